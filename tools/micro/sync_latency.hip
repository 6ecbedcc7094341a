// sync_latency.hip — how long after a kernel has finished does the host know? hipStreamSynchronize against a word the kernel writes to
// mapped host memory as its last act, for kernels of 5 us .. 5 ms. (Decides where elemhip waits by polling: engine.cpp `sync_poll`.)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/sync_latency.hip -o tools/micro/sync_latency_bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_spin(uint64_t ticks, uint32_t* flag, uint32_t value, uint64_t* devEnd) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (flag) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    if (devEnd) *devEnd = __builtin_amdgcn_s_memrealtime();
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint32_t* hflag; hipHostMalloc((void**)&hflag, 64, hipHostMallocMapped | hipHostMallocCoherent); *hflag = 0;
    uint32_t* dflag; hipHostGetDevicePointer((void**)&dflag, hflag, 0);
    uint32_t seq = 0;
    for (int warm = 0; warm < 50; ++warm) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 100ull, (uint32_t*)nullptr, 0u, (uint64_t*)nullptr); hipStreamSynchronize(s); }
    std::printf("kernel us | launch -> hipStreamSynchronize returns (p50 / p99) | launch -> polled word seen (p50 / p99) | difference p50\n");
    for (double us : {5.0, 20.0, 50.0, 100.0, 300.0, 1000.0, 5000.0}) {
        const int reps = us >= 1000.0 ? 200 : 1000;
        std::vector<double> a, b;
        for (int r = 0; r < reps; ++r) {
            double t0 = now_us();
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, (uint64_t)(us * 100.0), (uint32_t*)nullptr, 0u, (uint64_t*)nullptr);
            hipStreamSynchronize(s);
            a.push_back(now_us() - t0);
            const uint32_t want = ++seq;
            t0 = now_us();
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, (uint64_t)(us * 100.0), dflag, want, (uint64_t*)nullptr);
            while (__atomic_load_n(hflag, __ATOMIC_ACQUIRE) != want) __builtin_ia32_pause();
            b.push_back(now_us() - t0);
            if ((r & 63) == 63) hipStreamSynchronize(s);
        }
        hipStreamSynchronize(s);
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        auto p = [](std::vector<double>& v, double q) { return v[std::min(v.size() - 1, (size_t)(q * v.size()))]; };
        std::printf("%8.0f | %9.1f / %9.1f | %9.1f / %9.1f | %6.1f\n", us, p(a, 0.5), p(a, 0.99), p(b, 0.5), p(b, 0.99), p(a, 0.5) - p(b, 0.5));
    }
    return 0;
}
