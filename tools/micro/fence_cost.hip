// fence_cost.hip — what the cache-maintenance halves of agent / system scope fences cost on gfx950, and what a flag hand-over between
// two workgroups costs with and without them (same XCD / different XCDs). Decides how resident.hip synchronises its workgroups.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/fence_cost.hip -o tools/micro/fence_cost_bin && tools/micro/fence_cost_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint64_t ticks() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xFu; }

// one workgroup: `iters` times { dirty `bytes` of device memory; op }, ticks per op kind
__global__ void k_ops(float* buf, uint32_t floats, uint32_t iters, uint64_t* out) {
    for (int kind = 0; kind < 6; ++kind) {
        __syncthreads();
        const uint64_t t0 = ticks();
        for (uint32_t it = 0; it < iters; ++it) {
            for (uint32_t i = threadIdx.x; i < floats; i += blockDim.x) buf[i] = (float)(it + kind);
            switch (kind) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("buffer_wbl2 sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_inv sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); break;
            case 5: asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)\n\tbuffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) out[kind] = ticks() - t0;
    }
}

// ping-pong: workgroups `a` and `b` of the grid hand a 4 KB payload back and forth `rounds` times through a flag.
// mode 0: agent-scope release / acquire (what the compiler emits: wbl2 sc1 ... inv sc1)
// mode 1: stores completed (vmcnt 0) -> relaxed agent atomic flag; reader: relaxed agent atomic poll -> buffer_inv sc1 (no write-back)
// mode 2: as 1 but payload loads are agent-scope relaxed atomics themselves (sc1 loads), no invalidate at all
// `bad` counts payload words that were stale.
__global__ void k_pingpong(uint32_t a, uint32_t b, uint32_t rounds, int mode, uint32_t* flag, uint32_t* payload, uint64_t* out, uint32_t* xcc) {
    if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
    if (blockIdx.x != a && blockIdx.x != b) return;
    const bool first = blockIdx.x == a;
    uint32_t bad = 0;
    __shared__ uint32_t go;
    const uint64_t t0 = ticks();
    for (uint32_t r = 1; r <= rounds; ++r) {
        const bool mine = first == ((r & 1u) != 0u);          // odd rounds: a writes, b reads
        if (mine) {
            for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) payload[i] = r * 7u + i;
            if (mode == 0) {
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(flag, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(flag, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (threadIdx.x == 0) {
                uint64_t tw = ticks();
                if (mode == 0) { while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r) { if (ticks() - tw > 100000000ull) break; } }
                else { while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r) { if (ticks() - tw > 100000000ull) break; } }
                go = 1;
            }
            __syncthreads();
            if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            else if (mode == 1) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) {
                const uint32_t v = mode == 2 ? __hip_atomic_load(payload + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : payload[i];
                bad += v != r * 7u + i;
            }
            __syncthreads();
        }
    }
    const uint64_t t1 = ticks();
    atomicAdd(&out[2], (unsigned long long)bad);
    if (threadIdx.x == 0 && first) { out[0] = t1 - t0; out[1] = rounds; }
}

int main() {
    float* buf; uint64_t* out; uint32_t *flag, *payload, *xcc;
    hipMalloc(&buf, 1 << 20); hipMalloc(&out, 64); hipMalloc(&flag, 64); hipMalloc(&payload, 4096); hipMalloc(&xcc, 64 * 4);
    uint64_t h[8];
    const char* names[6] = {"vmcnt(0) only", "wbl2 sc1 (agent release)", "wbl2 sc0 sc1 (system release)", "inv sc1 (agent acquire)", "inv sc0 sc1 (system acquire)", "wbl2 sc1 + inv sc1"};
    for (uint32_t floats : {256u, 1024u, 65536u}) {
        hipLaunchKernelGGL(k_ops, dim3(1), dim3(256), 0, 0, buf, floats, 200u, out);
        hipDeviceSynchronize();
        hipMemcpy(h, out, 48, hipMemcpyDeviceToHost);
        for (int k = 0; k < 6; ++k) std::printf("dirty %7u B then %-32s %8.2f us per iteration\n", floats * 4, names[k], 0.01 * (double)h[k] / 200.0);
    }
    uint32_t hx[64];
    for (int mode = 0; mode < 3; ++mode) for (uint32_t b : {8u, 1u, 4u}) {
        hipMemset(flag, 0, 64); hipMemset(out, 0, 64);
        hipLaunchKernelGGL(k_pingpong, dim3(16), dim3(256), 0, 0, 0u, b, 2000u, mode, flag, payload, out, xcc);
        hipDeviceSynchronize();
        hipMemcpy(h, out, 24, hipMemcpyDeviceToHost); hipMemcpy(hx, xcc, 64, hipMemcpyDeviceToHost);
        std::printf("ping-pong mode %d workgroups 0 (xcc %u) <-> %u (xcc %u): %.2f us per hand-over of 4 KB, stale words %llu\n", mode, hx[0], b, hx[b],
                    0.01 * (double)h[0] / (double)h[1], (unsigned long long)h[2]);
    }
    std::printf("xcc of workgroups 0..15:"); for (int i = 0; i < 16; ++i) std::printf(" %u", hx[i]); std::printf("\n");
    return 0;
}
