// Does an s_wakeup from ANOTHER wave of the workgroup shorten a victim wave's `s_nop N`? (tools/tap_bisect.sh, r04: specialised
// kernels of tap islands mis-rendered / faulted only when other waves executed s_wakeup; the victim's tapOut stores go through an
// SGPR pair written by v_readfirstlane, which needs 5 wait states before the VMEM instruction reads it — the compiler covers them
// with `s_nop 4`.) Wave 0 alternates between two buffers: v_readfirstlane -> SGPR pair, `s_nop 4` (or 5 x `s_nop 0`, or the
// address in VGPRs), global_store, read back. A store that used the STALE pair lands in the other buffer. Waves 1..7 execute
// s_wakeup (or s_nop) in a loop meanwhile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(512) void victim(float* a, float* b, unsigned iters, unsigned mode, unsigned* out) {
    __shared__ volatile unsigned stop;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (threadIdx.x == 0) stop = 0u;
    __syncthreads();
    if (wave == 0) {
        unsigned bad = 0;
        for (unsigned it = 1; it <= iters; ++it) {
            float* p = (it & 1u) ? a : b;
            const unsigned lo = (unsigned)(uintptr_t)p, hi = (unsigned)((uintptr_t)p >> 32);
            const float val = (float)it;
            const unsigned off = lane * 4u;
            if (mode & 4u) __builtin_amdgcn_s_sleep(1);
            if (mode & 16u) {          // address in VGPRs: no SGPR hazard at all
                float* q = p + lane;
                asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(q), "v"(val) : "memory");
            } else if (mode & 8u) {    // the same wait states as five one-cycle instructions
                asm volatile("v_readfirstlane_b32 s20, %0\n\tv_readfirstlane_b32 s21, %1\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
                             "global_store_dword %2, %3, s[20:21]\n\ts_waitcnt vmcnt(0)" :: "v"(lo), "v"(hi), "v"(off), "v"(val) : "s20", "s21", "memory");
            } else {
                asm volatile("v_readfirstlane_b32 s20, %0\n\tv_readfirstlane_b32 s21, %1\n\ts_nop 4\n\t"
                             "global_store_dword %2, %3, s[20:21]\n\ts_waitcnt vmcnt(0)" :: "v"(lo), "v"(hi), "v"(off), "v"(val) : "s20", "s21", "memory");
            }
            const float got = __builtin_nontemporal_load(p + lane);
            if (got != val) ++bad;
        }
        out[lane] = bad;
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) stop = 1u;
    } else {
        while (stop == 0u) {
            if (mode & 1u) { asm volatile("s_wakeup\n\ts_nop 3\n\ts_wakeup\n\ts_nop 7\n\ts_wakeup" ::: "memory"); }
            else if (mode & 2u) { __builtin_amdgcn_s_sleep(1); }
            else { asm volatile("s_nop 3\n\ts_nop 7" ::: "memory"); }
        }
    }
}

int main(int argc, char** argv) {
    const unsigned iters = argc > 1 ? (unsigned)atoi(argv[1]) : 1000000u;
    float *a, *b; unsigned* out;
    hipMalloc(&a, 4096); hipMalloc(&b, 4096); hipMalloc(&out, 256);
    const struct { unsigned mode; const char* what; } cases[] = {
        {0u, "others: s_nop          | victim: s_nop 4"},
        {1u, "others: s_wakeup       | victim: s_nop 4"},
        {5u, "others: s_wakeup       | victim: s_sleep 1, then s_nop 4"},
        {9u, "others: s_wakeup       | victim: 5 x s_nop 0"},
        {13u, "others: s_wakeup       | victim: s_sleep 1, then 5 x s_nop 0"},
        {17u, "others: s_wakeup       | victim: address in VGPRs"},
        {2u, "others: s_sleep        | victim: s_nop 4"},
        {3u, "others: s_wakeup (+nop)| victim: s_nop 4 (again)"},
    };
    for (auto& c : cases) {
        hipMemset(a, 0, 4096); hipMemset(b, 0, 4096); hipMemset(out, 0, 256);
        hipLaunchKernelGGL(victim, dim3(1), dim3(512), 0, 0, a, b, iters, c.mode, out);
        hipError_t e = hipDeviceSynchronize();
        unsigned h[64] = {};
        hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
        unsigned worst = 0; for (unsigned v : h) worst = v > worst ? v : worst;
        printf("mode %2u  %-58s misplaced stores: %u of %u  (%s)\n", c.mode, c.what, worst, iters, hipGetErrorString(e));
        fflush(stdout);
        if (e != hipSuccess) break;
    }
    return 0;
}
