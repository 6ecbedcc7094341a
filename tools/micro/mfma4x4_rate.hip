// Issue rate of v_mfma_f32_4x4x1_16B_f32 (conv.hip partition MAC): 10 independent accumulators per wave, as in the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void rate(float* out, int iters, float a0, float b0) {
    f4v d[NACC];
    for (int i = 0; i < NACC; ++i) d[i] = f4v{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) d[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < NACC; ++i) s += d[i][0] + d[i][1] + d[i][2] + d[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[4096] = t1 - t0;
}
template <int NACC> void run(float* d, int threads) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(rate<NACC>, dim3(1), dim3(threads), 0, 0, d, iters, 1.0f, 0.5f); hipDeviceSynchronize(); }
    long long cyc; hipMemcpy(&cyc, (char*)d + 4096 * 8, 8, hipMemcpyDeviceToHost);
    const double per = (double)cyc / (iters * 2.0 * NACC);     // s_memtime ticks at 100 MHz: scale below
    printf("%d accumulators, %d waves per SIMD: %.2f clock64 ticks per MFMA per wave\n", NACC, threads / 256, per);
}
int main() {
    float* d; hipMalloc(&d, 1 << 20);
    run<10>(d, 256); run<10>(d, 512); run<10>(d, 1024); run<2>(d, 256); run<4>(d, 256); run<1>(d, 256);
    // calibration: a dependent chain of v_fma (4 cycles each on a lone wave... measured 4.07 in r02) is not repeated here;
    // wall-clock per launch instead:
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipEventRecord(e0); hipLaunchKernelGGL(rate<10>, dim3(1024), dim3(256), 0, 0, d, iters, 1.0f, 0.5f); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 1024.0 * 4 * iters * 20.0 * 512.0;
    printf("1024 workgroups x 4 waves (1 wave per SIMD): %.3f ms, %.1f TFLOP/s (peak 157)\n", ms, flops / (ms * 1e-3) / 1e12);
    hipEventRecord(e0); hipLaunchKernelGGL(rate<10>, dim3(1024), dim3(512), 0, 0, d, iters, 1.0f, 0.5f); hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    printf("1024 workgroups x 8 waves (2 per SIMD): %.3f ms, %.1f TFLOP/s\n", ms, 2 * flops / (ms * 1e-3) / 1e12);
    return 0;
}
