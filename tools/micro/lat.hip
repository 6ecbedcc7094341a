// Micro-benchmarks for the single-wave latency regime the island interpreter lives in.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ float lds[];
__global__ void lds_chase(unsigned* out, int n, int mode) {
    unsigned* l = (unsigned*)lds;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) l[i] = (i * 17 + 5) & 4095;
    __syncthreads();
    unsigned idx = (mode & 1) ? threadIdx.x : 0;
    long long t0 = clock64();
    if (mode & 2) {
        for (int i = 0; i < n; ++i) idx = __builtin_amdgcn_readfirstlane(l[idx]);   // uniform chase through SGPR
    } else {
        for (int i = 0; i < n; ++i) idx = l[idx];
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = idx; out[1] = (unsigned)(t1 - t0); }
}
__global__ void valu_chain(float* out, int n, float a, float b) {
    float x = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) { x = x * a; x = x + b; x = x * a; x = x + b; x = x * a; x = x + b; x = x * a; x = x + b; }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) ((unsigned*)out)[64] = (unsigned)(t1 - t0);
}
__global__ void valu_indep(float* out, int n, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) { x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b; x4 = x4 * a + b; x5 = x5 * a + b; x6 = x6 * a + b; x7 = x7 * a + b; }
    long long t1 = clock64();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) ((unsigned*)out)[64] = (unsigned)(t1 - t0);
}
__global__ void clock_cost(unsigned* out, int n) {
    long long t0 = clock64(); long long acc = 0;
    for (int i = 0; i < n; ++i) acc += clock64();
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = (unsigned)acc; out[1] = (unsigned)(t1 - t0); }
}
int main() {
    unsigned* d; hipMalloc(&d, 4096);
    unsigned h[128];
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(lds_chase, dim3(1), dim3(64), 16384, 0, d, 1000, mode); hipDeviceSynchronize(); }
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("lds_chase mode %d (bit0 per-lane idx, bit1 readfirstlane): %.1f cycles/read\n", mode, h[1] / 1000.0);
    }
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(valu_chain, dim3(1), dim3(64), 0, 0, (float*)d, 1000, 1.0001f, 0.5f); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost);
    printf("valu dependent chain (contract default): %.2f cycles/op (8 ops/iter)\n", h[64] / 8000.0);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(valu_indep, dim3(1), dim3(64), 0, 0, (float*)d, 1000, 1.0001f, 0.5f); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost);
    printf("valu 8 independent fma streams: %.2f cycles/fma\n", h[64] / 8000.0);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(clock_cost, dim3(1), dim3(64), 0, 0, d, 1000); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("clock64(): %.1f cycles/call\n", h[1] / 1000.0);
    // 4 waves on one CU all running the dependent chain
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(valu_chain, dim3(1), dim3(256), 0, 0, (float*)d, 1000, 1.0001f, 0.5f); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost);
    printf("valu dependent chain, 4 waves/WG: %.2f cycles/op\n", h[64] / 8000.0);
    return 0;
}
