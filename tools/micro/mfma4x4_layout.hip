// Register layout of v_mfma_f32_4x4x1_16B_f32 on gfx950 (conv.hip batch_mac_tile_mfma relies on it): A = lane 4 b + i, B = lane
// 4 b + j, D[i][j] of block b = register i of lane 4 b + j. A = 1 + i + 10 b, B = 100 (1 + j): D = A * B identifies (i, j, b).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int lane = threadIdx.x, b = lane >> 2, q = lane & 3;
    const float a = 1.0f + q + 10.0f * b, bb = 100.0f * (1 + q);
    f4v d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bb, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}
int main() {
    float* d; hipMalloc(&d, 1024); float h[256];
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d); hipDeviceSynchronize();
    hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int okA = 1, okB = 1;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
        const int b = lane >> 2, q = lane & 3;
        const float expectA = (1.0f + r + 10.0f * b) * 100.0f * (1 + q);   // register = A index i, lane = B index j
        const float expectB = (1.0f + q + 10.0f * b) * 100.0f * (1 + r);   // register = B index j, lane = A index i
        okA &= h[lane * 4 + r] == expectA; okB &= h[lane * 4 + r] == expectB;
    }
    printf("layout: register = A row i, lane = 4 b + j : %s\nlayout: register = B column j, lane = 4 b + i : %s\n", okA ? "YES" : "no", okB ? "YES" : "no");
    printf("lane 5 (block 1, q 1): %g %g %g %g\n", h[20], h[21], h[22], h[23]);
    return 0;
}
