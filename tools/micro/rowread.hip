// What the long-partition sums' access pattern gets from the memory system (r06): every thread reads ROWS values of V bytes, one per
// row, rows 33 280 B apart (the stored spectra: 4160 complex floats), a wave covers 64 x V contiguous bytes of a row — exactly
// elemhip_convolve_long_mac's loads — with all loads of a round in flight before the first use. Grid: 1024 workgroups x 256 threads
// (one round on the chip, four waves per SIMD). Variants: V = 8 (dwordx2, what the kernel does) / 16 (dwordx4: two bins per lane),
// ROWS = 31 per round, three rounds; data footprint 46 MB (the C3 set's U + G) so that it comes from L2 / MALL / HBM as in the kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/rowread.hip -o tools/micro/rowread_bin && tools/micro/rowread_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int ROWS>
__global__ __launch_bounds__(256) void rowread(const T* __restrict__ base, T* out, size_t rowStride, int rowsPerNode, int rounds) {
    const int node = blockIdx.x, slab = blockIdx.y, run = blockIdx.z;
    const T* p = base + (size_t)node * rowsPerNode * rowStride + (size_t)slab * 256 + threadIdx.x;
    T acc = T{};
    for (int r = 0; r < rounds; ++r) {
        T v[ROWS];
        const int row0 = (run * 16 + r * 8) % (rowsPerNode - ROWS);
#pragma unroll
        for (int i = 0; i < ROWS; ++i) v[i] = p[(size_t)(row0 + i) * rowStride];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) acc += v[i];
    }
    out[((size_t)(node * gridDim.y + slab) * gridDim.z + run) * 256 + threadIdx.x] = acc;
}

template <typename T>
static void run(const char* name, int slabs) {
    const int nodes = 8, runs = 8, rowsPerNode = 175;
    const size_t rowStride = 33280 / sizeof(T);
    T* d; T* o;
    const size_t bytes = (size_t)nodes * rowsPerNode * 33280;
    hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
    hipMalloc(&o, (size_t)nodes * slabs * runs * 256 * sizeof(T));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((rowread<T, 31>), dim3(nodes, slabs, runs), dim3(256), 0, 0, d, o, rowStride, rowsPerNode, 3);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double req = (double)nodes * slabs * runs * 256 * 31 * 3 * sizeof(T);
        std::printf("%-28s %7.2f us per launch   requested %6.1f MB   %6.2f TB/s from the caches' point of view\n", name, 1e3 * ms / 20, req / 1e6, req / (ms / 20 * 1e-3) / 1e12);
    }
    hipFree(d); hipFree(o);
}

int main() {
    run<float2>("8 B per lane, 16 slabs", 16);
    run<float4>("16 B per lane, 8 slabs", 8);
    return 0;
}
