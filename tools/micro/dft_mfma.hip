// DFT-as-GEMM on the f32 matrix cores vs the radix-4 Stockham FFT that conv.hip uses, for the 1024-point complex
// transforms of the `convolve` node ("run the experiment, adopt it or commit the numbers").
//
//   radix-4:  one workgroup (256 threads) per transform, 5 passes through LDS (conv.hip fft1024), ~51 kFLOP.
//   MFMA:     one WAVE per transform, N = 32 x 32:  X[c + 32 d] = sum_b F[b][d] W1024^(b c) sum_a F[c][a] x[32 a + b],
//             two complex 32x32x32 GEMMs on v_mfma_f32_32x32x2_f32 (4 real GEMMs each = 64 MFMAs per stage,
//             exact f32) with the twiddle multiply in between, ~0.52 MFLOP; four waves (one per SIMD) per workgroup.
// Both read a batch of transforms from global memory and write the spectra back; timed with HIP events.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/dft_mfma_bin tools/micro/dft_mfma.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef float c2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ c2 mk(float re, float im) { c2 v; v.x = re; v.y = im; return v; }
__device__ __forceinline__ c2 cmul(c2 a, c2 b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ c2 cadd(c2 a, c2 b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c2 csub(c2 a, c2 b) { return mk(a.x - b.x, a.y - b.y); }

__device__ c2 kW1024[1024];   // cis(-2 pi k / 1024)
__device__ c2 kF32[32 * 32];  // F[r][s] = cis(-2 pi r s / 32)

// ---- radix-4 Stockham, as in conv.hip ----
__device__ __forceinline__ void fft1024(c2* a, c2* b, const c2* W, uint32_t tid) {
#pragma unroll
    for (uint32_t s = 0; s < 5; ++s) {
        const uint32_t Ns = 1u << (2u * s), k = tid & (Ns - 1u), tw = 256u >> (2u * s);
        c2 v0 = a[tid], v1 = a[tid + 256u], v2 = a[tid + 512u], v3 = a[tid + 768u];
        if (s > 0) { v1 = cmul(v1, W[k * tw]); v2 = cmul(v2, W[2u * k * tw]); v3 = cmul(v3, W[3u * k * tw]); }
        const c2 a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), t = csub(v1, v3);
        const c2 a3 = mk(t.y, -t.x);
        const uint32_t idx = ((tid - k) << 2) + k;
        b[idx] = cadd(a0, a2); b[idx + Ns] = cadd(a1, a3); b[idx + 2u * Ns] = csub(a0, a2); b[idx + 3u * Ns] = csub(a1, a3);
        __syncthreads();
        c2* t2 = a; a = b; b = t2;
    }
}
__global__ __launch_bounds__(256) void k_radix4(const c2* in, c2* out, int perWg) {
    __shared__ c2 A[1024], B[1024], W[1024];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 1024; i += 256) W[i] = kW1024[i];
    for (int r = 0; r < perWg; ++r) {
        const size_t base = ((size_t)blockIdx.x * perWg + r) * 1024;
        __syncthreads();
        for (uint32_t i = tid; i < 1024; i += 256) A[i] = in[base + i];
        __syncthreads();
        fft1024(A, B, W, tid);          // 5 passes: result in B
        for (uint32_t i = tid; i < 1024; i += 256) out[base + i] = B[i];
    }
}

// ---- two 32x32 complex GEMMs on the f32 matrix cores, one wave per transform ----
__global__ __launch_bounds__(256) void k_mfma(const c2* in, c2* out, int perWave) {
    __shared__ float Fr[32 * 32], Fi[32 * 32];            // F[r][s]
    __shared__ c2 W[1024];
    __shared__ float Mr[4][32 * 33], Mi[4][32 * 33];      // per wave: the 32 x 32 matrix of the current stage (row stride 33: no bank conflicts)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    for (uint32_t i = tid; i < 1024; i += 256) { W[i] = kW1024[i]; Fr[i] = kF32[i].x; Fi[i] = kF32[i].y; }
    __syncthreads();
    float* mr = Mr[w]; float* mi = Mi[w];
    const uint32_t i31 = lane & 31u, hi = lane >> 5;       // A: row i31, k = k0 + hi;  B: k = k0 + hi, col i31
    for (int r = 0; r < perWave; ++r) {
        const size_t base = (((size_t)blockIdx.x * 4 + w) * perWave + r) * 1024;
        // Xm[a][b] = x[32 a + b] -> LDS
        for (uint32_t i = lane; i < 1024; i += 64) { const c2 v = in[base + i]; mr[(i >> 5) * 33 + (i & 31)] = v.x; mi[(i >> 5) * 33 + (i & 31)] = v.y; }
        __builtin_amdgcn_wave_barrier();
        // stage 1: Y[c][b] = sum_a F[c][a] Xm[a][b]
        f16v cr = {0}, ci = {0};
#pragma unroll
        for (uint32_t k0 = 0; k0 < 32; k0 += 2) {
            const float ar = Fr[i31 * 32 + k0 + hi], ai = Fi[i31 * 32 + k0 + hi];
            const float br = mr[(k0 + hi) * 33 + i31], bi = mi[(k0 + hi) * 33 + i31];
            cr = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, cr, 0, 0, 0);
            ci = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi, ci, 0, 0, 0);
            cr = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai, bi, cr, 0, 0, 0);
            ci = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, ci, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        // twiddle W1024^(b c), back to LDS as Y'[c][b]   (D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
#pragma unroll
        for (uint32_t reg = 0; reg < 16; ++reg) {
            const uint32_t c = (reg & 3u) + 8u * (reg >> 2) + 4u * hi, b = i31;
            const c2 y = cmul(mk(cr[reg], ci[reg]), W[(b * c) & 1023u]);
            mr[c * 33 + b] = y.x; mi[c * 33 + b] = y.y;
        }
        __builtin_amdgcn_wave_barrier();
        // stage 2: Z[c][d] = sum_b Y'[c][b] F[b][d]
        f16v zr = {0}, zi = {0};
#pragma unroll
        for (uint32_t k0 = 0; k0 < 32; k0 += 2) {
            const float ar = mr[i31 * 33 + k0 + hi], ai = mi[i31 * 33 + k0 + hi];
            const float br = Fr[(k0 + hi) * 32 + i31], bi = Fi[(k0 + hi) * 32 + i31];
            zr = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, zr, 0, 0, 0);
            zi = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi, zi, 0, 0, 0);
            zr = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai, bi, zr, 0, 0, 0);
            zi = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, zi, 0, 0, 0);
        }
        // X[c + 32 d] = Z[c][d]
#pragma unroll
        for (uint32_t reg = 0; reg < 16; ++reg) {
            const uint32_t c = (reg & 3u) + 8u * (reg >> 2) + 4u * hi, d = i31;
            out[base + c + 32u * d] = mk(zr[reg], zi[reg]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int main() {
    const int nWg = 1024, perWg = 64;                      // 65 536 transforms per launch
    const size_t n = (size_t)nWg * perWg;
    std::vector<c2> w(1024), f(1024), x(n * 1024);
    for (int k = 0; k < 1024; ++k) { const double a = -2.0 * M_PI * k / 1024.0; w[k] = c2{(float)cos(a), (float)sin(a)}; }
    for (int r = 0; r < 32; ++r) for (int s = 0; s < 32; ++s) { const double a = -2.0 * M_PI * ((r * s) % 32) / 32.0; f[r * 32 + s] = c2{(float)cos(a), (float)sin(a)}; }
    uint32_t st = 12345u;
    for (size_t i = 0; i < x.size(); ++i) { st = st * 1664525u + 1013904223u; const float a = (float)(st >> 8) / 8388608.0f - 1.0f; st = st * 1664525u + 1013904223u; x[i] = c2{a, (float)(st >> 8) / 8388608.0f - 1.0f}; }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(kW1024), w.data(), sizeof(c2) * 1024);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(kF32), f.data(), sizeof(c2) * 1024);
    c2 *dIn, *dA, *dB;
    (void)hipMalloc(&dIn, x.size() * sizeof(c2)); (void)hipMalloc(&dA, x.size() * sizeof(c2)); (void)hipMalloc(&dB, x.size() * sizeof(c2));
    (void)hipMemcpy(dIn, x.data(), x.size() * sizeof(c2), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto timeit = [&](auto launch) { launch(); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5.0; };
    const double msR = timeit([&] { hipLaunchKernelGGL(k_radix4, dim3(nWg), dim3(256), 0, 0, dIn, dA, perWg); });
    const double msM = timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(nWg), dim3(256), 0, 0, dIn, dB, perWg / 4); });
    // accuracy of both against a double-precision DFT of the first two transforms, and against each other over the batch
    std::vector<c2> ya(2048), yb(2048), allA(64 * 1024), allB(64 * 1024);
    (void)hipMemcpy(ya.data(), dA, 2048 * sizeof(c2), hipMemcpyDeviceToHost); (void)hipMemcpy(yb.data(), dB, 2048 * sizeof(c2), hipMemcpyDeviceToHost);
    double eA = 0, eB = 0, mag = 0;
    for (int t = 0; t < 2; ++t) for (int k = 0; k < 1024; ++k) {
        double re = 0, im = 0;
        for (int j = 0; j < 1024; ++j) { const double a = -2.0 * M_PI * ((k * j) % 1024) / 1024.0; re += x[t * 1024 + j].x * cos(a) - x[t * 1024 + j].y * sin(a); im += x[t * 1024 + j].x * sin(a) + x[t * 1024 + j].y * cos(a); }
        mag = fmax(mag, hypot(re, im));
        eA = fmax(eA, hypot(ya[t * 1024 + k].x - re, ya[t * 1024 + k].y - im)); eB = fmax(eB, hypot(yb[t * 1024 + k].x - re, yb[t * 1024 + k].y - im));
    }
    (void)hipMemcpy(allA.data(), dA + (n - 64) * 1024, allA.size() * sizeof(c2), hipMemcpyDeviceToHost); (void)hipMemcpy(allB.data(), dB + (n - 64) * 1024, allB.size() * sizeof(c2), hipMemcpyDeviceToHost);
    double dAB = 0; for (size_t i = 0; i < allA.size(); ++i) dAB = fmax(dAB, hypot(allA[i].x - allB[i].x, allA[i].y - allB[i].y));
    printf("1024-point complex DFT, %zu transforms per launch, 256 CUs; max |X| = %.1f\n", n, mag);
    printf("radix-4 Stockham in LDS (conv.hip), one workgroup per transform : %8.3f ms  = %6.1f ns per transform  (%5.1f M transforms/s)  max err vs f64 DFT %.2e\n", msR, 1e6 * msR / n, n / msR / 1e3, eA);
    printf("2 x complex 32x32x32 GEMM on v_mfma_f32_32x32x2_f32, one wave each : %8.3f ms  = %6.1f ns per transform  (%5.1f M transforms/s)  max err vs f64 DFT %.2e\n", msM, 1e6 * msM / n, n / msM / 1e3, eB);
    printf("MFMA / radix-4 time ratio %.2f;  the two agree to %.2e on the last 64 transforms\n", msM / msR, dAB);
    printf("arithmetic: radix-4 ~ 5 N log2 N = 51 kFLOP, GEMM form 2 x 4 x 2 x 32^3 = 524 kFLOP per transform (%.1f TFLOP/s achieved on the matrix cores)\n", 524288.0 * n / (msM * 1e-3) / 1e12);
    return 0;
}
