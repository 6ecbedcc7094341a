// fft4096.h — the transform core of the long-partition convolver (conv_long.inc): a 4096-point complex FFT for 256 threads
// (Stockham radix-16, three passes through one padded LDS buffer, a 16-point DFT per thread in registers) and the split steps
// that make it the 8192-point transform of REAL data, forward and inverse.
//
// Written against plain pointers so that the same code runs on the device (buf = LDS, one call per thread, __syncthreads between
// the phases) and on the host (tests/native/fft4096_host.cpp emulates the 256 threads phase by phase and checks the result
// against a double-precision DFT): the arithmetic that decides C3's parity is testable without a GPU.
//
// Conventions. w = exp(-2 pi i / 8192); `W` is the table w^j, j < 8192 (rounded from double on the host).
//   real forward   U[k] = 2 X[k],  k = 0 .. 4096, X = DFT_8192 of 8192 real samples x (packed as z[n] = x[2n] + i x[2n+1])
//   real inverse   out[n] = 8192 * IDFT_8192(Y)[n] for a Hermitian spectrum given as Y[k], k = 0 .. 4096
// so a spectrum product G * U comes back as 16384 x the circular convolution: the IR spectra carry the 1/16384 (engine.cpp).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define LFFT_FD __host__ __device__ __forceinline__
#else
#define LFFT_FD inline
#endif

namespace lfft {

typedef float c2 __attribute__((ext_vector_type(2)));   // x = re, y = im

constexpr uint32_t M = 4096;            // complex transform length
constexpr uint32_t N = 8192;            // real transform length
constexpr uint32_t kBins = M + 1;       // spectrum bins of a real transform
constexpr uint32_t kRow = 4160;         // row stride of a stored spectrum (c2 elements): 65 x 64
constexpr uint32_t kBuf = M + M / 16;   // padded LDS buffer (c2 elements)
constexpr uint32_t kThreads = 256;

LFFT_FD c2 mk(float re, float im) { c2 v; v.x = re; v.y = im; return v; }
// complex product with two fused multiply-adds (r06: the library is compiled with -ffp-contract=off for the bit-exact node kernels, which
// left every product here as mul, mul, sub / mul, mul, add — the transforms have a TOLERANCE (1e-6 on |y| <~ 1), not a bit pattern, to
// meet, and the fused form is both the more accurate and a third fewer instructions of kernels that are bound by instruction issue)
LFFT_FD c2 cmul(c2 a, c2 b) { return mk(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x)); }
LFFT_FD c2 cconj(c2 a) { return mk(a.x, -a.y); }
LFFT_FD c2 mul_mi(c2 a) { return mk(a.y, -a.x); }      // a * -i
LFFT_FD c2 mul_pi(c2 a) { return mk(-a.y, a.x); }      // a * +i
// element i of the transform sits at pad(i): one spare element per 16 keeps the stride-16 writes of pass 0 off a single bank pair
LFFT_FD uint32_t pad(uint32_t i) { return i + (i >> 4); }

// radix-4 butterfly, forward sign: (y0, y1, y2, y3) = DFT_4(x0, x1, x2, x3)
LFFT_FD void bfly4(c2& x0, c2& x1, c2& x2, c2& x3) {
    const c2 a0 = x0 + x2, a1 = x0 - x2, a2 = x1 + x3, a3 = mul_mi(x1 - x3);
    x0 = a0 + a2; x1 = a1 + a3; x2 = a0 - a2; x3 = a1 - a3;
}

// 16-point DFT in registers, natural order in and out. n = n1 + 4 n2, m = 4 m1 + m2:
//   X[4 m1 + m2] = sum_n1 W4^(n1 m1) [ W16^(n1 m2) sum_n2 x[n1 + 4 n2] W4^(n2 m2) ]
LFFT_FD void dft16(c2 (&v)[16]) {
    // W16^j = exp(-2 pi i j / 16), j = 1, 2, 3, 6, 9 (j = 4: -i, j = 0: 1); the other products used below are among these
    const float c1 = 0.92387953251128673848f, s1 = 0.38268343236508978178f, r2 = 0.70710678118654752440f;
    const c2 w1 = mk(c1, -s1), w2 = mk(r2, -r2), w3 = mk(s1, -c1), w6 = mk(-r2, -r2), w9 = mk(-c1, s1);
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) bfly4(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);     // v[n1 + 4 m2] = A[n1][m2]
    // twiddles W16^(n1 m2)
    v[1 + 4] = cmul(v[1 + 4], w1); v[1 + 8] = cmul(v[1 + 8], w2); v[1 + 12] = cmul(v[1 + 12], w3);
    v[2 + 4] = cmul(v[2 + 4], w2); v[2 + 8] = mul_mi(v[2 + 8]);   v[2 + 12] = cmul(v[2 + 12], w6);
    v[3 + 4] = cmul(v[3 + 4], w3); v[3 + 8] = cmul(v[3 + 8], w6); v[3 + 12] = cmul(v[3 + 12], w9);
    // second radix-4 over n1 for each m2: v[0 + 4 m2], v[1 + 4 m2], v[2 + 4 m2], v[3 + 4 m2] -> X[4 m1 + m2], m1 = 0..3
    c2 o[16];
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) {
        c2 y0 = v[4 * m2], y1 = v[4 * m2 + 1], y2 = v[4 * m2 + 2], y3 = v[4 * m2 + 3];
        bfly4(y0, y1, y2, y3);
        o[m2] = y0; o[4 + m2] = y1; o[8 + m2] = y2; o[12 + m2] = y3;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = o[i];
}

// ---- one Stockham radix-16 pass of thread `tid` (pass s = 0, 1, 2; Ns = 16^s), in three phases with a barrier between them ----
LFFT_FD void pass_read(const c2* buf, uint32_t tid, c2 (&v)[16]) {
    const uint32_t p0 = pad(tid);                                   // pad(tid + 256 r) = pad(tid) + 272 r
#pragma unroll
    for (uint32_t r = 0; r < 16; ++r) v[r] = buf[p0 + r * (kThreads + kThreads / 16u)];
}
// v[r] *= exp(-2 pi i r k / (16 Ns)), k = tid mod Ns. The factors come from three small tables (LDS on the device: 3 KB; the first
// build read them from the 8192-entry table in global memory, 15 scattered 8-byte loads per thread and pass):
//   pass 1 (Ns = 16):  exp(-2 pi i j / 256),  j = r k <= 225            -> T256[j]
//   pass 2 (Ns = 256): exp(-2 pi i j / 4096), j = r k <= 3825 = 64 h + l -> A64[h] * B64[l]   (one extra rounding: ~1e-7 relative)
constexpr uint32_t kTabT = 0, kTabA = 256, kTabB = 320, kTabSize = 384;
// table entry i (0 .. kTabSize) as an index into the 8192-entry table w^j
LFFT_FD uint32_t tab_source(uint32_t i) { return i < kTabA ? 32u * i : (i < kTabB ? 128u * (i - kTabA) : 2u * (i - kTabB)); }
// (the pass number as a template argument: strides, masks and LDS offsets fold into the instructions — with a run-time `s` the r05
//  kernels spent about a fifth of their instructions on index arithmetic; the run-time forms below dispatch to these)
template <uint32_t S, class TP>
LFFT_FD void pass_twiddle_s(c2 (&v)[16], uint32_t tid, TP tab) {
    if (S == 0u) return;
    if (S == 1u) {
        const uint32_t k = tid & 15u;
#pragma unroll
        for (uint32_t r = 1; r < 16; ++r) v[r] = cmul(v[r], tab[kTabT + r * k]);
    } else {
        const uint32_t k = tid & 255u;
#pragma unroll
        for (uint32_t r = 1; r < 16; ++r) { const uint32_t j = r * k; v[r] = cmul(v[r], cmul(tab[kTabA + (j >> 6)], tab[kTabB + (j & 63u)])); }
    }
}
template <uint32_t S>
LFFT_FD void pass_write_s(c2* buf, uint32_t tid, const c2 (&v)[16]) {
    constexpr uint32_t Ns = S == 0u ? 1u : (S == 1u ? 16u : 256u);
    const uint32_t k = tid & (Ns - 1u);
    const uint32_t base = ((tid - k) << 4) + k;
    // pad(base + r Ns) = pad(base) + r (Ns + Ns / 16) for Ns >= 16 (r Ns is a multiple of 16), = pad(base) + r for Ns = 1 (base is)
    const uint32_t p0 = pad(base);
    constexpr uint32_t step = Ns == 1u ? 1u : Ns + Ns / 16u;
#pragma unroll
    for (uint32_t r = 0; r < 16; ++r) buf[p0 + r * step] = v[r];
}
template <class TP>
LFFT_FD void pass_twiddle(c2 (&v)[16], uint32_t tid, uint32_t s, TP tab) {
    if (s == 1u) pass_twiddle_s<1u>(v, tid, tab); else if (s == 2u) pass_twiddle_s<2u>(v, tid, tab);
}
LFFT_FD void pass_write(c2* buf, uint32_t tid, uint32_t s, const c2 (&v)[16]) {
    if (s == 0u) pass_write_s<0u>(buf, tid, v); else if (s == 1u) pass_write_s<1u>(buf, tid, v); else pass_write_s<2u>(buf, tid, v);
}

// ---- real forward: the two spectrum bins k and 4096 - k from the complex transform Z of z[n] = x[2n] + i x[2n + 1] ----
// U[k] = (Z[k] + conj Z[M-k]) - i w^k (Z[k] - conj Z[M-k]);  k in [0, 2048]; for k = 0 the partner bin is U[4096].
// One complex product serves both bins: with t = w^k (Z[k] - conj Z[M-k]) the partner is U[M-k] = conj((Z[k] + conj Z[M-k]) + i t)
// (w^(M-k) = -conj(w^k); r06 — the first build formed the second product separately).
template <class WP>
LFFT_FD void split_forward(const c2* buf, uint32_t k, WP W, c2& Uk, c2& Umk) {
    const c2 A = buf[pad(k)], B = buf[pad((M - k) & (M - 1u))];
    const c2 s = A + cconj(B), it = mul_pi(cmul(W[k], A - cconj(B)));
    Uk = s - it;
    Umk = cconj(s + it);
}
// ---- real inverse: the transform input conj(Zt[k]), conj(Zt[M-k]) from Y[k], Y[M-k] ----
// Zt[k] = (Y[k] + conj Y[M-k]) + i conj(w^k) (Y[k] - conj Y[M-k]); the inverse runs as conj(FFT(conj Zt)): out[2n] = Re F[n], out[2n+1] = -Im F[n]
// (again one product: conj(Zt[M-k]) = (Y[k] + conj Y[M-k]) - i t, t = conj(w^k) (Y[k] - conj Y[M-k]))
template <class WP>
LFFT_FD void split_inverse(c2 Yk, c2 Ymk, uint32_t k, WP W, c2& Zk, c2& Zmk) {
    const c2 s = Yk + cconj(Ymk), it = mul_pi(cmul(cconj(W[k]), Yk - cconj(Ymk)));
    Zk = cconj(s + it);
    Zmk = s - it;
}

} // namespace lfft
