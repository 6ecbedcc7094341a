// conv.hip — the `convolve` node (reference: wasm/Convolve.h:23-92 -> fftconvolver::TwoStageFFTConvolver,
// an un-vendored third-party library) as a uniformly partitioned frequency-domain convolver for gfx950.
//
// What the reference computes is a zero-latency linear convolution of the node's input with channel 0
// of a shared resource (trailing |h| < 1e-6 dropped), restarted from silence whenever `path` is set.
// The reference's head(512) / tail0(512, one 4096 period late) / tail(4096, two periods late) schedule
// exists to spread CPU work; on the GPU the same sum is evaluated with ONE partition size (512):
//
//     Y_b = H_0 X_b + H_1 X_{b-1} + sum_{p>=2} H_p X_{b-p}          (1024-point spectra, 512 bins + packed Nyquist)
//
//   * main workgroup (one per node, the latency path): FFT of the current input block, the two newest
//     products, + the pre-multiplied older-partition sum, inverse FFT, overlap-add.
//   * helper workgroups (8 bin groups x S slices per node, off the latency path): during block b they
//     build the older-partition sum of block b+1 from spectra that are already final; every IR spectrum
//     is read once per block, coalesced, by exactly one wave.
// A call of fewer than 512 frames re-transforms the partially filled input block exactly like the
// reference does (FFTConvolver::process with a partial _inputBuffer); a call that straddles two input
// blocks computes the second block's older-partition sum in the main workgroup (slow path, odd sizes only).
// Results differ from the reference's Ooura-FFT arithmetic by float rounding (~1e-7 relative); the parity
// bar for this node is 1e-6 abs on |y| <~ 1 (SURVEY.md 8(d) C3 note).
#include <hip/hip_runtime.h>

#include "device.h"
#include "launch.h"

using namespace elemhip;

namespace {

typedef __attribute__((address_space(1))) float* gfp;
typedef __attribute__((address_space(1))) const float* gcfp;
typedef __attribute__((address_space(1))) uint32_t* gup;
typedef __attribute__((address_space(1))) const uint32_t* gcup;
typedef float c2 __attribute__((ext_vector_type(2)));   // complex float (x = re, y = im)
typedef __attribute__((address_space(1))) c2* gf2p;
typedef __attribute__((address_space(1))) const c2* gcf2p;
__device__ __forceinline__ c2 mk(float re, float im) { c2 v; v.x = re; v.y = im; return v; }

__device__ c2 kTwiddle[conv::kFft];     // cis(-2 pi k / 1024), rounded from double on the host

__device__ __forceinline__ c2 cmul(c2 a, c2 b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// spectrum product; bin 0 carries two real bins (DC, Nyquist)
__device__ __forceinline__ c2 smul(c2 h, c2 x, bool packed) {
    return packed ? mk(h.x * x.x, h.y * x.y) : cmul(h, x);
}
__device__ __forceinline__ c2 cadd(c2 a, c2 b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c2 csub(c2 a, c2 b) { return mk(a.x - b.x, a.y - b.y); }

// 1024-point forward FFT, Stockham radix-4 autosort, 256 threads, 5 passes LDS a -> b -> a ...; the
// result lands in `b` (odd number of passes), in natural order.
__device__ __forceinline__ void fft1024(c2* a, c2* b, const c2* W, uint32_t tid) {
#pragma unroll
    for (uint32_t s = 0; s < 5; ++s) {
        const uint32_t Ns = 1u << (2u * s), k = tid & (Ns - 1u), tw = 256u >> (2u * s);
        c2 v0 = a[tid], v1 = a[tid + 256u], v2 = a[tid + 512u], v3 = a[tid + 768u];
        if (s > 0) { v1 = cmul(v1, W[k * tw]); v2 = cmul(v2, W[2u * k * tw]); v3 = cmul(v3, W[3u * k * tw]); }
        const c2 a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), t = csub(v1, v3);
        const c2 a3 = mk(t.y, -t.x);                      // (v1 - v3) * -i
        const uint32_t idx = ((tid - k) << 2) + k;
        b[idx] = cadd(a0, a2); b[idx + Ns] = cadd(a1, a3); b[idx + 2u * Ns] = csub(a0, a2); b[idx + 3u * Ns] = csub(a1, a3);
        __syncthreads();
        c2* t2 = a; a = b; b = t2;
    }
}

__device__ __forceinline__ bool root_running(gcup recs, uint32_t rootRec, uint32_t numOut) {   // Core.h:28-31, GraphRenderSequence.h:218-219
    gcup r = recs + rootRec * kRecDwords;
    const float tg = __uint_as_float(r[rec::ROOT_TARGET]), g = __uint_as_float(r[rec::ROOT_GAIN]);
    const int ch = (int)r[rec::ROOT_CHANNEL];
    return (tg > 0.5f || !(fabsf(tg - g) <= 1e-6f)) && ch >= 0 && (uint32_t)ch < numOut;
}

struct State {
    gup hdr; gcf2p H; gf2p X; gf2p pre; gf2p preSlow; gfp inbuf; gfp overlap;
    uint32_t P, S;
};
__device__ __forceinline__ State state_of(gup base) {
    State st;
    st.hdr = base; st.P = base[conv::H_P]; st.S = base[conv::H_S];
    gf2p f = (gf2p)(base + conv::kHeaderDwords);
    st.H = (gcf2p)f;
    st.X = f + (size_t)st.P * 512u;
    st.pre = st.X + (size_t)st.P * 512u;
    st.preSlow = st.pre + (size_t)2u * st.S * 512u;
    st.inbuf = (gfp)(st.preSlow + 512u);
    st.overlap = st.inbuf + 512u;
    return st;
}

// which kernel family renders this node's share of a launch set of `batch` blocks (conv_long.inc; the same answer in every kernel)
__device__ __forceinline__ bool conv_use_long(const State& st, uint32_t batch, uint32_t longMode) {
    return longMode != 0u && st.hdr[conv::H_Q] != 0u && batch >= 8u && (batch & 7u) == 0u;
}

// sum_{p=2}^{P-1} H_p[k] X_{b-p}[k] restricted to partitions p = 2 + q, q = q0, q0 + dq, ...
__device__ __forceinline__ c2 older_sum(const State& st, uint32_t b, uint32_t k, uint32_t q0, uint32_t dq) {
    const uint32_t P = st.P, bm = b % P;
    const bool packed = k == 0u;
    c2 acc = mk(0.0f, 0.0f);
    uint32_t q = q0;
    for (; q + 3u * dq + 2u < P; q += 4u * dq) {      // four independent load pairs in flight
        c2 h[4], x[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint32_t p = 2u + q + u * dq;
            const uint32_t slot = bm >= p ? bm - p : bm + P - p;
            h[u] = st.H[(size_t)p * 512u + k];
            x[u] = st.X[(size_t)slot * 512u + k];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) acc = cadd(acc, smul(h[u], x[u], packed));
    }
    for (; q + 2u < P; q += dq) {
        const uint32_t p = 2u + q;
        const uint32_t slot = bm >= p ? bm - p : bm + P - p;
        acc = cadd(acc, smul(st.H[(size_t)p * 512u + k], st.X[(size_t)slot * 512u + k], packed));
    }
    return acc;
}

__device__ void conv_main(const ConvDesc d, gup recs, gfp hbm, const Globals* g, c2* A, c2* B, c2* W, c2* Xk) {
    const uint32_t tid = threadIdx.x;
    const uint32_t n = g->numSamples, stride = g->blockStride;
    gfp out = hbm + (size_t)d.outHbm * stride;
    gcup r = (gcup)(recs + d.rec * kRecDwords);
    const uint64_t sp = (uint64_t)r[rec::CONV_STATE] | ((uint64_t)r[rec::CONV_STATE + 1] << 32);
    uint32_t inKind = d.inKind, inBuf = d.inIdx;
    if (inKind == 3u) { inKind = g->numIn > 0 ? 1u : 0u; inBuf = 0u; }   // leaf: host channel 0 (arena buffer 0)
    if (inKind == 5u) {                                                  // folded `in` node (Math.h:92-126): its channel, or silence
        const uint32_t ch = ((gcup)recs)[d.inIdx * kRecDwords + rec::P0];
        if (ch < g->numIn) { inKind = 1u; inBuf = ch; } else inKind = 4u;
    }
    // folded root (Core.h:66-78 / GainFade.h:56-72): gain ramp of this block
    float rootG = 1.0f, rootT = 1.0f, rootStep = 0.0f;
    if (d.fuseRootRec != kNone) {
        gcup rr = (gcup)(recs + d.fuseRootRec * kRecDwords);
        rootG = __uint_as_float(rr[rec::ROOT_GAIN]); rootT = __uint_as_float(rr[rec::ROOT_TARGET]); rootStep = __uint_as_float(rr[rec::ROOT_STEP]);
    }
    auto rootGain = [&](uint32_t frame) {
        if (d.fuseRootRec == kNone) return 1.0f;
        if (rootG == rootT) return rootT;
        const float v = rootG + rootStep * (float)(int)frame;
        return (v < 0.0f) ? 0.0f : ((1.0f < v) ? 1.0f : v);
    };
    if (sp == 0ull || inKind == 0u) {                                 // Convolve.h:70-71
        for (uint32_t i = tid; i < n; i += 256u) out[i] = 0.0f;
        return;
    }
    const State st = state_of((gup)reinterpret_cast<uint32_t*>(sp));
    if (st.P == 0u) {                                                 // empty / all-below-threshold IR: FFTConvolver with no segments
        for (uint32_t i = tid; i < n; i += 256u) out[i] = 0.0f;
        return;
    }
    gcfp in = (gcfp)(hbm + (size_t)(inKind == 1u ? inBuf : 0u) * stride);
    const float cval = inKind == 2u ? __uint_as_float(((gcup)recs)[d.inIdx * kRecDwords + rec::P0]) : 0.0f;
    for (uint32_t i = tid; i < conv::kFft; i += 256u) W[i] = kTwiddle[i];

    uint32_t fill = st.hdr[conv::H_FILL], blk = st.hdr[conv::H_BLK];
    const uint32_t blk0 = blk;
    uint32_t processed = 0;
    while (processed < n) {
        const uint32_t chunk = min(n - processed, 512u - fill);
        // where this block's older-partition sum comes from (read before the barriers below: the slow path updates the header)
        const bool haveHelpers = st.hdr[conv::H_PREVALID0 + (blk & 1u)] == blk;
        const bool haveSlow = st.hdr[conv::H_PRESLOW_FOR] == blk;
        // 1. time-domain input block, zero-padded to 1024
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q) {
            const uint32_t i = tid + 256u * q;
            float v = 0.0f;
            if (i < fill) v = st.inbuf[i];
            else if (i < fill + chunk) {
                const uint32_t j = processed + i - fill;
                v = inKind == 1u ? in[j] : cval;                      // inKind 4: cval == 0
                st.inbuf[i] = v;
            }
            A[i] = mk(v, 0.0f);
            A[i + 512u] = mk(0.0f, 0.0f);
        }
        __syncthreads();
        fft1024(A, B, W, tid);                                        // -> B
        // 2. keep the block spectrum (packed), form Y, lay out conj(Y) for the inverse transform
        const uint32_t prevSlot = (blk % st.P) == 0u ? st.P - 1u : (blk % st.P) - 1u;
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q) {
            const uint32_t k = tid + 256u * q;
            const bool packed = k == 0u;
            const c2 x = packed ? mk(B[0].x, B[512].x) : B[k];
            Xk[k] = x;
            c2 acc = mk(0.0f, 0.0f);
            if (st.P > 2u) {
                if (haveHelpers) {
                    gcf2p ps = (gcf2p)st.pre + (size_t)(blk & 1u) * st.S * 512u + k;
                    for (uint32_t s = 0; s < st.S; ++s) acc = cadd(acc, ps[(size_t)s * 512u]);
                } else if (haveSlow) {
                    acc = st.preSlow[k];
                } else {
                    acc = older_sum(st, blk, k, 0u, 1u);
                    st.preSlow[k] = acc;
                }
            }
            if (st.P > 1u) acc = cadd(acc, smul(st.H[512u + k], st.X[(size_t)prevSlot * 512u + k], packed));
            acc = cadd(acc, smul(st.H[k], x, packed));
            if (packed) { A[0] = mk(acc.x, 0.0f); A[512] = mk(acc.y, 0.0f); }
            else { A[k] = mk(acc.x, -acc.y); A[1024u - k] = acc; }
        }
        if (!haveHelpers && !haveSlow && st.P > 2u && tid == 0u) st.hdr[conv::H_PRESLOW_FOR] = blk;
        __syncthreads();
        fft1024(A, B, W, tid);                                        // -> B; y[i] = Re B[i] (H carries the 1/1024)
        // 3. overlap-add and emit
        for (uint32_t i = tid; i < chunk; i += 256u) {
            const float y = B[fill + i].x + st.overlap[fill + i];
            out[processed + i] = d.fuseRootRec == kNone ? y : y * rootGain(processed + i);
        }
        __syncthreads();
        fill += chunk;
        if (fill == 512u) {
            for (uint32_t j = tid; j < 512u; j += 256u) {
                st.overlap[j] = B[512u + j].x;
                st.X[(size_t)(blk % st.P) * 512u + j] = Xk[j];
            }
            // the node's last input blocks in the time domain: the history a long-partition launch set starts from (conv_long.inc)
            if (const uint32_t R = st.hdr[conv::H_HISTBLKS]) {
                gfp hist = st.overlap + 512u + (size_t)(blk % R) * 512u;
                for (uint32_t j = tid; j < 512u; j += 256u) hist[j] = st.inbuf[j];
            }
            fill = 0u; blk += 1u;
        }
        processed += chunk;
    }
    __syncthreads();
    if (tid == 0u) {
        st.hdr[conv::H_FILL_NEXT] = fill; st.hdr[conv::H_BLK_NEXT] = blk;
        // the helpers of this launch are producing the older-partition sum of block blk0 + 1
        st.hdr[conv::H_PREVALID0 + ((blk0 + 1u) & 1u)] = blk0 + 1u;
    }
}

__device__ void conv_helper(const ConvDesc d, uint32_t h, gup recs, const Globals* g, c2* red) {
    gcup r = (gcup)(recs + d.rec * kRecDwords);
    const uint64_t sp = (uint64_t)r[rec::CONV_STATE] | ((uint64_t)r[rec::CONV_STATE + 1] << 32);
    uint32_t inKind = d.inKind;
    if (inKind == 3u) inKind = g->numIn > 0 ? 1u : 0u;
    if (sp == 0ull || inKind == 0u) return;                           // (kind 5 always convolves: its channel or silence)
    const State st = state_of((gup)reinterpret_cast<uint32_t*>(sp));
    if (st.P <= 2u) return;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t k = (h % conv::kBinGroups) * 64u + lane;
    const uint32_t b = st.hdr[conv::H_BLK] + 1u;
    for (uint32_t s = h / conv::kBinGroups; s < st.S; s += d.slices) {
        const c2 acc = older_sum(st, b, k, s * 4u + w, 4u * st.S);
        __syncthreads();
        red[threadIdx.x] = acc;
        __syncthreads();
        if (w == 0u) {
            c2 t = cadd(cadd(red[lane], red[64u + lane]), cadd(red[128u + lane], red[192u + lane]));
            st.pre[((size_t)(b & 1u) * st.S + s) * 512u + k] = t;
        }
    }
}

} // namespace

__global__ __launch_bounds__(256)
void elemhip_convolve_kernel(PlanView pv, uint32_t* recs, float* hbm, const Globals* g, uint32_t workBegin) {
    __shared__ c2 A[conv::kFft], B[conv::kFft], W[conv::kFft], Xk[512];
    const uint32_t entry = pv.convWork[workBegin + blockIdx.x];
    const ConvDesc d = pv.convs[entry & 0xFFFFu];
    const uint32_t role = entry >> 16;
    if (!root_running((gcup)recs, d.rootRec, g->numOut)) return;
    if (role == 0u) conv_main(d, (gup)recs, (gfp)hbm, g, A, B, W, Xk);
    else conv_helper(d, role - 1u, (gup)recs, g, A);
}

// ---- multi-block launches (elemhip_process_blocks) --------------------------------------------------------------------
// With B whole 512-frame blocks in one launch set, every input block of the batch is known before the convolve level
// starts (the levels in front of it rendered all B blocks), so nothing is left of the per-block latency path:
//   K1  (node, j): spectrum of input block j -> Xnew[j]                                 (one forward FFT each, all parallel)
//   K2  (node, j): Y_j = sum_p H_p X_{b0+j-p} over ALL partitions (new spectra from Xnew, older ones from the ring),
//                  inverse FFT, head half -> the node's output buffer of block j, tail half -> tails[j + 1]
//   K3  (node, j): out_j += tails[j] (tails[0] = the overlap carried in), folded root gain; last B (<= P) spectra go
//                  into the ring, tails[B] becomes the carried overlap, the block counter advances by B.
// Three launches per launch set instead of two per block. The helpers' pre-multiplied sums are not produced: the first
// per-block call after a batch takes conv_main's own older_sum path once. The host only uses this path while every
// call so far rendered whole 512-frame blocks (fill == 0, Engine::convAligned).
// Scratch per node (floats): [0] b0 | 16 + Xnew[B][512] c2 | tails[B + 1][512] | Ysum[kMacParts][B][512] c2
namespace {

constexpr uint32_t kBatchHdr = 16;
constexpr uint32_t kMacParts = 2;       // workgroups the partitions of one (node, bin tile, 64-block chunk) are cut into
// the matrix-core MAC reads TILED copies: per 16-bin tile a contiguous slab of rows [kTileHist + maxBatch][16] (time - the launch
// set's first block + kTileHist) and of IR rows [kTileHist][16] — a workgroup's whole window is one contiguous stream, where the
// natural [row][512] layout hands out 128-byte pieces 4 KB apart
constexpr uint32_t kTileHist = 192;     // history rows (= the longest IR, in partitions, the matrix-core kernel takes)
__host__ __device__ __forceinline__ size_t batch_scratch_floats_512(uint32_t maxBatch) {
    return kBatchHdr + (size_t)maxBatch * 1024u + (size_t)(maxBatch + 1u) * 512u + (size_t)kMacParts * maxBatch * 1024u
         + (size_t)(kTileHist + maxBatch) * 1024u + (size_t)kTileHist * 1024u;
}

struct BatchCtx {
    State st; bool live; uint32_t inKind, inBuf; float cval; gfp scratch; gf2p xnew; gfp tails; gf2p ysum; float gain;
    gf2p xt, ht; uint32_t xtRows;      // tiled spectra [32 tiles][xtRows][16], tiled IR spectra [32][kTileHist][16]
};

// decode shared by the three kernels; live == false: the node writes zeros (Convolve.h:70-71) or does nothing
__device__ __forceinline__ bool batch_ctx(const ConvDesc& d, gup recs, const Globals* g, float* scratchAll, uint32_t convIdx, uint32_t maxBatch, BatchCtx& c, size_t perNode) {
    gcup r = (gcup)(recs + d.rec * kRecDwords);
    const uint64_t sp = (uint64_t)r[rec::CONV_STATE] | ((uint64_t)r[rec::CONV_STATE + 1] << 32);
    c.inKind = d.inKind; c.inBuf = d.inIdx;
    if (c.inKind == 3u) { c.inKind = g->numIn > 0 ? 1u : 0u; c.inBuf = 0u; }
    if (c.inKind == 5u) {
        const uint32_t ch = ((gcup)recs)[d.inIdx * kRecDwords + rec::P0];
        if (ch < g->numIn) { c.inKind = 1u; c.inBuf = ch; } else c.inKind = 4u;
    }
    c.cval = c.inKind == 2u ? __uint_as_float(((gcup)recs)[d.inIdx * kRecDwords + rec::P0]) : 0.0f;
    c.gain = 1.0f;
    if (d.fuseRootRec != kNone) c.gain = __uint_as_float(((gcup)recs)[d.fuseRootRec * kRecDwords + rec::ROOT_TARGET]);   // settled fades only (Engine::batchEligible)
    c.scratch = (gfp)(scratchAll + (size_t)convIdx * perNode);      // per node: the 512-partition area, then the long-partition area (conv_long.inc)
    c.xnew = (gf2p)(c.scratch + kBatchHdr);
    c.tails = c.scratch + kBatchHdr + (size_t)maxBatch * 1024u;
    c.ysum = (gf2p)(c.tails + (size_t)(maxBatch + 1u) * 512u);
    c.xtRows = kTileHist + maxBatch;
    c.xt = c.ysum + (size_t)kMacParts * maxBatch * 512u;
    c.ht = c.xt + (size_t)c.xtRows * 512u;
    c.live = false;
    if (sp == 0ull || c.inKind == 0u) return false;
    c.st = state_of((gup)reinterpret_cast<uint32_t*>(sp));
    if (c.st.P == 0u) return false;
    c.live = true;
    return true;
}

} // namespace

__global__ __launch_bounds__(256)
void elemhip_convolve_batch_fft(PlanView pv, uint32_t* recs, float* hbm, const Globals* g, uint32_t workBegin,
                                uint32_t arenaFloats, float* scratchAll, uint32_t maxBatch, uint32_t macMode, uint32_t longMode, size_t perNode) {
    __shared__ c2 A[conv::kFft], B[conv::kFft], W[conv::kFft];
    const uint32_t convIdx = pv.convWork[workBegin + blockIdx.x] & 0xFFFFu, j = blockIdx.y, tid = threadIdx.x;
    const ConvDesc d = pv.convs[convIdx];
    if (!root_running((gcup)recs, d.rootRec, g->numOut)) return;
    BatchCtx c;
    if (!batch_ctx(d, (gup)recs, g, scratchAll, convIdx, maxBatch, c, perNode)) return;
    if (conv_use_long(c.st, gridDim.y, longMode)) return;           // this node's share of the set goes through the long-partition kernels
    const uint32_t stride = g->blockStride;
    gcfp in = (gcfp)(hbm + (size_t)j * arenaFloats + (size_t)(c.inKind == 1u ? c.inBuf : 0u) * stride);
    for (uint32_t i = tid; i < conv::kFft; i += 256u) W[i] = kTwiddle[i];
#pragma unroll
    for (uint32_t q = 0; q < 2; ++q) {
        const uint32_t i = tid + 256u * q;
        A[i] = mk(c.inKind == 1u ? in[i] : c.cval, 0.0f);
        A[i + 512u] = mk(0.0f, 0.0f);
    }
    __syncthreads();
    fft1024(A, B, W, tid);
    const bool tiled = macMode != 0u && c.st.P <= kTileHist;     // the matrix-core MAC takes this node (elemhip_convolve_batch_mac)
#pragma unroll
    for (uint32_t q = 0; q < 2; ++q) {
        const uint32_t k = tid + 256u * q;
        const c2 v = k == 0u ? mk(B[0].x, B[512].x) : B[k];
        c.xnew[(size_t)j * 512u + k] = v;
        if (tiled) c.xt[((size_t)(k >> 4) * c.xtRows + kTileHist + j) * 16u + (k & 15u)] = v;
    }
    if (tiled) {   // the rows older than the set (the node's spectra ring) and the IR spectra, tile by tile: row `back` blocks back sits at kTileHist - back
        const uint32_t P = c.st.P, bm = c.st.hdr[conv::H_BLK] % P, nb = gridDim.y;
        for (uint32_t back = j + 1u; back <= P; back += nb) {
            const uint32_t slot = bm >= back ? bm - back : bm + P - back;
#pragma unroll
            for (uint32_t q = 0; q < 2; ++q) { const uint32_t k = tid + 256u * q; c.xt[((size_t)(k >> 4) * c.xtRows + kTileHist - back) * 16u + (k & 15u)] = ((gcf2p)c.st.X)[(size_t)slot * 512u + k]; }
        }
        for (uint32_t p = j; p < P; p += nb) {
#pragma unroll
            for (uint32_t q = 0; q < 2; ++q) { const uint32_t k = tid + 256u * q; c.ht[((size_t)(k >> 4) * kTileHist + p) * 16u + (k & 15u)] = ((gcf2p)c.st.H)[(size_t)p * 512u + k]; }
        }
    }
    if (j == 0u) {   // what the later kernels must not read from state another workgroup of theirs rewrites
        for (uint32_t i = tid; i < 512u; i += 256u) c.tails[i] = c.st.overlap[i];
        if (tid == 0u) ((gup)c.scratch)[0] = c.st.hdr[conv::H_BLK];
    }
}

// K2a: the partition sums of a launch set, H read ONCE per set. For one bin k the sums of the set's blocks are a convolution
// along the block index: Y_j[k] = sum_p H_p[k] X_{b0+j-p}[k]. A workgroup owns 16 bins of one node and (up to) 64
// consecutive blocks: thread (bin, jg) accumulates the four blocks j0 .. j0 + 3 (j0 = 4 jg) while p runs over the
// partitions — per step ONE new input spectrum value (the window X_{j0+q-p}, q = 0..3, slides by one: three of its
// values stay in registers), one H value shared by the 16 threads of the bin group, four complex multiply-adds.
// r02's kernel ran one workgroup per (node, block) over all partitions: every block re-read all of H and its whole input
// window (12.3 MB per block against 1.28 MB algorithmic); here a 64-block set reads H and the ring once and the 64
// new spectra once per bin tile.
constexpr uint32_t kMacU = 8, kMacR = 128, kMacDP = 3;     // partitions per chunk, ring rows, chunks the global loads run ahead
template <bool HasPacked>
__device__ __forceinline__ void batch_mac_tile(const BatchCtx& c, uint32_t tile, uint32_t jBase, uint32_t batch, uint32_t tid,
                                               uint32_t part, uint32_t maxBatch, c2 (&Xs)[kMacR][20], c2 (&Hs)[kMacDP + 1u][kMacU][16]) {
    // LDS: a ring of input-spectrum rows (16 bins each) indexed by block time, and two chunks of H rows. A chunk is 8
    // partitions; per chunk the workgroup brings in 8 new ring rows and 8 H rows — ONE 8-byte value per thread — while the
    // 64 x 16 outputs it accumulates read everything else from LDS.
    constexpr uint32_t U = kMacU, R = kMacR, DP = kMacDP;
    // (row pitch 20 values = 40 dwords: the four block groups of a wave read rows 4 apart — 160 dwords = 32 banks on — so each
    //  half-wave's 8-byte reads cover all 64 banks once; a 16-value pitch put all four on the same 32 banks)
    // (Xs / Hs belong to the kernel: as statics of this template they existed once per instantiation, 48 KB per workgroup
    //  instead of 24 — three workgroups per CU instead of six)
    const uint32_t bin = tid & 15u, row = tid >> 4, kb = tile * 16u + bin, jg = row;      // bin, load row / block group
    const uint32_t j0 = jBase + jg * 4u;
    const uint32_t P = c.st.P, b0 = c.st.hdr[conv::H_BLK];
    const bool packed = HasPacked && kb == 0u;                            // bin 0 carries two real bins (DC, Nyquist)
    gcf2p H = (gcf2p)c.st.H + kb, X = (gcf2p)c.st.X + kb, N = (gcf2p)c.xnew + kb;
    // x(t): input spectrum of block b0 + t — 0 <= t < batch: a block of this set; t < 0: the ring (slot of block b is b mod P)
    const uint32_t bm = b0 % P;
    auto xat = [&](int t) -> c2 {
        if (t >= (int)batch) return mk(0.0f, 0.0f);                      // (rows only the unused lanes of a ragged last group look at)
        if (t >= 0) return N[(size_t)t * 512u];
        uint32_t back = (uint32_t)(-t);                                   // 1 .. P - 1: blocks older than P never contribute
        if (back > P) back = P;                                           // (a padded step past the last partition: any valid slot)
        const uint32_t slot = bm >= back ? bm - back : bm + P - back;
        return X[(size_t)slot * 512u];
    };
    // the partitions are cut into kMacParts runs, a workgroup each; every run leaves its own partial sum, the inverse-FFT
    // kernel adds them up
    const uint32_t perPart = (P + kMacParts - 1u) / kMacParts, pBegin = part * perPart, pEnd = pBegin + perPart < P ? pBegin + perPart : P;
    gf2p Y = c.ysum + (size_t)part * maxBatch * 512u + kb;
    if (pBegin >= pEnd) {
        for (uint32_t q = 0; q < 4u; ++q) if (j0 + q < batch) Y[(size_t)(j0 + q) * 512u] = mk(0.0f, 0.0f);
        return;
    }
    auto hrow = [&](uint32_t p) -> c2 { return p < pEnd ? H[(size_t)p * 512u] : mk(0.0f, 0.0f); };   // partitions past the end multiply by zero
    // initial fill: ring rows t in [jBase - 8 DP - pBegin, jBase + 63 - pBegin] (the first DP chunks' new entries and every
    // thread's starting window), H rows of the first DP chunks
    {   // (every load issued before the first LDS write: the fill costs one memory round trip, not one per row)
        const int tLo = (int)jBase - (int)(U * DP) - (int)pBegin;
        constexpr uint32_t NX = (64u + U * DP + 15u) / 16u, NH = (U * DP + 15u) / 16u;
        c2 fx[NX], fh[NH];
#pragma unroll
        for (uint32_t i = 0; i < NX; ++i) { const uint32_t r = row + 16u * i; fx[i] = r < 64u + U * DP ? xat(tLo + (int)r) : mk(0.0f, 0.0f); }
#pragma unroll
        for (uint32_t i = 0; i < NH; ++i) { const uint32_t r = row + 16u * i; fh[i] = r < U * DP ? hrow(pBegin + r) : mk(0.0f, 0.0f); }
#pragma unroll
        for (uint32_t i = 0; i < NX; ++i) { const uint32_t r = row + 16u * i; if (r < 64u + U * DP) Xs[(uint32_t)(tLo + (int)r) & (R - 1u)][bin] = fx[i]; }
#pragma unroll
        for (uint32_t i = 0; i < NH; ++i) { const uint32_t r = row + 16u * i; if (r < U * DP) Hs[r / U][r % U][bin] = fh[i]; }
    }
    __syncthreads();
    auto xs = [&](int t) -> c2 { return Xs[(uint32_t)t & (R - 1u)][bin]; };
    c2 w0 = xs((int)j0 - (int)pBegin), w1 = xs((int)j0 + 1 - (int)pBegin), w2 = xs((int)j0 + 2 - (int)pBegin), w3 = xs((int)j0 + 3 - (int)pBegin);
    c2 a0 = mk(0.0f, 0.0f), a1 = a0, a2 = a0, a3 = a0;
    auto mac = [&](c2& acc, c2 h, c2 x) {
        if constexpr (!HasPacked) {
            // two v_pk_fma_f32 per complex multiply-add instead of four v_fma_f32 — the same four fused operations in the same
            // order per component: (re, im) += (h.re, h.re) * (x.re, x.im); (re, im) += (-h.im, h.im) * (x.im, x.re)
            typedef float pk2 __attribute__((ext_vector_type(2)));
            pk2 a = __builtin_bit_cast(pk2, acc);
            const pk2 hv = __builtin_bit_cast(pk2, h), xv = __builtin_bit_cast(pk2, x);
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
                : "+v"(a) : "v"(hv), "v"(xv));
            acc = __builtin_bit_cast(c2, a);
            return;
        }
        if (HasPacked && packed) { acc.x = __builtin_fmaf(h.x, x.x, acc.x); acc.y = __builtin_fmaf(h.y, x.y, acc.y); }
        else {
            acc.x = __builtin_fmaf(h.x, x.x, acc.x); acc.x = __builtin_fmaf(-h.y, x.y, acc.x);
            acc.y = __builtin_fmaf(h.x, x.y, acc.y); acc.y = __builtin_fmaf(h.y, x.x, acc.y);
        }
    };
    // this thread's piece of chunk `ck` (rows 0..7: a new ring row, rows 8..15: an H row)
    auto piece = [&](uint32_t ck) -> c2 {
        const uint32_t pk = pBegin + ck * U;
        if (pk >= pEnd) return mk(0.0f, 0.0f);
        return row < U ? xat((int)jBase - (int)U - (int)pk + (int)row) : hrow(pk + row - U);
    };
    auto stash = [&](uint32_t ck, c2 v) {
        const uint32_t pk = pBegin + ck * U;
        if (pk >= pEnd) return;
        if (row < U) Xs[(uint32_t)((int)jBase - (int)U - (int)pk + (int)row) & (R - 1u)][bin] = v; else Hs[ck % (DP + 1u)][row - U][bin] = v;
    };
    // chunk ck's piece is loaded while chunk ck - DP is computed and put into LDS DP - 1 chunks later (one barrier before its
    // first reader): the load has two whole chunks of arithmetic to come back
    c2 q1 = piece(DP), q2 = mk(0.0f, 0.0f);                  // in flight: chunk DP (for the end of chunk 0), chunk DP + 1 (end of chunk 1)
    uint32_t ck = 0u;
    for (uint32_t p0 = pBegin; p0 < pEnd; p0 += U, ++ck) {
        q2 = piece(ck + DP + 1u);
        const uint32_t hb = ck % (DP + 1u);
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const c2 h = Hs[hb][u][bin];
            const c2 xn = xs((int)j0 - 1 - (int)(p0 + u));                   // enters the window after partition p0 + u
            mac(a0, h, w0); mac(a1, h, w1); mac(a2, h, w2); mac(a3, h, w3);
            w3 = w2; w2 = w1; w1 = w0; w0 = xn;
        }
        stash(ck + DP, q1);
        q1 = q2;
        __syncthreads();
    }
    if (j0 < batch) Y[(size_t)j0 * 512u] = a0;
    if (j0 + 1u < batch) Y[(size_t)(j0 + 1u) * 512u] = a1;
    if (j0 + 2u < batch) Y[(size_t)(j0 + 2u) * 512u] = a2;
    if (j0 + 3u < batch) Y[(size_t)(j0 + 3u) * 512u] = a3;
}


// K2a on the matrix cores (north_star: "MFMA used only for the partitioned-convolution dense FFT x IR multiply"; the reference's
// call site is wasm/Convolve.h:49,73-84). For one bin the set's partition sums are a Toeplitz product along the block index:
// y[t] = sum_p h[p] x[t - p]. Take four consecutive taps and four consecutive input times: the outer product
//     D[m][n] += x[t0 - 4J - m] * h[4J + n]            (m, n = 0..3; J = tap group)
// accumulated over J holds, on its seven anti-diagonals, partial sums of y[t0 + n - m] — every one of the 16 products is a term
// the convolution needs, and every (t, p) pair is formed exactly once over the tiles t0 = 0, 4, 8, ...: y[t0 + n] = P_n(t0) +
// Q_n(t0 + 4) with P_n = sum of the diagonal n - m = n (n + m' < 4) and Q_n = the diagonal n - 4 of the NEXT tile. That is
// v_mfma_f32_4x4x1_16B_f32: sixteen independent 4x4x1 blocks per instruction = the sixteen bins of the workgroup's tile, A = one
// x value per lane (lane 4 b + m: bin b, time offset m), B = one h value per lane (lane 4 b + n), D = 4 registers (row m) — 256
// multiply-adds per instruction from two operand registers, and because A of (tile i, group J) is A of (tile i + 1, group J + 1)
// a wave keeps a window of x segments in registers and loads ONE new x segment and ONE h segment per 20 instructions. Complex =
// four real products (xr hr, xi (-hi) -> re; xi hr, xr hi -> im). f32 MFMA is an exact fmaf chain (MI355X_MICROARCH.md), so the
// result differs from the vector kernel's only in the order of the sum (taps by residue mod 4, then the diagonals).
// Same workgroup geometry, LDS ring and H staging as batch_mac_tile; a wave owns 16 consecutive blocks = 4 output tiles + the
// fifth tile whose lower diagonals complete the fourth.
typedef float f4v __attribute__((ext_vector_type(4)));
template <int V> struct MfmaIntC { static constexpr int value = V; };
template <int I, int N, class F> __device__ __forceinline__ void mfma_static_for(F&& f) { if constexpr (I < N) { f(MfmaIntC<I>{}); mfma_static_for<I + 1, N>(f); } }
template <int CTRL>
__device__ __forceinline__ float quad_rot(float v) {      // value of lane (n + k) % 4 of the same quad, k encoded in CTRL
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// The matrix-core variant is PERSISTENT over the block axis: a workgroup owns (node, 16-bin tile) and walks `chunks` runs of
// kMfmaOut = 128 blocks. LDS holds the tile's IR rows (loaded once) and a 512-row ring of input-spectrum rows, both read from the
// TILED copies the fft kernel leaves — contiguous streams; while a run's MFMAs execute, the next run's 128 new rows are already on
// their way (registers -> ring after the run). Every spectrum row and every IR row is read from memory ONCE per (node, tile).
// Measured on the way here (profiles/r04/c3_mac_variants.txt, 8 channels x 188 partitions, 1024-block sets): the vector kernel's
// chunked ring with MFMA arithmetic 131 us (= the vector kernel: both wait out a global round trip per 8-partition chunk);
// everything up front from the natural [row][512] layout, 64 blocks x half the partitions per workgroup 183 us (8192 workgroups x
// 33 KB in 128-byte pieces 4 KB apart); tiled copies, all partitions, 128 blocks per workgroup 162 us with a rotating register
// window (the compiler copied window AND accumulators every step) and 92 us with compile-time window slots; persistent: below.
constexpr uint32_t kMfmaSteps = kTileHist / 4u;                      // tap groups of 4
constexpr uint32_t kMfmaOut = 128;                                   // blocks per run
constexpr uint32_t kMfmaRing = 336;                                  // ring rows: a run reads kMfmaOut + 4 kMfmaSteps + 5 = 325 of them; the next
                                                                     // run's 128 new rows replace rows that are dead by then
constexpr uint32_t kMfmaLdsBytes = (kMfmaRing + kTileHist) * 16u * 8u;    // 42 KB + 24 KB: two workgroups per CU
template <bool HasPacked, bool Swap>
__device__ __forceinline__ void batch_mac_tile_mfma(const BatchCtx& c, uint32_t tile, uint32_t jFirst, uint32_t chunks, uint32_t batch, uint32_t tid,
                                                    c2 (*Xs)[16], c2 (*Hs)[16]) {
    constexpr int NT = 8;                                                   // output tiles per wave (+ 1 for the last tile's lower diagonals)
    const uint32_t bin = tid & 15u, row = tid >> 4;                         // load roles: 16 rows x 16 bins per pass = 2 KB contiguous
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t mb = lane >> 2, mq = lane & 3u;                          // compute roles: MFMA block (= bin of the tile) and row / column inside it
    const uint32_t P = c.st.P;
    gcf2p Xt = (gcf2p)c.xt + (size_t)tile * c.xtRows * 16u + bin, Ht = (gcf2p)c.ht + (size_t)tile * kTileHist * 16u + bin;
    // x(t): input spectrum of block (set start + t); t < 0: the rows the fft kernel copied out of the node's ring
    auto xat = [&](int t) -> c2 {
        if (t >= (int)batch || t < -(int)P) return mk(0.0f, 0.0f);         // (rows older than the IR is long meet zero taps only: keep NaNs out)
        return Xt[(size_t)((int)kTileHist + t) * 16u];
    };
    const uint32_t steps = (P + 3u) / 4u;                                   // <= kMfmaSteps (P <= kTileHist: checked by the caller)
    const int tBase = (int)jFirst - 3 * (int)kMfmaRing;                     // ring row of time t: (t - tBase) mod kMfmaRing (t - tBase > 0)
    auto ringRow = [&](int t) -> uint32_t { return (uint32_t)(t - tBase) % kMfmaRing; };                 // (loads and window set-up only: the tap loop steps a row index)
    {   // IR rows and the first run's window [jFirst - 4 steps - 4, jFirst + kMfmaOut]
        constexpr uint32_t NX = (kMfmaOut + 4u * kMfmaSteps + 8u + 15u) / 16u, NH = kTileHist / 16u;
        const int tLo = (int)jFirst - 4 * (int)steps - 4;
        const uint32_t rows = kMfmaOut + 4u * steps + 5u;
        c2 fx[NX], fh[NH];
#pragma unroll
        for (uint32_t i = 0; i < NX; ++i) { const uint32_t r = row + 16u * i; fx[i] = r < rows ? xat(tLo + (int)r) : mk(0.0f, 0.0f); }
#pragma unroll
        for (uint32_t i = 0; i < NH; ++i) { const uint32_t p = row + 16u * i; fh[i] = p < P ? Ht[(size_t)p * 16u] : mk(0.0f, 0.0f); }   // taps past the end multiply by zero
#pragma unroll
        for (uint32_t i = 0; i < NX; ++i) { const uint32_t r = row + 16u * i; if (r < rows) Xs[ringRow(tLo + (int)r)][bin] = fx[i]; }
#pragma unroll
        for (uint32_t i = 0; i < NH; ++i) Hs[row + 16u * i][bin] = fh[i];
    }
    __syncthreads();
    const bool packedLane = HasPacked && tile == 0u && mb == 0u;            // bin 0 carries two REAL bins (DC, Nyquist): (hr xr, hi xi)
    auto mma = [&](float a, float b, f4v acc) -> f4v {
        if constexpr (Swap) return __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, acc, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    };
    // anti-diagonals: lane n of a quad collects D[m][(n + m) % 4] from register m; n + m < 4 belongs to y[t0 + n] of this tile (P),
    // n + m >= 4 to y[t0 - 4 + n], i.e. to the previous tile's outputs (Q)
    auto diag = [&](const f4v& D, float& Pn, float& Qn) {
        const float v1 = quad_rot<0x39>(D.y), v2 = quad_rot<0x4E>(D.z), v3 = quad_rot<0x93>(D.w);   // quad_perm [1,2,3,0], [2,3,0,1], [3,0,1,2]
        Pn = D.x; Qn = 0.0f;
        if (mq + 1u < 4u) Pn += v1; else Qn += v1;
        if (mq + 2u < 4u) Pn += v2; else Qn += v2;
        if (mq + 3u < 4u) Pn += v3; else Qn += v3;
    };
    gf2p Y = c.ysum + tile * 16u + mb;                                      // one partition run: partial sum 0 is the sum
    constexpr int NS = NT + 1;
    for (uint32_t ch = 0; ch < chunks; ++ch) {
        const uint32_t jBase = jFirst + ch * kMfmaOut;
        if (jBase >= batch) break;
        // the next run's new rows (jBase + kMfmaOut, jBase + 2 kMfmaOut]: in flight during this run's arithmetic
        constexpr uint32_t NP = kMfmaOut / 16u;
        c2 pre[NP];
        const bool more = ch + 1u < chunks && jBase + kMfmaOut < batch;
        if (more) {
#pragma unroll
            for (uint32_t i = 0; i < NP; ++i) pre[i] = xat((int)(jBase + kMfmaOut + 1u + row + 16u * i));
        }
        const uint32_t Tw = jBase + wave * 32u;                             // this wave's first output block
        // x segment sigma of this wave: lane (b, m) holds x[Tw + 4 sigma - m] of bin b. The window of segments lives in NT + 1
        // registers used round-robin — segment sigma in slot sigma mod (NT + 1) — and the tap loop is unrolled NT + 1 times so that
        // every slot index is a compile-time constant.
        auto xseg = [&](int sigma) -> c2 { return Xs[ringRow((int)Tw + 4 * sigma - (int)mq)][mb]; };
        c2 W[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) W[i] = xseg(i);
        f4v Dr[NS], Di[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) { Dr[i] = f4v{0.0f, 0.0f, 0.0f, 0.0f}; Di[i] = Dr[i]; }
        // the h values and the entering x segment of tap group J + 1 are read from LDS while group J's MFMAs execute (a lone wave
        // per SIMD has nothing else to hide an LDS round trip behind)
        int xr = (int)ringRow((int)Tw - 4 - (int)mq);                       // ring row of segment sigma = -1; steps down by 4 per group
        c2 hN = Hs[mq][mb], xN = Xs[xr][mb];
        for (int J0 = 0; J0 < (int)steps; J0 += NS) {                       // tap groups J0 .. J0 + NT (taps 4 J .. 4 J + 3 each)
            mfma_static_for<0, NS>([&](auto Uc) {
                constexpr int u = decltype(Uc)::value;
                const int J = J0 + u;
                if (J >= (int)steps) return;
                const c2 h = hN, xn = xN;                                   // lane (b, n): h[4 J + n] of bin b; the segment entering after this group
                {
                    const int Jn = J + 1 < (int)steps ? J + 1 : J;          // (the last group re-reads its own rows: unused)
                    xr -= 4; if (xr < 0) xr += (int)kMfmaRing;
                    hN = Hs[4 * Jn + (int)mq][mb]; xN = Xs[xr][mb];
                }
                // the four real products of a complex multiply-add; the packed bin's lanes get (hr xr, hi xi) from the same four instructions
                const float hB1 = h.x, hB2 = packedLane ? 0.0f : -h.y, hB3 = packedLane ? h.y : h.x, hB4 = packedLane ? 0.0f : h.y;
                // tile i multiplies segment sigma = i - J: slot (i - u) mod NS (J0 is a multiple of NS)
                mfma_static_for<0, NS>([&](auto Ic) { constexpr int i = decltype(Ic)::value; constexpr int sl = (i - u + NS) % NS; Dr[i] = mma(W[sl].x, hB1, Dr[i]); });
                mfma_static_for<0, NS>([&](auto Ic) { constexpr int i = decltype(Ic)::value; constexpr int sl = (i - u + NS) % NS; Di[i] = mma(W[sl].y, hB3, Di[i]); });
                mfma_static_for<0, NS>([&](auto Ic) { constexpr int i = decltype(Ic)::value; constexpr int sl = (i - u + NS) % NS; Dr[i] = mma(W[sl].y, hB2, Dr[i]); });
                mfma_static_for<0, NS>([&](auto Ic) { constexpr int i = decltype(Ic)::value; constexpr int sl = (i - u + NS) % NS; Di[i] = mma(W[sl].x, hB4, Di[i]); });
                W[(NT - u + NS) % NS] = xn;                                 // the oldest segment (tile NT's) makes room for sigma = -(J + 1)
            });
        }
        float Pr[NS], Qr[NS], Pi[NS], Qi[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) { diag(Dr[i], Pr[i], Qr[i]); diag(Di[i], Pi[i], Qi[i]); }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const uint32_t t = Tw + 4u * (uint32_t)i + mq;
            if (t < batch) Y[(size_t)t * 512u] = mk(Pr[i] + Qr[i + 1], Pi[i] + Qi[i + 1]);
        }
        if (more) {   // every wave is done with this run's window: the new rows replace the oldest ones
            __syncthreads();
#pragma unroll
            for (uint32_t i = 0; i < NP; ++i) Xs[ringRow((int)(jBase + kMfmaOut + 1u + row + 16u * i))][bin] = pre[i];
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256)
void elemhip_convolve_batch_mac(PlanView pv, uint32_t* recs, float* hbm, const Globals* g, uint32_t workBegin,
                                uint32_t arenaFloats, float* scratchAll, uint32_t maxBatch, uint32_t batch, uint32_t mode, uint32_t longMode, size_t perNode) {
    const uint32_t convIdx = pv.convWork[workBegin + blockIdx.x] & 0xFFFFu, tile = blockIdx.y, tid = threadIdx.x;
    const uint32_t part = blockIdx.z % kMacParts, jBase = (blockIdx.z / kMacParts) * 64u;
    const ConvDesc d = pv.convs[convIdx];
    if (!root_running((gcup)recs, d.rootRec, g->numOut)) return;
    BatchCtx c;
    if (!batch_ctx(d, (gup)recs, g, scratchAll, convIdx, maxBatch, c, perNode)) return;     // (the ifft kernel writes the zeros)
    if (conv_use_long(c.st, batch, longMode)) return;
    // (mode != 0: the nodes whose IR fits the matrix-core kernel are rendered by elemhip_convolve_batch_mac_mfma)
    if (mode != 0u && c.st.P <= kTileHist) return;
    __shared__ __attribute__((aligned(16))) char ldsRaw[(kMacR * 20u + (kMacDP + 1u) * kMacU * 16u) * 8u];
    c2 (&Xs)[kMacR][20] = *reinterpret_cast<c2 (*)[kMacR][20]>(ldsRaw);
    c2 (&Hs)[kMacDP + 1u][kMacU][16] = *reinterpret_cast<c2 (*)[kMacDP + 1u][kMacU][16]>(ldsRaw + (size_t)kMacR * 20u * 8u);
    if (tile == 0u) batch_mac_tile<true>(c, tile, jBase, batch, tid, part, maxBatch, Xs, Hs);
    else batch_mac_tile<false>(c, tile, jBase, batch, tid, part, maxBatch, Xs, Hs);
}

// K2a on the matrix cores: one workgroup per (node, 16-bin tile, kMfmaOut blocks). mode (engine option "conv_mfma"): 1 = default
// (IRs of up to kTileHist partitions; longer ones stay with the vector kernel), 0 = packed vector FMAs only (r03), 2 = MFMA with the
// operand roles exchanged (the layout probe of the bring-up: wrong sums by construction).
__global__ __launch_bounds__(256)
void elemhip_convolve_batch_mac_mfma(PlanView pv, uint32_t* recs, float* hbm, const Globals* g, uint32_t workBegin,
                                     uint32_t arenaFloats, float* scratchAll, uint32_t maxBatch, uint32_t batch, uint32_t mode, uint32_t chunks, uint32_t longMode, size_t perNode) {
    const uint32_t convIdx = pv.convWork[workBegin + blockIdx.x] & 0xFFFFu, tile = blockIdx.y, tid = threadIdx.x, jFirst = blockIdx.z * chunks * kMfmaOut;
    const ConvDesc d = pv.convs[convIdx];
    if (jFirst >= batch || !root_running((gcup)recs, d.rootRec, g->numOut)) return;
    BatchCtx c;
    if (!batch_ctx(d, (gup)recs, g, scratchAll, convIdx, maxBatch, c, perNode) || c.st.P > kTileHist || conv_use_long(c.st, batch, longMode)) return;
    __shared__ __attribute__((aligned(16))) char ldsRaw[kMfmaLdsBytes];
    c2 (*Xm)[16] = reinterpret_cast<c2 (*)[16]>(ldsRaw);
    c2 (*Hm)[16] = reinterpret_cast<c2 (*)[16]>(ldsRaw + (size_t)kMfmaRing * 16u * 8u);
    if (tile == 0u) batch_mac_tile_mfma<true, false>(c, tile, jFirst, chunks, batch, tid, Xm, Hm);
    else batch_mac_tile_mfma<false, false>(c, tile, jFirst, chunks, batch, tid, Xm, Hm);
}

// K2b (node, j): inverse FFT of the block's partition sum; head half -> the node's output buffer of block j, tail half -> tails[j + 1]
__global__ __launch_bounds__(256)
void elemhip_convolve_batch_ifft(PlanView pv, uint32_t* recs, float* hbm, const Globals* g, uint32_t workBegin,
                                 uint32_t arenaFloats, float* scratchAll, uint32_t maxBatch, uint32_t macMode, uint32_t longMode, size_t perNode) {
    __shared__ c2 A[conv::kFft], B[conv::kFft], W[conv::kFft];
    const uint32_t convIdx = pv.convWork[workBegin + blockIdx.x] & 0xFFFFu, j = blockIdx.y, tid = threadIdx.x;
    const ConvDesc d = pv.convs[convIdx];
    if (!root_running((gcup)recs, d.rootRec, g->numOut)) return;
    BatchCtx c;
    const uint32_t stride = g->blockStride;
    gfp out = (gfp)hbm + (size_t)j * arenaFloats + (size_t)d.outHbm * stride;
    if (!batch_ctx(d, (gup)recs, g, scratchAll, convIdx, maxBatch, c, perNode)) {
        for (uint32_t i = tid; i < 512u; i += 256u) out[i] = 0.0f;
        return;
    }
    if (conv_use_long(c.st, gridDim.y, longMode)) return;
    for (uint32_t i = tid; i < conv::kFft; i += 256u) W[i] = kTwiddle[i];
    // bins 2 tid, 2 tid + 1 of the sum; the conjugate-symmetric half makes the 1024-point transform's input
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) const f4* gcf4p;
    f4 y = ((gcf4p)(c.ysum + (size_t)j * 512u))[tid];
    const uint32_t runs = (macMode != 0u && c.st.P <= kTileHist) ? 1u : kMacParts;      // the matrix-core MAC sums all partitions in one run
    for (uint32_t q = 1; q < runs; ++q) y += ((gcf4p)(c.ysum + ((size_t)q * maxBatch + j) * 512u))[tid];       // the runs' partial sums
    const c2 acc0 = mk(y.x, y.y), acc1 = mk(y.z, y.w);
    const uint32_t k0 = 2u * tid, k1 = k0 + 1u;
    if (tid == 0u) { A[0] = mk(acc0.x, 0.0f); A[512] = mk(acc0.y, 0.0f); }
    else { A[k0] = mk(acc0.x, -acc0.y); A[1024u - k0] = acc0; }
    A[k1] = mk(acc1.x, -acc1.y); A[1024u - k1] = acc1;
    __syncthreads();
    fft1024(A, B, W, tid);
    for (uint32_t i = tid; i < 512u; i += 256u) {
        out[i] = B[i].x;
        c.tails[(size_t)(j + 1u) * 512u + i] = B[512u + i].x;
    }
}

__global__ __launch_bounds__(256)
void elemhip_convolve_batch_finish(PlanView pv, uint32_t* recs, float* hbm, const Globals* g, uint32_t workBegin,
                                   uint32_t arenaFloats, float* scratchAll, uint32_t maxBatch, uint32_t longMode, size_t perNode) {
    const uint32_t convIdx = pv.convWork[workBegin + blockIdx.x] & 0xFFFFu, j = blockIdx.y, tid = threadIdx.x, batch = gridDim.y;
    const ConvDesc d = pv.convs[convIdx];
    if (!root_running((gcup)recs, d.rootRec, g->numOut)) return;
    BatchCtx c;
    if (!batch_ctx(d, (gup)recs, g, scratchAll, convIdx, maxBatch, c, perNode)) return;
    if (conv_use_long(c.st, batch, longMode)) return;
    const uint32_t stride = g->blockStride, P = c.st.P, b0 = ((gcup)c.scratch)[0];
    gfp out = (gfp)hbm + (size_t)j * arenaFloats + (size_t)d.outHbm * stride;
    // the node's last input blocks in the time domain (a later launch set may take the long-partition path: conv_long.inc)
    if (const uint32_t R = c.st.hdr[conv::H_HISTBLKS]) {
        if (j + R >= batch) {
            gcfp in = (gcfp)(hbm + (size_t)j * arenaFloats + (size_t)(c.inKind == 1u ? c.inBuf : 0u) * stride);
            gfp hist = c.st.overlap + 512u + (size_t)((b0 + j) % R) * 512u;
            for (uint32_t i = tid; i < 512u; i += 256u) hist[i] = c.inKind == 1u ? in[i] : c.cval;
        }
    }
    for (uint32_t i = tid; i < 512u; i += 256u) {
        const float y = out[i] + c.tails[(size_t)j * 512u + i];
        out[i] = d.fuseRootRec == kNone ? y : y * c.gain;
    }
    if (j + P >= batch)   // the newest min(B, P) spectra are the ones later blocks read
        for (uint32_t k = tid; k < 512u; k += 256u) c.st.X[(size_t)((b0 + j) % P) * 512u + k] = c.xnew[(size_t)j * 512u + k];
    if (j + 1u == batch) {
        for (uint32_t i = tid; i < 512u; i += 256u) c.st.overlap[i] = c.tails[(size_t)batch * 512u + i];
        if (tid == 0u) {
            c.st.hdr[conv::H_BLK] = b0 + batch; c.st.hdr[conv::H_BLK_NEXT] = b0 + batch;
            c.st.hdr[conv::H_FILL] = 0u; c.st.hdr[conv::H_FILL_NEXT] = 0u;
        }
    }
}

#include "conv_long.inc"

namespace elemhip {

void launch_convolve(hipStream_t s, const PlanView& pv, uint32_t* recs, float* hbm, const Globals* g,
                     uint32_t workBegin, uint32_t numWorkgroups) {
    hipLaunchKernelGGL(elemhip_convolve_kernel, dim3(numWorkgroups), dim3(256), 0, s, pv, recs, hbm, g, workBegin);
}

size_t convolve_batch_scratch_floats(uint32_t maxBatch, uint32_t longHistRows) {
    return batch_scratch_floats_512(maxBatch) + (longHistRows ? long_scratch_floats(maxBatch, longHistRows) : 0u);
}

// One launch set of the convolve nodes of a level. `longHistRows` != 0: nodes with long-partition spectra render whole sets of a
// multiple of 8 blocks through the long-partition kernels (conv_long.inc; longHistRows = the most tap groups any IR has, - 1);
// `anyShortPath`: some node of the level (or this set's size) still needs the 512-partition kernels.
void launch_convolve_batch(hipStream_t s, const PlanView& pv, uint32_t* recs, float* hbm, const Globals* g, uint32_t workBegin,
                           uint32_t numNodes, uint32_t batch, uint32_t arenaFloats, float* scratch, uint32_t maxBatch, uint32_t macMode,
                           bool anyShortIr, bool anyLongIr, uint32_t longHistRows, bool anyShortPath, uint32_t longStateBlocks, uint32_t longMacMode,
                           const float* inDirect, uint32_t numInCh, float* outDirect, uint32_t numOutCh) {
    const dim3 grid(numNodes, batch), block(256);
    const size_t perNode = convolve_batch_scratch_floats(maxBatch, longHistRows);
    const bool longSet = longHistRows != 0u && batch >= 8u && (batch & 7u) == 0u;
    const uint32_t longMode = longSet ? 1u : 0u;
    if (!longSet || anyShortPath) {
    hipLaunchKernelGGL(elemhip_convolve_batch_fft, grid, block, 0, s, pv, recs, hbm, g, workBegin, arenaFloats, scratch, maxBatch, macMode, longMode, perNode);
    // the partition sums: nodes whose IR has at most kTileHist partitions on the matrix cores (macMode != 0), the others through the
    // vector kernel — (node, 16-bin tile, 64-block chunk x partition run)
    if (macMode != 0u && anyShortIr) {
        // persistent over the block axis: as many runs per workgroup as leave about two workgroups per CU (256 CUs on this chip)
        const uint32_t runs = (batch + kMfmaOut - 1u) / kMfmaOut, pairs = numNodes * (conv::kBlock / 16u);
        uint32_t per = (runs * pairs) / 512u;      // (two workgroups per CU: LDS)
        per = per < 1u ? 1u : (per > runs ? runs : per);
        hipLaunchKernelGGL(elemhip_convolve_batch_mac_mfma, dim3(numNodes, conv::kBlock / 16u, (runs + per - 1u) / per), block, 0, s, pv, recs, hbm, g, workBegin, arenaFloats, scratch, maxBatch, batch, macMode, per, longMode, perNode);
    }
    if (macMode == 0u || anyLongIr)
    hipLaunchKernelGGL(elemhip_convolve_batch_mac, dim3(numNodes, conv::kBlock / 16u, kMacParts * ((batch + 63u) / 64u)), block, 0, s, pv, recs, hbm, g, workBegin, arenaFloats, scratch, maxBatch, batch, macMode, longMode, perNode);
    hipLaunchKernelGGL(elemhip_convolve_batch_ifft, grid, block, 0, s, pv, recs, hbm, g, workBegin, arenaFloats, scratch, maxBatch, macMode, longMode, perNode);
    hipLaunchKernelGGL(elemhip_convolve_batch_finish, grid, block, 0, s, pv, recs, hbm, g, workBegin, arenaFloats, scratch, maxBatch, longMode, perNode);
    }
    if (longSet) {
        const uint32_t chunks = batch / 8u;
        hipLaunchKernelGGL(elemhip_convolve_long_fft, dim3(numNodes, longHistRows + chunks), block, 0, s, pv, recs, hbm, g, workBegin, arenaFloats, scratch, maxBatch, batch, longHistRows, longMode, perNode, inDirect, numInCh);
        // the partition sums: LDS-tiled (mode 1) while a tile's rows fit 64 KB of LDS (C3: 40 KB), else the register kernel over L2
        const size_t macLds = long_mac_tile_lds_bytes(longHistRows + 1u);      // (Qp <= histRows + 1)
        if (macLds <= 64u * 1024u && longMacMode == 1u)
            hipLaunchKernelGGL(elemhip_convolve_long_mac_lds, dim3(numNodes, lfft::M / kTileBins, (chunks + kTileRun - 1u) / kTileRun), dim3(512), macLds, s, pv, recs, hbm, g, workBegin, arenaFloats, scratch, maxBatch, batch, longHistRows, longMode, perNode);
        else if (longMacMode == 2u)      // (A/B: runs of 32 chunks per thread — every spectrum row is read by 1.7 workgroups instead of 2.4, at two waves per SIMD)
        hipLaunchKernelGGL(elemhip_convolve_long_mac<32u>, dim3(numNodes, lfft::M / 256u, (chunks + 31u) / 32u), block, 0, s, pv, recs, hbm, g, workBegin, arenaFloats, scratch, maxBatch, batch, longHistRows, longMode, perNode, 0u);
        else
        hipLaunchKernelGGL(elemhip_convolve_long_mac<kLongRun>, dim3(numNodes, lfft::M / 256u, (chunks + kLongRun - 1u) / kLongRun), block, 0, s, pv, recs, hbm, g, workBegin, arenaFloats, scratch, maxBatch, batch, longHistRows, longMode, perNode, longMacMode == 3u ? 1u : 0u);
        hipLaunchKernelGGL(elemhip_convolve_long_ifft, dim3(numNodes, chunks), block, 0, s, pv, recs, hbm, g, workBegin, arenaFloats, scratch, maxBatch, batch, longHistRows, longMode, perNode, inDirect, numInCh, outDirect, numOutCh);
        (void)longStateBlocks;      // (the 512-partition state — spectra ring, overlap — is made on demand: launch_convolve_fix_overlap)
    }
}

// before the next 512-partition evaluation (a block-at-a-time launch, a set that does not take the long partitions) of convolve nodes
// a long-partition set rendered last: their overlap (conv_long.inc, elemhip_convolve_long_tail mode 1). `numWork` entries of the
// plan's conv work list from `workBegin` on are looked at; helper entries and nodes whose overlap is current return at once.
void launch_convolve_fix_overlap(hipStream_t s, const PlanView& pv, uint32_t* recs, float* hbm, const Globals* g, uint32_t workBegin,
                                 uint32_t numWork, float* scratch, uint32_t maxBatch, uint32_t longHistRows, uint32_t maxPartitions) {
    if (!numWork) return;
    const size_t perNode = convolve_batch_scratch_floats(maxBatch, longHistRows);
    // the 1024-point spectra of the last P blocks from the input ring, then the overlap of the last block
    hipLaunchKernelGGL(elemhip_convolve_long_state, dim3(numWork, maxPartitions ? maxPartitions : 1u), dim3(256), 0, s, pv, recs, hbm, g, workBegin, scratch, maxBatch, perNode);
    hipLaunchKernelGGL(elemhip_convolve_long_tail, dim3(numWork), dim3(256), 0, s, pv, recs, hbm, g, workBegin, scratch, maxBatch, perNode);
}

uint32_t convolve_long_tap_group() { return kLongTaps; }
uint32_t convolve_long_row_floats() { return kRow * 2u; }

uint32_t convolve_mfma_max_partitions() { return kTileHist; }

hipError_t upload_convolve_tables(const float* twiddleReIm /* 2 * 1024 floats */, const float* twiddle8192ReIm /* 2 * 8192 floats */) {
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(kTwiddle), twiddleReIm, sizeof(float) * 2 * conv::kFft);
    if (e != hipSuccess) return e;
    return hipMemcpyToSymbol(HIP_SYMBOL(kTw8192), twiddle8192ReIm, sizeof(float) * 2 * lfft::N);
}

} // namespace elemhip
