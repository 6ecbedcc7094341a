// kernels.hip — the block-render path as hand-written HIP for gfx950 (CDNA4).
//
// One launch renders one *island level*: every workgroup (4 wavefronts) interprets the task
// list of one island with that island's block buffers resident in LDS.  Stateless node loops
// (runtime/elem/builtins/Math.h etc.) run sample-parallel, 64 lanes x (samples/64); stateful
// recurrences (phasor, polyBLEP phase, one-pole, biquad, SVF, ...) run one node per lane with
// only the loop-carried state update on the serial chain and everything else hoisted into
// sample-parallel pre/post passes.  An epilogue workgroup then sums root buffers into the
// output bus, advances root fades and promotes feedback taps (GraphRenderSequence.h:268-309).
//
// PARITY RULES: compiled with -ffp-contract=off; float-state recurrences are
// op-for-op the reference's expressions, in the reference's order; nodes that compute in double
// in the reference (SVF, shelf, mm1p, prewarp) compute in double here.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include "device.h"
#include "launch.h"

using namespace elemhip;

extern __shared__ __attribute__((aligned(16))) float lds[];

namespace {

struct Ctx {
    uint32_t*       recs;
    float*          hbm;
    const Globals*  g;
    const uint32_t* operands;
    const uint32_t* lcg;      // [2*(kMaxBlock+1)] jump-ahead table for `rand`
    uint32_t        n;        // frames this block
    uint32_t        stride;   // floats per arena buffer
    uint32_t        numIn;    // host input channels
    uint32_t        lane;
};

#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ float    u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f)    { return __float_as_uint(f); }

__device__ __forceinline__ double rec_ld_f64(const uint32_t* r, uint32_t d) {
    return __hiloint2double((int)r[d + 1], (int)r[d]);
}
__device__ __forceinline__ void rec_st_f64(uint32_t* r, uint32_t d, double v) {
    r[d] = (uint32_t)__double2loint(v); r[d + 1] = (uint32_t)__double2hiint(v);
}
__device__ __forceinline__ float* rec_ptr(const uint32_t* r, uint32_t d) {
    uint64_t p = (uint64_t)r[d] | ((uint64_t)r[d + 1] << 32);
    return reinterpret_cast<float*>(p);
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) {   // std::clamp
    return (v < lo) ? lo : ((hi < v) ? hi : v);
}
__device__ __forceinline__ double clampd(double v, double lo, double hi) {
    return (v < lo) ? lo : ((hi < v) ? hi : v);
}

// ---- operand access ---------------------------------------------------------------------
__device__ __forceinline__ uint32_t member_nin(const Ctx& c, const Member& m) {
    return m.nin == kNone ? c.numIn : m.nin;
}
__device__ __forceinline__ uint32_t member_opnd(const Ctx& c, const Member& m, uint32_t k) {
    return c.operands[m.opnd + k];
}

// sample-parallel fetch: `o` is wave-uniform, so the kind switch is a scalar branch
__device__ __forceinline__ float fetch(const Ctx& c, uint32_t o, uint32_t i) {
    const uint32_t kind = o & kOpKindMask, v = o & kOpValMask;
    if (kind == kOpLds)   return lds[v + i];
    if (kind == kOpConst) return lds[v];
    if (kind == kOpHbm)   return c.hbm[(size_t)v * c.stride + i];
    return 0.0f;
}

__device__ __forceinline__ void put(const Ctx& c, const Member& m, uint32_t i, float y) {
    if (m.outLds != kNone) lds[m.outLds + i] = y;
    if (m.outHbm != kNone) c.hbm[(size_t)m.outHbm * c.stride + i] = y;
}

__device__ __forceinline__ void zero_fill(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) put(c, m, i, 0.0f);
}

// lane-per-node access: every operand of a serial op lives in LDS (the planner imports HBM
// operands first); value(t) = lds[base + t * step]
struct SIn { uint32_t base, step; };
__device__ __forceinline__ SIn sin_of(uint32_t o) {
    const uint32_t kind = o & kOpKindMask, v = o & kOpValMask;
    if (kind == kOpLds)   return SIn{v, 1u};
    if (kind == kOpConst) return SIn{v, 0u};
    return SIn{0u, 0u};   // LDS word 0 is kept at 0.0f
}
__device__ __forceinline__ float sget(SIn s, uint32_t t) { return lds[s.base + t * s.step]; }

// ---- stateless math (Math.h) -------------------------------------------------------------
__device__ __forceinline__ float unary_eval(uint16_t op, float x) {
    switch (op) {
        case OP_SIN:   return sinf(x);
        case OP_COS:   return cosf(x);
        case OP_TAN:   return tanf(x);
        case OP_TANH:  return tanhf(x);
        case OP_ASINH: return asinhf(x);
        case OP_LN:    return logf(x);
        case OP_LOG:   return log10f(x);
        case OP_LOG2:  return log2f(x);
        case OP_CEIL:  return ceilf(x);
        case OP_FLOOR: return floorf(x);
        case OP_ROUND: return roundf(x);
        case OP_SQRT:  return sqrtf(x);
        case OP_EXP:   return expf(x);
        default:       return fabsf(x);   // OP_ABS
    }
}

__device__ __forceinline__ float binary_eval(uint16_t op, float x, float y) {
    switch (op) {
        case OP_LE:  return (x < y)  ? 1.0f : 0.0f;
        case OP_LEQ: return (x <= y) ? 1.0f : 0.0f;
        case OP_GE:  return (x > y)  ? 1.0f : 0.0f;
        case OP_GEQ: return (x >= y) ? 1.0f : 0.0f;
        case OP_POW: // SafePow, Math.h:179-188
            if (x < 0.0f && y != floorf(y)) return 0.0f;
            return powf(x, y);
        case OP_EQ:  return (fabsf(x - y) <= FLT_EPSILON) ? 1.0f : 0.0f;
        case OP_AND: return (fabsf(1.0f - x) <= FLT_EPSILON && fabsf(1.0f - y) <= FLT_EPSILON) ? 1.0f : 0.0f;
        default:     return (fabsf(1.0f - x) <= FLT_EPSILON || fabsf(1.0f - y) <= FLT_EPSILON) ? 1.0f : 0.0f; // OP_OR
    }
}

__device__ __forceinline__ float reduce_eval(uint16_t op, float a, float b) {
    switch (op) {
        case OP_ADD: return a + b;
        case OP_SUB: return a - b;
        case OP_MUL: return a * b;
        case OP_DIV: return (b == 0.0f) ? 0.0f : a / b;   // SafeDivides, Math.h:135-140
        case OP_MOD: return fmodf(a, b);
        case OP_MIN: return (b < a) ? b : a;               // std::min
        default:     return (a < b) ? b : a;               // std::max
    }
}

// One switch per task, not per sample: the per-sample loops below are specialised by template.
template <uint16_t OPC>
__device__ void run_unary(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    if (member_nin(c, m) < 1) return zero_fill(c, m, s0, s1);
    const uint32_t o = member_opnd(c, m, 0);
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) put(c, m, i, unary_eval(OPC, fetch(c, o, i)));
}

template <uint16_t OPC>
__device__ void run_binary(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    if (member_nin(c, m) < 2) return zero_fill(c, m, s0, s1);
    const uint32_t o0 = member_opnd(c, m, 0), o1 = member_opnd(c, m, 1);
    for (uint32_t i = s0 + c.lane; i < s1; i += 64)
        put(c, m, i, binary_eval(OPC, fetch(c, o0, i), fetch(c, o1, i)));
}

// BinaryReducingNode (Math.h:59-89): strict left fold over the children, any fan-in.
template <uint16_t OPC>
__device__ void run_reduce(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    uint32_t nin = member_nin(c, m);
    if (nin < 1) return zero_fill(c, m, s0, s1);
    if (m.nin == kNone && nin > kMaxHostIn) nin = kMaxHostIn;
    float acc[8];
    const uint32_t o0 = member_opnd(c, m, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t i = s0 + c.lane + 64u * j;
        acc[j] = (i < s1) ? fetch(c, o0, i) : 0.0f;
    }
    for (uint32_t k = 1; k < nin; ++k) {
        const uint32_t o = member_opnd(c, m, k);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t i = s0 + c.lane + 64u * j;
            if (i < s1) acc[j] = reduce_eval(OPC, acc[j], fetch(c, o, i));
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t i = s0 + c.lane + 64u * j;
        if (i < s1) put(c, m, i, acc[j]);
    }
}

// IdentityNode `in` (Math.h:92-126): out = inputData[channel]; leaf => host input channel.
__device__ void run_in(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    const uint32_t ch = c.recs[m.rec * kRecDwords + rec::P0];   // static_cast<size_t>(int): negatives wrap large
    const uint32_t nin = member_nin(c, m);
    const bool neg = (int32_t)ch < 0;
    if (neg || ch >= nin || (m.nin == kNone && ch >= kMaxHostIn)) return zero_fill(c, m, s0, s1);
    const uint32_t o = member_opnd(c, m, ch);
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) put(c, m, i, fetch(c, o, i));
}

__device__ void run_copy(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    const uint32_t o = member_opnd(c, m, 0);
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) put(c, m, i, fetch(c, o, i));
}

// RootNode (Core.h:66-78) + GainFade::process (helpers/GainFade.h:56-72). The gain itself is
// advanced once per block by the epilogue, after every island has read it.
__device__ void run_root(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    if (member_nin(c, m) < 1) return zero_fill(c, m, s0, s1);
    const uint32_t* r = c.recs + m.rec * kRecDwords;
    const float g = u2f(r[rec::ROOT_GAIN]), tg = u2f(r[rec::ROOT_TARGET]), step = u2f(r[rec::ROOT_STEP]);
    const uint32_t o = member_opnd(c, m, 0);
    if (g == tg) {
        for (uint32_t i = s0 + c.lane; i < s1; i += 64) put(c, m, i, fetch(c, o, i) * tg);
    } else {
        for (uint32_t i = s0 + c.lane; i < s1; i += 64)
            put(c, m, i, fetch(c, o, i) * clampf(g + step * (float)(int)i, 0.0f, 1.0f));
    }
}

// CutoffPrewarpNode (filters/MultiMode1p.h:9-36): double internals.
__device__ void run_prewarp(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    if (member_nin(c, m) < 1) return zero_fill(c, m, s0, s1);
    const double T = 1.0 / c.g->sampleRate;
    const uint32_t o = member_opnd(c, m, 0);
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) {
        const double twoPi = 2.0 * 3.141592653589793238;
        const double wd = twoPi * (double)fetch(c, o, i);
        put(c, m, i, (float)tan(wd * T / 2.0));
    }
}

// SampleTimeNode (wasm/SampleTime.h:11-24), MetronomeNode (wasm/Metro.h:40-55)
__device__ void run_time(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    const int64_t st = c.g->sampleTime;
    for (uint32_t i = s0 + c.lane; i < s1; i += 64)
        put(c, m, i, (float)(double)((uint64_t)st + (uint64_t)i));
}
__device__ void run_metro(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    const uint32_t* r = c.recs + m.rec * kRecDwords;
    const int64_t is64 = (int64_t)((uint64_t)r[rec::P0] | ((uint64_t)r[rec::P1] << 32));
    const double is = (double)is64;
    const int64_t st = c.g->sampleTime;
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) {
        const double t = (double)((uint64_t)st + (uint64_t)i) / is;
        put(c, m, i, ((t - floor(t)) < 0.5) ? 1.0f : 0.0f);
    }
}

// UniformRandomNoiseNode (Noise.h:9-43): the LCG is affine mod 2^32, so sample i is an exact
// jump-ahead  s_{i+1} = A[i+1]*s_0 + C[i+1]  from a precomputed table — bit-identical, no chain.
__device__ void run_rand(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    const uint32_t seed = r[rec::S0];
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) {
        const uint32_t s = c.lcg[2 * (i + 1)] * seed + c.lcg[2 * (i + 1) + 1];
        put(c, m, i, (float)(int)((s >> 16) & 0x7FFF) / 32767.0f);
    }
    WAVE_SYNC();
    if (c.lane == 0) r[rec::S0] = c.lcg[2 * c.n] * seed + c.lcg[2 * c.n + 1];
}

// SingleSampleDelayNode (Delays.h:15-39): out[i] = (i ? in[i-1] : z); z = in[n-1]
__device__ void run_z(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    if (member_nin(c, m) < 1) return zero_fill(c, m, s0, s1);
    uint32_t* r = c.recs + m.rec * kRecDwords;
    const uint32_t o = member_opnd(c, m, 0);
    const float z = u2f(r[rec::S0]);
    const float last = (c.n > 0) ? fetch(c, o, c.n - 1) : z;
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) put(c, m, i, i ? fetch(c, o, i - 1) : z);
    WAVE_SYNC();
    if (c.lane == 0) r[rec::S0] = f2u(last);
}

// TapInNode / TapOutNode (Feedback.h:40-53, 111-126); buffers are always float.
__device__ void run_tapin(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    const uint32_t* r = c.recs + m.rec * kRecDwords;
    const float* shared = rec_ptr(r, rec::TAP_SHARED);
    if (!shared) return zero_fill(c, m, s0, s1);
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) put(c, m, i, shared[i]);
}
__device__ void run_tapout(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    const uint32_t* r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 1) return zero_fill(c, m, s0, s1);
    float* priv = rec_ptr(r, rec::TAP_PRIVATE);
    const uint32_t o = member_opnd(c, m, 0);
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) {
        const float x = fetch(c, o, i);
        priv[i] = x;
        put(c, m, i, x);
    }
}

// SampleDelayNode (Delays.h:177-272). The reference writes the block into the ring and then
// reads ring[(size + w0 - len + i) & mask]; for i >= len that is this block's in[i-len], for
// i < len it is older ring data the block's own writes cannot touch (size >= len + blockSize).
__device__ void run_sdelay(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    float* ring = rec_ptr(r, rec::RING_PTR);
    const int size = (int)r[rec::RING_SIZE];
    const int len  = (int)r[rec::RING_LEN];
    int w0 = (int)r[rec::RING_WRITE];
    if (r[rec::RING_RESET]) w0 = 0;
    if (member_nin(c, m) < 1 || size == 0 || ring == nullptr) {
        WAVE_SYNC();
        if (c.lane == 0) { r[rec::RING_RESET] = 0; r[rec::RING_WRITE] = (uint32_t)w0; }
        return zero_fill(c, m, s0, s1);
    }
    const int mask = size - 1;
    const uint32_t o = member_opnd(c, m, 0);
    const int readStart = w0 - len;
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) {
        float y;
        if (len >= 0 && (int)i >= len) y = fetch(c, o, i - (uint32_t)len);
        else                           y = ring[(size + readStart + (int)i) & mask];
        put(c, m, i, y);
    }
    WAVE_SYNC();
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) ring[(w0 + (int)i) & mask] = fetch(c, o, i);
    WAVE_SYNC();
    if (c.lane == 0) { r[rec::RING_RESET] = 0; r[rec::RING_WRITE] = (uint32_t)((w0 + (int)c.n) & mask); }
}

// VariableDelayNode (Delays.h:51-169). Every lane replays the write-index walk; when the
// smallest read offset in the block exceeds the block length no read can observe a write of
// this block, so reads/writes are sample-parallel; otherwise lane 0 walks the block serially.
__device__ void run_delay(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    float* ring = rec_ptr(r, rec::RING_PTR);
    const int size = (int)r[rec::RING_SIZE];
    int w0 = (int)r[rec::RING_WRITE];
    if (r[rec::RING_RESET]) w0 = 0;
    const uint32_t nin = member_nin(c, m);
    if (nin < 3) {
        WAVE_SYNC();
        if (c.lane == 0) { r[rec::RING_RESET] = 0; r[rec::RING_WRITE] = (uint32_t)w0; }
        return zero_fill(c, m, s0, s1);
    }
    const uint32_t oLen = member_opnd(c, m, 0), oFb = member_opnd(c, m, 1), oX = member_opnd(c, m, 2);
    if (size == 0 || ring == nullptr) {   // Delays.h:106-107 copies input 0
        for (uint32_t i = s0 + c.lane; i < s1; i += 64) put(c, m, i, fetch(c, oLen, i));
        WAVE_SYNC();
        if (c.lane == 0) { r[rec::RING_RESET] = 0; r[rec::RING_WRITE] = (uint32_t)w0; }
        return;
    }
    const float fsize = (float)size;
    // smallest clamped offset over the block
    float mn = FLT_MAX;
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) mn = fminf(mn, clampf(fetch(c, oLen, i), 0.0f, fsize));
    for (int d = 32; d >= 1; d >>= 1) mn = fminf(mn, __shfl_xor(mn, d));
    const bool parallel = (mn >= (float)(c.n + 2)) && ((uint32_t)size >= c.n);
    if (parallel) {
        for (uint32_t i = s0 + c.lane; i < s1; i += 64) {
            int w = w0 + (int)i; if (w >= size) w -= size;       // size >= n: at most one wrap
            const float offset = clampf(fetch(c, oLen, i), 0.0f, fsize);
            const float readFrac = (float)(size + w) - offset;
            const int readLeft = (int)readFrac;
            const int readRight = readLeft + 1;
            const float frac = readFrac - floorf(readFrac);
            const float left = ring[readLeft % size];
            const float right = ring[readRight % size];
            const float out = left + frac * (right - left);
            const float fb = clampf(fetch(c, oFb, i), -1.0f, 1.0f);
            const float in = fetch(c, oX, i) + fb * out;
            put(c, m, i, out);
            // all reads of the block are ordered before any write by the WAVE_SYNC below
            lds[m.scratch + i] = in;
        }
        WAVE_SYNC();
        for (uint32_t i = s0 + c.lane; i < s1; i += 64) {
            int w = w0 + (int)i; if (w >= size) w -= size;
            ring[w] = lds[m.scratch + i];
        }
        WAVE_SYNC();
        if (c.lane == 0) {
            int w = w0 + (int)c.n; if (w >= size) w -= size;
            r[rec::RING_RESET] = 0; r[rec::RING_WRITE] = (uint32_t)w;
        }
        return;
    }
    if (c.lane == 0) {
        int w = w0;
        for (uint32_t i = s0; i < s1; ++i) {
            const float offset = clampf(fetch(c, oLen, i), 0.0f, fsize);
            if (offset <= FLT_EPSILON) {
                const float in = fetch(c, oX, i);
                ring[w] = in;
                put(c, m, i, in);
                if (++w >= size) w -= size;
                continue;
            }
            const float readFrac = (float)(size + w) - offset;
            const int readLeft = (int)readFrac;
            const int readRight = readLeft + 1;
            const float frac = readFrac - floorf(readFrac);
            const float left = __builtin_nontemporal_load(&ring[readLeft % size]);
            const float right = __builtin_nontemporal_load(&ring[readRight % size]);
            const float out = left + frac * (right - left);
            const float fb = clampf(fetch(c, oFb, i), -1.0f, 1.0f);
            const float in = fetch(c, oX, i) + fb * out;
            __builtin_nontemporal_store(in, &ring[w]);
            put(c, m, i, out);
            if (++w >= size) w -= size;
        }
        r[rec::RING_RESET] = 0; r[rec::RING_WRITE] = (uint32_t)w;
    }
    WAVE_SYNC();
}

// ---- lane-per-node recurrences ---------------------------------------------------------------
// Each helper below is entered by lanes [0, count) of the task's wave; `m` is that lane's node.
// Loads of a chunk are issued before the dependent chain so LDS latency is paid once per chunk.
constexpr int CH = 8;

__device__ __forceinline__ void szero(const Member& m, uint32_t n) {
    for (uint32_t t = 0; t < n; ++t) lds[m.outLds + t] = 0.0f;
}

// PhasorNode (Core.h:85-136): step = f * (1/sr) in float; phase = next - floor(next)
template <bool WithReset>
__device__ void ser_phasor(const Ctx& c, const Member& m) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    const uint32_t nin = member_nin(c, m);
    if (nin < (WithReset ? 2u : 1u)) return szero(m, c.n);
    const SIn f = sin_of(member_opnd(c, m, 0));
    const SIn rs = WithReset ? sin_of(member_opnd(c, m, 1)) : SIn{0, 0};
    float phase = u2f(r[rec::S0]);
    float lastIn = u2f(r[rec::S1]);
    const float rsr = 1.0f / c.g->sampleRateF;
    for (uint32_t t0 = 0; t0 < c.n; t0 += CH) {
        float fq[CH], rv[CH], y[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const uint32_t t = min(t0 + j, c.n - 1);
            fq[j] = sget(f, t) * rsr;
            if (WithReset) rv[j] = sget(rs, t);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (WithReset && t0 + j < c.n) {
                const float dt = rv[j] - lastIn;     // Change::tick (helpers/Change.h:20-31)
                lastIn = rv[j];
                if (dt > 0.0f) phase = 0.0f;          // change(...) > 0.5  <=>  dt > 0
            }
            y[j] = phase;
            const float next = phase + fq[j];
            const float np = next - floorf(next);
            if (t0 + j < c.n) phase = np;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) if (t0 + j < c.n) lds[m.outLds + t0 + j] = y[j];
    }
    r[rec::S0] = f2u(phase);
    if (WithReset) r[rec::S1] = f2u(lastIn);
}

// OnePoleNode (Filters.h:13-39): z = x + p*z
__device__ void ser_pole(const Ctx& c, const Member& m) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 2) return szero(m, c.n);
    const SIn ps = sin_of(member_opnd(c, m, 0)), xs = sin_of(member_opnd(c, m, 1));
    float z = u2f(r[rec::S0]);
    for (uint32_t t0 = 0; t0 < c.n; t0 += CH) {
        float p[CH], x[CH], y[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) { const uint32_t t = min(t0 + j, c.n - 1); p[j] = sget(ps, t); x[j] = sget(xs, t); }
#pragma unroll
        for (int j = 0; j < CH; ++j) { const float nz = x[j] + p[j] * z; if (t0 + j < c.n) z = nz; y[j] = z; }
#pragma unroll
        for (int j = 0; j < CH; ++j) if (t0 + j < c.n) lds[m.outLds + t0 + j] = y[j];
    }
    r[rec::S0] = f2u(z);
}

// EnvelopeNode (Filters.h:46-79)
__device__ void ser_env(const Ctx& c, const Member& m) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 3) return szero(m, c.n);
    const SIn as = sin_of(member_opnd(c, m, 0)), rs = sin_of(member_opnd(c, m, 1)), xs = sin_of(member_opnd(c, m, 2));
    float z = u2f(r[rec::S0]);
    for (uint32_t t0 = 0; t0 < c.n; t0 += CH) {
        float ap[CH], rp[CH], vn[CH], y[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const uint32_t t = min(t0 + j, c.n - 1);
            ap[j] = sget(as, t); rp[j] = sget(rs, t); vn[j] = fabsf(sget(xs, t));
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float k = (fabsf(vn[j]) > z) ? ap[j] : rp[j];
            const float nz = k * (z - vn[j]) + vn[j];
            if (t0 + j < c.n) z = nz;
            y[j] = z;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) if (t0 + j < c.n) lds[m.outLds + t0 + j] = y[j];
    }
    r[rec::S0] = f2u(z);
}

// BiquadFilterNode (Filters.h:87-120), TDF-II with coefficient signals
__device__ void ser_biquad(const Ctx& c, const Member& m) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 6) return szero(m, c.n);
    SIn s[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = sin_of(member_opnd(c, m, k));
    float z1 = u2f(r[rec::S0]), z2 = u2f(r[rec::S1]);
    for (uint32_t t0 = 0; t0 < c.n; t0 += CH) {
        float b0x[CH], b1x[CH], b2x[CH], a1[CH], a2[CH], y[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const uint32_t t = min(t0 + j, c.n - 1);
            const float x = sget(s[5], t);
            b0x[j] = sget(s[0], t) * x; b1x[j] = sget(s[1], t) * x; b2x[j] = sget(s[2], t) * x;
            a1[j] = sget(s[3], t); a2[j] = sget(s[4], t);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float yy = b0x[j] + z1;
            const float nz1 = b1x[j] - a1[j] * yy + z2;
            const float nz2 = b2x[j] - a2[j] * yy;
            if (t0 + j < c.n) { z1 = nz1; z2 = nz2; }
            y[j] = yy;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) if (t0 + j < c.n) lds[m.outLds + t0 + j] = y[j];
    }
    r[rec::S0] = f2u(z1); r[rec::S1] = f2u(z2);
}

// CounterNode / AccumNode / LatchNode / MaxHold / OnceNode (Core.h:183-404)
__device__ void ser_counter(const Ctx& c, const Member& m) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 1) return szero(m, c.n);
    const SIn gs = sin_of(member_opnd(c, m, 0));
    float count = u2f(r[rec::S0]);
    for (uint32_t t = 0; t < c.n; ++t) {
        const float in = sget(gs, t);
        if ((1.0f - in) <= FLT_EPSILON) { lds[m.outLds + t] = count; count = count + 1.0f; }
        else { count = 0.0f; lds[m.outLds + t] = 0.0f; }
    }
    r[rec::S0] = f2u(count);
}

__device__ __forceinline__ float change_tick(float& lastIn, float xn) {   // helpers/Change.h:20-31
    const float dt = xn - lastIn;
    lastIn = xn;
    return (dt > 0.0f) ? 1.0f : ((dt < 0.0f) ? -1.0f : 0.0f);
}

__device__ void ser_accum(const Ctx& c, const Member& m) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 2) return szero(m, c.n);
    const SIn xs = sin_of(member_opnd(c, m, 0)), rs = sin_of(member_opnd(c, m, 1));
    float total = u2f(r[rec::S0]), lastIn = u2f(r[rec::S1]);
    for (uint32_t t = 0; t < c.n; ++t) {
        if (change_tick(lastIn, sget(rs, t)) > 0.5f) total = 0.0f;
        total += sget(xs, t);
        lds[m.outLds + t] = total;
    }
    r[rec::S0] = f2u(total); r[rec::S1] = f2u(lastIn);
}

__device__ void ser_latch(const Ctx& c, const Member& m) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 2) return szero(m, c.n);
    const SIn ls = sin_of(member_opnd(c, m, 0)), xs = sin_of(member_opnd(c, m, 1));
    float z = u2f(r[rec::S0]), hold = u2f(r[rec::S1]);
    for (uint32_t t = 0; t < c.n; ++t) {
        const float l = sget(ls, t), x = sget(xs, t);
        if (fabsf(z) <= FLT_EPSILON && l > FLT_EPSILON) hold = x;
        z = l;
        lds[m.outLds + t] = hold;
    }
    r[rec::S0] = f2u(z); r[rec::S1] = f2u(hold);
}

__device__ void ser_maxhold(const Ctx& c, const Member& m) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 2) return szero(m, c.n);
    const SIn xs = sin_of(member_opnd(c, m, 0)), rs = sin_of(member_opnd(c, m, 1));
    const uint32_t hts = r[rec::P0];
    float lastIn = u2f(r[rec::S0]); uint32_t at = r[rec::S1]; float mx = u2f(r[rec::S2]);
    for (uint32_t t = 0; t < c.n; ++t) {
        const float in = sget(xs, t), reset = sget(rs, t);
        if (change_tick(lastIn, reset) > 0.5f || ++at >= hts) { mx = in; at = 0; }
        else if (in > mx) { at = 0; mx = in; }
        lds[m.outLds + t] = mx;
    }
    r[rec::S0] = f2u(lastIn); r[rec::S1] = at; r[rec::S2] = f2u(mx);
}

__device__ void ser_once(const Ctx& c, const Member& m) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 1) return szero(m, c.n);
    const SIn xs = sin_of(member_opnd(c, m, 0));
    const bool isArmed = u2f(r[rec::S2]) != 0.0f;    // atomic<FloatType> armed, loaded once per block
    float gain = u2f(r[rec::S0]), lastIn = u2f(r[rec::S1]);
    bool disarm = false;
    for (uint32_t t = 0; t < c.n; ++t) {
        const float x = sget(xs, t);
        const float delta = change_tick(lastIn, x);
        if (isArmed && delta > 0.5f) { gain = 1.0f; disarm = true; }
        if (delta < -0.5f) gain = 0.0f;
        lds[m.outLds + t] = x * gain;
    }
    r[rec::S0] = f2u(gain); r[rec::S1] = f2u(lastIn);
    if (disarm) r[rec::S2] = f2u(0.0f);
}

// SequenceNode (Core.h:407-573)
__device__ void ser_seq(const Ctx& c, const Member& m) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    const float* seq = rec_ptr(r, rec::SEQ_PTR);
    const uint32_t len = r[rec::SEQ_LEN];
    uint32_t idx = r[rec::SEQ_INDEX];
    float holdValue = u2f(r[rec::SEQ_HOLDVAL]);
    bool first = r[rec::SEQ_FIRST] != 0;
    bool have = r[rec::SEQ_HAVE] != 0;        // activeSequence != nullptr
    if (r[rec::SEQ_PENDING]) {                // a new sequence arrived since the last block (:468-492)
        r[rec::SEQ_PENDING] = 0;
        have = true;
        // size_t % 0 is UB in the reference; an empty sequence leaves the index untouched here
        if (len) idx = idx % len;
        if (first && len) holdValue = seq[idx];
    }
    const uint32_t nin = member_nin(c, m);
    if (nin < 1 || !have) {
        r[rec::SEQ_INDEX] = idx; r[rec::SEQ_HOLDVAL] = f2u(holdValue); r[rec::SEQ_HAVE] = have;
        return szero(m, c.n);
    }
    const bool hasReset = nin > 1;
    const bool hold = r[rec::SEQ_HOLD] != 0, loop = r[rec::SEQ_LOOP] != 0;
    const uint32_t offset = r[rec::SEQ_OFFSET];
    const SIn ts = sin_of(member_opnd(c, m, 0));
    const SIn rs = hasReset ? sin_of(member_opnd(c, m, 1)) : SIn{0, 0};
    float chg = u2f(r[rec::SEQ_CHANGE]), rchg = u2f(r[rec::SEQ_RCHANGE]);
    for (uint32_t t = 0; t < c.n; ++t) {
        const float in = sget(ts, t);
        const float reset = hasReset ? sget(rs, t) : 0.0f;
        if (change_tick(rchg, reset) > 0.5f) idx = offset;
        if (change_tick(chg, in) > 0.5f) {
            // std::min(seqIndex, size - 1): size_t arithmetic, an empty sequence reads nothing here
            if (len) holdValue = seq[min(idx, len - 1)];
            first = true;
            if ((++idx >= len) && loop) idx = 0;
        }
        float y;
        if (idx < len) y = hold ? holdValue : holdValue * in;
        else           y = hold ? holdValue : 0.0f;
        lds[m.outLds + t] = y;
    }
    r[rec::SEQ_INDEX] = idx; r[rec::SEQ_HOLDVAL] = f2u(holdValue); r[rec::SEQ_FIRST] = first;
    r[rec::SEQ_CHANGE] = f2u(chg); r[rec::SEQ_RCHANGE] = f2u(rchg); r[rec::SEQ_HAVE] = 1;
}

// MultiMode1p (filters/MultiMode1p.h:38-113): double state; G = g/(1+g) hoisted into `pre`.
__device__ void ser_mm1p(const Ctx& c, const Member& m) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    const SIn xs = sin_of(member_opnd(c, m, 1));
    const uint32_t mode = r[rec::P0];
    const double* G = reinterpret_cast<const double*>(lds + m.scratch);
    double z = rec_ld_f64(r, rec::S0);
    for (uint32_t t = 0; t < c.n; ++t) {
        const float xn = sget(xs, t);
        const double v = ((double)xn - z) * G[t];
        const double lp = v + z;
        z = lp + v;
        float y;
        if (mode == 0)      y = (float)lp;
        else if (mode == 2) y = xn - (float)lp;
        else                y = (float)(lp + lp - (double)xn);
        lds[m.outLds + t] = y;
    }
    rec_st_f64(r, rec::S0, z);
}

// StateVariableFilterNode::tick (filters/SVF.h:48-70) / shelf (filters/SVFShelf.h:44-64).
// Scratch holds a1,a2,a3 (double) per sample from the coefficient pre-pass; for modes whose
// output needs k (and A) the chain stores v1,v2 back over a1,a2 and a parallel post-pass
// forms the output.
template <bool Direct>
__device__ void ser_svf_chain(const Ctx& c, const Member& m, uint32_t inIdx, uint32_t mode) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    const SIn xs = sin_of(member_opnd(c, m, inIdx));
    double* A1 = reinterpret_cast<double*>(lds + m.scratch);
    double* A2 = A1 + kSlotWords;       // each double array spans two float slots
    double* A3 = A2 + kSlotWords;
    double ic1 = rec_ld_f64(r, rec::S0), ic2 = rec_ld_f64(r, rec::S2);
    for (uint32_t t = 0; t < c.n; ++t) {
        const double a1 = A1[t], a2 = A2[t], a3 = A3[t];
        const double v0 = (double)sget(xs, t);
        const double v3 = v0 - ic2;
        const double v1 = ic1 * a1 + v3 * a2;
        const double v2 = ic2 + ic1 * a2 + v3 * a3;
        ic1 = v1 * 2.0 - ic1;
        ic2 = v2 * 2.0 - ic2;
        if (Direct) lds[m.outLds + t] = (float)(mode == 0 ? v2 : v1);
        else { A1[t] = v1; A2[t] = v2; }
    }
    rec_st_f64(r, rec::S0, ic1); rec_st_f64(r, rec::S2, ic2);
}

// PolyBlepOscillatorNode (Oscillators.h:19-94)
__device__ __forceinline__ float blep(float phase, float inc) {
    if (phase < inc) { const float p = phase / inc; return (2.0f - p) * p - 1.0f; }
    if (phase > (1.0f - inc)) { const float p = (phase - 1.0f) / inc; return (p + 2.0f) * p + 1.0f; }
    return 0.0f;
}

// serial part: only `phase += inc; if (phase >= 1) phase -= 1`. The out slot carries inc[t] in
// (from the pre-pass) and the pre-tick phase[t] out (for the post-pass).
__device__ void ser_blep_phase(const Ctx& c, const Member& m) {
    uint32_t* r = c.recs + m.rec * kRecDwords;
    float phase = u2f(r[rec::S0]);
    for (uint32_t t0 = 0; t0 < c.n; t0 += CH) {
        float inc[CH], y[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) inc[j] = lds[m.outLds + min(t0 + j, c.n - 1)];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            y[j] = phase;
            float np = phase + inc[j];
            if (np >= 1.0f) np -= 1.0f;
            if (t0 + j < c.n) phase = np;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) if (t0 + j < c.n) lds[m.outLds + t0 + j] = y[j];
    }
    r[rec::S0] = f2u(phase);
}

__device__ void ser_blep_acc(const Ctx& c, const Member& m) {   // triangle integrator (:58-59)
    uint32_t* r = c.recs + m.rec * kRecDwords;
    float acc = u2f(r[rec::S1]);
    for (uint32_t t0 = 0; t0 < c.n; t0 += CH) {
        float d[CH], y[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) d[j] = lds[m.outLds + min(t0 + j, c.n - 1)];
#pragma unroll
        for (int j = 0; j < CH; ++j) { const float na = acc + d[j]; if (t0 + j < c.n) acc = na; y[j] = acc; }
#pragma unroll
        for (int j = 0; j < CH; ++j) if (t0 + j < c.n) lds[m.outLds + t0 + j] = y[j];
    }
    r[rec::S1] = f2u(acc);
}

// ---- task dispatch ------------------------------------------------------------------------------
template <typename F>
__device__ __forceinline__ void for_members(const Ctx& c, const Member* members, const Task& t, F&& f) {
    for (uint32_t k = 0; k < t.count; ++k) { const Member m = members[t.first + k]; f(m); }
}

// Stateful task: parallel pre-pass over all members, lane-per-member chain, parallel post-pass.
__device__ void run_stateful(const Ctx& c, const Member* members, const Task& t) {
    const uint32_t n = c.n;
    const uint16_t op = t.opcode;
    // ---- pre-pass (sample-parallel) ----
    if (op == OP_BLEPSAW || op == OP_BLEPSQUARE || op == OP_BLEPTRIANGLE) {
        const float sr = c.g->sampleRateF;
        for_members(c, members, t, [&](const Member& m) {
            if (member_nin(c, m) < 1) return;
            const uint32_t o = member_opnd(c, m, 0);
            for (uint32_t i = c.lane; i < n; i += 64) lds[m.outLds + i] = fetch(c, o, i) / sr;
        });
    } else if (op == OP_SVF) {
        const double sr = c.g->sampleRate;
        for_members(c, members, t, [&](const Member& m) {
            if (member_nin(c, m) < 3) return;
            const uint32_t oF = member_opnd(c, m, 0), oQ = member_opnd(c, m, 1);
            double* A1 = reinterpret_cast<double*>(lds + m.scratch);
            double* A2 = A1 + kSlotWords; double* A3 = A2 + kSlotWords;
            for (uint32_t i = c.lane; i < n; i += 64) {   // updateCoeffs, SVF.h:72-80
                const double fc = (double)fetch(c, oF, i), q = (double)fetch(c, oQ, i);
                const double g = tan(3.14159265359 * clampd(fc, 20.0, sr / 2.0001) / sr);
                const double k = 1.0 / clampd(q, 0.25, 20.0);
                const double a1 = 1.0 / (1.0 + g * (g + k));
                const double a2 = g * a1;
                A1[i] = a1; A2[i] = a2; A3[i] = g * a2;
            }
        });
    } else if (op == OP_SVFSHELF) {
        const double sr = c.g->sampleRate;
        for_members(c, members, t, [&](const Member& m) {
            if (member_nin(c, m) < 4) return;
            const uint32_t mode = c.recs[m.rec * kRecDwords + rec::P0];
            const uint32_t oF = member_opnd(c, m, 0), oQ = member_opnd(c, m, 1), oG = member_opnd(c, m, 2);
            double* A1 = reinterpret_cast<double*>(lds + m.scratch);
            double* A2 = A1 + kSlotWords; double* A3 = A2 + kSlotWords;
            for (uint32_t i = c.lane; i < n; i += 64) {   // updateCoeffs, SVFShelf.h:66-83
                const double fc = (double)fetch(c, oF, i), q = (double)fetch(c, oQ, i), dB = (double)fetch(c, oG, i);
                const double A = pow(10.0, dB / 40.0);
                double g = tan(3.14159265359 * clampd(fc, 20.0, sr / 2.0001) / sr);
                double k = 1.0 / clampd(q, 0.25, 20.0);
                if (mode == 0) g /= A;
                if (mode == 1) g *= A;
                if (mode == 2) k /= A;
                const double a1 = 1.0 / (1.0 + g * (g + k));
                const double a2 = g * a1;
                A1[i] = a1; A2[i] = a2; A3[i] = g * a2;
            }
        });
    } else if (op == OP_MM1P) {
        for_members(c, members, t, [&](const Member& m) {
            if (member_nin(c, m) < 2) return;
            const uint32_t oG = member_opnd(c, m, 0);
            double* G = reinterpret_cast<double*>(lds + m.scratch);
            for (uint32_t i = c.lane; i < n; i += 64) {
                const double g = clampd((double)fetch(c, oG, i), 0.0, 0.9999);
                G[i] = g / (1.0 + g);
            }
        });
    }
    WAVE_SYNC();

    // ---- chain (lane-per-member) ----
    if (c.lane < t.count && n > 0) {
        const Member m = members[t.first + c.lane];
        switch (op) {
            case OP_PHASOR:   ser_phasor<false>(c, m); break;
            case OP_SPHASOR:  ser_phasor<true>(c, m); break;
            case OP_POLE:     ser_pole(c, m); break;
            case OP_ENV:      ser_env(c, m); break;
            case OP_BIQUAD:   ser_biquad(c, m); break;
            case OP_COUNTER:  ser_counter(c, m); break;
            case OP_ACCUM:    ser_accum(c, m); break;
            case OP_LATCH:    ser_latch(c, m); break;
            case OP_MAXHOLD:  ser_maxhold(c, m); break;
            case OP_ONCE:     ser_once(c, m); break;
            case OP_SEQ:      ser_seq(c, m); break;
            case OP_MM1P:
                if (member_nin(c, m) < 2) szero(m, n); else ser_mm1p(c, m);
                break;
            case OP_SVF: {
                if (member_nin(c, m) < 3) { szero(m, n); break; }
                const uint32_t mode = c.recs[m.rec * kRecDwords + rec::P0];
                if (mode <= 1) ser_svf_chain<true>(c, m, 2, mode); else ser_svf_chain<false>(c, m, 2, mode);
                break;
            }
            case OP_SVFSHELF:
                if (member_nin(c, m) < 4) szero(m, n); else ser_svf_chain<false>(c, m, 3, 0);
                break;
            case OP_BLEPSAW: case OP_BLEPSQUARE: case OP_BLEPTRIANGLE:
                if (member_nin(c, m) < 1) szero(m, n); else ser_blep_phase(c, m);
                break;
            default: break;
        }
    }
    WAVE_SYNC();

    // ---- post-pass (sample-parallel) ----
    if (op == OP_BLEPSAW || op == OP_BLEPSQUARE || op == OP_BLEPTRIANGLE) {
        const float sr = c.g->sampleRateF;
        for_members(c, members, t, [&](const Member& m) {
            if (member_nin(c, m) < 1) return;
            const uint32_t o = member_opnd(c, m, 0);
            for (uint32_t i = c.lane; i < n; i += 64) {
                const float inc = fetch(c, o, i) / sr;
                const float phase = lds[m.outLds + i];
                float y;
                if (op == OP_BLEPSAW) {
                    y = 2.0f * phase - 1.0f - blep(phase, inc);
                } else {
                    const float naive = phase < 0.5f ? 1.0f : -1.0f;
                    const float halfPhase = fmodf(phase + 0.5f, 1.0f);
                    const float square = naive + blep(phase, inc) - blep(halfPhase, inc);
                    y = (op == OP_BLEPSQUARE) ? square : (4.0f * inc * square);
                }
                lds[m.outLds + i] = y;
            }
        });
        if (op == OP_BLEPTRIANGLE) {
            WAVE_SYNC();
            if (c.lane < t.count && n > 0) {
                const Member m = members[t.first + c.lane];
                if (member_nin(c, m) >= 1) ser_blep_acc(c, m);
            }
        }
    } else if (op == OP_SVF) {
        for_members(c, members, t, [&](const Member& m) {
            if (member_nin(c, m) < 3) return;
            const uint32_t mode = c.recs[m.rec * kRecDwords + rec::P0];
            if (mode <= 1) return;
            const uint32_t oQ = member_opnd(c, m, 1), oX = member_opnd(c, m, 2);
            const double* V1 = reinterpret_cast<const double*>(lds + m.scratch);
            const double* V2 = V1 + kSlotWords;
            for (uint32_t i = c.lane; i < n; i += 64) {   // SVF.h:57-69
                const double k = 1.0 / clampd((double)fetch(c, oQ, i), 0.25, 20.0);
                const double v0 = (double)fetch(c, oX, i), v1 = V1[i], v2 = V2[i];
                float y;
                if (mode == 2)      y = (float)(v0 - k * v1 - v2);
                else if (mode == 3) y = (float)(v0 - k * v1);
                else                y = (float)(v0 - 2.0 * k * v1);
                lds[m.outLds + i] = y;
            }
        });
    } else if (op == OP_SVFSHELF) {
        for_members(c, members, t, [&](const Member& m) {
            if (member_nin(c, m) < 4) return;
            const uint32_t mode = c.recs[m.rec * kRecDwords + rec::P0];
            const uint32_t oQ = member_opnd(c, m, 1), oG = member_opnd(c, m, 2), oX = member_opnd(c, m, 3);
            const double* V1 = reinterpret_cast<const double*>(lds + m.scratch);
            const double* V2 = V1 + kSlotWords;
            for (uint32_t i = c.lane; i < n; i += 64) {   // SVFShelf.h:54-63
                const double A = pow(10.0, (double)fetch(c, oG, i) / 40.0);
                double k = 1.0 / clampd((double)fetch(c, oQ, i), 0.25, 20.0);
                if (mode == 2) k /= A;
                const double v0 = (double)fetch(c, oX, i), v1 = V1[i], v2 = V2[i];
                float y;
                if (mode == 2)      y = (float)(v0 + k * (A * A - 1.0) * v1);
                else if (mode == 0) y = (float)(v0 + k * (A - 1.0) * v1 + (A * A - 1.0) * v2);
                else                y = (float)(A * A * v0 + k * (1.0 - A) * A * v1 + (1.0 - A * A) * v2);
                lds[m.outLds + i] = y;
            }
        });
    }
    WAVE_SYNC();
    // stateful outputs are produced in LDS; the planner schedules an OP_COPY export if another
    // island consumes them, except for direct HBM outputs requested here
    for_members(c, members, t, [&](const Member& m) {
        if (m.outHbm == kNone) return;
        for (uint32_t i = c.lane; i < n; i += 64) c.hbm[(size_t)m.outHbm * c.stride + i] = lds[m.outLds + i];
    });
}

__device__ void run_task(const Ctx& c, const Member* members, const Task& t) {
    const uint32_t s0 = min((uint32_t)t.s0, c.n), s1 = min((uint32_t)t.s1, c.n);
#define PAR(OPC, FN) case OPC: for_members(c, members, t, [&](const Member& m) { FN(c, m, s0, s1); }); break;
#define UN(OPC)  case OPC: for_members(c, members, t, [&](const Member& m) { run_unary<OPC>(c, m, s0, s1); }); break;
#define BI(OPC)  case OPC: for_members(c, members, t, [&](const Member& m) { run_binary<OPC>(c, m, s0, s1); }); break;
#define RE(OPC)  case OPC: for_members(c, members, t, [&](const Member& m) { run_reduce<OPC>(c, m, s0, s1); }); break;
    switch (t.opcode) {
        UN(OP_SIN) UN(OP_COS) UN(OP_TAN) UN(OP_TANH) UN(OP_ASINH) UN(OP_LN) UN(OP_LOG) UN(OP_LOG2)
        UN(OP_CEIL) UN(OP_FLOOR) UN(OP_ROUND) UN(OP_SQRT) UN(OP_EXP) UN(OP_ABS)
        BI(OP_LE) BI(OP_LEQ) BI(OP_GE) BI(OP_GEQ) BI(OP_POW) BI(OP_EQ) BI(OP_AND) BI(OP_OR)
        RE(OP_ADD) RE(OP_SUB) RE(OP_MUL) RE(OP_DIV) RE(OP_MOD) RE(OP_MIN) RE(OP_MAX)
        PAR(OP_IN, run_in) PAR(OP_COPY, run_copy) PAR(OP_ROOT, run_root) PAR(OP_PREWARP, run_prewarp)
        PAR(OP_TIME, run_time) PAR(OP_METRO, run_metro) PAR(OP_RAND, run_rand) PAR(OP_Z, run_z)
        PAR(OP_TAPIN, run_tapin) PAR(OP_TAPOUT, run_tapout) PAR(OP_SDELAY, run_sdelay) PAR(OP_DELAY, run_delay)
        case OP_CONST: case OP_SR:   // materialised only when a consumer needs a real buffer
            for_members(c, members, t, [&](const Member& m) {
                const float v = u2f(c.recs[m.rec * kRecDwords + rec::P0]);
                for (uint32_t i = s0 + c.lane; i < s1; i += 64) put(c, m, i, v);
            });
            break;
        default: run_stateful(c, members, t); break;
    }
#undef PAR
#undef UN
#undef BI
#undef RE
}

// RootNode::stillRunning (Core.h:28-31) and the channel test of RootRenderSequence::process
// (GraphRenderSequence.h:214-219)
__device__ __forceinline__ bool root_running(const uint32_t* recs, uint32_t rootRec, uint32_t numOut) {
    const uint32_t* r = recs + rootRec * kRecDwords;
    const float tg = u2f(r[rec::ROOT_TARGET]), g = u2f(r[rec::ROOT_GAIN]);
    const bool on = tg > 0.5f;
    const bool settled = fabsf(tg - g) <= 1e-6f;
    const int ch = (int)r[rec::ROOT_CHANNEL];
    return (on || !settled) && ch >= 0 && (uint32_t)ch < numOut;
}

} // namespace

// ---- kernels ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads)
void elemhip_island_kernel(PlanView pv, uint32_t* recs, float* hbm, const Globals* g, const uint32_t* lcg,
                           uint32_t levelBegin) {
    const Island isl = pv.islands[pv.levelIslands[levelBegin + blockIdx.x]];
    if (!root_running(recs, isl.rootRec, g->numOut)) return;

    Ctx c;
    c.recs = recs; c.hbm = hbm; c.g = g; c.operands = pv.operands; c.lcg = lcg;
    c.n = g->numSamples; c.stride = g->blockStride; c.numIn = g->numIn;
    c.lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;

    // LDS word 0 (and 1) read as 0.0f; broadcast cells take their node's current value
    if (threadIdx.x < 2) lds[threadIdx.x] = 0.0f;
    for (uint32_t k = isl.constBegin + threadIdx.x; k < isl.constEnd; k += kThreads) {
        const ConstCell cc = pv.constCells[k];
        lds[cc.ldsWord] = u2f(recs[cc.rec * kRecDwords + rec::P0]);
    }
    __syncthreads();

    uint32_t stage = 0;
    for (uint32_t ti = isl.taskBegin; ti < isl.taskEnd; ++ti) {
        const Task t = pv.tasks[ti];
        while (stage < t.stage) { __syncthreads(); ++stage; }
        if (t.wave == wave) run_task(c, pv.members, t);
    }
}

// Epilogue: one workgroup. (1) zero + sum running roots into the output bus in render-sequence
// order (GraphRenderSequence.h:286-295, 227-231); (2) promote tap buffers of active roots
// (:306-308, Feedback.h:90-109); (3) advance root fades (GainFade.h:70-71); (4) advance the block.
__global__ __launch_bounds__(1024)
void elemhip_epilogue_kernel(PlanView pv, uint32_t* recs, const float* hbm, Globals* g, float* outRing) {
    const uint32_t n = g->numSamples, numOut = g->numOut, stride = g->blockStride;
    float* out = outRing + (size_t)g->blockSlot * numOut * stride;
    for (uint32_t idx = threadIdx.x; idx < numOut * n; idx += blockDim.x) {
        const uint32_t ch = idx / n, i = idx - ch * n;
        float acc = 0.0f;
        for (uint32_t r = 0; r < pv.numRoots; ++r) {
            const RootEntry re = pv.roots[r];
            if (!root_running(recs, re.rec, numOut)) continue;
            if (recs[re.rec * kRecDwords + rec::ROOT_CHANNEL] != ch) continue;
            acc += hbm[(size_t)re.hbm * stride + i];
        }
        out[(size_t)ch * stride + i] = acc;
    }
    for (uint32_t k = 0; k < pv.numTaps; ++k) {
        const TapEntry te = pv.taps[k];
        const uint32_t* rr = recs + te.rootRec * kRecDwords;
        // only sequences that ran this block hold fresh tap data; promotion needs root.active()
        if (!(u2f(rr[rec::ROOT_TARGET]) > 0.5f)) continue;
        const uint32_t* tr = recs + te.rec * kRecDwords;
        float* shared = rec_ptr(tr, rec::TAP_SHARED);
        const float* priv = rec_ptr(tr, rec::TAP_PRIVATE);
        if (!shared || !priv) continue;
        __syncthreads();   // earlier promotions into the same name complete first (last writer wins)
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) shared[i] = priv[i];
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < pv.numRoots; r += blockDim.x) {
        uint32_t* rr = recs + pv.roots[r].rec * kRecDwords;
        if (!root_running(recs, pv.roots[r].rec, numOut)) continue;
        const float gcur = u2f(rr[rec::ROOT_GAIN]), tg = u2f(rr[rec::ROOT_TARGET]), step = u2f(rr[rec::ROOT_STEP]);
        if (gcur != tg && (rr[rec::ROOT_HASIN] || g->numIn > 0) /* fade.process ran (Core.h:74-77) */)
            rr[rec::ROOT_GAIN] = f2u(clampf(gcur + step * (float)(int)n, 0.0f, 1.0f));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        g->sampleTime += (int64_t)n;
        g->blockSlot = (g->blockSlot + 1) % g->ringSlots;
    }
}

// Host -> device parameter patches, applied at a block boundary (the reference's per-node
// atomics / SPSC queues drained at the top of process(), e.g. Delays.h:92-95).
__global__ void elemhip_patch_kernel(const Patch* patches, uint32_t count, uint32_t* recs, uint32_t* globals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const Patch p = patches[i];
    if (p.kind == 0) recs[p.index] = p.value;
    else if (p.kind == 1) { if (u2f(recs[p.index]) == 0.0f) recs[p.index] = p.value; }   // OnceNode arm, Core.h:359-362
    else globals[p.index] = p.value;
}

// ---- host launchers ------------------------------------------------------------------------------------
namespace elemhip {

hipError_t configure_kernels(uint32_t maxLdsBytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(elemhip_island_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxLdsBytes);
}

void launch_level(hipStream_t s, const PlanView& pv, uint32_t* recs, float* hbm, const Globals* g, const uint32_t* lcg,
                  uint32_t levelBegin, uint32_t numIslands, uint32_t ldsBytes) {
    hipLaunchKernelGGL(elemhip_island_kernel, dim3(numIslands), dim3(kThreads), ldsBytes, s, pv, recs, hbm, g, lcg, levelBegin);
}

void launch_epilogue(hipStream_t s, const PlanView& pv, uint32_t* recs, const float* hbm, Globals* g, float* outRing) {
    hipLaunchKernelGGL(elemhip_epilogue_kernel, dim3(1), dim3(1024), 0, s, pv, recs, hbm, g, outRing);
}

void launch_patches(hipStream_t s, const Patch* patches, uint32_t count, uint32_t* recs, uint32_t* globals) {
    if (!count) return;
    hipLaunchKernelGGL(elemhip_patch_kernel, dim3((count + 255) / 256), dim3(256), 0, s, patches, count, recs, globals);
}

} // namespace elemhip
