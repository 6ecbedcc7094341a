// kernels.hip — the block-render path as hand-written HIP for gfx950 (CDNA4).
//
// One launch renders one *island level*: every workgroup (4 wavefronts) interprets the program
// of one island with that island's block buffers AND its program resident in LDS.  Stateless node
// loops (runtime/elem/builtins/Math.h etc.) run sample-parallel, 64 lanes x (samples/64);
// stateful recurrences (phasor, polyBLEP phase, one-pole, biquad, SVF, ...) run one node per lane
// with only the loop-carried state update on the serial chain, everything else hoisted into
// sample-parallel pre/post passes, and the chain's operands streamed from LDS with 16-byte reads
// one chunk ahead of the dependent arithmetic.  An epilogue workgroup then sums root buffers into
// the output bus, advances root fades and promotes feedback taps (GraphRenderSequence.h:268-309).
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

// Global-memory pointers carry an explicit address space. With plain (generic) pointers the
// optimiser merges `cond ? hbm[i] : lds[j]` into one FLAT load of a selected generic pointer,
// and a flat access to LDS costs ~700 cycles instead of ~100.
typedef float __attribute__((address_space(1)))*          gfp;
typedef const float __attribute__((address_space(1)))*    gcfp;
typedef uint32_t __attribute__((address_space(1)))*       gup;
typedef const uint32_t __attribute__((address_space(1)))* gcup;

struct Ctx {
    gup             recs;
    gfp             hbm;
    const Globals*  g;
    gcup            lcg;      // [2*(kMaxBlock+1)] jump-ahead table for `rand`
    uint32_t        members;  // LDS word offsets of the staged program tables
    uint32_t        operands;
    uint32_t        n;        // frames this block
    uint32_t        stride;   // floats per arena buffer
    uint32_t        numIn;    // host input channels
    uint32_t        lane;
    float           srF;
    double          sr;
    int64_t         sampleTime;   // of the block being rendered (a multi-block launch advances it per block)
};

#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))

__device__ __forceinline__ float    u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f)    { return __float_as_uint(f); }
__device__ __forceinline__ uint32_t ldsu(uint32_t w) { return __float_as_uint(lds[w]); }

__device__ __forceinline__ double rec_ld_f64(gcup r, uint32_t d) {
    return __hiloint2double((int)r[d + 1], (int)r[d]);
}
__device__ __forceinline__ void rec_st_f64(gup r, uint32_t d, double v) {
    r[d] = (uint32_t)__double2loint(v); r[d + 1] = (uint32_t)__double2hiint(v);
}
__device__ __forceinline__ gfp rec_ptr(gcup r, uint32_t d) {
    uint64_t p = (uint64_t)r[d] | ((uint64_t)r[d + 1] << 32);
    return (gfp)reinterpret_cast<float*>(p);
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) {   // std::clamp
    return (v < lo) ? lo : ((hi < v) ? hi : v);
}
__device__ __forceinline__ double clampd(double v, double lo, double hi) {
    return (v < lo) ? lo : ((hi < v) ? hi : v);
}

// ---- program access (staged in LDS) ------------------------------------------------------------
typedef float    v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef double   v2d __attribute__((ext_vector_type(2)));

// 16-byte LDS accesses: the compiler only emits ds_read_b128 / ds_write_b128 when told the
// address is aligned (every slot, scratch array and program table is laid out on 16 bytes)
__device__ __forceinline__ v4u lds4u(uint32_t w) { return *reinterpret_cast<const v4u*>(__builtin_assume_aligned(&lds[w], 16)); }
__device__ __forceinline__ v4f ld4(uint32_t w) { return *reinterpret_cast<const v4f*>(__builtin_assume_aligned(&lds[w], 16)); }
__device__ __forceinline__ void st4(uint32_t w, v4f v) { *reinterpret_cast<v4f*>(__builtin_assume_aligned(&lds[w], 16)) = v; }
__device__ __forceinline__ v2d ld2d(const double* p) { return *reinterpret_cast<const v2d*>(__builtin_assume_aligned(p, 16)); }

__device__ __forceinline__ Member member_uniform(const Ctx& c, uint32_t k) {   // same k on every lane
    const uint32_t w = c.members + k * 8u;
    const v4u a = lds4u(w);
    Member m;
    m.rec = UNI(a.x); m.opnd = UNI(a.y); m.nin = UNI(a.z); m.outLds = UNI(a.w);
    m.outHbm = UNI(ldsu(w + 4)); m.scratch = UNI(ldsu(w + 5));
    // pad0_/pad1_ carry the first two operand codes on the device
    m.pad0_ = UNI(ldsu(c.operands + m.opnd)); m.pad1_ = UNI(ldsu(c.operands + m.opnd + 1));
    return m;
}
__device__ __forceinline__ Member member_lane(const Ctx& c, uint32_t k) {      // per-lane k
    const uint32_t w = c.members + k * 8u;
    const v4u a = lds4u(w);
    Member m;
    m.rec = a.x; m.opnd = a.y; m.nin = a.z; m.outLds = a.w;
    m.outHbm = ldsu(w + 4); m.scratch = ldsu(w + 5);
    return m;
}
__device__ __forceinline__ uint32_t member_nin(const Ctx& c, const Member& m) {
    return m.nin == kNone ? c.numIn : m.nin;
}
__device__ __forceinline__ uint32_t opnd_uniform(const Ctx& c, const Member& m, uint32_t k) {
    if (k == 0) return m.pad0_;      // decoded with the task header / member
    if (k == 1) return m.pad1_;
    return UNI(ldsu(c.operands + m.opnd + k));
}
__device__ __forceinline__ uint32_t opnd_lane(const Ctx& c, const Member& m, uint32_t k) {
    return ldsu(c.operands + m.opnd + k);
}

// Sample-parallel operand: LDS buffer (step 1) / broadcast cell (step 0) share one addressing
// form, HBM buffers are the other; both are wave-uniform so the choice is a scalar branch made
// once per operand, outside the sample loop.
struct PIn { uint32_t base, step; gcfp g; };
__device__ __forceinline__ PIn pin_of(const Ctx& c, uint32_t o) {
    // arithmetic selects only: an if/else ladder here compiled to ~30 scalar instructions of control flow
    const uint32_t kind = o >> 30, v = o & kOpValMask;
    PIn p;
    p.step = (kind == 0u) ? 1u : 0u;                     // LDS buffer
    p.base = (kind <= 1u) ? v : 0u;                      // buffer / broadcast cell; otherwise LDS word 0 (= 0.0f)
    const uint64_t ga = (uint64_t)(uintptr_t)(float*)c.hbm + (uint64_t)v * c.stride * 4u;
    p.g = (gcfp)(float*)(uintptr_t)((kind == 2u) ? ga : 0ull);
    return p;
}
__device__ __forceinline__ float pget(const PIn& p, uint32_t i) { return p.g ? p.g[i] : lds[p.base + i * p.step]; }
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

// Canonical frame mapping of sample-parallel tasks: a task covers 64*V frames, V in {1,2,4,8}
// (the planner only emits such ranges), and lane l owns the V CONSECUTIVE frames i .. i+V-1 with
// i = s0 + l*V: one vector LDS/global access per operand, no per-frame control flow. Every
// sample-parallel op uses this mapping, which is what lets dependent ops of one stage run back
// to back on a wave without a barrier (each lane only re-reads what it wrote itself).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const v2f __attribute__((address_space(1)))* gcv2;
typedef const v4f __attribute__((address_space(1)))* gcv4;
typedef v2f __attribute__((address_space(1)))* gv2;
typedef v4f __attribute__((address_space(1)))* gv4;

template <int V>
__device__ __forceinline__ void vload(const PIn& p, uint32_t i, float (&x)[V]) {
    if (p.g) {
        if constexpr (V == 1) x[0] = p.g[i];
        else if constexpr (V == 2) { const v2f a = *(gcv2)(p.g + i); x[0] = a.x; x[1] = a.y; }
        else {
#pragma unroll
            for (int q = 0; q < V; q += 4) { const v4f a = *(gcv4)(p.g + i + q); x[q] = a.x; x[q + 1] = a.y; x[q + 2] = a.z; x[q + 3] = a.w; }
        }
    } else if (p.step) {
        const uint32_t w = p.base + i;
        if constexpr (V == 1) x[0] = lds[w];
        else if constexpr (V == 2) { const v2f a = *reinterpret_cast<const v2f*>(__builtin_assume_aligned(&lds[w], 8)); x[0] = a.x; x[1] = a.y; }
        else {
#pragma unroll
            for (int q = 0; q < V; q += 4) { const v4f a = ld4(w + q); x[q] = a.x; x[q + 1] = a.y; x[q + 2] = a.z; x[q + 3] = a.w; }
        }
    } else {
        const float v = lds[p.base];
#pragma unroll
        for (int q = 0; q < V; ++q) x[q] = v;
    }
}

// nlim = first frame that must NOT be written (block shorter than the task range => per-frame tail)
template <int V>
__device__ __forceinline__ void vstore(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim, const float (&y)[V]) {
    if (i + V <= nlim) {
        if (m.outLds != kNone) {
            const uint32_t w = m.outLds + i;
            if constexpr (V == 1) lds[w] = y[0];
            else if constexpr (V == 2) { v2f a; a.x = y[0]; a.y = y[1]; *reinterpret_cast<v2f*>(__builtin_assume_aligned(&lds[w], 8)) = a; }
            else {
#pragma unroll
                for (int q = 0; q < V; q += 4) { v4f a; a.x = y[q]; a.y = y[q + 1]; a.z = y[q + 2]; a.w = y[q + 3]; st4(w + q, a); }
            }
        }
        if (m.outHbm != kNone) {
            gfp g = c.hbm + (size_t)m.outHbm * c.stride + i;
            if constexpr (V == 1) g[0] = y[0];
            else if constexpr (V == 2) { v2f a; a.x = y[0]; a.y = y[1]; *(gv2)g = a; }
            else {
#pragma unroll
                for (int q = 0; q < V; q += 4) { v4f a; a.x = y[q]; a.y = y[q + 1]; a.z = y[q + 2]; a.w = y[q + 3]; *(gv4)(g + q) = a; }
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < V; ++q) if (i + q < nlim) put(c, m, i + q, y[q]);
    }
}

template <int V>
__device__ __forceinline__ void vzero(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    float z[V];
#pragma unroll
    for (int q = 0; q < V; ++q) z[q] = 0.0f;
    vstore<V>(c, m, i, nlim, z);
}

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

template <uint16_t OPC, int V>
__device__ __forceinline__ void run_unary(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    if (member_nin(c, m) < 1) return vzero<V>(c, m, i, nlim);
    float x[V];
    vload<V>(pin_of(c, opnd_uniform(c, m, 0)), i, x);
#pragma unroll
    for (int q = 0; q < V; ++q) x[q] = unary_eval(OPC, x[q]);
    vstore<V>(c, m, i, nlim, x);
}

template <uint16_t OPC, int V>
__device__ __forceinline__ void run_binary(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    if (member_nin(c, m) < 2) return vzero<V>(c, m, i, nlim);
    float x[V], y[V];
    vload<V>(pin_of(c, opnd_uniform(c, m, 0)), i, x);
    vload<V>(pin_of(c, opnd_uniform(c, m, 1)), i, y);
#pragma unroll
    for (int q = 0; q < V; ++q) x[q] = binary_eval(OPC, x[q], y[q]);
    vstore<V>(c, m, i, nlim, x);
}

// BinaryReducingNode (Math.h:59-89): strict left fold over the children, any fan-in. Operand
// codes are fetched 64 at a time (lane b reads code k+b with one LDS access, v_readlane
// broadcasts them) and the loads of a batch of children are all in flight before the ordered fold.
template <uint16_t OPC, int V>
__device__ __forceinline__ void run_reduce(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    uint32_t nin = member_nin(c, m);
    if (nin < 1) return vzero<V>(c, m, i, nlim);
    if (m.nin == kNone && nin > kMaxHostIn) nin = kMaxHostIn;
    float acc[V];
    vload<V>(pin_of(c, opnd_uniform(c, m, 0)), i, acc);
    if (nin == 2) {
        float y[V];
        vload<V>(pin_of(c, opnd_uniform(c, m, 1)), i, y);
#pragma unroll
        for (int q = 0; q < V; ++q) acc[q] = reduce_eval(OPC, acc[q], y[q]);
        return vstore<V>(c, m, i, nlim, acc);
    }
    constexpr int KB = (V <= 2) ? 16 : (V == 4 ? 8 : 4);   // children in flight (KB * V registers)
    uint32_t k = 1;
    // all HBM children (a mixer): one buffer descriptor over the arena, child offset in an SGPR —
    // 4 instructions per child (v_readlane, s_and, s_mul, buffer_load) instead of a generic decode
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((float*)c.hbm, 0, 0x7FFFFFFF, 0x00020000);
    const uint32_t strideBytes = c.stride * 4u;
    while (k < nin) {
        const uint32_t left = min(nin - k, 64u);
        const uint32_t mine = (c.lane < left) ? ldsu(c.operands + m.opnd + k + c.lane) : (uint32_t)kOpZero;
        const bool allHbm = (V <= 2) && __all(c.lane >= left || ((mine >> 30) == 2u && (mine & kOpValMask) < (0x7FFFFFFFu / strideBytes)));
        if (allHbm) {
            constexpr int KH = (V == 1) ? 64 : 32;   // one memory round trip per 64 children
            for (uint32_t b0 = 0; b0 < left; b0 += KH) {
                float v[KH][V];
#pragma unroll
                for (int b = 0; b < KH; ++b) {
                    if (b0 + b < left) {
                        const uint32_t code = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)(b0 + b));
                        const uint32_t soff = (code & kOpValMask) * strideBytes;
                        if constexpr (V == 1) v[b][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, i * 4u, soff, 0));
                        else { const v2f t2 = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, i * 4u, soff, 0)); v[b][0] = t2.x; v[b][V - 1] = t2.y; }
                    }
                }
#pragma unroll
                for (int b = 0; b < KH; ++b)
                    if (b0 + b < left) {
#pragma unroll
                        for (int q = 0; q < V; ++q) acc[q] = reduce_eval(OPC, acc[q], v[b][q]);
                    }
            }
        } else {
            for (uint32_t b0 = 0; b0 < left; b0 += KB) {
                float v[KB][V];
#pragma unroll
                for (int b = 0; b < KB; ++b)
                    if (b0 + b < left) vload<V>(pin_of(c, (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)(b0 + b))), i, v[b]);
#pragma unroll
                for (int b = 0; b < KB; ++b)
                    if (b0 + b < left) {
#pragma unroll
                        for (int q = 0; q < V; ++q) acc[q] = reduce_eval(OPC, acc[q], v[b][q]);
                    }
            }
        }
        k += left;
    }
    vstore<V>(c, m, i, nlim, acc);
}

// Per-frame form for the less common sample-parallel nodes: f(frame) -> value, same mapping.
template <int V, typename F>
__device__ __forceinline__ void vmap(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim, F&& f) {
    float y[V];
#pragma unroll
    for (int q = 0; q < V; ++q) y[q] = f(i + q);
    vstore<V>(c, m, i, nlim, y);
}

// IdentityNode `in` (Math.h:92-126): out = inputData[channel]; leaf => host input channel.
template <int V>
__device__ __forceinline__ void run_in(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    const uint32_t ch = UNI(c.recs[m.rec * kRecDwords + rec::P0]);
    const uint32_t nin = member_nin(c, m);
    const bool neg = (int32_t)ch < 0;   // static_cast<size_t>(negative int) is huge: zero-fill
    if (neg || ch >= nin || (m.nin == kNone && ch >= kMaxHostIn)) return vzero<V>(c, m, i, nlim);
    float x[V];
    vload<V>(pin_of(c, opnd_uniform(c, m, ch)), i, x);
    vstore<V>(c, m, i, nlim, x);
}

template <int V>
__device__ __forceinline__ void run_copy(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    float x[V];
    vload<V>(pin_of(c, opnd_uniform(c, m, 0)), i, x);
    vstore<V>(c, m, i, nlim, x);
}

// RootNode (Core.h:66-78) + GainFade::process (helpers/GainFade.h:56-72). The gain itself is
// advanced once per block by the epilogue, after every island has read it.
template <int V>
__device__ __forceinline__ void run_root(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    if (member_nin(c, m) < 1) return vzero<V>(c, m, i, nlim);
    gcup r = c.recs + m.rec * kRecDwords;
    const float g = u2f(UNI(r[rec::ROOT_GAIN])), tg = u2f(UNI(r[rec::ROOT_TARGET])), step = u2f(UNI(r[rec::ROOT_STEP]));
    float x[V];
    vload<V>(pin_of(c, opnd_uniform(c, m, 0)), i, x);
    if (g == tg) {
#pragma unroll
        for (int q = 0; q < V; ++q) x[q] = x[q] * tg;
    } else {
#pragma unroll
        for (int q = 0; q < V; ++q) x[q] = x[q] * clampf(g + step * (float)(int)(i + q), 0.0f, 1.0f);
    }
    vstore<V>(c, m, i, nlim, x);
}

// CutoffPrewarpNode (filters/MultiMode1p.h:9-36): double internals.
template <int V>
__device__ __forceinline__ void run_prewarp(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    if (member_nin(c, m) < 1) return vzero<V>(c, m, i, nlim);
    const double T = 1.0 / c.sr;
    float x[V];
    vload<V>(pin_of(c, opnd_uniform(c, m, 0)), i, x);
#pragma unroll
    for (int q = 0; q < V; ++q) {
        const double twoPi = 2.0 * 3.141592653589793238;
        const double wd = twoPi * (double)x[q];
        x[q] = (float)tan(wd * T / 2.0);
    }
    vstore<V>(c, m, i, nlim, x);
}

// SampleTimeNode (wasm/SampleTime.h:11-24), MetronomeNode (wasm/Metro.h:40-55)
template <int V>
__device__ __forceinline__ void run_time(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    const int64_t st = c.sampleTime;
    vmap<V>(c, m, i, nlim, [&](uint32_t t) { return (float)(double)((uint64_t)st + (uint64_t)t); });
}
template <int V>
__device__ __forceinline__ void run_metro(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    gcup r = c.recs + m.rec * kRecDwords;
    const int64_t is64 = (int64_t)((uint64_t)r[rec::P0] | ((uint64_t)r[rec::P1] << 32));
    const double is = (double)is64;
    const int64_t st = c.sampleTime;
    vmap<V>(c, m, i, nlim, [&](uint32_t t) {
        const double tt = (double)((uint64_t)st + (uint64_t)t) / is;
        return ((tt - floor(tt)) < 0.5) ? 1.0f : 0.0f;
    });
}

// TapInNode / TapOutNode (Feedback.h:40-53, 111-126); buffers are always float.
template <int V>
__device__ __forceinline__ void run_tapin(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    gcup r = c.recs + m.rec * kRecDwords;
    gcfp shared = rec_ptr(r, rec::TAP_SHARED);
    if (!shared) return vzero<V>(c, m, i, nlim);
    vmap<V>(c, m, i, nlim, [&](uint32_t t) { return shared[t]; });
}
template <int V>
__device__ __forceinline__ void run_tapout(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    gcup r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 1) return vzero<V>(c, m, i, nlim);
    gfp priv = rec_ptr(r, rec::TAP_PRIVATE);
    float x[V];
    vload<V>(pin_of(c, opnd_uniform(c, m, 0)), i, x);
#pragma unroll
    for (int q = 0; q < V; ++q) if (i + q < nlim) priv[i + q] = x[q];
    vstore<V>(c, m, i, nlim, x);
}

template <int V>
__device__ __forceinline__ void run_fill(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    const float v = u2f(UNI(c.recs[m.rec * kRecDwords + rec::P0]));
    vmap<V>(c, m, i, nlim, [&](uint32_t) { return v; });
}

// UniformRandomNoiseNode (Noise.h:9-43): the LCG is affine mod 2^32, so sample i is an exact
// jump-ahead  s_{i+1} = A[i+1]*s_0 + C[i+1]  from a precomputed table — bit-identical, no chain.
__device__ __forceinline__ void run_rand(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    gup r = c.recs + m.rec * kRecDwords;
    const uint32_t seed = r[rec::S0];
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) {
        const uint32_t s = c.lcg[2 * (i + 1)] * seed + c.lcg[2 * (i + 1) + 1];
        put(c, m, i, (float)(int)((s >> 16) & 0x7FFF) / 32767.0f);
    }
    WAVE_SYNC();
    if (c.lane == 0) r[rec::S0] = c.lcg[2 * c.n] * seed + c.lcg[2 * c.n + 1];
}

// SingleSampleDelayNode (Delays.h:15-39): out[i] = (i ? in[i-1] : z); z = in[n-1]
__device__ __forceinline__ void run_z(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    if (member_nin(c, m) < 1) return zero_fill(c, m, s0, s1);
    gup r = c.recs + m.rec * kRecDwords;
    const uint32_t o = opnd_uniform(c, m, 0);
    const float z = u2f(r[rec::S0]);
    const float last = (c.n > 0) ? fetch(c, o, c.n - 1) : z;
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) put(c, m, i, i ? fetch(c, o, i - 1) : z);
    WAVE_SYNC();
    if (c.lane == 0) r[rec::S0] = f2u(last);
}

// SampleDelayNode (Delays.h:177-272). The reference writes the block into the ring and then
// reads ring[(size + w0 - len + i) & mask]; for i >= len that is this block's in[i-len], for
// i < len it is older ring data the block's own writes cannot touch (size >= len + blockSize).
__device__ __forceinline__ void run_sdelay(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    gup r = c.recs + m.rec * kRecDwords;
    gfp ring = rec_ptr(r, rec::RING_PTR);
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
    const uint32_t o = opnd_uniform(c, m, 0);
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

// VariableDelayNode (Delays.h:51-169). When the smallest read offset in the block exceeds the
// block length no read can observe a write of this block, so reads then writes are
// sample-parallel; otherwise lane 0 walks the block serially.
__device__ __forceinline__ void run_delay(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    gup r = c.recs + m.rec * kRecDwords;
    gfp ring = rec_ptr(r, rec::RING_PTR);
    const int size = (int)r[rec::RING_SIZE];
    int w0 = (int)r[rec::RING_WRITE];
    if (r[rec::RING_RESET]) w0 = 0;
    const uint32_t nin = member_nin(c, m);
    if (nin < 3) {
        WAVE_SYNC();
        if (c.lane == 0) { r[rec::RING_RESET] = 0; r[rec::RING_WRITE] = (uint32_t)w0; }
        return zero_fill(c, m, s0, s1);
    }
    const uint32_t oLen = opnd_uniform(c, m, 0), oFb = opnd_uniform(c, m, 1), oX = opnd_uniform(c, m, 2);
    if (size == 0 || ring == nullptr) {   // Delays.h:106-107 copies input 0
        for (uint32_t i = s0 + c.lane; i < s1; i += 64) put(c, m, i, fetch(c, oLen, i));
        WAVE_SYNC();
        if (c.lane == 0) { r[rec::RING_RESET] = 0; r[rec::RING_WRITE] = (uint32_t)w0; }
        return;
    }
    const float fsize = (float)size;
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
            lds[m.scratch + i] = in;   // every read of the block precedes every write
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

// SampleSeqNode<F,false> (builtins/SampleSeq.h:169-404): k-rate control from in[0][0], two
// cross-fading BufferReader<float>s (detail::GainFade step 0.02/sample) added into a zeroed output.
// Every lane replays the (tiny, uniform) control logic; the fade ramps are produced serially only
// while a reader is still moving (<= 50 frames), the sample reads and the mix are frame-parallel.
__device__ __forceinline__ void run_sampleseq(const Ctx& c, const Member& m, uint32_t s0, uint32_t s1) {
    gup r = c.recs + m.rec * kRecDwords;
    const uint32_t n = c.n;
    const double dur = rec_ld_f64(r, rec::SSQ_DUR);
    double rtDur = rec_ld_f64(r, rec::SSQ_RTDUR);
    float gain[2], target[2], step[2]; uint32_t pos[2], bsz[2]; double start[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        gcup rr = r + rec::SSQ_READER0 + q * rec::SSQ_READER_DWORDS;
        gain[q] = u2f(rr[0]); target[q] = u2f(rr[1]); step[q] = u2f(rr[2]); pos[q] = rr[3];
        start[q] = rec_ld_f64(rr, 4); bsz[q] = rr[6];
    }
    int prev = (int)r[rec::SSQ_PREV], next = (int)r[rec::SSQ_NEXT];
    uint32_t active = r[rec::SSQ_ACTIVE], flags = r[rec::SSQ_FLAGS];
    auto resetReaders = [&]() {   // BufferReader::reset (:149-155)
        gain[0] = gain[1] = 0.0f; target[0] = target[1] = 0.0f; start[0] = start[1] = 0.0;
    };
    if (dur != rtDur) { resetReaders(); rtDur = dur; }                                       // :289-296
    const bool bufPending = r[rec::SSQ_BUFPENDING] != 0, seqPending = r[rec::SSQ_SEQPENDING] != 0;
    if (bufPending) { resetReaders(); flags |= 2u; }                                         // :298-304
    if (seqPending) { prev = -1; next = -1; flags |= 1u; }                                   // :306-316
    gcfp buf = rec_ptr(r, rec::SSQ_BUF);
    const uint32_t bufLen = r[rec::SSQ_BUFLEN], seqLen = r[rec::SSQ_SEQLEN];
    gcfp values = rec_ptr(r, rec::SSQ_SEQ) + 2u * seqLen;
    auto evTime = [&](uint32_t k) { gcup p = (gcup)(rec_ptr(r, rec::SSQ_SEQ)) + 2u * k; return rec_ld_f64(p, 0); };
    const bool ok = member_nin(c, m) >= 1 && (flags & 1u) && seqLen > 0 && (flags & 2u) && buf != nullptr && dur > 0.0;
    uint32_t avail[2] = {0u, 0u};
    float g0[2] = {0.0f, 0.0f};
    if (ok) {
        auto setTarget = [&](int q, float g) { target[q] = g; step[q] = (g < gain[q]) ? -fabsf(step[q]) : fabsf(step[q]); };   // :33-41
        auto toPos = [&](double p, uint32_t outOfRange) { return (p >= 0.0 && p < 4.0e9) ? (uint32_t)p : outOfRange; };
        const double t = (double)fetch(c, opnd_uniform(c, m, 0), 0);                         // :332
        const bool update = (prev < 0 && next < 0) || (prev >= 0 && t <= evTime((uint32_t)prev) + 1e-6)
                         || (next >= 0 && t >= evTime((uint32_t)next) - 1e-6);               // :337-339
        bool aligned = true;                                                                  // :94-103
        if (fabsf(target[active] - 1.0f) <= 1e-6f) {
            const double p = ((t - start[active]) / rtDur) * (double)(bsz[active] - 1u);
            const int np = (fabs(p) < 9.2e18) ? (int)(long long)p : 0;
            const int delta = (int)pos[active] - np;
            aligned = abs(delta) < 16;
        }
        if (update || !aligned) {                                                             // updateEventBoundaries :257-281
            uint32_t lo = 0, hi = seqLen;                                                     // upper_bound(t)
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (evTime(mid) > t) hi = mid; else lo = mid + 1; }
            next = lo < seqLen ? (int)lo : -1;
            if (lo == 0) { prev = -1; setTarget(0, 0.0f); setTarget(1, 0.0f); }
            else {
                prev = (int)lo - 1;
                if (active == 0) setTarget(0, 0.0f); else setTarget(1, 0.0f);
                active = (active + 1u) & 1u;
                if (fabsf(values[prev] - 1.0f) <= 1e-6f) {                                    // engage :75-83
                    const int q = (int)active;
                    const double st = evTime((uint32_t)prev);
                    if (q == 0) { start[0] = st; bsz[0] = bufLen; setTarget(0, 1.0f); } else { start[1] = st; bsz[1] = bufLen; setTarget(1, 1.0f); }
                    const double p = ((t - st) / rtDur) * (double)(bufLen - 1u);
                    const uint32_t np = min(toPos(p, bufLen), bufLen);
                    if (q == 0) pos[0] = np; else pos[1] = np;
                }
            }
        }
        // fade ramps: gains used at frame i, serial only while moving (detail::GainFade::operator() :43-51)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            avail[q] = (pos[q] < bsz[q]) ? min(n, bsz[q] - pos[q]) : 0u;
            float g = gain[q];
            uint32_t k = 0;
            for (; k < avail[q] && g != target[q]; ++k) {
                if (c.lane == 0) lds[m.scratch + q * kSlotWords + k] = g;
                g = clampf(g + step[q], 0.0f, 1.0f);
            }
            g0[q] = g;                       // constant from frame k on
            for (uint32_t i = k + c.lane; i < avail[q]; i += 64) lds[m.scratch + q * kSlotWords + i] = g;
            gain[q] = g;
        }
    }
    (void)g0;
    WAVE_SYNC();
    for (uint32_t i = s0 + c.lane; i < s1; i += 64) {
        float acc = 0.0f;                                                                     // :374-377
        if (ok) {
            if (i < avail[0]) acc += buf[pos[0] + i] * lds[m.scratch + i] ;
            if (i < avail[1]) acc += buf[pos[1] + i] * lds[m.scratch + kSlotWords + i];
        }
        put(c, m, i, acc);
    }
    WAVE_SYNC();
    if (c.lane == 0) {
        rec_st_f64(r, rec::SSQ_RTDUR, rtDur);
        r[rec::SSQ_PREV] = (uint32_t)prev; r[rec::SSQ_NEXT] = (uint32_t)next; r[rec::SSQ_ACTIVE] = active; r[rec::SSQ_FLAGS] = flags;
        r[rec::SSQ_BUFPENDING] = 0; r[rec::SSQ_SEQPENDING] = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            gup rr = r + rec::SSQ_READER0 + q * rec::SSQ_READER_DWORDS;
            rr[0] = f2u(gain[q]); rr[1] = f2u(target[q]); rr[2] = f2u(step[q]); rr[3] = pos[q] + avail[q];
            rec_st_f64(rr, 4, start[q]); rr[6] = bsz[q];
        }
    }
}

// ---- lane-per-node recurrences ---------------------------------------------------------------
// Entered by lanes [0, count) of the task's wave; `m` is that lane's node. Every operand of a
// chain lives in LDS (the planner imports HBM operands first): a block buffer (16-byte aligned,
// streamed 8 samples at a time with two ds_read_b128) or a broadcast cell (read once). Which
// operands are cells is a property of the TASK (the planner groups chain members by that mask),
// so the choice is a scalar branch.
constexpr int CH = 8;

struct SIn { uint32_t base; float cval; };

__device__ __forceinline__ SIn sin_of(uint32_t o) {
    const uint32_t kind = o & kOpKindMask, v = o & kOpValMask;
    SIn s;
    if (kind == kOpLds)        { s.base = v; s.cval = 0.0f; }
    else if (kind == kOpConst) { s.base = v; s.cval = lds[v]; }
    else                       { s.base = 0u; s.cval = 0.0f; }   // zero operand (mask bit set by the planner)
    return s;
}

// Chunked chain with the broadcast-cell mask CM known at compile time: cell operands are plain
// loop-invariant scalars (no loads, no register copies), buffer operands are streamed with two
// ds_read_b128 per 8 frames, one chunk ahead of the dependent arithmetic (ping-pong registers).
// `step(x[NIN]) -> y` carries the node state by reference.
template <int NIN, uint32_t CM, typename Step>
__device__ __forceinline__ void chain_loop_m(const SIn (&in)[NIN], uint32_t outBase, uint32_t n, Step&& step) {
    constexpr int NS0 = NIN - __builtin_popcount(CM & ((1u << NIN) - 1u));
    constexpr int NS = NS0 > 0 ? NS0 : 1;
    const uint32_t nFull = n & ~(uint32_t)(CH - 1);
    auto load = [&](uint32_t t0, float (&x)[NS][CH]) {
        int s_ = 0;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            if (!((CM >> k) & 1u)) {
                const v4f a = ld4(in[k].base + t0), b = ld4(in[k].base + t0 + 4);
                x[s_][0] = a.x; x[s_][1] = a.y; x[s_][2] = a.z; x[s_][3] = a.w;
                x[s_][4] = b.x; x[s_][5] = b.y; x[s_][6] = b.z; x[s_][7] = b.w;
                ++s_;
            }
        }
    };
    auto run8 = [&](const float (&x)[NS][CH], uint32_t t0) {
        float y[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            float xs[NIN];
            int s_ = 0;
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                if ((CM >> k) & 1u) xs[k] = in[k].cval;
                else xs[k] = x[s_++][j];
            }
            y[j] = step(xs);
        }
        v4f a, b;
        a.x = y[0]; a.y = y[1]; a.z = y[2]; a.w = y[3]; b.x = y[4]; b.y = y[5]; b.z = y[6]; b.w = y[7];
        st4(outBase + t0, a);
        st4(outBase + t0 + 4, b);
    };
    float A[NS][CH], B[NS][CH];
    uint32_t t0 = 0;
    if (nFull && NS0 > 0) load(0, A);
    while (t0 + 2 * CH <= nFull) {
        if (NS0 > 0) load(t0 + CH, B);
        run8(A, t0);
        if (NS0 > 0) load((t0 + 2 * CH < nFull) ? t0 + 2 * CH : t0, A);   // nothing left: harmless re-read
        run8(B, t0 + CH);
        t0 += 2 * CH;
    }
    if (t0 < nFull) { run8(A, t0); t0 += CH; }
    for (uint32_t t = nFull; t < n; ++t) {
        float xs[NIN];
#pragma unroll
        for (int k = 0; k < NIN; ++k) xs[k] = ((CM >> k) & 1u) ? in[k].cval : lds[in[k].base + t];
        lds[outBase + t] = step(xs);
    }
}

// Dispatch on the task's cell mask (wave-uniform): every combination for up to 3 operands; for the
// 6-operand biquad the two shapes that occur (coefficients constant / everything a signal).
template <int NIN, typename Step>
__device__ __forceinline__ void chain_loop(const SIn (&in)[NIN], uint32_t cmask, uint32_t outBase, uint32_t n, Step&& step) {
    cmask &= (1u << NIN) - 1u;
    if constexpr (NIN == 1) {
        if (cmask) chain_loop_m<1, 1u>(in, outBase, n, step); else chain_loop_m<1, 0u>(in, outBase, n, step);
    } else if constexpr (NIN == 2) {
        switch (cmask) {
            case 0: chain_loop_m<2, 0u>(in, outBase, n, step); break;
            case 1: chain_loop_m<2, 1u>(in, outBase, n, step); break;
            case 2: chain_loop_m<2, 2u>(in, outBase, n, step); break;
            default: chain_loop_m<2, 3u>(in, outBase, n, step); break;
        }
    } else if constexpr (NIN == 3) {
        switch (cmask) {
            case 0: chain_loop_m<3, 0u>(in, outBase, n, step); break;
            case 1: chain_loop_m<3, 1u>(in, outBase, n, step); break;
            case 2: chain_loop_m<3, 2u>(in, outBase, n, step); break;
            case 3: chain_loop_m<3, 3u>(in, outBase, n, step); break;
            case 4: chain_loop_m<3, 4u>(in, outBase, n, step); break;
            case 5: chain_loop_m<3, 5u>(in, outBase, n, step); break;
            case 6: chain_loop_m<3, 6u>(in, outBase, n, step); break;
            default: chain_loop_m<3, 7u>(in, outBase, n, step); break;
        }
    } else {
        static_assert(NIN == 6, "chain arity");
        if (cmask == 0x1Fu) chain_loop_m<6, 0x1Fu>(in, outBase, n, step);
        else {
            // mixed shapes: materialise the cells of constant operands as (degenerate) buffers is not
            // possible, so fall back to per-frame reads — correct for any mask, just slower
            for (uint32_t t = 0; t < n; ++t) {
                float xs[NIN];
#pragma unroll
                for (int k = 0; k < NIN; ++k) xs[k] = ((cmask >> k) & 1u) ? in[k].cval : lds[in[k].base + t];
                lds[outBase + t] = step(xs);
            }
        }
    }
}

__device__ __forceinline__ void szero(const Member& m, uint32_t n) {
    for (uint32_t t = 0; t < n; ++t) lds[m.outLds + t] = 0.0f;
}

__device__ __forceinline__ float change_tick(float& lastIn, float xn) {   // helpers/Change.h:20-31
    const float dt = xn - lastIn;
    lastIn = xn;
    return (dt > 0.0f) ? 1.0f : ((dt < 0.0f) ? -1.0f : 0.0f);
}

// PhasorNode (Core.h:85-136): step = f * (1/sr) in float; phase = next - floor(next)
__device__ __forceinline__ void ser_phasor(const Ctx& c, const Member& m, uint32_t cm) {
    gup r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 1) return szero(m, c.n);
    const SIn in[1] = {sin_of(opnd_lane(c, m, 0))};
    float phase = u2f(r[rec::S0]);
    const float rsr = 1.0f / c.srF;
    if ((cm & 1u) && in[0].cval * rsr >= 0.0f && phase >= 0.0f) {
        // constant non-negative step: next >= 0, so next - floor(next) is exactly v_fract_f32(next)
        const SIn st[1] = {SIn{0u, in[0].cval * rsr}};
        chain_loop_m<1, 1u>(st, m.outLds, c.n, [&](const float (&x)[1]) {
            const float y = phase;
            phase = __builtin_amdgcn_fractf(phase + x[0]);
            return y;
        });
    } else {
        chain_loop<1>(in, cm, m.outLds, c.n, [&](const float (&x)[1]) {
            const float stepv = x[0] * rsr;
            const float y = phase;
            const float next = phase + stepv;
            phase = next - floorf(next);
            return y;
        });
    }
    r[rec::S0] = f2u(phase);
}

__device__ __forceinline__ void ser_sphasor(const Ctx& c, const Member& m, uint32_t cm) {
    gup r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 2) return szero(m, c.n);
    const SIn in[2] = {sin_of(opnd_lane(c, m, 0)), sin_of(opnd_lane(c, m, 1))};
    float phase = u2f(r[rec::S0]), lastIn = u2f(r[rec::S1]);
    const float rsr = 1.0f / c.srF;
    chain_loop<2>(in, cm, m.outLds, c.n, [&](const float (&x)[2]) {
        if (change_tick(lastIn, x[1]) > 0.5f) phase = 0.0f;
        const float stepv = x[0] * rsr;
        const float y = phase;
        const float next = phase + stepv;
        phase = next - floorf(next);
        return y;
    });
    r[rec::S0] = f2u(phase); r[rec::S1] = f2u(lastIn);
}

// OnePoleNode (Filters.h:13-39): z = x + p*z
__device__ __forceinline__ void ser_pole(const Ctx& c, const Member& m, uint32_t cm) {
    gup r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 2) return szero(m, c.n);
    const SIn in[2] = {sin_of(opnd_lane(c, m, 0)), sin_of(opnd_lane(c, m, 1))};
    float z = u2f(r[rec::S0]);
    chain_loop<2>(in, cm, m.outLds, c.n, [&](const float (&x)[2]) { z = x[1] + x[0] * z; return z; });
    r[rec::S0] = f2u(z);
}

// EnvelopeNode (Filters.h:46-79)
__device__ __forceinline__ void ser_env(const Ctx& c, const Member& m, uint32_t cm) {
    gup r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 3) return szero(m, c.n);
    const SIn in[3] = {sin_of(opnd_lane(c, m, 0)), sin_of(opnd_lane(c, m, 1)), sin_of(opnd_lane(c, m, 2))};
    float z = u2f(r[rec::S0]);
    chain_loop<3>(in, cm, m.outLds, c.n, [&](const float (&x)[3]) {
        const float vn = fabsf(x[2]);
        if (fabsf(vn) > z) z = x[0] * (z - vn) + vn;
        else               z = x[1] * (z - vn) + vn;
        return z;
    });
    r[rec::S0] = f2u(z);
}

// BiquadFilterNode (Filters.h:87-120), TDF-II with coefficient signals
__device__ __forceinline__ void ser_biquad(const Ctx& c, const Member& m, uint32_t cm) {
    gup r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 6) return szero(m, c.n);
    SIn in[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) in[k] = sin_of(opnd_lane(c, m, k));
    float z1 = u2f(r[rec::S0]), z2 = u2f(r[rec::S1]);
    chain_loop<6>(in, cm, m.outLds, c.n, [&](const float (&x)[6]) {
        const float xx = x[5];
        const float y = x[0] * xx + z1;
        z1 = x[1] * xx - x[3] * y + z2;
        z2 = x[2] * xx - x[4] * y;
        return y;
    });
    r[rec::S0] = f2u(z1); r[rec::S1] = f2u(z2);
}

// CounterNode / AccumNode / LatchNode / MaxHold / OnceNode (Core.h:183-404)
__device__ __forceinline__ void ser_counter(const Ctx& c, const Member& m, uint32_t cm) {
    gup r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 1) return szero(m, c.n);
    const SIn in[1] = {sin_of(opnd_lane(c, m, 0))};
    float count = u2f(r[rec::S0]);
    chain_loop<1>(in, cm, m.outLds, c.n, [&](const float (&x)[1]) {
        if ((1.0f - x[0]) <= FLT_EPSILON) { const float y = count; count = count + 1.0f; return y; }
        count = 0.0f;
        return 0.0f;
    });
    r[rec::S0] = f2u(count);
}

__device__ __forceinline__ void ser_accum(const Ctx& c, const Member& m, uint32_t cm) {
    gup r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 2) return szero(m, c.n);
    const SIn in[2] = {sin_of(opnd_lane(c, m, 0)), sin_of(opnd_lane(c, m, 1))};
    float total = u2f(r[rec::S0]), lastIn = u2f(r[rec::S1]);
    chain_loop<2>(in, cm, m.outLds, c.n, [&](const float (&x)[2]) {
        if (change_tick(lastIn, x[1]) > 0.5f) total = 0.0f;
        total += x[0];
        return total;
    });
    r[rec::S0] = f2u(total); r[rec::S1] = f2u(lastIn);
}

__device__ __forceinline__ void ser_latch(const Ctx& c, const Member& m, uint32_t cm) {
    gup r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 2) return szero(m, c.n);
    const SIn in[2] = {sin_of(opnd_lane(c, m, 0)), sin_of(opnd_lane(c, m, 1))};
    float z = u2f(r[rec::S0]), hold = u2f(r[rec::S1]);
    chain_loop<2>(in, cm, m.outLds, c.n, [&](const float (&x)[2]) {
        if (fabsf(z) <= FLT_EPSILON && x[0] > FLT_EPSILON) hold = x[1];
        z = x[0];
        return hold;
    });
    r[rec::S0] = f2u(z); r[rec::S1] = f2u(hold);
}

__device__ __forceinline__ void ser_maxhold(const Ctx& c, const Member& m, uint32_t cm) {
    gup r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 2) return szero(m, c.n);
    const SIn in[2] = {sin_of(opnd_lane(c, m, 0)), sin_of(opnd_lane(c, m, 1))};
    const uint32_t hts = r[rec::P0];
    float lastIn = u2f(r[rec::S0]); uint32_t at = r[rec::S1]; float mx = u2f(r[rec::S2]);
    chain_loop<2>(in, cm, m.outLds, c.n, [&](const float (&x)[2]) {
        if (change_tick(lastIn, x[1]) > 0.5f || ++at >= hts) { mx = x[0]; at = 0; }
        else if (x[0] > mx) { at = 0; mx = x[0]; }
        return mx;
    });
    r[rec::S0] = f2u(lastIn); r[rec::S1] = at; r[rec::S2] = f2u(mx);
}

__device__ __forceinline__ void ser_once(const Ctx& c, const Member& m, uint32_t cm) {
    gup r = c.recs + m.rec * kRecDwords;
    if (member_nin(c, m) < 1) return szero(m, c.n);
    const SIn in[1] = {sin_of(opnd_lane(c, m, 0))};
    const bool isArmed = u2f(r[rec::S2]) != 0.0f;    // atomic<FloatType> armed, loaded once per block
    float gain = u2f(r[rec::S0]), lastIn = u2f(r[rec::S1]);
    bool disarm = false;
    chain_loop<1>(in, cm, m.outLds, c.n, [&](const float (&x)[1]) {
        const float delta = change_tick(lastIn, x[0]);
        if (isArmed && delta > 0.5f) { gain = 1.0f; disarm = true; }
        if (delta < -0.5f) gain = 0.0f;
        return x[0] * gain;
    });
    r[rec::S0] = f2u(gain); r[rec::S1] = f2u(lastIn);
    if (disarm) r[rec::S2] = f2u(0.0f);
}

// SequenceNode (Core.h:407-573)
__device__ __forceinline__ void ser_seq(const Ctx& c, const Member& m, uint32_t cm) {
    gup r = c.recs + m.rec * kRecDwords;
    gcfp seq = rec_ptr(r, rec::SEQ_PTR);
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
    const SIn in[2] = {sin_of(opnd_lane(c, m, 0)), hasReset ? sin_of(opnd_lane(c, m, 1)) : SIn{0u, 0.0f}};
    const uint32_t cm2 = hasReset ? cm : (cm | 2u);
    float chg = u2f(r[rec::SEQ_CHANGE]), rchg = u2f(r[rec::SEQ_RCHANGE]);
    chain_loop<2>(in, cm2, m.outLds, c.n, [&](const float (&x)[2]) {
        if (change_tick(rchg, x[1]) > 0.5f) idx = offset;
        if (change_tick(chg, x[0]) > 0.5f) {
            // std::min(seqIndex, size - 1): size_t arithmetic, an empty sequence reads nothing here
            if (len) holdValue = seq[min(idx, len - 1)];
            first = true;
            if ((++idx >= len) && loop) idx = 0;
        }
        if (idx < len) return hold ? holdValue : holdValue * x[0];
        return hold ? holdValue : 0.0f;
    });
    r[rec::SEQ_INDEX] = idx; r[rec::SEQ_HOLDVAL] = f2u(holdValue); r[rec::SEQ_FIRST] = first;
    r[rec::SEQ_CHANGE] = f2u(chg); r[rec::SEQ_RCHANGE] = f2u(rchg); r[rec::SEQ_HAVE] = 1;
}

// PolyBlepOscillatorNode (Oscillators.h:19-94)
__device__ __forceinline__ float blep(float phase, float inc) {
    if (phase < inc) { const float p = phase / inc; return (2.0f - p) * p - 1.0f; }
    if (phase > (1.0f - inc)) { const float p = (phase - 1.0f) / inc; return (p + 2.0f) * p + 1.0f; }
    return 0.0f;
}

// serial part: only `phase += inc; if (phase >= 1) phase -= 1`. The out slot carries inc[t] in
// (from the pre-pass) and the pre-tick phase[t] out (for the post-pass).
__device__ __forceinline__ void ser_blep_phase(const Ctx& c, const Member& m, uint32_t cm) {
    gup r = c.recs + m.rec * kRecDwords;
    float phase = u2f(r[rec::S0]);
    if (cm & 1u) {
        // constant frequency: the increment is one scalar, no pre-pass, nothing to load
        const float inc = lds[opnd_lane(c, m, 0) & kOpValMask] / c.srF;
        const SIn in[1] = {SIn{0u, inc}};
        if (inc >= 0.0f && inc < 1.0f && phase >= 0.0f && phase < 1.0f) {
            // t = phase + inc lies in [0, 2): `if (t >= 1) t -= 1` == t - floor(t) exactly == v_fract_f32
            chain_loop_m<1, 1u>(in, m.outLds, c.n, [&](const float (&x)[1]) {
                const float y = phase;
                phase = __builtin_amdgcn_fractf(phase + x[0]);
                return y;
            });
        } else {
            chain_loop_m<1, 1u>(in, m.outLds, c.n, [&](const float (&x)[1]) {
                const float y = phase;
                phase += x[0];
                if (phase >= 1.0f) phase -= 1.0f;
                return y;
            });
        }
    } else {
        const SIn in[1] = {SIn{m.outLds, 0.0f}};
        chain_loop_m<1, 0u>(in, m.outLds, c.n, [&](const float (&x)[1]) {
            const float y = phase;
            phase += x[0];
            if (phase >= 1.0f) phase -= 1.0f;
            return y;
        });
    }
    r[rec::S0] = f2u(phase);
}

__device__ __forceinline__ void ser_blep_acc(const Ctx& c, const Member& m) {   // triangle integrator (:58-59)
    gup r = c.recs + m.rec * kRecDwords;
    float acc = u2f(r[rec::S1]);
    const SIn in[1] = {SIn{m.outLds, 0.0f}};
    chain_loop<1>(in, 0u, m.outLds, c.n, [&](const float (&x)[1]) { acc += x[0]; return acc; });
    r[rec::S1] = f2u(acc);
}

// ---- double-state filters as wave scans -----------------------------------------------------------
// SVF / shelf / mm1p keep their state in double in the reference (filters/SVF.h:112-120,
// MultiMode1p.h:110); their per-sample update is an affine map of the state whose coefficients
// depend only on the inputs, so one wave evaluates a whole block as a scan of affine maps:
// lane l owns samples [8l, 8l+8), composes them locally, a 6-step Kogge-Stone scan over the 64
// lane aggregates yields each lane's incoming state, and the lane then replays its 8 samples to
// form the outputs. Re-association moves results by O(1e-16) relative — far inside the 1e-6 bar
// for these (stable, contractive) recurrences — and removes the 512-step f64 dependency chain.
struct Aff2 { double m11, m12, m21, m22, q1, q2; };   // s' = M s + q

__device__ __forceinline__ Aff2 aff2_after(const Aff2& f, const Aff2& g) {   // apply f, then g
    Aff2 r;
    r.m11 = g.m11 * f.m11 + g.m12 * f.m21; r.m12 = g.m11 * f.m12 + g.m12 * f.m22;
    r.m21 = g.m21 * f.m11 + g.m22 * f.m21; r.m22 = g.m21 * f.m12 + g.m22 * f.m22;
    r.q1 = g.m11 * f.q1 + g.m12 * f.q2 + g.q1;
    r.q2 = g.m21 * f.q1 + g.m22 * f.q2 + g.q2;
    return r;
}
__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    const int lo = __shfl_up(__double2loint(v), d), hi = __shfl_up(__double2hiint(v), d);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_f64(double v, int l) {
    const int lo = __shfl(__double2loint(v), l), hi = __shfl(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// Exclusive scan over lanes of per-lane aggregates; returns the composition of lanes [0, lane).
__device__ __forceinline__ Aff2 aff2_exclusive_scan(Aff2 a, uint32_t lane, Aff2& total) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Aff2 p;
        p.m11 = shfl_up_f64(a.m11, d); p.m12 = shfl_up_f64(a.m12, d); p.m21 = shfl_up_f64(a.m21, d);
        p.m22 = shfl_up_f64(a.m22, d); p.q1 = shfl_up_f64(a.q1, d); p.q2 = shfl_up_f64(a.q2, d);
        if (lane >= (uint32_t)d) a = aff2_after(p, a);
    }
    total.m11 = shfl_f64(a.m11, 63); total.m12 = shfl_f64(a.m12, 63); total.m21 = shfl_f64(a.m21, 63);
    total.m22 = shfl_f64(a.m22, 63); total.q1 = shfl_f64(a.q1, 63); total.q2 = shfl_f64(a.q2, 63);
    Aff2 e;
    e.m11 = shfl_up_f64(a.m11, 1); e.m12 = shfl_up_f64(a.m12, 1); e.m21 = shfl_up_f64(a.m21, 1);
    e.m22 = shfl_up_f64(a.m22, 1); e.q1 = shfl_up_f64(a.q1, 1); e.q2 = shfl_up_f64(a.q2, 1);
    if (lane == 0) { e.m11 = 1.0; e.m12 = 0.0; e.m21 = 0.0; e.m22 = 1.0; e.q1 = 0.0; e.q2 = 0.0; }
    return e;
}

// SVF / shelf coefficient pre-pass (updateCoeffs, filters/SVF.h:72-80, SVFShelf.h:66-83): the
// double tan / pow / divisions are the expensive, state-free part, so they run sample-parallel on
// every free wave one stage before the scan and land in the member's scratch as doubles:
// a1,a2,a3 (svf) or a1,a2,a3,k,A (shelf), each array spanning two float slots.
template <bool Shelf, int V>
__device__ __forceinline__ void run_svf_coef(const Ctx& c, const Member& m, uint32_t i, uint32_t nlim) {
    if (member_nin(c, m) < (Shelf ? 4u : 3u)) return;
    const uint32_t mode = Shelf ? UNI(c.recs[m.rec * kRecDwords + rec::P0]) : 0u;
    double* A1 = reinterpret_cast<double*>(lds + m.scratch);
    double* A2 = A1 + kSlotWords; double* A3 = A2 + kSlotWords;
    double* KK = A3 + kSlotWords; double* AA = KK + kSlotWords;
    const double sr = c.sr;
    float fcv[V], qv[V], gv[V];
    vload<V>(pin_of(c, opnd_uniform(c, m, 0)), i, fcv);
    vload<V>(pin_of(c, opnd_uniform(c, m, 1)), i, qv);
    if (Shelf) vload<V>(pin_of(c, opnd_uniform(c, m, 2)), i, gv);
#pragma unroll
    for (int q_ = 0; q_ < V; ++q_) {
        const uint32_t t = i + q_;
        const double fc = (double)fcv[q_], q = (double)qv[q_];
        double g = tan(3.14159265359 * clampd(fc, 20.0, sr / 2.0001) / sr);
        double k = 1.0 / clampd(q, 0.25, 20.0);
        if (Shelf) {
            const double A = pow(10.0, (double)gv[q_] / 40.0);
            if (mode == 0) g /= A;
            if (mode == 1) g *= A;
            if (mode == 2) k /= A;
            if (t < nlim) { KK[t] = k; AA[t] = A; }
        }
        const double a1 = 1.0 / (1.0 + g * (g + k));
        const double a2 = g * a1;
        if (t < nlim) { A1[t] = a1; A2[t] = a2; A3[t] = g * a2; }
    }
}

// StateVariableFilterNode (filters/SVF.h:48-105) and StateVariableShelfFilterNode
// (filters/SVFShelf.h:44-124), whole wave on one node.
//   tick:  v3 = v0 - ic2; v1 = ic1*a1 + v3*a2; v2 = ic2 + ic1*a2 + v3*a3; ic1 = 2 v1 - ic1; ic2 = 2 v2 - ic2
//   =>     ic1' = (2a1-1) ic1 - 2a2 ic2 + 2a2 v0 ;  ic2' = 2a2 ic1 + (1-2a3) ic2 + 2a3 v0
template <bool Shelf>
__device__ __forceinline__ void scan_svf(const Ctx& c, const Member& m) {
    gup r = c.recs + m.rec * kRecDwords;
    const uint32_t n = c.n, need = Shelf ? 4u : 3u;
    if (member_nin(c, m) < need) { for (uint32_t i = c.lane; i < n; i += 64) lds[m.outLds + i] = 0.0f; return; }
    const uint32_t mode = UNI(r[rec::P0]);
    const PIn pQ = pin_of(c, opnd_uniform(c, m, 1));
    const PIn pX = pin_of(c, opnd_uniform(c, m, Shelf ? 3 : 2));
    const double* A1 = reinterpret_cast<const double*>(lds + m.scratch);
    const double* A2 = A1 + kSlotWords; const double* A3 = A2 + kSlotWords;
    const double* KK = A3 + kSlotWords; const double* AA = KK + kSlotWords;
    const uint32_t t0 = c.lane * 8u;
    double a1[8], a2[8], a3[8], v0[8];
    {   // lane-owned 8 samples: 16-byte LDS reads
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const v2d p1 = ld2d(&A1[t0 + j]);
            const v2d p2 = ld2d(&A2[t0 + j]);
            const v2d p3 = ld2d(&A3[t0 + j]);
            a1[j] = p1.x; a1[j + 1] = p1.y; a2[j] = p2.x; a2[j + 1] = p2.y; a3[j] = p3.x; a3[j + 1] = p3.y;
        }
        if (pX.g) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v0[j] = (double)pX.g[t0 + j];
        } else if (pX.step) {
            const v4f xa = ld4(pX.base + t0);
            const v4f xb = ld4(pX.base + t0 + 4);
            v0[0] = xa.x; v0[1] = xa.y; v0[2] = xa.z; v0[3] = xa.w; v0[4] = xb.x; v0[5] = xb.y; v0[6] = xb.z; v0[7] = xb.w;
        } else {
            const double x = (double)lds[pX.base];
#pragma unroll
            for (int j = 0; j < 8; ++j) v0[j] = x;
        }
    }
    Aff2 agg; agg.m11 = 1.0; agg.m12 = 0.0; agg.m21 = 0.0; agg.m22 = 1.0; agg.q1 = 0.0; agg.q2 = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (t0 + j < n) {
            Aff2 s;
            s.m11 = 2.0 * a1[j] - 1.0; s.m12 = -2.0 * a2[j];
            s.m21 = 2.0 * a2[j];       s.m22 = 1.0 - 2.0 * a3[j];
            s.q1 = 2.0 * a2[j] * v0[j]; s.q2 = 2.0 * a3[j] * v0[j];
            agg = aff2_after(agg, s);
        }
    }
    Aff2 total;
    const Aff2 pre = aff2_exclusive_scan(agg, c.lane, total);
    const double s1 = rec_ld_f64(r, rec::S0), s2 = rec_ld_f64(r, rec::S2);
    double ic1 = pre.m11 * s1 + pre.m12 * s2 + pre.q1;
    double ic2 = pre.m21 * s1 + pre.m22 * s2 + pre.q2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t t = t0 + j;
        if (t < n) {
            const double v3 = v0[j] - ic2;
            const double v1 = ic1 * a1[j] + v3 * a2[j];
            const double v2 = ic2 + ic1 * a2[j] + v3 * a3[j];
            ic1 = v1 * 2.0 - ic1;
            ic2 = v2 * 2.0 - ic2;
            float y;
            if (!Shelf) {   // SVF.h:57-69
                if (mode == 0)      y = (float)v2;
                else if (mode == 1) y = (float)v1;
                else {
                    const float qf = pQ.g ? pQ.g[t] : lds[pQ.base + t * pQ.step];
                    const double k = 1.0 / clampd((double)qf, 0.25, 20.0);
                    if (mode == 2)      y = (float)(v0[j] - k * v1 - v2);
                    else if (mode == 3) y = (float)(v0[j] - k * v1);
                    else                y = (float)(v0[j] - 2.0 * k * v1);
                }
            } else {        // SVFShelf.h:54-63
                const double A = AA[t], k = KK[t];
                if (mode == 2)      y = (float)(v0[j] + k * (A * A - 1.0) * v1);
                else if (mode == 0) y = (float)(v0[j] + k * (A - 1.0) * v1 + (A * A - 1.0) * v2);
                else                y = (float)(A * A * v0[j] + k * (1.0 - A) * A * v1 + (1.0 - A * A) * v2);
            }
            lds[m.outLds + t] = y;
        }
    }
    if (c.lane == 0) {
        rec_st_f64(r, rec::S0, total.m11 * s1 + total.m12 * s2 + total.q1);
        rec_st_f64(r, rec::S2, total.m21 * s1 + total.m22 * s2 + total.q2);
    }
}

// MultiMode1p (filters/MultiMode1p.h:79-107): v = (x - z) G; lp = v + z; z' = lp + v
//   =>  z' = (1 - 2G) z + 2G x  (1-D affine scan); lp is formed from the pre-sample state.
__device__ __forceinline__ void scan_mm1p(const Ctx& c, const Member& m) {
    gup r = c.recs + m.rec * kRecDwords;
    const uint32_t n = c.n;
    if (member_nin(c, m) < 2) { for (uint32_t i = c.lane; i < n; i += 64) lds[m.outLds + i] = 0.0f; return; }
    const uint32_t mode = UNI(r[rec::P0]);
    const PIn pGn = pin_of(c, opnd_uniform(c, m, 0)), pX = pin_of(c, opnd_uniform(c, m, 1));
    const uint32_t t0 = c.lane * 8u;
    double G[8]; float xs[8];
    double am = 1.0, aq = 0.0;   // z' = am z + aq over this lane's samples
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t t = t0 + j;
        const bool live = t < n;
        const uint32_t tc = live ? t : 0u;
        auto rd = [&](const PIn& p) { return p.g ? p.g[tc] : lds[p.base + tc * p.step]; };
        const double g = clampd((double)rd(pGn), 0.0, 0.9999);
        G[j] = g / (1.0 + g);
        xs[j] = rd(pX);
        if (live) {
            const double mm = 1.0 - 2.0 * G[j], qq = 2.0 * G[j] * (double)xs[j];
            am = mm * am; aq = mm * aq + qq;
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double pm = shfl_up_f64(am, d), pq = shfl_up_f64(aq, d);
        if (c.lane >= (uint32_t)d) { aq = am * pq + aq; am = am * pm; }
    }
    const double tm = shfl_f64(am, 63), tq = shfl_f64(aq, 63);
    double em = shfl_up_f64(am, 1), eq = shfl_up_f64(aq, 1);
    if (c.lane == 0) { em = 1.0; eq = 0.0; }
    const double z0 = rec_ld_f64(r, rec::S0);
    double z = em * z0 + eq;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t t = t0 + j;
        if (t < n) {
            const float xn = xs[j];
            const double v = ((double)xn - z) * G[j];
            const double lp = v + z;
            z = lp + v;
            float y;
            if (mode == 0)      y = (float)lp;
            else if (mode == 2) y = xn - (float)lp;
            else                y = (float)(lp + lp - (double)xn);
            lds[m.outLds + t] = y;
        }
    }
    if (c.lane == 0) rec_st_f64(r, rec::S0, tm * z0 + tq);
}

// ---- task dispatch ------------------------------------------------------------------------------
struct TaskU { uint32_t opcode, stage, flags, s0, s1, first, count, o0, o1, outLds, nin, outHbm; };

// Visit the members of a task (wave-uniform). Sample-parallel math ops of a single-member task
// (`lite`) take everything from the task header; other ops also need the member's record / scratch.
template <bool Lite, typename F>
__device__ __forceinline__ void for_members_x(const Ctx& c, const TaskU& t, F&& f) {
    if (Lite && t.count == 1 && t.nin != 0xFFFEu) {
        Member m;
        m.rec = kNone; m.opnd = kNone; m.scratch = kNone;
        m.nin = t.nin == 0xFFFFu ? (uint32_t)kNone : t.nin;
        m.outLds = t.outLds == 0xFFFFu ? (uint32_t)kNone : t.outLds;
        m.outHbm = t.outHbm; m.pad0_ = t.o0; m.pad1_ = t.o1;
        if (m.nin > 2u && m.nin != kNone) {   // operands beyond the inlined two live in the table
            m.opnd = UNI(ldsu(c.members + t.first * 8u + 1u));
        } else if (m.nin == kNone) {
            m.opnd = UNI(ldsu(c.members + t.first * 8u + 1u));
        }
        f(m);
        return;
    }
    for (uint32_t k = 0; k < t.count; ++k) { const Member m = member_uniform(c, t.first + k); f(m); }
}
template <typename F>
__device__ __forceinline__ void for_members(const Ctx& c, const TaskU& t, F&& f) { for_members_x<false>(c, t, f); }

// Stateful task: whole-wave scans for the double-state filters; otherwise a sample-parallel
// pre-pass over all members, the lane-per-member chain, and a sample-parallel post-pass.
__device__ __forceinline__ void run_stateful(const Ctx& c, const TaskU& t) {
    const uint32_t n = c.n;
    const uint32_t op = t.opcode;
    const uint32_t cm = t.flags;
    if (op == OP_SVF || op == OP_SVFSHELF || op == OP_MM1P) {
        for_members(c, t, [&](const Member& m) {
            if (op == OP_SVF) scan_svf<false>(c, m);
            else if (op == OP_SVFSHELF) scan_svf<true>(c, m);
            else scan_mm1p(c, m);
        });
    } else {
        const bool isBlep = (op == OP_BLEPSAW || op == OP_BLEPSQUARE || op == OP_BLEPTRIANGLE);
        if (isBlep && !(cm & 1u)) {   // pre-pass: inc[t] = f/sr into the out slot (skipped for a constant frequency)
            const float sr = c.srF;
            for_members(c, t, [&](const Member& m) {
                if (member_nin(c, m) < 1) return;
                const PIn p = pin_of(c, opnd_uniform(c, m, 0));
                if (n == 512) {   // full block: 8 consecutive frames per lane, vector LDS traffic
                    float x[8];
                    vload<8>(p, c.lane * 8u, x);
                    v4f a, b;
                    a.x = x[0] / sr; a.y = x[1] / sr; a.z = x[2] / sr; a.w = x[3] / sr;
                    b.x = x[4] / sr; b.y = x[5] / sr; b.z = x[6] / sr; b.w = x[7] / sr;
                    st4(m.outLds + c.lane * 8u, a); st4(m.outLds + c.lane * 8u + 4u, b);
                } else {
                    for (uint32_t i = c.lane; i < n; i += 64) lds[m.outLds + i] = pget(p, i) / sr;
                }
            });
            WAVE_SYNC();
        }
        if (c.lane < t.count && n > 0) {
            const Member m = member_lane(c, t.first + c.lane);
            switch (op) {
                case OP_PHASOR:   ser_phasor(c, m, cm); break;
                case OP_SPHASOR:  ser_sphasor(c, m, cm); break;
                case OP_POLE:     ser_pole(c, m, cm); break;
                case OP_ENV:      ser_env(c, m, cm); break;
                case OP_BIQUAD:   ser_biquad(c, m, cm); break;
                case OP_COUNTER:  ser_counter(c, m, cm); break;
                case OP_ACCUM:    ser_accum(c, m, cm); break;
                case OP_LATCH:    ser_latch(c, m, cm); break;
                case OP_MAXHOLD:  ser_maxhold(c, m, cm); break;
                case OP_ONCE:     ser_once(c, m, cm); break;
                case OP_SEQ:      ser_seq(c, m, cm); break;
                case OP_BLEPSAW: case OP_BLEPSQUARE: case OP_BLEPTRIANGLE:
                    if (member_nin(c, m) < 1) szero(m, n); else ser_blep_phase(c, m, cm);
                    break;
                default: break;
            }
        }
        WAVE_SYNC();
        if (isBlep) {   // post-pass: waveform from (phase, inc)
            const float sr = c.srF;
            for_members(c, t, [&](const Member& m) {
                if (member_nin(c, m) < 1) return;
                const PIn p = pin_of(c, opnd_uniform(c, m, 0));
                auto wave = [&](float f, float phase) {
                    const float inc = f / sr;
                    if (op == OP_BLEPSAW) return 2.0f * phase - 1.0f - blep(phase, inc);
                    const float naive = phase < 0.5f ? 1.0f : -1.0f;
                    const float halfPhase = fmodf(phase + 0.5f, 1.0f);
                    const float square = naive + blep(phase, inc) - blep(halfPhase, inc);
                    return (op == OP_BLEPSQUARE) ? square : (4.0f * inc * square);
                };
                if (n == 512) {
                    float f[8], ph[8];
                    vload<8>(p, c.lane * 8u, f);
                    vload<8>(PIn{m.outLds, 1u, (gcfp)nullptr}, c.lane * 8u, ph);
                    v4f a, b;
                    a.x = wave(f[0], ph[0]); a.y = wave(f[1], ph[1]); a.z = wave(f[2], ph[2]); a.w = wave(f[3], ph[3]);
                    b.x = wave(f[4], ph[4]); b.y = wave(f[5], ph[5]); b.z = wave(f[6], ph[6]); b.w = wave(f[7], ph[7]);
                    st4(m.outLds + c.lane * 8u, a); st4(m.outLds + c.lane * 8u + 4u, b);
                } else {
                    for (uint32_t i = c.lane; i < n; i += 64) lds[m.outLds + i] = wave(pget(p, i), lds[m.outLds + i]);
                }
            });
            if (op == OP_BLEPTRIANGLE) {
                WAVE_SYNC();
                if (c.lane < t.count && n > 0) {
                    const Member m = member_lane(c, t.first + c.lane);
                    if (member_nin(c, m) >= 1) ser_blep_acc(c, m);
                }
            }
        }
    }
    WAVE_SYNC();
    // chain outputs are produced in LDS; copy out the ones another island (or the epilogue) reads
    for_members(c, t, [&](const Member& m) {
        if (m.outHbm == kNone) return;
        for (uint32_t i = c.lane; i < n; i += 64) c.hbm[(size_t)m.outHbm * c.stride + i] = lds[m.outLds + i];
    });
}

// Fast path for the overwhelmingly common sample-parallel task: one node, one or two operands that
// all live in LDS (buffers / broadcast cells), enough inputs for the node's arity. The planner
// marks such tasks (flags bit 7); everything they need is in the 8-dword header, so there is no
// operand-kind decoding and no member/operand table access: load, apply, store.
template <int V>
__device__ __forceinline__ void fast_load(uint32_t o, uint32_t i, float (&x)[V]) {
    const uint32_t w = o & kOpValMask;
    if ((o >> 30) == 0u) {
        if constexpr (V == 1) x[0] = lds[w + i];
        else if constexpr (V == 2) { const v2f a = *reinterpret_cast<const v2f*>(__builtin_assume_aligned(&lds[w + i], 8)); x[0] = a.x; x[1] = a.y; }
        else {
#pragma unroll
            for (int q = 0; q < V; q += 4) { const v4f a = ld4(w + i + q); x[q] = a.x; x[q + 1] = a.y; x[q + 2] = a.z; x[q + 3] = a.w; }
        }
    } else {
        const float v = lds[(o >> 30) == 1u ? w : 0u];
#pragma unroll
        for (int q = 0; q < V; ++q) x[q] = v;
    }
}

template <int V>
__device__ __forceinline__ void run_fast(const Ctx& c, const TaskU& t, uint32_t i) {
    float x[V], y[V];
    fast_load<V>(t.o0, i, x);
    const uint32_t op = t.opcode;
    if (op >= OP_LE) {   // binary / two-operand reduce: one switch per task, straight-line loops inside each case
        fast_load<V>(t.o1, i, y);
        switch (op) {
#define FB(OPC, EXPR) case OPC: _Pragma("unroll") for (int q = 0; q < V; ++q) x[q] = (EXPR); break;
            FB(OP_ADD, x[q] + y[q]) FB(OP_SUB, x[q] - y[q]) FB(OP_MUL, x[q] * y[q])
            FB(OP_DIV, reduce_eval(OP_DIV, x[q], y[q])) FB(OP_MIN, reduce_eval(OP_MIN, x[q], y[q])) FB(OP_MAX, reduce_eval(OP_MAX, x[q], y[q]))
            FB(OP_MOD, fmodf(x[q], y[q]))
            FB(OP_LE, binary_eval(OP_LE, x[q], y[q])) FB(OP_LEQ, binary_eval(OP_LEQ, x[q], y[q])) FB(OP_GE, binary_eval(OP_GE, x[q], y[q]))
            FB(OP_GEQ, binary_eval(OP_GEQ, x[q], y[q])) FB(OP_POW, binary_eval(OP_POW, x[q], y[q])) FB(OP_EQ, binary_eval(OP_EQ, x[q], y[q]))
            FB(OP_AND, binary_eval(OP_AND, x[q], y[q]))
            default: _Pragma("unroll") for (int q = 0; q < V; ++q) x[q] = binary_eval(OP_OR, x[q], y[q]); break;
#undef FB
        }
    } else {
        switch (op) {   // one switch per task, the loops inside each case are straight-line
#define FU(OPC) case OPC: _Pragma("unroll") for (int q = 0; q < V; ++q) x[q] = unary_eval(OPC, x[q]); break;
            FU(OP_SIN) FU(OP_COS) FU(OP_TAN) FU(OP_TANH) FU(OP_ASINH) FU(OP_LN) FU(OP_LOG) FU(OP_LOG2)
            FU(OP_CEIL) FU(OP_FLOOR) FU(OP_ROUND) FU(OP_SQRT) FU(OP_EXP)
            default: _Pragma("unroll") for (int q = 0; q < V; ++q) x[q] = fabsf(x[q]); break;
#undef FU
        }
    }
    if (t.outLds != 0xFFFFu) {
        const uint32_t w = t.outLds + i;
        if constexpr (V == 1) lds[w] = x[0];
        else if constexpr (V == 2) { v2f a; a.x = x[0]; a.y = x[1]; *reinterpret_cast<v2f*>(__builtin_assume_aligned(&lds[w], 8)) = a; }
        else {
#pragma unroll
            for (int q = 0; q < V; q += 4) { v4f a; a.x = x[q]; a.y = x[q + 1]; a.z = x[q + 2]; a.w = x[q + 3]; st4(w + q, a); }
        }
    }
    if (t.outHbm != kNone) {
        gfp g = c.hbm + (size_t)t.outHbm * c.stride + i;
        if constexpr (V == 1) g[0] = x[0];
        else if constexpr (V == 2) { v2f a; a.x = x[0]; a.y = x[1]; *(gv2)g = a; }
        else {
#pragma unroll
            for (int q = 0; q < V; q += 4) { v4f a; a.x = x[q]; a.y = x[q + 1]; a.z = x[q + 2]; a.w = x[q + 3]; *(gv4)(g + q) = a; }
        }
    }
}

// sample-parallel opcodes on the canonical mapping, V consecutive frames per lane
template <int V>
__device__ __forceinline__ void run_par(const Ctx& c, const TaskU& t, uint32_t i, uint32_t nlim) {
#define PAR(OPC, FN) case OPC: for_members(c, t, [&](const Member& m) { FN<V>(c, m, i, nlim); }); break;
#define UN(OPC)  case OPC: for_members_x<true>(c, t, [&](const Member& m) { run_unary<OPC, V>(c, m, i, nlim); }); break;
#define BI(OPC)  case OPC: for_members_x<true>(c, t, [&](const Member& m) { run_binary<OPC, V>(c, m, i, nlim); }); break;
#define RE(OPC)  case OPC: for_members_x<true>(c, t, [&](const Member& m) { run_reduce<OPC, V>(c, m, i, nlim); }); break;
    switch (t.opcode) {
        UN(OP_SIN) UN(OP_COS) UN(OP_TAN) UN(OP_TANH) UN(OP_ASINH) UN(OP_LN) UN(OP_LOG) UN(OP_LOG2)
        UN(OP_CEIL) UN(OP_FLOOR) UN(OP_ROUND) UN(OP_SQRT) UN(OP_EXP) UN(OP_ABS)
        BI(OP_LE) BI(OP_LEQ) BI(OP_GE) BI(OP_GEQ) BI(OP_POW) BI(OP_EQ) BI(OP_AND) BI(OP_OR)
        RE(OP_ADD) RE(OP_SUB) RE(OP_MUL) RE(OP_DIV) RE(OP_MOD) RE(OP_MIN) RE(OP_MAX)
        PAR(OP_IN, run_in) PAR(OP_COPY, run_copy) PAR(OP_ROOT, run_root) PAR(OP_PREWARP, run_prewarp)
        PAR(OP_TIME, run_time) PAR(OP_METRO, run_metro) PAR(OP_TAPIN, run_tapin) PAR(OP_TAPOUT, run_tapout)
        PAR(OP_CONST, run_fill) PAR(OP_SR, run_fill)
        case OP_SVF_COEF:   for_members(c, t, [&](const Member& m) { run_svf_coef<false, V>(c, m, i, nlim); }); break;
        case OP_SHELF_COEF: for_members(c, t, [&](const Member& m) { run_svf_coef<true, V>(c, m, i, nlim); }); break;
        default: break;
    }
#undef PAR
#undef UN
#undef BI
#undef RE
}

__device__ __forceinline__ void run_task(const Ctx& c, const TaskU& t, uint32_t lo, uint32_t hi) {
    switch (t.opcode) {
        // whole-block, single-wave ops with their own state handling
        case OP_RAND:   for_members(c, t, [&](const Member& m) { run_rand(c, m, 0, c.n); }); return;
        case OP_Z:      for_members(c, t, [&](const Member& m) { run_z(c, m, 0, c.n); }); return;
        case OP_SDELAY: for_members(c, t, [&](const Member& m) { run_sdelay(c, m, 0, c.n); }); return;
        case OP_DELAY:  for_members(c, t, [&](const Member& m) { run_delay(c, m, 0, c.n); }); return;
        case OP_SAMPLESEQ: for_members(c, t, [&](const Member& m) { run_sampleseq(c, m, 0, c.n); }); return;
        case OP_PHASOR: case OP_SPHASOR: case OP_COUNTER: case OP_ACCUM: case OP_LATCH: case OP_MAXHOLD: case OP_ONCE:
        case OP_SEQ: case OP_POLE: case OP_ENV: case OP_BIQUAD: case OP_MM1P: case OP_SVF: case OP_SVFSHELF:
        case OP_BLEPSAW: case OP_BLEPSQUARE: case OP_BLEPTRIANGLE:
            run_stateful(c, t);
            return;
        default: break;
    }
    // sample-parallel: frames [s0, s1) of this workgroup's slice, 64*V of them
    const uint32_t s0 = max(t.s0, lo), s1 = min(t.s1, hi);
    if (s0 >= s1 || s0 >= c.n) return;
    const uint32_t V = (s1 - s0) >> 6;
    const uint32_t nlim = min(s1, c.n);
    const uint32_t i = s0 + c.lane * V;
    switch (V) {
        case 1: run_par<1>(c, t, i, nlim); break;
        case 2: run_par<2>(c, t, i, nlim); break;
        case 4: run_par<2>(c, t, i, nlim); run_par<2>(c, t, i + 2, nlim); break;   // same lane, same 4 frames
        case 8: run_par<8>(c, t, i, nlim); break;
        default: break;   // the planner only emits 64, 128, 256 or 512-frame ranges
    }
}

// RootNode::stillRunning (Core.h:28-31) and the channel test of RootRenderSequence::process
// (GraphRenderSequence.h:214-219)
__device__ __forceinline__ bool root_running(gcup recs, uint32_t rootRec, uint32_t numOut) {
    gcup r = recs + rootRec * kRecDwords;
    const float tg = u2f(r[rec::ROOT_TARGET]), g = u2f(r[rec::ROOT_GAIN]);
    const bool on = tg > 0.5f;
    const bool settled = fabsf(tg - g) <= 1e-6f;
    const int ch = (int)r[rec::ROOT_CHANNEL];
    return (on || !settled) && ch >= 0 && (uint32_t)ch < numOut;
}

} // namespace

// ---- kernels ---------------------------------------------------------------------------------------
// One launch renders one island level for `batch` consecutive blocks (1 = the realtime path).
//   * batch == 1, or a stateless island: stages are separated by workgroup barriers; a stateless island's
//     blocks are spread over gridDim.y (nothing carries over from block to block).
//   * stateful island, batch > 1: software pipeline over blocks. The island keeps `copies` (D) blocks in
//     flight, block b in buffer set b % D with its own program copy. The planner cuts the S stages into D phases
//     of consecutive stages (balanced by estimated cost); in macro-step m every wave runs its phase-p tasks of block m - p. A task of
//     stage s waits (LDS counters) until the previous non-empty stage of ITS block is complete and until
//     block b - D has left the pipeline, so its buffer set is free. Every wait refers to work of an earlier
//     macro/micro step and all waves walk the steps in the same order, so the schedule cannot deadlock.
//     Node state lives in the node records; the same wave renders a node for every block, in block order.
__device__ __forceinline__ void wait_counter(uint32_t word, uint32_t need) {
    uint32_t* p = reinterpret_cast<uint32_t*>(lds + word);
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
}

__global__ __launch_bounds__(kThreads)
void elemhip_island_kernel(PlanView pv, uint32_t* recs, float* hbm, const Globals* g, const uint32_t* lcg,
                           uint32_t levelBegin, uint32_t batch, uint32_t arenaFloats) {
    const uint64_t tStart = clock64();
    const uint32_t entry = pv.levelIslands[levelBegin + blockIdx.x];
    const Island isl = pv.islands[entry & 0xFFFFFFu];
    const uint32_t splitIdx = entry >> 24;
    const uint32_t numOut = g->numOut;
    if (!root_running((gcup)recs, isl.rootRec, numOut)) return;
    const bool pipe = batch > 1u && !isl.stateless;
    if (pipe && blockIdx.y != 0u) return;                       // a stateful island renders all its blocks in one workgroup
    if (!pipe && blockIdx.y >= batch) return;

    // stage the island program in LDS (one coalesced copy), then fill the broadcast cells
    uint32_t* progLds = reinterpret_cast<uint32_t*>(lds + isl.ldsProg);
    gcup prog = (gcup)(pv.prog + isl.progBegin);
    for (uint32_t k = threadIdx.x; k < isl.progDwords; k += kThreads) progLds[k] = prog[k];
    if (threadIdx.x < kSlot0) lds[threadIdx.x] = 0.0f;
    for (uint32_t k = threadIdx.x; k < isl.numStages * isl.copies; k += kThreads) reinterpret_cast<uint32_t*>(lds + isl.ldsCounters)[k] = 0u;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < isl.numCells; k += kThreads) {
        const uint32_t word = progLds[isl.cellOff + 2 * k], rec_ = progLds[isl.cellOff + 2 * k + 1];
        lds[word] = u2f(((gcup)recs)[rec_ * kRecDwords + rec::P0]);
    }
    __syncthreads();

    Ctx c;
    c.recs = (gup)recs; c.hbm = (gfp)hbm; c.g = g; c.lcg = (gcup)lcg;
    c.members = isl.ldsProg + isl.memOff; c.operands = isl.ldsProg + isl.opndOff;
    c.n = g->numSamples; c.stride = g->blockStride; c.numIn = g->numIn;
    c.lane = threadIdx.x & 63u;
    c.srF = g->sampleRateF; c.sr = g->sampleRate;
    const int64_t sampleTime0 = g->sampleTime;
    c.sampleTime = sampleTime0;
    const uint32_t wave = UNI(threadIdx.x >> 6);
    // frames this workgroup renders (whole block unless the island is split)
    const uint32_t lo = isl.split > 1 ? (splitIdx * c.stride) / isl.split : 0u;
    const uint32_t hi = isl.split > 1 ? ((splitIdx + 1) * c.stride) / isl.split : 0xFFFFu;

    unsigned long long* trace = (blockIdx.x == 0 && blockIdx.y == 0 && g->trace) ? reinterpret_cast<unsigned long long*>(g->trace) : nullptr;
    const uint64_t tProlog = clock64();
    const uint32_t S = isl.numStages, D = pipe ? isl.copies : 1u;
    const uint32_t tabT = isl.ldsProg + isl.stageOff, tabPrev = tabT + S, tabBegin = tabPrev + S + wave * (S + 1u);

    const uint32_t lastStage = S - 1u;                           // the planner never leaves the last stage empty
    const uint32_t myBlocks = (batch - blockIdx.y + gridDim.y - 1u) / gridDim.y;
    const uint32_t steps = myBlocks * S;
    uint32_t traceSlot = 0;
    // pipelined walk: this wave's own (stage, phase) slots, macro-step by macro-step
    const uint32_t walkOffs = isl.ldsProg + isl.schedOff;
    const uint32_t e0 = UNI(ldsu(walkOffs + wave)), e1 = UNI(ldsu(walkOffs + wave + 1u));
    const uint32_t walkBase = walkOffs + kWaves + 1u;            // the planner pads in front of the offsets: entries are 16-byte aligned
    const uint32_t lastT = UNI(ldsu(tabT + lastStage));
    // Inside a macro-step the wave polls its slots and runs whichever is ready: slots of one macro-step belong to
    // different blocks, so their order is free, and a slot stalled on another wave must not hold back work for a
    // younger block that is already runnable. The next macro-step starts when every slot of this one has run.
    const uint32_t myEntries = e1 - e0;
    const bool ooo = pipe && myEntries > 1u && myEntries <= 32u;
    const uint32_t fullMask = myEntries >= 32u ? 0xFFFFFFFFu : ((1u << myEntries) - 1u);
    const uint32_t totalMs = (pipe && myEntries > 0u) ? batch + D - 1u : 0u;
    uint32_t pm = 0, pe = e0, doneMask = 0u;
    bool progress = false;
    uint32_t bbi = 0, bs_ = 0;                  // barrier walk: block of this workgroup, stage
    for (uint32_t it = 0; pipe ? pm < totalMs : it < steps; ++it) {
        uint32_t b, s, tb, te, use = 0u, copy = 0u;
        if (pipe) {
            const uint32_t idx = pe - e0, m = pm;
            const v4u ea = lds4u(walkBase + pe * 8u), eb = lds4u(walkBase + pe * 8u + 4u);
            bool wrapped = false;
            if (++pe == e1) { pe = e0; wrapped = true; }
            s = UNI(ea.x); b = m - UNI(ea.y);                    // wraps when m < phase: caught by b >= batch
            tb = UNI(ea.z); te = UNI(ea.w);
            const uint32_t prev = UNI(eb.x), prevT = UNI(eb.y);
            use = D == 4u ? b >> 2 : D == 5u ? b / 5u : D == 3u ? b / 3u : D == 2u ? b >> 1 : D == 6u ? b / 6u : b;
            copy = b - use * D;
            // completion counters are per (stage, buffer set): block b is use number b / D of set b % D. (One counter per
            // stage would let a wave that runs a block ahead satisfy the count meant for a slower wave's task.)
            const uint32_t wPrev = isl.ldsCounters + prev * D + copy, wLast = isl.ldsCounters + lastStage * D + copy;
            bool run = false;
            if (ooo) {
                const uint32_t bit = 1u << idx;
                if (!(doneMask & bit)) {
                    if (b >= batch) doneMask |= bit;
                    else {
                        uint32_t* cp = reinterpret_cast<uint32_t*>(lds);
                        const bool ready = (prev == kNone || __hip_atomic_load(cp + wPrev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (use + 1u) * prevT)
                                        && (use == 0u || __hip_atomic_load(cp + wLast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= use * lastT);
                        if (ready) { doneMask |= bit; run = true; }
                    }
                }
                if (doneMask == fullMask) { doneMask = 0u; ++pm; pe = e0; progress = false; }
                else if (run) progress = true;
                else if (wrapped) { if (!progress) __builtin_amdgcn_s_sleep(2); progress = false; }
            } else {
                if (wrapped) ++pm;
                if (b < batch) {
                    if (prev != kNone) wait_counter(wPrev, (use + 1u) * prevT);
                    if (use > 0u) wait_counter(wLast, use * lastT);
                    run = true;
                }
            }
            if (!run) continue;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            b = blockIdx.y + bbi * gridDim.y;
            s = bs_;
            if (++bs_ == S) { bs_ = 0u; ++bbi; }
            if (it > 0u) __syncthreads();
            tb = UNI(ldsu(tabBegin + s)); te = UNI(ldsu(tabBegin + s + 1u));
            if (tb == te) continue;
        }
        const uint32_t progBase = isl.ldsProg + copy * isl.copyDwords;
        c.members = progBase + isl.memOff; c.operands = progBase + isl.opndOff;
        c.hbm = (gfp)hbm + (size_t)b * arenaFloats;
        c.sampleTime = sampleTime0 + (int64_t)b * (int64_t)c.n;
        for (uint32_t ti = tb; ti < te; ++ti) {
            const v4u h = lds4u(progBase + ti * 8u), h2 = lds4u(progBase + ti * 8u + 4u);
            const uint32_t d0 = UNI(h.x), d1 = UNI(h.y), d6 = UNI(h2.z);
            TaskU t;
            t.opcode = d0 & 0xFFFFu; t.stage = (d0 >> 16) & 0xFFu; t.flags = d0 >> 24;
            t.s0 = d1 & 0xFFFFu; t.s1 = d1 >> 16;
            t.first = UNI(h.z); t.count = UNI(h.w);
            t.o0 = UNI(h2.x); t.o1 = UNI(h2.y); t.outLds = d6 & 0xFFFFu; t.nin = d6 >> 16; t.outHbm = UNI(h2.w);
            // ELEMHIP trace hook: [wave][slot] = {opcode | stage << 16, start, end} in shader clocks
            const uint64_t t0 = trace ? clock64() : 0;
            if ((t.flags & 0x80u) && t.s1 <= c.n && isl.split <= 1u) {
                const uint32_t V = (t.s1 - t.s0) >> 6, i = t.s0 + c.lane * V;
                if (V == 2u) run_fast<2>(c, t, i);
                else if (V == 8u) run_fast<8>(c, t, i);
                else if (V == 4u) { run_fast<2>(c, t, i); run_fast<2>(c, t, i + 2u); }
                else run_fast<1>(c, t, i);
            } else
            run_task(c, t, lo, hi);
            if (trace) {
                const uint64_t t1 = clock64();
                if (c.lane == 0 && traceSlot < 62) {
                    unsigned long long* w = trace + (size_t)wave * 192 + 3 * (traceSlot + 2);
                    w[0] = d0 | ((unsigned long long)b << 32); w[1] = t0; w[2] = t1;
                }
                ++traceSlot;
            }
        }
        if (pipe) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (c.lane == 0)
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(lds + isl.ldsCounters + s * D + copy), te - tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (trace && c.lane == 0) {
        unsigned long long* w = trace + (size_t)wave * 192;
        w[0] = traceSlot; w[1] = tStart; w[2] = tProlog; w[3] = clock64();
    }
}

// Epilogue: one workgroup. (1) zero + sum running roots into the output bus in render-sequence
// order (GraphRenderSequence.h:286-295, 227-231); (2) promote tap buffers of active roots
// (:306-308, Feedback.h:90-109); (3) advance root fades (GainFade.h:70-71); (4) advance the block.
// output bus of one block: zero + sum the running roots per channel in render-sequence order
__device__ __forceinline__ void bus_sum(const PlanView& pv, gcup recs, gcfp hbm, gfp out, uint32_t n, uint32_t numOut, uint32_t stride) {
    __shared__ int rootChan[1024];        // channel of root r if it ran this block, else -1
    __shared__ uint16_t chanList[1024];   // running roots grouped by channel, render-sequence order inside a channel
    __shared__ uint16_t chanStart[kMaxOut + 1];
    const uint32_t nr = min(pv.numRoots, 1024u);
    for (uint32_t r = threadIdx.x; r < nr; r += blockDim.x) {
        const uint32_t rr = pv.roots[r].rec;
        rootChan[r] = root_running(recs, rr, numOut) ? (int)recs[rr * kRecDwords + rec::ROOT_CHANNEL] : -1;
    }
    __syncthreads();
    if (threadIdx.x < numOut) {           // thread ch counts its roots
        uint32_t cnt = 0;
        for (uint32_t r = 0; r < nr; ++r) cnt += rootChan[r] == (int)threadIdx.x;
        chanStart[threadIdx.x + 1] = (uint16_t)cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) { chanStart[0] = 0; for (uint32_t c = 0; c < numOut; ++c) chanStart[c + 1] += chanStart[c]; }
    __syncthreads();
    if (threadIdx.x < numOut) {
        uint32_t q = chanStart[threadIdx.x];
        for (uint32_t r = 0; r < nr; ++r) if (rootChan[r] == (int)threadIdx.x) chanList[q++] = (uint16_t)r;
    }
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < numOut * n; idx += blockDim.x) {
        const uint32_t ch = idx / n, i = idx - ch * n;
        float acc = 0.0f;
        for (uint32_t q = chanStart[ch]; q < chanStart[ch + 1]; ++q) acc += hbm[(size_t)pv.roots[chanList[q]].hbm * stride + i];
        for (uint32_t r = nr; r < pv.numRoots; ++r) {   // > 1024 roots: slow path
            const RootEntry re = pv.roots[r];
            if (root_running(recs, re.rec, numOut) && recs[re.rec * kRecDwords + rec::ROOT_CHANNEL] == ch)
                acc += hbm[(size_t)re.hbm * stride + i];
        }
        out[(size_t)ch * stride + i] = acc;
    }
}

// Epilogue of a multi-block launch: workgroup b sums block b's roots into output-ring slot b. The host only
// batches blocks while every running root's fade is settled and the plan has no taps / convolvers, so there
// is no per-block state to advance besides the sample clock.
__global__ __launch_bounds__(1024)
void elemhip_epilogue_batch_kernel(PlanView pv, uint32_t* recs_, const float* hbm_, Globals* g, float* outRing,
                                   uint32_t batch, uint32_t arenaFloats) {
    const uint32_t b = blockIdx.x;
    const uint32_t n = g->numSamples, numOut = min(g->numOut, (uint32_t)kMaxOut), stride = g->blockStride;
    bus_sum(pv, (gcup)recs_, (gcfp)hbm_ + (size_t)b * arenaFloats, (gfp)(outRing + (size_t)b * numOut * stride), n, numOut, stride);
    if (b == 0 && threadIdx.x == 0) { g->sampleTime += (int64_t)n * (int64_t)batch; g->blockSlot = 0; }
}

__global__ __launch_bounds__(1024)
void elemhip_epilogue_kernel(PlanView pv, uint32_t* recs_, const float* hbm_, Globals* g, float* outRing) {
    gup recs = (gup)recs_; gcfp hbm = (gcfp)hbm_;
    const uint32_t n = g->numSamples, numOut = min(g->numOut, (uint32_t)kMaxOut), stride = g->blockStride;
    gfp out = (gfp)(outRing + (size_t)g->blockSlot * numOut * stride);
    bus_sum(pv, recs, hbm, out, n, numOut, stride);
    for (uint32_t k = 0; k < pv.numTaps; ++k) {
        const TapEntry te = pv.taps[k];
        gcup rr = recs + te.rootRec * kRecDwords;
        // promotion needs root.active() (GraphRenderSequence.h:200-210)
        if (!(u2f(rr[rec::ROOT_TARGET]) > 0.5f)) continue;
        gcup tr = recs + te.rec * kRecDwords;
        gfp shared = rec_ptr(tr, rec::TAP_SHARED);
        gcfp priv = rec_ptr(tr, rec::TAP_PRIVATE);
        if (!shared || !priv) continue;
        __syncthreads();   // earlier promotions into the same name complete first (last writer wins)
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) shared[i] = priv[i];
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < pv.numRoots; r += blockDim.x) {
        gup rr = recs + pv.roots[r].rec * kRecDwords;
        if (!root_running(recs, pv.roots[r].rec, numOut)) continue;
        const float gcur = u2f(rr[rec::ROOT_GAIN]), tg = u2f(rr[rec::ROOT_TARGET]), step = u2f(rr[rec::ROOT_STEP]);
        if (gcur != tg && (rr[rec::ROOT_HASIN] || g->numIn > 0) /* fade.process ran (Core.h:74-77) */)
            rr[rec::ROOT_GAIN] = f2u(clampf(gcur + step * (float)(int)n, 0.0f, 1.0f));
    }
    // convolve nodes: commit the input-block position their main workgroups reached (conv.hip)
    for (uint32_t k = threadIdx.x; k < pv.numConvs; k += blockDim.x) {
        gup st = (gup)rec_ptr(recs + pv.convs[k].rec * kRecDwords, rec::CONV_STATE);
        if (!st) continue;
        st[conv::H_FILL] = st[conv::H_FILL_NEXT];
        st[conv::H_BLK] = st[conv::H_BLK_NEXT];
    }
    const uint32_t nextSlot = (g->blockSlot + 1) % g->ringSlots;
    if (g->inRing && nextSlot < g->inBlocks) {   // stage the next block's host inputs (elemhip_process_blocks)
        const uint32_t words = g->numIn * stride;
        gcfp src = (gcfp)reinterpret_cast<const float*>(g->inRing) + (size_t)nextSlot * words;
        gfp dst = (gfp)const_cast<float*>(hbm_);
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        g->sampleTime += (int64_t)n;
        g->blockSlot = nextSlot;
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
                  uint32_t levelBegin, uint32_t numIslands, uint32_t ldsBytes, uint32_t batch, uint32_t arenaFloats) {
    // gridDim.y: blocks of a batch that stateless islands render concurrently
    const uint32_t gy = batch > 1u ? (batch < 8u ? batch : 8u) : 1u;
    hipLaunchKernelGGL(elemhip_island_kernel, dim3(numIslands, gy), dim3(kThreads), ldsBytes, s, pv, recs, hbm, g, lcg, levelBegin, batch, arenaFloats);
}

void launch_epilogue(hipStream_t s, const PlanView& pv, uint32_t* recs, const float* hbm, Globals* g, float* outRing) {
    hipLaunchKernelGGL(elemhip_epilogue_kernel, dim3(1), dim3(1024), 0, s, pv, recs, hbm, g, outRing);
}

void launch_epilogue_batch(hipStream_t s, const PlanView& pv, uint32_t* recs, const float* hbm, Globals* g, float* outRing,
                           uint32_t batch, uint32_t arenaFloats) {
    hipLaunchKernelGGL(elemhip_epilogue_batch_kernel, dim3(batch), dim3(1024), 0, s, pv, recs, hbm, g, outRing, batch, arenaFloats);
}

void launch_patches(hipStream_t s, const Patch* patches, uint32_t count, uint32_t* recs, uint32_t* globals) {
    if (!count) return;
    hipLaunchKernelGGL(elemhip_patch_kernel, dim3((count + 255) / 256), dim3(256), 0, s, patches, count, recs, globals);
}

} // namespace elemhip
