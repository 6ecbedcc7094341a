// engine.cpp — instruction interpreter, node table, device-resident node records and the
// per-block launch sequence. See engine.h for the mapping onto runtime/elem/Runtime.h.
#include "engine.h"
#include <thread>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "launch.h"

namespace elemhip {

// ELEMHIP_DEBUG_SYNC=1 (fault hunting on the GPU box): every launch is followed by a device synchronise and a line on stderr, so
// the last line printed before a "Memory access fault" abort names the kernel that faulted.
static bool debugSyncOn() { static const bool on = std::getenv("ELEMHIP_DEBUG_SYNC") != nullptr; return on; }
static void debugSync(const char* what, unsigned a = 0, unsigned b = 0) {
    if (!debugSyncOn()) return;
    std::fprintf(stderr, "[elemhip sync] %s %u %u ...", what, a, b); std::fflush(stderr);
    const hipError_t e = hipDeviceSynchronize();
    std::fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); std::fflush(stderr);
}

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    std::fprintf(stderr, "[elemhip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return kHipError; } } while (0)
#define HIP_WARN(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    std::fprintf(stderr, "[elemhip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)

const char* describe(int c) {
    switch (c) {   // 0..8: runtime/elem/Types.h:62-85
        case 0: return "Ok";
        case 1: return "Node type not recognized";
        case 2: return "Node not found";
        case 3: return "Attempting to create a node that already exists";
        case 4: return "Attempting to create a node type that already exists";
        case 5: return "Invalid value type for the given node property";
        case 6: return "Invalid value for the given node property";
        case 7: return "Invariant violation";
        case 8: return "Invalid instruction format";
        case kHipError: return "HIP runtime error";
        case kNoDevice: return "No HIP device available";
        case kBlockTooLarge: return "numSamples exceeds the block size the runtime was created with";
        case kTooManyChannels: return "Too many host channels";
        case kUnsupportedGraph: return "Graph uses a construct the HIP engine does not support";
        case kJsonParseError: return "Failed to parse json string";
        default: return "Return code not recognized";
    }
}

// Registry: reference node-type names (DefaultNodeTypes.h:49-144, wasm/Main.cpp:47-61) -> opcode
static const std::unordered_map<std::string, uint16_t>& opTable() {
    static const std::unordered_map<std::string, uint16_t> t = {
        {"in", OP_IN}, {"sin", OP_SIN}, {"cos", OP_COS}, {"tan", OP_TAN}, {"tanh", OP_TANH}, {"asinh", OP_ASINH},
        {"ln", OP_LN}, {"log", OP_LOG}, {"log2", OP_LOG2}, {"ceil", OP_CEIL}, {"floor", OP_FLOOR}, {"round", OP_ROUND},
        {"sqrt", OP_SQRT}, {"exp", OP_EXP}, {"abs", OP_ABS},
        {"le", OP_LE}, {"leq", OP_LEQ}, {"ge", OP_GE}, {"geq", OP_GEQ}, {"pow", OP_POW}, {"eq", OP_EQ}, {"and", OP_AND}, {"or", OP_OR},
        {"add", OP_ADD}, {"sub", OP_SUB}, {"mul", OP_MUL}, {"div", OP_DIV}, {"mod", OP_MOD}, {"min", OP_MIN}, {"max", OP_MAX},
        {"root", OP_ROOT}, {"const", OP_CONST}, {"phasor", OP_PHASOR}, {"sphasor", OP_SPHASOR}, {"sr", OP_SR}, {"seq", OP_SEQ},
        {"counter", OP_COUNTER}, {"accum", OP_ACCUM}, {"latch", OP_LATCH}, {"maxhold", OP_MAXHOLD}, {"once", OP_ONCE}, {"rand", OP_RAND},
        {"delay", OP_DELAY}, {"sdelay", OP_SDELAY}, {"z", OP_Z},
        {"pole", OP_POLE}, {"env", OP_ENV}, {"biquad", OP_BIQUAD}, {"prewarp", OP_PREWARP}, {"mm1p", OP_MM1P}, {"svf", OP_SVF}, {"svfshelf", OP_SVFSHELF},
        {"tapIn", OP_TAPIN}, {"tapOut", OP_TAPOUT},
        {"blepsaw", OP_BLEPSAW}, {"blepsquare", OP_BLEPSQUARE}, {"bleptriangle", OP_BLEPTRIANGLE},
        {"mc.table", OP_TABLE}, {"mc.sample", OP_MCSAMPLE}, {"mc.sampleseq", OP_SAMPLESEQ}, {"time", OP_TIME}, {"metro", OP_METRO}, {"sampleseq", OP_SAMPLESEQ}, {"convolve", OP_CONVOLVE}, {"table", OP_TABLE}, {"seq2", OP_SEQ2}, {"sparseq2", OP_SPARSEQ2}, {"sparseq", OP_SPARSEQ}, {"capture", OP_CAPTURE}, {"mc.capture", OP_CAPTURE}, {"sample", OP_SAMPLE}, {"meter", OP_METER}, {"snapshot", OP_SNAPSHOT}, {"scope", OP_SCOPE},
    };
    return t;
}

static inline uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

static int bitceil(int n) {   // builtins/helpers/BitUtils.h:9-20
    if ((n & (n - 1)) == 0) return n;
    int o = 1;
    while (o < n) o <<= 1;
    return o;
}

static inline float clampf(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }

static double msToStep(double sr, double ms) {   // helpers/GainFade.h:10-12
    return ms > 1e-6 ? 1.0 / (sr * ms / 1000.0) : 1.0;
}

ProgHeap::~ProgHeap() { if (dev) (void)hipFree(dev); }

Plan::~Plan() {
    if (graphExec) (void)hipGraphExecDestroy(graphExec);
    if (specGraphExec) (void)hipGraphExecDestroy(specGraphExec);
    if (dev.ptr) { if (pool) pool->give(dev); else (void)hipFree(dev.ptr); }
}

DevBuf TablePool::take(size_t bytes) {
    {
        std::lock_guard<std::mutex> lock(m);
        size_t best = free.size();
        for (size_t k = 0; k < free.size(); ++k)
            if (free[k].bytes >= bytes && free[k].bytes <= 4 * bytes && (best == free.size() || free[k].bytes < free[best].bytes)) best = k;
        if (best != free.size()) { DevBuf b = free[best]; free.erase(free.begin() + (long)best); return b; }
    }
    DevBuf b;
    b.bytes = bytes + bytes / 4 + 4096;    // (the next plan of a live graph is a little bigger or smaller)
    if (hipMalloc(&b.ptr, b.bytes) != hipSuccess) { b.ptr = nullptr; b.bytes = 0; }
    return b;
}

void TablePool::give(DevBuf b) {
    std::lock_guard<std::mutex> lock(m);
    free.push_back(b);
    if (free.size() > 8) { (void)hipFree(free.front().ptr); free.erase(free.begin()); }
}

TablePool::~TablePool() { for (DevBuf& b : free) (void)hipFree(b.ptr); }

// ---------------------------------------------------------------------------------------------
Engine::Engine(double sr, int bs, int dev) : sampleRate(sr), blockSize(bs), device(dev) {
    auto fail = [&](int code) { initErr = code; };
    // A block's buffers are LDS slots of at most kMaxBlock frames. A longer host block is rendered as k equal slices — the smallest k
    // that divides it into slices of 64 .. kMaxBlock frames (1024 -> 2 x 512, 700 -> 2 x 350, 1023 -> 3 x 341; r04 took multiples of
    // 512 only) — the sample-rate nodes cannot tell; taps, whose delay IS the host's block (Feedback.h:29-31, 66-67, 103-104), keep
    // shared buffers of the HOST's block size and every slice reads / writes its own stretch of them (setTapSlice): the engine's own
    // block size is the slice, the host's the limit of a process() call. A size no such k divides (a prime above 512: 521, 1031) is
    // rendered as slices of kMaxBlock frames and a shorter last one — to the kernels the same as a host that calls process() with
    // fewer frames than the block size (r04 / early r05 refused such sizes).
    hostBlockSize = bs;
    if (bs > (int)kMaxBlock && bs <= 64 * (int)kMaxBlock) {
        bool equal = false;
        for (int k = (bs + (int)kMaxBlock - 1) / (int)kMaxBlock; k <= bs / 64; ++k)
            if (bs % k == 0) { bs /= k; blockSize = bs; equal = true; break; }
        if (!equal) { bs = (int)kMaxBlock; blockSize = bs; }
    }
    if (const char* e = std::getenv("ELEMHIP_SPECIALIZE")) specialize = std::max(0, std::min(2, std::atoi(e)));
    if (const char* e = std::getenv("ELEMHIP_SYNC_POLL")) syncPoll = std::atoi(e) != 0;
    if (const char* e = std::getenv("ELEMHIP_FUSE_EPILOGUE")) fuseEpilogue = std::atoi(e) != 0;   // (a native host without access to the options)
    if (const char* e = std::getenv("ELEMHIP_RESIDENT")) residentOpt = std::atoi(e) != 0;   // (a native host without access to the options)
    if (const char* e = std::getenv("ELEMHIP_PLAN_CACHE")) planCache = std::max(0, std::min(2, std::atoi(e)));   // 2: verify mode (tests)
    if (bs <= 0 || bs > (int)kMaxBlock) { fail(kBlockTooLarge); return; }
    if (dev == -1) {
        // "dry" engine: host logic only (instruction decode, graph mutation, plan build, gc) with
        // no device behind it. It cannot render: process() returns kNoDevice. Used by CPU-only tests.
        dry = true;
        hGlobals = Globals{};
        hGlobals.numSamples = (uint32_t)bs; hGlobals.ringSlots = 1; hGlobals.blockStride = (uint32_t)bs;
        hGlobals.sampleRateF = (float)sr; hGlobals.sampleRate = sr;
        return;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { fail(kNoDevice); return; }
    if (dev < 0 || dev >= count) { fail(kNoDevice); return; }
    if (hipSetDevice(dev) != hipSuccess) { fail(kHipError); return; }
    if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { fail(kHipError); return; }
    ownStream = true;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) cuCount = cus; }

    recCapacity = 8192;
    if (hipMalloc(&dRecs, (size_t)recCapacity * kRecDwords * 4) != hipSuccess) { fail(kHipError); return; }
    (void)hipMemsetAsync(dRecs, 0, (size_t)recCapacity * kRecDwords * 4, stream);
    (void)hipStreamSynchronize(stream);
    shadow.reserve((size_t)recCapacity * kRecDwords);

    patchCap = 1u << 16;
    if (hipHostMalloc((void**)&hPatches, sizeof(Patch) * patchCap, hipHostMallocDefault) != hipSuccess) { fail(kHipError); return; }

    hGlobals = Globals{};
    hGlobals.sampleTime = 0;
    hGlobals.numSamples = (uint32_t)bs;
    hGlobals.ringSlots = 1;
    hGlobals.blockStride = (uint32_t)bs;
    hGlobals.sampleRateF = (float)sr;
    hGlobals.sampleRate = sr;
    if (hipMalloc(&dGlobals, sizeof(Globals)) != hipSuccess) { fail(kHipError); return; }
    (void)hipMemcpy(dGlobals, &hGlobals, sizeof(Globals), hipMemcpyHostToDevice);

    {   // FFT twiddles for `convolve` (conv.hip): cis(-2 pi k / 1024) rounded from double
        std::vector<float> tw(2 * conv::kFft);
        for (uint32_t k = 0; k < conv::kFft; ++k) {
            const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)conv::kFft;
            tw[2 * k] = (float)std::cos(a); tw[2 * k + 1] = (float)std::sin(a);
        }
        std::vector<float> tw8(2 * 8192);     // ... and cis(-2 pi j / 8192) for the long-partition transforms (conv_long.inc, fft4096.h)
        for (uint32_t k = 0; k < 8192; ++k) {
            const double a = -2.0 * 3.14159265358979323846 * (double)k / 8192.0;
            tw8[2 * k] = (float)std::cos(a); tw8[2 * k + 1] = (float)std::sin(a);
        }
        if (upload_convolve_tables(tw.data(), tw8.data()) != hipSuccess) { fail(kHipError); return; }
    }
    // LCG jump-ahead table for `rand` (Noise.h:28-32): s_k = A[k]*s_0 + C[k] (mod 2^32)
    std::vector<uint32_t> lcg(2 * (kMaxBlock + 1));
    uint32_t A = 1, Cc = 0;
    for (uint32_t k = 0; k <= kMaxBlock; ++k) {
        lcg[2 * k] = A; lcg[2 * k + 1] = Cc;
        A = 214013u * A; Cc = 214013u * Cc + 2531011u;
    }
    if (hipMalloc(&dLcg, lcg.size() * 4) != hipSuccess) { fail(kHipError); return; }
    (void)hipMemcpy(dLcg, lcg.data(), lcg.size() * 4, hipMemcpyHostToDevice);

    if (ensureHbm(kMaxHostIn + 64) != kOk) { fail(kHipError); return; }
    if (ensureOutRing((size_t)8 * bs) != kOk) { fail(kHipError); return; }
    if (const char* e = std::getenv("ELEMHIP_NO_GRAPH")) useGraph = !(e[0] == '1');
}

Engine::~Engine() {
    for (auto& kv : nodes) if (kv.second.hostInst && kv.second.hostVt && kv.second.hostVt->destroy) kv.second.hostVt->destroy(kv.second.hostInst, kv.second.hostVt->user);
    if (dry) {
        for (auto& kv : nodes) std::free(kv.second.ring.ptr);
        for (auto& kv : resources) { std::free(kv.second->dev.ptr); for (DevBuf& d : kv.second->devCh) std::free(d.ptr); }
        return;
    }
    residentStop();
    if (stream) (void)hipStreamSynchronize(stream);
    if (hResident) (void)hipHostFree(hResident);
    if (hDone) (void)hipHostFree(hDone);
    if (hResIn) (void)hipHostFree(hResIn);
    if (hResOut) (void)hipHostFree(hResOut);
    if (dResidentSync) (void)hipFree(dResidentSync);
    current.reset(); pending.reset();
    for (auto& kv : nodes) if (kv.second.ring.ptr) (void)hipFree(kv.second.ring.ptr);
    for (auto& kv : resources) { if (kv.second->dev.ptr) (void)hipFree(kv.second->dev.ptr); for (DevBuf& d : kv.second->devCh) if (d.ptr) (void)hipFree(d.ptr); }
    freeDeferred();
    if (dRecs) (void)hipFree(dRecs);
    if (dGlobals) (void)hipFree(dGlobals);
    if (dLcg) (void)hipFree(dLcg);
    if (dConvScratch) (void)hipFree(dConvScratch);
    if (dHbm) (void)hipFree(dHbm);
    if (dOutRing) (void)hipFree(dOutRing);
    for (hipEvent_t e : profEvents) (void)hipEventDestroy(e);
    for (hipEvent_t e : auxDone) (void)hipEventDestroy(e);
    for (hipStream_t s2 : auxStreams) (void)hipStreamDestroy(s2);
    if (forkEvent) (void)hipEventDestroy(forkEvent);
    if (hPatches) (void)hipHostFree(hPatches);
    if (hOut) (void)hipHostFree(hOut);
    if (hIn) (void)hipHostFree(hIn);
    if (ioStream) (void)hipStreamSynchronize(ioStream);
    for (int k = 0; k < 2; ++k) {
        if (hStageOut[k]) (void)hipHostFree(hStageOut[k]);
        if (hStageIn[k]) (void)hipHostFree(hStageIn[k]);
        if (dStageOut[k]) (void)hipFree(dStageOut[k]);
        if (dStageIn[k]) (void)hipFree(dStageIn[k]);
        if (evIn[k]) (void)hipEventDestroy(evIn[k]);
        if (evRendered[k]) (void)hipEventDestroy(evRendered[k]);
        if (evOut[k]) (void)hipEventDestroy(evOut[k]);
    }
    if (ioStream) (void)hipStreamDestroy(ioStream);
    if (relayStream) { (void)hipStreamSynchronize(relayStream); (void)hipStreamDestroy(relayStream); }
    if (evRelay) (void)hipEventDestroy(evRelay);
    if (dRelay) (void)hipFree(dRelay);
    if (hRelay) (void)hipHostFree(hRelay);
    if (ownStream && stream) (void)hipStreamDestroy(stream);
}

// Every entry point that takes the render lock — except the per-block process() calls themselves — first asks a resident kernel
// (option "resident") to leave: it owns the engine's stream for as long as it lives.
struct RenderGuard {
    std::lock_guard<std::mutex> l;
    explicit RenderGuard(Engine& e) : l(e.mu) { e.residentStop(); }
};

void Engine::setStream(hipStream_t s) {
    RenderGuard lock(*this);
    if (dry) return;
    if (stream) (void)hipStreamSynchronize(stream);
    if (ownStream && stream) (void)hipStreamDestroy(stream);
    stream = s; ownStream = false;
    dropGraphs();
}

void Engine::dropGraphs() {
    if (!current) return;
    if (current->graphExec) { (void)hipGraphExecDestroy(current->graphExec); current->graphExec = nullptr; }
    if (current->specGraphExec) { (void)hipGraphExecDestroy(current->specGraphExec); current->specGraphExec = nullptr; }
}

void Engine::freeDeferred() {   // called right after a synchronize of every stream the engine renders on (`mu` held)
    patchCursor = 0;
    for (void* p : deferredFree) (void)hipFree(p);
    deferredFree.clear();
    retiredPlans.clear();         // plans replaced since the last synchronize: their tables / hipGraphs are idle now (~Plan frees them)
}

int Engine::ensureHbm(size_t buffers) {
    if (buffers <= hbmBuffers) return kOk;
    size_t want = std::max(buffers, hbmBuffers * 2);
    float* nb = nullptr;
    // + slack: vector loads of a 64*V-frame task may read (never write) past a short block's buffer
    const size_t floats = want * blockSize + 1024;
    HIP_OK(hipMalloc(&nb, floats * sizeof(float)));
    HIP_OK(hipMemsetAsync(nb, 0, floats * sizeof(float), stream));   // ordered with the kernels: they run on `stream` (non-blocking w.r.t. the null stream)
    if (dHbm) deferredFree.push_back(dHbm);
    dHbm = nb; hbmBuffers = want;
    dropGraphs();
    return kOk;
}

// Arena buffers a launch set of `blocks` blocks of plan `p` needs: one slice (host inputs + exports) per block, then the
// stream ring of the specialised kernels, one slice per buffer set an island may keep in flight.
size_t Engine::arenaBuffers(const Plan& p, size_t blocks) const {
    return (size_t)p.numHbmBuffers * blocks + (size_t)p.numStreamBuffers * (streamRing ? (size_t)p.maxCopies : blocks);
}
// The recurrence loops of the specialised kernels address the whole arena through 32-bit buffer offsets with the top bit
// reserved as "out of range": a launch set stays under 2 GB of arena (C2: 0.6 MB per block).
size_t Engine::maxSetBlocks(const Plan& p) const {
    const size_t cap = ((size_t)1 << 29) / (size_t)blockSize;                       // buffers of blockSize floats in 2 GB
    if (!streamRing) return std::max<size_t>(1, (cap - 2) / std::max<uint32_t>(1u, p.numHbmBuffers + p.numStreamBuffers));
    const size_t ring = (size_t)p.numStreamBuffers * p.maxCopies + 2;
    return cap > ring ? std::max<size_t>(1, (cap - ring) / std::max<uint32_t>(1u, p.numHbmBuffers)) : 1;
}

int Engine::ensureOutRing(size_t floats) {
    if (floats <= outRingFloats) return kOk;
    float* nb = nullptr;
    HIP_OK(hipMalloc(&nb, floats * sizeof(float)));
    HIP_OK(hipMemsetAsync(nb, 0, floats * sizeof(float), stream));
    if (dOutRing) deferredFree.push_back(dOutRing);
    dOutRing = nb; outRingFloats = floats;
    dropGraphs();
    return kOk;
}

uint32_t Engine::allocRec() {
    uint32_t r;
    if (!freeRecs.empty()) { r = freeRecs.back(); freeRecs.pop_back(); }
    else r = nextRec++;
    if (shadow.size() < (size_t)(r + 1) * kRecDwords) shadow.resize((size_t)(r + 1) * kRecDwords, 0u);
    std::fill(shadow.begin() + (size_t)r * kRecDwords, shadow.begin() + (size_t)(r + 1) * kRecDwords, 0u);
    if (freshFlag.size() <= r) freshFlag.resize((size_t)r + 1, 0);
    freshFlag[r] = 1;
    freshRecs.push_back(r);
    return r;
}

void Engine::writeParam(Node& n, uint32_t dword, uint32_t value) {
    const uint32_t idx = n.rec * kRecDwords + dword;
    shadow[idx] = value;
    // a record that has not been uploaded yet travels whole; otherwise patch the one dword
    if (!freshFlag[n.rec])
        patches.push_back(Patch{0u, idx, value, 0u});
    // a multi-output node: every channel's record — except the buffer slots (P0..P2: pointer, length), which differ per
    // channel and are written by writeChannelBuffer only (two patches for one dword in one flush have no order)
    if (dword > rec::P2) for (uint32_t cr : n.chanRecs) writeRec(cr, dword, value);
}

void Engine::writeRec(uint32_t rec, uint32_t dword, uint32_t value) {
    const uint32_t idx = rec * kRecDwords + dword;
    shadow[idx] = value;
    if (!freshFlag[rec]) patches.push_back(Patch{0u, idx, value, 0u});
}

// Channel `ch` of a shared resource on the device (channel 0 is Resource::dev). A channel the resource does not have
// reads as an empty buffer (AudioBufferResource.h:32-40).
int Engine::ensureResourceChannelOnDevice(const ResourcePtr& r, uint32_t ch, const void** ptr, uint32_t* len) {
    *ptr = nullptr; *len = 0;
    if (!r || ch >= r->channels.size()) return kOk;
    *len = (uint32_t)r->channels[ch].size();
    if (ch == 0) { int rc = ensureResourceOnDevice(r); *ptr = r->dev.ptr; return rc; }
    if (r->devCh.size() <= ch) r->devCh.resize(ch + 1);
    DevBuf& d = r->devCh[ch];
    if (!d.ptr) {
        const size_t floats = std::max<size_t>(r->channels[ch].size(), (size_t)blockSize);
        if (dry) { d.ptr = std::calloc(floats, sizeof(float)); std::memcpy(d.ptr, r->channels[ch].data(), r->channels[ch].size() * 4); }
        else {
            HIP_OK(hipMalloc(&d.ptr, floats * sizeof(float)));
            HIP_OK(hipMemsetAsync(d.ptr, 0, floats * sizeof(float), stream));
            HIP_OK(hipStreamSynchronize(stream));
            if (*len) HIP_OK(hipMemcpy(d.ptr, r->channels[ch].data(), (size_t)*len * sizeof(float), hipMemcpyHostToDevice));
        }
        d.bytes = floats * sizeof(float);
    }
    *ptr = d.ptr;
    return kOk;
}

// mc.table (mc/Table.h:12-85): output channel j is TableNode's lookup (Table.h:35-71) into channel j of the resource
void Engine::writeTableChannel(Node& n, uint32_t ch, uint32_t rec) {
    const void* ptr = nullptr; uint32_t len = 0;
    if (n.res) (void)ensureResourceChannelOnDevice(n.res, ch, &ptr, &len);
    const uint64_t v = (uint64_t)reinterpret_cast<uintptr_t>(ptr);
    writeRec(rec, rec::TBL_BUF, (uint32_t)(v & 0xFFFFFFFFu));
    writeRec(rec, rec::TBL_BUF + 1, (uint32_t)(v >> 32));
    writeRec(rec, rec::TBL_LEN, len);
}

// The record of output channel `ch` of a multi-output node. Every channel renders the node's algorithm on its own copy of
// the state (the reference keeps ONE state and loops over the channels inside process(): reader positions and fades
// evolve identically for every channel) and differs only in the resource channel it reads.
uint32_t Engine::channelRec(Node& n, uint32_t ch) {
    if (ch == 0 || !n.mc) return n.rec;
    while (n.chanRecs.size() < ch) {
        const uint32_t r = allocRec();
        n.chanRecs.push_back(r);
        // parameters and INITIAL state as the host last wrote them for channel 0 (pending-buffer flags included)
        std::memcpy(shadow.data() + (size_t)r * kRecDwords, shadow.data() + (size_t)n.rec * kRecDwords, kRecDwords * 4);
        if (n.op == OP_CAPTURE) shadow[(size_t)r * kRecDwords + rec::CAP_CH] = (uint32_t)n.chanRecs.size();   // mc.capture: passes input ch + 1 through (channel 0 records)
        else writeChannelBuffer(n, (uint32_t)n.chanRecs.size(), r);
        // channel 0 has been on the device already (it may be mid-playback): the new channel continues from channel 0's
        // LIVE reader state and consumed flags — the reference keeps one state for all channels (mc/Sample.h, mc/SampleSeq.h)
        // (mc.capture channels > 0 only pass an input through: they have no state to take over, and the clone — dwords P3.. of
        //  channel 0's LIVE record — would overwrite the CAP_CH just set with channel 0's: the new channel would pass input 1 through
        //  and record every block into the shared ring a second time)
        if (!freshFlag[n.rec] && n.op != OP_CAPTURE) recClones.push_back({n.rec, r});
    }
    return n.chanRecs[ch - 1];
}

// buffer pointer / length of resource channel `ch` into record `rec` (table, mc.sample, mc.sampleseq share the slots P0-P2)
void Engine::writeChannelBuffer(Node& n, uint32_t ch, uint32_t rec) {
    static_assert(rec::TBL_BUF == rec::SMP_BUF && rec::TBL_BUF == rec::SSQ_BUF && rec::TBL_LEN == rec::SMP_LEN && rec::TBL_LEN == rec::SSQ_BUFLEN, "shared buffer slots");
    writeTableChannel(n, ch, rec);
}

void Engine::writeParamPtr(Node& n, uint32_t dword, const void* p) {
    const uint64_t v = (uint64_t)reinterpret_cast<uintptr_t>(p);
    writeParam(n, dword, (uint32_t)(v & 0xFFFFFFFFu));
    writeParam(n, dword + 1, (uint32_t)(v >> 32));
}

int Engine::allocRing(Node& n, size_t floats) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(floats, 1) * sizeof(float);
    if (dry) {
        p = std::calloc(1, bytes);
        std::free(n.ring.ptr);
        n.ring.ptr = p; n.ring.bytes = bytes;
        return kOk;
    }
    HIP_OK(hipMalloc(&p, bytes));
    HIP_OK(hipMemsetAsync(p, 0, bytes, stream));
    HIP_OK(hipStreamSynchronize(stream));   // callers follow up with blocking hipMemcpy's into the buffer
    if (n.ring.ptr) deferredFree.push_back(n.ring.ptr);
    n.ring.ptr = p; n.ring.bytes = bytes;
    return kOk;
}

// A new impulse response = a new convolver starting from silence (Convolve.h:47-51: a fresh
// TwoStageFFTConvolver per `path` assignment). Builds the conv:: state: header + IR partition spectra.
int Engine::setConvolverIr(Node& n, const ResourcePtr& res) {
    std::vector<float> h;
    if (!res->channels.empty()) h = res->channels[0];
    // trailing |h| < 1e-6 is dropped by the two-stage convolver as a whole and again by each of its
    // three uniform convolvers over ir[0:4096), ir[4096:8192), ir[8192:) (fftconv_oracle.h)
    size_t len = h.size();
    while (len > 0 && std::fabs(h[len - 1]) < 0.000001f) --len;
    h.resize(len);
    for (size_t lo : {(size_t)0, (size_t)4096}) {
        size_t hi = std::min(len, lo + 4096);
        while (hi > lo && std::fabs(h[hi - 1]) < 0.000001f) h[--hi] = 0.0f;
    }
    const uint32_t B = conv::kBlock, N = conv::kFft;
    const uint32_t P = (uint32_t)((len + B - 1) / B);
    const uint32_t older = P > 2 ? P - 2 : 0;
    const uint32_t S = std::min<uint32_t>(conv::kMaxSlices, std::max<uint32_t>(1, (older + conv::kSlicePartitions - 1) / conv::kSlicePartitions));
    // Long partitions (conv_long.inc): launch sets of a multiple of 8 blocks evaluate an IR of at least kLongMinP 512-partitions as
    // Q partitions of 4096 samples (8192-point spectra, G rows padded with zero rows to a multiple of the MAC's tap group) from a
    // ring of the node's last R input blocks in the time domain; both live behind `overlap`.
    constexpr uint32_t kLongMinP = 32;
    const uint32_t tapGroup = convolve_long_tap_group(), rowFloats = convolve_long_row_floats();
    const uint32_t Q = (convLong && P >= kLongMinP) ? (uint32_t)((len + 4095) / 4096) : 0u;
    const uint32_t Qp = (Q + tapGroup - 1) / tapGroup * tapGroup;
    const uint32_t R = Q ? 8u * Qp + 8u : 0u;
    const size_t words512 = conv::kHeaderDwords + 2 * ((size_t)2 * P + 2 * S + 1) * 512 + 1024;
    const size_t words = words512 + (size_t)R * 512 + (size_t)Qp * rowFloats;
    std::vector<uint32_t> blob(conv::kHeaderDwords + (size_t)P * 1024, 0u);
    blob[conv::H_P] = P; blob[conv::H_S] = S; blob[conv::H_Q] = Q; blob[conv::H_HISTBLKS] = R;
    if (++convUid == 0u) convUid = 1u;
    blob[conv::H_UID] = convUid;
    n.convQp = Qp; n.convHistBlocks = R; n.convP = P;
    convMaxQp = std::max(convMaxQp, Qp);
    convMinP = std::min(convMinP, P); convMaxP = std::max(convMaxP, P);   // (over the engine's lifetime: which MAC kernels a launch set needs)
    // IR partition spectra in double, scaled by 1/1024 (exact), rounded to float, Nyquist packed into bin 0
    std::vector<std::complex<double>> a(N), tw(N / 2);
    for (uint32_t k = 0; k < N / 2; ++k) { const double ang = -2.0 * 3.14159265358979323846 * k / N; tw[k] = {std::cos(ang), std::sin(ang)}; }
    for (uint32_t p = 0; p < P; ++p) {
        for (uint32_t i = 0; i < N; ++i) { const size_t j = (size_t)p * B + i; a[i] = (i < B && j < len) ? (double)h[j] : 0.0; }
        for (uint32_t i = 1, j = 0; i < N; ++i) {            // bit reversal
            uint32_t bit = N >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(a[i], a[j]);
        }
        for (uint32_t m = 2; m <= N; m <<= 1)
            for (uint32_t s0 = 0; s0 < N; s0 += m)
                for (uint32_t k = 0; k < m / 2; ++k) {
                    const std::complex<double> u = a[s0 + k], t = a[s0 + k + m / 2] * tw[k * (N / m)];
                    a[s0 + k] = u + t; a[s0 + k + m / 2] = u - t;
                }
        float* dst = reinterpret_cast<float*>(blob.data() + conv::kHeaderDwords + (size_t)p * 1024);
        const double sc = 1.0 / (double)N;
        dst[0] = (float)(a[0].real() * sc); dst[1] = (float)(a[N / 2].real() * sc);
        for (uint32_t k = 1; k < N / 2; ++k) { dst[2 * k] = (float)(a[k].real() * sc); dst[2 * k + 1] = (float)(a[k].imag() * sc); }
    }
    // G_q = RFFT_8192([g_q | 0]) / 16384 in double, rounded to float: the device transforms return 16384 x the circular convolution (fft4096.h)
    std::vector<float> G((size_t)Qp * rowFloats, 0.0f);
    if (Q) {
        const uint32_t N8 = 8192;
        std::vector<std::complex<double>> a8(N8), tw8(N8 / 2);
        for (uint32_t k = 0; k < N8 / 2; ++k) { const double ang = -2.0 * 3.14159265358979323846 * k / N8; tw8[k] = {std::cos(ang), std::sin(ang)}; }
        for (uint32_t q = 0; q < Q; ++q) {
            for (uint32_t i = 0; i < N8; ++i) { const size_t j = (size_t)q * 4096 + i; a8[i] = (i < 4096 && j < len) ? (double)h[j] : 0.0; }
            for (uint32_t i = 1, j = 0; i < N8; ++i) {            // bit reversal
                uint32_t bit = N8 >> 1;
                for (; j & bit; bit >>= 1) j ^= bit;
                j ^= bit;
                if (i < j) std::swap(a8[i], a8[j]);
            }
            for (uint32_t m = 2; m <= N8; m <<= 1)
                for (uint32_t s0 = 0; s0 < N8; s0 += m)
                    for (uint32_t k = 0; k < m / 2; ++k) {
                        const std::complex<double> u = a8[s0 + k], t = a8[s0 + k + m / 2] * tw8[k * (N8 / m)];
                        a8[s0 + k] = u + t; a8[s0 + k + m / 2] = u - t;
                    }
            float* dst = G.data() + (size_t)q * rowFloats;
            for (uint32_t k = 0; k <= N8 / 2; ++k) { dst[2 * k] = (float)(a8[k].real() / 16384.0); dst[2 * k + 1] = (float)(a8[k].imag() / 16384.0); }
        }
    }
    int rc = allocRing(n, words);
    if (rc != kOk) return rc;
    const size_t gOff = (words512 + (size_t)R * 512) * 4;       // bytes: header + 512-partition state + input ring
    if (dry) { std::memcpy(n.ring.ptr, blob.data(), blob.size() * 4); if (Q) std::memcpy((char*)n.ring.ptr + gOff, G.data(), G.size() * 4); }
    else {
        HIP_OK(hipMemcpy(n.ring.ptr, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
        if (Q) HIP_OK(hipMemcpy((char*)n.ring.ptr + gOff, G.data(), G.size() * 4, hipMemcpyHostToDevice));
    }
    writeParamPtr(n, rec::CONV_STATE, n.ring.ptr);
    if (n.convSlices != S) { n.convSlices = S; planStale = true; }
    return kOk;
}

int Engine::ensureResourceOnDevice(const ResourcePtr& r) {
    if (r->dev.ptr) return kOk;
    const size_t have = r->channels.empty() ? 0 : r->channels[0].size();
    const size_t floats = std::max<size_t>(have, (size_t)blockSize);
    void* p = nullptr;
    if (dry) { r->dev.ptr = std::calloc(floats, sizeof(float)); r->dev.bytes = floats * sizeof(float); return kOk; }
    HIP_OK(hipMalloc(&p, floats * sizeof(float)));
    HIP_OK(hipMemsetAsync(p, 0, floats * sizeof(float), stream));
    HIP_OK(hipStreamSynchronize(stream));
    if (have) HIP_OK(hipMemcpy(p, r->channels[0].data(), have * sizeof(float), hipMemcpyHostToDevice));
    r->dev.ptr = p; r->dev.bytes = floats * sizeof(float);
    return kOk;
}

// SharedResourceMap::getTapResource (SharedResource.h:79-92)
ResourcePtr Engine::tapResource(const std::string& name) {
    auto it = resources.find(name);
    if (it != resources.end()) return it->second;
    auto r = std::make_shared<Resource>();
    r->channels.emplace_back((size_t)hostBlockSize, 0.0f);      // AudioBufferResource(1, getBlockSize()) (Feedback.h:29-31): the HOST's block
    r->isTap = true;
    resources.emplace(name, r);
    return r;
}

// GainFade::updateCurrentStep (helpers/GainFade.h:107-109)
void Engine::rootUpdateStep(Node& n) {
    n.step = (n.gain > n.target) ? n.outStep : n.inStep;
    writeParamF(n, rec::ROOT_TARGET, n.target);
    writeParamF(n, rec::ROOT_STEP, n.step);
}

// ---- instructions ------------------------------------------------------------------------------
int Engine::createNode(int32_t id, const std::string& type) {   // Runtime.h:293-313
    auto ht = hostTypes.find(type);
    if (ht != hostTypes.end()) {               // a registered call-out type (Runtime.h:480-487)
        if (nodes.find(id) != nodes.end()) return kNodeAlreadyExists;
        Node n;
        n.id = id; n.op = OP_HOST; n.rec = allocRec();
        n.hostVt = ht->second.get();
        n.hostInst = n.hostVt->create ? n.hostVt->create(id, sampleRate, blockSize, n.hostVt->user) : nullptr;
        nodes.emplace(id, std::move(n));
        return kOk;
    }
    auto it = opTable().find(type);
    if (it == opTable().end()) return kUnknownNodeType;
    if (nodes.find(id) != nodes.end()) return kNodeAlreadyExists;
    Node n;
    n.id = id; n.op = it->second; n.rec = allocRec();
    n.mc = type.compare(0, 3, "mc.") == 0;
    uint32_t* r = shadow.data() + (size_t)n.rec * kRecDwords;
    switch (n.op) {
        case OP_CONST: r[rec::P0] = fbits(1.0f); break;                           // Core.h:166
        case OP_SR:    r[rec::P0] = fbits((float)sampleRate); break;              // Core.h:178
        case OP_IN:    r[rec::P0] = 0u; break;                                    // Math.h:125
        case OP_ROOT: {                                                           // Core.h:80-82
            n.gain = 0.0f; n.target = 1.0f; n.channel = -1;
            n.inStep = (float)msToStep(sampleRate, 20);
            n.outStep = (float)((double)(-1.0f) * msToStep(sampleRate, 20));
            n.step = (n.gain > n.target) ? n.outStep : n.inStep;
            r[rec::ROOT_CHANNEL] = (uint32_t)-1; r[rec::ROOT_TARGET] = fbits(n.target);
            r[rec::ROOT_STEP] = fbits(n.step); r[rec::ROOT_GAIN] = fbits(n.gain);
            break;
        }
        case OP_MAXHOLD: r[rec::P0] = 0xFFFFFFFFu; break;                         // Core.h:336
        case OP_SEQ:     r[rec::SEQ_HOLD] = 0; r[rec::SEQ_LOOP] = 1; break;       // Core.h:566-568
        case OP_SEQ2:    r[rec::SEQ_HOLD] = 0; r[rec::SEQ_LOOP] = 1; break;       // Seq2.h:157-159
        case OP_SPARSEQ:                                                          // SparSeq.h:340-368: edgeCount = -1, no loop points, no held event
            r[rec::SQ_EDGES] = (uint32_t)-1; r[rec::SQ_HOLD] = (uint32_t)-1;
            r[rec::SQ_LOOP_START] = r[rec::SQ_LOOP_END] = (uint32_t)-1;
            break;
        case OP_SCOPE:   // Analyzers.h:142-149: ringBuffer(4) of 8192 frames, channels = 1, size = 512
            n.props["channels"] = Value::number(1.0); n.props["size"] = Value::number(512.0);
            break;
        case OP_SAMPLE:  // VariablePitchLerpReader(float sampleRate, ...): gainSmoothAlpha(1.0 - exp(-1.0 / (0.01 * sampleRate))), Sample.h:163
            r[rec::SMP_ALPHA] = fbits((float)(1.0 - std::exp(-1.0 / (0.01 * (double)(float)sampleRate)))); break;
        case OP_RAND:    r[rec::S0] = (uint32_t)std::rand(); break;               // Noise.h:42
        case OP_SAMPLESEQ:                                                        // SampleSeq.h:66-68: fade step 0.02
            r[rec::SSQ_PREV] = r[rec::SSQ_NEXT] = 0xFFFFFFFFu;
            r[rec::SSQ_READER0 + 2] = fbits(0.02f); r[rec::SSQ_READER0 + rec::SSQ_READER_DWORDS + 2] = fbits(0.02f);
            if (n.mc) {   // mc/SampleSeq.h:96: readers({MCBufferReader<float>(sr, 8.0), ...}) -> elem::GainFade(sr, 8 ms, 8 ms)
                const double fs = (double)(float)sampleRate;
                const float inS = (float)msToStep(fs, 8.0), outS = (float)((double)(-1.0f) * msToStep(fs, 8.0));
                r[rec::SSQ_FLAGS] = 4u;
                r[rec::SSQ_READER0 + 7] = fbits(inS); r[rec::SSQ_READER0 + rec::SSQ_READER_DWORDS + 7] = fbits(outS);
                r[rec::SSQ_READER0 + 2] = fbits(inS); r[rec::SSQ_READER0 + rec::SSQ_READER_DWORDS + 2] = fbits(inS);   // updateCurrentStep at rest
            }
            break;
        case OP_MCSAMPLE: {                                                       // mc/Sample.h:162-164: playbackRate = 1.0
            const double one = 1.0; uint64_t bits; std::memcpy(&bits, &one, 8);
            r[rec::MCS_RATE] = (uint32_t)(bits & 0xFFFFFFFFu); r[rec::MCS_RATE + 1] = (uint32_t)(bits >> 32);
            break;
        }
        case OP_METRO: {                                                          // wasm/Metro.h:15
            const int64_t is = (int64_t)std::max(2.0, 1000.0 * 0.001 * sampleRate);
            r[rec::P0] = (uint32_t)((uint64_t)is & 0xFFFFFFFFu); r[rec::P1] = (uint32_t)((uint64_t)is >> 32);
            n.props["interval"] = Value::number(1000.0);
            break;
        }
        default: break;
    }
    auto ins = nodes.emplace(id, std::move(n));
    Node& nn = ins.first->second;
    int rc = kOk;
    if (nn.op == OP_DELAY || nn.op == OP_SDELAY) {                                // Delays.h:56, 183: the default size is the HOST's block
        rc = setProperty(id, "size", Value::number((double)hostBlockSize));
    } else if (nn.op == OP_TAPOUT) {                                              // Feedback.h:66-67
        rc = allocRing(nn, (size_t)blockSize);
        if (rc == kOk) writeParamPtr(nn, rec::TAP_PRIVATE, nn.ring.ptr);
        tapNodeIds.push_back(id);
    } else if (nn.op == OP_TAPIN) {
        tapNodeIds.push_back(id);
    } else if (nn.op == OP_METER || nn.op == OP_SNAPSHOT) {                       // per-block readout log (device.h EVT_LOG): 1024 entries of 4 dwords
        rc = allocRing(nn, (size_t)kEventLogEntries * 4u);
        if (rc == kOk) { writeParamPtr(nn, rec::EVT_LOG, nn.ring.ptr); writeParam(nn, rec::EVT_LOGMASK, kEventLogEntries - 1u); }
    } else if (nn.op == OP_SCOPE) {                                               // Analyzers.h:145: MultiChannelRingBuffer(4) x 8192
        rc = allocRing(nn, 4u * 8192u);
        if (rc == kOk) writeParamPtr(nn, rec::SCP_RING, nn.ring.ptr);
    } else if (nn.op == OP_CAPTURE && !nn.mc) {                                   // Capture.h:17: ringBuffer(1, bitceil(sr)); (mc.capture: at commit)
        const size_t cap = (size_t)bitceil((int)(size_t)sampleRate);
        rc = allocRing(nn, cap);
        if (rc == kOk) { writeParamPtr(nn, rec::CAP_RING, nn.ring.ptr); writeParam(nn, rec::CAP_MASK, (uint32_t)(cap - 1)); }
    }
    return rc;
}

int Engine::appendChild(int32_t parent, int32_t child, int32_t channel) {   // Runtime.h:335-366
    auto p = nodes.find(parent);
    if (p == nodes.end()) return kNodeNotFound;
    auto c = nodes.find(child);
    if (c == nodes.end()) return kNodeNotFound;
    p->second.inlets.push_back(Inlet{child, (uint32_t)channel});
    c->second.outlets.push_back(Outlet{parent, (uint32_t)channel});
    return kOk;
}

int Engine::setProperty(int32_t id, const std::string& key, const Value& v) {   // Runtime.h:315-333
    auto it = nodes.find(id);
    if (it == nodes.end()) return kNodeNotFound;
    Node& n = it->second;
    if (n.op == OP_HOST) {                                         // GraphNode::setProperty of the user's node (GraphNode.h:49)
        if (n.hostVt && n.hostVt->setProperty) {
            std::string j;
            toJson(v, j);
            const int rc = n.hostVt->setProperty(n.hostInst, key.c_str(), j.c_str(), n.hostVt->user);
            if (rc != kOk) return rc;
        }
        n.props[key] = v;
        return kOk;
    }
    switch (n.op) {
        case OP_CONST:                                             // Core.h:142-152
            if (key == "value") {
                if (!v.isNumber()) return kInvalidPropertyType;
                writeParamF(n, rec::P0, (float)v.num);
            }
            break;
        case OP_IN:                                                // Math.h:95-105
            if (key == "channel") {
                if (!v.isNumber()) return kInvalidPropertyType;
                writeParam(n, rec::P0, (uint32_t)(int)v.num);
            }
            break;
        case OP_ROOT:                                              // Core.h:33-64
            if (key == "active") {
                if (!v.isBool()) return kInvalidPropertyType;
                n.target = v.b ? 1.0f : 0.0f;                      // fadeIn / fadeOut
                rootUpdateStep(n);
            }
            if (key == "channel") {
                if (!v.isNumber()) return kInvalidPropertyType;    // (reference: bad_variant_access)
                n.channel = (int)v.num;
                writeParam(n, rec::ROOT_CHANNEL, (uint32_t)n.channel);
            }
            if (key == "fadeInMs") {
                if (!v.isNumber()) return kInvalidPropertyType;
                n.inStep = (float)msToStep(sampleRate, v.num);
                rootUpdateStep(n);
            }
            if (key == "fadeOutMs") {
                if (!v.isNumber()) return kInvalidPropertyType;
                n.outStep = (float)((double)(-1.0f) * msToStep(sampleRate, v.num));
                rootUpdateStep(n);
            }
            break;
        case OP_MAXHOLD:                                           // Core.h:292-303
            if (key == "hold") {
                if (!v.isNumber()) return kInvalidPropertyType;
                const double h = sampleRate * 0.001 * v.num;
                writeParam(n, rec::P0, (uint32_t)h);
            }
            break;
        case OP_ONCE:                                              // Core.h:352-366
            if (key == "arm") {
                if (!v.isBool()) return kInvalidPropertyType;
                if (v.b) {
                    const uint32_t idx = n.rec * kRecDwords + rec::S2;
                    if (freshFlag[n.rec]) shadow[idx] = fbits(1.0f);
                    else patches.push_back(Patch{1u, idx, fbits(1.0f), 0u});
                }
            }
            break;
        case OP_SEQ2:                                              // Seq2.h:38-84 (same properties as seq)
        case OP_SEQ:                                               // Core.h:411-458
            if (key == "hold") { if (!v.isBool()) return kInvalidPropertyType; writeParam(n, rec::SEQ_HOLD, v.b ? 1u : 0u); }
            if (key == "loop") { if (!v.isBool()) return kInvalidPropertyType; writeParam(n, rec::SEQ_LOOP, v.b ? 1u : 0u); }
            if (key == "offset") {
                if (!v.isNumber()) return kInvalidPropertyType;
                if (v.num < 0.0) return kInvalidPropertyValue;
                writeParam(n, rec::SEQ_OFFSET, (uint32_t)(uint64_t)v.num);
            }
            if (key == "seq") {
                if (!v.isArray()) return kInvalidPropertyType;
                std::vector<float> data(v.arr.size());
                for (size_t i = 0; i < v.arr.size(); ++i) {
                    if (!v.arr[i].isNumber()) return kInvalidPropertyType;
                    data[i] = (float)v.arr[i].num;
                }
                int rc = allocRing(n, data.size());
                if (rc != kOk) return rc;
                if (!data.empty() && dry) std::memcpy(n.ring.ptr, data.data(), data.size() * 4);
                if (!data.empty() && !dry) HIP_OK(hipMemcpy(n.ring.ptr, data.data(), data.size() * 4, hipMemcpyHostToDevice));
                writeParamPtr(n, rec::SEQ_PTR, n.ring.ptr);
                writeParam(n, rec::SEQ_LEN, (uint32_t)data.size());
                writeParam(n, rec::SEQ_PENDING, 1u);
            }
            break;
        case OP_RAND:                                              // Noise.h:13-23
            if (key == "seed") {
                if (!v.isNumber()) return kInvalidPropertyType;
                writeParam(n, rec::S0, (uint32_t)(int64_t)v.num);
            }
            break;
        case OP_DELAY:                                             // Delays.h:59-82
            if (key == "size") {
                if (!v.isNumber()) return kInvalidPropertyType;
                const int size = (int)v.num;
                if (size < 0) return kInvalidPropertyValue;
                int rc = allocRing(n, (size_t)size);
                if (rc != kOk) return rc;
                writeParamPtr(n, rec::RING_PTR, n.ring.ptr);
                writeParam(n, rec::RING_SIZE, (uint32_t)size);
                writeParam(n, rec::RING_RESET, 1u);
            }
            break;
        case OP_SDELAY:                                            // Delays.h:186-216
            if (key == "size") {
                if (!v.isNumber()) return kInvalidPropertyType;
                const int len = (int)v.num;
                const int size = bitceil(len + blockSize);
                if (size < 0) return kInvalidPropertyValue;
                int rc = allocRing(n, (size_t)size);
                if (rc != kOk) return rc;
                writeParamPtr(n, rec::RING_PTR, n.ring.ptr);
                writeParam(n, rec::RING_SIZE, (uint32_t)size);
                writeParam(n, rec::RING_LEN, (uint32_t)len);
                writeParam(n, rec::RING_RESET, 1u);
            }
            break;
        case OP_SVF:                                               // filters/SVF.h:30-46
            if (key == "mode") {
                if (!v.isString()) return kInvalidPropertyType;
                int m = -1;
                if (v.str == "lowpass") m = 0; if (v.str == "bandpass") m = 1; if (v.str == "highpass") m = 2;
                if (v.str == "notch") m = 3; if (v.str == "allpass") m = 4;
                if (m >= 0) writeParam(n, rec::P0, (uint32_t)m);
            }
            break;
        case OP_SVFSHELF:                                          // filters/SVFShelf.h:29-42
            if (key == "mode") {
                if (!v.isString()) return kInvalidPropertyType;
                int m = -1;
                if (v.str == "lowshelf") m = 0; if (v.str == "highshelf") m = 1;
                if (v.str == "bell" || v.str == "peak") m = 2;
                if (m >= 0) writeParam(n, rec::P0, (uint32_t)m);
            }
            break;
        case OP_MM1P:                                              // filters/MultiMode1p.h:48-62
            if (key == "mode") {
                if (!v.isString()) return kInvalidPropertyType;
                int m = -1;
                if (v.str == "lowpass") m = 0; if (v.str == "highpass") m = 2; if (v.str == "allpass") m = 4;
                if (m >= 0) writeParam(n, rec::P0, (uint32_t)m);
            }
            break;
        case OP_TAPIN: case OP_TAPOUT:                             // Feedback.h:24-38, 70-84
            if (key == "name") {
                if (!v.isString()) return kInvalidPropertyType;
                ResourcePtr r = tapResource(v.str);
                int rc = ensureResourceOnDevice(r);
                if (rc != kOk) return rc;
                n.res = r;
                writeParamPtr(n, rec::TAP_SHARED, dry ? r->dev.ptr : (const void*)(reinterpret_cast<const float*>(r->dev.ptr) + tapSliceOff));
            }
            break;
        case OP_SAMPLE:                                            // Sample.h:25-75
            if (key == "path") {
                if (!v.isString()) return kInvalidPropertyType;
                auto rit = resources.find(v.str);
                if (rit == resources.end()) return kInvalidPropertyValue;
                int rc = ensureResourceOnDevice(rit->second);
                if (rc != kOk) return rc;
                n.res = rit->second;
                writeParamPtr(n, rec::SMP_BUF, n.res->dev.ptr);
                writeParam(n, rec::SMP_LEN, (uint32_t)(n.res->channels.empty() ? 0 : n.res->channels[0].size()));
                writeParam(n, rec::SMP_PENDING, 1u);
            }
            if (key == "mode") {
                if (!v.isString()) return kInvalidPropertyType;
                if (v.str == "trigger") writeParam(n, rec::SMP_MODE, 0u);
                if (v.str == "gate") writeParam(n, rec::SMP_MODE, 1u);
                if (v.str == "loop") writeParam(n, rec::SMP_MODE, 2u);
            }
            if (key == "startOffset" || key == "stopOffset") {
                if (!v.isNumber()) return kInvalidPropertyType;
                const int vi = (int)v.num;
                if (vi < 0) return kInvalidPropertyValue;
                writeParam(n, key == "startOffset" ? rec::SMP_START : rec::SMP_STOP, (uint32_t)vi);
            }
            break;
        case OP_MCSAMPLE:                                          // mc/Sample.h:22-76
            if (key == "path") {
                if (!v.isString()) return kInvalidPropertyType;
                auto rit = resources.find(v.str);
                if (rit == resources.end()) return kInvalidPropertyValue;
                int rc = ensureResourceOnDevice(rit->second);
                if (rc != kOk) return rc;
                n.res = rit->second;
                writeParamPtr(n, rec::SMP_BUF, n.res->dev.ptr);
                writeParam(n, rec::SMP_LEN, (uint32_t)(n.res->channels.empty() ? 0 : n.res->channels[0].size()));
                writeParam(n, rec::SMP_PENDING, 1u);
                for (size_t c = 0; c < n.chanRecs.size(); ++c) writeChannelBuffer(n, (uint32_t)c + 1u, n.chanRecs[c]);
            }
            if (key == "mode") {
                if (!v.isString()) return kInvalidPropertyType;
                if (v.str == "trigger") writeParam(n, rec::SMP_MODE, 0u);
                if (v.str == "gate") writeParam(n, rec::SMP_MODE, 1u);
                if (v.str == "loop") writeParam(n, rec::SMP_MODE, 2u);
            }
            if (key == "startOffset" || key == "stopOffset") {
                if (!v.isNumber()) return kInvalidPropertyType;
                const int vi = (int)v.num;
                if (vi < 0) return kInvalidPropertyValue;
                writeParam(n, key == "startOffset" ? rec::SMP_START : rec::SMP_STOP, (uint32_t)vi);
            }
            if (key == "playbackRate") {
                if (!v.isNumber()) return kInvalidPropertyType;
                uint64_t bits; std::memcpy(&bits, &v.num, 8);
                writeParam(n, rec::MCS_RATE, (uint32_t)(bits & 0xFFFFFFFFu));
                writeParam(n, rec::MCS_RATE + 1, (uint32_t)(bits >> 32));
            }
            break;
        case OP_SCOPE:                                             // Analyzers.h:151-173
            if (key == "size") { if (!v.isNumber()) return kInvalidPropertyType; if (v.num < 256 || v.num > 8192) return kInvalidPropertyValue; }
            if (key == "channels") { if (!v.isNumber()) return kInvalidPropertyType; if (v.num < 0 || v.num > 4) return kInvalidPropertyValue; }
            if (key == "name") { if (!v.isString()) return kInvalidPropertyType; }
            break;
        case OP_TABLE:                                             // Table.h:20-33
            if (key == "path") {
                if (!v.isString()) return kInvalidPropertyType;
                auto rit = resources.find(v.str);
                if (rit == resources.end()) return kInvalidPropertyValue;
                int rc = ensureResourceOnDevice(rit->second);
                if (rc != kOk) return rc;
                n.res = rit->second;
                writeParamPtr(n, rec::TBL_BUF, n.res->dev.ptr);
                writeParam(n, rec::TBL_LEN, (uint32_t)(n.res->channels.empty() ? 0 : n.res->channels[0].size()));
                for (size_t c = 0; c < n.chanRecs.size(); ++c) writeChannelBuffer(n, (uint32_t)c + 1u, n.chanRecs[c]);   // mc.table
            }
            break;
        case OP_SPARSEQ2:                                          // SparSeq2.h:20-54
            if (key == "seq") {
                if (!v.isArray()) return kInvalidPropertyType;
                std::map<double, float> events;
                for (const Value& e : v.arr) {
                    if (!e.isObject()) return kInvalidPropertyType;
                    const Value* val = e.find("value"); const Value* tm = e.find("time");
                    if (!val || !tm || !val->isNumber() || !tm->isNumber()) return kInvalidPropertyType;
                    events.insert({tm->num, (float)val->num});
                }
                const size_t len = events.size();
                std::vector<uint32_t> blob(len * 3 + 2, 0u);        // [len doubles][len floats]
                size_t k = 0;
                for (auto& kv : events) { std::memcpy(&blob[2 * k], &kv.first, 8); std::memcpy(&blob[2 * len + k], &kv.second, 4); ++k; }
                int rc = allocRing(n, blob.size());
                if (rc != kOk) return rc;
                if (dry) std::memcpy(n.ring.ptr, blob.data(), blob.size() * 4);
                else HIP_OK(hipMemcpy(n.ring.ptr, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
                writeParamPtr(n, rec::SPS_SEQ, n.ring.ptr);
                writeParam(n, rec::SPS_LEN, (uint32_t)len);
            }
            if (key == "interpolate") {
                if (!v.isNumber()) return kInvalidPropertyType;
                writeParam(n, rec::SPS_INTERP, (uint32_t)(int32_t)v.num);
            }
            break;
        case OP_SPARSEQ:                                           // SparSeq.h:40-131
            if (key == "offset") {
                if (!v.isNumber()) return kInvalidPropertyType;
                if (v.num < 0.0) return kInvalidPropertyValue;
                writeParam(n, rec::SQ_OFFSET, (uint32_t)(int32_t)(size_t)v.num);
            }
            if (key == "loop") {
                int32_t ls = -1, le = -1;
                if (!(v.type == Value::Null || (v.isBool() && !v.b))) {
                    if (!v.isArray()) return kInvalidPropertyType;
                    if (v.arr.size() < 2 || !v.arr[0].isNumber() || !v.arr[1].isNumber()) return kInvalidPropertyType;   // (the reference reads points[0], points[1] unchecked)
                    ls = (int32_t)v.arr[0].num; le = (int32_t)v.arr[1].num;
                }
                writeParam(n, rec::SQ_NEW_START, (uint32_t)ls);
                writeParam(n, rec::SQ_NEW_END, (uint32_t)le);
                writeParam(n, rec::SQ_LOOP_PENDING, 1u);
            }
            if (key == "follow") { if (!v.isBool()) return kInvalidPropertyType; writeParam(n, rec::SQ_FOLLOW, v.b ? 1u : 0u); }
            if (key == "interpolate") { if (!v.isNumber()) return kInvalidPropertyType; writeParam(n, rec::SQ_INTERP, (uint32_t)(int32_t)v.num); }
            if (key == "tickInterval") {
                if (!v.isNumber()) return kInvalidPropertyType;
                if (v.num < 0.0) return kInvalidPropertyValue;
                const double samples = (double)(float)sampleRate * v.num;       // GraphNode<float>::getSampleRate() * ti
                uint64_t bits; std::memcpy(&bits, &samples, 8);
                writeParam(n, rec::SQ_TICK, (uint32_t)(bits & 0xFFFFFFFFu));
                writeParam(n, rec::SQ_TICK + 1, (uint32_t)(bits >> 32));
            }
            if (key == "seq") {
                if (!v.isArray()) return kInvalidPropertyType;
                std::map<int32_t, float> events;                                 // std::map::insert: the first event of a tick time stays
                for (const Value& e : v.arr) {
                    if (!e.isObject()) return kInvalidPropertyType;
                    const Value* val = e.find("value"); const Value* tm = e.find("tickTime");
                    if (!val || !tm || !val->isNumber() || !tm->isNumber()) return kInvalidPropertyType;
                    events.insert({(int32_t)tm->num, (float)val->num});
                }
                const size_t len = events.size();
                std::vector<uint32_t> blob(2 * len + 1, 0u);                     // [len int32 tick times][len floats]
                size_t k = 0;
                for (auto& kv : events) { std::memcpy(&blob[k], &kv.first, 4); std::memcpy(&blob[len + k], &kv.second, 4); ++k; }
                int rc = allocRing(n, blob.size());
                if (rc != kOk) return rc;
                if (dry) std::memcpy(n.ring.ptr, blob.data(), blob.size() * 4);
                else HIP_OK(hipMemcpy(n.ring.ptr, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
                writeParamPtr(n, rec::SQ_SEQ, n.ring.ptr);
                writeParam(n, rec::SQ_LEN, (uint32_t)len);
                writeParam(n, rec::SQ_SEQ_PENDING, 1u);
            }
            break;
        case OP_CONVOLVE:                                          // wasm/Convolve.h:34-56
            if (key == "path") {
                if (!v.isString()) return kInvalidPropertyType;
                auto rit = resources.find(v.str);
                if (rit == resources.end()) return kInvalidPropertyValue;
                int rc = setConvolverIr(n, rit->second);
                if (rc != kOk) return rc;
            }
            break;
        case OP_SAMPLESEQ:                                         // SampleSeq.h:181-255
            if (key == "duration") {
                if (!v.isNumber()) return kInvalidPropertyType;
                if (v.num <= 0.0) return kInvalidPropertyValue;
                uint64_t bits; std::memcpy(&bits, &v.num, 8);
                writeParam(n, rec::SSQ_DUR, (uint32_t)(bits & 0xFFFFFFFFu));
                writeParam(n, rec::SSQ_DUR + 1, (uint32_t)(bits >> 32));
            }
            if (key == "path") {
                if (!v.isString()) return kInvalidPropertyType;
                auto rit = resources.find(v.str);
                if (rit == resources.end()) return kInvalidPropertyValue;
                int rc = ensureResourceOnDevice(rit->second);
                if (rc != kOk) return rc;
                n.res = rit->second;
                writeParamPtr(n, rec::SSQ_BUF, n.res->dev.ptr);
                writeParam(n, rec::SSQ_BUFLEN, (uint32_t)(n.res->channels.empty() ? 0 : n.res->channels[0].size()));
                writeParam(n, rec::SSQ_BUFPENDING, 1u);
                for (size_t c = 0; c < n.chanRecs.size(); ++c) writeChannelBuffer(n, (uint32_t)c + 1u, n.chanRecs[c]);   // mc.sampleseq
            }
            if (key == "seq") {
                if (!v.isArray()) return kInvalidPropertyType;
                std::map<double, float> events;                     // std::map::insert keeps a key's first entry
                for (const Value& e : v.arr) {
                    if (!e.isObject()) return kInvalidPropertyType;
                    const Value* val = e.find("value"); const Value* tm = e.find("time");
                    if (!val || !tm || !val->isNumber() || !tm->isNumber()) return kInvalidPropertyType;
                    events.insert({tm->num, (float)val->num});
                }
                const size_t len = events.size();
                std::vector<uint32_t> blob(len * 3 + 2, 0u);        // [len doubles][len floats]
                size_t k = 0;
                for (auto& kv : events) {
                    std::memcpy(&blob[2 * k], &kv.first, 8);
                    std::memcpy(&blob[2 * len + k], &kv.second, 4);
                    ++k;
                }
                int rc = allocRing(n, blob.size());
                if (rc != kOk) return rc;
                if (dry) std::memcpy(n.ring.ptr, blob.data(), blob.size() * 4);
                else HIP_OK(hipMemcpy(n.ring.ptr, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
                writeParamPtr(n, rec::SSQ_SEQ, n.ring.ptr);
                writeParam(n, rec::SSQ_SEQLEN, (uint32_t)len);
                writeParam(n, rec::SSQ_SEQPENDING, 1u);
            }
            break;
        case OP_METRO:                                             // wasm/Metro.h:18-34
            if (key == "interval") {
                if (!v.isNumber()) return kInvalidPropertyType;
                if (0 >= v.num) return kInvalidPropertyValue;
                const double is = v.num * 0.001 * sampleRate;
                const int64_t iv = (int64_t)std::max(2.0, is);
                writeParam(n, rec::P0, (uint32_t)((uint64_t)iv & 0xFFFFFFFFu));
                writeParam(n, rec::P1, (uint32_t)((uint64_t)iv >> 32));
            }
            break;
        default: break;
    }
    n.props[key] = v;   // GraphNode::setProperty (GraphNode.h:108-111)
    return kOk;
}

// `malformedTail`: the id list was cut at a non-number entry. Like the reference (Runtime.h:375-380) the roots in front
// of it have been activated by then, and the call fails before anything is deactivated or swapped.
int Engine::activateRoots(const std::vector<int32_t>& ids, bool malformedTail) {   // Runtime.h:368-433
    std::set<int32_t> active;
    for (int32_t id : ids) {
        auto it = nodes.find(id);
        if (it == nodes.end()) return kNodeNotFound;
        if (it->second.op == OP_ROOT) {
            setProperty(id, "active", Value::boolean(true));
            active.insert(id);
        }
    }
    if (malformedTail) return kInvalidInstructionFormat;
    for (int32_t id : currentRoots) {
        auto it = nodes.find(id);
        if (it == nodes.end() || it->second.op != OP_ROOT) continue;
        Node& n = it->second;
        if (active.count(id) == 0) setProperty(id, "active", Value::boolean(false));
        const bool on = n.target > 0.5f;
        const bool settled = std::fabs(n.target - n.gain) <= 1e-6f;
        if (on || !settled) active.insert(id);          // stillRunning(): keep fading roots
    }
    currentRoots.swap(active);
    shouldRebuild = true;
    return kOk;
}

// `renderLock` holds `mu` on entry and on return; buildPlan releases it while it plans (the render thread keeps
// rendering the current plan meanwhile — the role of the reference's SPSC sequence queue, Runtime.h:207-216, 277-285).
int Engine::commit(std::unique_lock<std::mutex>& renderLock) {   // Runtime.h:202-206
    if (shouldRebuild || rebuildOwed || (planStale && (current || pending))) {
        planStale = false;
        auto t0 = std::chrono::steady_clock::now();
        auto p = buildPlan(renderLock);
        // (not a reference code path: its buildRenderSequence cannot fail. The roots stay swapped as in the reference;
        // the rebuild stays owed so that the next commit retries instead of rendering the old sequence forever.)
        if (!p) { rebuildOwed = true; return kUnsupportedGraph; }
        rebuildOwed = false;
        // mc.capture: the reference (re)creates the node's multi-channel ring whenever a render sequence that holds it is pushed
        // (GraphRenderSequence.h:165-169 sets `_internal:numChildren`, mc/Capture.h:21-31 allocates children - 1 channels of
        // bitceil(sr) frames): unread samples are dropped, the change detector and the relay flag live on
        bool ringsReset = false;
        for (int32_t id : p->mcCaptureIds) {
            auto it = nodes.find(id);
            if (it == nodes.end() || it->second.op != OP_CAPTURE || !it->second.mc) continue;
            Node& n = it->second;
            const size_t chans = n.inlets.size() > 1 ? n.inlets.size() - 1 : 0, cap = (size_t)bitceil((int)(size_t)sampleRate);
            if (chans == 0) continue;
            if (n.ring.bytes != chans * cap * sizeof(float)) {
                const int rc = allocRing(n, chans * cap);
                if (rc != kOk) return rc;
                writeParamPtr(n, rec::CAP_RING, n.ring.ptr);
                for (uint32_t cr : n.chanRecs) { writeRec(cr, rec::CAP_RING, shadow[(size_t)n.rec * kRecDwords + rec::CAP_RING]); writeRec(cr, rec::CAP_RING + 1, shadow[(size_t)n.rec * kRecDwords + rec::CAP_RING + 1]); }
            }
            writeParam(n, rec::CAP_MASK, (uint32_t)(cap - 1));
            writeParam(n, rec::CAP_CHANS, (uint32_t)chans);
            writeParam(n, rec::CAP_WRITE, 0u); writeParam(n, rec::CAP_READ, 0u);
            ringsReset = true;
        }
        // (the reference drops the unread samples when the sequence is PUSHED, not when it is first rendered: an event poll between
        //  this commit and the next block finds the new ring empty — the resets go to the device now, behind the blocks in flight)
        if (ringsReset && !dry) { const int rc = flushPending(); if (rc != kOk) return rc; }
        pending = p;
        shouldRebuild = false;
        st.plansBuilt++;
        st.lastPlanBuildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return kOk;
}

int Engine::apply(const Value& batch) {   // Runtime.h:170-218
    std::lock_guard<std::mutex> control(ctl);
    std::unique_lock<std::mutex> lock(mu);
    if (!dry && hipSetDevice(device) != hipSuccess) return kHipError;
    residentStop();
    if (!batch.isArray()) return kInvalidInstructionFormat;
    shouldRebuild = false;   // a local in the reference: ACTIVATE_ROOTS and COMMIT must share a batch
    static const bool applyTiming = std::getenv("ELEMHIP_APPLY_TIMING") != nullptr;   // time per instruction kind of a batch, on stderr
    double kindUs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    struct Report {
        const double* us; bool on;
        ~Report() { if (on) std::fprintf(stderr, "[elemhip] apply: create %.1f delete %.1f append %.1f set %.1f activate %.1f commit %.1f us\n", us[0], us[1], us[2], us[3], us[4], us[5]); }
    } report{kindUs, applyTiming};
    for (const Value& next : batch.arr) {
        if (!next.isArray()) return kInvalidInstructionFormat;
        const auto& ar = next.arr;
        if (ar.empty() || !ar[0].isNumber()) return kInvalidInstructionFormat;
        const int cmd = (int)ar[0].num;
        const auto tCmd = applyTiming ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        struct Acc {
            double* slot; std::chrono::steady_clock::time_point t0; bool on;
            ~Acc() { if (on) *slot += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
        } acc{&kindUs[(cmd >= 0 && cmd < 6) ? cmd : 7], tCmd, applyTiming};
        int res = kOk;
        static const Value undef;
        auto arg = [&](size_t i) -> const Value& { return i < ar.size() ? ar[i] : undef; };
        switch (cmd) {
            case 0:   // CREATE_NODE
                if (!arg(1).isNumber() || !arg(2).isString()) { res = kInvalidInstructionFormat; break; }
                res = createNode((int32_t)arg(1).num, arg(2).str);
                break;
            case 3:   // SET_PROPERTY
                if (!arg(1).isNumber() || !arg(2).isString()) { res = kInvalidInstructionFormat; break; }
                res = setProperty((int32_t)arg(1).num, arg(2).str, arg(3));
                break;
            case 2:   // APPEND_CHILD
                if (!arg(1).isNumber() || !arg(2).isNumber() || !arg(3).isNumber()) { res = kInvalidInstructionFormat; break; }
                res = appendChild((int32_t)arg(1).num, (int32_t)arg(2).num, (int32_t)arg(3).num);
                break;
            case 4: { // ACTIVATE_ROOTS
                if (!arg(1).isArray()) { res = kInvalidInstructionFormat; break; }
                std::vector<int32_t> ids;
                bool bad = false;
                for (const Value& v : arg(1).arr) { if (!v.isNumber()) { bad = true; break; } ids.push_back((int32_t)v.num); }
                // the reference activates the roots preceding a malformed id before failing
                res = activateRoots(ids, bad);
                shouldRebuild = true;
                break;
            }
            case 5:   // COMMIT_UPDATES
                res = commit(lock);
                break;
            default: break;
        }
        if (res != kOk) return res;
    }
    return kOk;
}

// ---- event relay ------------------------------------------------------------------------------------
// ---- the event relay (Runtime.h:437-446, GraphRenderSequence.h:189-198, builtins/Analyzers.h, Capture.h, mc/Capture.h) -------------
//
// The reference drains lock-free queues: its audio thread never notices a relay. Here the readouts live in device memory, so
// the relay (a) takes the render lock only to ENQUEUE a device-side snapshot of the event nodes' records behind the blocks
// already queued (stream order makes it consistent) and, at the end, to queue the read-position updates as parameter patches;
// (b) waits for that snapshot, fetches the ring ranges it names and calls the host back WITHOUT the render lock — a render
// thread that calls process() meanwhile is not held up by a synchronise or a copy (r04 held `mu` across both).
// What a ring range holds cannot change under the copy: kernels only write ahead of the write position the snapshot shows,
// the relay is the only reader, and the rings themselves are freed under `ctl`, which the relay holds throughout.
//
// blockwise = false: the reference's processQueuedEvents — per node the NEWEST readout since the last relay.
// blockwise = true:  what the reference's offline caller produces by relaying after EVERY block (offline-renderer/index.ts:112-120),
//   reconstructed after a whole launch set from the per-block readout logs the kernels keep (device.h EVT_LOG): every block's
//   events in block order, nodes in render order inside a block. Exact while a relay window stays within eventWindowBlocks().
int Engine::processQueuedEvents(void (*cb)(const char*, const char*, void*), void* user, bool blockwise) {
    std::lock_guard<std::mutex> control(ctl);
    if (dry || !cb) return kOk;
    struct Item { Node* n; size_t recOff; uint32_t order; };
    std::vector<Item> items;
    std::shared_ptr<Plan> plan;
    uint64_t windowBlocks = 0, blocksNow = 0;
    // A host block longer than the engine's renders as k slices, each an engine block with readouts of its own; the reference's nodes
    // see ONE block (a meter reports min / max over all its frames, Analyzers.h:38-39; the relay runs once per host block): the slices
    // of a host block are put back together here — `hostEnds[h]` = slices of the window rendered when host block h ended.
    const bool sliced = hostBlockSize != blockSize;
    std::vector<uint64_t> hostEnds;
    {   // ---- (a) under the render lock: snapshot of the records, stream-ordered behind everything rendered so far ----
        RenderGuard lock(*this);
        if (!current) return kOk;
        if (hipSetDevice(device) != hipSuccess) return kHipError;
        plan = current;
        uint32_t order = 0;
        for (auto& en : plan->eventNodes) {
            auto nit = nodes.find(en.first), rit = nodes.find(en.second);
            if (nit == nodes.end() || rit == nodes.end()) continue;
            auto a = rit->second.props.find("active");                       // GraphRenderSequence.h:192
            if (a == rit->second.props.end() || !a->second.isBool() || !a->second.b) continue;
            bool seen = false;                                              // (an mc.* node is one plan entry per output channel: relayed once)
            for (const Item& it : items) if (it.n == &nit->second) { seen = true; break; }
            if (seen) continue;
            items.push_back({&nit->second, items.size() * kRecDwords * 4, order++});
        }
        blocksNow = st.blocksRendered;
        windowBlocks = blocksNow - relayBlocksMark;
        if (sliced) {
            for (uint64_t e : hostBlockEnds) if (e > relayBlocksMark && e <= blocksNow) hostEnds.push_back(e - relayBlocksMark);
            if (windowBlocks && (hostEnds.empty() || hostEnds.back() != windowBlocks)) hostEnds.push_back(windowBlocks);
        }
        if (items.empty()) { relayBlocksMark = blocksNow; hostBlockEnds.clear(); return kOk; }
        const size_t need = items.size() * kRecDwords * 4;
        if (need > relayBytes) {
            const size_t cap = std::max<size_t>(need * 2, 16384);
            uint8_t* d = nullptr; uint8_t* h = nullptr;
            HIP_OK(hipMalloc(reinterpret_cast<void**>(&d), cap));
            HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&h), cap, hipHostMallocDefault));
            if (dRelay) deferredFree.push_back(dRelay);
            if (hRelay) (void)hipHostFree(hRelay);
            dRelay = d; hRelay = h; relayBytes = cap;
        }
        if (!relayStream) {
            HIP_OK(hipStreamCreateWithFlags(&relayStream, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&evRelay, hipEventDisableTiming));
        }
        // (parameter patches still waiting for the next block — a read position written by the previous relay — go first)
        int rc = flushPending();
        if (rc != kOk) return rc;
        for (const Item& it : items)
            HIP_OK(hipMemcpyAsync(dRelay + it.recOff, dRecs + (size_t)it.n->rec * kRecDwords, kRecDwords * 4, hipMemcpyDeviceToDevice, stream));
        HIP_OK(hipEventRecord(evRelay, stream));
    }
    // ---- (b) without the render lock: wait for the snapshot, fetch what it names ----
    HIP_OK(hipStreamWaitEvent(relayStream, evRelay, 0));
    HIP_OK(hipMemcpyAsync(hRelay, dRelay, items.size() * kRecDwords * 4, hipMemcpyDeviceToHost, relayStream));
    HIP_OK(hipStreamSynchronize(relayStream));
    auto numStr = [](float v) { char b[64]; std::snprintf(b, sizeof b, "%.17g", (double)v); return std::string(b); };
    auto srcOf = [](const Node& n) {
        std::string src = "null";
        auto nm = n.props.find("name");
        if (nm != n.props.end() && nm->second.isString()) { src = "\""; for (char ch : nm->second.str) { if (ch == '"' || ch == '\\') src += '\\'; src += ch; } src += "\""; }
        return src;
    };
    // a stretch [from, from + count) of a device ring of `cap` entries of `entryFloats` floats each, wrapped, into `dst`
    auto fetchRing = [&](const float* base, uint32_t cap, uint32_t from, uint32_t count, uint32_t entryFloats, float* dst) -> bool {
        if (!count) return true;
        from %= cap;
        const uint32_t first = std::min(count, cap - from);
        if (hipMemcpyAsync(dst, base + (size_t)from * entryFloats, (size_t)first * entryFloats * 4, hipMemcpyDeviceToHost, relayStream) != hipSuccess) return false;
        if (count > first && hipMemcpyAsync(dst + (size_t)first * entryFloats, base, (size_t)(count - first) * entryFloats * 4, hipMemcpyDeviceToHost, relayStream) != hipSuccess) return false;
        return hipStreamSynchronize(relayStream) == hipSuccess;
    };
    struct Ev { uint64_t block; uint32_t order; const char* type; std::string json; };
    std::vector<Ev> evs;
    struct Wb { Node* n; uint32_t dword; uint32_t value; };
    std::vector<Wb> writeBack;
    const uint64_t lastSlice = windowBlocks ? windowBlocks - 1 : 0;
    // the HOST block (of this relay window) an engine block `fromEnd` blocks before the newest belongs to
    auto blockOf = [&](uint64_t fromEnd) -> uint64_t {
        const uint64_t s_ = fromEnd > lastSlice ? 0 : lastSlice - fromEnd;
        return sliced ? (uint64_t)(std::upper_bound(hostEnds.begin(), hostEnds.end(), s_) - hostEnds.begin()) : s_;
    };
    const uint64_t hostBlocks = sliced ? hostEnds.size() : windowBlocks;            // host blocks in the window
    const uint64_t lastBlock = hostBlocks ? hostBlocks - 1 : 0;
    const uint32_t hostFrames = (uint32_t)hostBlockSize;
    for (const Item& it : items) {
        Node& n = *it.n;
        const uint32_t* rc_ = reinterpret_cast<const uint32_t*>(hRelay + it.recOff);
        if (n.op == OP_SCOPE) {                                           // Analyzers.h:192-245, MultiChannelRingBuffer.h:61-86
            auto numOr = [&](const char* k, double dflt) { auto q = n.props.find(k); return (q != n.props.end() && q->second.isNumber()) ? q->second.num : dflt; };
            const size_t size = (size_t)numOr("size", 512.0), channels = std::min<size_t>(4, (size_t)numOr("channels", 1.0));
            const uint32_t cap = 8192u, mask = cap - 1u;
            const uint32_t wEnd = rc_[rec::SCP_WRITE];
            uint32_t r = rc_[rec::SCP_READ];
            if (!n.ring.ptr || size == 0 || size >= cap) continue;
            // blockwise: the write position after each block of the window (whole blocks of blockSize frames), oldest first
            const uint64_t steps = blockwise ? std::max<uint64_t>(1, std::min<uint64_t>(hostBlocks, (cap - 1) / hostFrames)) : 1;
            std::vector<std::pair<uint64_t, uint32_t>> emits;            // (host block, read position of the emitted frame)
            for (uint64_t s_ = 0; s_ < steps; ++s_) {
                const uint32_t w = (wEnd - (uint32_t)((steps - 1 - s_) * (uint64_t)hostFrames)) & mask;
                const uint32_t full = w > r ? w - r : ((cap - (r - w)) & mask);
                if (!(full > size)) continue;
                emits.emplace_back(lastBlock - std::min<uint64_t>(lastBlock, steps - 1 - s_), r);
                r = (uint32_t)((r + size) & mask);
            }
            if (emits.empty()) continue;
            const uint32_t r0 = emits.front().second, span = (uint32_t)(emits.size() * size);
            std::vector<float> host((size_t)channels * span);
            bool ok = true;
            for (size_t ch = 0; ch < channels && ok; ++ch) ok = fetchRing((const float*)n.ring.ptr + ch * cap, cap, r0, span, 1, host.data() + ch * span);
            if (!ok) return kHipError;
            const std::string src = srcOf(n);
            for (size_t e = 0; e < emits.size(); ++e) {
                std::string j = "{\"source\": " + src + ", \"data\": [";
                for (size_t ch = 0; ch < channels; ++ch) {
                    j += ch ? ", [" : "[";
                    for (size_t i = 0; i < size; ++i) { if (i) j += ", "; j += numStr(host[ch * span + e * size + i]); }
                    j += "]";
                }
                j += "]}";
                evs.push_back({emits[e].first, it.order, "scope", std::move(j)});
            }
            writeBack.push_back({&n, rec::SCP_READ, r});
            continue;
        }
        if (n.op == OP_CAPTURE) {                                         // Capture.h:60-95 / mc/Capture.h:107-146: drain the ring(s) into the relay, emit once the gate fell
            const uint32_t mask = shadow[(size_t)n.rec * kRecDwords + rec::CAP_MASK], cap = mask + 1u;
            const uint32_t chans = n.mc ? shadow[(size_t)n.rec * kRecDwords + rec::CAP_CHANS] : 1u;
            const uint32_t w = rc_[rec::CAP_WRITE], r = rc_[rec::CAP_READ], ready = rc_[rec::CAP_READY];
            const uint32_t avail = w > r ? w - r : ((cap - (r - w)) & mask);
            if (avail > 0 && n.ring.ptr && chans > 0) {
                if (n.mc) { if (n.relayCh.size() != chans) n.relayCh.resize(chans); }      // (pendingEventData.resize(numChansToRead))
                for (uint32_t k = 0; k < chans; ++k) {
                    std::vector<float>& dst = n.mc ? n.relayCh[k] : n.relay;
                    const size_t at = dst.size();
                    dst.resize(at + avail);
                    if (!fetchRing((const float*)n.ring.ptr + (size_t)k * cap, cap, r, avail, 1, dst.data() + at)) return kHipError;
                }
                writeBack.push_back({&n, rec::CAP_READ, (r + avail) & mask});
            }
            if (ready) {
                writeBack.push_back({&n, rec::CAP_READY, 0u});
                std::string j = "{\"source\": " + srcOf(n) + ", \"data\": [";
                if (n.mc) {
                    for (size_t k = 0; k < n.relayCh.size(); ++k) {
                        j += k ? ", [" : "[";
                        for (size_t i = 0; i < n.relayCh[k].size(); ++i) { if (i) j += ", "; j += numStr(n.relayCh[k][i]); }
                        j += "]";
                        n.relayCh[k].clear();
                    }
                } else {
                    for (size_t i = 0; i < n.relay.size(); ++i) { if (i) j += ", "; j += numStr(n.relay[i]); }
                    n.relay.clear();
                }
                j += "]}";
                evs.push_back({lastBlock, it.order, n.mc ? "mc.capture" : "capture", std::move(j)});
            }
            continue;
        }
        // meter / snapshot
        const uint32_t count = rc_[rec::EVT_COUNT];
        float fa, fb; std::memcpy(&fa, &rc_[rec::EVT_A], 4); std::memcpy(&fb, &rc_[rec::EVT_B], 4);
        const uint32_t lmask = shadow[(size_t)n.rec * kRecDwords + rec::EVT_LOGMASK], lcap = lmask + 1u;
        // The reference's readout queue (SingleWriterSingleReaderQueue.h, capacity 32) cannot tell "32 x k pushes since the last relay"
        // from "none": its write position is back on the read position, size() answers 0 and processEvents reports nothing — a meter
        // polled every 32nd block, a snapshot that latches exactly 32 times per block (a 3 kHz train at 48 kHz and 512 frames). Kept.
        auto wrapsToEmpty = [](uint32_t pushes) { return pushes != 0u && (pushes & 31u) == 0u; };
        if (n.op == OP_METER) {                                           // Analyzers.h:23-62
            const uint32_t fresh = count - n.eventCount;
            if (!fresh) continue;
            if (n.ring.ptr && (sliced || (blockwise && fresh > 1u))) {
                const uint32_t take = std::min(fresh, lcap);
                std::vector<uint32_t> e((size_t)take * 4);
                if (!fetchRing((const float*)n.ring.ptr, lcap, count - take, take, 4, reinterpret_cast<float*>(e.data()))) return kHipError;
                const std::string src = srcOf(n);
                // one readout per HOST block: the slices of a host block folded into one min / max (unsliced: every group is one entry)
                struct G { uint64_t block; float mn, mx; };
                std::vector<G> groups;
                for (uint32_t k = 0; k < take; ++k) {
                    float mn, mx; std::memcpy(&mn, &e[4 * k + 1], 4); std::memcpy(&mx, &e[4 * k + 2], 4);
                    const uint64_t b = blockOf(take - 1 - k);
                    if (!groups.empty() && groups.back().block == b) { G& g = groups.back(); if (mn < g.mn) g.mn = mn; if (mx > g.mx) g.mx = mx; }
                    else groups.push_back({b, mn, mx});
                }
                auto emit = [&](const G& g) { evs.push_back({g.block, it.order, "meter", "{\"min\": " + numStr(g.mn) + ", \"max\": " + numStr(g.mx) + ", \"source\": " + src + "}"}); };
                if (blockwise) for (const G& g : groups) emit(g);
                else if (!groups.empty() && !wrapsToEmpty((uint32_t)groups.size())) emit(groups.back());   // (the reference queued one readout per host block)
            } else if (!wrapsToEmpty(fresh)) evs.push_back({lastBlock, it.order, "meter", "{\"min\": " + numStr(fa) + ", \"max\": " + numStr(fb) + ", \"source\": " + srcOf(n) + "}"});
            n.eventCount = count;
        } else {                                                          // Analyzers.h:83-131
            const uint32_t blk = rc_[rec::EVT_BLK], logn = rc_[rec::EVT_LOGN];
            if (count == n.eventCount) { n.logRelayed = logn; continue; }
            const uint32_t fresh = logn - n.logRelayed;
            if (blockwise && n.ring.ptr && fresh >= 1u) {
                const uint32_t take = std::min(fresh, lcap);
                std::vector<uint32_t> e((size_t)take * 4);
                if (!fetchRing((const float*)n.ring.ptr, lcap, logn - take, take, 4, reinterpret_cast<float*>(e.data()))) return kHipError;
                const std::string src = srcOf(n);
                for (uint32_t k = 0; k < take;) {                        // (the log entries of one HOST block: its newest latch, its pushes summed)
                    const uint64_t b = blockOf((uint64_t)(blk - 1u - e[4 * k]));
                    uint32_t pushes = 0, last = k;
                    for (; k < take && blockOf((uint64_t)(blk - 1u - e[4 * k])) == b; ++k) { pushes += e[4 * k + 2]; last = k; }
                    float v; std::memcpy(&v, &e[4 * last + 1], 4);
                    if (wrapsToEmpty(pushes)) continue;                   // (per-block relay: the pushes of that block alone)
                    evs.push_back({b, it.order, "snapshot", "{\"source\": " + src + ", \"data\": " + numStr(v) + "}"});
                }
            } else if (!wrapsToEmpty(count - n.eventCount)) evs.push_back({lastBlock, it.order, "snapshot", "{\"source\": " + srcOf(n) + ", \"data\": " + numStr(fb) + "}"});
            n.eventCount = count; n.logRelayed = logn;
        }
    }
    {   // ---- (c) read positions back to the device: parameter patches, applied in stream order at once (no synchronise) ----
        RenderGuard lock(*this);
        for (const Wb& w : writeBack) writeParam(*w.n, w.dword, w.value);
        relayBlocksMark = blocksNow;
        while (!hostBlockEnds.empty() && hostBlockEnds.front() <= blocksNow) hostBlockEnds.pop_front();
        if (!writeBack.empty()) { const int rc = flushPending(); if (rc != kOk) return rc; }
    }
    // ---- (d) the host's callbacks, in block order (stable: nodes stay in render order inside a block) ----
    if (blockwise) std::stable_sort(evs.begin(), evs.end(), [](const Ev& a, const Ev& b) { return a.block != b.block ? a.block < b.block : a.order < b.order; });
    for (const Ev& e : evs) cb(e.type, e.json.c_str(), user);
    return kOk;
}

// How many blocks may pass between two blockwise relays for their result to be exactly the per-block relay's: the per-block
// readout logs hold 1024 blocks; a scope ring (8192 frames, `size` of them per event) must not overrun inside a window; a
// capture node's take is placed by the relay that sees its gate fall, so it wants a relay per block.
uint32_t Engine::eventWindowBlocks() {
    std::lock_guard<std::mutex> control(ctl);
    RenderGuard lock(*this);
    const std::shared_ptr<Plan> pl = pending ? pending : current;
    // in HOST blocks (what the caller counts in): a host block of k slices fills k entries of the per-block readout logs
    const uint32_t perHost = (uint32_t)((hostBlockSize + blockSize - 1) / blockSize);
    uint32_t w = std::max(1u, kEventLogEntries / std::max(1u, perHost));
    if (!pl) return w;
    for (auto& en : pl->eventNodes) {
        auto nit = nodes.find(en.first);
        if (nit == nodes.end()) continue;
        const Node& n = nit->second;
        if (n.op == OP_CAPTURE) return 1u;
        if (n.op == OP_SCOPE) {
            auto q = n.props.find("size");
            const double size = (q != n.props.end() && q->second.isNumber()) ? q->second.num : 512.0;
            // a scope whose `size` is below the block hands on less per relay than a block brings: its ring overruns under a per-block
            // relay too, and where it does depends on every single relay — only a relay per block reproduces that
            if (size < (double)hostBlockSize) return 1u;
            const double room = 8192.0 - 1.0 - std::max(1.0, size);
            w = std::min<uint32_t>(w, (uint32_t)std::max(1.0, std::floor(room / (double)hostBlockSize)));
        }
    }
    return w;
}

// ---- gc / resources -------------------------------------------------------------------------------
size_t Engine::gc(int32_t* out, size_t cap) {   // Runtime.h:220-272
    std::lock_guard<std::mutex> control(ctl);
    RenderGuard lock(*this);
    if (!dry) (void)hipSetDevice(device);
    std::vector<int32_t> pruned;
    for (auto it = nodes.begin(); it != nodes.end(); ++it) {
        const int32_t id = it->first;
        const bool held = (current && current->holdsNode(id)) || (pending && pending->holdsNode(id));
        if (!held) pruned.push_back(id);
    }
    std::set<int32_t> prunedSet(pruned.begin(), pruned.end());
    for (int32_t id : pruned) {
        Node& n = nodes.at(id);
        for (auto& inlet : n.inlets) {
            auto c = nodes.find(inlet.source);
            if (c == nodes.end() || prunedSet.count(inlet.source)) continue;
            auto& o = c->second.outlets;
            o.erase(std::remove_if(o.begin(), o.end(), [&](const Outlet& x) { return x.dest == id; }), o.end());
        }
    }
    for (int32_t id : pruned) {
        Node& n = nodes.at(id);
        // (`mu` free does NOT mean the device is idle: elemhip_process_blocks_host drops it between launch sets while one renders.
        //  Device memory is released after the next synchronize — freeDeferred — never under a kernel that may still read it.)
        if (n.ring.ptr) { if (dry) std::free(n.ring.ptr); else deferredFree.push_back(n.ring.ptr); }
        if (n.hostInst && n.hostVt && n.hostVt->destroy) n.hostVt->destroy(n.hostInst, n.hostVt->user);
        // drop queued writes aimed at the record before it is recycled
        const uint32_t lo = n.rec * kRecDwords, hi = lo + kRecDwords;
        patches.erase(std::remove_if(patches.begin(), patches.end(), [&](const Patch& p) { return p.kind != 2 && p.index >= lo && p.index < hi; }), patches.end());
        if (freshFlag[n.rec]) { freshRecs.erase(std::remove(freshRecs.begin(), freshRecs.end(), n.rec), freshRecs.end()); freshFlag[n.rec] = 0; }
        recClones.erase(std::remove_if(recClones.begin(), recClones.end(), [&](const std::pair<uint32_t, uint32_t>& c) { return c.first == n.rec; }), recClones.end());
        freeRecs.push_back(n.rec);
        for (uint32_t cr : n.chanRecs) {
            const uint32_t clo = cr * kRecDwords, chi = clo + kRecDwords;
            patches.erase(std::remove_if(patches.begin(), patches.end(), [&](const Patch& p) { return p.kind != 2 && p.index >= clo && p.index < chi; }), patches.end());
            if (freshFlag[cr]) { freshRecs.erase(std::remove(freshRecs.begin(), freshRecs.end(), cr), freshRecs.end()); freshFlag[cr] = 0; }
            freeRecs.push_back(cr);
        }
        if (n.op == OP_TAPIN || n.op == OP_TAPOUT) tapNodeIds.erase(std::remove(tapNodeIds.begin(), tapNodeIds.end(), id), tapNodeIds.end());
        convStaleNodes.erase(id);
        nodes.erase(id);
    }
    if (!pruned.empty()) { if (++nodesEpoch == 0u) nodesEpoch = 1u; }      // (Inlet::src memos name erased nodes now)
    std::sort(pruned.begin(), pruned.end());
    size_t k = 0;
    for (int32_t id : pruned) { if (out && k < cap) out[k] = id; ++k; }
    lastPruned.swap(pruned);
    return k;
}

size_t Engine::lastGc(int32_t* out, size_t cap) {
    std::lock_guard<std::mutex> control(ctl);
    RenderGuard lock(*this);
    size_t k = 0;
    for (int32_t id : lastPruned) { if (out && k < cap) out[k] = id; ++k; }
    return k;
}

bool Engine::hasNode(int32_t id) {
    std::lock_guard<std::mutex> control(ctl);
    RenderGuard lock(*this);
    return nodes.find(id) != nodes.end();
}

// Runtime::reset (Runtime.h:448-458) forwards to every node; of the built-ins only SampleNode reacts: both readers get
// noteOff(), i.e. target gain 0 (Sample.h:78-81, 174-177). The reader state lives in the node record.
void Engine::reset() {
    std::lock_guard<std::mutex> control(ctl);
    RenderGuard lock(*this);
    for (auto& kv : nodes) {
        Node& n = kv.second;
        if (n.op == OP_SAMPLE) {
            writeParamF(n, rec::SMP_READER0, 0.0f);
            writeParamF(n, rec::SMP_READER0 + rec::SMP_READER_DWORDS, 0.0f);
        } else if (n.op == OP_MCSAMPLE) {
            writeParam(n, rec::MCS_RESET, 1u);          // both readers noteOff() at the next block (mc/Sample.h:78-81)
        } else if (n.op == OP_HOST && n.hostVt && n.hostVt->reset) {
            n.hostVt->reset(n.hostInst, n.hostVt->user);
        }
    }
}

int Engine::registerNodeType(const std::string& type, const HostVTable& vt) {   // Runtime.h:480-487
    std::lock_guard<std::mutex> control(ctl);
    RenderGuard lock(*this);
    if (hostTypes.count(type) || opTable().count(type)) return kNodeTypeAlreadyExists;
    if (!vt.process) return kInvalidInstructionFormat;
    hostTypes.emplace(type, std::unique_ptr<HostVTable>(new HostVTable(vt)));
    return kOk;
}

static std::string idToHex(int32_t id) {   // Types.h:16-27
    char b[16]; std::snprintf(b, sizeof b, "%08x", (uint32_t)id);
    return b;
}

std::string Engine::snapshotJson() {   // Runtime.h:489-498: { nodeIdToHex(id): node.getProperties() }
    std::lock_guard<std::mutex> control(ctl);
    RenderGuard lock(*this);
    std::map<std::string, const Node*> sorted;
    for (auto& kv : nodes) sorted.emplace(idToHex(kv.first), &kv.second);
    std::string out = "{";
    bool first = true;
    for (auto& kv : sorted) {
        if (!first) out += ',';
        first = false;
        out += '"' + kv.first + "\":{";
        bool f2 = true;
        for (auto& pr : kv.second->props) {
            if (!f2) out += ',';
            f2 = false;
            toJson(Value::string(pr.first), out); out += ':'; toJson(pr.second, out);
        }
        out += '}';
    }
    out += '}';
    return out;
}

std::string Engine::sharedResourceKeysJson() {   // SharedResourceMap::keys (SharedResource.h)
    std::lock_guard<std::mutex> control(ctl);
    RenderGuard lock(*this);
    std::vector<std::string> keys;
    for (auto& kv : resources) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    std::string out = "[";
    for (size_t i = 0; i < keys.size(); ++i) { if (i) out += ','; toJson(Value::string(keys[i]), out); }
    out += ']';
    return out;
}

bool Engine::addSharedResource(const std::string& name, const float* const* ch, size_t nCh, size_t nSamples) {
    std::lock_guard<std::mutex> control(ctl);
    RenderGuard lock(*this);
    if (resources.count(name)) return false;                 // insert-only (SharedResource.h:61-63)
    auto r = std::make_shared<Resource>();
    for (size_t c = 0; c < nCh; ++c) r->channels.emplace_back(ch[c], ch[c] + nSamples);
    resources.emplace(name, r);
    return true;
}

void Engine::pruneSharedResources() {   // SharedResource.h:94-102
    std::lock_guard<std::mutex> control(ctl);
    RenderGuard lock(*this);
    if (!dry) (void)hipSetDevice(device);
    for (auto it = resources.begin(); it != resources.end();) {
        if (it->second.use_count() == 1) {
            if (it->second->dev.ptr) { if (dry) std::free(it->second->dev.ptr); else deferredFree.push_back(it->second->dev.ptr); }
            for (DevBuf& d : it->second->devCh) if (d.ptr) { if (dry) std::free(d.ptr); else deferredFree.push_back(d.ptr); }
            it->second->dev.ptr = nullptr;
            for (DevBuf& d : it->second->devCh) d.ptr = nullptr;
            it = resources.erase(it);
        } else ++it;
    }
}

int Engine::setOption(const std::string& key, double value) {
    std::lock_guard<std::mutex> control(ctl);
    RenderGuard lock(*this);
    if (key == "use_graph") { useGraph = value != 0.0; return kOk; }
    if (key == "sync_poll") { syncPoll = value != 0.0; return kOk; }   // elemhip_process: wait for the epilogue's word in mapped host memory (1) or synchronise the stream (0)
    // elemhip_process through a kernel that stays on the GPU between calls (resident.hip): opt-in, the GPU spins while the host is away
    if (key == "resident") { residentOpt = value != 0.0; residentStreak = 0; return kOk; }
    if (key == "resident_idle_us") { residentIdleUs = (uint32_t)std::max(10.0, std::min(5e6, value)); return kOk; }
    if (key == "resident_after") { residentAfter = (uint32_t)std::max(1.0, std::min(1e6, value)); return kOk; }
    if (key == "host_out_direct") { hostOutDirect = value != 0; return kOk; }   // elemhip_process: epilogue writes the pinned host block itself
    if (key == "conv_direct_io") { convDirectIo = value != 0; return kOk; }
    if (key == "conv_long_mac_lds") { convLongMacMode = (uint32_t)std::max(0.0, std::min(3.0, value)); return kOk; }   // long-partition sums: 0 the register kernel over L2 (runs of 16 chunks), 1 the LDS-tiled kernel, 2 the register kernel with runs of 32 chunks
    if (key == "conv_long") { convLong = value != 0; return kOk; }   // IRs set from now on get (or do not get) long-partition spectra; sets of older IRs keep theirs
    if (key == "conv_mfma") { convMfma = std::max(0, std::min(1, (int)value)); return kOk; }   // partition MAC of launch sets: 1 matrix cores (default), 0 packed vector FMAs
    if (key == "skip_idle_launches") { skipIdleLaunches = value != 0; dropGraphs(); return kOk; }
    if (key == "fuse_epilogue") { fuseEpilogue = value != 0; dropGraphs(); return kOk; }
    if (key == "spec_block_graph") { specBlockGraph = value != 0; dropGraphs(); return kOk; }   // elemhip_process: replay the launch set of one from a hipGraph
    if (key == "spec_blocks") { specBlocks = value != 0; return kOk; }      // elemhip_process through the specialised kernels when it can
    if (key == "batch_blocks") { batchBlocks = std::max(1, std::min(1024, (int)value)); return kOk; }      // blocks per multi-block launch (1 = off)
    if (key == "debug_build_delay_ms") { debugBuildDelayMs = std::max(0, (int)value); return kOk; }   // tests: stretches the unlocked part of a plan build
    if (key == "plan_cache") { planCache = std::max(0, std::min(2, (int)value)); islandCache.clear(); islandShapeCache.clear(); return kOk; }
    if (key == "plan_relocate") { relocatePrograms = value != 0; islandShapeCache.clear(); return kOk; }   // programs of structural twins renamed instead of scheduled again
    if (key == "fuse_svf_coef") { fuseSvfCoef = (uint32_t)std::max(0, std::min(2, (int)value)); planStale = true; return kOk; }
    if (key == "solo_waves") { soloWaves = (uint32_t)std::max(0, std::min(3, (int)value)); planStale = true; return kOk; }
    if (key == "mixer_split") { const int v = (int)value; mixerSplit = (v == 2 || v == 4 || v == 8) ? (uint32_t)v : 1u; planStale = true; return kOk; }
    if (key == "stateless_rows") { statelessRows = (uint32_t)std::max(1, std::min(64, (int)value)); return kOk; }   // gridDim.y of a multi-block launch: blocks that stateless islands render side by side
    if (key == "spec_waves_per_eu") { specWavesPerEu = std::max(0, std::min(8, (int)value)); specTextCache.clear(); islandCache.clear(); islandShapeCache.clear(); planStale = true; return kOk; }
    if (key == "pipeline_copies") { pipelineCopies = std::max(1, std::min(6, (int)value)); planStale = true; return kOk; }   // next commit re-plans
#ifdef ELEMHIP_EXPERIMENTAL
    if (key == "stream_ring") { streamRing = value != 0.0; return kOk; }   // 0: measurement only, needs kernels built with ELEMHIP_STREAM_PER_BLOCK
#else
    if (key == "stream_ring") return value != 0.0 ? kOk : kInvalidPropertyValue;   // (the per-block stream slices exist in EXPERIMENTAL builds only)
#endif
    if (key == "pack_islands") { packIslands = std::max(0, std::min(16, (int)value)); planStale = true; return kOk; }   // next commit re-plans
    if (key == "prog_heap_dwords") { progHeapCap = (size_t)std::max(0.0, value); progHeap.reset(); islandCache.clear(); planStale = true; return kOk; }   // 0: sized by the engine
    if (key == "pack_roots") { packRoots = value != 0; planStale = true; return kOk; }   // islands of different active roots may share a workgroup (C4: a root per render job)
    if (key == "pack_max") { packMax = std::max(1, std::min(16, (int)value)); planStale = true; return kOk; }
    if (key == "cu_count") { cuCount = std::max(1, (int)value); planStale = true; return kOk; }      // (dry handles / tests: the CU count the auto mode plans for)
    if (key == "chain_lds_out") { chainLdsOut = value != 0.0; planStale = true; return kOk; }
    if (key == "merge_phases") { mergePhases = value != 0.0; planStale = true; return kOk; }   // next commit re-plans
    if (key == "specialize") { specialize = std::max(0, std::min(2, (int)value)); planStale = true; return kOk; }   // next commit re-plans
    if (key == "profile_launches") {
        // 1: a HIP event pair around every launch of every launch set; N > 1: around those of every N-th set only (r06: an event record
        // costs ~4 us of stream time — 8.5 of an 80 us C3 set; the per-set mean is over the sampled sets, still inside the timed region)
        profileLaunches = value != 0.0;
        profileEvery = (uint32_t)std::max(1.0, std::min(1024.0, value));
        if (profileLaunches) { profMs.clear(); profSets = 0; profBlocks = 0; profSetCounter = 0; }
        return kOk;
    }
    if (key == "max_shape_launches") { maxShapeLaunches = std::max(1, std::min(64, (int)value)); dropGraphs(); return kOk; }
    if (key == "spec_lonely_blocks") { lonelyBlocks = std::max(0, (int)value); return kOk; }   // background mode: a one-off shape is queued for compilation once its
    if (key == "spec_lonely_ms") { lonelyMs = std::max(0, (int)value); return kOk; }           // plan has rendered this many blocks and been current this long
    // run-time compiler tunings (PROCESS-wide; bit-identical samples by construction, island_ops.inc): later plans compile with them
    if (key == "biquad_form") { if (!Jit::get().setTuning("ELEMHIP_BIQUAD_FORM", (int)value)) return kInvalidPropertyValue; specTextCache.clear(); islandCache.clear(); islandShapeCache.clear(); planStale = true; return kOk; }
    if (key == "wide_chain_depth") { if (!Jit::get().setTuning("ELEMHIP_WIDE_CHAIN_DEPTH", (int)value)) return kInvalidPropertyValue; specTextCache.clear(); islandCache.clear(); islandShapeCache.clear(); planStale = true; return kOk; }
    if (key == "jit_cache_entries") { Jit::get().setEntryCap((uint32_t)std::max(0.0, value)); return kOk; }   // PROCESS-wide: compiled shapes kept in memory (0: default 256)
    if (key == "time_batch") { timeBatch = std::max(1, std::min(256, (int)value)); return kOk; }
    if (key == "graph_blocks") { graphBlocks = std::max(1, (int)value); dropGraphs(); return kOk; }
    return kInvalidPropertyValue;
}

// ---- block rendering ----------------------------------------------------------------------------------
int Engine::flushPending() {
    residentStop();
    if (nextRec > recCapacity) {   // grow the record arena (device idle: we hold `mu` and sync every call)
        uint32_t cap = recCapacity;
        while (cap < nextRec) cap *= 2;
        uint32_t* nr = nullptr;
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipMalloc(&nr, (size_t)cap * kRecDwords * 4));
        HIP_OK(hipMemsetAsync(nr, 0, (size_t)cap * kRecDwords * 4, stream));
        HIP_OK(hipMemcpyAsync(nr, dRecs, (size_t)recCapacity * kRecDwords * 4, hipMemcpyDeviceToDevice, stream));
        HIP_OK(hipStreamSynchronize(stream));
        (void)hipFree(dRecs);
        dRecs = nr; recCapacity = cap;
        dropGraphs();
    }
    if (!freshRecs.empty()) {
        std::sort(freshRecs.begin(), freshRecs.end());
        size_t i = 0;
        while (i < freshRecs.size()) {
            size_t j = i + 1;
            while (j < freshRecs.size() && freshRecs[j] == freshRecs[j - 1] + 1) ++j;
            const uint32_t first = freshRecs[i];
            const size_t count = j - i;
            // through the pinned staging area the patches use (`shadow` is pageable and changes under the copy otherwise): a new
            // voice's records reach the device without waiting for the blocks that are still rendering
            constexpr size_t slotsPerRec = (kRecDwords * 4 + sizeof(Patch) - 1) / sizeof(Patch);
            size_t done = 0;
            while (done < count) {
                if (patchCursor + slotsPerRec > patchCap) { HIP_WARN(hipStreamSynchronize(stream)); patchCursor = 0; }
                const size_t fit = std::min(count - done, (patchCap - patchCursor) / slotsPerRec);
                std::memcpy(hPatches + patchCursor, shadow.data() + ((size_t)first + done) * kRecDwords, fit * kRecDwords * 4);
                HIP_WARN(hipMemcpyAsync(dRecs + ((size_t)first + done) * kRecDwords, hPatches + patchCursor, fit * kRecDwords * 4, hipMemcpyHostToDevice, stream));
                patchCursor += fit * slotsPerRec;
                done += fit;
            }
            i = j;
        }
        for (uint32_t r : freshRecs) freshFlag[r] = 0;
        freshRecs.clear();
    }
    if (deviceClockBehind && !nextSetDirect) {      // (enqueueBatch: direct-I/O convolver sets leave the device's sample clock to be caught up here)
        const uint64_t t = (uint64_t)hGlobals.sampleTime;
        patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, sampleTime) / 4), (uint32_t)(t & 0xFFFFFFFFu), 0u});
        patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, sampleTime) / 4 + 1), (uint32_t)(t >> 32), 0u});
        deviceClockBehind = false;
    }
    size_t off = 0;
    while (off < patches.size()) {
        // the patch kernel reads the pinned staging area when it RUNS: hand every launch its own
        // stretch of it and only rewind once the stream has drained
        if (patchCursor >= patchCap) { HIP_WARN(hipStreamSynchronize(stream)); patchCursor = 0; }
        const size_t cnt = std::min<size_t>(patchCap - patchCursor, patches.size() - off);
        std::memcpy(hPatches + patchCursor, patches.data() + off, cnt * sizeof(Patch));
        launch_patches(stream, hPatches + patchCursor, (uint32_t)cnt, dRecs, reinterpret_cast<uint32_t*>(dGlobals));
        debugSync("patches", (unsigned)cnt);
        patchCursor += cnt;
        off += cnt;
    }
    patches.clear();
    // new channel records of multi-output nodes that are already rendering: everything but the buffer slots, AFTER the
    // patches (a pending-buffer flag the host has just queued for channel 0 reaches the new channel with this copy)
    for (auto& cl : recClones)
        HIP_WARN(hipMemcpyAsync(dRecs + (size_t)cl.second * kRecDwords + rec::P3, dRecs + (size_t)cl.first * kRecDwords + rec::P3,
                                (kRecDwords - rec::P3) * 4, hipMemcpyDeviceToDevice, stream));
    recClones.clear();
    return kOk;
}

int Engine::setGlobalsFor(size_t nIn, size_t nOut, size_t n, int64_t sampleTime) {
    auto patchG = [&](size_t byteOff, uint32_t v) { patches.push_back(Patch{2u, (uint32_t)(byteOff / 4), v, 0u}); };
    if (hGlobals.numSamples != (uint32_t)n) { hGlobals.numSamples = (uint32_t)n; patchG(offsetof(Globals, numSamples), (uint32_t)n); }
    if (hGlobals.numIn != (uint32_t)nIn) { hGlobals.numIn = (uint32_t)nIn; patchG(offsetof(Globals, numIn), (uint32_t)nIn); }
    if (hGlobals.numOut != (uint32_t)nOut) { hGlobals.numOut = (uint32_t)nOut; patchG(offsetof(Globals, numOut), (uint32_t)nOut); }
    if (hGlobals.sampleTime != sampleTime) {
        hGlobals.sampleTime = sampleTime;
        patchG(offsetof(Globals, sampleTime), (uint32_t)((uint64_t)sampleTime & 0xFFFFFFFFu));
        patchG(offsetof(Globals, sampleTime) + 4, (uint32_t)((uint64_t)sampleTime >> 32));
    }
    return kOk;
}

void Engine::setInRing(const float* ring, uint32_t blocks) {
    const uint64_t v = (uint64_t)(uintptr_t)ring;
    if (hGlobals.inRing != v) {
        hGlobals.inRing = v;
        patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, inRing) / 4), (uint32_t)(v & 0xFFFFFFFFu), 0u});
        patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, inRing) / 4 + 1), (uint32_t)(v >> 32), 0u});
    }
    if (hGlobals.inBlocks != blocks) {
        hGlobals.inBlocks = blocks;
        patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, inBlocks) / 4), blocks, 0u});
    }
}

int Engine::swapInPending() {   // Runtime.h:277-285: newest sequence wins
    if (pending) {
        if (current && !dry) retiredPlans.push_back(std::move(current));   // its kernels may still be queued (host path): freed after the next synchronize
        current = pending;
        pending.reset();
        current->blocksAtAdoption = st.blocksRendered;
        current->adopted = std::chrono::steady_clock::now();
        st.numIslands = (uint32_t)current->islands.size();
        st.numLevels = (uint32_t)current->levelOffsets.size() - 1;
        st.numTasks = current->numTasks;
        st.numNodesInPlan = (uint32_t)current->nodeIds.size();
        st.maxLdsBytes = current->maxLdsBytes;
        st.numHbmBuffers = current->numHbmBuffers;
    }
    if (!current) return kOk;
    int rc = ensureHbm(arenaBuffers(*current, 1));
    if (rc != kOk) return rc;
    if (current->maxLdsBytes > maxLdsConfigured) {
        HIP_OK(configure_kernels(current->maxLdsBytes));
        maxLdsConfigured = current->maxLdsBytes;
    }
    return kOk;
}

void Engine::enqueueBlock(const Plan& p, float* outRing) {
    if (!outRing) outRing = dOutRing;
    const size_t L = p.levelOffsets.size() - 1;
    for (size_t l = 0; l < L; ++l) {
        const uint32_t b = p.levelOffsets[l], e = p.levelOffsets[l + 1];
        if (e > b) { launch_level(stream, p.view, dRecs, dHbm, dGlobals, dLcg, b, e - b, p.levelLdsBytes[l]); islandBlocksInterp += e - b; }
        debugSync("block: interpreter level", (unsigned)l, e - b);
        const uint32_t cb = p.convLevelOffsets[l], ce = p.convLevelOffsets[l + 1];
        if (ce > cb) launch_convolve(stream, p.view, dRecs, dHbm, dGlobals, cb, ce - cb);
        if (!p.hosts.empty()) (void)renderHostNodes(p, l);
    }
    launch_epilogue(stream, p.view, dRecs, dHbm, dGlobals, outRing, armFlag, armValue);
    if (armFlag) flagArmed = true;
    debugSync("block: epilogue");
}

// Call-out nodes of launch level `l` (GraphNode::process on the CPU, GraphNode.h:72): drain the stream, bring each node's
// input buffers to the host, run the user's process(), put its output block back into the arena. Slow by construction
// (two PCIe round trips and a pipeline drain per node and block) — the price of keeping custom CPU nodes usable.
int Engine::renderHostNodes(const Plan& p, size_t l) {
    bool any = false;
    for (const Plan::HostDesc& h : p.hosts) if (h.level == (uint32_t)l) { any = true; break; }
    if (!any) return kOk;
    HIP_OK(hipStreamSynchronize(stream));
    const size_t bs = (size_t)blockSize, n = hGlobals.numSamples, nInHost = hGlobals.numIn;
    for (const Plan::HostDesc& h : p.hosts) {
        if (h.level != (uint32_t)l) continue;
        auto nit = nodes.find(h.nodeId), rit = nodes.find(h.rootId);
        if (nit == nodes.end() || rit == nodes.end() || !nit->second.hostVt) continue;
        const Node& r = rit->second;
        const bool on = r.target > 0.5f, settled = std::fabs(r.target - r.gain) <= 1e-6f;
        if (!((on || !settled) && r.channel >= 0 && (uint32_t)r.channel < hGlobals.numOut)) continue;   // GraphRenderSequence.h:214-219
        const size_t k = h.leaf ? nInHost : h.inputs.size();
        hostIn.assign(std::max<size_t>(k, 1) * bs, 0.0f);
        hostOut.assign(bs, 0.0f);
        for (size_t j = 0; j < k; ++j) {
            float* dst = hostIn.data() + j * bs;
            if (h.leaf) { HIP_OK(hipMemcpyAsync(dst, dHbm + j * bs, n * sizeof(float), hipMemcpyDeviceToHost, stream)); continue; }
            const Plan::HostDesc::In& in = h.inputs[j];
            if (in.kind == 1) HIP_OK(hipMemcpyAsync(dst, dHbm + (size_t)in.idx * bs, n * sizeof(float), hipMemcpyDeviceToHost, stream));
            else if (in.kind == 2) { float v; std::memcpy(&v, &shadow[(size_t)in.idx * kRecDwords + rec::P0], 4); std::fill(dst, dst + n, v); }
            else if (in.kind == 3) {
                const uint32_t ch = shadow[(size_t)in.idx * kRecDwords + rec::P0];
                if (ch < nInHost) HIP_OK(hipMemcpyAsync(dst, dHbm + (size_t)ch * bs, n * sizeof(float), hipMemcpyDeviceToHost, stream));
            }
        }
        HIP_OK(hipStreamSynchronize(stream));
        std::vector<const float*> ptrs(std::max<size_t>(k, 1));
        for (size_t j = 0; j < k; ++j) ptrs[j] = hostIn.data() + j * bs;
        const Node& hn = nit->second;
        hn.hostVt->process(hn.hostInst, ptrs.data(), k, hostOut.data(), n, curBlockTime, h.active ? 1 : 0, hn.hostVt->user);
        HIP_OK(hipMemcpyAsync(dHbm + (size_t)h.outHbm * bs, hostOut.data(), n * sizeof(float), hipMemcpyHostToDevice, stream));
        HIP_OK(hipStreamSynchronize(stream));
    }
    return kOk;
}

// Host mirror of what the epilogue kernel does to each root's fade (GainFade.h:56-72), so that
// activateRoots()/gc() can evaluate stillRunning() without a device read-back.
void Engine::mirrorRootFades(const Plan& p, uint32_t n, uint32_t nOut, uint32_t nIn) {
    for (int32_t id : p.rootIds) {
        auto it = nodes.find(id);
        if (it == nodes.end()) continue;
        Node& r = it->second;
        const bool on = r.target > 0.5f;
        const bool settled = std::fabs(r.target - r.gain) <= 1e-6f;
        if (!((on || !settled) && r.channel >= 0 && (uint32_t)r.channel < nOut)) continue;
        if (r.gain != r.target && (!r.inlets.empty() || nIn > 0))
            r.gain = clampf(r.gain + r.step * (float)(int)n, 0.0f, 1.0f);
    }
}

// A host block longer than the engine's: slice `off / blockSize` of it renders the frames [off, off + blockSize) — and a tap's delay
// is the HOST's block (TapOutNode::promoteTapBuffers copies numSamples frames of its delay buffer to the shared one, TapInNode copies
// numSamples frames back out: Feedback.h:88-109, 40-54), so the slice reads and promotes ITS stretch of the shared buffers: the tap
// records' shared-buffer pointers move with the slice (parameter patches, applied in front of the slice's kernels).
void Engine::setTapSlice(size_t off) {
    if (off == tapSliceOff) return;
    tapSliceOff = off;
    if (dry) return;
    for (int32_t id : tapNodeIds) {
        auto it = nodes.find(id);
        if (it == nodes.end() || !it->second.res || !it->second.res->dev.ptr) continue;
        writeParamPtr(it->second, rec::TAP_SHARED, reinterpret_cast<const float*>(it->second.res->dev.ptr) + off);
    }
}

int Engine::process(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n, int64_t sampleTime) {
    if (hostBlockSize == blockSize) return processSlice(in, nIn, out, nOut, n, sampleTime);
    if (n > (size_t)hostBlockSize) return kBlockTooLarge;
    if (nIn > kMaxHostIn || nOut > kMaxOutBus) return kTooManyChannels;
    // a host block longer than the engine's: slice by slice (each slice is a block of its own to the kernels), the render lock held
    // and the newest sequence adopted ONCE for the whole host block — a commit, gc or event relay from another thread lands between
    // two host blocks, never inside one
    std::vector<const float*> ip(nIn);
    std::vector<float*> op(nOut);
    std::lock_guard<std::mutex> lock(mu);
    for (size_t off = 0; off < n || off == 0; off += (size_t)blockSize) {
        for (size_t c = 0; c < nIn; ++c) ip[c] = in[c] + off;
        for (size_t c = 0; c < nOut; ++c) op[c] = out[c] + off;
        if (!tapNodeIds.empty()) setTapSlice(off);
        const int rc = processSliceLocked(ip.data(), nIn, op.data(), nOut, std::min((size_t)blockSize, n - off), sampleTime + (int64_t)off, off == 0);
        if (rc != kOk) return rc;
    }
    noteHostBlockEnd();
    return kOk;
}

int Engine::processSlice(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n, int64_t sampleTime) {
    std::lock_guard<std::mutex> lock(mu);
    return processSliceLocked(in, nIn, out, nOut, n, sampleTime, true);
}

// (`mu` held.) `adopt`: take the newest render sequence — once per HOST block (Runtime.h:277-285): the later slices of a host block
// longer than the engine's render the sequence its first slice adopted, under the same hold of the lock.
int Engine::processSliceLocked(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n, int64_t sampleTime, bool adopt) {
    if (dry) return kNoDevice;
    if (hipSetDevice(device) != hipSuccess) return kHipError;
    if (n > (size_t)blockSize) return kBlockTooLarge;
    if (nIn > kMaxHostIn || nOut > kMaxOutBus) return kTooManyChannels;
    if (n != conv::kBlock) convAligned = false;   // a convolver's input block may now be partly filled at a call boundary
    if (residentLive && (pending || !residentOpt)) residentStop();
    int rc = adopt ? swapInPending() : kOk;
    if (rc != kOk) return rc;
    if (!current) return kOk;   // no render sequence yet: outputs untouched (Runtime.h:287-289)
    Plan& p = *current;
    if (p.packedRootChannels > 0 && nOut < (size_t)p.packedRootChannels) return kInvalidPropertyValue;   // (`pack_roots`, plan.cpp)

    if (hGlobals.ringSlots != 1 || hGlobals.blockSlot != 0) {
        hGlobals.ringSlots = 1; hGlobals.blockSlot = 0;
        patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, ringSlots) / 4), 1u, 0u});
        patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, blockSlot) / 4), 0u, 0u});
    }
    setGlobalsFor(nIn, nOut, n, sampleTime);
    setInRing(nullptr, 0);
    rc = ensureOutRing(std::max<size_t>(nOut, 1) * blockSize);
    if (rc != kOk) return rc;
    // option "resident": a run of plain blocks (nothing to flush, every fade settled, the same plan and channel counts) is handed
    // to the resident kernel through mapped host memory — no launch, no stream synchronise
    {
        const bool plain = residentOpt && n == (size_t)blockSize && residentEligible(p, nIn, nOut);
        if (residentLive && (!plain || residentPlan != &p || residentNIn != nIn || residentNOut != nOut)) residentStop();
        if (!plain) residentStreak = 0;
        else if (!residentLive && ++residentStreak > residentAfter) {
            rc = residentStart(p, nIn, nOut);
            if (rc != kOk) return rc;
        }
        if (residentLive) {
            rc = residentBlock(in, nIn, out, nOut, n);
            if (rc != kResidentGone) return rc;
        }
    }
    // A whole block of a settled sequence whose island shapes are all compiled goes through the specialised kernels as a
    // launch set of one (stages pipelined inside the workgroup, no interpreter image); everything else block by block.
    const bool specLevels = n == (size_t)blockSize && specBlockOk(p);
    const bool specBlock = specLevels && batchEligible(p, nOut, true);
    const bool specFade = specLevels && !specBlock;      // root fades running: same level launches, the per-block epilogue behind them
    if (specLevels) {
        rc = ensureHbm(arenaBuffers(p, 1));
        if (rc != kOk) return rc;
    }

    if (nIn > 0) {
        const size_t floats = nIn * (size_t)blockSize;
        if (floats > hInFloats) {
            if (hIn) (void)hipHostFree(hIn);
            HIP_OK(hipHostMalloc((void**)&hIn, floats * sizeof(float), hipHostMallocDefault));
            hInFloats = floats;
        }
        for (size_t c = 0; c < nIn; ++c) {
            std::memcpy(hIn + c * blockSize, in[c], n * sizeof(float));
            if (n < (size_t)blockSize) std::memset(hIn + c * blockSize + n, 0, (blockSize - n) * sizeof(float));
        }
        HIP_OK(hipMemcpyAsync(dHbm, hIn, floats * sizeof(float), hipMemcpyHostToDevice, stream));
    }
    rc = flushPending();
    if (rc != kOk) return rc;
    curBlockTime = sampleTime;
    // the epilogue writes the output block straight into the pinned host buffer (mapped into the device's address space): one
    // launch and one PCIe write burst less than rendering into HBM and copying back
    float* outDev = nullptr;
    if (nOut > 0) {
        const size_t floats = std::max<size_t>(nOut, 1) * (size_t)blockSize;
        if (floats > hOutFloats) {
            HIP_OK(hipStreamSynchronize(stream));
            if (hOut) (void)hipHostFree(hOut);
            hOut = nullptr; hOutFloats = 0; hOutDev = nullptr;
            // mapped AND coherent, said out loud: with `sync_poll` the host reads this block when the epilogue's word arrives, before the
            // kernel has ended — no kernel-end release stands behind the samples, only the epilogue's own system-scope fence
            HIP_OK(hipHostMalloc((void**)&hOut, floats * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
            hOutFloats = floats;
            if (hipHostGetDevicePointer((void**)&hOutDev, hOut, 0) != hipSuccess) hOutDev = nullptr;
        }
        outDev = hostOutDirect ? hOutDev : nullptr;
    }
    // the call ends when the epilogue's word arrives (sync_poll): only when the epilogue writes the host's block itself, nothing is
    // being profiled and no hipGraph replays the launches (a captured launch would publish a stale value)
    const bool graphPath = specBlock && specBlockGraph && useGraph && !profileLaunches && !debugSyncOn();
    flagArmed = false; armFlag = nullptr;
    if (syncPoll && outDev && !graphPath && !profileLaunches && !debugSyncOn() && p.hosts.empty()) {
        if (!hDone) {
            HIP_OK(hipHostMalloc((void**)&hDone, 64, hipHostMallocMapped | hipHostMallocCoherent));
            *hDone = 0;
            HIP_OK(hipHostGetDevicePointer((void**)&dDone, hDone, 0));
        }
        armFlag = dDone; armValue = ++doneSeq;
    }
    if (specBlock && specBlockGraph && useGraph && !profileLaunches && !debugSyncOn()) {
        // the launch set of one (level launches, side-stream forks and joins, batch epilogue) replayed from a captured graph
        float* const target = outDev ? outDev : dOutRing;
        if (!p.specGraphExec || p.specGraphOut != target || p.specGraphNumOut != (uint32_t)nOut) {   // (which launches are left out depends on the output count)
            if (p.specGraphExec) { (void)hipGraphExecDestroy(p.specGraphExec); p.specGraphExec = nullptr; }
            hipGraph_t graph = nullptr;
            HIP_OK(hipStreamSynchronize(stream));
            const uint64_t before = st.specLaunches;
            HIP_OK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            enqueueBatch(p, 1u, outDev);
            HIP_OK(hipStreamEndCapture(stream, &graph));
            HIP_OK(hipGraphInstantiate(&p.specGraphExec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
            p.specGraphOut = target; p.specGraphNumOut = (uint32_t)nOut; p.specGraphLaunches = (uint32_t)(st.specLaunches - before);
            st.specLaunches = before;
            st.graphCaptures++;
        }
        HIP_OK(hipGraphLaunch(p.specGraphExec, stream));
        st.specLaunches += p.specGraphLaunches; st.graphReplays++;
    } else if (specBlock) enqueueBatch(p, 1u, outDev);
    else if (specFade) enqueueSpecBlock(p, outDev);
    else { fixConvOverlaps(p); enqueueBlock(p, outDev); }
    if (nOut > 0 && !outDev) HIP_OK(hipMemcpyAsync(hOut, dOutRing, nOut * (size_t)blockSize * sizeof(float), hipMemcpyDeviceToHost, stream));
    armFlag = nullptr;
    bool arrived = false;
    if (flagArmed) {
        // spin on the epilogue's word (2 ms at most: a graph that takes longer gains nothing from it; a kernel that faulted never
        // publishes — the synchronise below reports it)
        const uint32_t want = armValue;
        uint32_t spins = 0;
        std::chrono::steady_clock::time_point t0;
        for (;;) {
            if (__atomic_load_n(hDone, __ATOMIC_ACQUIRE) == want) { arrived = true; break; }
            __builtin_ia32_pause();
            if ((++spins & 0x3FFu) == 0u) {
                const auto now = std::chrono::steady_clock::now();
                if (spins == 0x400u) t0 = now;
                else if (now - t0 > std::chrono::milliseconds(2)) break;
            }
        }
        flagArmed = false;
        syncPolls++;
        if (!arrived) syncPollFallbacks++;
    }
    // (every 256th call still synchronises: the runtime retires its launch bookkeeping there; so does a call with plans / buffers
    //  waiting to be released — freeDeferred's contract is a REAL synchronise of the stream, not the polled word)
    bool synced = false;
    if (!arrived || !deferredFree.empty() || !retiredPlans.empty() || (++unsyncedCalls & 255u) == 0u) {
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipGetLastError());
        synced = true;
    }
    if (profUsed) profCollect();
    for (size_t c = 0; c < nOut; ++c) std::memcpy(out[c], hOut + c * blockSize, n * sizeof(float));
    mirrorRootFades(p, (uint32_t)n, (uint32_t)nOut, (uint32_t)nIn);
    hGlobals.sampleTime += (int64_t)n;
    st.blocksRendered++;
    promoteDeferredShapes();
    // the word arrived: the armed epilogue is the last work of this call on `stream`, which is in order and joined every side stream
    // in front of it — the patch uploads of this call have been consumed, so the staging cursor may start over; nothing is freed
    if (synced) freeDeferred(); else patchCursor = 0;
    return kOk;
}

// ---- option "resident" (resident.hip) -------------------------------------------------------------------------------------------
bool Engine::residentEligible(const Plan& p, size_t nIn, size_t nOut) const {
    if (!p.convs.empty() || !p.hosts.empty() || profileLaunches || debugSyncOn() || hGlobals.trace) return false;
    if (!patches.empty() || !freshRecs.empty() || !recClones.empty() || deviceClockBehind) return false;   // (something flushPending still has to bring to the device)
    if (p.view.numRoots > kResidentMaxRoots || p.levelOffsets.size() < 2 || p.levelOffsets.size() - 1 > kResidentMaxLevels) return false;
    if (p.levelOffsets.back() == 0u) return false;
    return batchEligible(p, nOut, true);      // every running root's fade settled: the epilogue has no per-root state to advance
}

int Engine::residentStart(const Plan& p, size_t nIn, size_t nOut) {
    if (!hResident) {
        HIP_OK(hipHostMalloc((void**)&hResident, sizeof(ResidentCtl), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_OK(hipHostGetDevicePointer((void**)&dResidentCtl, hResident, 0));
        HIP_OK(hipMalloc((void**)&dResidentSync, 64));
    }
    const size_t inF = std::max<size_t>(nIn, 1) * (size_t)blockSize, outF = std::max<size_t>(nOut, 1) * (size_t)blockSize;
    if (inF > resInFloats) {
        if (hResIn) (void)hipHostFree(hResIn);
        hResIn = nullptr; resInFloats = 0;
        HIP_OK(hipHostMalloc((void**)&hResIn, inF * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_OK(hipHostGetDevicePointer((void**)&dResIn, hResIn, 0));
        resInFloats = inF;
    }
    if (outF > resOutFloats) {
        if (hResOut) (void)hipHostFree(hResOut);
        hResOut = nullptr; resOutFloats = 0;
        HIP_OK(hipHostMalloc((void**)&hResOut, outF * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_OK(hipHostGetDevicePointer((void**)&dResOut, hResOut, 0));
        resOutFloats = outF;
    }
    ResidentLevels lv{};
    lv.count = (uint32_t)p.levelOffsets.size() - 1;
    uint32_t widest = 1;
    for (uint32_t l = 0; l <= lv.count; ++l) lv.offset[l] = p.levelOffsets[l];
    for (uint32_t l = 0; l < lv.count; ++l) widest = std::max(widest, lv.offset[l + 1] - lv.offset[l]);
    // one workgroup per compute unit at most: every workgroup has to be ON the device for the barriers between the levels to complete
    const uint32_t groups = std::min<uint32_t>(widest, (uint32_t)std::max(1, cuCount));
    if (p.maxLdsBytes > residentLdsConfigured) {
        HIP_OK(configure_resident(p.maxLdsBytes));
        residentLdsConfigured = p.maxLdsBytes;
    }
    std::memset(hResident, 0, sizeof(ResidentCtl));
    HIP_OK(hipMemsetAsync(dResidentSync, 0, 64, stream));
    const uint64_t idleTicks = (uint64_t)residentIdleUs * 100ull, hangTicks = 5ull * 100000000ull;   // s_memrealtime: 100 MHz
    HIP_OK(launch_resident(stream, p.view, dRecs, dHbm, dGlobals, dLcg, lv, groups, std::max<uint32_t>(p.maxLdsBytes, 1024u), dResidentCtl, dResIn, dResOut,
                           dResidentSync, idleTicks, hangTicks));
    residentTicksBody = residentTicksEpilogue = 0;
    residentLive = true; residentSeq = 0; residentPlan = &p; residentNIn = nIn; residentNOut = nOut;
    st.residentLaunches++;
    return kOk;
}

void Engine::residentStop() {
    if (!residentLive) return;
    __atomic_store_n(&hResident->seq, kResidentQuit, __ATOMIC_RELEASE);
    HIP_WARN(hipStreamSynchronize(stream));
    if (__atomic_load_n(&hResident->exited, __ATOMIC_ACQUIRE) == 2u)
        std::fprintf(stderr, "[elemhip] resident kernel: a device-wide barrier timed out; the block it was rendering is lost\n");
    residentLive = false; residentStreak = 0; residentPlan = nullptr;
    static const bool trace = std::getenv("ELEMHIP_RESIDENT_TRACE") != nullptr;
    if (trace && residentSeq)
        std::fprintf(stderr, "[elemhip] resident kernel left after %u blocks: levels %.2f us, epilogue %.2f us per block (device clock, from the block number's arrival)\n",
                     residentSeq, 0.01 * (double)residentTicksBody / residentSeq, 0.01 * (double)residentTicksEpilogue / residentSeq);
}

int Engine::residentBlock(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n) {
    for (size_t c = 0; c < nIn; ++c) std::memcpy(hResIn + c * blockSize, in[c], n * sizeof(float));
    const uint32_t seq = ++residentSeq;
    __atomic_store_n(&hResident->seq, seq, __ATOMIC_RELEASE);
    uint32_t spins = 0;
    std::chrono::steady_clock::time_point t0;
    for (;;) {
        if (__atomic_load_n(&hResident->done, __ATOMIC_ACQUIRE) == seq) break;
        if (__atomic_load_n(&hResident->exited, __ATOMIC_ACQUIRE) != 0u) {
            // it left: by itself (idle) before it saw this block, or on a barrier time-out
            const bool rendered = __atomic_load_n(&hResident->done, __ATOMIC_ACQUIRE) == seq;
            const uint32_t code = hResident->exited;
            HIP_WARN(hipStreamSynchronize(stream));
            residentLive = false; residentStreak = 0; residentPlan = nullptr;
            if (rendered) break;
            if (code == 2u) return kHipError;
            return kResidentGone;
        }
        __builtin_ia32_pause();
        if ((++spins & 0xFFFu) == 0u) {
            const auto now = std::chrono::steady_clock::now();
            if (spins == 0x1000u) t0 = now;
            else if (now - t0 > std::chrono::seconds(10)) {      // (the kernel's own time-outs are 5 s: this is a kernel that never started)
                __atomic_store_n(&hResident->seq, kResidentQuit, __ATOMIC_RELEASE);
                HIP_WARN(hipStreamSynchronize(stream));
                residentLive = false; residentStreak = 0; residentPlan = nullptr;
                return kHipError;
            }
        }
    }
    for (size_t c = 0; c < nOut; ++c) std::memcpy(out[c], hResOut + c * blockSize, n * sizeof(float));
    residentTicksBody += hResident->ticksBody; residentTicksEpilogue += hResident->ticksEpilogue;
    curBlockTime = hGlobals.sampleTime;
    hGlobals.sampleTime += (int64_t)n;
    st.blocksRendered++; st.residentBlocks++;
    islandBlocksInterp += current->levelOffsets.back();
    return kOk;
}

int Engine::timeLaunches(size_t nOut, size_t numBlocks, float* msOut, size_t cap) {
    RenderGuard lock(*this);
    if (dry) return -kNoDevice;
    if (hipSetDevice(device) != hipSuccess) return -kHipError;
    int rc = swapInPending();
    if (rc != kOk) return -rc;
    if (!current) return 0;
    const Plan& p = *current;
    const size_t L = p.levelOffsets.size() - 1;
    if (cap < L + 1) return -kInvalidPropertyValue;
    rc = ensureOutRing(std::max<size_t>(nOut, 1) * blockSize);
    if (rc != kOk) return -rc;
    if (hGlobals.ringSlots != 1 || hGlobals.blockSlot != 0) {
        hGlobals.ringSlots = 1; hGlobals.blockSlot = 0;
        patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, ringSlots) / 4), 1u, 0u});
        patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, blockSlot) / 4), 0u, 0u});
    }
    setGlobalsFor(0, nOut, (size_t)blockSize, hGlobals.sampleTime);
    setInRing(nullptr, 0);
    rc = flushPending();
    if (rc != kOk) return -rc;
    std::vector<hipEvent_t> ev(2 * (L + 2));   // + one empty pair: the cost of the event pair itself
    for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) return -kHipError;
    std::vector<double> acc(L + 2, 0.0);
    // timeBatch > 1: time the multi-block launches elemhip_process_blocks issues (msOut = per LAUNCH of `batch` blocks)
    const uint32_t batch = (timeBatch > 1 && p.convs.empty() && batchEligible(p, nOut)) ? (uint32_t)std::min<size_t>((size_t)timeBatch, maxSetBlocks(p)) : 1u;
    const uint32_t arenaFloats = batch > 1 ? p.numHbmBuffers * (uint32_t)blockSize : 0u;
    if (batch > 1) {
        rc = ensureHbm(arenaBuffers(p, batch)); if (rc != kOk) return -rc;
        rc = ensureOutRing(std::max<size_t>(nOut, 1) * blockSize * batch); if (rc != kOk) return -rc;
    }
    lastTimeBatch = batch;
    fixConvOverlaps(p);
    for (size_t b = 0; b < numBlocks; ++b) {
        for (size_t l = 0; l < L; ++l) {
            const uint32_t lb = p.levelOffsets[l], le = p.levelOffsets[l + 1];
            (void)hipEventRecord(ev[2 * l], stream);
            if (le > lb) { if (batch > 1) launchLevelBatch(p, l, batch, arenaFloats); else launch_level(stream, p.view, dRecs, dHbm, dGlobals, dLcg, lb, le - lb, p.levelLdsBytes[l], batch, arenaFloats, statelessRows); }
            if (p.convLevelOffsets[l + 1] > p.convLevelOffsets[l])
                launch_convolve(stream, p.view, dRecs, dHbm, dGlobals, p.convLevelOffsets[l], p.convLevelOffsets[l + 1] - p.convLevelOffsets[l]);
            (void)hipEventRecord(ev[2 * l + 1], stream);
        }
        (void)hipEventRecord(ev[2 * L], stream);
        if (batch > 1) launch_epilogue_batch(stream, p.view, dRecs, dHbm, dGlobals, dOutRing, batch, arenaFloats);
        else launch_epilogue(stream, p.view, dRecs, dHbm, dGlobals, dOutRing);
        (void)hipEventRecord(ev[2 * L + 1], stream);
        (void)hipEventRecord(ev[2 * L + 2], stream);
        (void)hipEventRecord(ev[2 * L + 3], stream);
        if (hipStreamSynchronize(stream) != hipSuccess) return -kHipError;
        for (size_t l = 0; l <= L + 1; ++l) { float ms = 0; (void)hipEventElapsedTime(&ms, ev[2 * l], ev[2 * l + 1]); acc[l] += ms; }
        for (uint32_t k = 0; k < batch; ++k) mirrorRootFades(p, (uint32_t)blockSize, (uint32_t)nOut, 0);
        hGlobals.sampleTime += (int64_t)blockSize * batch;
        st.blocksRendered += batch;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    // an empty event pair measures the marker-to-marker cost that every timed launch also pays
    const double empty = acc[L + 1] / (double)std::max<size_t>(numBlocks, 1);
    for (size_t l = 0; l <= L; ++l) msOut[l] = (float)std::max(0.0, acc[l] / (double)std::max<size_t>(numBlocks, 1) - empty);
    if (cap > L + 1) msOut[L + 1] = (float)empty;
    if (cap > L + 2) msOut[L + 2] = (float)batch;   // blocks per timed launch
    return (int)(L + 1);
}

int Engine::traceLevel(size_t nOut, uint32_t level, unsigned long long* out, size_t cap) {
    RenderGuard lock(*this);
    if (dry) return kNoDevice;
    if (hipSetDevice(device) != hipSuccess) return kHipError;
    int rc = swapInPending();
    if (rc != kOk) return rc;
    if (!current || cap < kWaves * 192) return kInvalidPropertyValue;
    const Plan& p = *current;
    const size_t L = p.levelOffsets.size() - 1;
    rc = ensureOutRing(std::max<size_t>(nOut, 1) * blockSize);
    if (rc != kOk) return rc;
    unsigned long long* dTrace = nullptr;
    HIP_OK(hipMalloc(&dTrace, kWaves * 192 * 8));
    HIP_OK(hipMemset(dTrace, 0, kWaves * 192 * 8));
    if (hGlobals.ringSlots != 1 || hGlobals.blockSlot != 0) {
        hGlobals.ringSlots = 1; hGlobals.blockSlot = 0;
        patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, ringSlots) / 4), 1u, 0u});
        patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, blockSlot) / 4), 0u, 0u});
    }
    setGlobalsFor(0, nOut, (size_t)blockSize, hGlobals.sampleTime);
    setInRing(nullptr, 0);
    rc = flushPending();
    if (rc != kOk) return rc;
    const uint64_t tp = (uint64_t)reinterpret_cast<uintptr_t>(dTrace);
    const uint32_t batch = (timeBatch > 1 && p.convs.empty() && batchEligible(p, nOut)) ? (uint32_t)std::min<size_t>((size_t)timeBatch, maxSetBlocks(p)) : 1u;
    const uint32_t arenaFloats = batch > 1 ? p.numHbmBuffers * (uint32_t)blockSize : 0u;
    if (batch > 1) {
        rc = ensureHbm(arenaBuffers(p, batch)); if (rc != kOk) return rc;
        rc = ensureOutRing(std::max<size_t>(nOut, 1) * blockSize * batch); if (rc != kOk) return rc;
    }
    for (size_t l = 0; l < L; ++l) {
        const uint32_t lb = p.levelOffsets[l], le = p.levelOffsets[l + 1];
        const uint64_t v = (l == level) ? tp : 0;
        HIP_OK(hipMemcpyAsync(reinterpret_cast<char*>(dGlobals) + offsetof(Globals, trace), &v, 8, hipMemcpyHostToDevice, stream));
        if (le > lb) { if (batch > 1) launchLevelBatch(p, l, batch, arenaFloats); else launch_level(stream, p.view, dRecs, dHbm, dGlobals, dLcg, lb, le - lb, p.levelLdsBytes[l], batch, arenaFloats, statelessRows); }
    }
    const uint64_t zero = 0;
    HIP_OK(hipMemcpyAsync(reinterpret_cast<char*>(dGlobals) + offsetof(Globals, trace), &zero, 8, hipMemcpyHostToDevice, stream));
    if (batch > 1) launch_epilogue_batch(stream, p.view, dRecs, dHbm, dGlobals, dOutRing, batch, arenaFloats);
    else launch_epilogue(stream, p.view, dRecs, dHbm, dGlobals, dOutRing);
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(out, dTrace, kWaves * 192 * 8, hipMemcpyDeviceToHost));
    (void)hipFree(dTrace);
    for (uint32_t k = 0; k < batch; ++k) mirrorRootFades(p, (uint32_t)blockSize, (uint32_t)nOut, 0);
    hGlobals.sampleTime += (int64_t)blockSize * batch;
    st.blocksRendered += batch;
    return kOk;
}

// A multi-block launch carries no per-block root/tap/convolver bookkeeping: it is used only while every
// running root's fade is settled (Core.h:28-31) and the plan has neither taps nor convolvers.
// every island shape of the sequence has its kernel loaded on this device (stateless islands stay with the interpreter kernel)
bool Engine::specReady(const Plan& p) const {
    if (specialize == 0 || p.shapes.empty()) return false;
    bool any = false;
    for (const Plan::SpecShape& sh : p.shapes) {
        if (sh.entry->function(device)) any = true;
        else if (!sh.optional) return false;
    }
    return any;
}

void Engine::promoteDeferredShapes() {
    Plan* p = current.get();
    if (!p || !p->deferredShapes) return;
    if (st.blocksRendered - p->blocksAtAdoption < (uint64_t)std::max(0, lonelyBlocks)) return;
    if (std::chrono::steady_clock::now() - p->adopted < std::chrono::milliseconds(std::max(0, lonelyMs))) return;
    for (Plan::SpecShape& sh : p->shapes)
        if (sh.deferred) { Jit::get().promote(sh.entry); sh.deferred = false; }
    p->deferredShapes = 0;
}

bool Engine::anyRootRuns(const std::vector<int32_t>& rootIds, size_t nOut) const {
    for (int32_t id : rootIds) {
        auto it = nodes.find(id);
        if (it == nodes.end()) return true;          // (cannot tell: launch)
        const Node& r = it->second;
        const bool on = r.target > 0.5f, settled = std::fabs(r.target - r.gain) <= 1e-6f;
        if ((on || !settled) && r.channel >= 0 && (uint32_t)r.channel < nOut) return true;
    }
    return false;
}

// (oneBlock: a launch set of ONE — the batch epilogue promotes the taps after it like the per-block epilogue does, so tap
// pairs that do not sit in one island are no obstacle)
bool Engine::batchEligible(const Plan& p, size_t nOut, bool oneBlock) const {
    if ((!oneBlock && !p.taps.empty() && !p.tapsInSets) || !p.hosts.empty()) return false;
    // convolvers: the multi-block kernels (conv.hip) assume every node's 512-frame input block is empty at the start of a
    // launch set, i.e. that every call so far rendered whole 512-frame blocks
    if (!p.convs.empty() && !(convAligned && blockSize == (int)conv::kBlock)) return false;
    for (int32_t id : p.rootIds) {
        auto it = nodes.find(id);
        if (it == nodes.end()) return false;
        const Node& r = it->second;
        const bool on = r.target > 0.5f;
        const bool settled = std::fabs(r.target - r.gain) <= 1e-6f;
        const bool running = (on || !settled) && r.channel >= 0 && (uint32_t)r.channel < nOut;
        if (running && r.gain != r.target) return false;
    }
    return true;
}

hipEvent_t Engine::profEvent() {
    if (profUsed == profEvents.size()) { hipEvent_t e = nullptr; (void)hipEventCreate(&e); profEvents.push_back(e); }
    return profEvents[profUsed++];
}

// after a stream synchronize: fold the event pairs of this call into the per-level sums
void Engine::profCollect() {
    for (size_t k = 0; k + 1 < profUsed; k += 2) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, profEvents[k], profEvents[k + 1]) != hipSuccess) continue;
        const uint32_t slot = profSlots[k / 2];
        if (profMs.size() <= slot) profMs.resize(slot + 1, 0.0);
        profMs[slot] += ms;
    }
    profUsed = 0; profSlots.clear();
}

// debug / tests: program text and compile state of the k-th specialised shape of the newest plan
int Engine::specInfo(size_t k, std::string* source, std::string* log, int* state, uint32_t* islands) {
    std::lock_guard<std::mutex> control(ctl);
    RenderGuard lock(*this);
    const std::shared_ptr<Plan> pl = pending ? pending : current;
    if (!pl || k >= pl->shapes.size()) return -1;
    const Plan::SpecShape& sh = pl->shapes[k];
    if (source) *source = sh.entry->fullText();
    if (log) *log = sh.entry->log;
    if (state) *state = sh.entry->state.load();
    if (islands) *islands = sh.count;
    return (int)pl->shapes.size();
}

int Engine::launchProfile(double* msOut, size_t cap, uint64_t* launchSets, uint64_t* blocks) {
    RenderGuard lock(*this);
    if (launchSets) *launchSets = profSets;
    if (blocks) *blocks = profBlocks;
    const size_t n = std::min(cap, profMs.size());
    for (size_t i = 0; i < n; ++i) msOut[i] = profMs[i];
    return (int)profMs.size();
}

// One launch level of a multi-block launch. When every island shape of the level has its specialised kernel compiled
// (jit.cpp) the level runs as one launch per shape plus an interpreter launch for the islands no shape covers
// (stateless mixers and roots); until then the whole level goes through the interpreter kernel.
bool Engine::launchLevelBatch(const Plan& p, size_t l, uint32_t batch, uint32_t arenaFloats, float* epiOut) {
    const uint32_t b = p.levelOffsets[l], e = p.levelOffsets[l + 1];
    if (e <= b) return false;
    bool spec = specialize != 0 && !p.shapes.empty();
    std::vector<std::pair<hipFunction_t, const Plan::SpecShape*>> fns;   // function null: the shape is not compiled (yet)
    struct InterpRun { uint32_t begin, count; };                          // contiguous stretches of specLists the interpreter kernel renders
    std::vector<InterpRun> interpRuns;
    if (spec) {
        bool any = false;
        int specTaken = 0;
        for (const Plan::SpecShape& sh : p.shapes) {
            if (sh.level != (uint32_t)l) continue;
            // (the level's shapes are listed biggest first, plan.cpp: the first `maxShapeLaunches` compiled ones get their kernels)
            hipFunction_t fn = specTaken < maxShapeLaunches ? sh.entry->function(device) : nullptr;
            if (fn) ++specTaken;
            any = any || fn != nullptr;
            // A launch whose islands all belong to roots that do not run (a replaced root once its fade-out has settled stays in
            // the plan until the next commit) would start workgroups that return at once — and, as a second launch of its
            // level, cost a fork to a side stream and a join. The host mirrors the root fades (mirrorRootFades), and launch
            // sets are only rendered while every running root's fade is settled: what runs does not change inside a set.
            if (skipIdleLaunches && !anyRootRuns(sh.roots, hGlobals.numOut)) { st.idleLaunchesSkipped++; continue; }
            if (fn) { fns.emplace_back(fn, &sh); continue; }
            // not compiled (yet), or beyond the launch cap: joins the interpreter run that ends where its list begins, or starts one
            if (!interpRuns.empty() && interpRuns.back().begin + interpRuns.back().count == sh.listBegin) interpRuns.back().count += sh.count;
            else interpRuns.push_back({sh.listBegin, sh.count});
        }
        if (!any) spec = false;
    }
    if (!spec) { launch_level(stream, p.view, dRecs, dHbm, dGlobals, dLcg, b, e - b, p.levelLdsBytes[l], batch, arenaFloats, statelessRows); islandBlocksInterp += (uint64_t)(e - b) * batch; debugSync("set: interpreter level", (unsigned)l, batch); return false; }
    // The launches of one level are independent of each other (different islands): with more than one they go to side
    // streams forked from / joined to the engine's stream, so two shapes of 64 islands each fill 128 CUs at once
    // instead of 64 CUs twice.
    uint32_t rb = p.restOffsets[l], re = p.restOffsets[l + 1];
    if (re > rb && skipIdleLaunches && l < p.restRoots.size() && !anyRootRuns(p.restRoots[l], hGlobals.numOut)) { re = rb; st.idleLaunchesSkipped++; }
    const size_t launches = fns.size() + interpRuns.size() + (re > rb ? 1 : 0);
    if (launches == 0) return false;
    const bool fork = launches > 1;
    const bool fused = epiOut != nullptr && batch == 1u && launches == 1 && fns.size() == 1 && fns[0].first != nullptr;
    if (fork) {
        while (auxStreams.size() < launches - 1) {
            hipStream_t s2 = nullptr; hipEvent_t ev = nullptr;
            // a side stream must not share a hardware queue with the engine's stream (the runtime hands queues out round-robin
            // per priority class; in a process with many streams two shapes of C4 landed on one queue and ran back to back,
            // 12.9 -> 25 us per block): side streams alternate between the two other priority classes
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            const int prio = (auxStreams.size() % 2 == 0) ? greatest : least;
            if (least == greatest || hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, prio) != hipSuccess)
                HIP_WARN(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            HIP_WARN(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            auxStreams.push_back(s2); auxDone.push_back(ev);
        }
        if (!forkEvent) HIP_WARN(hipEventCreateWithFlags(&forkEvent, hipEventDisableTiming));
        HIP_WARN(hipEventRecord(forkEvent, stream));
    }
    size_t k = 0;
    auto streamFor = [&](size_t idx) -> hipStream_t {
        if (!fork || idx == 0) return stream;
        hipStream_t s2 = auxStreams[idx - 1];
        HIP_WARN(hipStreamWaitEvent(s2, forkEvent, 0));
        return s2;
    };
    for (const InterpRun& r : interpRuns) {   // these shapes' islands go through the interpreter kernel (their lists have the levelIslands entry format)
        PlanView pv = p.view;
        pv.levelIslands = p.dSpecLists;
        launch_level(streamFor(k++), pv, dRecs, dHbm, dGlobals, dLcg, r.begin, r.count, p.levelLdsBytes[l], batch, arenaFloats, statelessRows);
        islandBlocksInterp += (uint64_t)r.count * batch;
    }
    for (auto& f : fns) {
        hipStream_t st_ = streamFor(k++);
        PlanView pv = p.view;
        uint32_t* recs = dRecs; float* hbm = dHbm; const Globals* g = dGlobals; const uint32_t* lcg = dLcg;
        const uint32_t* list = p.dSpecLists + f.second->listBegin;
        // the stream ring sits behind the `batch` block slices of this launch set
        uint32_t bt = batch, af = arenaFloats, sb = batch * arenaFloats, ss = p.numStreamBuffers * (uint32_t)blockSize;
        uint32_t eg = fused ? f.second->count : 0u;
        float* eo = fused ? epiOut : nullptr;
        uint32_t* ef = (fused && armFlag) ? armFlag : nullptr;      // the fused tail publishes elemhip_process' completion word itself
        uint32_t ev = armValue;
        void* args[] = {&pv, &recs, &hbm, &g, &lcg, &list, &bt, &af, &sb, &ss, &eg, &eo, &ef, &ev};
        const uint32_t gy = f.second->stateless ? std::max(1u, std::min(batch, statelessRows)) : 1u;
        HIP_WARN(hipModuleLaunchKernel(f.first, f.second->count, gy, 1, kThreads, 1, 1, 0, st_, args, nullptr));
        st.specLaunches++;
        islandBlocksSpec += (uint64_t)f.second->count * batch;
        debugSync("set: specialised shape", f.second->count, batch);
    }
    if (re > rb) {
        PlanView pv = p.view;
        pv.levelIslands = p.dRestIslands;
        launch_level(streamFor(k++), pv, dRecs, dHbm, dGlobals, dLcg, rb, re - rb, p.levelLdsBytes[l], batch, arenaFloats, statelessRows);
        islandBlocksInterp += (uint64_t)(re - rb) * batch;
        debugSync("set: interpreter rest", re - rb, batch);
    }
    if (fork) {
        for (size_t i = 1; i < launches; ++i) {
            HIP_WARN(hipEventRecord(auxDone[i - 1], auxStreams[i - 1]));
            HIP_WARN(hipStreamWaitEvent(stream, auxDone[i - 1], 0));
        }
    }
    return fused;
}

// the convolve nodes of level l over a whole launch set: four launches (fft, mac, ifft, finish: conv.hip, "multi-block launches")
void Engine::launchConvolveBatch(const Plan& p, size_t l, uint32_t batch, uint32_t arenaFloats) {
    const uint32_t cb = p.convLevelOffsets[l], ce = p.convLevelOffsets[l + 1];
    uint32_t mains = 0;
    while (cb + mains < ce && (p.convWork[cb + mains] >> 16) == 0u) ++mains;   // main entries lead a level's work list
    if (!mains) return;
    // which nodes of the level have long-partition spectra (an IR can be replaced without a re-plan: looked up per launch set)
    bool anyShortPath = false;
    uint32_t stateBlocks = 1;
    for (uint32_t k = 0; k < mains; ++k) {
        const uint32_t ci = p.convWork[cb + k] & 0xFFFFu;
        auto it = ci < p.convNodeIds.size() ? nodes.find(p.convNodeIds[ci]) : nodes.end();
        if (it == nodes.end() || it->second.convQp == 0u) { anyShortPath = true; continue; }
        stateBlocks = std::max(stateBlocks, std::max(it->second.convHistBlocks, it->second.convP));
    }
    const uint32_t longRows = (convLong && convMaxQp) ? convMaxQp - 1u : 0u;
    const bool longSet = longRows && batch >= 8u && (batch & 7u) == 0u;
    {   // the scratch holds per-node headers (which convolver's spectra its ring carries from set to set, conv_long.inc): they mean
        // something only under the layout they were written for — a new allocation or another set geometry starts from zeroed scratch
        const uint64_t key = ((uint64_t)(uint32_t)batchBlocks << 32) | longRows;
        if (key != convScratchKey && dConvScratch) { HIP_WARN(hipMemsetAsync(dConvScratch, 0, convScratchFloats * sizeof(float), stream)); convScratchKey = key; }
    }
    if (longSet && stateBlocks > 1u) {
        convLongSets++;
        // staleness is a fact of a NODE (its header's H_OVL_STALE), not of the engine: a plan without this node may render
        // block-at-a-time in between, and the node must still be repaired when a later plan brings it back
        for (uint32_t k = 0; k < mains; ++k) {
            const uint32_t ci = p.convWork[cb + k] & 0xFFFFu;
            if (ci < p.convNodeIds.size()) convStaleNodes.insert(p.convNodeIds[ci]);
        }
    }
    if (!longSet) fixConvOverlaps(p);
    launch_convolve_batch(stream, p.view, dRecs, dHbm, dGlobals, cb, mains, batch, arenaFloats, dConvScratch, (uint32_t)batchBlocks, (uint32_t)convMfma,
                          convMinP <= convolve_mfma_max_partitions(), convMaxP > convolve_mfma_max_partitions(), longRows, anyShortPath, stateBlocks, convLongMacMode,
                          longSet ? setInDirect : nullptr, setNumIn, longSet ? setOutDirect : nullptr, setNumOut);
}

// A plan that consists of long-partition convolvers only (BASELINE configs[2]: root(convolve(in)) per channel — in and root folded into
// the node's launches) needs neither the copy of the caller's input blocks into the arenas nor the bus-sum epilogue of a launch set:
// the long-partition kernels read the [block][channel][frame] input where it lies, and when every output channel of the call is the
// folded root of exactly one running convolver they write the caller's output buffer themselves (r05: 8.4 + 9.5 us of a 118 us C3 set).
void Engine::chooseConvDirectIo(const Plan& p, size_t nIn, size_t nOut, uint32_t batch, bool haveIn, bool& dIn, bool& dOut) {
    dIn = dOut = false;
    if (!convDirectIo || !convLong || !convMaxQp || p.convs.empty() || !p.levelIslands.empty() || !p.hosts.empty() || !p.taps.empty()) return;
    if (batch < 8u || (batch & 7u) != 0u) return;
    std::vector<uint32_t> fused;
    for (size_t ci = 0; ci < p.convs.size(); ++ci) {
        auto it = ci < p.convNodeIds.size() ? nodes.find(p.convNodeIds[ci]) : nodes.end();
        if (it == nodes.end() || it->second.convQp == 0u) return;            // a node without long-partition spectra: the 512-partition kernels read the arenas
        if (p.convs[ci].fuseRootRec != kNone) fused.push_back(p.convs[ci].fuseRootRec);
    }
    dIn = haveIn;
    std::vector<uint8_t> seen(nOut, 0);
    size_t covered = 0;
    for (int32_t id : p.rootIds) {
        auto it = nodes.find(id);
        if (it == nodes.end()) return;
        const Node& r = it->second;
        const bool on = r.target > 0.5f, settled = std::fabs(r.target - r.gain) <= 1e-6f;
        if (!((on || !settled) && r.channel >= 0 && (size_t)r.channel < nOut)) continue;   // not running (or a channel the call does not ask for)
        if (std::find(fused.begin(), fused.end(), r.rec) == fused.end() || seen[(size_t)r.channel]) return;
        seen[(size_t)r.channel] = 1; ++covered;
    }
    dOut = covered == nOut && nOut > 0;
}

void Engine::fixConvOverlaps(const Plan& p) {
    if (convStaleNodes.empty() || p.convs.empty() || !dConvScratch) return;
    bool any = false;
    for (int32_t id : p.convNodeIds) any = convStaleNodes.erase(id) > 0 || any;      // (the kernels return early for nodes whose header is not stale)
    if (!any) return;
    launch_convolve_fix_overlap(stream, p.view, dRecs, dHbm, dGlobals, 0u, (uint32_t)p.convWork.size(), dConvScratch, (uint32_t)batchBlocks,
                                (convLong && convMaxQp) ? convMaxQp - 1u : 0u, convMaxP);
}

void Engine::enqueueBatch(const Plan& p, uint32_t batch, float* outRing) {
    if (!outRing) outRing = dOutRing;
    const uint32_t arenaFloats = p.numHbmBuffers * (uint32_t)blockSize;
    const size_t L = p.levelOffsets.size() - 1;
    const bool prof = profileLaunches && (profSetCounter++ % profileEvery) == 0u;
    // a launch set of ONE block (elemhip_process): the last level's kernel ends with the epilogue when it can (spec_epilogue_tail)
    const bool mayFuse = fuseEpilogue && batch == 1u && L > 0 && p.convs.empty() && p.taps.empty() && p.roots.size() <= 32 && !debugSyncOn();
    bool fused = false;
    for (size_t l = 0; l < L; ++l) {
        const uint32_t b = p.levelOffsets[l], e = p.levelOffsets[l + 1];
        if (e <= b && p.convLevelOffsets[l + 1] <= p.convLevelOffsets[l]) continue;
        if (prof) (void)hipEventRecord(profEvent(), stream);
        fused = launchLevelBatch(p, l, batch, arenaFloats, (mayFuse && l + 1 == L) ? outRing : nullptr);
        launchConvolveBatch(p, l, batch, arenaFloats);
        if (prof) { (void)hipEventRecord(profEvent(), stream); profSlots.push_back((uint32_t)l); }
    }
    if (prof && !setOutDirect) (void)hipEventRecord(profEvent(), stream);      // (a direct-I/O set has no epilogue to bracket: two stream operations less per set)
    if (setOutDirect) {
        // the convolvers wrote the caller's buffer themselves (chooseConvDirectIo): what is left of the epilogue is the device's sample
        // clock, moved on by a parameter patch (applied in stream order with the next call's patches)
        // (r06: ... and only once something is about to READ it — a stream of such sets has no island, no epilogue and no other
        //  reader of the clock, and a patch launch per set was 4 us of kernel plus its launch gap in front of every 70 us of work:
        //  flushPending brings the device's clock up to the host's before the first launch that is not another direct set)
        deviceClockBehind = true;
    } else if (!fused) {
        launch_epilogue_batch(stream, p.view, dRecs, dHbm, dGlobals, outRing, batch, arenaFloats, armFlag, armValue);
        if (armFlag && batch == 1u) flagArmed = true;
    } else { st.fusedEpilogues++; if (armFlag && batch == 1u) flagArmed = true; }
    debugSync("set: epilogue", batch);
    if (prof && !setOutDirect) { (void)hipEventRecord(profEvent(), stream); profSlots.push_back((uint32_t)L); }
    if (prof) { profSets++; profBlocks += batch; if (profMs.size() <= L) profMs.resize(L + 1, 0.0); }     // (slot L = the epilogue, 0 for direct sets)
}

bool Engine::specBlockOk(const Plan& p) const { return specBlocks && p.convs.empty() && p.hosts.empty() && specReady(p); }

void Engine::enqueueSpecBlock(const Plan& p, float* outRing) {
    if (!outRing) outRing = dOutRing;
    const uint32_t arenaFloats = p.numHbmBuffers * (uint32_t)blockSize;
    const size_t L = p.levelOffsets.size() - 1;
    for (size_t l = 0; l < L; ++l) (void)launchLevelBatch(p, l, 1u, arenaFloats);
    launch_epilogue(stream, p.view, dRecs, dHbm, dGlobals, outRing, armFlag, armValue);
    if (armFlag) flagArmed = true;
    debugSync("block: specialised levels + epilogue");
    st.specFadeBlocks++;
}

int Engine::processBlocks(const float* inDev, size_t nIn, float* outDev, size_t nOut, size_t numBlocks, int64_t sampleTime) {
    RenderGuard lock(*this);
    if (dry) return kNoDevice;
    if (hostBlockSize != blockSize) return kBlockTooLarge;     // (the device-resident layout is [block][channel][blockSize <= 512])
    if (hipSetDevice(device) != hipSuccess) return kHipError;
    int rc = enqueueBlocks(inDev, nIn, outDev, nOut, numBlocks, sampleTime);
    if (rc != kOk) return rc;
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipGetLastError());
    if (profUsed) profCollect();
    freeDeferred();
    return kOk;
}

// `mu` held, device current. Everything is enqueued on `stream`; the caller synchronises.
int Engine::enqueueBlocks(const float* inDev, size_t nIn, float* outDev, size_t nOut, size_t numBlocks, int64_t sampleTime) {
    if (nIn > kMaxHostIn || nOut > kMaxOutBus) return kTooManyChannels;
    armFlag = nullptr; flagArmed = false;      // (no launch set publishes elemhip_process' completion word)
    int rc = swapInPending();
    if (rc != kOk) return rc;
    if (!current || numBlocks == 0) return kOk;
    Plan& p = *current;
    if (p.packedRootChannels > 0 && nOut < (size_t)p.packedRootChannels) {      // (plan.cpp `pack_roots`: a packed root would not run in the reference)
        std::fprintf(stderr, "[elemhip] pack_roots: this plan needs calls with at least %d output channels\n", p.packedRootChannels);
        return kInvalidPropertyValue;
    }
    const size_t bs = (size_t)blockSize;
    const bool graphOk = useGraph && p.hosts.empty() && !debugSyncOn();   // call-out nodes synchronise inside a block: nothing to capture
    const bool haveIn = nIn > 0 && inDev != nullptr;
    const size_t G = graphOk ? (size_t)graphBlocks : 1;
    rc = ensureOutRing(std::max<size_t>(nOut, 1) * bs * G);
    if (rc != kOk) return rc;

    setGlobalsFor(nIn, nOut, bs, sampleTime);
    size_t done = 0;
    while (done < numBlocks) {
        if (batchBlocks > 1 && numBlocks - done > 1 && batchEligible(p, nOut)) {
            // ---- multi-block launches: one kernel per level renders `chunk` blocks (kernels.hip) ----
            const size_t setCap = std::min((size_t)batchBlocks, maxSetBlocks(p));
            const size_t chunk = std::min(setCap, numBlocks - done);
            rc = ensureHbm(arenaBuffers(p, setCap));
            if (rc != kOk) return rc;
            rc = ensureOutRing(std::max<size_t>(nOut, 1) * bs * (size_t)batchBlocks);
            if (rc != kOk) return rc;
            if (!p.convs.empty()) {
                const size_t need = p.convs.size() * convolve_batch_scratch_floats((uint32_t)batchBlocks, (convLong && convMaxQp) ? convMaxQp - 1u : 0u);
                if (need > convScratchFloats) {
                    HIP_OK(hipStreamSynchronize(stream));
                    if (dConvScratch) (void)hipFree(dConvScratch);
                    dConvScratch = nullptr; convScratchFloats = 0;
                    HIP_OK(hipMalloc(&dConvScratch, need * sizeof(float)));
                    convScratchFloats = need;
                    convScratchKey = ~0ull;        // (its per-node headers are garbage: zeroed before the first launch that reads them)
                }
            }
            setInRing(nullptr, 0);
            hGlobals.blockSlot = 0;
            bool dIn = false, dOut = false;
            chooseConvDirectIo(p, nIn, nOut, (uint32_t)chunk, haveIn, dIn, dOut);
            float* const outTarget = (outDev && nOut > 0) ? outDev + done * nOut * bs : nullptr;
            setInDirect = dIn ? inDev + done * nIn * bs : nullptr; setNumIn = (uint32_t)nIn;
            setOutDirect = dOut ? (outTarget ? outTarget : dOutRing) : nullptr; setNumOut = (uint32_t)nOut;
            if (dIn || dOut) convDirectSets++;
            nextSetDirect = dOut;
            if (haveIn && !dIn)   // host inputs of block b -> arena buffers 0..nIn-1 of block b's arena
                HIP_OK(hipMemcpy2DAsync(dHbm, (size_t)p.numHbmBuffers * bs * sizeof(float), inDev + done * nIn * bs, nIn * bs * sizeof(float),
                                        nIn * bs * sizeof(float), chunk, hipMemcpyDeviceToDevice, stream));
            rc = flushPending();
            nextSetDirect = false;
            if (rc != kOk) return rc;
            // the set's epilogue sums the roots straight into the caller's [block][channel][frame] buffer (r04: into the engine's ring and
            // a device-to-device copy behind it — 5.6 us of a 135 us C3 set, 100 us of a C4 set)
            enqueueBatch(p, (uint32_t)chunk, outTarget);
            setInDirect = nullptr; setOutDirect = nullptr;
            hGlobals.sampleTime += (int64_t)(chunk * bs);
            done += chunk;
            st.blocksRendered += chunk;
            promoteDeferredShapes();
            st.batchLaunches++;
            continue;
        }
        if (specBlockOk(p)) {
            // ---- one block through the specialised kernels (root fades running, or a call of one block) ----
            rc = ensureHbm(arenaBuffers(p, 1));
            if (rc != kOk) return rc;
            if (hGlobals.ringSlots != 1u || hGlobals.blockSlot != 0u) {
                hGlobals.ringSlots = 1u; hGlobals.blockSlot = 0u;
                patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, ringSlots) / 4), 1u, 0u});
                patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, blockSlot) / 4), 0u, 0u});
            }
            setInRing(nullptr, 0);
            if (haveIn) HIP_OK(hipMemcpyAsync(dHbm, inDev + done * nIn * bs, nIn * bs * sizeof(float), hipMemcpyDeviceToDevice, stream));
            rc = flushPending();
            if (rc != kOk) return rc;
            curBlockTime = hGlobals.sampleTime;
            if (batchEligible(p, nOut, true)) enqueueBatch(p, 1u); else enqueueSpecBlock(p);
            if (outDev && nOut > 0)
                HIP_OK(hipMemcpyAsync(outDev + done * nOut * bs, dOutRing, nOut * bs * sizeof(float), hipMemcpyDeviceToDevice, stream));
            mirrorRootFades(p, (uint32_t)bs, (uint32_t)nOut, (uint32_t)nIn);
            hGlobals.sampleTime += (int64_t)bs;
            done += 1;
            st.blocksRendered += 1;
            promoteDeferredShapes();
            continue;
        }
        const size_t chunk = std::min(G, numBlocks - done);
        // ring geometry for this chunk
        if (hGlobals.ringSlots != (uint32_t)G || hGlobals.blockSlot != 0) {
            hGlobals.ringSlots = (uint32_t)G; hGlobals.blockSlot = 0;
            patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, ringSlots) / 4), (uint32_t)G, 0u});
            patches.push_back(Patch{2u, (uint32_t)(offsetof(Globals, blockSlot) / 4), 0u, 0u});
        }
        // host inputs: block 0 of the chunk is copied here, the epilogue of block k stages block k + 1
        setInRing(haveIn ? inDev + done * nIn * bs : nullptr, haveIn ? (uint32_t)chunk : 0u);
        if (haveIn) HIP_OK(hipMemcpyAsync(dHbm, inDev + done * nIn * bs, nIn * bs * sizeof(float), hipMemcpyDeviceToDevice, stream));
        rc = flushPending();
        if (rc != kOk) return rc;
        fixConvOverlaps(p);      // (convolve nodes a long-partition set rendered last: their overlap, before block-at-a-time launches read it)
        // (a plan is captured at its third block-at-a-time chunk: a live graph's plan renders the two blocks of its root fades this
        //  way, then launch sets take over and the next commit replaces it — a capture would be made and thrown away every time)
        if (graphOk && chunk == G && (p.graphExec || ++p.blockChunks > 2u)) {
            if (!p.graphExec || p.graphBlocks != (int)G) {
                if (p.graphExec) { (void)hipGraphExecDestroy(p.graphExec); p.graphExec = nullptr; }
                hipGraph_t graph = nullptr;
                HIP_OK(hipStreamSynchronize(stream));
                const auto tc0 = std::chrono::steady_clock::now();
                const uint64_t ibi = islandBlocksInterp;
                HIP_OK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
                for (size_t b = 0; b < G; ++b) enqueueBlock(p);
                HIP_OK(hipStreamEndCapture(stream, &graph));
                islandBlocksInterp = ibi;                                   // (counted per replay below)
                HIP_OK(hipGraphInstantiate(&p.graphExec, graph, nullptr, nullptr, 0));
                (void)hipGraphDestroy(graph);
                p.graphBlocks = (int)G;
                st.graphCaptures++;
                st.lastGraphCaptureMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count();
            }
            HIP_OK(hipGraphLaunch(p.graphExec, stream));
            st.graphReplays++;
            islandBlocksInterp += (uint64_t)G * p.levelIslands.size();
        } else {
            for (size_t b = 0; b < chunk; ++b) { curBlockTime = hGlobals.sampleTime + (int64_t)(b * bs); enqueueBlock(p); }
        }
        if (outDev && nOut > 0)
            HIP_OK(hipMemcpyAsync(outDev + done * nOut * bs, dOutRing, chunk * nOut * bs * sizeof(float), hipMemcpyDeviceToDevice, stream));
        // host mirrors of device-side block state
        for (size_t b = 0; b < chunk; ++b) mirrorRootFades(p, (uint32_t)bs, (uint32_t)nOut, (uint32_t)nIn);
        hGlobals.sampleTime += (int64_t)(chunk * bs);
        hGlobals.blockSlot = (uint32_t)((hGlobals.blockSlot + chunk) % G);
        done += chunk;
        st.blocksRendered += chunk;
        promoteDeferredShapes();
    }
    return kOk;
}

int Engine::ensureHostStaging(size_t outFloats, size_t inFloats) {
    if (!ioStream) {
        HIP_OK(hipStreamCreateWithFlags(&ioStream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIP_OK(hipEventCreateWithFlags(&evIn[k], hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&evRendered[k], hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&evOut[k], hipEventDisableTiming));
        }
    }
    auto grow = [&](float* (&h)[2], float* (&d)[2], size_t& have, size_t want) -> int {
        if (want <= have) return kOk;
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipStreamSynchronize(ioStream));
        // the new buffers first: a failed allocation leaves the old pair (and `have`) as they were
        float* nh[2] = {nullptr, nullptr}; float* nd[2] = {nullptr, nullptr};
        bool ok = true;
        for (int k = 0; k < 2 && ok; ++k)
            ok = hipHostMalloc((void**)&nh[k], want * sizeof(float), hipHostMallocDefault) == hipSuccess && hipMalloc(&nd[k], want * sizeof(float)) == hipSuccess;
        if (!ok) {
            for (int k = 0; k < 2; ++k) { if (nh[k]) (void)hipHostFree(nh[k]); if (nd[k]) (void)hipFree(nd[k]); }
            (void)hipGetLastError();
            return kHipError;
        }
        for (int k = 0; k < 2; ++k) {
            if (h[k]) (void)hipHostFree(h[k]);
            if (d[k]) (void)hipFree(d[k]);
            h[k] = nh[k]; d[k] = nd[k];
        }
        have = want;
        return kOk;
    };
    int rc = grow(hStageOut, dStageOut, stageOutFloats, outFloats);
    if (rc != kOk) return rc;
    return grow(hStageIn, dStageIn, stageInFloats, inFloats);
}

// Runtime::process for a whole offline render (offline-renderer/index.ts:87-133): planar host arrays of `numFrames` frames.
// Set k (up to `batch_blocks` blocks) is gathered into pinned half k % 2, copied in on the copy stream, rendered on the
// engine's stream into device half k % 2, copied out on the copy stream and scattered to the caller's arrays while set
// k + 1 renders. The render lock is taken per set: a commit on another thread lands between two sets (block boundary).
int Engine::processBlocksHost(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t numFrames, int64_t sampleTime) {
    if (dry) return kNoDevice;
    if (nIn > kMaxHostIn || nOut > kMaxOutBus) return kTooManyChannels;
    if ((nIn && !in) || (nOut && !out)) return kInvalidInstructionFormat;
    const size_t bs = (size_t)blockSize;
    // whole HOST blocks, like the reference's block loop (a host block = hostBlockSize / blockSize engine blocks)
    const size_t hb = (size_t)hostBlockSize;
    bool tapSlices = hb % bs != 0;          // ragged slices (a host block that no k divides evenly): launch sets hold whole engine blocks
    if (hb != bs && !tapSlices) { std::lock_guard<std::mutex> lock(mu); tapSlices = !tapNodeIds.empty(); }
    if (tapSlices) {
        // taps under a host block longer than the engine's: every slice needs its own stretch of the shared tap buffers (setTapSlice),
        // which launch sets do not do — host block by host block through process()
        std::vector<const float*> ip(nIn);
        std::vector<float*> op(nOut);
        std::vector<float> tailIn, tailOut;
        for (size_t f0 = 0; f0 < numFrames; f0 += hb) {
            const size_t nf = std::min(hb, numFrames - f0);
            for (size_t c = 0; c < nIn; ++c) ip[c] = in[c] + f0;
            for (size_t c = 0; c < nOut; ++c) op[c] = out[c] + f0;
            if (nf < hb) {
                // the last, partly filled host block is still a whole block to the engine (offline-renderer/index.ts:104-131: inputs
                // padded with zeros, the frames beyond the caller's arrays dropped)
                tailIn.assign(nIn * hb, 0.0f); tailOut.assign(nOut * hb, 0.0f);
                for (size_t c = 0; c < nIn; ++c) { std::memcpy(tailIn.data() + c * hb, in[c] + f0, nf * sizeof(float)); ip[c] = tailIn.data() + c * hb; }
                for (size_t c = 0; c < nOut; ++c) op[c] = tailOut.data() + c * hb;
            }
            const int rc = process(ip.data(), nIn, op.data(), nOut, hb, sampleTime + (int64_t)f0);
            if (rc != kOk) return rc;
            if (nf < hb) for (size_t c = 0; c < nOut; ++c) std::memcpy(out[c] + f0, tailOut.data() + c * hb, nf * sizeof(float));
        }
        return kOk;
    }
    const size_t numBlocks = ((numFrames + hb - 1) / hb) * (hb / bs);
    if (numBlocks == 0) return kOk;
    size_t setBlocks;
    {
        RenderGuard lock(*this);
        if (hipSetDevice(device) != hipSuccess) return kHipError;
        setBlocks = (size_t)std::max(1, batchBlocks);
        if (setBlocks < 8) setBlocks = std::min<size_t>(64, numBlocks);     // per-block launch path: still stage whole chunks
        setBlocks = std::min(setBlocks, numBlocks);
        // launch sets hold whole HOST blocks: a newer render sequence is adopted at a set boundary (enqueueBlocks), and the reference
        // swaps sequences at host-block boundaries only (Runtime.h:277-285)
        if (hb > bs) setBlocks = std::min(numBlocks, std::max(hb / bs, setBlocks / (hb / bs) * (hb / bs)));
        int rc = ensureHostStaging(setBlocks * std::max<size_t>(nOut, 1) * bs, setBlocks * std::max<size_t>(nIn, 1) * bs);
        if (rc != kOk) return rc;
    }
    const size_t numSets = (numBlocks + setBlocks - 1) / setBlocks;
    auto scatter = [&](size_t k) {     // pinned half -> the caller's planar arrays
        const size_t b0 = k * setBlocks, nb = std::min(setBlocks, numBlocks - b0);
        const float* src = hStageOut[k & 1];
        auto part = [&](size_t bBegin, size_t bEnd) {
            for (size_t b = bBegin; b < bEnd; ++b) {
                const size_t f0 = (b0 + b) * bs;
                if (f0 >= numFrames) break;                     // (the engine blocks that only fill up the last host block)
                const size_t n = std::min(bs, numFrames - f0);
                for (size_t c = 0; c < nOut; ++c) std::memcpy(out[c] + f0, src + (b * nOut + c) * bs, n * sizeof(float));
            }
        };
        // One thread copies ~10 GB/s; a set of many channels (C4: 128 outputs x 1024 blocks = 268 MB per 12.6 ms of rendering)
        // needs more than that to stay hidden behind the next set, so big sets are cut over a few threads by block range.
        const size_t bytes = nb * nOut * bs * sizeof(float);
        size_t threads = bytes >= (16u << 20) ? std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency() / 4)) : 1;
        threads = std::min(threads, nb);
        if (threads <= 1) return part(0, nb);
        std::vector<std::thread> pool;
        for (size_t t = 1; t < threads; ++t) pool.emplace_back(part, nb * t / threads, nb * (t + 1) / threads);
        part(0, nb / threads);
        for (auto& th : pool) th.join();
    };
    int result = kOk;
    size_t issued = 0, scattered = 0;
    // a failing HIP call ends the loop; the tail below drains both streams, releases what was deferred and reports the code —
    // `out` then holds the sets scattered so far (whole launch sets, in order), nothing is left in flight
#define HOST_TRY(call) { if ((call) != hipSuccess) { std::fprintf(stderr, "[elemhip] %s failed: %s\n", #call, hipGetErrorString(hipGetLastError())); result = kHipError; break; } }
    for (size_t k = 0; k < numSets; ++k) {
        const size_t b0 = k * setBlocks, nb = std::min(setBlocks, numBlocks - b0);
        const int half = (int)(k & 1);
        if (nIn) {
            if (k >= 2) HOST_TRY(hipEventSynchronize(evIn[half]));     // the H2D of set k - 2 has left this pinned half
            float* dst = hStageIn[half];
            for (size_t b = 0; b < nb; ++b) {
                const size_t f0 = (b0 + b) * bs;
                const size_t n = f0 < numFrames ? std::min(bs, numFrames - f0) : 0;
                for (size_t c = 0; c < nIn; ++c) {
                    float* d = dst + (b * nIn + c) * bs;
                    if (n) std::memcpy(d, in[c] + f0, n * sizeof(float));
                    if (n < bs) std::memset(d + n, 0, (bs - n) * sizeof(float));
                }
            }
        }
        {
            RenderGuard lock(*this);
            if (hipSetDevice(device) != hipSuccess) { result = kHipError; break; }
            if (nIn) {
                // (the copy stream is in order: this H2D runs behind the D2H of set k - 2, which waited for that set's render,
                //  the last reader of this device half)
                HOST_TRY(hipMemcpyAsync(dStageIn[half], hStageIn[half], nb * nIn * bs * sizeof(float), hipMemcpyHostToDevice, ioStream));
                HOST_TRY(hipEventRecord(evIn[half], ioStream));
                HOST_TRY(hipStreamWaitEvent(stream, evIn[half], 0));
            }
            if (k >= 2) HOST_TRY(hipStreamWaitEvent(stream, evOut[half], 0));   // the D2H of set k - 2 has drained this device half
            int rc = enqueueBlocks(nIn ? dStageIn[half] : nullptr, nIn, nOut ? dStageOut[half] : nullptr, nOut, nb,
                                   sampleTime + (int64_t)(b0 * bs));
            if (rc != kOk) { result = rc; break; }
            if (hb > bs) {                                      // (whole host blocks of hb / bs slices each: where each one ended, for the event relay)
                const uint64_t per = hb / bs, base = st.blocksRendered - nb;
                for (uint64_t e = per; e <= nb; e += per) { if (hostBlockEnds.size() >= 65536) hostBlockEnds.pop_front(); hostBlockEnds.push_back(base + e); }
            }
            HOST_TRY(hipEventRecord(evRendered[half], stream));
            HOST_TRY(hipStreamWaitEvent(ioStream, evRendered[half], 0));
            if (nOut) HOST_TRY(hipMemcpyAsync(hStageOut[half], dStageOut[half], nb * nOut * bs * sizeof(float), hipMemcpyDeviceToHost, ioStream));
            HOST_TRY(hipEventRecord(evOut[half], ioStream));
            issued = k + 1;
        }
        if (k >= 1) {   // set k - 1 arrives while set k renders
            HOST_TRY(hipEventSynchronize(evOut[(k - 1) & 1]));
            if (nOut) scatter(k - 1);
            scattered = k;
        }
    }
#undef HOST_TRY
    if (issued > scattered && result == kOk) {       // the last set (every earlier one was scattered while its successor rendered)
        const size_t last = issued - 1;
        if (hipEventSynchronize(evOut[last & 1]) != hipSuccess) result = kHipError;
        else if (nOut) scatter(last);
    }
    {
        RenderGuard lock(*this);
        if (hipStreamSynchronize(stream) != hipSuccess) result = result == kOk ? kHipError : result;
        if (hipStreamSynchronize(ioStream) != hipSuccess) result = result == kOk ? kHipError : result;
        if (hipGetLastError() != hipSuccess && result == kOk) result = kHipError;
        if (profUsed) profCollect();
        freeDeferred();
    }
    return result;
}

} // namespace elemhip
