// oracle/elem_oracle.cpp — CPU restatement of the block-render path.  TEST INFRASTRUCTURE ONLY.
//
// A plain scalar interpreter of the reference's algorithm for Runtime::process() and the graph
// mutation that feeds it, written from the reference's behaviour (file:line cited per function;
// paths relative to /root/reference).  Nothing under elementary_amd/ includes, links or calls this
// file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load the resulting
// oracle/libelemoracle.so, and only as the checker.
//
// Pinning: tests/test_oracle_golden.py checks this restatement against the golden vectors the
// reference's own jest suites hold for the path (tests/golden/*.json, transcribed from
// js/packages/offline-renderer/__tests__/__snapshots__) and, where oracle/_ref was built, against the
// unmodified reference engine on randomized graphs (tests/test_oracle_vs_ref.py).
//
// Build: g++ -std=c++17 -O2 -ffp-contract=off (oracle/Makefile `port`). No FMA contraction: the
// reference's float recurrences are only reproducible op for op.
#include "fftconv_oracle.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

// ---- minimal JSON value (every number is a double, like runtime/elem/JSON.h:91-104) ----------
struct JV {
    enum T { Undef, Null, Bool, Num, Str, Arr, Obj } t = Undef;
    bool b = false; double n = 0; std::string s;
    std::vector<JV> a; std::vector<std::pair<std::string, JV>> o;
};

struct JP {
    const char* p; const char* e;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool str(std::string& out) {
        if (p >= e || *p != '"') return false;
        ++p;
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) {
                ++p;
                switch (*p) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'u': {
                        if (e - p < 5) return false;
                        unsigned cp = (unsigned)std::strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                        p += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += *p;
                }
                ++p;
            } else out += *p++;
        }
        if (p >= e) return false;
        ++p;
        return true;
    }
    bool val(JV& v) {
        ws();
        if (p >= e) return false;
        if (*p == '[') {
            ++p; v.t = JV::Arr; ws();
            if (p < e && *p == ']') { ++p; return true; }
            for (;;) {
                v.a.emplace_back();
                if (!val(v.a.back())) return false;
                ws();
                if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == ']') { ++p; return true; }
                return false;
            }
        }
        if (*p == '{') {
            ++p; v.t = JV::Obj; ws();
            if (p < e && *p == '}') { ++p; return true; }
            for (;;) {
                ws();
                std::string k;
                if (!str(k)) return false;
                ws();
                if (p >= e || *p != ':') return false;
                ++p;
                v.o.emplace_back(k, JV());
                if (!val(v.o.back().second)) return false;
                ws();
                if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == '}') { ++p; return true; }
                return false;
            }
        }
        if (*p == '"') { v.t = JV::Str; return str(v.s); }
        if (!std::strncmp(p, "true", 4)) { v.t = JV::Bool; v.b = true; p += 4; return true; }
        if (!std::strncmp(p, "false", 5)) { v.t = JV::Bool; v.b = false; p += 5; return true; }
        if (!std::strncmp(p, "null", 4)) { v.t = JV::Null; p += 4; return true; }
        char* end = nullptr;
        std::string tmp(p, std::min<size_t>(64, (size_t)(e - p)));
        v.n = std::strtod(tmp.c_str(), &end);
        if (end == tmp.c_str()) return false;
        p += end - tmp.c_str();
        v.t = JV::Num;
        return true;
    }
};

enum Kind {
    K_IN, K_SIN, K_COS, K_TAN, K_TANH, K_ASINH, K_LN, K_LOG, K_LOG2, K_CEIL, K_FLOOR, K_ROUND, K_SQRT, K_EXP, K_ABS,
    K_LE, K_LEQ, K_GE, K_GEQ, K_POW, K_EQ, K_AND, K_OR,
    K_ADD, K_SUB, K_MUL, K_DIV, K_MOD, K_MIN, K_MAX,
    K_ROOT, K_CONST, K_PHASOR, K_SPHASOR, K_SR, K_SEQ, K_COUNTER, K_ACCUM, K_LATCH, K_MAXHOLD, K_ONCE, K_RAND,
    K_DELAY, K_SDELAY, K_Z, K_POLE, K_ENV, K_BIQUAD, K_PREWARP, K_MM1P, K_SVF, K_SVFSHELF, K_TAPIN, K_TAPOUT,
    K_BLEPSAW, K_BLEPSQUARE, K_BLEPTRIANGLE, K_TIME, K_METRO, K_SAMPLESEQ, K_CONVOLVE, K_TABLE, K_SEQ2, K_SPARSEQ2, K_SAMPLE, K_METER, K_SNAPSHOT, K_SCOPE, K_SPARSEQ, K_CAPTURE,
};

// registry names: runtime/elem/DefaultNodeTypes.h:49-144 (hot-path subset) + wasm/Main.cpp:47-61
const std::unordered_map<std::string, Kind>& registry() {
    static const std::unordered_map<std::string, Kind> r = {
        {"in", K_IN}, {"sin", K_SIN}, {"cos", K_COS}, {"tan", K_TAN}, {"tanh", K_TANH}, {"asinh", K_ASINH}, {"ln", K_LN},
        {"log", K_LOG}, {"log2", K_LOG2}, {"ceil", K_CEIL}, {"floor", K_FLOOR}, {"round", K_ROUND}, {"sqrt", K_SQRT},
        {"exp", K_EXP}, {"abs", K_ABS}, {"le", K_LE}, {"leq", K_LEQ}, {"ge", K_GE}, {"geq", K_GEQ}, {"pow", K_POW},
        {"eq", K_EQ}, {"and", K_AND}, {"or", K_OR}, {"add", K_ADD}, {"sub", K_SUB}, {"mul", K_MUL}, {"div", K_DIV},
        {"mod", K_MOD}, {"min", K_MIN}, {"max", K_MAX}, {"root", K_ROOT}, {"const", K_CONST}, {"phasor", K_PHASOR},
        {"sphasor", K_SPHASOR}, {"sr", K_SR}, {"seq", K_SEQ}, {"counter", K_COUNTER}, {"accum", K_ACCUM},
        {"latch", K_LATCH}, {"maxhold", K_MAXHOLD}, {"once", K_ONCE}, {"rand", K_RAND}, {"delay", K_DELAY},
        {"sdelay", K_SDELAY}, {"z", K_Z}, {"pole", K_POLE}, {"env", K_ENV}, {"biquad", K_BIQUAD}, {"prewarp", K_PREWARP},
        {"mm1p", K_MM1P}, {"svf", K_SVF}, {"svfshelf", K_SVFSHELF}, {"tapIn", K_TAPIN}, {"tapOut", K_TAPOUT},
        {"blepsaw", K_BLEPSAW}, {"blepsquare", K_BLEPSQUARE}, {"bleptriangle", K_BLEPTRIANGLE}, {"time", K_TIME},
        {"metro", K_METRO}, {"sampleseq", K_SAMPLESEQ}, {"convolve", K_CONVOLVE}, {"table", K_TABLE}, {"seq2", K_SEQ2}, {"sparseq2", K_SPARSEQ2}, {"sample", K_SAMPLE}, {"meter", K_METER}, {"snapshot", K_SNAPSHOT}, {"scope", K_SCOPE}, {"sparseq", K_SPARSEQ}, {"capture", K_CAPTURE},
    };
    return r;
}

using Buf = std::shared_ptr<std::vector<float>>;

// detail::GainFade + detail::BufferReader<float> of builtins/SampleSeq.h:29-166
struct SeqReader {
    float gain = 0, target = 0, step = 0.02f;
    const float* buffer = nullptr; size_t bufferSize = 0, position = 0;
    double sampleDuration = 0, startTime = 0;
    void setTarget(float g) { target = g; step = (target < gain) ? -std::abs(step) : std::abs(step); }          // :33-41
    bool on() const { return std::abs(target - 1.0f) <= 1e-6f; }                                                 // :53-55
    void engage(double start, double now, const float* b, size_t size) {                                         // :75-83
        startTime = start; buffer = b; bufferSize = size; setTarget(1.0f);
        const double p = ((now - startTime) / sampleDuration) * (double)(bufferSize - 1u);
        position = (p >= 0.0 && p < 1.8e19) ? (size_t)p : bufferSize;   // size_t cast of a negative double: out of range
        position = std::min(std::max<size_t>(position, 0), bufferSize);
    }
    bool aligned(double t) const {                                                                               // :94-103
        if (!on()) return true;
        const double p = ((t - startTime) / sampleDuration) * (double)(bufferSize - 1u);
        const int np = (std::abs(p) < 9.2e18) ? (int)(int64_t)p : 0;   // x86-64 size_t cast of a double, then the :99 int cast
        const int delta = (int)position - np;
        return std::abs(delta) < 16;
    }
    void readAdding(float* out, size_t n) {                                                                      // :105-110, :43-51
        for (size_t i = 0; i < n && position < bufferSize; ++i) {
            const float x = buffer[position++];
            float y;
            if (gain == target) y = gain * x;
            else { y = x * gain; gain = std::min(std::max(gain + step, 0.0f), 1.0f); }
            out[i] += y;
        }
    }
    void reset(double dur) { gain = 0; target = 0; sampleDuration = dur; startTime = 0; }                        // :149-155
};

struct Node {
    int32_t id = 0; Kind kind = K_CONST;
    std::vector<std::pair<int32_t, uint32_t>> inlets;    // (source, outlet channel)
    std::vector<std::pair<int32_t, uint32_t>> outlets;   // (destination, outlet channel)
    std::map<std::string, JV> props;
    // params
    float value = 1.0f; int channel = 0; int mode = 0; uint32_t holdSamples = 0xFFFFFFFFu;
    bool hold = false, loop = true; size_t seqOffset = 0; int64_t interval = 0; int sdelayLen = 0;
    // root fade (helpers/GainFade.h)
    float gain = 0, target = 1, step = 0, inStep = 0, outStep = 0; int rootChannel = -1;
    // state
    float f0 = 0, f1 = 0, f2 = 0; double d0 = 0, d1 = 0; uint32_t u0 = 0, u1 = 0; float armed = 0;
    bool firstPulse = false, haveSeq = false, pendingSeq = false, pendingRing = false;
    std::vector<float> ring, newRing; int writeIndex = 0;
    std::vector<float> seq, newSeq; size_t seqIndex = 0;
    std::vector<float> tapPrivate; Buf tapShared, pendingTap; bool tapPending = false;
    // sampleseq (SampleSeq.h:169-404)
    double sampleDuration = 0, rtSampleDuration = 0;
    std::vector<std::pair<double, float>> seqEvents, newSeqEvents; bool pendingEvents = false, haveEvents = false;
    int prevEvent = -1, nextEvent = -1;            // -1 == end()
    Buf sampleBuf, pendingSampleBuf; bool samplePending = false;
    SeqReader readers[2]; size_t activeReader = 0; size_t sampleLen = 0, pendingSampleLen = 0;
    size_t sampleBufSize() const { return sampleLen; }
    int32_t interp = 0;          // sparseq2
    // sparseq (SparSeq.h:17-372): change queue restated as "pending" fields drained at the top of process()
    std::map<int32_t, float> tickSeq, newTickSeq; bool pendingTickSeq = false, haveTickSeq = false;
    int32_t loopStart = -1, loopEnd = -1, newLoopStart = -1, newLoopEnd = -1; bool loopEvent = false, pendingLoop = false;
    int32_t pendLoopStart = -1, pendLoopEnd = -1;
    bool follow = false; int32_t holdOrder = 0; double tickInterval = 0.0;
    int32_t edgeCount = -1; size_t samplesSinceEdge = 0; bool holdValid = false; int32_t holdKey = 0;
    // capture (Capture.h:13-104): 128-frame scratch -> ring of bitceil(sr) frames -> relay
    std::vector<float> capRing, capRelay; size_t capW = 0, capR = 0, capScratchSize = 0; float capScratch[128]; bool capReady = false;
    // meter / snapshot readouts (Analyzers.h): the relay reports the newest one and clears the queue
    bool haveReadout = false; float roMin = 0, roMax = 0, roVal = 0;
    // scope (Analyzers.h:137-250): MultiChannelRingBuffer(4) of 8192 frames
    std::vector<float> scopeRing; size_t scopeW = 0, scopeR = 0;
    // sample (Sample.h:22-231): two VariablePitchLerpReader<float>
    struct LerpReader { float targetGain = 0, gain = 0; double pos = 0; bool hasBuffer = false; } lerp[2];
    size_t currentReader = 0, startOffset = 0, stopOffset = 0; int sampleMode = 0;
    // convolve (wasm/Convolve.h:23-92)
    std::shared_ptr<fftconv_oracle::TwoStageConvolver> convolver, pendingConvolver;
    std::vector<float> out;      // this node's block buffer (one per node, never aliased)
};

struct RootSeq { int32_t root; std::vector<int32_t> order; };
struct Sequence { std::vector<RootSeq> roots; std::set<int32_t> ids; };

inline float clampf(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }
inline double clampd(double v, double lo, double hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }
inline double msToStep(double sr, double ms) { return ms > 1e-6 ? 1.0 / (sr * ms / 1000.0) : 1.0; }   // GainFade.h:10-12
inline int bitceil(int n) { if ((n & (n - 1)) == 0) return n; int o = 1; while (o < n) o <<= 1; return o; }   // BitUtils.h:9-20
inline float changeTick(float& last, float x) {   // helpers/Change.h:20-31
    const float dt = x - last; last = x;
    return dt > 0 ? 1.0f : (dt < 0 ? -1.0f : 0.0f);
}
inline float blep(float phase, float inc) {        // Oscillators.h:23-38
    if (phase < inc) { const float p = phase / inc; return (2.0f - p) * p - 1.0f; }
    if (phase > (1.0f - inc)) { const float p = (phase - 1.0f) / inc; return (p + 2.0f) * p + 1.0f; }
    return 0.0f;
}

struct Oracle {
    double sr; int bs;
    std::unordered_map<int32_t, Node> nodes;
    std::set<int32_t> currentRoots;
    std::unordered_map<std::string, Buf> resources;
    std::unordered_map<std::string, size_t> resourceLen;   // true sample count (tap-sized padding excluded)
    std::shared_ptr<Sequence> current, pending;

    Oracle(double sampleRate, int blockSize) : sr(sampleRate), bs(blockSize) {}

    // ---- Runtime::createNode (Runtime.h:293-313) + node constructors ------------------------
    int createNode(int32_t id, const std::string& type) {
        auto it = registry().find(type);
        if (it == registry().end()) return 1;
        if (nodes.count(id)) return 3;
        Node n; n.id = id; n.kind = it->second; n.out.assign((size_t)bs, 0.0f);
        switch (n.kind) {
            case K_ROOT:                                        // Core.h:80-82
                n.gain = 0; n.target = 1; n.rootChannel = -1;
                n.inStep = (float)msToStep(sr, 20); n.outStep = (float)((double)(-1.0f) * msToStep(sr, 20));
                n.step = n.inStep;
                break;
            case K_RAND: n.u0 = (uint32_t)std::rand(); break;   // Noise.h:42
            case K_DELAY: n.ring.assign((size_t)bs, 0.0f); n.props["size"].t = JV::Num; n.props["size"].n = bs; break;      // Delays.h:56
            case K_SDELAY: n.sdelayLen = bs; n.ring.assign((size_t)bitceil(bs + bs), 0.0f); n.props["size"].t = JV::Num; n.props["size"].n = bs; break; // :183
            case K_TAPOUT: n.tapPrivate.assign((size_t)bs, 0.0f); break;   // Feedback.h:66-67
            case K_METRO: n.interval = (int64_t)std::max(2.0, 1000.0 * 0.001 * sr); break;   // wasm/Metro.h:15
            default: break;
        }
        nodes.emplace(id, std::move(n));
        return 0;
    }

    Buf tapResource(const std::string& name) {   // SharedResource.h:79-92
        auto it = resources.find(name);
        if (it != resources.end()) return it->second;
        auto r = std::make_shared<std::vector<float>>((size_t)bs, 0.0f);
        resources.emplace(name, r);
        return r;
    }

    // ---- per-type setProperty; returns the reference's ReturnCode (Types.h:51-86) ------------
    int setProperty(int32_t id, const std::string& key, const JV& v) {
        auto it = nodes.find(id);
        if (it == nodes.end()) return 2;
        Node& n = it->second;
        const bool num = v.t == JV::Num, boo = v.t == JV::Bool, str = v.t == JV::Str;
        switch (n.kind) {
            case K_CONST: if (key == "value") { if (!num) return 5; n.value = (float)v.n; } break;        // Core.h:142-152
            case K_IN: if (key == "channel") { if (!num) return 5; n.channel = (int)v.n; } break;        // Math.h:95-105
            case K_ROOT:                                                                                  // Core.h:33-64
                if (key == "active") { if (!boo) return 5; n.target = v.b ? 1.0f : 0.0f; n.step = n.gain > n.target ? n.outStep : n.inStep; }
                if (key == "channel") { if (!num) return 5; n.rootChannel = (int)v.n; }
                if (key == "fadeInMs") { if (!num) return 5; n.inStep = (float)msToStep(sr, v.n); n.step = n.gain > n.target ? n.outStep : n.inStep; }
                if (key == "fadeOutMs") { if (!num) return 5; n.outStep = (float)((double)(-1.0f) * msToStep(sr, v.n)); n.step = n.gain > n.target ? n.outStep : n.inStep; }
                break;
            case K_MAXHOLD: if (key == "hold") { if (!num) return 5; n.holdSamples = (uint32_t)(sr * 0.001 * v.n); } break;   // Core.h:292-303
            case K_ONCE: if (key == "arm") { if (!boo) return 5; if (n.armed == 0.0f) n.armed = v.b ? 1.0f : 0.0f; } break;   // Core.h:352-366
            case K_SEQ2:                                                                                  // Seq2.h:38-84 (same properties)
            case K_SEQ:                                                                                   // Core.h:411-458
                if (key == "hold") { if (!boo) return 5; n.hold = v.b; }
                if (key == "loop") { if (!boo) return 5; n.loop = v.b; }
                if (key == "offset") { if (!num) return 5; if (v.n < 0) return 6; n.seqOffset = (size_t)v.n; }
                if (key == "seq") {
                    if (v.t != JV::Arr) return 5;
                    std::vector<float> d(v.a.size());
                    for (size_t i = 0; i < v.a.size(); ++i) { if (v.a[i].t != JV::Num) return 5; d[i] = (float)v.a[i].n; }
                    n.newSeq.swap(d); n.pendingSeq = true;
                }
                break;
            case K_RAND: if (key == "seed") { if (!num) return 5; n.u0 = (uint32_t)(int64_t)v.n; } break;   // Noise.h:13-23
            case K_DELAY:                                                                                 // Delays.h:59-82
                if (key == "size") { if (!num) return 5; const int size = (int)v.n; if (size < 0) return 6; n.newRing.assign((size_t)size, 0.0f); n.pendingRing = true; }
                break;
            case K_SDELAY:                                                                                // Delays.h:186-216
                if (key == "size") {
                    if (!num) return 5;
                    const int len = (int)v.n; const int size = bitceil(len + bs);
                    if (size < 0) return 6;
                    n.newRing.assign((size_t)size, 0.0f); n.pendingRing = true; n.sdelayLen = len;
                }
                break;
            case K_SVF:                                                                                   // filters/SVF.h:30-46
                if (key == "mode") { if (!str) return 5;
                    if (v.s == "lowpass") n.mode = 0; if (v.s == "bandpass") n.mode = 1; if (v.s == "highpass") n.mode = 2;
                    if (v.s == "notch") n.mode = 3; if (v.s == "allpass") n.mode = 4; }
                break;
            case K_SVFSHELF:                                                                              // filters/SVFShelf.h:29-42
                if (key == "mode") { if (!str) return 5;
                    if (v.s == "lowshelf") n.mode = 0; if (v.s == "highshelf") n.mode = 1; if (v.s == "bell" || v.s == "peak") n.mode = 2; }
                break;
            case K_MM1P:                                                                                  // filters/MultiMode1p.h:48-62
                if (key == "mode") { if (!str) return 5;
                    if (v.s == "lowpass") n.mode = 0; if (v.s == "highpass") n.mode = 2; if (v.s == "allpass") n.mode = 4; }
                break;
            case K_TAPIN: case K_TAPOUT:                                                                  // Feedback.h:24-38, 70-84
                if (key == "name") { if (!str) return 5; n.pendingTap = tapResource(v.s); n.tapPending = true; }
                break;
            case K_SAMPLESEQ:                                                                             // SampleSeq.h:181-255
                if (key == "duration") { if (!num) return 5; if (v.n <= 0.0) return 6; n.sampleDuration = v.n; }
                if (key == "path") {
                    if (!str) return 5;
                    auto r = resources.find(v.s);
                    if (r == resources.end()) return 6;
                    n.pendingSampleBuf = r->second; n.samplePending = true; n.pendingSampleLen = resourceLen[v.s];
                }
                if (key == "seq") {
                    if (v.t != JV::Arr) return 5;
                    std::map<double, float> m;
                    for (const JV& e : v.a) {
                        if (e.t != JV::Obj) return 5;
                        const JV* val = nullptr; const JV* tm = nullptr;
                        for (auto& kv : e.o) { if (kv.first == "value") val = &kv.second; if (kv.first == "time") tm = &kv.second; }
                        if (!val || !tm || val->t != JV::Num || tm->t != JV::Num) return 5;
                        m.insert({tm->n, (float)val->n});   // std::map::insert keeps the first entry of a key
                    }
                    n.newSeqEvents.assign(m.begin(), m.end()); n.pendingEvents = true;
                }
                break;
            case K_SAMPLE:                                                                                // Sample.h:25-75
                if (key == "path") {
                    if (!str) return 5;
                    auto r = resources.find(v.s);
                    if (r == resources.end()) return 6;
                    n.pendingSampleBuf = r->second; n.samplePending = true; n.pendingSampleLen = resourceLen[v.s];
                }
                if (key == "mode") { if (!str) return 5; if (v.s == "trigger") n.sampleMode = 0; if (v.s == "gate") n.sampleMode = 1; if (v.s == "loop") n.sampleMode = 2; }
                if (key == "startOffset") { if (!num) return 5; const int vi = (int)v.n; if (vi < 0) return 6; n.startOffset = (size_t)vi; }
                if (key == "stopOffset") { if (!num) return 5; const int vi = (int)v.n; if (vi < 0) return 6; n.stopOffset = (size_t)vi; }
                break;
            case K_SPARSEQ:                                                                               // SparSeq.h:40-131
                if (key == "offset") { if (!num) return 5; if (v.n < 0) return 6; n.seqOffset = (size_t)v.n; }
                if (key == "loop") {
                    if (v.t == JV::Null || (boo && !v.b)) { n.newLoopStart = -1; n.newLoopEnd = -1; n.loopEvent = true; }
                    else {
                        if (v.t != JV::Arr) return 5;
                        if (v.a.size() < 2 || v.a[0].t != JV::Num || v.a[1].t != JV::Num) return 5;
                        n.newLoopStart = (int32_t)v.a[0].n; n.newLoopEnd = (int32_t)v.a[1].n; n.loopEvent = true;
                    }
                }
                if (key == "follow") { if (!boo) return 5; n.follow = v.b; }
                if (key == "interpolate") { if (!num) return 5; n.holdOrder = (int32_t)v.n; }
                if (key == "tickInterval") { if (!num) return 5; if (v.n < 0) return 6; n.tickInterval = (double)(float)sr * v.n; }
                if (key == "seq") {
                    if (v.t != JV::Arr) return 5;
                    std::map<int32_t, float> m;
                    for (const JV& e : v.a) {
                        if (e.t != JV::Obj) return 5;
                        const JV* val = nullptr; const JV* tm = nullptr;
                        for (auto& kv : e.o) { if (kv.first == "value") val = &kv.second; if (kv.first == "tickTime") tm = &kv.second; }
                        if (!val || !tm || val->t != JV::Num || tm->t != JV::Num) return 5;
                        m.insert({(int32_t)tm->n, (float)val->n});
                    }
                    n.newTickSeq.swap(m); n.pendingTickSeq = true;
                }
                break;
            case K_SCOPE:                                                                                 // Analyzers.h:151-173
                if (key == "size") { if (!num) return 5; if (v.n < 256 || v.n > 8192) return 6; }
                if (key == "channels") { if (!num) return 5; if (v.n < 0 || v.n > 4) return 6; }
                if (key == "name") { if (!str) return 5; }
                break;
            case K_TABLE:                                                                                 // Table.h:20-33
                if (key == "path") {
                    if (!str) return 5;
                    auto r = resources.find(v.s);
                    if (r == resources.end()) return 6;
                    n.pendingSampleBuf = r->second; n.samplePending = true; n.pendingSampleLen = resourceLen[v.s];
                }
                break;
            case K_SPARSEQ2:                                                                              // SparSeq2.h:20-54
                if (key == "seq") {
                    if (v.t != JV::Arr) return 5;
                    std::map<double, float> m;
                    for (const JV& e : v.a) {
                        if (e.t != JV::Obj) return 5;
                        const JV* val = nullptr; const JV* tm = nullptr;
                        for (auto& kv : e.o) { if (kv.first == "value") val = &kv.second; if (kv.first == "time") tm = &kv.second; }
                        if (!val || !tm || val->t != JV::Num || tm->t != JV::Num) return 5;
                        m.insert({tm->n, (float)val->n});
                    }
                    n.newSeqEvents.assign(m.begin(), m.end()); n.pendingEvents = true;
                }
                if (key == "interpolate") { if (!num) return 5; n.interp = (int32_t)v.n; }
                break;
            case K_CONVOLVE:                                                                              // wasm/Convolve.h:34-56
                if (key == "path") {
                    if (!str) return 5;
                    auto r = resources.find(v.s);
                    if (r == resources.end()) return 6;
                    auto co = std::make_shared<fftconv_oracle::TwoStageConvolver>();
                    co->init(512, 4096, r->second->data(), resourceLen[v.s]);
                    n.pendingConvolver = co;
                }
                break;
            case K_METRO:                                                                                 // wasm/Metro.h:18-34
                if (key == "interval") { if (!num) return 5; if (0 >= v.n) return 6; n.interval = (int64_t)std::max(2.0, v.n * 0.001 * sr); }
                break;
            default: break;
        }
        n.props[key] = v;
        return 0;
    }

    int appendChild(int32_t parent, int32_t child, int32_t ch) {   // Runtime.h:335-366
        if (!nodes.count(parent) || !nodes.count(child)) return 2;
        nodes.at(parent).inlets.emplace_back(child, (uint32_t)ch);
        nodes.at(child).outlets.emplace_back(parent, (uint32_t)ch);
        return 0;
    }

    static bool stillRunning(const Node& r) {   // Core.h:28-31, GainFade.h:98-104
        return r.target > 0.5f || !(std::fabs(r.target - r.gain) <= 1e-6f);
    }

    int activateRoots(const std::vector<int32_t>& ids) {   // Runtime.h:368-433
        std::set<int32_t> active;
        JV t; t.t = JV::Bool; t.b = true; JV f; f.t = JV::Bool; f.b = false;
        for (int32_t id : ids) {
            if (!nodes.count(id)) return 2;
            if (nodes.at(id).kind == K_ROOT) { setProperty(id, "active", t); active.insert(id); }
        }
        for (int32_t id : currentRoots) {
            auto it = nodes.find(id);
            if (it == nodes.end() || it->second.kind != K_ROOT) continue;
            if (!active.count(id)) setProperty(id, "active", f);
            if (stillRunning(it->second)) active.insert(id);
        }
        currentRoots.swap(active);
        return 0;
    }

    void traverse(std::unordered_set<int32_t>& visited, std::vector<int32_t>& order, int32_t id) {   // Runtime.h:502-518
        if (visited.count(id)) return;
        for (auto& in : nodes.at(id).inlets) if (nodes.count(in.first)) traverse(visited, order, in.first);
        order.push_back(id);
        visited.insert(id);
    }

    std::shared_ptr<Sequence> buildRenderSequence() {   // Runtime.h:520-577
        auto seq = std::make_shared<Sequence>();
        std::vector<int32_t> front, back;
        for (int32_t id : currentRoots) {
            Node& r = nodes.at(id);
            if (r.kind != K_ROOT) continue;
            auto a = r.props.find("active");
            const bool active = a != r.props.end() && a->second.t == JV::Bool && a->second.b;
            if (active) front.push_back(id); else back.push_back(id);
        }
        std::reverse(front.begin(), front.end());   // std::list::push_front
        front.insert(front.end(), back.begin(), back.end());
        std::unordered_set<int32_t> visited;
        for (int32_t rid : front) {
            RootSeq rs; rs.root = rid;
            traverse(visited, rs.order, rid);
            for (int32_t id : rs.order) seq->ids.insert(id);
            seq->roots.push_back(std::move(rs));
        }
        return seq;
    }

    int apply(const JV& batch) {   // Runtime.h:170-218
        if (batch.t != JV::Arr) return 8;
        bool shouldRebuild = false;
        for (const JV& in : batch.a) {
            if (in.t != JV::Arr || in.a.empty() || in.a[0].t != JV::Num) return 8;
            static const JV undef;
            auto arg = [&](size_t i) -> const JV& { return i < in.a.size() ? in.a[i] : undef; };
            int res = 0;
            switch ((int)in.a[0].n) {
                case 0: if (arg(1).t != JV::Num || arg(2).t != JV::Str) res = 8; else res = createNode((int32_t)arg(1).n, arg(2).s); break;
                case 3: if (arg(1).t != JV::Num || arg(2).t != JV::Str) res = 8; else res = setProperty((int32_t)arg(1).n, arg(2).s, arg(3)); break;
                case 2: if (arg(1).t != JV::Num || arg(2).t != JV::Num || arg(3).t != JV::Num) res = 8;
                        else res = appendChild((int32_t)arg(1).n, (int32_t)arg(2).n, (int32_t)arg(3).n); break;
                case 4: {
                    if (arg(1).t != JV::Arr) { res = 8; break; }
                    std::vector<int32_t> ids; bool bad = false;
                    for (const JV& v : arg(1).a) { if (v.t != JV::Num) { bad = true; break; } ids.push_back((int32_t)v.n); }
                    res = activateRoots(ids);
                    if (!res && bad) res = 8;
                    shouldRebuild = true;
                    break;
                }
                case 5: if (shouldRebuild) pending = buildRenderSequence(); break;
                default: break;
            }
            if (res) return res;
        }
        return 0;
    }

    // ---- one node, one block: the per-sample loops of runtime/elem/builtins/** ------------------------
    void processNode(Node& n, const float* const* in, size_t nIn, size_t N, int64_t sampleTime) {
        float* out = n.out.data();
        auto zero = [&]() { std::fill_n(out, N, 0.0f); };
        const float srF = (float)sr;
        switch (n.kind) {
            case K_CONST: for (size_t i = 0; i < N; ++i) out[i] = n.value; break;                         // Core.h:154-163
            case K_SR: for (size_t i = 0; i < N; ++i) out[i] = (float)sr; break;                          // Core.h:173-180
            case K_IN: {                                                                                  // Math.h:107-123
                const size_t ch = (size_t)n.channel;
                if (ch >= nIn) { zero(); break; }
                for (size_t i = 0; i < N; ++i) out[i] = in[ch][i];
                break;
            }
            case K_SIN: case K_COS: case K_TAN: case K_TANH: case K_ASINH: case K_LN: case K_LOG: case K_LOG2:
            case K_CEIL: case K_FLOOR: case K_ROUND: case K_SQRT: case K_EXP: case K_ABS: {               // Math.h:9-28
                if (nIn < 1) { zero(); break; }
                for (size_t i = 0; i < N; ++i) {
                    const float x = in[0][i];
                    float y;
                    switch (n.kind) {
                        case K_SIN: y = std::sin(x); break; case K_COS: y = std::cos(x); break; case K_TAN: y = std::tan(x); break;
                        case K_TANH: y = std::tanh(x); break; case K_ASINH: y = std::asinh(x); break; case K_LN: y = std::log(x); break;
                        case K_LOG: y = std::log10(x); break; case K_LOG2: y = std::log2(x); break; case K_CEIL: y = std::ceil(x); break;
                        case K_FLOOR: y = std::floor(x); break; case K_ROUND: y = std::round(x); break; case K_SQRT: y = std::sqrt(x); break;
                        case K_EXP: y = std::exp(x); break; default: y = std::abs(x); break;
                    }
                    out[i] = y;
                }
                break;
            }
            case K_LE: case K_LEQ: case K_GE: case K_GEQ: case K_POW: case K_EQ: case K_AND: case K_OR: { // Math.h:30-57, 142-188
                if (nIn < 2) { zero(); break; }
                const float eps = std::numeric_limits<float>::epsilon();
                for (size_t i = 0; i < N; ++i) {
                    const float x = in[0][i], y = in[1][i];
                    float r;
                    switch (n.kind) {
                        case K_LE: r = x < y; break; case K_LEQ: r = x <= y; break; case K_GE: r = x > y; break; case K_GEQ: r = x >= y; break;
                        case K_POW: r = (x < 0.0f && y != std::floor(y)) ? 0.0f : std::pow(x, y); break;
                        case K_EQ: r = std::abs(x - y) <= eps; break;
                        case K_AND: r = (std::abs(1.0f - x) <= eps) && (std::abs(1.0f - y) <= eps); break;
                        default: r = (std::abs(1.0f - x) <= eps) || (std::abs(1.0f - y) <= eps); break;
                    }
                    out[i] = r;
                }
                break;
            }
            case K_ADD: case K_SUB: case K_MUL: case K_DIV: case K_MOD: case K_MIN: case K_MAX: {         // Math.h:59-89, 128-177
                if (nIn < 1) { zero(); break; }
                for (size_t i = 0; i < N; ++i) out[i] = in[0][i];
                for (size_t k = 1; k < nIn; ++k)
                    for (size_t i = 0; i < N; ++i) {
                        const float a = out[i], b = in[k][i];
                        float r;
                        switch (n.kind) {
                            case K_ADD: r = a + b; break; case K_SUB: r = a - b; break; case K_MUL: r = a * b; break;
                            case K_DIV: r = (b == 0.0f) ? 0.0f : a / b; break; case K_MOD: r = std::fmod(a, b); break;
                            case K_MIN: r = std::min(a, b); break; default: r = std::max(a, b); break;
                        }
                        out[i] = r;
                    }
                break;
            }
            case K_ROOT: {                                                                                // Core.h:66-78, GainFade.h:56-72
                if (nIn < 1) { zero(); break; }
                if (n.gain == n.target) { for (size_t i = 0; i < N; ++i) out[i] = in[0][i] * n.target; break; }
                for (size_t i = 0; i < N; ++i) out[i] = in[0][i] * clampf(n.gain + n.step * (float)(int)i, 0.0f, 1.0f);
                n.gain = clampf(n.gain + n.step * (float)(int)N, 0.0f, 1.0f);
                break;
            }
            case K_PHASOR: case K_SPHASOR: {                                                              // Core.h:85-136
                const bool withReset = n.kind == K_SPHASOR;
                if (nIn < (withReset ? 2u : 1u)) { zero(); break; }
                const float rsr = 1.0f / srF;
                for (size_t i = 0; i < N; ++i) {
                    if (withReset && changeTick(n.f1, in[1][i]) > 0.5f) n.f0 = 0.0f;
                    const float stp = in[0][i] * rsr;
                    out[i] = n.f0;
                    const float next = n.f0 + stp;
                    n.f0 = next - std::floor(next);
                }
                break;
            }
            case K_COUNTER: {                                                                             // Core.h:183-215
                if (nIn < 1) { zero(); break; }
                for (size_t i = 0; i < N; ++i) {
                    if ((1.0f - in[0][i]) <= std::numeric_limits<float>::epsilon()) { out[i] = n.f0; n.f0 = n.f0 + 1.0f; }
                    else { n.f0 = 0.0f; out[i] = 0.0f; }
                }
                break;
            }
            case K_ACCUM: {                                                                               // Core.h:217-248
                if (nIn < 2) { zero(); break; }
                for (size_t i = 0; i < N; ++i) {
                    if (changeTick(n.f1, in[1][i]) > 0.5f) n.f0 = 0.0f;
                    n.f0 += in[0][i];
                    out[i] = n.f0;
                }
                break;
            }
            case K_LATCH: {                                                                               // Core.h:250-286
                if (nIn < 2) { zero(); break; }
                const float eps = std::numeric_limits<float>::epsilon();
                for (size_t i = 0; i < N; ++i) {
                    if (std::abs(n.f0) <= eps && in[0][i] > eps) n.f1 = in[1][i];
                    n.f0 = in[0][i];
                    out[i] = n.f1;
                }
                break;
            }
            case K_MAXHOLD: {                                                                             // Core.h:288-339
                if (nIn < 2) { zero(); break; }
                for (size_t i = 0; i < N; ++i) {
                    const float x = in[0][i];
                    if (changeTick(n.f0, in[1][i]) > 0.5f || ++n.u0 >= n.holdSamples) { n.f1 = x; n.u0 = 0; }
                    else if (x > n.f1) { n.u0 = 0; n.f1 = x; }
                    out[i] = n.f1;
                }
                break;
            }
            case K_ONCE: {                                                                                // Core.h:368-398
                if (nIn < 1) { zero(); break; }
                const bool isArmed = n.armed != 0.0f;
                for (size_t i = 0; i < N; ++i) {
                    const float d = changeTick(n.f1, in[0][i]);
                    if (isArmed && d > 0.5f) { n.f0 = 1.0f; n.armed = 0.0f; }
                    if (d < -0.5f) n.f0 = 0.0f;
                    out[i] = in[0][i] * n.f0;
                }
                break;
            }
            case K_SEQ: {                                                                                 // Core.h:460-555
                if (n.pendingSeq) {
                    n.seq.swap(n.newSeq); n.pendingSeq = false; n.haveSeq = true;
                    if (!n.seq.empty()) n.seqIndex = n.seqIndex % n.seq.size();
                    if (n.firstPulse && !n.seq.empty()) n.f0 = n.seq[n.seqIndex];
                }
                if (nIn < 1 || !n.haveSeq) { zero(); break; }
                const bool hasReset = nIn > 1;
                const size_t len = n.seq.size();
                for (size_t i = 0; i < N; ++i) {
                    const float x = in[0][i], reset = hasReset ? in[1][i] : 0.0f;
                    if (changeTick(n.f2, reset) > 0.5f) n.seqIndex = n.seqOffset;
                    if (changeTick(n.f1, x) > 0.5f) {
                        if (len) n.f0 = n.seq[std::min(n.seqIndex, len - 1)];
                        n.firstPulse = true;
                        if ((++n.seqIndex >= len) && n.loop) n.seqIndex = 0;
                    }
                    if (n.seqIndex < len) out[i] = n.hold ? n.f0 : n.f0 * x;
                    else out[i] = n.hold ? n.f0 : 0.0f;
                }
                break;
            }
            case K_RAND:                                                                                  // Noise.h:28-40
                for (size_t i = 0; i < N; ++i) {
                    n.u0 = 214013u * n.u0 + 2531011u;
                    out[i] = (float)(int)((n.u0 >> 16) & 0x7FFF) / (float)0x7FFF;
                }
                break;
            case K_Z:                                                                                     // Delays.h:15-39
                if (nIn < 1) { zero(); break; }
                for (size_t i = 0; i < N; ++i) { out[i] = n.f0; n.f0 = in[0][i]; }
                break;
            case K_DELAY: {                                                                               // Delays.h:84-161
                if (n.pendingRing) { n.ring.swap(n.newRing); n.pendingRing = false; n.writeIndex = 0; }
                if (nIn < 3) { zero(); break; }
                const int size = (int)n.ring.size();
                float* d = n.ring.data();
                if (size == 0) { std::copy_n(in[0], N, out); break; }
                for (size_t i = 0; i < N; ++i) {
                    const float offset = clampf(in[0][i], 0.0f, (float)size);
                    if (offset <= std::numeric_limits<float>::epsilon()) {
                        d[n.writeIndex] = in[2][i]; out[i] = in[2][i];
                        if (++n.writeIndex >= size) n.writeIndex -= size;
                        continue;
                    }
                    const float readFrac = (float)(size + n.writeIndex) - offset;
                    const int readLeft = (int)readFrac, readRight = readLeft + 1;
                    const float frac = readFrac - std::floor(readFrac);
                    const float left = d[readLeft % size], right = d[readRight % size];
                    const float o = left + frac * (right - left);
                    const float fb = clampf(in[1][i], -1.0f, 1.0f);
                    d[n.writeIndex] = in[2][i] + fb * o;
                    out[i] = o;
                    if (++n.writeIndex >= size) n.writeIndex -= size;
                }
                break;
            }
            case K_SDELAY: {                                                                              // Delays.h:218-264
                if (n.pendingRing) { n.ring.swap(n.newRing); n.pendingRing = false; n.writeIndex = 0; }
                const int size = (int)n.ring.size();
                if (nIn < 1 || size == 0) { zero(); break; }
                const int mask = size - 1, len = n.sdelayLen;
                float* d = n.ring.data();
                const int readStart = n.writeIndex - len;
                for (size_t i = 0; i < N; ++i) { d[n.writeIndex] = in[0][i]; n.writeIndex = (n.writeIndex + 1) & mask; }
                for (int i = 0; i < (int)N; ++i) out[i] = d[(size + readStart + i) & mask];
                break;
            }
            case K_POLE:                                                                                  // Filters.h:13-39
                if (nIn < 2) { zero(); break; }
                for (size_t i = 0; i < N; ++i) { n.f0 = in[1][i] + in[0][i] * n.f0; out[i] = n.f0; }
                break;
            case K_ENV:                                                                                   // Filters.h:46-79
                if (nIn < 3) { zero(); break; }
                for (size_t i = 0; i < N; ++i) {
                    const float vn = std::abs(in[2][i]);
                    if (std::abs(vn) > n.f0) n.f0 = in[0][i] * (n.f0 - vn) + vn;
                    else n.f0 = in[1][i] * (n.f0 - vn) + vn;
                    out[i] = n.f0;
                }
                break;
            case K_BIQUAD:                                                                                // Filters.h:87-120
                if (nIn < 6) { zero(); break; }
                for (size_t i = 0; i < N; ++i) {
                    const float x = in[5][i];
                    const float y = in[0][i] * x + n.f0;
                    n.f0 = in[1][i] * x - in[3][i] * y + n.f1;
                    n.f1 = in[2][i] * x - in[4][i] * y;
                    out[i] = y;
                }
                break;
            case K_PREWARP: {                                                                             // filters/MultiMode1p.h:9-36
                if (nIn < 1) { zero(); break; }
                const double T = 1.0 / sr;
                for (size_t i = 0; i < N; ++i) {
                    const double twoPi = 2.0 * 3.141592653589793238;
                    const double wd = twoPi * (double)in[0][i];
                    out[i] = (float)std::tan(wd * T / 2.0);
                }
                break;
            }
            case K_MM1P:                                                                                  // filters/MultiMode1p.h:64-107
                if (nIn < 2) { zero(); break; }
                for (size_t i = 0; i < N; ++i) {
                    const double g = clampd((double)in[0][i], 0.0, 0.9999);
                    const float xn = in[1][i];
                    const double G = g / (1.0 + g);
                    const double v = ((double)xn - n.d0) * G;
                    const double lp = v + n.d0;
                    n.d0 = lp + v;
                    if (n.mode == 0) out[i] = (float)lp;
                    else if (n.mode == 2) out[i] = xn - (float)lp;
                    else out[i] = (float)(lp + lp - xn);
                }
                break;
            case K_SVF:                                                                                   // filters/SVF.h:48-105
                if (nIn < 3) { zero(); break; }
                for (size_t i = 0; i < N; ++i) {
                    const double fc = in[0][i], q = in[1][i];
                    const double g = std::tan(3.14159265359 * clampd(fc, 20.0, sr / 2.0001) / sr);
                    const double k = 1.0 / clampd(q, 0.25, 20.0);
                    const double a1 = 1.0 / (1.0 + g * (g + k)), a2 = g * a1, a3 = g * a2;
                    const float v0 = in[2][i];
                    const double v3 = v0 - n.d1;
                    const double v1 = n.d0 * a1 + v3 * a2;
                    const double v2 = n.d1 + n.d0 * a2 + v3 * a3;
                    n.d0 = v1 * 2.0 - n.d0;
                    n.d1 = v2 * 2.0 - n.d1;
                    switch (n.mode) {
                        case 0: out[i] = (float)v2; break; case 1: out[i] = (float)v1; break;
                        case 2: out[i] = (float)(v0 - k * v1 - v2); break; case 3: out[i] = (float)(v0 - k * v1); break;
                        default: out[i] = (float)(v0 - 2.0 * k * v1); break;
                    }
                }
                break;
            case K_SVFSHELF:                                                                              // filters/SVFShelf.h:44-124
                if (nIn < 4) { zero(); break; }
                for (size_t i = 0; i < N; ++i) {
                    const double fc = in[0][i], q = in[1][i], dB = in[2][i];
                    const double A = std::pow(10, dB / 40.0);
                    double g = std::tan(3.14159265359 * clampd(fc, 20.0, sr / 2.0001) / sr);
                    double k = 1.0 / clampd(q, 0.25, 20.0);
                    if (n.mode == 0) g /= A;
                    if (n.mode == 1) g *= A;
                    if (n.mode == 2) k /= A;
                    const double a1 = 1.0 / (1.0 + g * (g + k)), a2 = g * a1, a3 = g * a2;
                    const float v0 = in[3][i];
                    const double v3 = v0 - n.d1;
                    const double v1 = n.d0 * a1 + v3 * a2;
                    const double v2 = n.d1 + n.d0 * a2 + v3 * a3;
                    n.d0 = v1 * 2.0 - n.d0;
                    n.d1 = v2 * 2.0 - n.d1;
                    if (n.mode == 2) out[i] = (float)(v0 + k * (A * A - 1.0) * v1);
                    else if (n.mode == 0) out[i] = (float)(v0 + k * (A - 1.0) * v1 + (A * A - 1.0) * v2);
                    else out[i] = (float)(A * A * v0 + k * (1.0 - A) * A * v1 + (1.0 - A * A) * v2);
                }
                break;
            case K_TAPIN:                                                                                 // Feedback.h:40-53
                if (n.tapPending) { n.tapShared = n.pendingTap; n.tapPending = false; }
                if (!n.tapShared) { zero(); break; }
                for (size_t i = 0; i < N; ++i) out[i] = (*n.tapShared)[i];
                break;
            case K_TAPOUT:                                                                                // Feedback.h:111-126
                if (nIn < 1 || N > n.tapPrivate.size()) { zero(); break; }
                for (size_t i = 0; i < N; ++i) { n.tapPrivate[i] = in[0][i]; out[i] = in[0][i]; }
                break;
            case K_BLEPSAW: case K_BLEPSQUARE: case K_BLEPTRIANGLE: {                                     // Oscillators.h:41-94
                if (nIn < 1) { zero(); break; }
                for (size_t i = 0; i < N; ++i) {
                    const float inc = in[0][i] / srF;
                    const float phase = n.f0;
                    float y;
                    if (n.kind == K_BLEPSAW) y = 2.0f * phase - 1.0f - blep(phase, inc);
                    else {
                        const float naive = phase < 0.5f ? 1.0f : -1.0f;
                        const float halfPhase = std::fmod(phase + 0.5f, 1.0f);
                        const float square = naive + blep(phase, inc) - blep(halfPhase, inc);
                        if (n.kind == K_BLEPSQUARE) y = square;
                        else { n.f1 += 4.0f * inc * square; y = n.f1; }
                    }
                    n.f0 += inc;
                    if (n.f0 >= 1.0f) n.f0 -= 1.0f;
                    out[i] = y;
                }
                break;
            }
            case K_SAMPLESEQ: {                                                                           // SampleSeq.h:283-379
                const double dur = n.sampleDuration;
                if (dur != n.rtSampleDuration) { n.readers[0].reset(dur); n.readers[1].reset(dur); n.rtSampleDuration = dur; }
                if (n.samplePending) { n.sampleBuf = n.pendingSampleBuf; n.sampleLen = n.pendingSampleLen; n.samplePending = false; n.readers[0].reset(dur); n.readers[1].reset(dur); }
                if (n.pendingEvents) { n.seqEvents.swap(n.newSeqEvents); n.pendingEvents = false; n.haveEvents = true; n.prevEvent = n.nextEvent = -1; }
                if (nIn < 1 || !n.haveEvents || n.seqEvents.empty() || !n.sampleBuf || dur <= 0.0) { zero(); break; }
                const auto& ev = n.seqEvents;
                const double t = (double)in[0][0];
                const bool update = (n.prevEvent < 0 && n.nextEvent < 0)
                    || (n.prevEvent >= 0 && t <= ev[(size_t)n.prevEvent].first + 1e-6)
                    || (n.nextEvent >= 0 && t >= ev[(size_t)n.nextEvent].first - 1e-6);
                if (update || !n.readers[n.activeReader].aligned(t)) {                                    // updateEventBoundaries :257-281
                    size_t ub = 0;
                    while (ub < ev.size() && !(ev[ub].first > t)) ++ub;                                   // upper_bound
                    n.nextEvent = ub < ev.size() ? (int)ub : -1;
                    if (ub == 0) { n.prevEvent = -1; n.readers[0].setTarget(0.0f); n.readers[1].setTarget(0.0f); }
                    else {
                        n.prevEvent = (int)ub - 1;
                        n.readers[n.activeReader].setTarget(0.0f);
                        n.activeReader = (n.activeReader + 1) & 1;
                        if (std::abs(ev[(size_t)n.prevEvent].second - 1.0f) <= 1e-6f)
                            n.readers[n.activeReader].engage(ev[(size_t)n.prevEvent].first, t, n.sampleBuf->data(), n.sampleBufSize());
                    }
                }
                zero();
                n.readers[0].readAdding(out, N);
                n.readers[1].readAdding(out, N);
                break;
            }
            case K_SCOPE: {                                                                               // Analyzers.h:175-190, MultiChannelRingBuffer.h:34-59
                if (nIn < 1) { zero(); break; }
                std::copy_n(in[0], N, out);
                const size_t cap = 8192, mask = cap - 1;
                if (n.scopeRing.empty()) n.scopeRing.assign(4 * cap, 0.0f);
                const size_t w = n.scopeW, r = n.scopeR;
                const size_t freeSlots = r > w ? r - w : cap - (w - r);
                const size_t nw = (w + N) & mask;
                for (size_t ch = 0; ch < std::min<size_t>(4, nIn); ++ch)
                    for (size_t i = 0; i < N; ++i) n.scopeRing[ch * cap + ((w + i) & mask)] = in[ch][i];
                n.scopeW = nw;
                n.scopeR = N >= freeSlots ? ((nw + 1) & mask) : r;
                break;
            }
            case K_SPARSEQ: {                                                                             // SparSeq.h:201-335
                const int32_t offset = (int32_t)n.seqOffset;
                auto getTickTime = [&]() -> int32_t {                                                     // :154-199
                    int32_t tickTime = offset + n.edgeCount;
                    const int32_t ls = n.loopStart, le = n.loopEnd;
                    if (ls > -1 && le > -1 && tickTime >= le) {
                        const int32_t dur = le - ls;
                        if (dur > 0) {
                            if (n.pendingLoop) {
                                n.loopStart = n.pendLoopStart; n.loopEnd = n.pendLoopEnd; n.pendingLoop = false;
                                const int32_t nls = n.loopStart, nle = n.loopEnd;
                                if (nls == -1 && nle == -1) return tickTime;
                                tickTime = nls + ((nle - nls) != 0 ? (tickTime - le) % (nle - nls) : 0);   // (% 0: undefined in the reference)
                            } else {
                                tickTime = ls + ((tickTime - le) % dur);
                            }
                            n.edgeCount = tickTime - offset;
                        }
                    }
                    return tickTime;
                };
                auto findTickValue = [&](int32_t tickTime) {                                              // :133-152
                    auto& m = n.tickSeq;
                    if (m.empty()) { n.holdValid = false; return; }
                    auto it2 = m.upper_bound(tickTime);
                    if (it2 == m.begin()) { if (it2->first == 0) { n.holdValid = true; n.holdKey = 0; } else n.holdValid = false; return; }
                    --it2; n.holdValid = true; n.holdKey = it2->first;
                };
                int32_t tickTime = getTickTime();
                if (n.pendingTickSeq || n.loopEvent) {
                    if (n.pendingTickSeq) { n.tickSeq.swap(n.newTickSeq); n.pendingTickSeq = false; n.haveTickSeq = true; }
                    if (n.loopEvent) { n.pendingLoop = true; n.pendLoopStart = n.newLoopStart; n.pendLoopEnd = n.newLoopEnd; n.loopEvent = false; }
                    if (n.haveTickSeq) findTickValue(tickTime); else n.holdValid = false;
                }
                if (n.pendingLoop && ((n.loopStart == -1 && n.loopEnd == -1) || !n.follow)) {
                    n.loopStart = n.pendLoopStart; n.loopEnd = n.pendLoopEnd; n.pendingLoop = false;
                    tickTime = getTickTime();
                }
                if (nIn < 1 || !n.haveTickSeq) { zero(); break; }
                const bool hasReset = nIn > 1;
                for (size_t i = 0; i < N; ++i) {
                    n.samplesSinceEdge++;
                    const float x = in[0][i], reset = hasReset ? in[1][i] : 0.0f;
                    const bool trig = changeTick(n.f1, x) > 0.5f, rst = changeTick(n.f2, reset) > 0.5f;
                    if (rst) n.edgeCount = 0;
                    if (trig) {
                        n.edgeCount = rst ? 0 : n.edgeCount + 1;
                        n.samplesSinceEdge = 0;
                        tickTime = getTickTime();
                        findTickValue(tickTime);
                    }
                    if (!n.holdValid) { out[i] = 0.0f; continue; }
                    auto hv = n.tickSeq.find(n.holdKey);
                    if (n.holdOrder == 1) {
                        auto hr = std::next(hv);
                        if (hr == n.tickSeq.end()) { out[i] = hv->second; continue; }
                        const int32_t tl = hv->first, tr = hr->first;
                        const float lv = hv->second, rv = hr->second;
                        double alpha = (double)std::max(0, tickTime - tl) / (double)(tr - tl);
                        if (n.tickInterval > 0.0) alpha += (std::min((double)n.samplesSinceEdge, n.tickInterval) / n.tickInterval) / (double)(tr - tl);
                        out[i] = (float)(lv + alpha * (rv - lv));
                    } else {
                        out[i] = hv->second;
                    }
                }
                break;
            }
            case K_CAPTURE: {                                                                             // Capture.h:21-58
                if (nIn < 2) { zero(); break; }
                std::copy_n(in[1], N, out);
                if (n.capRing.empty()) n.capRing.assign((size_t)bitceil((int)(size_t)sr), 0.0f);
                const size_t cap = n.capRing.size(), mask = cap - 1;
                for (size_t i = 0; i < N; ++i) {
                    const bool g = (bool)in[0][i];
                    const bool falling = changeTick(n.f1, in[0][i]) < -0.5f;
                    if (falling || n.capScratchSize >= 128) {                                             // MultiChannelRingBuffer::write (:34-59)
                        const size_t w = n.capW, r = n.capR, cnt = n.capScratchSize;
                        const size_t freeSlots = r > w ? r - w : cap - (w - r);
                        const size_t nw = (w + cnt) & mask;
                        for (size_t k = 0; k < cnt; ++k) n.capRing[(w + k) & mask] = n.capScratch[k];
                        n.capW = nw;
                        n.capR = cnt >= freeSlots ? ((nw + 1) & mask) : r;
                        n.capScratchSize = 0;
                        if (falling) n.capReady = true;
                    }
                    if (g) n.capScratch[n.capScratchSize++] = in[1][i];
                }
                break;
            }
            case K_METER: {                                                                               // Analyzers.h:23-41
                if (nIn < 1) { zero(); break; }
                std::copy_n(in[0], N, out);
                const auto mm = std::minmax_element(in[0], in[0] + N);
                n.roMin = *mm.first; n.roMax = *mm.second; n.haveReadout = true;
                break;
            }
            case K_SNAPSHOT: {                                                                            // Analyzers.h:83-110
                if (nIn < 2) { zero(); break; }
                const float eps = std::numeric_limits<float>::epsilon();
                for (size_t i = 0; i < N; ++i) {
                    const float l = in[0][i], x = in[1][i];
                    if (std::abs(n.f0) <= eps && l > eps) { n.roVal = x; n.haveReadout = true; }
                    n.f0 = l;
                    out[i] = x;
                }
                break;
            }
            case K_SAMPLE: {                                                                              // Sample.h:82-139, 178-221
                if (n.samplePending) {
                    n.sampleBuf = n.pendingSampleBuf; n.sampleLen = n.pendingSampleLen; n.samplePending = false;
                    for (auto& rd : n.lerp) { rd = Node::LerpReader(); rd.hasBuffer = true; }
                }
                if (nIn < 1 || !n.sampleBuf) { zero(); break; }
                const float alpha = (float)(1.0 - std::exp(-1.0 / (0.01 * (double)(float)sr)));           // :163, sampleRate is FloatType
                const bool loop = n.sampleMode == 2;
                const size_t len = n.sampleLen, ostart = n.startOffset, ostop = n.stopOffset;
                const float* data = n.sampleBuf->data();
                auto tick = [&](Node::LerpReader& rd, float step) -> float {
                    if (!rd.hasBuffer || rd.pos < 0.0 || (rd.gain == 0.0f && rd.targetGain == 0.0f)) return 0.0f;
                    if (rd.pos >= (double)(len - ostop)) { if (!loop) return 0.0f; rd.pos = (double)ostart; }
                    size_t left = (size_t)rd.pos, right = left + 1;
                    const float frac = (float)(rd.pos - (double)left);
                    if (left >= len) left -= len;
                    if (right >= len) right -= len;
                    if (len) { left %= len; right %= len; }   // out-of-bounds reads in the reference (start offset beyond the buffer)
                    const float out = rd.gain * (data[left] + frac * (data[right] - data[left]));
                    const bool settled = std::abs(rd.targetGain - rd.gain) <= std::numeric_limits<float>::epsilon();
                    rd.pos = rd.pos + (double)step;
                    rd.gain = settled ? rd.targetGain : rd.gain + alpha * (rd.targetGain - rd.gain);
                    rd.gain = clampf(rd.gain, 0.0f, 1.0f);
                    return out;
                };
                const bool hasRate = nIn >= 2;
                for (size_t i = 0; i < N; ++i) {
                    const float cv = changeTick(n.f1, in[0][i]);
                    const float rate = hasRate ? in[1][i] : 1.0f;
                    if (cv > 0.5f) {
                        n.lerp[n.currentReader & 1].targetGain = 0.0f;
                        auto& nr = n.lerp[++n.currentReader & 1];
                        nr.targetGain = 1.0f; nr.pos = (double)(float)ostart;                              // pos = FloatType(startOffset) with ReaderType's float
                    }
                    if (cv < -0.5f && n.sampleMode != 0) n.lerp[n.currentReader & 1].targetGain = 0.0f;
                    const float a = tick(n.lerp[0], rate);
                    out[i] = a + tick(n.lerp[1], rate);
                }
                break;
            }
            case K_TABLE: {                                                                               // Table.h:35-71
                if (n.samplePending) { n.sampleBuf = n.pendingSampleBuf; n.sampleLen = n.pendingSampleLen; n.samplePending = false; }
                const int size = (int)n.sampleLen;
                if (nIn == 0 || !n.sampleBuf || size == 0) { zero(); break; }
                const float* buf = n.sampleBuf->data();
                for (size_t i = 0; i < N; ++i) {
                    const float readPos = clampf(in[0][i], 0.0f, 1.0f) * (float)(size - 1);
                    const int readLeft = (int)readPos, readRight = readLeft + 1;
                    const float frac = readPos - std::floor(readPos);
                    const float left = buf[readLeft % size], right = buf[readRight % size];
                    out[i] = left + frac * (right - left);
                }
                break;
            }
            case K_SEQ2: {                                                                                // Seq2.h:87-147
                if (n.pendingSeq) { n.seq.swap(n.newSeq); n.pendingSeq = false; n.haveSeq = true; }
                if (nIn < 1 || !n.haveSeq) { zero(); break; }
                const bool hasReset = nIn > 1;
                const size_t len = n.seq.size();
                for (size_t i = 0; i < N; ++i) {
                    const float x = in[0][i], reset = hasReset ? in[1][i] : 0.0f;
                    if (changeTick(n.f1, x) > 0.5f) n.seqIndex++;          // seqIndex holds edgeCount here
                    if (changeTick(n.f2, reset) > 0.5f) n.seqIndex = 0;
                    const size_t idx = n.seqOffset + n.seqIndex;
                    // an empty sequence is UB in the reference (% 0, at(size - 1)); it emits 0 here
                    const float next = len == 0 ? 0.0f : (idx < len) ? n.seq[idx] : (n.loop ? n.seq[idx % len] : (n.hold ? n.seq[len - 1] : 0.0f));
                    out[i] = n.hold ? next : next * x;
                }
                break;
            }
            case K_SPARSEQ2: {                                                                            // SparSeq2.h:68-127
                if (n.pendingEvents) { n.seqEvents.swap(n.newSeqEvents); n.pendingEvents = false; n.haveEvents = true; n.prevEvent = n.nextEvent = -1; }
                if (nIn < 1 || !n.haveEvents || n.seqEvents.empty()) { zero(); break; }
                const auto& ev = n.seqEvents;
                const bool interp = n.interp == 1;
                for (size_t i = 0; i < N; ++i) {
                    const double t = (double)in[0][i];
                    const bool update = (n.prevEvent < 0 && n.nextEvent < 0)
                        || (n.prevEvent >= 0 && t <= ev[(size_t)n.prevEvent].first + 1e-9)
                        || (n.nextEvent >= 0 && t >= ev[(size_t)n.nextEvent].first - 1e-9);
                    if (update) {                                                                         // :56-66
                        size_t ub = 0;
                        while (ub < ev.size() && !(ev[ub].first > t)) ++ub;
                        n.nextEvent = ub < ev.size() ? (int)ub : -1;
                        n.prevEvent = ub == 0 ? -1 : (int)ub - 1;
                    }
                    if (n.prevEvent < 0) { out[i] = 0.0f; continue; }
                    if (n.nextEvent < 0) { out[i] = ev[(size_t)n.prevEvent].second; continue; }
                    const auto& p = ev[(size_t)n.prevEvent]; const auto& q = ev[(size_t)n.nextEvent];
                    const double alpha = interp ? ((t - p.first) / (q.first - p.first)) : 0.0;
                    out[i] = p.second + (float)alpha * (q.second - p.second);
                }
                break;
            }
            case K_CONVOLVE:                                                                              // wasm/Convolve.h:58-84
                if (n.pendingConvolver) { n.convolver = n.pendingConvolver; n.pendingConvolver.reset(); }
                if (nIn == 0 || !n.convolver) { zero(); break; }
                n.convolver->process(in[0], out, N);
                break;
            case K_TIME:                                                                                  // wasm/SampleTime.h:14-22
                for (size_t i = 0; i < N; ++i) out[i] = (float)(double)((uint64_t)sampleTime + (uint64_t)i);
                break;
            case K_METRO: {                                                                               // wasm/Metro.h:36-55
                const double is = (double)n.interval;
                for (size_t i = 0; i < N; ++i) {
                    const double t = (double)((uint64_t)sampleTime + (uint64_t)i) / is;
                    out[i] = (float)((t - std::floor(t)) < 0.5);
                }
                break;
            }
        }
    }

    // ---- Runtime::processQueuedEvents -> RootRenderSequence::processQueuedEvents (Runtime.h:437-446, GraphRenderSequence.h:189-198)
    void events(void (*cb)(const char*, const char*, void*), void* user) {
        if (!current) return;
        auto numStr = [](float v) { char b[64]; std::snprintf(b, sizeof b, "%.17g", (double)v); return std::string(b); };
        for (RootSeq& rs : current->roots) {
            Node& root = nodes.at(rs.root);
            auto a = root.props.find("active");
            if (a == root.props.end() || a->second.t != JV::Bool || !a->second.b) continue;
            for (int32_t id : rs.order) {
                Node& n = nodes.at(id);
                if (n.kind == K_SCOPE) {                                                                  // Analyzers.h:192-245
                    auto numOr = [&](const char* k, double d) { auto q = n.props.find(k); return (q != n.props.end() && q->second.t == JV::Num) ? q->second.n : d; };
                    const size_t size = (size_t)numOr("size", 512.0), channels = (size_t)numOr("channels", 1.0), cap = 8192, mask = cap - 1;
                    const size_t w = n.scopeW, r = n.scopeR;
                    const size_t full = w > r ? w - r : ((cap - (r - w)) & mask);
                    if (!(full > size) || n.scopeRing.empty()) continue;
                    std::string src = "null";
                    auto nm = n.props.find("name");
                    if (nm != n.props.end() && nm->second.t == JV::Str) { src = "\""; for (char ch : nm->second.s) { if (ch == '"' || ch == '\\') src += '\\'; src += ch; } src += "\""; }
                    std::string j = "{\"source\": " + src + ", \"data\": [";
                    for (size_t ch = 0; ch < channels; ++ch) {
                        j += ch ? ", [" : "[";
                        for (size_t i = 0; i < size; ++i) { if (i) j += ", "; j += numStr(n.scopeRing[ch * cap + ((r + i) & mask)]); }
                        j += "]";
                    }
                    j += "]}";
                    n.scopeR = (r + size) & mask;
                    cb("scope", j.c_str(), user);
                    continue;
                }
                if (n.kind == K_CAPTURE) {                                                                // Capture.h:60-95
                    if (!n.capRing.empty()) {
                        const size_t cap = n.capRing.size(), mask = cap - 1, w = n.capW, r = n.capR;
                        const size_t avail = w > r ? w - r : ((cap - (r - w)) & mask);
                        for (size_t k = 0; k < avail; ++k) n.capRelay.push_back(n.capRing[(r + k) & mask]);
                        n.capR = (r + avail) & mask;
                    }
                    if (n.capReady) {
                        n.capReady = false;
                        std::string src = "null";
                        auto nm = n.props.find("name");
                        if (nm != n.props.end() && nm->second.t == JV::Str) { src = "\""; for (char ch : nm->second.s) { if (ch == '"' || ch == '\\') src += '\\'; src += ch; } src += "\""; }
                        std::string j = "{\"source\": " + src + ", \"data\": [";
                        for (size_t k = 0; k < n.capRelay.size(); ++k) { if (k) j += ", "; j += numStr(n.capRelay[k]); }
                        j += "]}";
                        n.capRelay.clear();
                        cb("capture", j.c_str(), user);
                    }
                    continue;
                }
                if ((n.kind != K_METER && n.kind != K_SNAPSHOT) || !n.haveReadout) continue;
                n.haveReadout = false;
                std::string src = "null";
                auto nm = n.props.find("name");
                if (nm != n.props.end() && nm->second.t == JV::Str) {
                    src = "\"";
                    for (char ch : nm->second.s) { if (ch == '"' || ch == '\\') src += '\\'; src += ch; }
                    src += "\"";
                }
                if (n.kind == K_METER) {                                                                  // Analyzers.h:43-62
                    const std::string j = "{\"min\": " + numStr(n.roMin) + ", \"max\": " + numStr(n.roMax) + ", \"source\": " + src + "}";
                    cb("meter", j.c_str(), user);
                } else {                                                                                  // Analyzers.h:112-131
                    const std::string j = "{\"source\": " + src + ", \"data\": " + numStr(n.roVal) + "}";
                    cb("snapshot", j.c_str(), user);
                }
            }
        }
    }

    // ---- Runtime::process -> GraphRenderSequence::process (Runtime.h:274-290, GraphRenderSequence.h:212-309)
    int process(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t N, int64_t sampleTime) {
        if (pending) { current = pending; pending.reset(); }
        if (!current) return 0;
        if (N > (size_t)bs) return 102;
        for (size_t c = 0; c < nOut; ++c) std::fill_n(out[c], N, 0.0f);
        std::vector<const float*> ptrs;
        for (RootSeq& rs : current->roots) {
            Node& root = nodes.at(rs.root);
            const int ch = root.rootChannel;
            if (!stillRunning(root) || ch < 0 || (size_t)ch >= nOut) continue;
            for (int32_t id : rs.order) {
                Node& n = nodes.at(id);
                if (n.inlets.empty()) { processNode(n, in, nIn, N, sampleTime); continue; }   // leaf: host inputs (:126-136)
                ptrs.clear();
                for (auto& e : n.inlets) {
                    static const std::vector<float> zeros(4096, 0.0f);
                    auto s = nodes.find(e.first);
                    ptrs.push_back((s != nodes.end() && e.second == 0) ? s->second.out.data() : zeros.data());
                }
                processNode(n, ptrs.data(), ptrs.size(), N, sampleTime);
            }
            for (size_t j = 0; j < N; ++j) out[ch][j] += root.out[j];
        }
        for (RootSeq& rs : current->roots) {   // promoteTapBuffers (:200-210, Feedback.h:90-109)
            if (!(nodes.at(rs.root).target > 0.5f)) continue;
            for (int32_t id : rs.order) {
                Node& n = nodes.at(id);
                if (n.kind != K_TAPOUT) continue;
                if (n.tapPending) { n.tapShared = n.pendingTap; n.tapPending = false; }
                if (n.tapShared) std::copy_n(n.tapPrivate.data(), N, n.tapShared->data());
            }
        }
        return 0;
    }

    size_t gc(int32_t* out, size_t cap) {   // Runtime.h:220-272
        std::vector<int32_t> pruned;
        for (auto& kv : nodes) {
            const bool held = (current && current->ids.count(kv.first)) || (pending && pending->ids.count(kv.first));
            if (!held) pruned.push_back(kv.first);
        }
        std::set<int32_t> ps(pruned.begin(), pruned.end());
        for (int32_t id : pruned)
            for (auto& e : nodes.at(id).inlets) {
                auto c = nodes.find(e.first);
                if (c == nodes.end() || ps.count(e.first)) continue;
                auto& o = c->second.outlets;
                o.erase(std::remove_if(o.begin(), o.end(), [&](auto& x) { return x.first == id; }), o.end());
            }
        for (int32_t id : pruned) nodes.erase(id);
        std::sort(pruned.begin(), pruned.end());
        size_t k = 0;
        for (int32_t id : pruned) { if (out && k < cap) out[k] = id; ++k; }
        return k;
    }
};

} // namespace

extern "C" {

void* elemoracle_create(double sampleRate, int blockSize) { return new Oracle(sampleRate, blockSize); }
void elemoracle_destroy(void* h) { delete static_cast<Oracle*>(h); }
int elemoracle_apply_instructions_json(void* h, const char* json, size_t len) {
    JV v; JP p{json, json + len};
    if (!p.val(v)) return 105;
    return static_cast<Oracle*>(h)->apply(v);
}
int elemoracle_process(void* h, const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n, int64_t st) {
    return static_cast<Oracle*>(h)->process(in, nIn, out, nOut, n, st);
}
int elemoracle_add_shared_resource(void* h, const char* name, const float* const* ch, size_t nCh, size_t nSamples) {
    auto* o = static_cast<Oracle*>(h);
    if (o->resources.count(name)) return 0;   // insert-only (SharedResource.h:61-63)
    auto r = std::make_shared<std::vector<float>>(nCh ? std::vector<float>(ch[0], ch[0] + nSamples) : std::vector<float>());
    o->resourceLen[name] = r->size();
    if (r->size() < (size_t)o->bs) r->resize((size_t)o->bs, 0.0f);
    o->resources.emplace(name, r);
    return 1;
}
void elemoracle_prune_shared_resources(void* h) {   // SharedResource.h:94-102
    auto* o = static_cast<Oracle*>(h);
    // holders: tapIn / tapOut nodes
    for (auto it = o->resources.begin(); it != o->resources.end();) { if (it->second.use_count() == 1) it = o->resources.erase(it); else ++it; }
}
size_t elemoracle_gc(void* h, int32_t* out, size_t cap) { return static_cast<Oracle*>(h)->gc(out, cap); }
int elemoracle_process_queued_events(void* h, void (*cb)(const char*, const char*, void*), void* user) {
    static_cast<Oracle*>(h)->events(cb, user); return 0;
}
void elemoracle_reset(void*) {}   // Runtime.h:448-458: only SampleNode (out of scope) reacts

} // extern "C"
