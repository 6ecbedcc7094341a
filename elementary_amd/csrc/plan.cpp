// plan.cpp — render-sequence build: node table -> islands -> device task tables.
//
// Step 1 restates Runtime::buildRenderSequence / traverse (runtime/elem/Runtime.h:502-577):
// roots sorted active-first, DFS post-order per root, children in inlet order, `visited` shared
// across roots.  Any topological order yields identical samples (buffers are never aliased,
// GraphRenderSequence.h:121-124); the order still decides which root's sequence OWNS a shared
// node, and therefore which nodes stop rendering when an inactive root finishes its fade
// (GraphRenderSequence.h:212-219) and in which order tap buffers are promoted.
//
// Steps 2-5 are new: cluster each root sequence into islands (one workgroup, buffers in LDS),
// schedule each island's nodes into barrier-separated stages of wave tasks, allocate LDS slots
// by liveness, and order islands into launch levels.
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <cstdio>
#include <functional>
#include <numeric>
#include <unordered_set>

#include <hip/hip_runtime.h>

#include "engine.h"

namespace elemhip {

namespace {

enum Kind : uint8_t { K_CONST, K_PAR, K_SINGLE, K_CHAIN, K_CONV, K_HOST };

Kind kindOf(uint16_t op) {
    switch (op) {
        case OP_CONST: case OP_SR: return K_CONST;
        case OP_RAND: case OP_Z: case OP_SDELAY: case OP_DELAY: case OP_SAMPLESEQ: case OP_METER: case OP_SNAPSHOT: case OP_SCOPE: case OP_CAPTURE: return K_SINGLE;
        case OP_CONVOLVE: return K_CONV;   // always an island of its own, rendered by conv.hip
        case OP_HOST: return K_HOST;       // always an island of its own, rendered on the CPU between launch levels
        case OP_PHASOR: case OP_SPHASOR: case OP_COUNTER: case OP_ACCUM: case OP_LATCH: case OP_MAXHOLD:
        case OP_ONCE: case OP_SEQ: case OP_SEQ2: case OP_SPARSEQ: case OP_SAMPLE: case OP_MCSAMPLE: case OP_POLE: case OP_ENV: case OP_BIQUAD: case OP_MM1P: case OP_SVF:
        case OP_SVFSHELF: case OP_BLEPSAW: case OP_BLEPSQUARE: case OP_BLEPTRIANGLE: case OP_PHASE:
            return K_CHAIN;
        default: return K_PAR;
    }
}

uint32_t scratchSlots(uint16_t op) {
    switch (op) {
        case OP_SVF: return 6;        // a1,a2,a3 as double (coefficient pre-pass -> scan)
        case OP_SVFSHELF: return 10;  // a1,a2,a3,k,A
        case OP_DELAY: return 1;
        case OP_SAMPLESEQ: return 2;  // per-reader fade gains
        case OP_SAMPLE: case OP_MCSAMPLE: return 6;     // per reader: read index, fraction, gain (serial pass -> gather pass)
        default: return 0;
    }
}

// inputs a leaf node of this type would read from the host channels (its reference arity)
uint32_t leafArity(uint16_t op) {
    switch (op) {
        case OP_PHASOR: case OP_COUNTER: case OP_ONCE: case OP_BLEPSAW: case OP_BLEPSQUARE: case OP_BLEPTRIANGLE:
        case OP_Z: case OP_SDELAY: case OP_PREWARP: case OP_ROOT: case OP_TAPOUT: case OP_SAMPLESEQ: case OP_MCSAMPLE: return 1;
        case OP_SPHASOR: case OP_ACCUM: case OP_LATCH: case OP_MAXHOLD: case OP_SEQ: case OP_SEQ2: case OP_SPARSEQ: case OP_CAPTURE: case OP_SAMPLE: case OP_POLE: case OP_MM1P: case OP_SNAPSHOT: return 2;
        case OP_ENV: case OP_SVF: case OP_DELAY: return 3;
        case OP_SCOPE: return 4;
        case OP_SVFSHELF: return 4;
        case OP_BIQUAD: return 6;
        case OP_LE: case OP_LEQ: case OP_GE: case OP_GEQ: case OP_POW: case OP_EQ: case OP_AND: case OP_OR: return 2;
        case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: case OP_MOD: case OP_MIN: case OP_MAX: case OP_IN: return kMaxHostIn;
        case OP_CONST: case OP_SR: case OP_RAND: case OP_TIME: case OP_METRO: case OP_TAPIN: return 0;
        default: return 1;   // unary math
    }
}

// blepsaw / blepsquare take two stages: the phase recurrence (one lane, serial), then the waveform as a sample-parallel
// task (OP_SAW_SHAPE / OP_SQUARE_SHAPE) that any wave can run — the recurrence is the longest serial item of a synth voice
// and bounds a pipelined island's block rate, so nothing else rides on its wave.
bool blepSplit(uint16_t op) { return op == OP_BLEPSAW || op == OP_BLEPSQUARE; }

uint32_t leafArityOfOp(uint16_t op);
} // namespace
uint32_t leafArityForCodegen(uint16_t op) { return leafArityOfOp(op); }
std::string emitSpecSource(const Island& I, const std::vector<Task>& tasks, const SpecProgram& sp,
                           const std::vector<uint32_t>& stageTab, uint32_t blockSize, uint32_t wavesPerEu);   // codegen.cpp
namespace {
uint32_t leafArityOfOp(uint16_t op) {
    if (op == OP_SAW_SHAPE || op == OP_SQUARE_SHAPE || op == OP_PHASE) return 1;
    if (op == OP_SVF_COEF) return leafArity(OP_SVF);
    if (op == OP_SHELF_COEF) return leafArity(OP_SVFSHELF);
    return leafArity(op);
}

// estimated shader cycles of one task on a lone wave (measured on the pipelined C2 voice island, tests/_trace.py pipe32)
constexpr uint32_t kSlotGap = 2500u;   // hand-over into a (stage, wave) slot: publish, poll, acquire
uint32_t taskCost(uint16_t op, uint32_t units, uint32_t count) {
    const uint32_t gap = 900u;                                                              // header decode + dispatch around every task
    if (op == OP_SVF_COEF || op == OP_SHELF_COEF) return gap + 2300u * units * count;      // double tan + divides per frame
    if (op == OP_SVF || op == OP_SVFSHELF || op == OP_MM1P) return gap + 9500u * count;    // wave scan
    if (op == OP_SAW_SHAPE || op == OP_SQUARE_SHAPE) return gap + 500u * units * count;
    if (op == OP_BLEPSAW || op == OP_BLEPSQUARE || op == OP_PHASE) return gap + 12500u;     // phase recurrence only
    if (op == OP_BLEPTRIANGLE) return gap + 30000u;
    if (op == OP_POLE || op == OP_ENV || op == OP_BIQUAD) return gap + 15300u;
    if (kindOf(op) == K_CHAIN) return gap + 12000u;
    if (kindOf(op) == K_SINGLE) return gap + 4000u * count;
    if (op == OP_ROOT) return gap + 1000u + 250u * units * count;
    if (op >= OP_SIN && op <= OP_EXP) return gap + 900u + 280u * units * count;            // tanh: 3.1 k for a whole block
    return gap + 900u + 80u * units * count;                                                // light op: 1.5 k for a whole block
}

struct NI {                      // per-node planning info
    Node* n = nullptr;
    int seq = 0;                 // owning root sequence
    int pos = 0;                 // position in the global render order
    Kind kind = K_PAR;
    int island = -1;
    int level = 0;               // stage inside the island
    int sub = 0;                 // depth inside a fused run of sample-parallel ops of one stage
    bool needLds = false;
    bool exported = false;
    uint32_t lds = kNone;        // LDS word offset of the output slot
    uint32_t hbm = kNone;        // HBM arena index
    uint32_t scratch = kNone;
    int lastUse = 0;             // last in-island consumer stage
    int fusedRoot = -1;          // convolve: NI index of the root whose gain this node applies itself (the root has no task)
    bool elided = false;         // `in` leaf read directly by convolvers / root folded into its convolver: never a task
    uint32_t ch = 0;             // output channel of a multi-output node this entry renders
    uint32_t rec = kNone;        // node record (a multi-output node has one per channel)
};

struct IslandBuild {
    std::vector<int> nodes;      // NI indices in render order
    int seq = 0;
    int level = 0;               // launch level
    std::vector<int> deps;       // islands it imports from
};

// (node id, channel) -> NI index: open addressing, sized once per build. The planner asks this table about every inlet in every
// phase (~20 lookups per node and build); a node-per-entry std::unordered_map made the render-order phase allocation-bound.
struct FlatIdx {
    struct Slot { int64_t first; int second; };
    std::vector<Slot> tab;
    uint64_t mask = 0;
    static constexpr int64_t kEmpty = INT64_MIN;
    static uint64_t hash(int64_t k) { uint64_t h = (uint64_t)k * 0x9E3779B97F4A7C15ull; return h ^ (h >> 29); }
    void reserve(size_t n) { size_t cap = 64; while (cap < 2 * n) cap <<= 1; tab.assign(cap, Slot{kEmpty, 0}); mask = cap - 1; used = 0; }
    size_t used = 0;
    int& operator[](int64_t k) {
        if (2 * (used + 1) > tab.size()) {   // (multi-output nodes add an entry per channel: grow, keep the load under one half)
            std::vector<Slot> old;
            old.swap(tab);
            tab.assign(std::max<size_t>(64, 2 * old.size()), Slot{kEmpty, 0}); mask = tab.size() - 1; used = 0;
            for (const Slot& o : old) if (o.first != kEmpty) (*this)[o.first] = o.second;
        }
        for (uint64_t i = hash(k) & mask;; i = (i + 1) & mask) {
            if (tab[i].first == k) return tab[i].second;
            if (tab[i].first == kEmpty) { tab[i].first = k; ++used; return tab[i].second; }
        }
    }
    const Slot* find(int64_t k) const {
        if (tab.empty()) return nullptr;
        for (uint64_t i = hash(k) & mask;; i = (i + 1) & mask) {
            if (tab[i].first == k) return &tab[i];
            if (tab[i].first == kEmpty) return nullptr;
        }
    }
    const Slot* end() const { return nullptr; }
    size_t count(int64_t k) const { return find(k) ? 1 : 0; }
    int at(int64_t k) const { const Slot* s = find(k); if (!s) throw std::out_of_range("plan index"); return s->second; }
};

struct UF {
    std::vector<int> p;
    int find(int x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
    int add() { p.push_back((int)p.size()); return (int)p.size() - 1; }
};

} // namespace

struct PlanBuilder {
    Engine& e;
    explicit PlanBuilder(Engine& eng) : e(eng) {}

    std::vector<NI> ni;
    // (node id, output channel) -> NI index. Multi-output nodes (mc.*, GraphRenderSequence.h:15-24) are planned as one
    // single-output entry per channel; every other node only has channel 0, so an inlet that names another channel of it
    // finds nothing and reads as a missing input, like before.
    FlatIdx idx;
    static int64_t K(int32_t id, uint32_t ch = 0) { return ((int64_t)id << 8) | (int64_t)(ch & 0xFFu); }
    std::vector<std::vector<int>> seqNodes; // per root sequence
    std::vector<Node*> seqRoots;

    // (node, channel) of an inlet -> planner entry, through the memo the render-order walk left in the inlet (no hashing): what
    // srcOf(in) answers, for the ~20 questions per inlet the phases ask
    struct Hit {
        int second; bool ok;
        const Hit* operator->() const { return this; }
        bool operator==(const FlatIdx::Slot* p) const { return p == nullptr && !ok; }
        bool operator!=(const FlatIdx::Slot* p) const { return !(*this == p); }
    };
    uint32_t buildEpoch = 0;
    Hit srcOf(const Inlet& in) const {
        const Node* c = in.srcEpoch == e.nodesEpoch ? in.src : nullptr;
        if (c) {
            if (c->planVisited != buildEpoch || in.channel >= c->planChans) return Hit{-1, false};
            return Hit{c->planIdx + (int)in.channel, true};
        }
        const FlatIdx::Slot* s = idx.find(K(in.source, in.channel));      // (an inlet of a node the walk did not reach, a missing source)
        return s ? Hit{s->second, true} : Hit{-1, false};
    }

    void traverse(uint32_t epoch, std::vector<Node*>& order, Node* root) {
        // iterative DFS post-order, children in inlet order (Runtime.h:502-518); visited / on-stack are epoch marks in the nodes
        struct Frame { Node* n; size_t next; };
        if (root->planVisited == epoch) return;
        std::vector<Frame> st;
        st.push_back({root, 0});
        root->planOnStack = epoch;
        while (!st.empty()) {
            Frame& f = st.back();
            Node& n = *f.n;
            if (f.next < n.inlets.size()) {
                const Inlet& in = n.inlets[f.next++];
                Node* c = in.srcEpoch == e.nodesEpoch ? in.src : nullptr;
                if (!c) {                                    // (misses are not remembered: the node may be created later)
                    auto it = e.nodes.find(in.source);
                    if (it == e.nodes.end()) continue;
                    c = &it->second; in.src = c; in.srcEpoch = e.nodesEpoch;
                }
                if (c->planVisited == epoch || c->planOnStack == epoch) continue;
                c->planOnStack = epoch;
                st.push_back({c, 0});
            } else {
                order.push_back(f.n);
                n.planVisited = epoch;
                n.planOnStack = 0;
                st.pop_back();
            }
        }
    }

    bool splitCoefStage = false;
    bool wantSpec = false;                  // also write the specialised-kernel text of every pipelined island
    uint32_t packK = 1;                     // merge up to this many same-shape islands of a launch level into one (lane-packing); 0 = as many as it takes
    uint32_t packMax = 2, cuCount = 256;    // ... to bring the fullest launch level down to the CU count, at most packMax
    bool packRoots = false;                 // option "pack_roots": merge across root sequences (active roots only)
    uint32_t packedIslands = 0;             // out: islands that disappeared into another
    uint32_t minPackedCopies = 0;           // out: fewest buffer sets of an island that carries more than one original island
    uint32_t statefulIslandsMax = 0;        // out: most stateful islands of one launch level (before packing)
    std::shared_ptr<Plan> build(uint32_t maxIslandNodes, uint32_t maxCopies);
};

std::shared_ptr<Plan> PlanBuilder::build(uint32_t maxIslandNodes, uint32_t maxCopies) {
    auto plan = std::make_shared<Plan>();
    Plan& p = *plan;
    const uint32_t bs = (uint32_t)e.blockSize;

    static const bool planTiming = std::getenv("ELEMHIP_PLAN_TIMING") != nullptr;   // phase times of a build on stderr
    auto tPhase = std::chrono::steady_clock::now();
    int phaseNo = 0;
    auto phase = [&](const char* name) {
        const auto now = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(now - tPhase).count();
        if (phaseNo < 4) p.buildUs[phaseNo++] = us;
        if (planTiming) std::fprintf(stderr, "[elemhip] plan %-18s %7.3f ms\n", name, us * 1e-3);
        tPhase = now;
    };
    // ---- 1. render order ---------------------------------------------------------------------
    std::vector<Node*> sortedRoots;   // std::list push_front/push_back in Runtime.h:544-559
    {
        std::vector<Node*> front, back;
        for (int32_t id : e.currentRoots) {
            auto it = e.nodes.find(id);
            if (it == e.nodes.end() || it->second.op != OP_ROOT) continue;
            auto a = it->second.props.find("active");
            const bool active = a != it->second.props.end() && a->second.isBool() && a->second.b;
            if (active) front.push_back(&it->second); else back.push_back(&it->second);
        }
        std::reverse(front.begin(), front.end());
        sortedRoots = front;
        sortedRoots.insert(sortedRoots.end(), back.begin(), back.end());
    }
    if (++e.planEpoch == 0u) { for (auto& kv : e.nodes) kv.second.planVisited = kv.second.planOnStack = 0u; e.planEpoch = 1u; }
    const uint32_t epoch = e.planEpoch;
    buildEpoch = epoch;
    idx.reserve(2 * e.nodes.size() + 64);   // (one entry per node and output channel; the table doubles that again)
    ni.reserve(e.nodes.size() + 16);
    std::vector<int32_t> planNodeIds;
    planNodeIds.reserve(e.nodes.size());
    for (size_t s = 0; s < sortedRoots.size(); ++s) {
        std::vector<Node*> order;
        traverse(epoch, order, sortedRoots[s]);
        seqRoots.push_back(sortedRoots[s]);
        seqNodes.emplace_back();
        for (Node* np : order) {
            Node& node = *np;
            const int32_t id = node.id;
            uint32_t numOuts = 1;                       // getRequiredOutputChannels (GraphRenderSequence.h:15-24)
            if (node.mc) for (auto& o : node.outlets) numOuts = std::max(numOuts, std::min<uint32_t>(o.channel + 1u, 16u));
            node.planIdx = (int32_t)ni.size(); node.planChans = numOuts;
            for (uint32_t ch = 0; ch < numOuts; ++ch) {
                NI x;
                x.n = &node;
                x.seq = (int)s;
                x.pos = (int)ni.size();
                x.kind = kindOf(node.op);
                x.ch = ch;
                if (ch == 0 || !node.mc) x.rec = node.rec;
                else { std::lock_guard<std::mutex> render(e.mu); x.rec = e.channelRec(node, ch); }   // (build runs without `mu`)
                idx[K(id, ch)] = (int)ni.size();
                seqNodes.back().push_back((int)ni.size());
                ni.push_back(x);
            }
            planNodeIds.push_back(id);
            if (node.op == OP_CAPTURE && node.mc) p.mcCaptureIds.push_back(id);
        }
    }
    // (no duplicates: the visit marks are shared by all root sequences, a node belongs to the first sequence that reaches it)
    p.nodeIds.swap(planNodeIds);

    // ---- 1b. nodes folded into the convolve launch ------------------------------------------------------
    // `root(convolve(in))` is the whole graph of a convolution reverb channel: three launch levels for one
    // kernel's worth of work. An `in` leaf whose only in-plan consumers are convolvers is read by them straight
    // from the host-input arena, and a root whose only input is a convolver owned by the same sequence (and
    // consumed by nothing else) has its fade applied by that convolver, which writes the root's buffer.
    for (NI& x : ni) {
        if (x.n->op != OP_IN || !x.n->inlets.empty()) continue;
        bool any = false, all = true;
        for (auto& o : x.n->outlets) {
            auto it = idx.find(K(o.dest));
            if (it == idx.end()) continue;
            any = true;
            if (ni[it->second].n->op != OP_CONVOLVE) all = false;
        }
        if (any && all) { x.kind = K_CONST; x.elided = true; }
    }
    for (size_t sq = 0; sq < seqRoots.size(); ++sq) {
        Node* r = seqRoots[sq];
        if (r->inlets.size() != 1 || r->inlets[0].channel != 0) continue;
        auto it = idx.find(K(r->inlets[0].source));
        if (it == idx.end()) continue;
        NI& c = ni[it->second];
        if (c.kind != K_CONV || c.seq != (int)sq) continue;
        size_t consumers = 0;
        for (auto& o : c.n->outlets) if (idx.count(K(o.dest))) ++consumers;
        if (consumers != 1) continue;
        NI& rn = ni[idx.at(K(r->id))];
        rn.kind = K_CONST; rn.elided = true;
        c.fusedRoot = idx.at(K(r->id));
    }

    phase("render order");
    // ---- 2. islands ---------------------------------------------------------------------------------
    UF uf;
    std::vector<uint32_t> weight;            // per island representative
    std::vector<std::vector<int>> succ;      // island DAG (by representative at insertion time; small unsorted sets)
    std::vector<int> ilevel;                 // running estimate of each island's launch level
    std::vector<uint32_t> seenMark;          // reaches(): visit marks by epoch (no hash set per query)
    uint32_t seenEpoch = 0;
    uf.p.reserve(ni.size()); weight.reserve(ni.size()); succ.reserve(ni.size()); ilevel.reserve(ni.size()); seenMark.reserve(ni.size());
    auto rep = [&](int i) { return uf.find(i); };
    auto newIsland = [&]() { int i = uf.add(); weight.push_back(0); succ.emplace_back(); ilevel.push_back(0); seenMark.push_back(0); return i; };
    using Small = std::vector<int>;          // a handful of island ids: linear search beats a tree
    auto has = [](const Small& v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };
    auto put = [&](Small& v, int x) { if (!has(v, x)) v.push_back(x); };
    auto putSucc = [&](int from, int to) { Small& v = succ[(size_t)from]; if (v.size() >= 64 || !has(v, to)) v.push_back(to); };   // (a duplicate edge is harmless; a fan-out of thousands stays linear)

    auto reaches = [&](const Small& from, const Small& targets, const Small& skip) {
        // is any island of `targets` reachable from `from` without starting inside `skip`?
        std::vector<int> stack;
        ++seenEpoch;
        auto first = [&](int r) { if (seenMark[(size_t)r] == seenEpoch) return false; seenMark[(size_t)r] = seenEpoch; return true; };
        for (int f : from) for (int s : succ[f]) { int r = rep(s); if (!has(skip, r) && first(r)) stack.push_back(r); }
        size_t budget = 20000;
        while (!stack.empty()) {
            if (budget-- == 0) return true;   // give up: treat as unsafe
            int x = stack.back(); stack.pop_back();
            if (has(targets, x)) return true;
            for (int s : succ[x]) { int r = rep(s); if (has(targets, r)) return true; if (first(r)) stack.push_back(r); }
        }
        return false;
    };

    Small deps, foreign, others;
    for (size_t k = 0; k < ni.size(); ++k) {
        NI& x = ni[k];
        if (x.kind == K_CONST) continue;
        const uint32_t w = 1 + scratchSlots(x.n->op);
        deps.clear(); foreign.clear();
        for (auto& in : x.n->inlets) {
            auto it = srcOf(in);
            if (it == idx.end()) continue;
            NI& s = ni[it->second];
            if (s.kind == K_CONST) continue;
            if (s.seq != x.seq) { put(foreign, rep(s.island)); continue; }
            put(deps, rep(s.island));
        }
        std::sort(deps.begin(), deps.end()); std::sort(foreign.begin(), foreign.end());   // (the order a std::set walked them in)
        int target = -1;
        const bool sealed = x.kind == K_CONV || x.kind == K_HOST;
        if (!deps.empty() && !sealed) {
            uint32_t total = w;
            for (int d : deps) total += weight[d];
            const Small dv(deps);
            if (total <= maxIslandNodes && (deps.size() == 1 || !reaches(dv, deps, deps))) {
                // a foreign (other-sequence) producer must not be downstream of the merged set
                if (foreign.empty() || !reaches(dv, foreign, {})) {
                    target = dv[0];
                    for (size_t j = 1; j < dv.size(); ++j) {
                        int a = rep(target), b = rep(dv[j]);
                        if (a == b) continue;
                        uf.p[b] = a;
                        ilevel[a] = std::max(ilevel[a], ilevel[b]);
                        weight[a] += weight[b];
                        for (int sb : succ[b]) putSucc(a, sb);
                        target = a;
                    }
                    target = rep(target);
                }
            }
            if (target < 0) {
                // join the single producer island that none of the others is downstream of
                // (and only if that does not push the island to a later launch level: a mixer fed
                // by many peer islands gets its own island instead of stalling one of its peers)
                int best = -1;
                for (int d : dv) {
                    if (weight[d] + w > maxIslandNodes) continue;
                    bool raises = false;
                    for (int o : deps) if (o != d && ilevel[o] >= ilevel[d]) { raises = true; break; }
                    if (!raises) for (int o : foreign) if (ilevel[o] >= ilevel[d]) { raises = true; break; }
                    if (raises) continue;
                    others.clear();
                    for (int o : deps) if (o != d) others.push_back(o);
                    for (int f : foreign) put(others, f);
                    if (!others.empty() && reaches({d}, others, {})) continue;
                    if (best < 0 || weight[d] > weight[best]) best = d;
                }
                target = best;
            }
        }
        if (target < 0) {
            target = newIsland();
            for (int d : deps) ilevel[target] = std::max(ilevel[target], ilevel[rep(d)] + 1);
            for (int f : foreign) ilevel[target] = std::max(ilevel[target], ilevel[rep(f)] + 1);
        } else {
            for (int f : foreign) ilevel[target] = std::max(ilevel[target], ilevel[rep(f)] + 1);
        }
        x.island = target;
        weight[target] += w;
        if (sealed) weight[target] += maxIslandNodes;   // nothing joins a convolve island
        for (int d : deps) if (rep(d) != target) putSucc(rep(d), target);
        for (int f : foreign) if (rep(f) != target) putSucc(rep(f), target);
    }

    // canonical island list
    std::vector<IslandBuild> ib;
    std::vector<int> islandOf(uf.p.size(), -1);   // uf representative -> dense index (in order of first appearance)
    for (size_t k = 0; k < ni.size(); ++k) {
        NI& x = ni[k];
        if (x.kind == K_CONST) continue;
        const int r = rep(x.island);
        if (islandOf[(size_t)r] < 0) { islandOf[(size_t)r] = (int)ib.size(); ib.emplace_back(); ib.back().seq = x.seq; }
        x.island = islandOf[(size_t)r];
        ib[x.island].nodes.push_back((int)k);
    }

    // exports, in-island consumers, island deps
    for (size_t k = 0; k < ni.size(); ++k) {
        NI& x = ni[k];
        if (x.kind == K_CONST) continue;
        if (x.n->op == OP_ROOT || x.kind == K_CONV || x.kind == K_HOST) x.exported = true;
        if (x.kind == K_CHAIN) x.needLds = true;
    }
    for (size_t k = 0; k < ni.size(); ++k) {
        NI& x = ni[k];
        if (x.kind == K_CONST) continue;
        for (auto& in : x.n->inlets) {
            auto it = srcOf(in);
            if (it == idx.end()) continue;
            NI& s = ni[it->second];
            if (s.kind == K_CONST) continue;
            if (s.island == x.island) s.needLds = true;
            else {
                s.exported = true;
                auto& d = ib[x.island].deps;
                if (std::find(d.begin(), d.end(), s.island) == d.end()) d.push_back(s.island);
            }
        }
    }

    // launch levels (Kahn); a leftover island would mean a cycle slipped through clustering
    {
        std::vector<int> indeg(ib.size(), 0);
        std::vector<std::vector<int>> out(ib.size());
        for (size_t i = 0; i < ib.size(); ++i) for (int d : ib[i].deps) { out[d].push_back((int)i); indeg[i]++; }
        std::vector<int> q;
        for (size_t i = 0; i < ib.size(); ++i) if (!indeg[i]) q.push_back((int)i);
        size_t seen = 0;
        while (seen < q.size()) {
            int i = q[seen++];
            for (int o : out[i]) { ib[o].level = std::max(ib[o].level, ib[i].level + 1); if (--indeg[o] == 0) q.push_back(o); }
        }
        if (seen != ib.size()) { std::fprintf(stderr, "[elemhip] plan: island graph is cyclic\n"); return nullptr; }
    }
    // ---- 2b. lane-packing of isomorphic islands -----------------------------------------------------------------------
    // A stateful island is one workgroup, and its float recurrences run one NODE PER LANE: a voice's envelope pole keeps a
    // whole wavefront busy with one lane. When a launch level has more such islands than the chip has CUs (512 voices, 1024
    // render jobs) the islands of one shape are merged K at a time: the recurrence tasks of the merged island carry K lanes at
    // the price of one (same-opcode chain members of a stage share a task anyway), only the sample-parallel work grows K-fold.
    // Islands of one launch level never depend on each other, so any such merge keeps the island graph acyclic. Only islands
    // of ONE root sequence are merged: an island renders while its root runs (GraphRenderSequence.h:214-219), and two roots
    // may stop at different blocks — independent render jobs with a root each (C4) stay one per workgroup.
    packedIslands = 0;
    std::vector<uint32_t> packCount(ib.size(), 1u);        // original islands inside each island
    {   // how many stateful islands does the fullest launch level hold?  packK = 0: as many per island as it takes to fit the CUs
        std::map<int, uint32_t> perLevel;
        for (size_t i = 0; i < ib.size(); ++i) {
            bool stateful = false;
            for (int k : ib[i].nodes) if (ni[k].kind != K_PAR || ni[k].n->op == OP_TAPIN || ni[k].n->op == OP_TAPOUT) stateful = true;
            if (stateful) perLevel[ib[i].level]++;
        }
        statefulIslandsMax = 0;
        for (auto& kv : perLevel) statefulIslandsMax = std::max(statefulIslandsMax, kv.second);
        if (packK == 0) packK = std::max(1u, std::min(packMax, (statefulIslandsMax + cuCount - 1u) / std::max(1u, cuCount)));
    }
    if (packK > 1) {
        struct Key { int level, seq; uint64_t shape; bool operator<(const Key& o) const { return level != o.level ? level < o.level : seq != o.seq ? seq < o.seq : shape < o.shape; } };
        std::map<Key, std::vector<int>> groups;
        // `pack_roots` (opt-in): islands of DIFFERENT root sequences may share a workgroup when every one of those roots is active
        // at plan time — an active root keeps running until the next commit (a change of root targets always arrives with one), so
        // "renders while ITS root runs" (GraphRenderSequence.h:214-219) is the same fact for all members. The other half of that
        // test, root channel < the caller's output count, is a per-call fact: the engine refuses a call that asks for fewer outputs
        // than the packed roots' channels need (Plan::packedRootChannels).
        std::vector<char> seqActive(seqRoots.size(), 0);
        for (size_t sq = 0; sq < seqRoots.size(); ++sq) {
            auto a = seqRoots[sq]->props.find("active");
            seqActive[sq] = (a != seqRoots[sq]->props.end() && a->second.isBool() && a->second.b && seqRoots[sq]->channel >= 0) ? 1 : 0;
        }
        for (size_t i = 0; i < ib.size(); ++i) {
            bool stateful = false, sealed = false;
            uint64_t h = 1469598103934665603ull;
            for (int k : ib[i].nodes) {
                const NI& x = ni[k];
                if (x.kind == K_CONV || x.kind == K_HOST) sealed = true;
                if (x.kind != K_PAR || x.n->op == OP_TAPIN || x.n->op == OP_TAPOUT) stateful = true;
                h ^= (uint64_t)x.n->op | ((uint64_t)x.n->inlets.size() << 16); h *= 1099511628211ull;
            }
            if (sealed || !stateful) continue;
            groups[Key{ib[i].level, (packRoots && seqActive[(size_t)ib[i].seq]) ? -1 : ib[i].seq, h}].push_back((int)i);
        }
        std::vector<int> mergedInto(ib.size(), -1);
        for (auto& kv : groups) {
            std::vector<int>& g = kv.second;
            for (size_t q = 0; q + 1 < g.size(); q += packK) {
                const int head = g[q];
                for (size_t j = q + 1; j < std::min(g.size(), q + (size_t)packK); ++j) {
                    const int from = g[j];
                    ib[head].nodes.insert(ib[head].nodes.end(), ib[from].nodes.begin(), ib[from].nodes.end());
                    for (int d : ib[from].deps) if (std::find(ib[head].deps.begin(), ib[head].deps.end(), d) == ib[head].deps.end()) ib[head].deps.push_back(d);
                    ib[from].nodes.clear(); ib[from].deps.clear();
                    mergedInto[from] = head;
                    packCount[head] += 1u;
                    ++packedIslands;
                    if (ib[from].seq != ib[head].seq)
                        p.packedRootChannels = std::max(p.packedRootChannels, 1 + std::max(seqRoots[(size_t)ib[from].seq]->channel, seqRoots[(size_t)ib[head].seq]->channel));
                }
                std::sort(ib[head].nodes.begin(), ib[head].nodes.end());          // NI indices are render-order positions
            }
        }
        if (packedIslands) {   // compact the island list, renumber
            std::vector<int> newIdx(ib.size(), -1);
            std::vector<IslandBuild> nb;
            std::vector<uint32_t> pc;
            for (size_t i = 0; i < ib.size(); ++i) if (mergedInto[i] < 0) { newIdx[i] = (int)nb.size(); nb.push_back(std::move(ib[i])); pc.push_back(packCount[i]); }
            packCount.swap(pc);
            for (size_t i = 0; i < newIdx.size(); ++i) if (mergedInto[i] >= 0) newIdx[i] = newIdx[mergedInto[i]];
            for (IslandBuild& B : nb) {
                for (int& d : B.deps) d = newIdx[d];
                std::sort(B.deps.begin(), B.deps.end());
                B.deps.erase(std::unique(B.deps.begin(), B.deps.end()), B.deps.end());
            }
            for (NI& x : ni) if (x.kind != K_CONST && x.island >= 0) x.island = newIdx[x.island];
            ib.swap(nb);
        }
    }
    int numLevels = 0;
    for (auto& i : ib) numLevels = std::max(numLevels, i.level + 1);

    // HBM arena indices: level-major so one level's exports are contiguous
    {
        std::vector<int> order(ib.size());
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ib[a].level < ib[b].level; });
        uint32_t next = kMaxHostIn;
        for (int i : order) for (int k : ib[i].nodes) if (ni[k].exported) ni[k].hbm = next++;
        for (NI& x : ni) if (x.elided && x.n->op == OP_ROOT) x.hbm = next++;   // written by the root's convolver
        p.numHbmBuffers = next;
    }

    phase("islands");
    // ---- taps inside launch sets (Feedback.h:90-126, GraphRenderSequence.h:297-308) --------------------
    // A tapIn of block b+1 reads what the tapOut of its name received in block b (the end-of-block promotion). A launch
    // set renders its blocks level by level, so that hand-over can only happen inside ONE island: the island keeps a single
    // block in flight (copies = 1: block b+1 starts when block b has left it), the tapOut sits at least one stage behind its
    // tapIns, and a tapIn reads the tapOut's private buffer for every block but the set's first (run_tapin); the batch
    // epilogue promotes once, after the set. Anything else — a name written by two tapOuts, or read in another island
    // than the one that writes it — keeps the plan on the block-at-a-time path (Engine::batchEligible).
    std::vector<int> tapWriter(ni.size(), -1);      // tapIn NI index -> NI index of the tapOut it is paired with
    std::vector<char> islandPairsTaps(ib.size(), 0);
    p.tapsInSets = true;
    p.tapPairs.clear();
    {
        std::map<const void*, std::vector<int>> outs, ins;
        for (size_t k = 0; k < ni.size(); ++k) {
            if (ni[k].island < 0 || !ni[k].n->res) continue;
            if (ni[k].n->op == OP_TAPOUT) outs[ni[k].n->res.get()].push_back((int)k);
            if (ni[k].n->op == OP_TAPIN) ins[ni[k].n->res.get()].push_back((int)k);
        }
        for (auto& kv : outs) {
            std::vector<int> uniq;                   // (a tapOut shared by two roots appears once per NI entry of its id)
            for (int k : kv.second) { bool seen = false; for (int u : uniq) seen = seen || ni[u].n == ni[k].n; if (!seen) uniq.push_back(k); }
            if (uniq.size() > 1) { p.tapsInSets = false; continue; }
            auto it = ins.find(kv.first);
            if (it == ins.end()) continue;
            for (int k : it->second) {
                if (ni[k].island != ni[uniq[0]].island) { p.tapsInSets = false; continue; }
                tapWriter[(size_t)k] = uniq[0];
                islandPairsTaps[(size_t)ni[k].island] = 1;
                ni[(size_t)uniq[0]].needLds = true;   // the hand-over goes through the tapOut's LDS slot (below)
            }
        }
        for (size_t k = 0; k < ni.size(); ++k)
            if (ni[k].island >= 0 && ni[k].n->op == OP_TAPIN) p.tapPairs.push_back({ni[k].n->id, p.tapsInSets && tapWriter[k] >= 0 ? ni[(size_t)tapWriter[k]].n->id : 0});
    }
    // ---- 3/4. per-island schedule, LDS allocation, task emission -------------------------------------
    p.islands.resize(ib.size());
    p.islandProg.assign(ib.size(), nullptr);
    p.islandRoot.assign(ib.size(), 0);
    std::vector<uint32_t> scheduled;         // islands whose program this build made (p.prog holds them, progBegin relative to it)
    std::vector<uint32_t> canonRecs, canonHbms;   // the island's records / arena buffers in the order its canonical walk meets them
    std::vector<uint32_t> relocated;         // plan_cache = 2: the twin's renamed program, to be compared with the fresh schedule
    std::vector<int> convLevel;              // launch level of p.convs[i]
    for (size_t ii = 0; ii < ib.size(); ++ii) {
        IslandBuild& B = ib[ii];
        Island& I = p.islands[ii];
        I.rootRec = seqRoots[B.seq]->rec;
        p.islandRoot[ii] = seqRoots[B.seq]->id;
        if (ni[B.nodes[0]].kind == K_CONV) {     // one node, no island program: a ConvDesc instead
            NI& x = ni[B.nodes[0]];
            ConvDesc d{};
            d.rec = x.n->rec; d.outHbm = x.hbm; d.rootRec = I.rootRec; d.slices = std::max<uint32_t>(1, x.n->convSlices);
            d.fuseRootRec = kNone;
            if (x.fusedRoot >= 0) { d.outHbm = ni[x.fusedRoot].hbm; d.fuseRootRec = ni[x.fusedRoot].n->rec; }
            if (x.n->inlets.empty()) d.inKind = 3;                                      // leaf: host input 0
            else {
                const Inlet& in = x.n->inlets[0];
                auto it = srcOf(in);
                if (it == idx.end()) d.inKind = 4;
                else if (ni[it->second].elided) { d.inKind = 5; d.inIdx = ni[it->second].n->rec; }   // host channel named by the `in` record
                else if (ni[it->second].kind == K_CONST) { d.inKind = 2; d.inIdx = ni[it->second].n->rec; }
                else { d.inKind = 1; d.inIdx = ni[it->second].hbm; }
            }
            I = Island{};
            I.rootRec = d.rootRec; I.split = 0;                                         // no island-kernel workgroup
            convLevel.push_back(B.level);
            p.convs.push_back(d);
            p.convNodeIds.push_back(x.n->id);
            continue;
        }

        if (ni[B.nodes[0]].kind == K_HOST) {     // call-out node (Runtime::registerNodeType): no device program, a HostDesc for the engine
            NI& x = ni[B.nodes[0]];
            Plan::HostDesc d{};
            d.nodeId = x.n->id; d.rootId = seqRoots[B.seq]->id; d.outHbm = x.hbm; d.level = (uint32_t)B.level;
            {
                auto a = seqRoots[B.seq]->props.find("active");
                d.active = a != seqRoots[B.seq]->props.end() && a->second.isBool() && a->second.b;
            }
            d.leaf = x.n->inlets.empty();
            for (auto& in : x.n->inlets) {
                auto it = srcOf(in);
                if (it == idx.end()) d.inputs.push_back({0, 0u, 0.0f});
                else if (ni[it->second].kind == K_CONST && !ni[it->second].elided) d.inputs.push_back({2, ni[it->second].n->rec, 0.0f});
                else if (ni[it->second].elided && ni[it->second].n->op == OP_IN) d.inputs.push_back({3, ni[it->second].n->rec, 0.0f});
                else d.inputs.push_back({1, ni[it->second].hbm, 0.0f});
            }
            I = Island{};
            I.rootRec = seqRoots[B.seq]->rec; I.split = 0;
            p.hosts.push_back(std::move(d));
            continue;
        }

        // ---- island program cache (commit -> first block): a re-plan after "one voice replaced" meets the other
        // islands unchanged. Everything below is a function of the island's nodes (ids, opcodes, records, edges), of where its
        // exports / imports sit in the arena, of the stream buffers handed out so far and of the planner options: keyed by a hash
        // of exactly that, an unchanged island takes its Island header, program blob and kernel text from the previous build.
        // `plan_cache` = 2 schedules anyway and compares (tests).
        const uint32_t streamStart = p.numStreamBuffers;
        const auto tIsl0 = std::chrono::steady_clock::now();
        uint64_t ikey = 0, skey = 0;
        std::shared_ptr<IslandProgram> cached, twin;     // exact hit (program on the device already) / structural twin (its program, renamed)
        if (e.planCache != 0) {
            uint64_t h = 1469598103934665603ull;
            auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; h ^= h >> 29; };
            mix(bs); mix(maxCopies); mix(splitCoefStage); mix(wantSpec); mix(e.fuseSvfCoef); mix(e.mergePhases); mix(e.soloWaves);
            mix(e.mixerSplit); mix(e.chainLdsOut); mix(packCount[ii]); mix((uint32_t)islandPairsTaps[ii]); mix(streamStart);
            mix(B.nodes.size());       // (the owning root's record is not an input of the schedule: Island::rootRec is re-made per plan)
            for (int k : B.nodes) {
                const NI& x = ni[k];
                mix((uint32_t)x.n->id); mix(x.n->op); mix(x.rec); mix(x.ch); mix((uint32_t)x.kind); mix(x.exported); mix(x.hbm); mix(x.elided);
                mix((uint32_t)(x.fusedRoot >= 0 ? ni[(size_t)x.fusedRoot].n->rec : kNone)); mix(x.needLds);
                mix((uint32_t)(tapWriter[(size_t)k] >= 0 ? ni[(size_t)tapWriter[(size_t)k]].n->id : 0));
                mix(x.n->inlets.size());
                for (auto& in : x.n->inlets) {
                    mix((uint32_t)in.source); mix(in.channel);
                    auto it = srcOf(in);
                    if (it == idx.end()) { mix(0xDEADu); continue; }
                    const NI& sn = ni[it->second];
                    mix((uint32_t)sn.kind); mix(sn.island == x.island); mix(sn.hbm); mix(sn.rec); mix(sn.elided); mix(sn.n->op);
                }
            }
            ikey = h;
            // the same walk with every record / arena buffer replaced by the ordinal of its first appearance: the island's STRUCTURE
            // (only for islands the exact key does not find: an unchanged island of a live graph pays for one walk, not two)
            auto it = e.islandCache.find(ikey);
            const bool exactKnown = e.planCache == 1 && it != e.islandCache.end();
            if (e.relocatePrograms && !exactKnown) {
                uint64_t g = 1469598103934665603ull;
                auto smix = [&](uint64_t v) { g ^= v; g *= 1099511628211ull; g ^= g >> 29; };
                canonRecs.clear(); canonHbms.clear();
                auto recOrd = [&](uint32_t r) -> uint32_t {
                    if (r == kNone) return kNone;
                    for (size_t q = 0; q < canonRecs.size(); ++q) if (canonRecs[q] == r) return (uint32_t)q;
                    canonRecs.push_back(r); return (uint32_t)canonRecs.size() - 1u;
                };
                auto hbmOrd = [&](uint32_t b) -> uint32_t {
                    if (b == kNone) return kNone;
                    if (b < kMaxHostIn) return 0x40000000u | b;             // host input slots are the same for everybody
                    for (size_t q = 0; q < canonHbms.size(); ++q) if (canonHbms[q] == b) return (uint32_t)q;
                    canonHbms.push_back(b); return (uint32_t)canonHbms.size() - 1u;
                };
                auto posIn = [&](int niIndex) -> uint32_t {          // position of an island member (B.nodes is sorted)
                    auto f = std::lower_bound(B.nodes.begin(), B.nodes.end(), niIndex);
                    return (f != B.nodes.end() && *f == niIndex) ? (uint32_t)(f - B.nodes.begin()) : 0xFFFFu;
                };
                smix(bs); smix(maxCopies); smix(splitCoefStage); smix(wantSpec); smix(e.fuseSvfCoef); smix(e.mergePhases); smix(e.soloWaves);
                smix(e.mixerSplit); smix(e.chainLdsOut); smix(packCount[ii]); smix((uint32_t)islandPairsTaps[ii]);
                smix(B.nodes.size());
                for (int k : B.nodes) {
                    const NI& x = ni[k];
                    smix(x.n->op); smix(recOrd(x.rec)); smix(x.ch); smix((uint32_t)x.kind); smix(x.exported); smix(hbmOrd(x.hbm)); smix(x.elided);
                    smix(x.fusedRoot >= 0 ? recOrd(ni[(size_t)x.fusedRoot].n->rec) : kNone); smix(x.needLds);
                    const int tw = tapWriter[(size_t)k];
                    smix(tw >= 0 ? (posIn(tw) != 0xFFFFu ? posIn(tw) : 0x10000u | recOrd(ni[(size_t)tw].rec)) : 0xFFFFFFu);
                    smix(x.n->inlets.size());
                    for (auto& in : x.n->inlets) {
                        smix(in.channel);
                        auto it2 = srcOf(in);
                        if (it2 == idx.end()) { smix(0xDEADu); continue; }
                        const NI& sn = ni[it2->second];
                        const bool inside = sn.island == x.island;
                        smix((uint32_t)sn.kind); smix(inside); smix(inside ? posIn(it2->second) : 0xFFFFu);
                        smix(hbmOrd(sn.hbm)); smix(recOrd(sn.rec)); smix(sn.elided); smix(sn.n->op);
                    }
                }
                skey = g;
            }
            if (it != e.islandCache.end() && it->second->heap == e.progHeap) {
                // the key is 64 bits of hash: a hit is only taken when the members it was built from are the members in front of us
                const std::vector<uint32_t>& m = it->second->members;
                bool same = m.size() == 4 * B.nodes.size();
                for (size_t q = 0; same && q < B.nodes.size(); ++q) {
                    const NI& x = ni[B.nodes[q]];
                    same = m[4 * q] == (uint32_t)x.n->id && m[4 * q + 1] == ((uint32_t)x.n->op | (x.ch << 16)) && m[4 * q + 2] == x.rec && m[4 * q + 3] == x.hbm;
                }
                if (same) cached = it->second; else e.st.planCacheMismatches++;
            }
            if (cached && e.planCache == 1) {
                I = cached->I;
                I.progBegin = cached->heapBegin;          // the program is on the device already
                I.rootRec = seqRoots[B.seq]->rec;
                p.islandProg[ii] = cached;
                p.progDwordsTotal += cached->blob.size();
                p.numStreamBuffers += cached->streamDelta;
                if (packCount[ii] > 1u) minPackedCopies = minPackedCopies ? std::min(minPackedCopies, I.copies) : I.copies;
                p.maxCopies = std::max(p.maxCopies, I.copies);
                p.maxLdsBytes = std::max(p.maxLdsBytes, I.ldsWords * 4u);
                if (cached->spec) { if (p.specText.size() < ib.size()) p.specText.resize(ib.size()); p.specText[ii] = cached->spec; }
                p.numTasks += I.numTasks; p.numMembers += cached->numMembers; p.numOperands += cached->numOperands;
                e.st.planIslandsReused++;
                continue;
            }
            // ---- same structure, other nodes (another voice of the patch; the voice that replaces one): the twin's program with the
            // i-th record / arena buffer it names replaced by this island's i-th, stream buffers moved to this island's first. Any
            // reference the canonical walk did not meet makes the attempt fail and the island is scheduled as usual.
            if (skey != 0 && !cached) {
                auto ts = e.islandShapeCache.find(skey);
                if (ts != e.islandShapeCache.end() && ts->second->canonRecs.size() == canonRecs.size() && ts->second->canonHbms.size() == canonHbms.size() &&
                    ts->second->members.size() == 4 * B.nodes.size()) {
                    // the structural key is 64 bits of hash too: before another island's program is renamed into this one, the two
                    // must at least agree member by member on opcode and output channel
                    bool same = true;
                    for (size_t q = 0; same && q < B.nodes.size(); ++q) {
                        const NI& x = ni[B.nodes[q]];
                        same = ts->second->members[4 * q + 1] == ((uint32_t)x.n->op | (x.ch << 16));
                    }
                    if (same) twin = ts->second; else e.st.planRelocationMismatches++;
                }
            }
            if (twin) {
                std::vector<uint32_t> blob(twin->blob);
                const Island& T = twin->I;
                bool ok = true;
                auto mapRec = [&](uint32_t r) -> uint32_t {
                    for (size_t q = 0; q < twin->canonRecs.size(); ++q) if (twin->canonRecs[q] == r) return canonRecs[q];
                    ok = false; return r;
                };
                auto mapHbm = [&](uint32_t b) -> uint32_t {          // an arena index as stored in outHbm fields / operand values (kNone handled by callers)
                    if (b & kOpStream) return kOpStream | ((b & ~kOpStream) - twin->streamStart + streamStart);
                    if (b < kMaxHostIn) return b;
                    for (size_t q = 0; q < twin->canonHbms.size(); ++q) if (twin->canonHbms[q] == b) return canonHbms[q];
                    ok = false; return b;
                };
                auto mapOpnd = [&](uint32_t o) -> uint32_t { return (o & kOpKindMask) == kOpHbm ? (kOpHbm | mapHbm(o & kOpValMask)) : o; };
                const uint32_t numMembers = twin->numMembers, numOperands = twin->numOperands;
                for (uint32_t d = 0; d < T.copies && ok; ++d) {
                    uint32_t* c0 = blob.data() + (size_t)d * T.copyDwords;
                    for (uint32_t t = 0; t < T.numTasks; ++t) {
                        Task* tk = reinterpret_cast<Task*>(c0 + t * 8u);
                        tk->o0 = mapOpnd(tk->o0); tk->o1 = mapOpnd(tk->o1);
                        if (tk->outHbm != kNone) tk->outHbm = mapHbm(tk->outHbm);
                    }
                    for (uint32_t m = 0; m < numMembers; ++m) {
                        Member* mb = reinterpret_cast<Member*>(c0 + T.memOff + m * 8u);
                        if (mb->outHbm != kNone) mb->outHbm = mapHbm(mb->outHbm);
                    }
                    for (uint32_t o = 0; o < numOperands; ++o) c0[T.opndOff + o] = mapOpnd(c0[T.opndOff + o]);
                }
                for (uint32_t q = 0; q < T.numCells && ok; ++q) blob[T.cellOff + 2u * q + 1u] = mapRec(blob[T.cellOff + 2u * q + 1u]);
                for (uint32_t q = 0; q < T.numRecs && ok; ++q) blob[T.recOff + q] = mapRec(blob[T.recOff + q]);
                // behind the record table: the specialised variant's arena table, then its operand table (to the end of the program)
                const uint32_t tail0 = T.recOff + T.numRecs, tailN = T.progDwords - tail0;
                if (ok && tailN < twin->specHbmTab) ok = false;
                for (uint32_t q = 0; q < twin->specHbmTab && ok; ++q) blob[tail0 + q] = mapHbm(blob[tail0 + q]);
                for (uint32_t q = twin->specHbmTab; q < tailN && ok; ++q) blob[tail0 + q] = mapOpnd(blob[tail0 + q]);
                if (!ok) { twin.reset(); e.st.planRelocationMismatches++; }
                else if (e.planCache == 1) {
                    I = T;
                    I.progBegin = (uint32_t)p.prog.size();      // (local to this build's staging; moved into the heap below)
                    I.rootRec = seqRoots[B.seq]->rec;
                    p.prog.insert(p.prog.end(), blob.begin(), blob.end());
                    auto ent = std::make_shared<IslandProgram>();
                    ent->I = I; ent->blob.swap(blob); ent->spec = twin->spec;
                    ent->numMembers = twin->numMembers; ent->numOperands = twin->numOperands; ent->streamDelta = twin->streamDelta;
                    ent->specHbmTab = twin->specHbmTab;
                    ent->canonRecs = canonRecs; ent->canonHbms = canonHbms; ent->streamStart = streamStart; ent->shapeKey = skey;
                    ent->members.reserve(4 * B.nodes.size());
                    for (int k : B.nodes) { const NI& x = ni[k]; ent->members.insert(ent->members.end(), {(uint32_t)x.n->id, (uint32_t)x.n->op | (x.ch << 16), x.rec, x.hbm}); }
                    p.islandProg[ii] = ent; scheduled.push_back((uint32_t)ii);
                    p.progDwordsTotal += ent->blob.size();
                    p.numStreamBuffers += ent->streamDelta;
                    if (packCount[ii] > 1u) minPackedCopies = minPackedCopies ? std::min(minPackedCopies, I.copies) : I.copies;
                    p.maxCopies = std::max(p.maxCopies, I.copies);
                    p.maxLdsBytes = std::max(p.maxLdsBytes, I.ldsWords * 4u);
                    if (ent->spec) { if (p.specText.size() < ib.size()) p.specText.resize(ib.size()); p.specText[ii] = ent->spec; }
                    p.numTasks += I.numTasks; p.numMembers += ent->numMembers; p.numOperands += ent->numOperands;
                    e.islandCache[ikey] = ent;
                    e.st.planIslandsRelocated++;
                    continue;
                } else relocated.swap(blob);                    // plan_cache = 2: schedule anyway, then compare with this
            }
        }
        // imports needed in LDS: external producers (or host inputs) feeding chain members
        struct Import { uint32_t hbm; int lastUse; uint32_t lds; };
        std::vector<Import> imports;
        auto importFor = [&](uint32_t hbm) -> int {
            for (size_t q = 0; q < imports.size(); ++q) if (imports[q].hbm == hbm) return (int)q;
            imports.push_back(Import{hbm, 0, kNone});
            return (int)imports.size() - 1;
        };
        // stages
        bool anyImport = false;
        for (int k : B.nodes) {
            NI& x = ni[k];
            if (x.kind != K_CHAIN) continue;
            if (x.n->inlets.empty()) { if (leafArity(x.n->op) > 0) anyImport = true; continue; }
            for (auto& in : x.n->inlets) {
                auto it = srcOf(in);
                if (it == idx.end()) continue;
                NI& s = ni[it->second];
                if (s.kind != K_CONST && s.island != x.island) anyImport = true;
            }
        }
        const int base = anyImport ? 1 : 0;
        // svf with its coefficient pre-pass inside the scan (scan_svf, island_ops.inc): no pre-pass task, no 6-slot scratch
        const bool fuseCoef = e.fuseSvfCoef == 1 || (e.fuseSvfCoef == 2 && packCount[ii] > 1u);
        auto coefFused = [&](uint16_t op) { return fuseCoef && op == OP_SVF; };
        int maxStage = 0;
        for (int k : B.nodes) {
            NI& x = ni[k];
            int lv = base, sub = 0;
            for (auto& in : x.n->inlets) {
                auto it = srcOf(in);
                if (it == idx.end()) continue;
                NI& s = ni[it->second];
                if (s.kind == K_CONST || s.island != x.island) continue;
                // Sample-parallel ops are lane-local (lane l reads and writes only samples l + 64j of
                // its slice), so a sample-parallel consumer of a sample-parallel producer can run in
                // the SAME stage on the same wave, right after it, with no barrier in between.
                const bool xsvf = (x.n->op == OP_SVF || x.n->op == OP_SVFSHELF) && !coefFused(x.n->op);   // its coefficient pre-pass is such an op too
                // (xsvf && splitCoefStage: the pre-pass gets a stage of its own, so its light producers run unsplit on one
                // wave instead of four times with four times the per-task overhead)
                // (a split oscillator's waveform task is sample-parallel as well: its consumers may follow it in its stage)
                const bool fuse = (x.kind == K_PAR || (xsvf && !splitCoefStage)) && (s.kind == K_PAR || blepSplit(s.n->op));
                const int need = fuse ? s.level : s.level + 1;
                if (need > lv) { lv = need; sub = 0; }
                if (fuse && s.level == lv) sub = std::max(sub, s.sub + 1);
            }
            x.sub = sub;
            if (x.n->op == OP_TAPOUT && islandPairsTaps[ii]) {   // behind every tapIn (a leaf: stage `base`) that reads its buffer
                bool paired = false;
                for (int k2 : B.nodes) paired = paired || (tapWriter[(size_t)k2] >= 0 && ni[(size_t)tapWriter[(size_t)k2]].n == x.n);
                if (paired && lv < base + 1) { lv = base + 1; x.sub = 0; }
            }
            // svf / shelf take two stages: sample-parallel coefficient pre-pass, then the scan
            if ((x.n->op == OP_SVF || x.n->op == OP_SVFSHELF) && !coefFused(x.n->op)) lv += 1;
            if (blepSplit(x.n->op)) lv += 1;   // recurrence at lv - 1, waveform (the node's output) at lv
            x.level = lv;
            x.lastUse = lv;
            maxStage = std::max(maxStage, lv);
        }
        if (maxStage > 250) { std::fprintf(stderr, "[elemhip] plan: island too deep\n"); return nullptr; }
        for (int k : B.nodes) {
            NI& x = ni[k];
            for (auto& in : x.n->inlets) {
                auto it = srcOf(in);
                if (it == idx.end()) continue;
                NI& s = ni[it->second];
                if (s.kind == K_CONST) continue;
                if (s.island == x.island) s.lastUse = std::max(s.lastUse, x.level);
                else if (x.kind == K_CHAIN) { Import& im = imports[importFor(s.hbm)]; im.lastUse = std::max(im.lastUse, x.level); }
            }
            if (x.kind == K_CHAIN && x.n->inlets.empty()) {
                const uint32_t ar = std::min<uint32_t>(leafArity(x.n->op), kMaxHostIn);
                for (uint32_t c = 0; c < ar; ++c) { Import& im = imports[importFor(c)]; im.lastUse = std::max(im.lastUse, x.level); }
            }
        }

        // LDS slots by liveness. Words 0..3 = zero cell; slots start at word kSlot0.
        // Two regions keep the allocation dense: buffers that live for one stage (and the multi-slot scratch areas, which
        // need consecutive slots) recycle the low region; buffers that stay live across several stages go to their own
        // region, so they never leave holes in front of a scratch request. (C2 voice: 13 slots -> 10, i.e. one more block in
        // flight in the same LDS.)
        std::vector<int> slotFreeAt[2];   // per region: stage from which the slot is free again
        const uint32_t kLongBit = 1u << 31;
        auto takeSlots = [&](int stage, uint32_t count, int lastUse) -> uint32_t {
            const int region = (count == 1 && lastUse - stage >= 2) ? 1 : 0;
            std::vector<int>& fr = slotFreeAt[region];
            const uint32_t tag = region ? kLongBit : 0u;
            // `count` consecutive slots free at `stage`
            const size_t S = fr.size();
            for (size_t s0 = 0; s0 + count <= S; ++s0) {
                bool ok = true;
                for (uint32_t c = 0; c < count; ++c) if (fr[s0 + c] > stage) { ok = false; break; }
                if (ok) { for (uint32_t c = 0; c < count; ++c) fr[s0 + c] = lastUse + 1; return tag | (uint32_t)s0; }
            }
            // extend (reuse a free tail if there is one)
            size_t s0 = S;
            while (s0 > 0 && fr[s0 - 1] <= stage && S - (s0 - 1) <= count) --s0;
            fr.resize(s0 + count, 0);
            for (uint32_t c = 0; c < count; ++c) fr[s0 + c] = lastUse + 1;
            return tag | (uint32_t)s0;
        };
        for (auto& im : imports) im.lds = takeSlots(0, 1, im.lastUse);
        // A paired tapOut's output slot belongs to it for the whole block, every block: with one block in flight the slot still
        // holds block b when the tapIns of block b + 1 run (stage `base`, before the tapOut's own stage), so the hand-over never
        // leaves LDS — the same completion counters that order every other buffer of the island order it (run_tapin).
        std::vector<int> pairedOut;
        if (islandPairsTaps[ii])
            for (int k : B.nodes) if (tapWriter[(size_t)k] >= 0 && std::find(pairedOut.begin(), pairedOut.end(), tapWriter[(size_t)k]) == pairedOut.end()) pairedOut.push_back(tapWriter[(size_t)k]);
        for (int k : pairedOut) ni[(size_t)k].lds = takeSlots(0, 1, maxStage);
        for (int stage = base; stage <= maxStage; ++stage) {
            for (int k : B.nodes) {   // coefficient scratch of the svf's that scan in the NEXT stage
                NI& x = ni[k];
                if (x.level != stage + 1 || (x.n->op != OP_SVF && x.n->op != OP_SVFSHELF) || coefFused(x.n->op)) continue;
                x.scratch = takeSlots(stage, scratchSlots(x.n->op), stage + 1);
            }
            for (int k : B.nodes) {   // out slot of a split oscillator: carries the phase from the recurrence stage on
                NI& x = ni[k];
                if (x.level == stage + 1 && blepSplit(x.n->op) && x.needLds) x.lds = takeSlots(stage, 1, x.lastUse);
            }
            for (int k : B.nodes) {
                NI& x = ni[k];
                if (x.level != stage) continue;
                if (x.needLds && !blepSplit(x.n->op) && std::find(pairedOut.begin(), pairedOut.end(), k) == pairedOut.end()) x.lds = takeSlots(stage, 1, x.lastUse);
                const uint32_t sc = scratchSlots(x.n->op);
                if (sc && x.n->op != OP_SVF && x.n->op != OP_SVFSHELF) x.scratch = takeSlots(stage, sc, stage);
            }
        }
        {   // slot index -> LDS word: short-lived region first, long-lived region behind it
            const uint32_t nShort = (uint32_t)slotFreeAt[0].size();
            auto resolve = [&](uint32_t v) { return v == kNone ? v : kSlot0 + ((v & kLongBit) ? nShort + (v & ~kLongBit) : v) * kSlotWords; };
            for (auto& im : imports) im.lds = resolve(im.lds);
            for (int k : B.nodes) { ni[k].lds = resolve(ni[k].lds); ni[k].scratch = resolve(ni[k].scratch); }
            // a paired tapIn names its writer's slot (Member::scratch; run_tapin reads it from the set's second block on)
            for (int k : B.nodes) if (tapWriter[(size_t)k] >= 0) ni[k].scratch = ni[(size_t)tapWriter[(size_t)k]].lds;
        }
        const uint32_t slotArea = (uint32_t)(slotFreeAt[0].size() + slotFreeAt[1].size()) * kSlotWords;      // block-buffer words of one copy
        bool statelessIsland = true;
        for (int k : B.nodes) if (ni[k].kind != K_PAR || ni[k].n->op == OP_TAPIN || ni[k].n->op == OP_TAPOUT) statelessIsland = false;
        // blocks kept in flight by a multi-block launch: as many buffer sets as fit in ~140 KB of LDS (one such workgroup per CU)
        uint32_t copies = 1;
        if (!statelessIsland && slotArea > 0) copies = std::max<uint32_t>(1, std::min<uint32_t>(maxCopies, (35u * 1024u) / slotArea));
        if (islandPairsTaps[ii]) copies = 1;          // a feedback loop through a tap: one block in flight
        const uint32_t slotWords = kSlot0 + copies * slotArea;                    // first word after every copy's buffers

        // island-local program tables
        std::vector<Task> tasks;
        std::vector<int> taskWave;               // executing wave of tasks[i]
        std::vector<Member> members;
        std::vector<uint32_t> operands;
        std::vector<int> memberNode;             // NI index of members[i] (-1: an import copy)
        std::vector<int> operandSrc;             // operands[i]: NI index of the in-island producer, -2 - import index for an imported buffer, -1 otherwise
        std::vector<ConstCell> cells;
        // broadcast cells for const-like producers
        std::unordered_map<uint32_t, uint32_t> cellOf;   // rec -> lds word
        auto cellFor = [&](Node* c) -> uint32_t {
            auto it = cellOf.find(c->rec);
            if (it != cellOf.end()) return it->second;
            const uint32_t word = slotWords + (uint32_t)cellOf.size();
            cellOf.emplace(c->rec, word);
            cells.push_back(ConstCell{word, c->rec});
            return word;
        };

        std::vector<uint32_t> recTable;                  // local record index -> global record
        std::unordered_map<uint32_t, uint32_t> localRec;
        auto localOf = [&](uint32_t rec) {
            auto it = localRec.find(rec);
            if (it != localRec.end()) return it->second;
            localRec.emplace(rec, (uint32_t)recTable.size());
            recTable.push_back(rec);
            return (uint32_t)recTable.size() - 1u;
        };
        auto makeMember = [&](NI& x) -> Member {
            Member m{};
            memberNode.push_back((int)(&x - ni.data()));
            m.rec = localOf(x.rec);
            m.opnd = (uint32_t)operands.size();
            m.outLds = x.needLds ? x.lds : kNone;
            m.outHbm = x.exported ? x.hbm : kNone;
            m.scratch = x.scratch;
            if (x.n->inlets.empty()) {
                m.nin = kNone;
                const uint32_t ar = std::min<uint32_t>(leafArity(x.n->op), kMaxHostIn);
                for (uint32_t c = 0; c < ar; ++c) {
                    if (x.kind == K_CHAIN) { operandSrc.push_back(-2 - importFor(c)); operands.push_back(kOpLds | imports[importFor(c)].lds); }
                    else { operandSrc.push_back(-1); operands.push_back(kOpHbm | c); }
                }
                return m;
            }
            m.nin = (uint32_t)x.n->inlets.size();
            for (auto& in : x.n->inlets) {
                auto it = srcOf(in);
                if (it == idx.end()) { operandSrc.push_back(-1); operands.push_back(kOpZero); continue; }
                NI& s = ni[it->second];
                if (s.kind == K_CONST) { operandSrc.push_back(-1); operands.push_back(kOpConst | cellFor(s.n)); }
                else if (s.island == x.island) { operandSrc.push_back(it->second); operands.push_back(kOpLds | s.lds); }
                else if (x.kind == K_CHAIN) { operandSrc.push_back(-2 - importFor(s.hbm)); operands.push_back(kOpLds | imports[importFor(s.hbm)].lds); }
                else { operandSrc.push_back(-1); operands.push_back(kOpHbm | s.hbm); }
            }
            return m;
        };

        // A pure sample-parallel island that streams many arena buffers (a mixer: no imports, its children are read
        // straight from HBM by the reduce) is bound by memory round trips, not by issue: its stages are cut into 64-frame
        // runs over ALL eight waves, so one workgroup keeps 8 x 64 child loads in flight.
        bool mixerLike = false;
        if (statelessIsland && imports.empty() && bs >= 128) {
            size_t direct = 0;
            for (int k : B.nodes)
                for (auto& in : ni[k].n->inlets) {
                    auto it = srcOf(in);
                    if (it != idx.end() && ni[it->second].kind != K_CONST && ni[it->second].island != ni[k].island) direct++;
                }
            mixerLike = direct >= 8;
        }
        // ---- tasks, stage by stage ----
        // A sample-parallel task covers 64*V frames with V in {1,2,4,8} (lane l owns V consecutive
        // frames). The block is cut into 64-frame units handed out as power-of-two runs.
        uint32_t* loadOut = nullptr;
        auto emitRanges = [&](uint16_t op, int stage, uint32_t first, uint32_t count, const std::vector<int>& waves) {
            const uint32_t units = (bs + 63) / 64;                      // 64-frame units (8 for a 512 block)
            std::vector<std::pair<uint32_t, uint32_t>> runs;            // (first unit, units)
            if (units == 8 && waves.size() == 3) runs = {{0, 4}, {4, 2}, {6, 2}};
            else {
                uint32_t f = 1;
                while (f * 2 <= waves.size() && f * 2 <= units) f *= 2; // power-of-two wave count
                uint32_t per = (units + f - 1) / f, p2 = 1;
                while (p2 < per) p2 *= 2;                               // units per wave, power of two
                for (uint32_t u = 0; u < units; u += p2) runs.push_back({u, std::min(p2, units - u)});
            }
            for (size_t w = 0; w < runs.size(); ++w) {
                uint32_t u0 = runs[w].first, un = runs[w].second;
                // a non power-of-two tail (block sizes that are not 64 * 2^k) is cut further
                while (un) {
                    uint32_t take = 1; while (take * 2 <= un && take * 2 <= 8) take *= 2;
                    tasks.push_back(Task{op, (uint8_t)stage, 0, (uint16_t)(u0 * 64), (uint16_t)((u0 + take) * 64), first, count, 0, 0, 0, 0, 0});
                    taskWave.push_back(waves[w % waves.size()]);
                    if (loadOut) loadOut[waves[w % waves.size()]] += taskCost(op, take, count);
                    u0 += take; un -= take;
                }
            }
        };
        if (!imports.empty()) {
            const uint32_t first = (uint32_t)members.size();
            for (auto& im : imports) {
                Member m{};
                m.rec = 0; m.opnd = (uint32_t)operands.size(); m.nin = 1; m.outLds = im.lds; m.outHbm = kNone; m.scratch = kNone;
                operandSrc.push_back(-1);
                operands.push_back(kOpHbm | im.hbm);
                memberNode.push_back(-1);
                members.push_back(m);
            }
            emitRanges(OP_COPY, 0, first, (uint32_t)imports.size(), {0, 1, 2, 3});
        }
        auto constMaskOf = [&](const NI& x) -> uint32_t {
            uint32_t mask = 0;
            for (size_t q = 0; q < x.n->inlets.size() && q < 8; ++q) {
                auto it = srcOf(x.n->inlets[q]);
                if (it == idx.end()) { mask |= 1u << q; continue; }   // zero operand
                if (ni[it->second].kind == K_CONST) mask |= 1u << q;
            }
            return mask;
        };
        // constant-frequency phasors and blepsaw / blepsquare phase recurrences of one stage share ONE task (device.h OP_PHASE)
        auto phaseMergeable = [&](const NI& x) {
            const uint16_t op = x.n->op;
            return e.mergePhases && (op == OP_PHASOR || blepSplit(op)) && !x.n->inlets.empty() && (constMaskOf(x) & 1u);
        };
        uint32_t waveLoad[kWaves] = {};   // estimated cycles per block
        // waves that would stay empty with one wave per (stage, task group): a small island hands them to its heavy
        // sample-parallel stages (a finer split), where a big one needs every wave for a slot of its own
        int spareWaves = (int)kWaves;
        {
            std::set<std::pair<int, uint32_t>> groups;
            std::map<int, uint32_t> parTotal;
            for (int k : B.nodes) {
                const NI& x = ni[k];
                const uint16_t op = x.n->op;
                if (x.kind == K_CHAIN) groups.insert({x.level - (blepSplit(op) ? 1 : 0), 0x10000u | (phaseMergeable(x) ? (uint32_t)OP_PHASE : (uint32_t)op)});
                else if (x.kind == K_SINGLE) groups.insert({x.level, 0x20000u | (uint32_t)k});
                else if (x.kind == K_PAR) parTotal[x.level] += taskCost(op, 8, 1);
                if (blepSplit(op)) parTotal[x.level] += taskCost(op == OP_BLEPSAW ? OP_SAW_SHAPE : OP_SQUARE_SHAPE, 8, 1);
                if (op == OP_SVF && !coefFused(op)) parTotal[x.level - 1] += taskCost(OP_SVF_COEF, 8, 1);
                if (op == OP_SVFSHELF) parTotal[x.level - 1] += taskCost(OP_SHELF_COEF, 8, 1);
            }
            spareWaves -= (int)groups.size();
            for (auto& kv : parTotal) spareWaves -= kv.second >= 36000u ? 4 : (kv.second >= 18000u ? 2 : 1);
        }
        loadOut = waveLoad;
        for (int stage = base; stage <= maxStage; ++stage) {
            std::map<uint32_t, std::vector<int>> chain;   // key: opcode | constMask << 16
            std::map<uint16_t, std::vector<int>> single;
            std::map<std::pair<int, uint16_t>, std::vector<int>> par;   // (fusion depth, opcode): emitted in dependency order
            std::vector<int> phasePhasors, phaseOscs;          // members of this stage's OP_PHASE task
            for (int k : B.nodes) {
                NI& x = ni[k];
                if (x.level == stage + 1 && x.n->op == OP_SVF && !coefFused(OP_SVF)) par[{1 << 20, OP_SVF_COEF}].push_back(k);
                if (x.level == stage + 1 && x.n->op == OP_SVFSHELF) par[{1 << 20, OP_SHELF_COEF}].push_back(k);
                if (x.level == stage + 1 && blepSplit(x.n->op)) {
                    if (phaseMergeable(x)) phaseOscs.push_back(k);
                    else chain[(uint32_t)x.n->op | (constMaskOf(x) << 16)].push_back(k);
                }
                if (x.level != stage) continue;
                if (blepSplit(x.n->op)) par[{0, x.n->op == OP_BLEPSAW ? OP_SAW_SHAPE : OP_SQUARE_SHAPE}].push_back(k);
                else if (x.n->op == OP_PHASOR && phaseMergeable(x)) phasePhasors.push_back(k);
                else if (x.kind == K_CHAIN) chain[(uint32_t)x.n->op | (constMaskOf(x) << 16)].push_back(k);
                else if (x.kind == K_SINGLE) single[x.n->op].push_back(k);
                else par[{x.sub, x.n->op}].push_back(k);
            }
            // Wave assignment. Inside one block the tasks of a stage want different waves; across the blocks of
            // a pipelined multi-block launch each wave's TOTAL work per block bounds the throughput, so the
            // heavy (serial) tasks go to the wave with the least work so far, idle-in-this-stage waves first.
            uint32_t busy[kWaves] = {};
            // In a pipelined island the long serial tasks get waves 0..3 to themselves and the sample-parallel work
            // runs on waves 4..7: a light task on the critical path of an older block never queues behind a
            // 15-20 k-cycle recurrence of a younger one.
            const int serialWaves = (int)kWaves;   // (dedicating waves 0..3 to the serial tasks was measured: no gain, worse balance)
            auto pickWave = [&]() {
                int b = 0;
                for (int w = 1; w < serialWaves; ++w)
                    if (busy[w] < busy[b] || (busy[w] == busy[b] && waveLoad[w] < waveLoad[b])) b = w;
                return b;
            };
            {   // OP_PHASE: up to 64 lanes per task, phasors in front (Task::s0 = how many)
                size_t pi = 0, oi = 0;
                while (pi < phasePhasors.size() || oi < phaseOscs.size()) {
                    const uint32_t first = (uint32_t)members.size();
                    uint32_t np = 0, cnt = 0;
                    for (; pi < phasePhasors.size() && cnt < 64; ++pi, ++np, ++cnt) members.push_back(makeMember(ni[phasePhasors[pi]]));
                    for (; oi < phaseOscs.size() && cnt < 64; ++oi, ++cnt) members.push_back(makeMember(ni[phaseOscs[oi]]));
                    const int w = pickWave();
                    busy[w] += 1; waveLoad[w] += taskCost(OP_PHASE, 8, 1);
                    tasks.push_back(Task{OP_PHASE, (uint8_t)stage, 1u, (uint16_t)np, (uint16_t)bs, first, cnt, 0, 0, 0, 0, 0});
                    taskWave.push_back(w);
                }
            }
            for (auto& kv : chain) {
                const uint16_t cop = (uint16_t)(kv.first & 0xFFFFu);
                // the double-state filters are wave scans, one node after the other: every node is a task of its own, so that the
                // nodes of a stage (a packed island has one per voice) spread over the waves instead of queueing on one
                const size_t lanes = (cop == OP_SVF || cop == OP_SVFSHELF || cop == OP_MM1P) ? 1 : 64;
                for (size_t off = 0; off < kv.second.size(); off += lanes) {
                    const uint32_t cnt = (uint32_t)std::min<size_t>(lanes, kv.second.size() - off);
                    const uint32_t first = (uint32_t)members.size();
                    for (uint32_t c = 0; c < cnt; ++c) members.push_back(makeMember(ni[kv.second[off + c]]));
                    const int w = pickWave();
                    busy[w] += 1; waveLoad[w] += taskCost(cop, 8, (cop == OP_SVF || cop == OP_SVFSHELF || cop == OP_MM1P) ? cnt : 1);
                    tasks.push_back(Task{cop, (uint8_t)stage, (uint8_t)(kv.first >> 16), 0, (uint16_t)bs, first, cnt, 0, 0, 0, 0, 0});
                    taskWave.push_back(w);
                }
            }
            for (auto& kv : single) {
                for (int k : kv.second) {
                    const uint32_t first = (uint32_t)members.size();
                    members.push_back(makeMember(ni[k]));
                    const int w = pickWave();
                    busy[w] += 1; waveLoad[w] += taskCost(kv.first, 8, 1);
                    tasks.push_back(Task{kv.first, (uint8_t)stage, 0, 0, (uint16_t)bs, first, 1, 0, 0, 0, 0, 0});
                    taskWave.push_back(w);
                }
            }
            std::vector<int> freeWaves;
            for (int w = 0; w < (int)kWaves; ++w) if (busy[w] == 0) freeWaves.push_back(w);
            if (freeWaves.empty()) freeWaves.push_back(pickWave());
            std::sort(freeWaves.begin(), freeWaves.end(), [&](int a, int b) { return waveLoad[a] < waveLoad[b]; });
            // Light sample-parallel ops cost mostly per-task overhead. In a pipelined island (blocks overlap, so
            // there is always other work for the other waves) a stage's ops therefore run unsplit on ONE wave
            // unless their cost says otherwise (below).
            std::vector<int> parWaves = freeWaves;
            if (parWaves.size() > 4 && !mixerLike) parWaves.resize(4);   // a finer split only multiplies per-task overhead
            if (copies > 1) {
                // split a stage's sample-parallel work so that one part is about as long as a serial recurrence
                // (~12 k cycles): light stages run unsplit, a filter-coefficient pre-pass on two waves. A finer split
                // shortens the stage but makes its completion wait for the slowest of more, unrelated wave queues.
                uint32_t total = 0;
                for (auto& kv : par) total += taskCost(kv.first.second, 8, (uint32_t)kv.second.size());
                size_t f = total >= 36000u ? 4 : (total >= 18000u ? 2 : 1);
                if (f == 2 && spareWaves >= 2) { f = 4; spareWaves -= 2; }
                else if (f == 1 && total >= 9000u && spareWaves >= 1) { f = 2; spareWaves -= 1; }
                f = std::min(f, parWaves.size());
                parWaves.resize(f);
            }
            for (auto& kv : par) {
                const uint32_t first = (uint32_t)members.size();
                for (int k : kv.second) members.push_back(makeMember(ni[k]));
                emitRanges(kv.first.second, stage, first, (uint32_t)kv.second.size(), parWaves);
            }
        }
        if (copies > 1) {
            // Pipelined island: re-assign whole (stage, wave) slots to waves, longest first (LPT), now that every slot's cost is
            // known — the stage-by-stage choice above cannot see that, e.g., the long final stage is still to come and
            // parks it on the wave that already carries the coefficient stage. Slots of one stage keep distinct waves.
            struct Slot { int stage, wave; uint32_t cost; };
            std::vector<Slot> slots;
            for (size_t q = 0; q < tasks.size(); ++q) {
                const uint32_t cst = taskCost(tasks[q].opcode, ((uint32_t)tasks[q].s1 - tasks[q].s0 + 63u) / 64u, tasks[q].count);
                bool found = false;
                for (Slot& sl : slots) if (sl.stage == tasks[q].stage && sl.wave == taskWave[q]) { sl.cost += cst; found = true; break; }
                if (!found) slots.push_back(Slot{tasks[q].stage, taskWave[q], kSlotGap + cst});
            }
            std::vector<size_t> order(slots.size());
            std::iota(order.begin(), order.end(), (size_t)0);
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return slots[a].cost > slots[b].cost; });
            uint32_t load[kWaves] = {};
            std::map<int, uint32_t> usedInStage;                       // stage -> mask of waves taken
            std::map<std::pair<int, int>, int> remap;                  // (stage, old wave) -> new wave
            for (size_t o : order) {
                const Slot& sl = slots[o];
                uint32_t& used = usedInStage[sl.stage];
                // Waves w and w + 4 of the workgroup share a SIMD, and two busy waves on one SIMD slow each other once they mix
                // memory instructions into their VALU stream (a lone-wave recurrence next to another recurrence: ~20 % slower
                // than next to a light slot). Ties on the wave's own load go to the wave whose SIMD mate carries the least:
                // the four heaviest slots land on four different SIMDs, the fifth joins the lightest of them.
                int best = -1;
                auto better = [&](int w, int b) { return load[w] < load[b] || (load[w] == load[b] && load[w ^ 4] < load[b ^ 4]); };
                for (int w = 0; w < (int)kWaves; ++w) if (!((used >> w) & 1u) && (best < 0 || better(w, best))) best = w;
                if (best < 0) for (int w = 0; w < (int)kWaves; ++w) if (best < 0 || better(w, best)) best = w;
                used |= 1u << best; load[best] += sl.cost;
                remap[{sl.stage, sl.wave}] = best;
            }
            for (size_t q = 0; q < tasks.size(); ++q) taskWave[q] = remap.at({(int)tasks[q].stage, taskWave[q]});
            // Program waves w and w + 4 share a SIMD, and a wave loses issue slots to its mate for about half the mate's busy
            // time (specialised C2 voice, tools/spec_trace.py: a recurrence wave alone on its SIMD 7.3-8.8 k cycles per block,
            // 9.4-10.8 k next to a sample-parallel wave of 3.7-5.2 k). Recurrence waves bound the block rate, so each gets a
            // SIMD of its own as far as SIMDs go, the heaviest of them next to the lightest sample-parallel wave.
            {
                bool serial[kWaves] = {};
                for (size_t q = 0; q < tasks.size(); ++q) if (kindOf(tasks[q].opcode) == K_CHAIN) serial[taskWave[q]] = true;
                std::vector<int> ser, par;
                for (int w = 0; w < (int)kWaves; ++w) (serial[w] ? ser : par).push_back(w);
                std::stable_sort(ser.begin(), ser.end(), [&](int a, int b) { return load[a] > load[b]; });
                std::stable_sort(par.begin(), par.end(), [&](int a, int b) { return load[a] < load[b]; });
                if (ser.size() <= 4 && !ser.empty()) {
                    int renum[kWaves];
                    std::vector<int> rest;                       // waves still to place, mates first
                    size_t pi = 0;
                    for (size_t k = 0; k < ser.size(); ++k) { renum[ser[k]] = (int)k; if (pi < par.size()) renum[par[pi++]] = (int)k + 4; }
                    int freeIdx[kWaves]; int nf = 0;
                    bool taken[kWaves] = {};
                    for (size_t k = 0; k < ser.size(); ++k) { taken[k] = true; if (k < par.size()) taken[k + 4] = true; }
                    for (int w = 0; w < (int)kWaves; ++w) if (!taken[w]) freeIdx[nf++] = w;
                    for (int f = 0; pi < par.size() && f < nf; ++f) renum[par[pi++]] = freeIdx[f];
                    // `solo_waves` = n: the n heaviest recurrence waves keep their SIMD to themselves — the mate's tasks move
                    // to the lightest sample-parallel wave that is not such a mate (it then owns two slots per block)
                    const size_t solo = std::min<size_t>({(size_t)e.soloWaves, ser.size(), par.size() > 1 ? par.size() - 1 : 0});
                    if (solo > 0) {
                        uint32_t ld[kWaves] = {};
                        for (int w = 0; w < (int)kWaves; ++w) ld[renum[w]] = load[w];
                        for (size_t k = 0; k < solo; ++k) {
                            const int mate = renum[par[k]];
                            int best = -1;
                            for (size_t j = solo; j < par.size(); ++j) { const int w = renum[par[j]]; if (best < 0 || ld[w] < ld[best]) best = w; }
                            if (best < 0) break;
                            for (int w = 0; w < (int)kWaves; ++w) if (renum[w] == mate) renum[w] = best;
                            ld[best] += ld[mate]; ld[mate] = 0;
                        }
                    }
                    for (size_t q = 0; q < tasks.size(); ++q) taskWave[q] = renum[taskWave[q]];
                }
            }
        }
        {   // per-wave task lists: sort by (wave, stage), keep emission order inside a (wave, stage)
            std::vector<size_t> order(tasks.size());
            std::iota(order.begin(), order.end(), (size_t)0);
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
                return taskWave[a] != taskWave[b] ? taskWave[a] < taskWave[b] : tasks[a].stage < tasks[b].stage; });
            std::vector<Task> sorted; sorted.reserve(tasks.size());
            for (uint32_t w = 0; w <= kWaves; ++w) I.waveTask[w] = 0;
            for (size_t q : order) { sorted.push_back(tasks[q]); I.waveTask[taskWave[q] + 1]++; }
            for (uint32_t w = 0; w < kWaves; ++w) I.waveTask[w + 1] += I.waveTask[w];
            tasks.swap(sorted);
        }
        for (Task& t : tasks) {   // inline member 0 into the task header
            const Member& m0 = members[t.first];
            const uint32_t nops = m0.nin == kNone ? std::min<uint32_t>(leafArityOfOp(t.opcode), kMaxHostIn) : m0.nin;
            t.o0 = nops > 0 ? operands[m0.opnd] : (uint32_t)kOpZero;
            t.o1 = nops > 1 ? operands[m0.opnd + 1] : (uint32_t)kOpZero;
            t.outLds16 = m0.outLds == kNone ? (uint16_t)0xFFFF : (uint16_t)m0.outLds;
            t.nin16 = m0.nin == kNone ? (uint16_t)0xFFFF : (m0.nin >= 0xFFFE ? (uint16_t)0xFFFE : (uint16_t)m0.nin);
            t.outHbm = m0.outHbm;
            // fast sample-parallel task (kernels.hip run_fast): single member, math opcode, all
            // operands in LDS, arity satisfied (no zero-fill case), not a leaf
            const bool unary = t.opcode >= OP_SIN && t.opcode <= OP_ABS;
            const bool binary = t.opcode >= OP_LE && t.opcode <= OP_OR;
            const bool reduce2 = t.opcode >= OP_ADD && t.opcode <= OP_MAX;
            auto inLds = [](uint32_t o) { return (o & kOpKindMask) != kOpHbm; };
            if (t.count == 1 && m0.nin != kNone && (m0.outLds == kNone || m0.outLds < 0xFFFFu) &&
                ((unary && m0.nin >= 1 && inLds(t.o0)) || (binary && m0.nin >= 2 && inLds(t.o0) && inLds(t.o1)) ||
                 (reduce2 && m0.nin == 2 && inLds(t.o0) && inLds(t.o1))) &&
                (m0.outLds != kNone || m0.outHbm != kNone))
                t.flags |= 0x80u;
        }
        if (copies > 1) {   // recurrence tasks that have a wave to themselves render all blocks of a launch in one go (kTaskOwnsWave)
            for (uint32_t w = 0; w < kWaves; ++w) {
                if (I.waveTask[w + 1] - I.waveTask[w] != 1u) continue;
                Task& t = tasks[I.waveTask[w]];
                const uint16_t op = t.opcode;
                const bool plain = op == OP_PHASOR || op == OP_SPHASOR || op == OP_POLE || op == OP_ENV || op == OP_BIQUAD || op == OP_COUNTER ||
                                   op == OP_ACCUM || op == OP_LATCH || op == OP_MAXHOLD || op == OP_PHASE;
                const bool osc = blepSplit(op) && (t.flags & 1u);            // constant frequency: no per-block pre-pass
                if (!plain && !osc) continue;
                bool ok = (t.flags & 0xC0u) == 0u;
                for (uint32_t k = 0; k < t.count && ok; ++k) {
                    const Member& m = members[t.first + k];
                    if (m.outHbm != kNone || m.outLds == kNone || m.nin == kNone || m.nin < leafArityOfOp(op)) ok = false;
                }
                if (ok) t.flags |= (uint8_t)kTaskOwnsWave;
            }
        }
        // (A mixer used to run as 8 workgroups, each rendering a 64-frame slice with ONE active wave; the slices are now the
        // eight waves of one workgroup — same latency for a single block, and a multi-block launch gives every block its own
        // workgroup. Level 1 of C2: 48 -> 22 us per launch set. `mixer_split` = 2 keeps two workgroups of four active waves: the
        // single-block launch stays at 13 us where one workgroup of eight needs 21.)
        I.split = 1u;
        if (mixerLike) {   // slices must be whole 64-frame units: the largest divisor of the block's unit count within the option
            const uint32_t units = bs / 64u;
            uint32_t parts = bs % 64u == 0u ? std::max(1u, std::min(e.mixerSplit, units)) : 1u;
            while (parts > 1u && units % parts != 0u) --parts;
            I.split = parts;
        }
        // stage tables: tasks per stage, previous non-empty stage, per-wave first task of each stage
        const uint32_t S = (uint32_t)maxStage + 1;
        std::vector<uint32_t> stageTab(2 * S + kWaves * (S + 1) + copies + 1, 0u);
        uint32_t schedRel = 0;
        for (const Task& t : tasks) stageTab[t.stage]++;
        {
            uint32_t prev = kNone;
            for (uint32_t st = 0; st < S; ++st) { stageTab[S + st] = prev; if (stageTab[st]) prev = st; }
            for (uint32_t w = 0; w < kWaves; ++w) {
                uint32_t* begin = stageTab.data() + 2 * S + w * (S + 1);
                uint32_t ti = I.waveTask[w];
                for (uint32_t st = 0; st <= S; ++st) {
                    while (ti < I.waveTask[w + 1] && tasks[ti].stage < st) ++ti;
                    begin[st] = ti;
                }
            }
        }
        {   // phases of the block pipeline: `copies` runs of consecutive stages minimising the costliest run,
            // a stage costing what its busiest wave spends in it (the stages of one phase run back to back)
            std::vector<uint32_t> cost(S, 0u);
            {
                std::vector<std::array<uint32_t, kWaves>> perWave(S);
                for (auto& a : perWave) a.fill(0u);
                for (size_t q = 0; q < tasks.size(); ++q) {
                    const Task& t = tasks[q];
                    uint32_t w = 0; while (w + 1 < kWaves && q >= I.waveTask[w + 1]) ++w;
                    const uint16_t op = t.opcode;
                    const uint32_t units = ((uint32_t)t.s1 - t.s0 + 63u) / 64u;
                    const uint32_t cst = taskCost(op, units, t.count);
                    perWave[t.stage][w] += cst;
                }
                for (uint32_t st = 0; st < S; ++st) cost[st] = 1500u + *std::max_element(perWave[st].begin(), perWave[st].end());
            }
            const uint32_t P = std::min<uint32_t>(copies, S);
            // dp[p][j]: best max-run cost covering stages [0, j) with p runs
            const uint64_t INF = ~0ull;
            std::vector<std::vector<uint64_t>> dp(P + 1, std::vector<uint64_t>(S + 1, INF));
            std::vector<std::vector<uint32_t>> cut(P + 1, std::vector<uint32_t>(S + 1, 0u));
            std::vector<uint64_t> pre(S + 1, 0);
            for (uint32_t j = 0; j < S; ++j) pre[j + 1] = pre[j] + cost[j];
            dp[0][0] = 0;
            for (uint32_t pp = 1; pp <= P; ++pp)
                for (uint32_t j = pp; j <= S; ++j)
                    for (uint32_t i0 = pp - 1; i0 < j; ++i0) {
                        if (dp[pp - 1][i0] == INF) continue;
                        const uint64_t v = std::max(dp[pp - 1][i0], pre[j] - pre[i0]);
                        if (v < dp[pp][j]) { dp[pp][j] = v; cut[pp][j] = i0; }
                    }
            uint32_t* phase = stageTab.data() + 2 * S + kWaves * (S + 1);
            for (uint32_t d = 0; d <= copies; ++d) phase[d] = S;     // unused trailing phases are empty
            uint32_t j = S;
            for (uint32_t pp = P; pp >= 1; --pp) { phase[pp] = j; j = cut[pp][j]; }
            phase[0] = 0;
            for (uint32_t d = P + 1; d <= copies; ++d) phase[d] = S;
        }
        {   // per-wave walk of the block pipeline: only the (stage, phase) slots where the wave has tasks, in the
            // order the kernel visits them inside a macro-step (stage offset inside the phase ascending, oldest
            // block = highest phase first); 8 dwords each: stage, phase, first task, end task, prev stage, its task count
            const size_t base = 2 * S + kWaves * (S + 1);
            std::vector<uint32_t> phaseOf(S, 0u);
            for (uint32_t d = 0; d < copies; ++d) for (uint32_t st = stageTab[base + d]; st < stageTab[base + d + 1]; ++st) phaseOf[st] = d;
            std::vector<uint32_t> offs(kWaves + 1, 0u), entries;
            for (uint32_t w = 0; w < kWaves; ++w) {
                const uint32_t* begin = stageTab.data() + 2 * S + w * (S + 1);
                std::vector<uint32_t> mine;
                for (uint32_t st = 0; st < S; ++st) if (begin[st + 1] > begin[st]) mine.push_back(st);
                std::stable_sort(mine.begin(), mine.end(), [&](uint32_t a, uint32_t b) {
                    const uint32_t sla = a - stageTab[base + phaseOf[a]], slb = b - stageTab[base + phaseOf[b]];
                    return sla != slb ? sla < slb : phaseOf[a] > phaseOf[b]; });
                // the kernel polls a wave's slots in table order and runs the first ready one: later stages first, so the
                // oldest blocks drain (and release their buffer sets) before younger ones are started
                std::sort(mine.begin(), mine.end(), [](uint32_t a, uint32_t b) { return a > b; });
                for (uint32_t st : mine) {
                    const uint32_t prev = stageTab[S + st];
                    const uint32_t e[8] = {st, phaseOf[st], begin[st], begin[st + 1], prev, prev == kNone ? 0u : stageTab[prev], 0u, 0u};
                    entries.insert(entries.end(), e, e + 8);
                }
                offs[w + 1] = (uint32_t)entries.size() / 8u;
            }
            while ((stageTab.size() + offs.size()) % 4) stageTab.push_back(0u);   // entries are read with 16-byte LDS loads
            schedRel = (uint32_t)stageTab.size();
            stageTab.insert(stageTab.end(), offs.begin(), offs.end());
            stageTab.insert(stageTab.end(), entries.begin(), entries.end());
        }
        // ---- specialised-kernel variant of the program (codegen.cpp): recurrence streams through the HBM arena --------------
        // On a lone wavefront every instruction costs ~4 cycles and an LDS access 12-50, so a specialised kernel moves the
        // blocks a float recurrence reads and writes through L2 instead: inputs arrive by scalar loads (16 frames per
        // instruction), outputs leave as lane-0 global stores. The interpreter program above is left as it is (same LDS
        // layout); the variant only re-routes operands / outputs and names the arena buffers it needs extra.
        SpecProgram sp;
        // (stateless islands too — mixers, root gains: their specialised kernel has no block pipeline, blocks go over gridDim.y)
        const bool specIsland = wantSpec && bs % 64u == 0u;   // (vector loads of a 64-frame unit must stay inside their arena buffer)
        if (specIsland) {
            sp.members = members; sp.operands = operands;
            sp.gdirect.assign(tasks.size(), 0);
            std::unordered_map<int, uint32_t> streamOf;        // NI index -> arena buffer carrying the node's output
            auto stream = [&](int k) -> uint32_t {
                if (ni[k].exported) return ni[k].hbm;
                auto it = streamOf.find(k);
                if (it != streamOf.end()) return it->second;
                const uint32_t b = kOpStream | p.numStreamBuffers++;     // a buffer of the stream ring (device.h kOpStream)
                streamOf.emplace(k, b);
                return b;
            };
            auto streamFamily = [&](const Task& t) {
                switch (t.opcode) {
                    case OP_PHASOR: case OP_SPHASOR: case OP_POLE: case OP_ENV: case OP_BIQUAD: case OP_COUNTER: case OP_ACCUM:
                    case OP_LATCH: case OP_MAXHOLD: case OP_PHASE: return true;
                    case OP_BLEPSAW: case OP_BLEPSQUARE: return (t.flags & 1u) != 0u;   // constant frequency: no in-place pre-pass
                    default: return false;
                }
            };
            std::unordered_map<int, uint32_t> phaseStream;      // split oscillator: arena buffer of its phase (recurrence -> waveform task)
            std::unordered_set<int> streamed;                   // nodes whose output lives in the arena only (no LDS copy in the variant)
            for (size_t q = 0; q < tasks.size(); ++q) {
                const Task& t = tasks[q];
                if (!streamFamily(t)) continue;
                bool ok = true;
                for (uint32_t k = 0; k < t.count; ++k) {
                    const Member& m = members[t.first + k];
                    if (m.nin == kNone || m.nin < leafArityOfOp(t.opcode) || memberNode[t.first + k] < 0) ok = false;
                }
                if (!ok) continue;
                // (option "chain_lds_out": only the operands stream through the arena; the output block stays in its LDS slot)
                sp.gdirect[q] = e.chainLdsOut ? 0 : 1;
                for (uint32_t k = 0; k < t.count; ++k) {
                    const uint32_t mi = t.first + k;
                    const int x = memberNode[mi];
                    const bool osc = blepSplit(ni[x].n->op);            // (an OP_PHASE task carries both kinds)
                    if (!e.chainLdsOut) {
                        uint32_t b;
                        if (osc) { b = kOpStream | p.numStreamBuffers++; phaseStream.emplace(x, b); }
                        else { b = stream(x); streamed.insert(x); }
                        sp.members[mi].outHbm = b;
                    }
                    for (uint32_t j = 0; j < members[mi].nin; ++j) {
                        const uint32_t oi = members[mi].opnd + j;
                        if ((operands[oi] & kOpKindMask) != kOpLds) continue;
                        const int src = operandSrc[oi];
                        if (src >= 0) sp.operands[oi] = kOpHbm | stream(src);
                        else if (src <= -2) sp.operands[oi] = kOpHbm | imports[(size_t)(-2 - src)].hbm;
                    }
                }
            }
            // producers of streamed operands also write the arena; consumers of streamed nodes read it
            for (size_t mi = 0; mi < members.size(); ++mi) {
                const int x = memberNode[mi];
                if (x < 0) continue;
                auto it = streamOf.find(x);
                if (it != streamOf.end() && sp.members[mi].outHbm == kNone) sp.members[mi].outHbm = it->second;
            }
            for (size_t oi = 0; oi < operands.size(); ++oi) {
                const int src = operandSrc[oi];
                if (src >= 0 && streamed.count(src) && (operands[oi] & kOpKindMask) == kOpLds) sp.operands[oi] = kOpHbm | stream(src);
            }
            // the waveform task of a streamed oscillator reads the phase from the arena (member operand slot 5)
            sp.phaseOp.assign(members.size(), (uint32_t)kOpZero);
            for (size_t q = 0; q < tasks.size(); ++q) {
                if (tasks[q].opcode != OP_SAW_SHAPE && tasks[q].opcode != OP_SQUARE_SHAPE) continue;
                for (uint32_t k = 0; k < tasks[q].count; ++k) {
                    auto it = phaseStream.find(memberNode[tasks[q].first + k]);
                    if (it != phaseStream.end()) sp.phaseOp[tasks[q].first + k] = kOpHbm | it->second;
                }
            }
            // the oscillator's recurrence member and its waveform member are the same node: only the waveform member exports it
            for (size_t q = 0; q < tasks.size(); ++q)
                if ((blepSplit(tasks[q].opcode) || tasks[q].opcode == OP_PHASE) && !sp.gdirect[q])
                    for (uint32_t k = 0; k < tasks[q].count; ++k) {
                        const int x = memberNode[tasks[q].first + k];
                        if (x >= 0 && blepSplit(ni[x].n->op)) sp.members[tasks[q].first + k].outHbm = members[tasks[q].first + k].outHbm;
                    }
            // arena table: every absolute arena index the variant names, in order of first appearance
            auto ref = [&](uint32_t abs) { for (uint32_t v : sp.hbmTab) if (v == abs) return; sp.hbmTab.push_back(abs); };
            for (const Member& m : sp.members) if (m.outHbm != kNone) ref(m.outHbm);
            for (uint32_t o : sp.operands) if ((o & kOpKindMask) == kOpHbm) ref(o & kOpValMask);
            for (uint32_t o : sp.phaseOp) if ((o & kOpKindMask) == kOpHbm) ref(o & kOpValMask);
        }

        // pack the blob: copies x [tasks | members | operands] | cells | stage tables
        static_assert(sizeof(Task) == 32 && sizeof(Member) == 32 && sizeof(ConstCell) == 8, "program layout");
        I.progBegin = (uint32_t)p.prog.size();
        I.numTasks = (uint32_t)tasks.size();
        I.memOff = I.numTasks * 8u;
        I.opndOff = I.memOff + (uint32_t)members.size() * 8u;
        I.copyDwords = (I.opndOff + (uint32_t)operands.size() + 3u) & ~3u;
        I.copies = copies;
        if (packCount[ii] > 1u) minPackedCopies = minPackedCopies ? std::min(minPackedCopies, copies) : copies;
        p.maxCopies = std::max(p.maxCopies, copies);
        I.slotArea = slotArea;
        I.stateless = statelessIsland ? 1u : 0u;
        I.cellOff = I.copyDwords * copies;
        I.numCells = (uint32_t)cells.size();
        I.stageOff = (I.cellOff + I.numCells * 2u + 3u) & ~3u;
        I.schedOff = I.stageOff + schedRel;
        I.recOff = I.stageOff + (uint32_t)stageTab.size();
        I.numRecs = (uint32_t)recTable.size();
        // + the specialised variant's arena table and its operand table (same indexing as the interpreter's; wide fan-in ops
        //   fetch operand codes from the staged table at run time)
        I.progDwords = I.recOff + I.numRecs + (uint32_t)sp.hbmTab.size() + (uint32_t)sp.operands.size();
        p.prog.resize((size_t)I.progBegin + I.progDwords);
        for (uint32_t d = 0; d < copies; ++d) {
            const uint32_t off = d * slotArea;
            auto adjOp = [&](uint32_t o) { return (o & kOpKindMask) == kOpLds ? o + off : o; };
            std::vector<Task> tc(tasks);
            std::vector<Member> mc(members);
            std::vector<uint32_t> oc(operands);
            for (Task& t : tc) { t.o0 = adjOp(t.o0); t.o1 = adjOp(t.o1); if (t.outLds16 != 0xFFFFu) t.outLds16 = (uint16_t)(t.outLds16 + off); }
            for (Member& m : mc) { if (m.outLds != kNone) m.outLds += off; if (m.scratch != kNone) m.scratch += off; }
            for (uint32_t& o : oc) o = adjOp(o);
            uint32_t* blob = p.prog.data() + I.progBegin + (size_t)d * I.copyDwords;
            if (!tc.empty()) std::memcpy(blob, tc.data(), tc.size() * sizeof(Task));
            if (!mc.empty()) std::memcpy(blob + I.memOff, mc.data(), mc.size() * sizeof(Member));
            if (!oc.empty()) std::memcpy(blob + I.opndOff, oc.data(), oc.size() * 4);
        }
        {
            uint32_t* blob = p.prog.data() + I.progBegin;
            if (!cells.empty()) std::memcpy(blob + I.cellOff, cells.data(), cells.size() * sizeof(ConstCell));
            std::memcpy(blob + I.stageOff, stageTab.data(), stageTab.size() * 4);
            if (!recTable.empty()) std::memcpy(blob + I.recOff, recTable.data(), recTable.size() * 4);
            if (!sp.hbmTab.empty()) std::memcpy(blob + I.recOff + I.numRecs, sp.hbmTab.data(), sp.hbmTab.size() * 4);
            if (!sp.operands.empty()) std::memcpy(blob + I.recOff + I.numRecs + sp.hbmTab.size(), sp.operands.data(), sp.operands.size() * 4);
        }
        while (p.prog.size() % 4) p.prog.push_back(0);   // keep every blob 16-byte aligned
        I.numStages = S;
        I.ldsProg = (slotWords + (uint32_t)cellOf.size() + 3u) & ~3u;
        I.ldsCounters = (I.ldsProg + I.progDwords + 3u) & ~3u;
        I.ldsNext = (I.ldsCounters + S * copies + 3u) & ~3u;
        I.ldsRecs = (I.ldsNext + S * kWaves + 3u) & ~3u;
        I.ldsWords = (I.ldsRecs + I.numRecs * kRecDwords + 3u) & ~3u;
        p.maxLdsBytes = std::max(p.maxLdsBytes, I.ldsWords * 4u);
        if (specIsland && (I.split == 1u || statelessIsland)) {
            if (p.specText.size() < ib.size()) p.specText.resize(ib.size());
            // signature of everything the text is a function of (arena indices by their position in the island's arena table)
            uint64_t h = 1469598103934665603ull;
            auto mix = [&](uint32_t v) { h ^= v; h *= 1099511628211ull; h ^= h >> 29; };
            auto arenaPos = [&](uint32_t abs) { for (size_t k = 0; k < sp.hbmTab.size(); ++k) if (sp.hbmTab[k] == abs) return (uint32_t)k | (abs & kOpStream); return 0xFFFFu; };
            const uint32_t* iw = reinterpret_cast<const uint32_t*>(&I);
            for (size_t k = 0; k < sizeof(Island) / 4; ++k) if (k != offsetof(Island, progBegin) / 4 && k != offsetof(Island, rootRec) / 4) mix(iw[k]);
            for (const Task& t : tasks) { const uint32_t* w = reinterpret_cast<const uint32_t*>(&t); for (int k = 0; k < 8; ++k) mix(k == 7 ? 0u : w[k]); }   // (t.outHbm: absolute, not part of the text)
            for (const Member& m : sp.members) { mix(m.rec); mix(m.opnd); mix(m.nin); mix(m.outLds); mix(m.outHbm == kNone ? kNone : arenaPos(m.outHbm)); mix(m.scratch); }
            for (uint32_t o : sp.operands) mix((o & kOpKindMask) == kOpHbm ? (kOpHbm | arenaPos(o & kOpValMask)) : o);
            for (uint32_t o : sp.phaseOp) mix((o & kOpKindMask) == kOpHbm ? (kOpHbm | arenaPos(o & kOpValMask)) : o);
            for (uint8_t g : sp.gdirect) mix(g);
            for (uint32_t k = 0; k < 2 * S; ++k) mix(stageTab[k]);
            mix(bs); mix((uint32_t)sp.hbmTab.size()); mix((uint32_t)e.specWavesPerEu);
            auto it = e.specTextCache.find(h);
            if (it == e.specTextCache.end()) {
                auto txt = std::make_shared<SpecText>();
                txt->text = emitSpecSource(I, tasks, sp, stageTab, bs, (uint32_t)e.specWavesPerEu);
                it = e.specTextCache.emplace(h, std::move(txt)).first;
            }
            p.specText[ii] = it->second;
        }
        p.numTasks += I.numTasks; p.numMembers += (uint32_t)members.size(); p.numOperands += (uint32_t)operands.size();
        {
            auto ent = std::make_shared<IslandProgram>();
            ent->I = I;
            ent->blob.assign(p.prog.begin() + I.progBegin, p.prog.end());
            ent->members.reserve(4 * B.nodes.size());
            for (int k : B.nodes) { const NI& x = ni[k]; ent->members.insert(ent->members.end(), {(uint32_t)x.n->id, (uint32_t)x.n->op | (x.ch << 16), x.rec, x.hbm}); }
            p.islandProg[ii] = ent; scheduled.push_back((uint32_t)ii);
            p.progDwordsTotal += ent->blob.size();
            ent->specHbmTab = (uint32_t)sp.hbmTab.size();
            if (skey != 0) {
                ent->canonRecs = canonRecs; ent->canonHbms = canonHbms; ent->streamStart = streamStart; ent->shapeKey = skey;
                if (e.planCache != 0 && !e.islandShapeCache.count(skey)) e.islandShapeCache[skey] = ent;
            }
            if (!relocated.empty()) {       // plan_cache = 2: the twin's renamed program must be what was just scheduled
                if (relocated != ent->blob) { e.st.planRelocationMismatches++; std::fprintf(stderr, "[elemhip] plan cache: island %zu: the relocated program of its twin differs from its own schedule\n", ii); }
                else e.st.planIslandsRelocated++;
                relocated.clear();
            }
            if (ii < p.specText.size()) ent->spec = p.specText[ii];
            ent->numMembers = (uint32_t)members.size(); ent->numOperands = (uint32_t)operands.size();
            ent->streamDelta = p.numStreamBuffers - streamStart;
            if (cached) {          // plan_cache = 2: the cached program must be what was just scheduled
                Island a = cached->I, b2 = I;
                a.progBegin = b2.progBegin = 0u; a.rootRec = b2.rootRec = 0u;
                const bool same = std::memcmp(&a, &b2, sizeof(Island)) == 0 && cached->blob == ent->blob && cached->spec == ent->spec &&
                                  cached->streamDelta == ent->streamDelta && cached->numMembers == ent->numMembers;
                if (!same) { e.st.planCacheMismatches++; std::fprintf(stderr, "[elemhip] plan cache: island %zu differs from its cached program\n", ii); }
            }
            if (e.planCache != 0) { e.islandCache[ikey] = std::move(ent); e.st.planIslandsScheduled++; }
        }
        if (planTiming) std::fprintf(stderr, "[elemhip] plan   island %zu scheduled in %.3f ms (%zu nodes)\n", ii, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tIsl0).count(), B.nodes.size());
    }

    // the scheduled islands' programs go to the heap as one contiguous upload; everything else is there already
    {
        ProgHeap& H = *e.progHeap;
        const size_t need = p.prog.size();
        if (H.usedDwords + need > H.capDwords) { p.heapOverflowDwords = std::max<size_t>(need, 1); return plan; }
        const uint32_t base = (uint32_t)H.usedDwords;
        H.usedDwords += need;
        for (uint32_t ii : scheduled) {
            p.islands[ii].progBegin += base;
            p.islandProg[ii]->heap = e.progHeap; p.islandProg[ii]->heapBegin = p.islands[ii].progBegin;
        }
        if (H.dev && need && hipMemcpy(H.dev + base, p.prog.data(), need * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        p.progHeap = e.progHeap;
        p.prog.clear(); p.prog.shrink_to_fit();
    }
    phase("island programs");
    // ---- 5. launch levels, roots, taps ------------------------------------------------------------------
    p.islandLevel.resize(ib.size());
    for (size_t i = 0; i < ib.size(); ++i) p.islandLevel[i] = (uint32_t)ib[i].level;
    p.levelOffsets.assign((size_t)numLevels + 1, 0);
    p.levelLdsBytes.assign((size_t)numLevels, 0);
    for (size_t i = 0; i < ib.size(); ++i) p.levelOffsets[(size_t)ib[i].level + 1] += p.islands[i].split;
    for (int l = 0; l < numLevels; ++l) p.levelOffsets[(size_t)l + 1] += p.levelOffsets[l];
    p.levelIslands.resize(p.levelOffsets.back());
    {
        std::vector<uint32_t> cursor(p.levelOffsets.begin(), p.levelOffsets.end() - 1);
        for (size_t i = 0; i < ib.size(); ++i) {
            for (uint32_t k = 0; k < p.islands[i].split; ++k) p.levelIslands[cursor[ib[i].level]++] = (uint32_t)i | (k << 24);
            p.levelLdsBytes[ib[i].level] = std::max(p.levelLdsBytes[ib[i].level], p.islands[i].ldsWords * 4u);
        }
    }
    if (p.convs.size() > 0xFFFFu) { std::fprintf(stderr, "[elemhip] plan: too many convolve nodes\n"); return nullptr; }
    p.convLevelOffsets.assign((size_t)numLevels + 1, 0);
    for (int l = 0; l < numLevels; ++l) {    // per level: every node's main workgroup first, then the helpers
        for (size_t c = 0; c < p.convs.size(); ++c) if (convLevel[c] == l) p.convWork.push_back((uint32_t)c);
        for (size_t c = 0; c < p.convs.size(); ++c) if (convLevel[c] == l)
            for (uint32_t h = 0; h < conv::kBinGroups * p.convs[c].slices; ++h) p.convWork.push_back((uint32_t)c | ((h + 1u) << 16));
        p.convLevelOffsets[(size_t)l + 1] = (uint32_t)p.convWork.size();
    }
    for (size_t s = 0; s < seqRoots.size(); ++s) {
        Node* r = seqRoots[s];
        NI& x = ni[idx.at(K(r->id))];
        p.roots.push_back(RootEntry{r->rec, x.hbm});
        p.rootIds.push_back(r->id);
        for (int k : seqNodes[s]) if (ni[k].n->op == OP_TAPOUT) p.taps.push_back(TapEntry{ni[k].n->rec, r->rec});
        for (int k : seqNodes[s]) if (ni[k].n->op == OP_METER || ni[k].n->op == OP_SNAPSHOT || ni[k].n->op == OP_SCOPE || ni[k].n->op == OP_CAPTURE) p.eventNodes.push_back({ni[k].n->id, r->id});
    }
    phase("levels, roots");
    return plan;
}

static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

std::shared_ptr<Plan> Engine::buildPlan(std::unique_lock<std::mutex>& renderLock) {
    // Planning reads the node table (structure owned by `ctl`, which the caller holds) and writes only the new Plan:
    // it runs WITHOUT the render lock, so process* calls keep rendering the current plan while a commit is planned.
    struct Relock {
        std::unique_lock<std::mutex>& l;
        explicit Relock(std::unique_lock<std::mutex>& l_) : l(l_) { l.unlock(); }
        ~Relock() { if (!l.owns_lock()) l.lock(); }
    } relock(renderLock);
    if (debugBuildDelayMs > 0) std::this_thread::sleep_for(std::chrono::milliseconds(debugBuildDelayMs));
    std::shared_ptr<Plan> plan;
    const uint32_t ldsLimit = 160u * 1024u - 1024u;
    // the text cache is only ever trimmed BETWEEN builds (texts in use stay alive through the shared objects the plan holds)
    if (specTextCache.size() > 4096) specTextCache.clear();
    if (islandCache.size() > 2048) islandCache.clear();      // (entries of replaced islands are never looked up again: ~8 KB each)
    if (islandShapeCache.size() > 512) islandShapeCache.clear();
    // Lane-packing (option "pack_islands": 0 auto, 1 off, K): the first attempt packs as the option says; a packed island
    // that does not fit in LDS, or fits with a single buffer set (no blocks in flight: the stages of a block would run back
    // to back), sends the build back with one island fewer per pack.
    // room in the program heap for this build's new programs: a few islands on a live graph, all of them on a first build
    auto freshHeap = [&](size_t needDwords) -> bool {
        auto h = std::make_shared<ProgHeap>();
        h->capDwords = std::max<size_t>((size_t)4 << 20, 8 * needDwords);       // 16 MB, or eight builds' worth
        if (progHeapCap) h->capDwords = std::max(progHeapCap, needDwords);       // option "prog_heap_dwords" (tests: a heap that keeps running out)
        else if (dry) h->capDwords = (size_t)1 << 30;
        if (dry) {}
        else if (hipMalloc(reinterpret_cast<void**>(&h->dev), h->capDwords * 4) != hipSuccess) return false;
        islandCache.clear();                   // (its entries name offsets of the old heap; plans in flight keep that heap alive)
        progHeap = std::move(h);
        st.progHeaps++;
        return true;
    };
    if (!progHeap || progHeap->capDwords - progHeap->usedDwords < (progHeapCap ? 0 : std::max<size_t>((size_t)1 << 18, 2 * lastPlanProgDwords)))
        if (!freshHeap(lastPlanProgDwords)) return nullptr;
    uint32_t packK = (uint32_t)packIslands;
    for (;;) {
        uint32_t usedK = 1, minCopies = 0;
        for (uint32_t limit = 56; limit >= 4; limit /= 2) {
            PlanBuilder b(*this);
            // a dry handle (no device) only generates / compiles kernels when asked to wait for them (cache warming, tests)
            b.wantSpec = specialize != 0 && (!dry || specialize >= 2);
            b.packK = packK; b.packMax = (uint32_t)std::max(1, packMax); b.cuCount = (uint32_t)std::max(1, cuCount); b.packRoots = packRoots;
            plan = b.build(limit, (uint32_t)std::max(1, pipelineCopies));
            if (plan && plan->heapOverflowDwords) {        // a graph far bigger than the last one: a heap sized for it, same attempt again
                if (!freshHeap(plan->heapOverflowDwords)) return nullptr;
                PlanBuilder b2(*this);
                b2.wantSpec = b.wantSpec; b2.packK = packK; b2.packMax = b.packMax; b2.cuCount = b.cuCount; b2.packRoots = packRoots;
                plan = b2.build(limit, (uint32_t)std::max(1, pipelineCopies));
                if (plan && plan->heapOverflowDwords) plan.reset();
                if (!plan) return nullptr;
                b.packK = b2.packK; b.minPackedCopies = b2.minPackedCopies; b.packedIslands = b2.packedIslands;
            }
            if (!plan) return nullptr;
            usedK = b.packK; minCopies = b.minPackedCopies;
            plan->packK = b.packedIslands ? usedK : 1u;
            if (plan->maxLdsBytes <= ldsLimit) break;
            plan.reset();
            if (usedK > 1) break;          // (a packed island too big for LDS: pack fewer rather than cut the islands up)
        }
        if (usedK > 1 && (!plan || (minCopies != 0 && minCopies < 2 && pipelineCopies >= 2))) { plan.reset(); packK = usedK - 1; continue; }
        break;
    }
    if (!plan) { std::fprintf(stderr, "[elemhip] plan: could not fit islands into LDS\n"); return nullptr; }
    Plan& p = *plan;
    lastPlanProgDwords = p.progDwordsTotal;

    // pack + upload the tables
    const auto tTables = std::chrono::steady_clock::now();
    size_t off = 0;
    auto place = [&](size_t bytes) { size_t o = off; off = align16(off + bytes); return o; };
    const size_t oIslands = place(p.islands.size() * sizeof(Island));
    const size_t oLevel = place(p.levelIslands.size() * 4);
    const size_t oRoots = place(p.roots.size() * sizeof(RootEntry));
    const size_t oTaps = place(p.taps.size() * sizeof(TapEntry));
    const size_t oConvs = place(p.convs.size() * sizeof(ConvDesc));
    const size_t oConvWork = place(p.convWork.size() * 4);
    // ---- specialised kernels: group each level's islands by generated text, queue the shapes for compilation ----
    double jitWaitMs = -1.0;
    // Call-out nodes synchronise with the host inside a block: their plans render block at a time through the interpreter kernels.
    // (Plans with tap nodes keep their specialised kernels: a paired tapIn takes its block from the tapOut's LDS slot, an unpaired
    // one from the shared buffer the per-block promotion fills — neither needs anything the specialised kernels lack.)
    if (!p.hosts.empty()) p.specText.clear();
    {
        const size_t L = p.levelOffsets.size() - 1;
        p.restOffsets.assign(L + 1, 0);
        p.restRoots.assign(L, {});
        std::vector<uint8_t> covered(p.islands.size(), 0);
        for (size_t l = 0; l < L; ++l) {
            // one shape per kernel: islands are grouped by the cache key of their text (two cached text objects can carry the same
            // text — the text cache is keyed by a signature that also covers what the text leaves out); a split island appears
            // once per part in levelIslands and once here
            std::map<std::string, std::pair<SpecText*, std::vector<uint32_t>>> byText;
            std::vector<uint8_t> seen(p.islands.size(), 0);
            for (uint32_t q = p.levelOffsets[l]; q < p.levelOffsets[l + 1]; ++q) {
                const uint32_t isl = p.levelIslands[q] & 0xFFFFFFu;
                if (!(isl < p.specText.size() && p.specText[isl]) || covered[isl] || seen[isl]) continue;
                seen[isl] = 1;
                SpecText& tx = *p.specText[isl];
                const uint32_t ldsW = p.islands[isl].ldsWords;
                if (tx.key.empty() || tx.keyLdsWords != ldsW) { tx.key = Jit::get().keyFor(tx.text, ldsW, (uint32_t)blockSize); tx.keyLdsWords = ldsW; }
                auto& slot = byText[tx.key];
                slot.first = &tx; slot.second.push_back(isl);
            }
            // the level's shapes, the ones with the most islands first (ties: by key): a level renders at most `max_shape_launches`
            // specialised launches, the biggest shapes — everything behind them in this list is ONE contiguous run the interpreter
            // kernel takes in one launch (Engine::launchLevelBatch). A live graph of structurally different voices used to give every
            // voice whose kernel had been compiled a launch (and a side stream) of its own: 60 launches per level and set, 150 us per block.
            std::vector<std::pair<const std::string*, std::pair<SpecText*, std::vector<uint32_t>>*>> order;
            for (auto& kvk : byText) order.emplace_back(&kvk.first, &kvk.second);
            std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.second->second.size() > b.second->second.size(); });
            for (auto& ord : order) {
                auto& kv = *ord.second;
                // background mode: a shape only one island has (a voice that is fading out next to its replacement, a
                // one-off graph) is not worth a compile at commit time: its plan may be gone in 30 ms. It gets a DEFERRED entry —
                // known to the kernel cache, not queued — and renders through the interpreter kernel; once this plan has rendered
                // for a while (Engine::promoteDeferredShapes: `spec_lonely_blocks` blocks and `spec_lonely_ms` of wall clock) the
                // shape is queued behind everything else, so a static single-patch graph leaves the interpreter too (r04: never)
                const uint32_t ldsW = p.islands[kv.second[0]].ldsWords;
                SpecText& tx = *kv.first;
                // ... and at once when it keeps coming back: a live graph that replaces a voice per commit meets the same one-off shape
                // (the old voice fading out behind its own mixer and root) in every plan — the second plan that wants it has it compiled
                const bool lonely = specialize == 1 && kv.second.size() < 2;
                const bool defer = lonely && !Jit::get().knownKey(tx.key) && Jit::get().sighting(tx.key) < 2u;
                Plan::SpecShape sh;
                sh.optional = lonely;
                sh.deferred = defer;
                sh.entry = Jit::get().requestKey(tx.key, tx.text, ldsW, (uint32_t)blockSize, defer);
                if (!dry) sh.entry->wantOn(device);
                if (defer) p.deferredShapes++;
                sh.level = (uint32_t)l; sh.listBegin = (uint32_t)p.specLists.size();
                sh.stateless = p.islands[kv.second[0]].stateless != 0u;
                // (levelIslands entry format: island | split part << 24 — a split island is one workgroup per part)
                for (uint32_t isl : kv.second) {
                    for (uint32_t k = 0; k < std::max(1u, p.islands[isl].split); ++k) p.specLists.push_back(isl | (k << 24));
                    covered[isl] = 1;
                    if (std::find(sh.roots.begin(), sh.roots.end(), p.islandRoot[isl]) == sh.roots.end()) sh.roots.push_back(p.islandRoot[isl]);
                }
                sh.count = (uint32_t)p.specLists.size() - sh.listBegin;
                p.shapes.push_back(std::move(sh));
            }
            for (uint32_t q = p.levelOffsets[l]; q < p.levelOffsets[l + 1]; ++q) {
                const uint32_t isl = p.levelIslands[q] & 0xFFFFFFu;
                if (covered[isl]) continue;
                p.restIslands.push_back(p.levelIslands[q]);
                auto& rr = p.restRoots[l];
                if (std::find(rr.begin(), rr.end(), p.islandRoot[isl]) == rr.end()) rr.push_back(p.islandRoot[isl]);
            }
            p.restOffsets[l + 1] = (uint32_t)p.restIslands.size();
        }
        p.specText.clear(); p.specText.shrink_to_fit();
        if (specialize >= 2) {
            const auto t0 = std::chrono::steady_clock::now();
            for (auto& sh : p.shapes) (void)Jit::get().wait(sh.entry);
            jitWaitMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    }
    const size_t oSpecLists = place(p.specLists.size() * 4);
    const size_t oRest = place(p.restIslands.size() * 4);
    std::vector<uint8_t> host(std::max<size_t>(off, 16), 0);
    auto put = [&](size_t o, const void* src, size_t bytes) { if (bytes) std::memcpy(host.data() + o, src, bytes); };
    put(oIslands, p.islands.data(), p.islands.size() * sizeof(Island));
    put(oLevel, p.levelIslands.data(), p.levelIslands.size() * 4);
    put(oRoots, p.roots.data(), p.roots.size() * sizeof(RootEntry));
    put(oTaps, p.taps.data(), p.taps.size() * sizeof(TapEntry));
    put(oConvs, p.convs.data(), p.convs.size() * sizeof(ConvDesc));
    put(oConvWork, p.convWork.data(), p.convWork.size() * 4);
    put(oSpecLists, p.specLists.data(), p.specLists.size() * 4);
    put(oRest, p.restIslands.data(), p.restIslands.size() * 4);

    const auto tUpload = std::chrono::steady_clock::now();
    p.buildUs[4] = std::chrono::duration<double, std::micro>(tUpload - tTables).count();
    // The tables go to the device BEFORE the render lock is taken again (r06): nothing a render call touches is involved — the pool
    // has a lock of its own, the copy runs on the null stream, the new plan is not visible to anybody yet — so what a commit still
    // does under `mu` behind a plan build is a handful of stores (the upload was 10 of the ~12 us it held).
    if (!dry) {
        p.pool = tablePool;
        p.dev = tablePool->take(host.size());
        if (!p.dev.ptr) return nullptr;
        if (hipMemcpy(p.dev.ptr, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        const uint8_t* d = static_cast<const uint8_t*>(p.dev.ptr);
        p.view.islands = reinterpret_cast<const Island*>(d + oIslands);
        p.view.levelIslands = reinterpret_cast<const uint32_t*>(d + oLevel);
        p.view.prog = p.progHeap->dev;          // island programs live in the engine's program heap (Island::progBegin is relative to it)
        p.view.roots = reinterpret_cast<const RootEntry*>(d + oRoots);
        p.view.taps = reinterpret_cast<const TapEntry*>(d + oTaps);
        p.view.convs = reinterpret_cast<const ConvDesc*>(d + oConvs);
        p.view.convWork = reinterpret_cast<const uint32_t*>(d + oConvWork);
        p.dSpecLists = reinterpret_cast<const uint32_t*>(d + oSpecLists);
        p.dRestIslands = reinterpret_cast<const uint32_t*>(d + oRest);
        p.view.numConvs = (uint32_t)p.convs.size();
        p.view.numRoots = (uint32_t)p.roots.size();
        p.view.numTaps = (uint32_t)p.taps.size();
    }
    p.buildUs[5] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tUpload).count();
    renderLock.lock();   // ---- from here on: render-side state ----
    residentStop();
    st.specShapes = (uint32_t)p.shapes.size(); st.specIslands = (uint32_t)p.specLists.size();
    if (jitWaitMs >= 0.0) st.lastJitWaitMs = jitWaitMs;
    // root records learn whether fade.process() runs on them (Core.h:74-77)
    for (int32_t id : p.rootIds) {
        Node& r = nodes.at(id);
        const uint32_t has = r.inlets.empty() ? 0u : 1u;
        if (shadow[r.rec * kRecDwords + rec::ROOT_HASIN] != has) writeParam(r, rec::ROOT_HASIN, has);
    }
    return plan;
}

} // namespace elemhip

namespace elemhip {

// JSON description of the render plan (islands, levels, tasks) for host-logic tests and debugging.
std::string Engine::describePlan() {
    std::lock_guard<std::mutex> lock(mu);
    // the newest plan (pending if a commit has not been rendered yet); nothing is adopted here
    const std::shared_ptr<Plan> shown = pending ? pending : current;
    if (!shown) return "null";
    const Plan& p = *shown;
    std::string s = "{";
    auto kv = [&](const char* k, uint64_t v, bool comma = true) { s += "\"" + std::string(k) + "\":" + std::to_string(v) + (comma ? "," : ""); };
    kv("num_islands", p.islands.size()); kv("num_levels", p.levelOffsets.size() - 1); kv("num_tasks", p.numTasks);
    kv("num_members", p.numMembers); kv("num_operands", p.numOperands); kv("num_nodes", p.nodeIds.size());
    kv("plan_islands_relocated", st.planIslandsRelocated); kv("plan_relocation_mismatches", st.planRelocationMismatches);
    kv("plan_spec_texts", specTextCache.size()); kv("plan_spec_fade_blocks", st.specFadeBlocks); kv("plan_idle_launches_skipped", st.idleLaunchesSkipped); kv("plan_fused_epilogues", st.fusedEpilogues); kv("plan_prog_heaps", st.progHeaps); kv("plan_prog_heap_used_dwords", p.progHeap ? p.progHeap->usedDwords : 0);
    kv("plan_islands_reused", st.planIslandsReused); kv("plan_islands_scheduled", st.planIslandsScheduled); kv("plan_cache_mismatches", st.planCacheMismatches);
    kv("num_hbm_buffers", p.numHbmBuffers); kv("num_stream_buffers", p.numStreamBuffers); kv("pack_k", p.packK);
    kv("max_lds_bytes", p.maxLdsBytes); kv("num_roots", p.roots.size());
    s += "\"build_us\":{";
    {
        static const char* names[6] = {"render_order", "islands", "island_programs", "levels_roots", "shapes_tables", "upload"};
        for (int k = 0; k < 6; ++k) { char b[64]; std::snprintf(b, sizeof b, "%s\"%s\":%.1f", k ? "," : "", names[k], p.buildUs[k]); s += b; }
    }
    s += "},";
    {   // which kernels rendered (island x block units since the handle was made), this plan's shapes, the process-wide kernel cache
        const JitStats js = Jit::get().stats();
        const uint64_t tot = islandBlocksSpec + islandBlocksInterp;
        char b[1024];
        uint32_t ready = 0, waiting = 0, deferred = 0;
        for (const Plan::SpecShape& sh : p.shapes) { const int stt = sh.entry->state.load(); if (stt == 1) ++ready; else if (stt == 2) ++deferred; else if (stt == 0) ++waiting; }
        std::snprintf(b, sizeof b, "\"island_blocks_spec\":%llu,\"island_blocks_interp\":%llu,\"interp_block_fraction\":%.6f,"
                      "\"shapes\":{\"total\":%zu,\"ready\":%u,\"compiling\":%u,\"deferred\":%u},"
                      "\"jit\":{\"entries\":%llu,\"entry_cap\":%u,\"modules_loaded\":%llu,\"compiles\":%llu,\"disk_hits\":%llu,\"failed\":%llu,\"evictions\":%llu,"
                      "\"abandoned\":%llu,\"queued\":%llu,\"deferred\":%llu,\"promoted\":%llu,\"compile_ms_mean\":%.1f,\"compile_ms_max\":%.1f,\"compile_ms_last\":%.1f,"
                      "\"text_bytes_held\":%llu,\"code_bytes_held\":%llu,\"disk_bytes\":%llu,\"disk_cap_bytes\":%llu,\"disk_files_removed\":%llu,\"workers\":%u},",
                      (unsigned long long)islandBlocksSpec, (unsigned long long)islandBlocksInterp, tot ? (double)islandBlocksInterp / (double)tot : 0.0,
                      p.shapes.size(), ready, waiting, deferred,
                      (unsigned long long)js.entries, js.entryCap, (unsigned long long)js.modulesLoaded, (unsigned long long)js.compiles, (unsigned long long)js.diskHits,
                      (unsigned long long)js.failed, (unsigned long long)js.evictions, (unsigned long long)js.abandoned, (unsigned long long)js.queued,
                      (unsigned long long)js.deferred, (unsigned long long)js.promoted, js.compiles ? js.compileMsTotal / (double)js.compiles : 0.0, js.compileMsMax, js.compileMsLast,
                      (unsigned long long)js.sourceBytesHeld, (unsigned long long)js.codeBytesHeld, (unsigned long long)js.diskBytes, (unsigned long long)js.diskCapBytes,
                      (unsigned long long)js.diskFilesRemoved, js.workers);
        s += b;
    }
    kv("sync_poll", syncPoll ? 1 : 0); kv("sync_polls", syncPolls); kv("sync_poll_fallbacks", syncPollFallbacks);
    kv("resident", residentOpt ? 1 : 0); kv("resident_launches", st.residentLaunches); kv("resident_blocks", st.residentBlocks);
    kv("conv_direct_io_sets", convDirectSets); kv("conv_long_sets", convLongSets); kv("conv_long", convLong ? 1 : 0); kv("conv_max_long_tap_rows", convMaxQp);
    kv("num_taps", p.taps.size()); kv("taps_in_sets", p.tapsInSets ? 1 : 0); kv("num_tap_nodes", p.taps.size() + p.tapPairs.size()); kv("num_convs", p.convs.size()); kv("conv_workgroups", p.convWork.size());
    s += "\"level_sizes\":[";
    for (size_t l = 0; l + 1 < p.levelOffsets.size(); ++l) { if (l) s += ","; s += std::to_string(p.levelOffsets[l + 1] - p.levelOffsets[l]); }
    s += "],\"root_ids\":[";
    for (size_t i = 0; i < p.rootIds.size(); ++i) { if (i) s += ","; s += std::to_string(p.rootIds[i]); }
    s += "],\"islands\":[";
    for (size_t i = 0; i < p.islands.size(); ++i) {
        const Island& I = p.islands[i];
        if (i) s += ",";
        s += "{\"tasks\":" + std::to_string(I.numTasks) + ",\"stages\":" + std::to_string(I.numStages) +
             ",\"lds_bytes\":" + std::to_string(I.ldsWords * 4) + ",\"consts\":" + std::to_string(I.numCells) +
             ",\"prog_dwords\":" + std::to_string(I.progDwords) + ",\"copies\":" + std::to_string(I.copies) +
             ",\"stateless\":" + std::to_string(I.stateless) + ",\"phases\":[";
        if (p.islandProg[i]) {   // first stage of each pipeline phase (stage tables of the island's program blob)
            const uint32_t* tab = p.islandProg[i]->blob.data() + I.stageOff + 2 * I.numStages + kWaves * (I.numStages + 1);
            for (uint32_t d = 0; d <= I.copies; ++d) { if (d) s += ","; s += std::to_string(tab[d]); }
        }
        s += "],\"waves\":[";
        for (uint32_t w = 0; w < kWaves; ++w) {   // per program wave: [opcode, stage] of its tasks in program order
            if (w) s += ",";
            s += "[";
            for (uint32_t t = I.waveTask[w]; p.islandProg[i] && t < I.waveTask[w + 1]; ++t) {
                const uint32_t d0 = p.islandProg[i]->blob[t * 8u];
                if (t != I.waveTask[w]) s += ",";
                s += "[" + std::to_string(d0 & 0xFFFFu) + "," + std::to_string((d0 >> 16) & 0xFFu) + "]";
            }
            s += "]";
        }
        s += "]}";
        if (i >= 63) break;
    }
    s += "]}";
    return s;
}

} // namespace elemhip
