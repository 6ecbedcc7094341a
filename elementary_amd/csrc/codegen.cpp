// codegen.cpp — writes the per-island-SHAPE part of a specialised kernel (island_spec.inc explains the scheme).
//
// Input: one island's program exactly as the planner built it for the interpreter (tasks sorted by (wave, stage),
// members, operand codes of buffer set 0, stage tables). Output: HIP source text that declares, as compile-time
// constants, what the interpreter reads from LDS tables at run time. Nothing that differs between two islands of the
// same shape appears in the text — record numbers are island-local, HBM arena indices and the root record are read
// from the staged tables at run time — so 256 voices of one synth patch share one kernel (the text is the cache key).
#include <cstdio>
#include <sstream>
#include <string>
#include <vector>

#include "engine.h"

namespace elemhip {

uint32_t leafArityForCodegen(uint16_t op);   // plan.cpp

static std::string u(uint32_t v) {
    if (v == kNone) return "kNone";
    char b[32]; std::snprintf(b, sizeof b, "%uu", v); return b;
}

// Arena buffer indices differ between the instances of a shape: the text names position k of the island's arena table
// (appended to its program blob, staged in LDS with it) instead of the index itself.
// A stream-ring buffer keeps its kOpStream bit in the TEXT (a compile-time fact of the shape: which base pointer an access
// uses folds away); the table entry holds the index inside the ring slice.
static std::string arenaRef(const SpecProgram& sp, uint32_t abs, uint32_t tabWord) {
    for (size_t k = 0; k < sp.hbmTab.size(); ++k)
        if (sp.hbmTab[k] == abs) {
            const std::string e = "UNI(ldsu(" + u(tabWord + (uint32_t)k) + "))";
            return (abs & kOpStream) ? "(kOpStream | (" + e + " & (kOpStream - 1u)))" : "(" + e + " & (kOpStream - 1u))";
        }
    return "0u";
}

// operand code -> expression of (c, off)
static std::string opndExpr(const SpecProgram& sp, uint32_t code, uint32_t tabWord) {
    const uint32_t kind = code & kOpKindMask, v = code & kOpValMask;
    if (kind == kOpLds)   return "((kOpLds | " + u(v) + ") + off)";
    if (kind == kOpConst) return "(kOpConst | " + u(v) + ")";
    if (kind == kOpHbm)   return "(kOpHbm | (" + arenaRef(sp, v, tabWord) + " & kOpValMask))";   // (the mask lets the compiler see the kind bits)
    return "(uint32_t)kOpZero";
}

std::string emitSpecSource(const Island& I, const std::vector<Task>& tasks, const SpecProgram& sp,
                           const std::vector<uint32_t>& stageTab, uint32_t blockSize, uint32_t wavesPerEu) {
    const std::vector<Member>& members = sp.members;
    const std::vector<uint32_t>& operands = sp.operands;
    const uint32_t tabWord = I.ldsProg + I.recOff + I.numRecs;   // LDS word of the staged arena table
    std::ostringstream o;
    const uint32_t S = I.numStages;
    uint32_t lastStage = S - 1u;
    o << "namespace gen {\n";
    // ---- tasks ----
    for (size_t q = 0; q < tasks.size(); ++q) {
        const Task& t = tasks[q];
        o << "struct T" << q << " {\n";
        o << "    static constexpr uint32_t opcode = " << t.opcode << "u, flags = " << (uint32_t)(t.flags & 0x3Fu) << "u, s0 = " << t.s0
          << "u, s1 = " << t.s1 << "u, count = " << t.count << "u;\n";
        o << "    static constexpr bool gdirect = " << (sp.gdirect[q] ? "true" : "false") << ";\n";
        o << "    template <int K> static __device__ __forceinline__ Member member(const Ctx& c, uint32_t off) {\n        Member m;\n";
        for (uint32_t k = 0; k < t.count; ++k) {
            const uint32_t mi = t.first + k;
            const Member& m = members[mi];
            const uint32_t nops = m.nin == kNone ? std::min<uint32_t>(leafArityForCodegen(t.opcode), kMaxHostIn) : m.nin;
            o << "        " << (k ? "else " : "") << "if constexpr (K == " << k << ") {\n";
            o << "            m.rec = " << u(m.rec) << "; m.opnd = " << u(m.opnd) << "; m.nin = " << u(m.nin) << ";\n";
            o << "            m.outLds = " << (m.outLds == kNone ? std::string("kNone") : u(m.outLds) + " + off") << ";\n";
            o << "            m.outHbm = " << (m.outHbm == kNone ? std::string("kNone") : arenaRef(sp, m.outHbm, tabWord)) << ";\n";
            o << "            m.scratch = " << (m.scratch == kNone ? std::string("kNone") : u(m.scratch) + " + off") << ";\n";
            for (uint32_t j = 0; j < 6; ++j) {
                std::string e = j < nops ? opndExpr(sp, operands[m.opnd + j], tabWord) : std::string("(uint32_t)kOpZero");
                if (j == 5 && nops <= 5 && (sp.phaseOp[mi] & kOpKindMask) == kOpHbm) e = opndExpr(sp, sp.phaseOp[mi], tabWord);   // streamed oscillator phase
                o << "            m.sops[" << j << "] = " << e << ";\n";
            }
            o << "            m.gdirect = " << (sp.gdirect[q] ? "1u" : "0u") << "; m.cnt = " << t.count << "u;\n";
            o << "        }\n";
        }
        o << "        m.pad0_ = m.sops[0]; m.pad1_ = m.sops[1]; m.off = off;\n        return m;\n    }\n";
        o << "    template <int K> static constexpr bool hasHbm() { return ";
        for (uint32_t k = 0; k < t.count; ++k) o << (k ? " : " : "") << "K == " << k << " ? " << (members[t.first + k].outHbm != kNone ? "true" : "false");
        o << " : false; }\n";
        o << "    static __device__ __forceinline__ Member member_lane(const Ctx& c, uint32_t off, uint32_t lane) {\n        Member m = member<0>(c, off);\n";
        for (uint32_t k = 1; k < t.count && k < 64u; ++k) o << "        m = spec_sel(lane >= " << k << "u, member<" << k << ">(c, off), m);\n";
        o << "        return m;\n    }\n};\n";
    }
    // ---- (wave, stage) slots, later stages first ----
    int waveSlots[kWaves];
    for (uint32_t w = 0; w < kWaves; ++w) {
        std::vector<uint32_t> stages;
        for (uint32_t q = I.waveTask[w]; q < I.waveTask[w + 1]; ++q)
            if (stages.empty() || stages.back() != tasks[q].stage) stages.push_back(tasks[q].stage);
        waveSlots[w] = (int)stages.size();
        for (int j = 0; j < (int)stages.size(); ++j) {
            const uint32_t st = stages[stages.size() - 1 - (size_t)j];
            const uint32_t prev = stageTab[S + st];
            uint32_t ntasks = 0;
            bool writes = false;     // the slot stores into arena buffers (read by other waves of the same launch, or by later launches)
            bool allDirect = true;   // ... and every task of it is a streamed recurrence (its block goes straight to the arena)
            bool plainFamily = true; // ... whose whole state lives in registers across blocks (island_ops.inc chain_blocks callers)
            o << "template <> struct Slot<" << w << ", " << j << "> {\n";
            std::ostringstream body;
            for (uint32_t q = I.waveTask[w]; q < I.waveTask[w + 1]; ++q)
                if (tasks[q].stage == st) {
                    body << "        spec_task<T" << q << ">(c, off);\n"; ++ntasks;
                    if (!sp.gdirect[q]) allDirect = false;
                    switch (tasks[q].opcode) {
                        case OP_PHASOR: case OP_SPHASOR: case OP_POLE: case OP_ENV: case OP_BIQUAD: case OP_COUNTER: case OP_ACCUM: case OP_LATCH:
                        case OP_MAXHOLD: case OP_PHASE: break;
                        case OP_BLEPSAW: case OP_BLEPSQUARE: if (!(tasks[q].flags & 1u)) plainFamily = false; break;
                        default: plainFamily = false;
                    }
                    // every member's operands: a broadcast cell, zero, or an arena stream (never an LDS buffer)
                    for (uint32_t k = 0; k < tasks[q].count; ++k) {
                        const Member& mm = members[tasks[q].first + k];
                        if (mm.nin == kNone) { plainFamily = false; continue; }
                        for (uint32_t j = 0; j < mm.nin; ++j) if ((operands[mm.opnd + j] & kOpKindMask) == kOpLds) plainFamily = false;
                    }
                    for (uint32_t k = 0; k < tasks[q].count; ++k) if (members[tasks[q].first + k].outHbm != kNone) writes = true;
                }
            o << "    static constexpr uint32_t stage = " << st << "u, prev = " << u(prev) << ", prevT = " << (prev == kNone ? 0u : stageTab[prev])
              << "u, ntasks = " << ntasks << "u;\n    static constexpr bool writesStreams = " << (writes ? "true" : "false")
              << ", deferPublish = " << (writes && allDirect ? "true" : "false")
              << ", persistent = " << (stages.size() == 1 && ntasks == 1 && writes && allDirect && plainFamily && I.copies > 1 ? "true" : "false") << ";\n";
            o << "    static __device__ __forceinline__ void run(const Ctx& c, uint32_t off) {\n" << body.str() << "    }\n};\n";
        }
    }
    o << "struct P {\n    static constexpr uint32_t S = " << S << "u, D = " << I.copies << "u, slotArea = " << I.slotArea << "u, ldsProg = " << I.ldsProg
      << "u, memOff = " << I.memOff << "u, opndOff = " << I.opndOff << "u, cellOff = " << I.cellOff << "u, numCells = " << I.numCells
      << "u, recOff = " << I.recOff << "u, numRecs = " << I.numRecs << "u, progDwords = " << I.progDwords << "u, ldsCounters = " << I.ldsCounters
      << "u, ldsRecs = " << I.ldsRecs << "u, specOpndOff = " << (I.recOff + I.numRecs + (uint32_t)sp.hbmTab.size()) << "u, lastStage = " << lastStage << "u, lastT = " << stageTab[lastStage] << "u, block = " << blockSize << "u, split = " << std::max(1u, I.split) << "u;\n";
    o << "    static constexpr bool stateless = " << (I.stateless ? "true" : "false") << ";\n";
    // A stateless island whose eight waves run the SAME tasks of every stage on consecutive frame runs (a mixer: 64-frame runs of one
    // 128-input fold) gets ONE copy of each task body, the wave's frame offset a run-time value: eight per-wave instantiations of a
    // 128-child fold were 70 KB of straight-line code that a single-block launch executes once, from a cold instruction cache
    // (37 us for the mixer level of a synchronous process() call against 18 us through the interpreter kernel).
    bool uniform = I.stateless != 0u;
    std::vector<std::vector<uint32_t>> perStage(S);          // wave 0's tasks per stage
    uint32_t runLen = 0;
    if (uniform) {
        for (uint32_t w = 0; w < kWaves && uniform; ++w) {
            std::vector<std::vector<uint32_t>> mine(S);
            for (uint32_t q = I.waveTask[w]; q < I.waveTask[w + 1]; ++q) mine[tasks[q].stage].push_back(q);
            if (w == 0) { perStage = mine; for (auto& v : mine) for (uint32_t q : v) { const uint32_t len = (uint32_t)tasks[q].s1 - tasks[q].s0; if (!runLen) runLen = len; if (len != runLen || tasks[q].s0 != 0) uniform = false; } continue; }
            for (uint32_t st = 0; st < S && uniform; ++st) {
                if (mine[st].size() != perStage[st].size()) { uniform = false; break; }
                for (size_t k = 0; k < mine[st].size(); ++k) {
                    const Task& a = tasks[perStage[st][k]]; const Task& b = tasks[mine[st][k]];
                    if (a.opcode != b.opcode || a.first != b.first || a.count != b.count || (a.flags & 0x3Fu) != (b.flags & 0x3Fu) ||
                        (uint32_t)b.s0 != w * runLen || (uint32_t)b.s1 != (w + 1u) * runLen) uniform = false;
                }
            }
        }
        if (!runLen) uniform = false;
    }
    o << "    static constexpr bool uniform = " << (uniform ? "true" : "false") << ";\n";
    o << "    template <int STAGE> static __device__ __forceinline__ void ustage(const Ctx& c, uint32_t wave) {\n";
    if (uniform)
        for (uint32_t st = 0; st < S; ++st) {
            o << "        if constexpr (STAGE == " << st << ") {\n";
            for (uint32_t q : perStage[st]) o << "            spec_task_at<T" << q << ">(c, wave * " << runLen << "u);\n";
            o << "        }\n";
        }
    o << "    }\n";
    o << "    static constexpr int waveSlots[8] = {";
    for (uint32_t w = 0; w < kWaves; ++w) o << (w ? ", " : "") << waveSlots[w];
    o << "};\n};\n} // namespace gen\n";
    // option "spec_waves_per_eu" (0: the compiler's choice): ask for this many waves per SIMD, i.e. cap the registers so that TWO eight-wave
    // workgroups fit a CU (4 -> 128 VGPRs); with islands of <= 80 KB of LDS a level of more islands than CUs then renders in one round
    o << "extern \"C\" __global__ __launch_bounds__(512) ";
    if (wavesPerEu) o << "__attribute__((amdgpu_waves_per_eu(" << wavesPerEu << ", " << wavesPerEu << "))) ";
    o << "void elemhip_spec_island(PlanView pv, uint32_t* recs, float* hbm, const Globals* g,\n"
         "        const uint32_t* lcg, const uint32_t* islandList, uint32_t batch, uint32_t arenaFloats, uint32_t streamBase, uint32_t streamSlice,\n"
         "        uint32_t epiGroups, float* epiOut, uint32_t* epiFlag, uint32_t epiValue) {\n"
         "    spec_island_main<gen::P>(pv, recs, hbm, g, lcg, islandList, batch, arenaFloats, streamBase, streamSlice);\n"
         "    spec_epilogue_tail(pv, recs, hbm, g, epiGroups, epiOut, epiFlag, epiValue);\n}\n";
    return o.str();
}

} // namespace elemhip
