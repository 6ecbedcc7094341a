// device.h — plan encoding shared by the host graph compiler (plan.cpp) and the HIP
// block-render kernels (kernels.hip).
//
// Vocabulary
//   node record  : 32 dwords of persistent per-node device storage: [0..7] params written by
//                  the host on setProperty (the reference's per-node atomics / SPSC queues,
//                  e.g. builtins/Core.h:165-166, Delays.h:59-95), [8..31] DSP state that only
//                  the render kernels touch (phasor phase, filter z's, ...).  Records outlive
//                  plan rebuilds and root deactivation; they die in gc() (Runtime.h:220-272).
//   island       : a connected cluster of nodes rendered by ONE workgroup with all of its
//                  intermediate block buffers in LDS; only buffers consumed by another island
//                  are exported to the HBM buffer arena.
//   task         : a group of 1..64 same-opcode nodes of one island stage. Stateless ops run
//                  sample-parallel (64 lanes x samples); stateful recurrences run lane-per-node.
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#ifndef __HIPCC_RTC__
#include "conv_state.h"     // namespace conv: the convolve node's state layout (not part of the run-time compiled text)
#endif

namespace elemhip {

enum : uint32_t {
    kRecDwords   = 32,     // dwords per node record
    kParam0      = 0,      // first param dword
    kState0      = 8,      // first state dword
    kWaves       = 8,      // waves per island workgroup: two per SIMD, so a wave stalled on a serial recurrence leaves issue slots
    kThreads     = 512,
    kSlotWords   = 516,    // LDS words per block-buffer slot: 512 + 4 keeps slots 16-byte aligned and makes
                           // lane-per-node ds_read_b128 (lane stride 516 words = 4 banks mod 64) conflict-free
    kSlot0       = 4,      // first slot word; words 0..3 read as 0.0f
    kMaxBlock    = 512,    // max frames per block the LDS slots are sized for
    kMaxOut      = 256,    // output bus channels per process call
    kMaxHostIn   = 32,     // host input channels addressable by leaf nodes (Types.h:142 uses 32 too)
    kNone        = 0xFFFFFFFFu,
};

// Operand encoding: [31:30] kind, [29:0] value
enum : uint32_t {
    kOpLds   = 0u << 30,   // value = LDS word offset of a block buffer (stride 1)
    kOpConst = 1u << 30,   // value = LDS word offset of one broadcast cell (stride 0)
    kOpHbm   = 2u << 30,   // value = buffer index in the HBM buffer arena (index * blockStride)
    kOpZero  = 3u << 30,   // reads as 0.0f
    kOpKindMask = 3u << 30,
    kOpValMask  = ~(3u << 30),
    // kOpHbm values (and Member::outHbm) with this bit name a buffer of the STREAM ring instead of the block's arena slice:
    // buffers that only carry a block from one wave of an island to another (the blocks a float recurrence reads / writes in
    // a specialised kernel). Block b of an island with D buffer sets uses ring slice b % D, so those lines are rewritten in
    // L2 before they are ever evicted; only exports (read by another island, a later launch or the epilogue) need a slice
    // per block of the launch set.
    kOpStream   = 1u << 29,
};

// Opcodes. The first block mirrors the registry names of runtime/elem/DefaultNodeTypes.h:49-144
// (hot-path subset) plus the wasm-host nodes time/metro/convolve (wasm/Main.cpp:47-61).
enum Op : uint16_t {
    OP_INVALID = 0,
    OP_CONST, OP_SR, OP_IN,
    OP_SIN, OP_COS, OP_TAN, OP_TANH, OP_ASINH, OP_LN, OP_LOG, OP_LOG2,
    OP_CEIL, OP_FLOOR, OP_ROUND, OP_SQRT, OP_EXP, OP_ABS,
    OP_LE, OP_LEQ, OP_GE, OP_GEQ, OP_POW, OP_EQ, OP_AND, OP_OR,
    OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MOD, OP_MIN, OP_MAX,
    OP_ROOT, OP_PHASOR, OP_SPHASOR, OP_COUNTER, OP_ACCUM, OP_LATCH, OP_MAXHOLD, OP_ONCE, OP_SEQ, OP_RAND,
    OP_DELAY, OP_SDELAY, OP_Z, OP_POLE, OP_ENV, OP_BIQUAD, OP_PREWARP, OP_MM1P, OP_SVF, OP_SVFSHELF,
    OP_TAPIN, OP_TAPOUT, OP_SAMPLESEQ, OP_BLEPSAW, OP_BLEPSQUARE, OP_BLEPTRIANGLE,
    OP_TIME, OP_METRO, OP_CONVOLVE,
    OP_TABLE, OP_SEQ2, OP_SPARSEQ2, OP_SAMPLE,
    OP_METER, OP_SNAPSHOT, OP_SCOPE,              // event side-channel (Analyzers.h)
    OP_MCSAMPLE,                                  // one output channel of mc.sample (mc/Sample.h); mc.table and
                                                  // mc.sampleseq channels reuse the table / sampleseq ops (flag in the record)
    OP_SPARSEQ,                                   // SparSeq.h:17-372 (tick-time keyed sequence: loop points, follow, interpolation)
    OP_CAPTURE,                                   // Capture.h:13-104 (gated recording relayed as a "capture" event)
    // plan pseudo-ops
    OP_COPY,          // out = in0 (import HBM->LDS, export LDS->HBM)
    OP_SVF_COEF,      // SVF coefficient pre-pass (a1,a2,a3 as double into the member's scratch), sample-parallel
    OP_SHELF_COEF,    // shelf variant (a1,a2,a3,k,A)
    OP_SAW_SHAPE,     // blepsaw waveform from (phase, frequency), sample-parallel, one stage after the phase recurrence
    OP_SQUARE_SHAPE,  // blepsquare variant
    OP_PHASE,         // constant-frequency phase recurrences of one stage merged into ONE lane-per-node task: `phasor` members
                      // first (Task::s0 of them: step = f * (1 / sr), Core.h:85-136), then the phase halves of blepsaw / blepsquare
                      // (inc = f / sr, Oscillators.h:60-66). Same loop for every lane — phase' = fract(phase + inc) — so a synth
                      // voice's gate phasor and its two oscillators cost one recurrence wave instead of two.
    OP_COUNT_
};

// One node of a task (8 dwords in the island program, 16-byte aligned).
struct Member {
    uint32_t rec;        // node record: local index into the island's staged record table (Island::recOff)
    uint32_t opnd;       // first operand (index into the island's operand array)
    uint32_t nin;        // operand count; kNone => leaf: operands are host inputs 0..nIn-1
    uint32_t outLds;     // LDS word offset of the output slot, or kNone
    uint32_t outHbm;     // HBM arena buffer index to (also) write, or kNone
    uint32_t scratch;    // LDS word offset of op scratch, or kNone
    uint32_t pad0_, pad1_;
#ifdef ELEMHIP_SPEC
    // specialised kernels (island_spec.inc): the member is built from compile-time constants; its first operand codes
    // travel with it (already moved to the block's buffer set), `off` = LDS word offset of that buffer set
    uint32_t sops[6];
    uint32_t off;
    uint32_t gdirect;    // recurrence member: the block goes straight to HBM arena buffer `outHbm` (no LDS copy of it exists)
    uint32_t cnt;        // members of the task (lanes >= cnt mirror the last member's arithmetic but skip its memory traffic)
#endif
};

// 8 dwords in the island program (two 16-byte LDS reads). The second half repeats what the
// common single-member task needs from its member and operand tables, so decoding such a task
// costs one LDS round trip instead of three.
struct Task {
    uint16_t opcode;
    uint8_t  stage;      // barrier epoch inside the island
    uint8_t  flags;      // chain tasks: bit k set = operand k of every member is a broadcast cell
    uint16_t s0, s1;     // sample range [s0, s1) for sample-parallel tasks
    uint32_t first;      // first member (index into the island's member array)
    uint32_t count;      // members in this task
    uint32_t o0, o1;     // member 0: first two operand codes (kOpZero when absent)
    uint16_t outLds16;   // member 0: LDS output word (0xFFFF = none)
    uint16_t nin16;      // member 0: operand count (0xFFFF = leaf / host inputs, 0xFFFE = too many: see member)
    uint32_t outHbm;     // member 0: HBM arena buffer (kNone = none)
};

// An island's program is one contiguous dword blob  [tasks | members | operands | const cells]
// that the workgroup copies into LDS once, so the interpreter never waits on global memory
// for metadata (a dependent scalar load per task cost ~0.7 us each).
struct Island {
    uint32_t progBegin;            // dword offset of the blob in PlanView::prog
    uint32_t progDwords;
    uint32_t numTasks;             // per copy; sorted by (wave, stage); wave w owns tasks [waveTask[w], waveTask[w+1])
    uint32_t waveTask[kWaves + 1];
    uint32_t split;                // > 1: the island is pure sample-parallel and runs as `split` workgroups,
                                   // workgroup k rendering frames [k, k+1) * blockSize / split; 0: no island workgroup (convolve)
    uint32_t memOff;               // dword offsets inside one program copy
    uint32_t opndOff;
    uint32_t cellOff;              // dword offset (whole blob) of the ConstCell table
    uint32_t numCells;             // ConstCell pairs: LDS broadcast cells to fill at start
    uint32_t ldsProg;              // LDS word where the blob is staged (after slots and cells)
    uint32_t ldsWords;             // total dynamic LDS words (blob + stage counters included)
    uint32_t rootRec;              // record of the RootNode whose render sequence owns the island
    uint32_t numStages;
    // multi-block launches (elemhip_process_blocks): a stateful island keeps `copies` blocks in flight, each with
    // its own set of LDS block buffers and its own program copy [tasks | members | operands] addressing that set;
    // stages of one block are ordered by LDS completion counters instead of workgroup barriers (kernels.hip).
    uint32_t copies;
    uint32_t copyDwords;           // dwords per program copy
    uint32_t stageOff;             // dword offset (whole blob) of the stage tables: tasksInStage[S] | prevNonEmpty[S] | begin[4][S+1] | phaseStart[copies+1]
    uint32_t stateless;            // 1: only stateless sample-parallel nodes: blocks of a batch may run in any order / in parallel
    uint32_t ldsCounters;          // LDS word of the completion counters [numStages][copies]
    uint32_t schedOff;             // dword offset (whole blob): walkOffsets[kWaves+1] | walk entries (8 dwords each), 16-byte aligned
    uint32_t ldsNext;              // LDS word of the per-wave next-block counters [kWaves][numStages] of the dataflow walk
    uint32_t recOff;               // dword offset (whole blob) of the island's record table: global record index per local index
    uint32_t numRecs;              // Member.rec is a LOCAL index into that table: the records are staged in LDS for the launch
    uint32_t ldsRecs;              // LDS word of the staged records [numRecs][kRecDwords]
    uint32_t slotArea;             // LDS words of one copy's block buffers: copy d's buffers sit d * slotArea words behind copy 0's
};

// Task flag bit 6 (recurrence tasks of a pipelined island): the task is the only one of its wave, exports nothing and keeps
// its whole state in registers, so it renders every block of the launch itself — publish, wait for the next block's inputs,
// run again — instead of returning to the walk between blocks (island.inc chain_blocks).
constexpr uint32_t kTaskOwnsWave = 0x40u;

struct ConstCell {
    uint32_t ldsWord;    // destination LDS word
    uint32_t rec;        // source record (param dword 0 holds the value)
};

// Root bookkeeping for the epilogue (GraphRenderSequence.h:227-231, 297-308).
struct RootEntry {
    uint32_t rec;        // root record: p0 channel, p1 target gain, p2 step, s8 current gain
    uint32_t hbm;        // HBM arena buffer holding the root's faded output
};

// convolve (wasm/Convolve.h:23-92): one descriptor per node of the plan; rendered by the
// partitioned-FFT kernel in conv.hip instead of an island program.
struct ConvDesc {
    uint32_t rec;        // node record: p0/p1 = device pointer to the node's convolver state (conv:: layout)
    uint32_t inKind;     // 0 no input channel: zero output, state untouched; 1 HBM arena buffer; 2 const (record);
                         // 3 leaf: host input 0 when the block has host inputs, else as 0; 4 silent input;
                         // 5 the host channel named by an `in` record (inIdx) that the planner folded into this launch
    uint32_t inIdx;
    uint32_t outHbm;     // HBM arena buffer receiving the output
    uint32_t rootRec;    // owning root (the node renders only while that root runs)
    uint32_t slices;     // helper workgroups per bin group the plan launches for this node
    uint32_t fuseRootRec; // kNone, or the record of a root folded into this node: its fade is applied here (Core.h:66-78)
    uint32_t pad_;
};

struct TapEntry {
    uint32_t rec;        // tapOut record: p0/p1 shared tap buffer ptr, p2/p3 private delay buffer ptr
    uint32_t rootRec;    // owning root (promotion only while that root is active)
};

// Engine-wide block state kept on the device so a block's launch sequence is replayable.
struct Globals {
    int64_t  sampleTime;   // wasm/Main.cpp:206-215 userData
    uint32_t numSamples;   // frames in the current block (<= blockSize)
    uint32_t numIn;        // host input channels this block
    uint32_t numOut;       // host output channels this block
    uint32_t blockSlot;    // which slot of the output ring the epilogue writes
    uint32_t ringSlots;
    uint32_t blockStride;  // floats between arena buffers (= blockSize)
    float    sampleRateF;
    uint32_t inBlocks;     // multi-block path: input blocks available behind inRing
    double   sampleRate;
    uint64_t trace;        // debug: device pointer to a per-task timestamp log for workgroup 0 of every launch, or 0
    uint64_t inRing;       // multi-block path: device pointer to [inBlocks][numIn][blockStride] host-input blocks; the
                           // epilogue stages the NEXT block's inputs into arena buffers 0..numIn-1 (0 = host copies them)
    uint32_t epiTicket;    // fused epilogue (island_spec.inc spec_epilogue_tail): workgroups of the last level that have finished
    uint32_t pad_;
};

// Device view of a compiled plan (all pointers are device pointers).
struct PlanView {
    const Island*    islands;
    const uint32_t*  levelIslands;   // per launch level: one entry per workgroup = island | (splitIndex << 24)
    const uint32_t*  prog;           // island program blobs
    const RootEntry* roots;          // in render-sequence order
    const TapEntry*  taps;           // in render-sequence order
    const ConvDesc*  convs;          // convolve nodes of the plan
    const uint32_t*  convWork;       // per launch level: one entry per conv workgroup = conv index | role << 16
                                     // (role 0 = the node's main workgroup, role h > 0 = helper h - 1)
    uint32_t numRoots;
    uint32_t numTaps;
    uint32_t numConvs;
    uint32_t pad_;
};

struct Patch {
    uint32_t kind;       // 0: recs[index] = value; 1: once-arm (Core.h:352-363); 2: globals dword
    uint32_t index;      // dword index into the record arena / globals
    uint32_t value;
    uint32_t pad_;
};

// Record layouts (dword offsets). Doubles sit on even dwords.
namespace rec {
enum : uint32_t {
    // generic
    P0 = 0, P1 = 1, P2 = 2, P3 = 3, P4 = 4, P5 = 5, P6 = 6, P7 = 7,
    S0 = 8, S1 = 9, S2 = 10, S3 = 11, S4 = 12, S5 = 13, S6 = 14, S7 = 15,
    // root (Core.h:15-83, helpers/GainFade.h)
    ROOT_CHANNEL = P0, ROOT_TARGET = P1, ROOT_STEP = P2, ROOT_HASIN = P3, ROOT_GAIN = S0,
    // delay / sdelay rings: device pointer (2 dwords), size, length, reset-pending flag
    RING_PTR = P0, RING_SIZE = P2, RING_LEN = P3, RING_RESET = P4, RING_WRITE = S0,
    // seq (Core.h:407-573)
    SEQ_HOLD = P0, SEQ_LOOP = P1, SEQ_OFFSET = P2, SEQ_PTR = P4, SEQ_LEN = P6, SEQ_PENDING = P7,
    SEQ_INDEX = S0, SEQ_HOLDVAL = S1, SEQ_FIRST = S2, SEQ_CHANGE = S3, SEQ_RCHANGE = S4, SEQ_HAVE = S5,
    // taps (Feedback.h)
    TAP_SHARED = P0, TAP_PRIVATE = P2,
    // table (Table.h:17-76): sample buffer pointer + true length
    TBL_BUF = P0, TBL_LEN = P2,
    // sparseq2 (SparSeq2.h:17-141): interpolate flag, event table [len doubles | len floats]
    SPS_INTERP = P0, SPS_SEQ = P4, SPS_LEN = P6,
    // sparseq (SparSeq.h): host-written: offset, follow, interpolate, new-sequence flag, event table [len int32 tick times | len
    // floats], its length, new-loop-points flag + the points, tickInterval in samples (double); state: edge count, samples since
    // the clock edge, held event index (-1: none), change detectors, have-sequence, loop points, pending loop points
    SQ_OFFSET = P0, SQ_FOLLOW = P1, SQ_INTERP = P2, SQ_SEQ_PENDING = P3, SQ_SEQ = P4, SQ_LEN = P6, SQ_LOOP_PENDING = P7,
    SQ_EDGES = S0, SQ_SINCE = S1, SQ_HOLD = S2, SQ_CHANGE = S3, SQ_RCHANGE = S4, SQ_HAVE = S5, SQ_LOOP_START = S6, SQ_LOOP_END = S7,
    SQ_NEW_START = 16, SQ_NEW_END = 17, SQ_TICK = 18, SQ_PEND_FLAG = 20, SQ_PEND_START = 21, SQ_PEND_END = 22,
    // capture (Capture.h): device ring of bitceil(sr) floats, its mask; ring write / read positions (MultiChannelRingBuffer.h), frames
    // in the 128-frame scratch (they sit in the ring ahead of the write position), change detector, relay-ready flag
    CAP_RING = P0, CAP_MASK = P2, CAP_WRITE = 8, CAP_READ = 9, CAP_SCRATCH = 10, CAP_CHANGE = 11, CAP_READY = 12,
    CAP_CHANS = P3,      // mc.capture: capture channels (= children - 1; 0: the mono `capture` node), ring k at CAP_RING + k * (CAP_MASK + 1) floats
    CAP_CH = P4,         // mc.capture: the output channel this record renders (channel 0's record carries the recording state)
    // meter: S0 min, S1 max, S2 readout count; snapshot: S0 previous trigger sample, S1 captured value, S2 capture count
    EVT_A = 8, EVT_B = 9, EVT_COUNT = 10,
    // ... and a per-BLOCK readout log (r05): ring of 4-dword entries {block number of this node, a, b, -} behind EVT_LOG with EVT_LOGMASK + 1
    // entries — meter: one entry per block (min, max), entry index = block number; snapshot: one entry per block that latched (value of the
    // block's last latch, latches in the block), EVT_LOGN entries so far, EVT_BLK blocks so far. The reference's offline caller relays events
    // after EVERY block (offline-renderer/index.ts:112-120); the log lets a host that rendered a whole launch set do the same afterwards.
    EVT_LOG = P0, EVT_LOGMASK = P2, EVT_BLK = 11, EVT_LOGN = 12,
    // scope: device ring [4 channels][8192] (MultiChannelRingBuffer.h), write / read positions shared with the host relay
    SCP_RING = P0, SCP_WRITE = 8, SCP_READ = 9,
    // sample (Sample.h:22-231): buffer, length, new-buffer flag, mode (0 trigger, 1 gate, 2 loop), offsets, gain smoothing alpha;
    // state: change detector, current reader, two readers {target gain, gain, pos (double)}
    SMP_BUF = P0, SMP_LEN = P2, SMP_PENDING = P3, SMP_MODE = P4, SMP_START = P5, SMP_STOP = P6, SMP_ALPHA = P7,
    SMP_CHANGE = 8, SMP_CURRENT = 9, SMP_HAVE = 10, SMP_READER0 = 12, SMP_READER_DWORDS = 4,   // target, gain, pos lo, pos hi
    // mc.sample channel (mc/Sample.h:17-291): params as `sample` (P7 unused); state: change detector, current reader, have-buffer,
    // pending reset (Runtime::reset), two readers {target, gain, step, pos (double), start offset, stop offset, loop}, playback rate (double)
    MCS_RESET = 11, MCS_READER0 = 12, MCS_READER_DWORDS = 8, MCS_RATE = 28,
    // seq2 (Seq2.h:35-166) shares seq's parameter layout; state: S0 edge count, S3/S4 change detectors, S5 have-sequence
    // convolve: device pointer to the conv:: state
    CONV_STATE = P0,
    // sampleseq (SampleSeq.h:169-404): sample buffer, event table [len doubles | len floats], k-rate state, two readers
    SSQ_BUF = P0, SSQ_BUFLEN = P2, SSQ_BUFPENDING = P3, SSQ_SEQ = P4, SSQ_SEQLEN = P6, SSQ_SEQPENDING = P7,
    SSQ_DUR = 8, SSQ_RTDUR = 10, SSQ_PREV = 12, SSQ_NEXT = 13, SSQ_ACTIVE = 14, SSQ_FLAGS = 15,
    SSQ_READER0 = 16, SSQ_READER_DWORDS = 8,   // per reader: gain, target, step, position, startTime(2), bufferSize, -
    // SSQ_FLAGS bit 2: the mc.sampleseq flavour (mc/SampleSeq.h): readers fade with elem::GainFade (helpers/GainFade.h) whose
    // fade-in / fade-out steps sit in the spare dword 7 of reader 0 / reader 1
};
}

} // namespace elemhip
