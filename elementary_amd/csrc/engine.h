// engine.h — host side of the HIP block-render engine.
//
// `Engine` re-creates the behaviour of `elem::Runtime<float>` (runtime/elem/Runtime.h:39-153) on
// top of device-resident node records and compiled render plans:
//   applyInstructions  -> Engine::apply        (Runtime.h:170-218)
//   process            -> Engine::process      (Runtime.h:274-290)
//   gc / reset / shared resources              (Runtime.h:220-272, 448-477)
// The render sequence the reference builds as a list of closures (Runtime.h:520-577,
// GraphRenderSequence.h:107-187) becomes a `Plan`: islands of nodes rendered by one workgroup
// each, grouped into launch levels (plan.cpp).
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <deque>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "device.h"
#include "json.h"
#include "jit.h"

namespace elemhip {

// Return codes 0..8 are the reference's (runtime/elem/Types.h:51-86); >= 100 are ours.
enum ReturnCode : int {
    kOk = 0, kUnknownNodeType = 1, kNodeNotFound = 2, kNodeAlreadyExists = 3, kNodeTypeAlreadyExists = 4,
    kInvalidPropertyType = 5, kInvalidPropertyValue = 6, kInvariantViolation = 7, kInvalidInstructionFormat = 8,
    kHipError = 100, kNoDevice = 101, kBlockTooLarge = 102, kTooManyChannels = 103, kUnsupportedGraph = 104,
    kJsonParseError = 105,
};
const char* describe(int code);

// Host call-out node types (Runtime::registerNodeType, runtime/elem/Runtime.h:105-106; GraphNode.h:19-96): a node whose
// process() runs on the CPU between two launch levels. OP_HOST never reaches a kernel, so it is not a device opcode.
constexpr uint16_t OP_HOST = 0x7F00;
struct HostVTable {
    void* (*create)(int32_t nodeId, double sampleRate, int blockSize, void* user) = nullptr;
    void  (*destroy)(void* node, void* user) = nullptr;
    int   (*setProperty)(void* node, const char* key, const char* jsonValue, void* user) = nullptr;
    void  (*process)(void* node, const float* const* in, size_t nIn, float* out, size_t numSamples, int64_t sampleTime, int active, void* user) = nullptr;
    void  (*reset)(void* node, void* user) = nullptr;
    void* user = nullptr;
};

struct DevBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
};

// Shared resource = immutable float channel arrays (SharedResource.h:15-24, AudioBufferResource.h),
// mirrored into device memory on first use.
struct Resource {
    std::vector<std::vector<float>> channels;
    DevBuf dev;                 // channel 0, uploaded lazily
    std::vector<DevBuf> devCh;  // channels >= 1 (multi-output nodes), uploaded lazily; devCh[0] unused
    bool isTap = false;         // created by getTapResource (mutable feedback buffer)
};
using ResourcePtr = std::shared_ptr<Resource>;

struct Node;
// (`src` / `srcEpoch`: the planner's memo of nodes.find(source) — valid while Engine::nodesEpoch, which every erase from the node
//  table bumps, equals srcEpoch; node addresses are stable under insertion)
struct Inlet  { int32_t source; uint32_t channel; mutable Node* src = nullptr; mutable uint32_t srcEpoch = 0; };
struct Outlet { int32_t dest; uint32_t channel; };

struct Node {
    int32_t id = 0;
    uint32_t planVisited = 0, planOnStack = 0;   // plan-build scratch (PlanBuilder::traverse): epoch marks instead of hash sets
    int32_t planIdx = -1;                        // ... planner entry of output channel 0 in the build whose epoch is planVisited
    uint32_t planChans = 0;                      // ... and how many channel entries follow it
    uint16_t op = OP_INVALID;
    uint32_t rec = kNone;
    std::vector<Inlet> inlets;
    std::vector<Outlet> outlets;
    std::map<std::string, Value> props;
    // RootNode / GainFade host mirror (helpers/GainFade.h): the device advances the same floats
    float gain = 0.0f, target = 1.0f, step = 0.0f, inStep = 0.0f, outStep = 0.0f;
    int channel = -1;
    // device-side resources owned by the node
    DevBuf ring;                // delay / sdelay ring, tapOut private buffer, seq data
    ResourcePtr res;            // tap buffer / sample data held by the node
    uint32_t eventCount = 0;    // meter / snapshot: readouts already relayed by processQueuedEvents
    uint32_t logRelayed = 0;    // snapshot: entries of the per-block readout log already relayed
    uint32_t convSlices = 1;    // convolve: helper slices its current impulse response wants (conv.hip)
    uint32_t convQp = 0, convHistBlocks = 0, convP = 0;   // convolve: long-partition tap rows (0: none), blocks of its input ring, 512-partitions (conv_long.inc)
    bool mc = false;            // multi-output node (mc.*): one record per output channel, planned as one entry per channel
    std::vector<uint32_t> chanRecs;   // records of output channels 1, 2, ... (allocated when a plan first needs them)
    std::vector<std::vector<float>> relayCh;   // mc.capture: one relay per capture channel (pendingEventData, mc/Capture.h:152)
    std::vector<float> relay;   // OP_CAPTURE: samples drained from the device ring, waiting for the gate's falling edge (Capture.h:102)
    void* hostInst = nullptr;   // OP_HOST: the instance its type's create() returned
    const HostVTable* hostVt = nullptr;
};

struct Plan;   // plan.cpp

// Generated text of one island-program signature (codegen.cpp) and, once computed, its kernel cache key (jit.cpp). Shared by
// Engine::specTextCache and the plans being built; the key is written under the control lock.
struct SpecText {
    std::string text;
    std::string key;          // empty until buildPlan first needs it
    uint32_t keyLdsWords = 0; // the LDS footprint the key was computed for
};

// One island's scheduled program as the planner left it (plan.cpp "island program cache"): an island whose nodes, edges,
// arena positions and options are unchanged at the next build takes these instead of being scheduled again.
// Table buffers of retired plans, kept for the next plan (a live graph re-plans dozens of times a second; hipMalloc / hipFree
// cost more than the upload, and hipFree synchronises the device). A buffer comes back when its plan dies, which is after the
// synchronize that follows the plan's last launch (Engine::freeDeferred) or before any launch (a pending plan replaced).
struct TablePool {
    std::mutex m;
    std::vector<DevBuf> free;
    DevBuf take(size_t bytes);           // a pooled buffer of at least `bytes` (not more than 4x), or a fresh allocation; ptr null = out of memory
    void give(DevBuf b);
    ~TablePool();
};

// Device home of the island programs. Blobs are appended and never rewritten, so a re-plan uploads only the programs of the
// islands it had to schedule (the replaced voice) while kernels of the current plan keep reading their own blobs; unchanged
// islands are referenced where they already sit. Replaced islands leave garbage behind: when the heap runs out the engine
// starts a fresh one and forgets the cache (plans keep the heap they point into alive).
struct ProgHeap {
    uint32_t* dev = nullptr;             // null on a dry engine (offsets are still handed out)
    size_t capDwords = 0, usedDwords = 0;
    ~ProgHeap();
};

struct IslandProgram {
    Island I;                            // progBegin / rootRec are re-made per plan
    std::vector<uint32_t> blob;          // host copy of the program (16-byte padded): describePlan, plan_cache = 2
    std::vector<uint32_t> members;       // (node id, opcode, record, arena buffer) of every member in render order: checked on a cache hit (the key is a hash)
    // relocation (plan.cpp, "same island, other nodes"): the records and arena buffers the program names, in the order a canonical
    // walk of the island meets them, and the first stream buffer it was given — an island of the same STRUCTURE (another voice of
    // the patch) takes this program with the i-th record / buffer replaced by its own i-th
    std::vector<uint32_t> canonRecs, canonHbms;
    uint32_t streamStart = 0;
    uint32_t specHbmTab = 0;             // entries of the specialised variant's arena table (behind the record table; its operand table follows)
    uint64_t shapeKey = 0;
    std::shared_ptr<ProgHeap> heap;      // where the device copy lives ...
    uint32_t heapBegin = 0;              // ... as a dword offset (= Island::progBegin of every plan that uses it)
    std::shared_ptr<SpecText> spec;      // specialised-kernel text of its shape (null: none)
    uint32_t numMembers = 0, numOperands = 0, streamDelta = 0;
};

struct Stats {
    uint64_t blocksRendered = 0;
    uint64_t plansBuilt = 0;
    double   lastPlanBuildMs = 0.0;
    uint32_t numIslands = 0, numLevels = 0, numTasks = 0, numNodesInPlan = 0, maxLdsBytes = 0, numHbmBuffers = 0;
    uint64_t graphReplays = 0, graphCaptures = 0, batchLaunches = 0;
    uint64_t specFadeBlocks = 0;         // blocks rendered by the specialised kernels while root fades were running (per-block epilogue)
    uint64_t idleLaunchesSkipped = 0;    // launches left out because no root of their islands ran
    uint64_t fusedEpilogues = 0;         // launch sets of one whose last level kernel ran the epilogue
    uint64_t progHeaps = 0;              // program heaps started (1 = the first still serves)
    uint64_t planIslandsReused = 0, planIslandsScheduled = 0, planCacheMismatches = 0;   // island program cache (plan.cpp)
    uint64_t planIslandsRelocated = 0, planRelocationMismatches = 0;                      // ... programs taken from a structural twin
    uint64_t specLaunches = 0;             // launches of run-time specialised island kernels
    uint32_t specShapes = 0, specIslands = 0;
    double   lastJitWaitMs = 0.0;
    double   lastGraphCaptureMs = 0.0;     // capture + instantiate of the per-block hipGraph of the current plan
    uint64_t residentLaunches = 0, residentBlocks = 0;   // option "resident": launches of the resident kernel / blocks it rendered
};

struct ResidentCtl;     // launch.h
struct RenderGuard;     // engine.cpp

class Engine {
    friend struct RenderGuard;
public:
    Engine(double sampleRate, int blockSize, int device);
    ~Engine();
    int initError() const { return initErr; }

    int apply(const Value& batch);
    int createNode(int32_t id, const std::string& type);
    int appendChild(int32_t parent, int32_t child, int32_t channel);
    int setProperty(int32_t id, const std::string& key, const Value& v);
    int activateRoots(const std::vector<int32_t>& ids, bool malformedTail = false);
    int commit(std::unique_lock<std::mutex>& renderLock);

    // one block, host buffers (Runtime::process)
    int process(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n, int64_t sampleTime);
    int processSlice(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n, int64_t sampleTime);
    int processSliceLocked(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n, int64_t sampleTime, bool adopt);   // n <= blockSize
    // numBlocks consecutive full blocks, device-resident output `outDev[block][nOut][blockSize]`
    // (may be null: render only) and optional device-resident input `inDev[block][nIn][blockSize]`
    int processBlocks(const float* inDev, size_t nIn, float* outDev, size_t nOut, size_t numBlocks, int64_t sampleTime);
    // the same for HOST buffers, planar like Runtime::process (Runtime.h:51-57) over `numFrames` = any number of frames: the
    // offline caller's block loop (offline-renderer/index.ts:87-133) in one call. Renders ceil(numFrames / blockSize) full
    // blocks (a short tail of the inputs is zero-padded, the outputs receive numFrames frames). Launch sets are staged
    // through pinned double buffers: the D2H of set k and the H2D of set k + 1 run on a copy stream while set k + 1 renders.
    int processBlocksHost(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t numFrames, int64_t sampleTime);
    // render `numBlocks` blocks with a HIP event pair around every kernel launch; msOut[l] = mean
    // duration of launch level l (l < numLevels), msOut[numLevels] = epilogue. Returns levels + 1.
    int timeLaunches(size_t nOut, size_t numBlocks, float* msOut, size_t cap);
    // debug: render one block of launch level `level` tracing workgroup 0; out = 4 waves x 192 u64
    int traceLevel(size_t nOut, uint32_t level, unsigned long long* out, size_t cap);

    bool addSharedResource(const std::string& name, const float* const* ch, size_t nCh, size_t nSamples);
    void pruneSharedResources();
    size_t gc(int32_t* out, size_t cap);
    size_t lastGc(int32_t* out, size_t cap);          // the ids the most recent gc() pruned (ascending)
    bool hasNode(int32_t id);
    void reset();
    // Runtime::processQueuedEvents (Runtime.h:64, 437-446): relays the newest meter / snapshot readout of every such node
    // of the current render sequence whose root is active. cb(type, json payload, user).
    // `blockwise`: every block's events since the last relay in block order, as a host that relayed after each block would have
    // got them (offline-renderer/index.ts:112-120), from the kernels' per-block readout logs
    int processQueuedEvents(void (*cb)(const char*, const char*, void*), void* user, bool blockwise = false);
    uint32_t eventWindowBlocks();          // blocks a blockwise relay window may span and still be exact for the newest plan's event nodes
    void setStream(hipStream_t s);
    const Stats& stats() const { return st; }
    // dry-engine introspection for host-logic tests: adopt the pending plan and describe it as JSON
    std::string describePlan();
    int setOption(const std::string& key, double value);
    uint32_t lastTimedBatch() const { return lastTimeBatch; }
    // option "profile_launches": HIP event pairs around every multi-block launch of processBlocks (same stream, inside
    // whatever region the caller times). msOut[l] = summed ms of level l, msOut[levels] = epilogue; launchSets = sets timed.
    int registerNodeType(const std::string& type, const HostVTable& vt);   // Runtime.h:480-487
    std::string snapshotJson();                        // Runtime::snapshot (Runtime.h:110, 489-498)
    std::string sharedResourceKeysJson();              // Runtime::getSharedResourceMapKeys (Runtime.h:94)
    int specInfo(size_t k, std::string* source, std::string* log, int* state, uint32_t* islands);
    int launchProfile(double* msOut, size_t cap, uint64_t* launchSets, uint64_t* blocks);

private:
    friend struct PlanBuilder;
    int initErr = 0;
    bool dry = false;                      // device == -1: host logic only, cannot render
    double sampleRate;
    int blockSize;                         // the engine's block: <= kMaxBlock frames (one LDS slot per buffer)
    std::vector<int32_t> tapNodeIds;       // every tapIn / tapOut node alive (setTapSlice)
    size_t tapSliceOff = 0;                // frames: where in the shared tap buffers the tap records point right now
    void setTapSlice(size_t off);
    int hostBlockSize = 0;                 // what the host created the runtime with: blockSize, or a multiple of kMaxBlock rendered in slices
    int device;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    // Two locks. `ctl` serialises the control plane (applyInstructions, gc, resources, options, snapshots ...) and owns
    // the node table's structure; `mu` guards what the render calls touch (current / pending plan, patch list, record
    // shadow, device buffers, stream). Order: ctl, then mu. process* take `mu` only; a commit drops `mu` while it plans.
    std::mutex ctl;
    std::mutex mu;
    Stats st;

    std::unordered_map<int32_t, Node> nodes;
    uint32_t nodesEpoch = 1;               // bumped whenever a node leaves `nodes` (Inlet::src memos of the planner)
    std::shared_ptr<TablePool> tablePool = std::make_shared<TablePool>();
    std::shared_ptr<ProgHeap> progHeap;    // island programs on the device (plan.cpp); owned by the mutator side (`ctl`)
    std::unordered_map<uint64_t, std::shared_ptr<IslandProgram>> islandShapeCache;   // structure -> a program of that structure (relocated for its peers)
    bool relocatePrograms = true;          // option "plan_relocate"
    size_t lastPlanProgDwords = 0, progHeapCap = 0;
    std::set<int32_t> currentRoots;
    std::unordered_map<std::string, ResourcePtr> resources;
    std::unordered_map<std::string, std::unique_ptr<HostVTable>> hostTypes;
    std::vector<float> hostIn, hostOut;    // staging for call-out nodes
    // generated text per island-program signature: 256 voices (and every re-plan of a live graph) format their text once
    // (the kernel cache key of a text lives NEXT TO the text, in the same shared object: nothing is keyed by an address)
    std::unordered_map<uint64_t, std::shared_ptr<SpecText>> specTextCache;
    // per-island programs of earlier builds (plan.cpp "island program cache"); same locking as specTextCache
    std::unordered_map<uint64_t, std::shared_ptr<IslandProgram>> islandCache;
    int planCache = 1;                     // 0 off, 1 reuse unchanged islands' programs, 2 schedule anyway and compare (tests)
    uint32_t planEpoch = 0;                // PlanBuilder::traverse marks
    int64_t curBlockTime = 0;              // sample time of the block being enqueued (call-out nodes get it as userData)
    bool shouldRebuild = false;
    bool rebuildOwed = false;              // a commit failed to build its plan: the next commit retries even without ACTIVATE_ROOTS
    std::vector<int32_t> lastPruned;
    bool planStale = false;                // a property changed the launch shape of the plan (convolve IR length)

    // record arena
    uint32_t* dRecs = nullptr;
    uint32_t recCapacity = 0;
    std::vector<uint32_t> shadow;          // host copy of every record as last written by the host
    std::vector<uint32_t> freeRecs;
    uint32_t nextRec = 0;
    std::vector<uint32_t> freshRecs;       // records to upload whole before the next block
    std::vector<std::pair<uint32_t, uint32_t>> recClones;   // (channel-0 record, new channel record): device-side copy of the live state at the next flush
    std::vector<uint8_t> freshFlag;        // freshFlag[rec] != 0 while rec is in freshRecs
    std::vector<Patch> patches;            // param patches to apply before the next block
    Patch* hPatches = nullptr;             // pinned staging
    uint32_t patchCap = 0;
    size_t patchCursor = 0;                // next free staging slot (rewound after a stream sync)

    // device globals, host mirror
    Globals* dGlobals = nullptr;
    Globals hGlobals{};
    uint32_t* dLcg = nullptr;

    // buffers
    float* dHbm = nullptr; size_t hbmBuffers = 0;
    float* dOutRing = nullptr; size_t outRingFloats = 0;
    float* hOut = nullptr; size_t hOutFloats = 0;     // pinned
    float* hOutDev = nullptr;                          // hOut as the device sees it (mapped)
    float* hIn = nullptr; size_t hInFloats = 0;       // pinned
    std::vector<void*> deferredFree;
    // the event relay's snapshot (processQueuedEvents): records copied device-side in stream order, fetched on a stream of its own
    hipStream_t relayStream = nullptr; hipEvent_t evRelay = nullptr;
    uint8_t* dRelay = nullptr; uint8_t* hRelay = nullptr; size_t relayBytes = 0;
    uint64_t relayBlocksMark = 0;          // Stats::blocksRendered at the last relay: a blockwise relay's window starts here
    // A host block longer than the engine's is k slices = k engine blocks, and the reference's nodes queue their readouts per HOST block
    // (one meter readout over all its frames, Analyzers.h:38-39): Stats::blocksRendered at the end of every host block rendered since the
    // last relay (kept only while hostBlockSize != blockSize; `mu` held), so that a relay can put the slices back together.
    std::deque<uint64_t> hostBlockEnds;
    void noteHostBlockEnd() { if (hostBlockSize != blockSize) { if (hostBlockEnds.size() >= 65536) hostBlockEnds.pop_front(); hostBlockEnds.push_back(st.blocksRendered); } }
    std::vector<std::shared_ptr<Plan>> retiredPlans;   // replaced plans whose launches may still be in flight; released by freeDeferred()

    // host-buffer launch sets (processBlocksHost): copy stream, pinned + device staging halves, hand-over events
    hipStream_t ioStream = nullptr;
    float* hStageOut[2] = {nullptr, nullptr}; float* hStageIn[2] = {nullptr, nullptr};
    float* dStageOut[2] = {nullptr, nullptr}; float* dStageIn[2] = {nullptr, nullptr};
    size_t stageOutFloats = 0, stageInFloats = 0;
    hipEvent_t evIn[2] = {nullptr, nullptr}, evRendered[2] = {nullptr, nullptr}, evOut[2] = {nullptr, nullptr};
    int ensureHostStaging(size_t outFloats, size_t inFloats);
    // enqueue `numBlocks` blocks on `stream` (no synchronise at the end): the body of processBlocks
    int enqueueBlocks(const float* inDev, size_t nIn, float* outDev, size_t nOut, size_t numBlocks, int64_t sampleTime);

    std::shared_ptr<Plan> current, pending;
    uint32_t maxLdsConfigured = 0;
    // ---- elemhip_process ends by spinning on a word the block's epilogue kernel publishes to mapped host memory behind the output block,
    // not by synchronising the stream (option "sync_poll", default 1; island.inc publish_done) ----
    bool syncPoll = true;
    uint32_t* hDone = nullptr; uint32_t* dDone = nullptr;     // the word: host / device address
    uint32_t doneSeq = 0;
    uint32_t* armFlag = nullptr; uint32_t armValue = 0;        // what the epilogue launch of the call being enqueued publishes (null: nothing)
    bool flagArmed = false;                                    // ... and whether an epilogue kernel took it
    uint32_t unsyncedCalls = 0;                                // calls since the stream was last synchronised for real
    uint64_t syncPolls = 0, syncPollFallbacks = 0;
    // ---- option "resident": elemhip_process through a kernel that stays on the GPU between calls (resident.hip) ----
    bool residentOpt = false;
    uint32_t residentIdleUs = 2000;        // option "resident_idle_us": the kernel leaves by itself after this long without a block
    uint32_t residentAfter = 3;            // ... and is launched once this many plain blocks in a row went through the launch path
    bool residentLive = false;             // the kernel is (or may still be) on `stream`
    uint32_t residentSeq = 0, residentStreak = 0;
    const Plan* residentPlan = nullptr; size_t residentNIn = 0, residentNOut = 0;
    ResidentCtl* hResident = nullptr; ResidentCtl* dResidentCtl = nullptr;   // mapped host memory, host / device address
    float* hResIn = nullptr; float* hResOut = nullptr; const float* dResIn = nullptr; float* dResOut = nullptr; size_t resInFloats = 0, resOutFloats = 0;
    unsigned long long* dResidentSync = nullptr;
    uint32_t residentLdsConfigured = 0;
    uint64_t residentTicksBody = 0, residentTicksEpilogue = 0;   // ELEMHIP_RESIDENT_TRACE: device-side phase times of this residency
    bool residentEligible(const Plan& p, size_t nIn, size_t nOut) const;
    int residentStart(const Plan& p, size_t nIn, size_t nOut);
    void residentStop();                   // (`mu` held) ask the kernel to leave and wait until it has; a no-op when it is not there
    // renders one block through the live kernel: kOk, an error, or kResidentGone — it left (idle) before it saw the block: launch path
    int residentBlock(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n);
    static constexpr int kResidentGone = -1000;
    bool useGraph = true;
    int  graphBlocks = 8;
    bool hostOutDirect = true;             // process(): the epilogue kernel writes into the mapped pinned output block
    uint32_t convMinP = 0xFFFFFFFFu, convMaxP = 0;   // fewest / most partitions of any impulse response set so far
    bool convLong = true;                  // option "conv_long": launch sets of a multiple of 8 blocks render IRs of >= 32 partitions with 4096-sample partitions (conv_long.inc)
    bool deviceClockBehind = false;        // direct-I/O convolver sets moved hGlobals.sampleTime on without patching the device's copy (flushPending catches it up)
    bool nextSetDirect = false;            // ... unless the launch being prepared is another such set
    uint32_t convUid = 0;                  // conv::H_UID of the newest convolver state
    uint64_t convScratchKey = ~0ull;       // (batch_blocks, long history rows) the scratch headers were zeroed for
    std::set<int32_t> convStaleNodes;      // convolve nodes last rendered by a long-partition set: their `overlap` is made on demand (fixConvOverlaps), per node
    void fixConvOverlaps(const Plan& p);   // ... before the next 512-partition evaluation (block-at-a-time launches, sets that are no multiple of 8 blocks)
    uint32_t convLongMacMode = 0;          // option "conv_long_mac_lds": 0 the register kernel (ships), 1 the LDS-tiled kernel (r06: 64 bins x 32 chunks; as fast, a third of the L2 traffic), 2 runs of 32, 3 zigzag
    bool convDirectIo = true;              // option "conv_direct_io": a plan of long-partition convolvers only reads the caller's input / writes the caller's output in place
    // the launch set being enqueued (enqueueBlocks -> enqueueBatch -> launchConvolveBatch): where its convolvers read / write directly
    const float* setInDirect = nullptr; float* setOutDirect = nullptr; uint32_t setNumIn = 0, setNumOut = 0;
    uint64_t convDirectSets = 0;
    void chooseConvDirectIo(const Plan& p, size_t nIn, size_t nOut, uint32_t batch, bool haveIn, bool& dIn, bool& dOut);
    uint64_t convLongSets = 0;             // launch sets in which some node took the long-partition kernels (describe_plan)
    uint32_t convMaxQp = 0;                // most long-partition tap rows of any impulse response set so far (sizes the scratch)
    int convMfma = 1;                      // conv.hip elemhip_convolve_batch_mac: 1 v_mfma_f32_4x4x1_16B_f32 Toeplitz tiles, 0 v_pk_fma_f32 (r03)
    bool skipIdleLaunches = true;          // option "skip_idle_launches": launches of a level whose islands all belong to roots that do not run are left out
    bool fuseEpilogue = false;             // option "fuse_epilogue": elemhip_process' launch set of one ends in the last level's kernel (no epilogue launch);
                                           // measured break-even (the ticket's release / acquire costs what the dependent launch did): off
    bool specBlockGraph = false;           // option "spec_block_graph": replay elemhip_process' launch set of one from a captured hipGraph
    bool specBlocks = true;                // process(): whole blocks of a settled, fully compiled sequence use the specialised kernels
    int  batchBlocks = 64;                 // blocks per multi-block launch in processBlocks (1 = per-block launches)
    int  specWavesPerEu = 0;               // option "spec_waves_per_eu": amdgpu_waves_per_eu of the specialised kernels (0: the compiler's choice; 4: two workgroups per CU)
    int  pipelineCopies = 6;               // blocks a stateful island keeps in flight inside a multi-block launch
    bool convAligned = true;               // every process call so far rendered whole 512-frame blocks (conv.hip batch path)
    float* dConvScratch = nullptr; size_t convScratchFloats = 0;
    uint32_t fuseSvfCoef = 2;              // plan.cpp: svf coefficient pre-pass inside the scan: 0 never, 1 always, 2 in lane-packed islands
    uint32_t soloWaves = 0;                // plan.cpp: heaviest recurrence waves that get no SIMD mate
    uint32_t mixerSplit = 2;               // workgroups a mixer island is cut into (plan.cpp; each renders blockSize / split frames on 8 / split waves)
    bool streamRing = true;                // stream buffers of the specialised kernels live in a ring of `copies` slices (0: one slice per block; measurement)
    int  packIslands = 0;                  // option "pack_islands": same-shape islands merged into one workgroup (0 auto: when a launch level has more
                                           // stateful islands than the device has CUs; 1 never; K: K per island)
    bool packRoots = false;                // option "pack_roots": lane-packing may merge islands of different ACTIVE roots (render jobs with a root each)
    int  packMax = 2, cuCount = 256;       // auto mode: at most packMax per island (measured on C2: 2 per island pays, 3 leaves two buffer sets and loses); CUs of the device
    bool chainLdsOut = false;              // option "chain_lds_out" (experiment): streamed recurrences write their block to LDS, only their operands come through the arena
    bool mergePhases = true;               // option "merge_phases": constant-frequency phasors and oscillator phases of a stage share one recurrence task (OP_PHASE)
    uint32_t statelessRows = 64;           // gridDim.y of a multi-block launch: blocks that stateless islands render side by side
    int  timeBatch = 1;
    int  specialize = 1;                   // 0: interpreter kernels only; 1: specialised kernels compiled in the background and used
                                           // once ready; 2: commit() waits for them (deterministic: tests, benchmarks)
    std::vector<hipStream_t> auxStreams;   // side streams for the independent launches of one level (launchLevelBatch)
    std::vector<hipEvent_t> auxDone;
    hipEvent_t forkEvent = nullptr;
    uint32_t profileEvery = 1, profSetCounter = 0;   // option "profile_launches" = N: events around every N-th launch set
    bool profileLaunches = false;
    std::vector<double> profMs;            // per level + epilogue, summed over the launch sets profiled so far
    uint64_t profSets = 0, profBlocks = 0;
    std::vector<hipEvent_t> profEvents;    // event pool of the current processBlocks call
    size_t profUsed = 0;
    std::vector<uint32_t> profSlots;       // per used pair: which profMs slot it feeds
    hipEvent_t profEvent();
    void profCollect();
    uint32_t lastTimeBatch = 1;            // blocks per launch the last timeLaunches actually used                    // timeLaunches: blocks per timed launch

    uint32_t allocRec();
    uint32_t channelRec(Node& n, uint32_t ch);           // record of output channel `ch` of a (multi-output) node
    void writeRec(uint32_t rec, uint32_t dword, uint32_t value);
    void writeTableChannel(Node& n, uint32_t ch, uint32_t rec);
    void writeChannelBuffer(Node& n, uint32_t ch, uint32_t rec);
    int  ensureResourceChannelOnDevice(const ResourcePtr& r, uint32_t ch, const void** ptr, uint32_t* len);
    void writeParam(Node& n, uint32_t dword, uint32_t value);
    void writeParamF(Node& n, uint32_t dword, float value) { uint32_t u; memcpy(&u, &value, 4); writeParam(n, dword, u); }
    void writeParamPtr(Node& n, uint32_t dword, const void* p);
    int  allocRing(Node& n, size_t floats);
    int  ensureResourceOnDevice(const ResourcePtr& r);
    int  setConvolverIr(Node& n, const ResourcePtr& res);
    ResourcePtr tapResource(const std::string& name);
    void rootUpdateStep(Node& n);
    int  flushPending();                   // fresh records + patches -> device (stream-ordered)
    void freeDeferred();
    void dropGraphs();                     // the current plan's captured graphs (a pointer or launch geometry baked into them changed)
    int  ensureHbm(size_t buffers);
    size_t arenaBuffers(const Plan& p, size_t blocks) const;
    size_t maxSetBlocks(const Plan& p) const;
    int  ensureOutRing(size_t floats);
    int  swapInPending();
    void enqueueBlock(const Plan& p, float* outRing = nullptr);
    int  renderHostNodes(const Plan& p, size_t level);   // call-out nodes of one launch level (synchronises the stream)
    void enqueueBatch(const Plan& p, uint32_t batch, float* outRing = nullptr);
    // one block of a fully compiled plan whose root fades are still running: the specialised level launches of a set of one, then
    // the PER-BLOCK epilogue (fades advance, taps promoted) — the two blocks after every commit of a live graph
    void enqueueSpecBlock(const Plan& p, float* outRing = nullptr);
    bool specBlockOk(const Plan& p) const;
    void launchConvolveBatch(const Plan& p, size_t l, uint32_t batch, uint32_t arenaFloats);
    // specialised kernels when ready, else the interpreter. `epiOut` non-null: if the level is one specialised launch, let its last
    // workgroup run the epilogue into `epiOut` (island_spec.inc spec_epilogue_tail); returns whether it will
    bool launchLevelBatch(const Plan& p, size_t level, uint32_t batch, uint32_t arenaFloats, float* epiOut = nullptr);
    bool batchEligible(const Plan& p, size_t nOut, bool oneBlock = false) const;
    bool specReady(const Plan& p) const;
    // background mode: the one-off shapes of the current plan are queued for compilation (behind everything else) once the plan
    // has rendered `lonelyBlocks` blocks and has been current for `lonelyMs` — a plan a live graph replaces 30 ms later never
    // gets there, a static patch does within its first second (plan.cpp "deferred")
    void promoteDeferredShapes();
    int lonelyBlocks = 64, lonelyMs = 30;  // options "spec_lonely_blocks", "spec_lonely_ms"
    int maxShapeLaunches = 6;              // option "max_shape_launches": specialised launches per level (the shapes with the most islands); the rest -> one interpreter launch
    uint64_t islandBlocksSpec = 0, islandBlocksInterp = 0;   // island x block units rendered by specialised / interpreter kernels (describe_plan)
    bool anyRootRuns(const std::vector<int32_t>& rootIds, size_t nOut) const;   // host mirror of spec_root_running (Core.h:28-31, GraphRenderSequence.h:214-219)
    void mirrorRootFades(const Plan& p, uint32_t n, uint32_t nOut, uint32_t nIn);
    int  setGlobalsFor(size_t nIn, size_t nOut, size_t n, int64_t sampleTime);
    void setInRing(const float* ring, uint32_t blocks);
    std::shared_ptr<Plan> buildPlan(std::unique_lock<std::mutex>& renderLock);
    int debugBuildDelayMs = 0;
};

// The specialised-kernel variant of one island's program (plan.cpp builds it, codegen.cpp turns it into text).
struct SpecProgram {
    std::vector<Member> members;           // same indexing as the interpreter's tables
    std::vector<uint32_t> operands;
    std::vector<uint8_t> gdirect;          // per task: a recurrence that streams its block straight to the arena
    std::vector<uint32_t> phaseOp;         // per member: waveform tasks of streamed oscillators read the phase from here
    std::vector<uint32_t> hbmTab;          // absolute arena indices named by the variant (appended to the island's blob)
};

// plan.cpp
struct Plan {
    std::vector<Island> islands;
    std::vector<uint32_t> levelIslands;
    std::vector<uint32_t> levelOffsets;    // numLevels + 1
    std::vector<uint32_t> levelLdsBytes;
    std::vector<uint32_t> prog;            // program blobs of the islands THIS build scheduled (staging for the heap upload)
    std::shared_ptr<ProgHeap> progHeap;    // Island::progBegin is an offset into it
    std::vector<std::shared_ptr<IslandProgram>> islandProg;   // per island (null: convolve / call-out islands)
    size_t progDwordsTotal = 0;            // all islands' programs (sizes the next heap)
    size_t heapOverflowDwords = 0;         // build failed: the heap lacked room for this many dwords
    uint32_t numTasks = 0, numMembers = 0, numOperands = 0;
    std::vector<RootEntry> roots;
    std::vector<TapEntry> taps;
    int packedRootChannels = 0;                              // > 0: islands of different roots share workgroups; a call must ask for at least this many outputs
    bool tapsInSets = true;                                  // every tapIn / tapOut pair sits in one island: launch sets may render this plan (plan.cpp)
    std::vector<std::pair<int32_t, int32_t>> tapPairs;       // (tapIn node id, id of the in-island tapOut whose private buffer it reads inside a launch set, or 0)
    std::vector<std::pair<int32_t, int32_t>> eventNodes;   // (node id, owning root id) of meter / snapshot nodes, render order
    std::vector<ConvDesc> convs;           // convolve nodes (conv.hip)
    std::vector<int32_t> convNodeIds;      // ... and their node ids (same order)
    std::vector<uint32_t> convWork;        // conv workgroups, level-major
    std::vector<uint32_t> convLevelOffsets; // numLevels + 1
    std::vector<int32_t> rootIds;          // same order as `roots`
    std::vector<uint32_t> islandLevel;     // launch level of each island
    std::vector<int32_t> islandRoot;       // node id of the root whose sequence owns the island (packed islands: of the head island)
    std::vector<std::vector<int32_t>> restRoots;   // per level: the roots that own the islands of the interpreter launch (launchLevelBatch skips idle launches)
    mutable std::vector<int32_t> nodeIds;  // every node the render sequence references (gc); sorted by the first holdsNode (gc is rare, a build is not)
    mutable bool nodeIdsSorted = false;
    bool holdsNode(int32_t id) const {
        if (!nodeIdsSorted) { std::sort(nodeIds.begin(), nodeIds.end()); nodeIdsSorted = true; }
        return std::binary_search(nodeIds.begin(), nodeIds.end(), id);
    }
    std::vector<int32_t> mcCaptureIds;     // mc.capture nodes of the sequence (their rings are re-made when it is pushed: Engine::commit)
    double buildUs[6] = {0, 0, 0, 0, 0, 0}; // where the build went: render order, islands, island programs, levels + roots, shapes + tables, upload
    uint32_t numHbmBuffers = kMaxHostIn;       // per block of a launch set: host inputs + exports
    uint32_t numStreamBuffers = 0;             // per slice of the stream ring (specialised kernels; device.h kOpStream)
    uint32_t maxCopies = 1;                    // most buffer sets an island of the plan keeps in flight = slices of the stream ring
    uint32_t packK = 1;                        // islands merged per workgroup by the lane-packing step (1 = none were)
    uint32_t maxLdsBytes = 0;
    // device copies
    DevBuf dev;                            // one allocation holding all tables (from / back to `pool`)
    std::shared_ptr<TablePool> pool;
    PlanView view{};
    // host call-out nodes (OP_HOST): rendered on the CPU after the kernels of their launch level
    struct HostDesc {
        int32_t nodeId; int32_t rootId;
        struct In { int kind; uint32_t idx; float value; };   // 0 zero, 1 HBM arena buffer, 2 constant, 3 host input channel idx
        std::vector<In> inputs;
        bool leaf = false;         // no inlets: the node reads the host input channels (GraphRenderSequence.h:107-141)
        uint32_t outHbm; uint32_t level; bool active;
    };
    std::vector<HostDesc> hosts;
    // specialised kernels (jit.cpp): islands of one launch level with the same generated program text share a kernel
    struct SpecShape {
        std::shared_ptr<SpecEntry> entry;
        uint32_t level = 0;
        uint32_t listBegin = 0, count = 0;     // its workgroups: specLists[listBegin, listBegin + count) (island | split part << 24)
        bool stateless = false;                // no block pipeline: the launch spreads the blocks of a set over gridDim.y
        bool optional = false;                 // a one-island shape of a background-mode plan: while it compiles its island renders through
                                               // the interpreter kernel and the plan still counts as ready (Engine::specReady)
        bool deferred = false;                 // ... and its compile has not even been queued: Engine::promoteDeferredShapes does that once
                                               // the plan has rendered for a while
        std::vector<int32_t> roots;            // the roots that own its islands: when none of them runs the launch is skipped
    };
    std::vector<SpecShape> shapes;
    uint32_t deferredShapes = 0;               // shapes whose compile waits for this plan to prove that it stays (0 again once promoted)
    uint64_t blocksAtAdoption = 0;             // Engine stats' blocksRendered when this plan became the current one
    std::chrono::steady_clock::time_point adopted{};   // when it became the current plan
    std::vector<uint32_t> specLists;           // island indices, shape-major
    std::vector<uint32_t> restIslands;         // per level: the levelIslands entries no shape covers (interpreter launch)
    std::vector<uint32_t> restOffsets;         // numLevels + 1
    std::vector<std::shared_ptr<SpecText>> specText;   // per island: generated text (null = interpreter only), consumed by buildPlan
    const uint32_t* dSpecLists = nullptr;      // device copies (inside `dev`)
    const uint32_t* dRestIslands = nullptr;
    // captured launch sequence for multi-block offline rendering
    hipGraphExec_t graphExec = nullptr;
    int graphBlocks = 0;
    uint32_t blockChunks = 0;               // block-at-a-time chunks rendered before the capture
    // elemhip_process of a settled, fully compiled sequence: the launch set of ONE (levels + batch epilogue) as a captured graph
    hipGraphExec_t specGraphExec = nullptr;
    float* specGraphOut = nullptr;          // the output pointer baked into it
    uint32_t specGraphNumOut = 0;           // ... and the output count it was captured for
    uint32_t specGraphLaunches = 0;         // specialised launches it replays (stats)
    ~Plan();
};

} // namespace elemhip
