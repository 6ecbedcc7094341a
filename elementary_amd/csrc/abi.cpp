// abi.cpp — the extern "C" surface declared in include/elemhip.h.
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/elemhip.h"
#include "engine.h"
#include "launch.h"

using elemhip::Engine;
using elemhip::Value;

struct elemhip_s {
    Engine engine;
    // typed path: root ids staged by elemhip_activate_roots until the next elemhip_commit of THIS handle (any thread)
    std::mutex stagedMu;
    std::vector<int32_t> stagedRoots;
    bool haveStaged = false;
    elemhip_s(double sr, int bs, int dev) : engine(sr, bs, dev) {}
};

static std::atomic<int> g_lastCreateError{0};

extern "C" {

elemhip_t* elemhip_create(double sampleRate, int blockSize, int deviceOrdinal) {
    elemhip_s* h = new (std::nothrow) elemhip_s(sampleRate, blockSize, deviceOrdinal);
    if (!h) { g_lastCreateError = elemhip::kHipError; return nullptr; }
    if (h->engine.initError()) { g_lastCreateError = h->engine.initError(); delete h; return nullptr; }
    g_lastCreateError = 0;
    return h;
}

void elemhip_destroy(elemhip_t* h) { delete h; }
int elemhip_last_create_error(void) { return g_lastCreateError.load(); }

int elemhip_apply_instructions_json(elemhip_t* h, const char* json, size_t len) {
    if (!h || !json) return elemhip::kInvalidInstructionFormat;
    Value v;
    elemhip::JsonParser parser(json, len);
    if (!parser.parse(v)) return elemhip::kJsonParseError;
    return h->engine.apply(v);
}

static int applyOne(elemhip_t* h, Value&& instr) {
    Value batch; batch.type = Value::Array;
    batch.arr.push_back(std::move(instr));
    return h->engine.apply(batch);
}

int elemhip_create_node(elemhip_t* h, int32_t id, const char* type) {
    if (!h || !type) return elemhip::kInvalidInstructionFormat;
    Value i; i.type = Value::Array;
    i.arr = {Value::number(0), Value::number(id), Value::string(type)};
    return applyOne(h, std::move(i));
}

int elemhip_append_child(elemhip_t* h, int32_t parent, int32_t child, int32_t ch) {
    if (!h) return elemhip::kInvalidInstructionFormat;
    Value i; i.type = Value::Array;
    i.arr = {Value::number(2), Value::number(parent), Value::number(child), Value::number(ch)};
    return applyOne(h, std::move(i));
}

int elemhip_set_property_json(elemhip_t* h, int32_t id, const char* key, const char* json, size_t len) {
    if (!h || !key || !json) return elemhip::kInvalidInstructionFormat;
    // a bare scalar is valid here, so wrap it to reuse the array parser
    std::string wrapped = "[" + std::string(json, len) + "]";
    Value v;
    elemhip::JsonParser parser(wrapped.data(), wrapped.size());
    if (!parser.parse(v) || v.arr.size() != 1) return elemhip::kJsonParseError;
    Value i; i.type = Value::Array;
    i.arr = {Value::number(3), Value::number(id), Value::string(key), v.arr[0]};
    return applyOne(h, std::move(i));
}

// ACTIVATE_ROOTS and COMMIT_UPDATES only rebuild when they share a batch (Runtime.h:172,199-205),
// so the typed path keeps the pair together: activate validates and stages the ids in the handle,
// commit sends [4,...],[5].
int elemhip_activate_roots(elemhip_t* h, const int32_t* ids, size_t n) {
    if (!h || (!ids && n)) return elemhip::kInvalidInstructionFormat;
    for (size_t i = 0; i < n; ++i) if (!h->engine.hasNode(ids[i])) return elemhip::kNodeNotFound;   // Runtime.h:386-390
    std::lock_guard<std::mutex> lock(h->stagedMu);
    h->stagedRoots.assign(ids, ids + n);
    h->haveStaged = true;
    return elemhip::kOk;
}

int elemhip_commit(elemhip_t* h) {
    if (!h) return elemhip::kInvalidInstructionFormat;
    Value batch; batch.type = Value::Array;
    {
        std::lock_guard<std::mutex> lock(h->stagedMu);
        if (h->haveStaged) {
            Value roots; roots.type = Value::Array;
            for (int32_t id : h->stagedRoots) roots.arr.push_back(Value::number(id));
            Value a; a.type = Value::Array;
            a.arr = {Value::number(4), roots};
            batch.arr.push_back(std::move(a));
            h->haveStaged = false;
        }
    }
    Value c; c.type = Value::Array; c.arr = {Value::number(5)};
    batch.arr.push_back(std::move(c));
    return h->engine.apply(batch);
}

int elemhip_process(elemhip_t* h, const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n, int64_t st) {
    if (!h) return elemhip::kInvalidInstructionFormat;
    return h->engine.process(in, nIn, out, nOut, n, st);
}

int elemhip_process_blocks(elemhip_t* h, const float* inDev, size_t nIn, float* outDev, size_t nOut, size_t numBlocks, int64_t st) {
    if (!h) return elemhip::kInvalidInstructionFormat;
    return h->engine.processBlocks(inDev, nIn, outDev, nOut, numBlocks, st);
}

int elemhip_process_blocks_host(elemhip_t* h, const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t numFrames, int64_t st) {
    if (!h) return elemhip::kInvalidInstructionFormat;
    return h->engine.processBlocksHost(in, nIn, out, nOut, numFrames, st);
}

int elemhip_add_shared_resource(elemhip_t* h, const char* name, const float* const* ch, size_t nCh, size_t nSamples) {
    if (!h || !name) return 0;
    return h->engine.addSharedResource(name, ch, nCh, nSamples) ? 1 : 0;
}

void elemhip_prune_shared_resources(elemhip_t* h) { if (h) h->engine.pruneSharedResources(); }
size_t elemhip_gc(elemhip_t* h, int32_t* out, size_t cap) { return h ? h->engine.gc(out, cap) : 0; }
size_t elemhip_last_gc(elemhip_t* h, int32_t* out, size_t cap) { return h ? h->engine.lastGc(out, cap) : 0; }
void elemhip_reset(elemhip_t* h) { if (h) h->engine.reset(); }
const char* elemhip_describe(int code) { return elemhip::describe(code); }

int elemhip_register_node_type(elemhip_t* h, const char* type, const elemhip_node_type* vt) {
    if (!h || !type || !vt) return elemhip::kInvalidInstructionFormat;
    elemhip::HostVTable t;
    t.create = vt->create; t.destroy = vt->destroy; t.setProperty = vt->set_property; t.process = vt->process; t.reset = vt->reset; t.user = vt->user;
    return h->engine.registerNodeType(type, t);
}

static size_t copyOut(const std::string& s, char* buf, size_t cap) {
    if (buf && cap) { const size_t n = s.size() < cap - 1 ? s.size() : cap - 1; std::memcpy(buf, s.data(), n); buf[n] = 0; }
    return s.size() + 1;
}
size_t elemhip_snapshot_json(elemhip_t* h, char* buf, size_t cap) { return h ? copyOut(h->engine.snapshotJson(), buf, cap) : 0; }
size_t elemhip_shared_resource_keys_json(elemhip_t* h, char* buf, size_t cap) { return h ? copyOut(h->engine.sharedResourceKeysJson(), buf, cap) : 0; }

int elemhip_process_queued_events(elemhip_t* h, elemhip_event_cb cb, void* user) {
    if (!h) return elemhip::kInvalidInstructionFormat;
    return h->engine.processQueuedEvents(cb, user);
}

int elemhip_process_queued_events_blockwise(elemhip_t* h, elemhip_event_cb cb, void* user) {
    if (!h) return elemhip::kInvalidInstructionFormat;
    return h->engine.processQueuedEvents(cb, user, true);
}
uint32_t elemhip_event_window_blocks(elemhip_t* h) { return h ? h->engine.eventWindowBlocks() : 0u; }

int elemhip_get_stats(elemhip_t* h, elemhip_stats* out) {
    if (!h || !out) return elemhip::kInvalidInstructionFormat;
    const elemhip::Stats& s = h->engine.stats();
    out->blocks_rendered = s.blocksRendered; out->plans_built = s.plansBuilt; out->last_plan_build_ms = s.lastPlanBuildMs;
    out->num_islands = s.numIslands; out->num_levels = s.numLevels; out->num_tasks = s.numTasks;
    out->num_nodes_in_plan = s.numNodesInPlan; out->max_lds_bytes = s.maxLdsBytes; out->num_hbm_buffers = s.numHbmBuffers;
    out->graph_replays = s.graphReplays; out->graph_captures = s.graphCaptures;
    out->batch_launches = s.batchLaunches;
    out->spec_launches = s.specLaunches; out->spec_shapes = s.specShapes; out->spec_islands = s.specIslands;
    out->last_jit_wait_ms = s.lastJitWaitMs; out->last_graph_capture_ms = s.lastGraphCaptureMs;
    out->resident_launches = s.residentLaunches; out->resident_blocks = s.residentBlocks;
    return elemhip::kOk;
}

int elemhip_time_launches(elemhip_t* h, size_t nOut, size_t numBlocks, float* msOut, size_t cap) {
    if (!h || !msOut) return -elemhip::kInvalidInstructionFormat;
    return h->engine.timeLaunches(nOut, numBlocks, msOut, cap);
}

int elemhip_get_launch_profile(elemhip_t* h, double* msOut, size_t cap, uint64_t* launchSets, uint64_t* blocks) {
    if (!h || !msOut) return -elemhip::kInvalidInstructionFormat;
    return h->engine.launchProfile(msOut, cap, launchSets, blocks);
}

int elemhip_trace_level(elemhip_t* h, size_t nOut, uint32_t level, unsigned long long* out, size_t cap) {
    if (!h || !out) return elemhip::kInvalidInstructionFormat;
    return h->engine.traceLevel(nOut, level, out, cap);
}

// Debug/test hook: JSON description of the current render plan. Returns bytes needed.
size_t elemhip_describe_plan(elemhip_t* h, char* buf, size_t cap) {
    if (!h) return 0;
    const std::string s = h->engine.describePlan();
    if (buf && cap) { const size_t n = s.size() < cap - 1 ? s.size() : cap - 1; std::memcpy(buf, s.data(), n); buf[n] = 0; }
    return s.size() + 1;
}

// Debug/test hook: the k-th specialised island shape of the newest plan. Returns the number of shapes (or -1);
// *state = 0 compiling, 1 ready, -1 failed; the program text / compiler log are copied when buffers are given.
int elemhip_spec_info(elemhip_t* h, size_t k, char* src, size_t srcCap, char* log, size_t logCap, int* state, uint32_t* islands) {
    if (!h) return -1;
    std::string s, l;
    const int n = h->engine.specInfo(k, src ? &s : nullptr, log ? &l : nullptr, state, islands);
    auto copy = [](const std::string& from, char* to, size_t cap) { if (to && cap) { const size_t m = from.size() < cap - 1 ? from.size() : cap - 1; std::memcpy(to, from.data(), m); to[m] = 0; } };
    copy(s, src, srcCap); copy(l, log, logCap);
    return n;
}

int elemhip_set_stream(elemhip_t* h, void* stream) {
    if (!h) return elemhip::kInvalidInstructionFormat;
    h->engine.setStream(static_cast<hipStream_t>(stream));
    return elemhip::kOk;
}

int elemhip_sum_buses(int deviceOrdinal, void* stream, float* dst, const float* const* partials, size_t nPartials, size_t nFloats) {
    if (!dst || !partials || nPartials == 0 || nPartials > 64) return elemhip::kInvalidInstructionFormat;
    if (hipSetDevice(deviceOrdinal) != hipSuccess) return elemhip::kHipError;
    if (elemhip::launch_bus_sum(static_cast<hipStream_t>(stream), dst, partials, (uint32_t)nPartials, nFloats) != hipSuccess) return elemhip::kHipError;
    return elemhip::kOk;
}

int elemhip_set_option(elemhip_t* h, const char* key, double value) {
    if (!h || !key) return elemhip::kInvalidInstructionFormat;
    return h->engine.setOption(key, value);
}

} // extern "C"
