/* elemhip.h — C-ABI of the MI355X block-render engine (libelemhip.so).
 *
 * Drop-in boundary for ONE path of elemaudio/elementary: elem::Runtime<float>::process() and the
 * graph-mutation calls that feed it.  Every entry point replaces one member of
 * `elem::Runtime<float>` (runtime/elem/Runtime.h:39-153); the reference line it stands in for is
 * cited next to it.  Plain pointers and sizes only; the caller owns every pointer it passes and
 * the engine copies whatever it keeps.  Integer return codes 0..8 are the reference's
 * `elem::ReturnCode` (runtime/elem/Types.h:51-86); codes >= 100 are HIP-side failures.
 *
 * Threading contract: one mutator thread (apply/add/gc) + one render thread (process*) per
 * handle; mutations become visible atomically at the next process call (reference: SPSC queue of
 * render sequences, Runtime.h:133,204,277-285).  Two locks inside the engine: a control lock
 * serialises the mutator-side calls and owns the node table; a render lock guards what process*
 * touches.  process* take the render lock only, and a commit releases it while it plans the new
 * render sequence, so a render call never waits for a plan build; the mutator can
 * wait for a render call in flight (at most one launch set), never the other way round.
 *
 * The engine has NO CPU fallback: every block is rendered by HIP kernels for gfx950 — the
 * ahead-of-time island kernels (elementary_amd/csrc/kernels.hip, kernels_rt.hip: island.inc +
 * island_ops.inc), the convolve kernels (conv.hip) and the per-island-shape kernels compiled at run
 * time from island_spec.inc + generated text (codegen.cpp, jit.cpp); elemhip_create fails (NULL)
 * when no gfx950 device is usable.
 *
 * Block size: the reference sizes its buffers to any blockSize (Runtime.h:44); this engine keeps a
 * block's buffers in LDS slots of at most 512 frames. blockSize <= 512: as given. Above 512 (up to 32768): accepted when the block
 * splits into k EQUAL slices of 64 .. 512 frames (1024 = 2 x 512, 700 = 2 x 350, 1023 = 3 x 341; the smallest such k is taken):
 * elemhip_process / elemhip_process_blocks_host render such a block slice by slice — the same samples for every node that works at
 * the sample rate; tapIn / tapOut, whose delay IS the host's block (Feedback.h:29-31, 88-109), keep tap buffers of the HOST's block
 * size and every slice reads / promotes its own stretch of them (r04 refused such graphs with code 104);
 * `meter` reports the last slice of a block, and the device-resident elemhip_process_blocks (whose layout is in blocks) answers
 * 102. A size above 512 that no such k divides (a prime: 521, 1031): slices of 512 frames and a shorter last one. Above 32768:
 * elemhip_create fails (code 102). Nodes that size themselves by the block (`delay` / `sdelay` without a `size`, tap buffers) use
 * the HOST's block size, as in the reference (Delays.h:56, 183; Feedback.h:29-31).
 */
#ifndef ELEMHIP_H
#define ELEMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct elemhip_s elemhip_t;

typedef struct elemhip_stats {
    uint64_t blocks_rendered;
    uint64_t plans_built;
    double   last_plan_build_ms;
    uint32_t num_islands, num_levels, num_tasks, num_nodes_in_plan, max_lds_bytes, num_hbm_buffers;
    uint64_t graph_replays, graph_captures;
    uint64_t batch_launches;      /* multi-block launch groups issued by elemhip_process_blocks */
    uint64_t spec_launches;       /* launches of run-time specialised island kernels */
    uint32_t spec_shapes, spec_islands;   /* distinct specialised island shapes / islands they cover in the current plan */
    double   last_jit_wait_ms;    /* time the last commit waited for kernel compilation (option "specialize" = 2) */
    double   last_graph_capture_ms; /* hipGraph capture + instantiate of the current plan's per-block launch sequence */
    uint64_t resident_launches, resident_blocks;   /* option "resident": launches of the resident kernel / blocks it rendered */
} elemhip_stats;

/* Runtime(double sampleRate, int blockSize)                      runtime/elem/Runtime.h:44,157-166
 * blockSize is the MAX frames per process call (see "Block size" above). Returns NULL on failure; the reason is
 * available from elemhip_last_create_error(). */
elemhip_t* elemhip_create(double sampleRate, int blockSize, int deviceOrdinal);
void       elemhip_destroy(elemhip_t*);
int        elemhip_last_create_error(void);

/* int applyInstructions(js::Array const& batch)                  Runtime.h:48,170-218
 * `utf8_json` is the same JSON text the cli host hands to elem::js::parseJSON
 * (cli/Benchmark.cpp:40-43): [[0,id,"type"],[2,parent,child,outCh],[3,id,"key",value],[4,[roots]],[5]] */
int elemhip_apply_instructions_json(elemhip_t*, const char* utf8_json, size_t len);

/* Typed fast path for the same five instructions                 Runtime.h:123-126,293-433 */
int elemhip_create_node(elemhip_t*, int32_t id, const char* type);
int elemhip_append_child(elemhip_t*, int32_t parent, int32_t child, int32_t childOutputChannel);
int elemhip_set_property_json(elemhip_t*, int32_t id, const char* key, const char* json_value, size_t len);
int elemhip_activate_roots(elemhip_t*, const int32_t* ids, size_t n);
int elemhip_commit(elemhip_t*);

/* void process(const F** in, size_t nIn, F** out, size_t nOut, size_t numSamples, void* userData)
 *                                                                Runtime.h:51-57,274-290
 * Planar host buffers; outputs are overwritten. `sampleTime` is the one thing the reference's
 * hosts pass as userData (wasm/Main.cpp:206-215, consumed by wasm/SampleTime.h, wasm/Metro.h). */
int elemhip_process(elemhip_t*, const float* const* in, size_t nIn, float* const* out, size_t nOut,
                    size_t numSamples, int64_t sampleTime);

/* Offline rendering (js/packages/offline-renderer/index.ts:87-133 block loop): `numBlocks`
 * consecutive full blocks in one call, buffers resident in HBM.
 *   in_dev  : device pointer [numBlocks][nIn][blockSize] or NULL when nIn == 0
 *   out_dev : device pointer [numBlocks][nOut][blockSize], or NULL to render without copying out */
int elemhip_process_blocks(elemhip_t*, const float* in_dev, size_t nIn, float* out_dev, size_t nOut,
                           size_t numBlocks, int64_t sampleTime);

/* The same block loop over HOST buffers: what OfflineRenderer.process(inputs, outputs) does with the reference engine
 * (js/packages/offline-renderer/index.ts:87-133: ceil(numFrames / blockSize) FULL blocks, a short input tail zero-padded,
 * every block copied to the caller's arrays) and what a host gets from calling Runtime::process (Runtime.h:51-57) in a loop.
 *   in  : nIn planar channel arrays of numFrames frames each (caller-owned, pageable is fine), NULL when nIn == 0
 *   out : nOut planar channel arrays of numFrames frames each; overwritten
 * No HIP types cross the boundary. Launch sets of `batch_blocks` blocks go through pinned double buffers: the D2H of set k
 * and the H2D of set k + 1 run on a copy stream while set k + 1 renders, so the samples land in host memory at the
 * device-resident rate. `sampleTime` = sample time of the first frame (userData of the reference hosts). */
int elemhip_process_blocks_host(elemhip_t*, const float* const* in, size_t nIn, float* const* out, size_t nOut,
                                size_t numFrames, int64_t sampleTime);

/* bool addSharedResource(name, unique_ptr<SharedResource>)       Runtime.h:83,461-465 (insert-only) */
int    elemhip_add_shared_resource(elemhip_t*, const char* name, const float* const* channels, size_t nCh, size_t nSamples);
/* void pruneSharedResources()                                    Runtime.h:89,467-471 */
void   elemhip_prune_shared_resources(elemhip_t*);
/* std::set<NodeId> gc()                                          Runtime.h:76,220-272
 * Prunes every unreferenced node and returns how many; the first min(count, cap) ids (ascending) land in prunedOut.
 * When count > cap the full set of THAT pass can still be read with elemhip_last_gc (same ordering). */
size_t elemhip_gc(elemhip_t*, int32_t* prunedOut, size_t cap);
size_t elemhip_last_gc(elemhip_t*, int32_t* prunedOut, size_t cap);
/* void reset()                                                   Runtime.h:70,448-458 */
void   elemhip_reset(elemhip_t*);

/* int registerNodeType(std::string const& type, NodeFactoryFn&&)  Runtime.h:105-106,480-487 (code 4 when the name is taken)
 * A custom node type whose instances run ON THE CPU (the reference's GraphNode plug-in interface, GraphNode.h:19-96): the
 * engine renders everything upstream on the GPU, drains the stream, hands the node its input blocks on the host, calls
 * process() and uploads the output block — a call-out per node and block, slow by construction, kept so that hosts with
 * their own C++ nodes (wasm/Main.cpp:47-61 registers three this way) keep working. Plans containing such nodes render
 * block by block (no multi-block launches, no hipGraph replay).
 *   create       NodeFactoryFn(NodeId, sampleRate, blockSize)                          GraphNode.h:27
 *   set_property GraphNode::setProperty(key, js::Value) with the value as JSON text    GraphNode.h:49; return a ReturnCode
 *   process      GraphNode::process(BlockContext): planar inputs, ONE output channel   GraphNode.h:72, Types.h:91-101
 *                (`sample_time` is what the reference's hosts pass as userData, `active` = BlockContext::active)
 *   reset        GraphNode::reset                                                      GraphNode.h:86 (may be NULL) */
typedef struct elemhip_node_type {
    void* (*create)(int32_t node_id, double sample_rate, int block_size, void* user);
    void  (*destroy)(void* node, void* user);
    int   (*set_property)(void* node, const char* key, const char* json_value, void* user);
    void  (*process)(void* node, const float* const* in, size_t n_in, float* out, size_t num_samples, int64_t sample_time, int active, void* user);
    void  (*reset)(void* node, void* user);
    void* user;
} elemhip_node_type;
int elemhip_register_node_type(elemhip_t*, const char* type, const elemhip_node_type* vt);

/* js::Object snapshot()                                          Runtime.h:110,489-498
 * {"<8-digit hex node id>": {property: value, ...}, ...} as JSON text; returns the bytes needed (including the NUL). */
size_t elemhip_snapshot_json(elemhip_t*, char* buf, size_t cap);
/* SharedResourceMap::KeyViewType getSharedResourceMapKeys()      Runtime.h:94 — a JSON array of names */
size_t elemhip_shared_resource_keys_json(elemhip_t*, char* buf, size_t cap);

/* ReturnCode::describe                                           Types.h:62-85 */
const char* elemhip_describe(int code);
int  elemhip_get_stats(elemhip_t*, elemhip_stats* out);
/* Measurement hook: render `numBlocks` blocks (no host inputs) with a HIP event pair around every
 * kernel launch on the engine's stream. msOut[l] = mean ms of launch level l, msOut[levels] = the
 * epilogue kernel. Returns the number of entries written (levels + 1) or a negated error code. */
int  elemhip_time_launches(elemhip_t*, size_t nOut, size_t numBlocks, float* msOut, size_t cap);
/* Measurement hook for timed regions: after elemhip_set_option("profile_launches", 1) every multi-block launch that
 * elemhip_process_blocks issues is bracketed by a HIP event pair on the engine's stream. msOut[l] = summed ms of launch
 * level l, msOut[levels] = the epilogue kernel; *launchSets / *blocks = launch sets and blocks covered. Returns the number
 * of entries available (levels + 1). Setting the option to 1 again clears the sums. */
int  elemhip_get_launch_profile(elemhip_t*, double* msOut, size_t cap, uint64_t* launchSets, uint64_t* blocks);
/* Tracing hook: render one block while workgroup 0 of launch level `level` logs shader-clock
 * timestamps per task. out[wave*192 + 0..3] = {tasks, kernel start, prologue end, kernel end};
 * out[wave*192 + 3*(k+2) + 0..2] = {opcode | stage<<16 | flags<<24, start, end} for the wave's k-th task. */
/* void processQueuedEvents(std::function<void(std::string const&, js::Value)>&&)   runtime/elem/Runtime.h:64, 437-446
 * Non-render thread. Relays the newest readout of every `meter` / `snapshot` node (builtins/Analyzers.h) of the current
 * render sequence whose root is active: cb(type, JSON payload, user), e.g. ("meter", {"min":..,"max":..,"source":name|null}). */
typedef void (*elemhip_event_cb)(const char* type, const char* json_payload, void* user);
int  elemhip_process_queued_events(elemhip_t*, elemhip_event_cb cb, void* user);
/* The relay of a caller that renders many blocks per call and still wants what the reference's offline renderer delivers by
 * calling processQueuedEvents after EVERY block (js/packages/offline-renderer/index.ts:112-120): every block's events since the
 * last relay, in block order, nodes in render order inside a block — reconstructed from per-block readout logs the kernels keep
 * (builtins/Analyzers.h:23-62, 83-131: `meter` queues one readout per block, `snapshot` one per latch; a per-block relay hands on the
 * newest of each block). Exact while no more than elemhip_event_window_blocks() blocks pass between two relays (1024 for meter /
 * snapshot; less with a `scope` — its 8192-frame ring must not overrun inside a window — and 1 with a `capture` node, whose take
 * belongs to the block in which the gate fell). Neither relay holds up a render thread: the readouts are snapshotted in stream order
 * and fetched on a stream of their own (runtime/elem/Runtime.h:437-446 drains lock-free queues). */
int  elemhip_process_queued_events_blockwise(elemhip_t*, elemhip_event_cb cb, void* user);
uint32_t elemhip_event_window_blocks(elemhip_t*);

int  elemhip_trace_level(elemhip_t*, size_t nOut, uint32_t level, unsigned long long* out, size_t cap);
/* Debug/test hook: JSON description of the current render plan (islands, launch levels, LDS).
 * deviceOrdinal == -1 at create time gives a "dry" handle that runs all host logic (instruction
 * decode, graph mutation, plan build, gc) without a GPU; it cannot render (process returns 101). */
size_t elemhip_describe_plan(elemhip_t*, char* buf, size_t cap);
/* Debug/test hook: program text, compiler log and state (0 compiling, 1 ready, -1 failed) of the k-th specialised
 * island shape of the newest plan; returns the number of shapes or -1. */
int  elemhip_spec_info(elemhip_t*, size_t k, char* src, size_t srcCap, char* log, size_t logCap, int* state, uint32_t* islands);
/* Run the launches on a caller-owned hipStream_t (e.g. torch's current stream). */
int  elemhip_set_stream(elemhip_t*, void* hipStream);
/* Multi-GPU hosts: the path shards over independent units (voices, render jobs: SURVEY 8(e)) and its one exchange step is the sum
 * of the ranks' output buses. The reference has no counterpart (one thread, one bus: GraphRenderSequence.h:286-290 zeroes it,
 * every root adds into it). Once the ranks' buses sit on one device (hipMemcpyPeer, or ncclSend / ncclRecv — INTEGRATION.md
 * section 6) this adds them in the order given: dst = ((partials[0] + partials[1]) + partials[2]) + ... — the same bits on every
 * run, unlike a ring all-reduce. Device pointers, `nFloats` each; at most 64 partials; asynchronous on `hipStream` (NULL: the
 * null stream); no engine handle needed. */
int  elemhip_sum_buses(int deviceOrdinal, void* hipStream, float* dst, const float* const* partials, size_t nPartials, size_t nFloats);
/* Tunables (the full table with defaults: INTEGRATION.md section 5): "batch_blocks" (blocks per multi-block launch, 1 ... 1024),
 * "specialize" (0 interpreter kernels only, 1 per-island-shape kernels compiled in the background and used once ready, 2 commit
 * waits for them), "spec_blocks" / "host_out_direct" (elemhip_process through the specialised kernels / output written straight
 * into pinned host memory), "use_graph" / "graph_blocks" (per-block launch path replayed from a hipGraph), "stateless_rows",
 * "mixer_split", "pipeline_copies", "merge_phases", "stream_ring", "pack_islands" / "pack_max" / "cu_count" (lane-packing of
 * isomorphic islands), "profile_launches", "time_batch", "chain_lds_out" (measurement). Unknown keys return code 6. */
int  elemhip_set_option(elemhip_t*, const char* key, double value);

#ifdef __cplusplus
}
#endif
#endif /* ELEMHIP_H */
