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
 * render sequences, Runtime.h:133,204,277-285).  Calls are serialised internally by a mutex.
 *
 * The engine has NO CPU fallback: every block is rendered by the HIP kernels of
 * elementary_amd/csrc/kernels.hip; elemhip_create fails (NULL) when no gfx950 device is usable.
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
} elemhip_stats;

/* Runtime(double sampleRate, int blockSize)                      runtime/elem/Runtime.h:44,157-166
 * blockSize is the MAX frames per process call (<= 512). Returns NULL on failure; the reason is
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

int  elemhip_trace_level(elemhip_t*, size_t nOut, uint32_t level, unsigned long long* out, size_t cap);
/* Debug/test hook: JSON description of the current render plan (islands, launch levels, LDS).
 * deviceOrdinal == -1 at create time gives a "dry" handle that runs all host logic (instruction
 * decode, graph mutation, plan build, gc) without a GPU; it cannot render (process returns 101). */
size_t elemhip_describe_plan(elemhip_t*, char* buf, size_t cap);
/* Run the launches on a caller-owned hipStream_t (e.g. torch's current stream). */
int  elemhip_set_stream(elemhip_t*, void* hipStream);
/* Tunables: "use_graph" (0/1), "graph_blocks" (blocks per captured hipGraph). */
int  elemhip_set_option(elemhip_t*, const char* key, double value);

#ifdef __cplusplus
}
#endif
#endif /* ELEMHIP_H */
