// launch.h — host-callable launchers for the kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include "device.h"

namespace elemhip {

// Output bus channels per process call (the epilogue kernels' thread-per-channel tables; device.h's kMaxOut = 256 was the r01-r03
// limit and is still what the JIT-compiled island kernels see — they never look at it)
constexpr uint32_t kMaxOutBus = 1024;
constexpr uint32_t kEventLogEntries = 1024;   // per-block readout log of a meter / snapshot node (device.h EVT_LOG), a power of two

// ---- the resident form of the single-block path (resident.hip) ----
constexpr uint32_t kResidentQuit = 0xFFFFFFFFu;   // block number that asks the kernel to leave
constexpr uint32_t kResidentMaxLevels = 32;
constexpr uint32_t kResidentMaxRoots = 64;
struct ResidentCtl {              // mapped, coherent host memory; one cache line per direction
    uint32_t seq;                 // host -> device: number of the block to render next (1, 2, ...; kResidentQuit: leave)
    uint32_t pad0[15];
    uint32_t done;                // device -> host: number of the block whose output is in the host's output block
    uint32_t exited;              // device -> host: 0 running, 1 left (asked to, or idle), 2 a device-wide barrier timed out (block lost)
    uint32_t ticksBody, ticksEpilogue;   // of the last block: block number seen -> levels rendered -> output block written (10 ns units)
    uint32_t pad1[12];
};
struct ResidentLevels { uint32_t count; uint32_t offset[kResidentMaxLevels + 1]; };   // entries of PlanView::levelIslands per launch level
hipError_t configure_resident(uint32_t maxLdsBytes);
hipError_t launch_resident(hipStream_t s, const PlanView& pv, uint32_t* recs, float* hbm, Globals* g, const uint32_t* lcg,
                           const ResidentLevels& lv, uint32_t groups, uint32_t ldsBytes, ResidentCtl* ctlDev, const float* inHostDev,
                           float* outHostDev, unsigned long long* sync, uint64_t idleTicks, uint64_t hangTicks);

hipError_t configure_kernels(uint32_t maxLdsBytes);
void launch_level(hipStream_t s, const PlanView& pv, uint32_t* recs, float* hbm, const Globals* g, const uint32_t* lcg,
                  uint32_t levelBegin, uint32_t numIslands, uint32_t ldsBytes, uint32_t batch = 1, uint32_t arenaFloats = 0,
                  uint32_t statelessRows = 8);
// (`doneFlag`: a word in mapped host memory the epilogue of a ONE-block set publishes `doneValue` to behind its output, or null)
void launch_epilogue_batch(hipStream_t s, const PlanView& pv, uint32_t* recs, const float* hbm, Globals* g, float* outRing,
                           uint32_t batch, uint32_t arenaFloats, uint32_t* doneFlag = nullptr, uint32_t doneValue = 0u);
void launch_epilogue(hipStream_t s, const PlanView& pv, uint32_t* recs, const float* hbm, Globals* g, float* outRing,
                     uint32_t* doneFlag = nullptr, uint32_t doneValue = 0u);
hipError_t configure_kernels_rt(uint32_t maxLdsBytes);
void launch_level_rt(hipStream_t s, const PlanView& pv, uint32_t* recs, float* hbm, const Globals* g, const uint32_t* lcg,
                     uint32_t levelBegin, uint32_t numIslands, uint32_t ldsBytes);
void launch_convolve(hipStream_t s, const PlanView& pv, uint32_t* recs, float* hbm, const Globals* g,
                     uint32_t workBegin, uint32_t numWorkgroups);
size_t convolve_batch_scratch_floats(uint32_t maxBatch, uint32_t longHistRows);   // per convolve node (longHistRows: 0 = no long-partition area)
void launch_convolve_batch(hipStream_t s, const PlanView& pv, uint32_t* recs, float* hbm, const Globals* g, uint32_t workBegin,
                           uint32_t numNodes, uint32_t batch, uint32_t arenaFloats, float* scratch, uint32_t maxBatch, uint32_t macMode,
                           bool anyShortIr, bool anyLongIr, uint32_t longHistRows, bool anyShortPath, uint32_t longStateBlocks, uint32_t longMacMode,
                           const float* inDirect, uint32_t numInCh, float* outDirect, uint32_t numOutCh);
void launch_convolve_fix_overlap(hipStream_t s, const PlanView& pv, uint32_t* recs, float* hbm, const Globals* g, uint32_t workBegin,
                                 uint32_t numWork, float* scratch, uint32_t maxBatch, uint32_t longHistRows, uint32_t maxPartitions);
uint32_t convolve_mfma_max_partitions();   // IRs of up to this many 512-tap partitions take the matrix-core MAC
uint32_t convolve_long_tap_group();        // long-partition IR spectra are allocated in multiples of this many rows
uint32_t convolve_long_row_floats();       // floats per long-partition spectrum row (4097 bins, padded)
hipError_t upload_convolve_tables(const float* twiddleReIm, const float* twiddle8192ReIm);
void launch_patches(hipStream_t s, const Patch* patches, uint32_t count, uint32_t* recs, uint32_t* globals);
hipError_t launch_bus_sum(hipStream_t s, float* dst, const float* const* partials, uint32_t count, size_t n);   // dst = ((p0 + p1) + p2) + ... (rank order)

} // namespace elemhip
