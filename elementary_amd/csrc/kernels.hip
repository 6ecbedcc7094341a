// kernels.hip — throughput build of the block-render kernels: the 8-wave (512-thread) island kernel that
// pipelines the blocks of a multi-block launch, plus the epilogue / patch kernels and the host launchers.
// The device code itself is island.inc, shared with kernels_rt.hip.
#define ELEMHIP_ISLAND_THREADS 512
#define ELEMHIP_ISLAND_KERNEL elemhip_island_kernel
#define ELEMHIP_AUX 1
#include "island.inc"
