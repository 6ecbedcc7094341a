// kernels_rt.hip — single-block (realtime path) build of the island kernel: 4 waves per workgroup, each
// covering two of the plan's 8 program waves. With 256 threads the register allocator has the whole 512-entry
// file per lane and the kernel needs no scratch; the 512-thread build spills ~212 B/lane, which costs a
// one-block launch several microseconds.
#define ELEMHIP_ISLAND_THREADS 256
#define ELEMHIP_ISLAND_KERNEL elemhip_island_kernel_rt
#include "island.inc"
