// tests/native/rccl_bus_sum.cpp — the multi-GPU exchange step of INTEGRATION.md §6 as a COMPILED translation unit: every rank renders its
// share of the voices into its own output bus (elemhip_process_blocks), rank 0 collects the buses over RCCL (xGMI point-to-point)
// and adds them in rank order (elemhip_sum_buses: the same bits on every run). tests/test_host_logic.py compiles and LINKS this file
// against librccl and libelemhip (it is not run: the authoring container has no GPU and the test boxes have one) so that the snippet
// in the documentation cannot rot. north_star: "RCCL over xGMI only for the final output-bus reduce".
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <vector>

#include <elemhip.h>

// one exchange: `bus` (device, n floats) of every rank -> `out` on rank 0 = ((bus_0 + bus_1) + ...) in rank order
extern "C" int elemhip_example_ordered_bus_sum(ncclComm_t comm, int rank, int world, int device, hipStream_t stream,
                                               const float* bus, float* const* part /* rank 0: world - 1 receive buffers at [1..] */,
                                               float* out, size_t n) {
    ncclResult_t rc = ncclGroupStart();
    if (rc != ncclSuccess) return (int)rc;
    if (rank == 0) { for (int r = 1; r < world; ++r) (void)ncclRecv(part[r], n, ncclFloat, r, comm, stream); }
    else (void)ncclSend(bus, n, ncclFloat, 0, comm, stream);
    rc = ncclGroupEnd();
    if (rc != ncclSuccess) return (int)rc;
    if (rank != 0) return 0;
    std::vector<const float*> parts((size_t)world);
    parts[0] = bus;
    for (int r = 1; r < world; ++r) parts[(size_t)r] = part[r];
    return elemhip_sum_buses(device, (void*)stream, out, parts.data(), (size_t)world, n);
}

// the one-call alternative when the last bit may differ from run to run (what bench.py --gpus N times)
extern "C" int elemhip_example_bus_reduce(ncclComm_t comm, hipStream_t stream, const float* bus, float* out, size_t n) {
    return (int)ncclReduce(bus, out, n, ncclFloat, ncclSum, 0, comm, stream);
}

int main() { std::printf("compiled and linked: %p %p\n", (void*)&elemhip_example_ordered_bus_sum, (void*)&elemhip_example_bus_reduce); return 0; }
