// resident.hip — the resident form of the single-block render path (option `resident`, VERDICT r04 "next" #7).
//
// elemhip_process() otherwise costs a host two or more kernel launches and a stream synchronise per 512-frame block — ~17 us of the
// 24 us a call takes on the cli benchmark's graph, where the reference's own call (Runtime.h:275-291, one walk of the render
// sequence on the calling thread) takes 19-28 us on a host core. This kernel stays on the GPU between calls instead: workgroup 0
// watches a word in mapped host memory; when the host publishes a block number it brings the block's input channels in, every
// workgroup renders its islands level by level (the same island body as kernels_rt.hip, behind device-wide barriers where a plan has
// more than one level or more islands than workgroups), workgroup 0 sums the roots into the host's mapped output block
// (GraphRenderSequence.h:286-295, 227-231), promotes the tap buffers (:297-308), advances the sample clock and publishes the block
// number back. No launch, no synchronise: the host spins on that word.
//
// A kernel that waits for a host is a kernel that can wait forever, so every wait in here is bounded by the realtime counter
// (s_memrealtime, 100 MHz): idle for `idleTicks` -> workgroup 0 tells the others to leave and the kernel ends (exit code 1; the host
// launches it again when it next has a run of plain blocks); a device-wide barrier that does not complete within `hangTicks` ->
// everybody leaves with exit code 2 and the engine reports an error (the block is lost). The host asks it to leave (block number
// kResidentQuit) before anything else touches the engine's stream — a commit, a property, an event relay, a launch set.
//
// What is rendered is what elemhip_island_kernel_rt + elemhip_epilogue_kernel render, on the same records and arena: the engine only
// goes resident while every running root's fade is settled and the plan has no convolvers or call-out nodes (enqueueBlock's other
// duties), so the epilogue here is the bus sum, the tap promotion and the clock.
#define ELEMHIP_ISLAND_THREADS 256
#define ELEMHIP_RESIDENT 1
#include "island.inc"

namespace {

__device__ __forceinline__ uint64_t now_ticks() { return __builtin_amdgcn_s_memrealtime(); }

// Device-wide barrier between the levels of a block: a monotonic arrival count in device memory (`epoch` barriers x G arrivals).
// Release / acquire at agent scope: the XCDs' L2s do not snoop each other, what a level exports to the arena has to be written
// back before the next level of another XCD reads it.
__device__ __forceinline__ bool grid_barrier(unsigned long long* count, unsigned long long* abortWord, uint64_t target, uint64_t hangTicks) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(count, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t t0 = now_ticks();
        while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {   // (the acquire is the fence below)
            if (__hip_atomic_load(abortWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull || now_ticks() - t0 > hangTicks) {
                __hip_atomic_store(abortWord, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        reinterpret_cast<uint32_t*>(lds)[1] = ok ? 1u : 0u;
    }
    __syncthreads();
    ok = reinterpret_cast<uint32_t*>(lds)[1] != 0u;
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}

// The output bus of the block, summed per channel over the running roots in render-sequence order (bus_sum of island.inc without its
// static tables: this kernel's LDS is the islands'). `out` is the host's block in mapped, coherent host memory: 16 bytes per lane,
// a wave's stores cover whole cache lines (single-float stores reach host memory as 1024 partial-line writes per block and took
// 100 us to drain: gpurun r05g).
__device__ __forceinline__ void resident_bus_sum(const PlanView& pv, gcup recs, gcfp hbm, gfp out, uint32_t n, uint32_t numOut, uint32_t stride) {
    int* rootChan = reinterpret_cast<int*>(lds) + 4;          // [numRoots] (<= kResidentMaxRoots)
    const uint32_t nr = pv.numRoots;
    if (threadIdx.x < nr) {
        const uint32_t rr = pv.roots[threadIdx.x].rec;
        rootChan[threadIdx.x] = root_running(recs, rr, numOut) ? (int)recs[rr * kRecDwords + rec::ROOT_CHANNEL] : -1;
    }
    __syncthreads();
    if (((n | stride) & 3u) == 0u && ((uintptr_t)out & 15u) == 0u && ((uintptr_t)hbm & 15u) == 0u) {
        typedef float bf4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(1))) const bf4* gcf4p;
        typedef __attribute__((address_space(1))) bf4* gf4p;
        const uint32_t n4 = n >> 2;
        for (uint32_t idx = threadIdx.x; idx < numOut * n4; idx += blockDim.x) {
            const uint32_t ch = idx / n4, i = (idx - ch * n4) << 2;
            bf4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            for (uint32_t r = 0; r < nr; ++r) if (rootChan[r] == (int)ch) acc += *(gcf4p)(hbm + (size_t)pv.roots[r].hbm * stride + i);
            *(gf4p)(out + (size_t)ch * stride + i) = acc;
        }
    } else {
        for (uint32_t idx = threadIdx.x; idx < numOut * n; idx += blockDim.x) {
            const uint32_t ch = idx / n, i = idx - ch * n;
            float acc = 0.0f;
            for (uint32_t r = 0; r < nr; ++r) if (rootChan[r] == (int)ch) acc += hbm[(size_t)pv.roots[r].hbm * stride + i];
            out[(size_t)ch * stride + i] = acc;
        }
    }
    __syncthreads();
}

} // namespace

__global__ __launch_bounds__(ELEMHIP_ISLAND_THREADS)
void elemhip_resident_kernel(PlanView pv, uint32_t* recs, float* hbm, Globals* g, const uint32_t* lcg, ResidentLevels lv,
                             ResidentCtl* ctl, const float* inHost, float* outHost, unsigned long long* sync,
                             uint64_t idleTicks, uint64_t hangTicks) {
    if (__builtin_amdgcn_groupstaticsize() != 0u) __builtin_trap();
    const uint32_t G = gridDim.x, w = blockIdx.x;
    uint32_t* ldsu_ = reinterpret_cast<uint32_t*>(lds);
    uint32_t last = 0u;                       // the block number rendered last (the host starts at 1)
    uint64_t epoch = 0;                       // device-wide barriers passed
    uint32_t code = 1u;
    uint64_t tSeen = 0, tBody = 0;
    for (;;) {
        // ---- the next block number: workgroup 0 from the host, the others from workgroup 0
        if (threadIdx.x == 0) {
            uint32_t seq = last;
            const uint64_t t0 = now_ticks();
            if (w == 0) {
                for (;;) {
                    seq = __hip_atomic_load(&ctl->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (seq != last) { tSeen = now_ticks(); break; }
                    if (now_ticks() - t0 > idleTicks) { seq = kResidentQuit; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            } else {
                for (;;) {
                    seq = (uint32_t)__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (acquire: the fence below)
                    if (seq != last) break;
                    if (now_ticks() - t0 > idleTicks + hangTicks) { seq = kResidentQuit; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            ldsu_[0] = seq;
        }
        __syncthreads();
        const uint32_t seq = UNI(ldsu_[0]);
        __syncthreads();
        if (w == 0) {
            if (seq != kResidentQuit) {
                // the host's input channels: mapped host memory -> arena buffers 0 .. numIn - 1 (what the H2D copy of the launch path does)
                const uint32_t words = g->numIn * g->blockStride;
                if (words) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");         // (0.4 us: tools/micro/fence_cost.hip)
                    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) hbm[i] = __builtin_nontemporal_load(inHost + i);
                }
            }
            if (G > 1u) {
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(&sync[0], (unsigned long long)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (seq == kResidentQuit) break;
        if (G > 1u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (one workgroup: its own L1 is all there is between two blocks)
        // ---- the block, level by level
        bool ok = true;
        for (uint32_t l = 0; l < lv.count && ok; ++l) {
            for (uint32_t i = lv.offset[l] + w; i < lv.offset[l + 1]; i += G) {
                island_body(pv, recs, hbm, g, lcg, 0u, i, 1u, 0u);
                __syncthreads();
            }
            if (G > 1u) { ++epoch; ok = grid_barrier(&sync[1], &sync[2], epoch * G, hangTicks); }
        }
        if (!ok) { code = 2u; break; }
        // ---- epilogue
        if (w == 0) {
            if (threadIdx.x == 0) tBody = now_ticks();
            const uint32_t n = g->numSamples, numOut = g->numOut, stride = g->blockStride;
            resident_bus_sum(pv, (gcup)recs, (gcfp)hbm, (gfp)outHost, n, numOut, stride);
            promote_taps(pv, (gup)recs, n);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");              // every thread's stores of the output block have completed (0.5 us) ...
            __syncthreads();
            if (threadIdx.x == 0) {
                g->sampleTime += (int64_t)n;
                const uint64_t tEnd = now_ticks();
                __hip_atomic_store(&ctl->ticksBody, (uint32_t)(tBody - tSeen), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&ctl->ticksEpilogue, (uint32_t)(tEnd - tBody), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&ctl->done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before the block number goes out
            }
        }
        last = seq;
    }
    if (w == 0 && threadIdx.x == 0) __hip_atomic_store(&ctl->exited, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

namespace elemhip {

hipError_t configure_resident(uint32_t maxLdsBytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(elemhip_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxLdsBytes);
}

hipError_t launch_resident(hipStream_t s, const PlanView& pv, uint32_t* recs, float* hbm, Globals* g, const uint32_t* lcg,
                           const ResidentLevels& lv, uint32_t groups, uint32_t ldsBytes, ResidentCtl* ctlDev, const float* inHostDev,
                           float* outHostDev, unsigned long long* sync, uint64_t idleTicks, uint64_t hangTicks) {
    hipLaunchKernelGGL(elemhip_resident_kernel, dim3(groups), dim3(ELEMHIP_ISLAND_THREADS), ldsBytes, s, pv, recs, hbm, g, lcg, lv, ctlDev,
                       inHostDev, outHostDev, sync, idleTicks, hangTicks);
    return hipGetLastError();
}

} // namespace elemhip
