// Second single-wave micro-benchmark: what LDS / SMEM / VMEM / cross-lane instructions cost a lone wavefront when they
// are interleaved with a dependent VALU chain (8 chain ops + 1 probe instruction per group), with one or all lanes active.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
extern __shared__ __attribute__((aligned(16))) float lds[];
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

#define CH8 "v_mul_f32 v21, v20, v22\n v_add_f32 v22, v23, v21\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, v23, v21\n" \
            "v_mul_f32 v21, v20, v22\n v_add_f32 v22, v23, v21\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, v23, v21\n"
#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "memory", "vcc", "scc"

// EXECM: 0 = all lanes, 1 = lane 0 only. ADDR: 0 = every lane the same address, 1 = lane * 16 bytes
template <int PROBE, int EXECM, int ADDR>
__global__ void probe(float* out, const float* gsrc, float* gdst, int reps) {
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.001f * (i & 511);
    __syncthreads();
    unsigned a = ADDR ? (threadIdx.x & 63) * 16u : 0u;
    unsigned long long t0 = 0, t1 = 0;
    float z = out[0];
        asm volatile(
        "v_mov_b32 v8, %3\n v_mov_b32 v9, %3\n v_mov_b32 v20, 0x3f7fdf3b\n v_mov_b32 v22, %2\n v_mov_b32 v23, 0.5\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n"
        "s_mov_b64 s[30:31], exec\n"
        ".if %c7\n s_mov_b64 exec, 1\n .endif\n"
        "s_memtime s[24:25]\n s_waitcnt lgkmcnt(0)\n"
        "s_mov_b32 s20, %6\n"
        "1:\n"
        R64(
        CH8
        ".if %c8 == 1\n ds_read_b128 v[10:13], v8\n .endif\n"
        ".if %c8 == 2\n ds_write_b128 v8, v[10:13] offset:16384\n .endif\n"
        ".if %c8 == 3\n ds_read_b64 v[10:11], v8\n .endif\n"
        ".if %c8 == 4\n ds_write_b64 v8, v[10:11] offset:16384\n .endif\n"
        ".if %c8 == 5\n ds_read_b32 v10, v8\n .endif\n"
        ".if %c8 == 6\n ds_write_b32 v8, v10 offset:16384\n .endif\n"
        ".if %c8 == 7\n s_load_dwordx16 s[32:47], %4, 0x0\n .endif\n"
        ".if %c8 == 8\n global_store_dwordx4 v9, v[10:13], %5\n .endif\n"
        ".if %c8 == 9\n global_load_dwordx4 v[14:17], v9, %4\n .endif\n"
        ".if %c8 == 10\n v_readlane_b32 s22, v22, 5\n .endif\n"
        ".if %c8 == 11\n v_writelane_b32 v24, s22, 7\n .endif\n"
        ".if %c8 == 12\n s_waitcnt lgkmcnt(0)\n .endif\n"
        ".if %c8 == 13\n ds_read_b128 v[10:13], v8\n ds_write_b128 v8, v[14:17] offset:16384\n .endif\n"
        ".if %c8 == 14\n s_nop 1\n .endif\n"
        ".if %c8 == 15\n v_mov_b32_dpp v24, v22 wave_shr:1 row_mask:0xf bank_mask:0xf\n .endif\n"
        ".if %c8 == 16\n ds_write_b128 v8, v[10:13] offset:16384\n s_waitcnt lgkmcnt(0)\n .endif\n"
        ".if %c8 == 17\n ds_read_b128 v[10:13], v8\n s_waitcnt lgkmcnt(0)\n .endif\n"
        ".if %c8 == 18\n v_add_u32 v8, 0, v8\n .endif\n"
        ".if %c8 == 19\n global_store_dwordx4 v9, v[10:13], %5\n s_waitcnt vmcnt(0)\n .endif\n"
        ".if %c8 == 20\n s_load_dwordx16 s[32:47], %4, 0x0\n s_waitcnt lgkmcnt(0)\n .endif\n"
        ".if %c8 == 21\n s_dcache_inv\n .endif\n"
        ".if %c8 == 22\n s_sleep 0\n .endif\n"
        )
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "s_memtime s[26:27]\n s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 exec, s[30:31]\n"
        "v_mov_b32 %0, s24\n v_mov_b32 %1, s26\n v_add_f32 v22, v22, v10\n v_add_f32 v22, v22, v14\n v_mov_b32 %2, v22\n"
        : "=v"(*(unsigned*)&t0), "=v"(*(unsigned*)&t1), "+v"(z)
        : "v"(a), "s"(gsrc), "s"(gdst), "s"(reps), "n"(EXECM), "n"(PROBE)
        : CLOB);
    if (threadIdx.x == 0) { out[1] = z; ((unsigned*)out)[2] = (unsigned)t1 - (unsigned)t0; }
}

// DPP hop chain: lane k computes step k (idempotent recompute of the lanes before it). NOP = wait states inserted between
// the add that writes Z and the DPP multiply that reads it. Checks the 64 results against a serial evaluation.
template <int NOP>
__global__ void dpp_chain(float* out, int reps) {
    const unsigned lane = threadIdx.x & 63;
    float x = 0.001f * lane + 0.1f, p = 0.9995f, z = 0.25f, t = p * 0.25f;   // T[0] = p * z_prev preset; Z don't care
    unsigned t0 = 0, t1 = 0;
    asm volatile(
        "s_memtime s[24:25]\n s_waitcnt lgkmcnt(0)\n"
        "s_mov_b32 s20, %6\n"
        "1:\n"
        R64(
        "v_mul_f32_dpp %3, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32 %2, %5, %3\n"
        ".if %c7 == 1\n s_nop 0\n .endif\n"
        ".if %c7 == 2\n s_nop 1\n .endif\n"
        )
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
        "s_memtime s[26:27]\n s_waitcnt lgkmcnt(0)\n"
        "v_mov_b32 %0, s24\n v_mov_b32 %1, s26\n"
        : "=v"(t0), "=v"(t1), "+v"(z), "+v"(t)
        : "v"(p), "v"(x), "s"(reps), "n"(NOP)
        : "s20", "s24", "s25", "s26", "s27", "scc", "memory");
    out[8 + lane] = z;
    if (threadIdx.x == 0) ((unsigned*)out)[2] = t1 - t0;
}

static float *d, *gs, *gd;
static float h[128];
template <int P, int E, int A> void run(const char* what) {
    for (int k = 0; k < 2; ++k) { hipLaunchKernelGGL((probe<P, E, A>), dim3(1), dim3(64), 65536, 0, d, gs, gd, 8); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double per = (double)((unsigned*)h)[2] / (8.0 * 64.0);
    printf("%-64s %s %s  %7.2f cycles per group of 8 chain ops + probe\n", what, E ? "1 lane  " : "64 lanes", A ? "distinct" : "same    ", per);
}
#define RUN3(P, what) run<P, 1, 0>(what); run<P, 0, 0>(what); run<P, 0, 1>(what);
int main() {
    hipMalloc(&d, 4096); hipMemset(d, 0, 4096); hipMalloc(&gs, 1 << 16); hipMemset(gs, 0, 1 << 16); hipMalloc(&gd, 1 << 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<0, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<0, 1, 0>("chain only"); run<0, 0, 0>("chain only");
    RUN3(1, "+ ds_read_b128") RUN3(2, "+ ds_write_b128") RUN3(3, "+ ds_read_b64") RUN3(4, "+ ds_write_b64") RUN3(5, "+ ds_read_b32") RUN3(6, "+ ds_write_b32")
    RUN3(13, "+ ds_read_b128 + ds_write_b128")
    run<16, 1, 0>("+ ds_write_b128 + lgkmcnt(0)"); run<17, 1, 0>("+ ds_read_b128 + lgkmcnt(0)");
    run<7, 1, 0>("+ s_load_dwordx16 (no wait)"); run<20, 1, 0>("+ s_load_dwordx16 + lgkmcnt(0)"); run<21, 1, 0>("+ s_dcache_inv");
    RUN3(8, "+ global_store_dwordx4") run<19, 1, 0>("+ global_store_dwordx4 + vmcnt(0)"); RUN3(9, "+ global_load_dwordx4")
    run<10, 1, 0>("+ v_readlane_b32"); run<11, 1, 0>("+ v_writelane_b32"); run<12, 1, 0>("+ s_waitcnt lgkmcnt(0) (nothing pending)");
    run<14, 1, 0>("+ s_nop 1"); run<15, 0, 0>("+ v_mov_b32_dpp wave_shr:1"); run<18, 1, 0>("+ v_add_u32"); run<22, 1, 0>("+ s_sleep 0");
    // serial reference for the DPP chain
    float ref[64]; { float z = 0.25f; for (int k = 0; k < 64; ++k) { volatile float t = 0.9995f * z; z = (0.001f * k + 0.1f) + t; ref[k] = z; } }
    auto chk = [&](const char* what, int nop) {
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0; for (int k = 0; k < 64; ++k) if (h[8 + k] != ref[k]) ++bad;
        printf("DPP hop chain, %-22s %7.2f cycles/step, %d of 64 lanes differ from the serial chain\n", what, (double)((unsigned*)h)[2] / (1.0 * 64.0), bad);
    };
    hipLaunchKernelGGL(dpp_chain<0>, dim3(1), dim3(64), 0, 0, d, 1); hipDeviceSynchronize(); chk("no wait states", 0);
    hipLaunchKernelGGL(dpp_chain<1>, dim3(1), dim3(64), 0, 0, d, 1); hipDeviceSynchronize(); chk("s_nop 0", 1);
    hipLaunchKernelGGL(dpp_chain<2>, dim3(1), dim3(64), 0, 0, d, 1); hipDeviceSynchronize(); chk("s_nop 1", 2);
    return 0;
}
