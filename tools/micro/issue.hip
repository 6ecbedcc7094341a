// Single-wave issue / dependent-latency micro-benchmarks for gfx950: what one lone wavefront (the regime every
// serial recurrence of the island kernel lives in) pays per dependent VALU op, and how many independent
// instructions ride along for free. Output: cycles per op / per recurrence step (s_memtime ticks = shader cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

extern __shared__ __attribute__((aligned(16))) float lds[];

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

// mode-selected straight-line bodies, 256 repetitions each, `reps` loop iterations
template <int MODE>
__global__ void valu(float* out, int reps, float p, float xin, int lanes) {
    float z = out[0], x = xin, t = 0.0f, u = 1.0f, w = 2.0f;
    double zd = (double)z, pd = (double)p, xd = (double)xin;
    if ((int)(threadIdx.x & 63) >= lanes) return;
    unsigned long long t0 = now();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0)  asm volatile(R256("v_add_f32 %0, %0, %1\n") : "+v"(z) : "v"(x));
        if (MODE == 1)  asm volatile(R256("v_mul_f32 %1, %2, %0\nv_add_f32 %0, %3, %1\n") : "+v"(z), "+v"(t) : "v"(p), "v"(x));
        if (MODE == 2)  asm volatile(R256("v_mul_f32 %1, %2, %0\nv_mov_b32 %4, %3\nv_add_f32 %0, %3, %1\nv_mov_b32 %5, %3\n") : "+v"(z), "+v"(t) : "v"(p), "v"(x), "v"(u), "v"(w));
        if (MODE == 3)  asm volatile(R256("v_mul_f32 %1, %2, %0\nv_mov_b32 %4, %3\nv_mov_b32 %5, %3\nv_add_f32 %0, %3, %1\nv_mov_b32 %4, %3\nv_mov_b32 %5, %3\n") : "+v"(z), "+v"(t) : "v"(p), "v"(x), "v"(u), "v"(w));
        if (MODE == 4)  asm volatile(R256("v_mul_f32 %1, %2, %0\ns_nop 0\nv_add_f32 %0, %3, %1\ns_nop 0\n") : "+v"(z), "+v"(t) : "v"(p), "v"(x));
        if (MODE == 5)  asm volatile(R256("v_add_f32 %0, %0, %1\nv_fract_f32 %0, %0\n") : "+v"(z) : "v"(x));
        if (MODE == 6)  asm volatile(R256("v_fma_f64 %0, %0, %1, %2\n") : "+v"(zd) : "v"(pd), "v"(xd));
        if (MODE == 7)  asm volatile(R256("v_mul_f64 %0, %0, %1\nv_add_f64 %0, %0, %2\n") : "+v"(zd) : "v"(pd), "v"(xd));
        if (MODE == 8)  asm volatile(R256("v_mul_f32 %1, %2, %0\ns_mov_b32 s20, s21\nv_add_f32 %0, %3, %1\ns_mov_b32 s20, s21\n") : "+v"(z), "+v"(t) : "v"(p), "v"(x) : "s20");
        if (MODE == 9)  asm volatile(R256("v_mul_f32 %1, %2, %0\nv_mul_f32 %4, %2, %5\nv_add_f32 %0, %3, %1\nv_add_f32 %5, %3, %4\n") : "+v"(z), "+v"(t) : "v"(p), "v"(x), "v"(u), "v"(w));   // two interleaved chains
        if (MODE == 10) asm volatile(R256("v_mul_f32 %1, %2, %0\nv_add_f32 %0, %3, %1\n") : "+v"(z), "+v"(t) : "s"(p), "v"(x));   // coefficient in an SGPR
        if (MODE == 11) asm volatile(R256("v_mul_f32_e64 %1, %2, %0\nv_add_f32_e64 %0, %3, %1\n") : "+v"(z), "+v"(t) : "v"(p), "v"(x));   // VOP3 encodings
        if (MODE == 12) asm volatile(R256("v_fma_f32 %0, %1, %0, %2\n") : "+v"(z) : "v"(p), "v"(x));
        if (MODE == 13) asm volatile(R256("v_mul_f32 %1, %2, %0\nv_mov_b32 %4, %3\nv_mov_b32 %5, %3\nv_mov_b32 %4, %3\nv_add_f32 %0, %3, %1\nv_mov_b32 %4, %3\nv_mov_b32 %5, %3\nv_mov_b32 %4, %3\n") : "+v"(z), "+v"(t) : "v"(p), "v"(x), "v"(u), "v"(w));
        // biquad TDF-II critical cycle (Filters.h:104-110): y = b0 x + z1; z1 = b1 x - a1 y + z2; z2 = b2 x - a2 y
        if (MODE == 14) asm volatile(R256("v_add_f32 %1, %4, %0\nv_mul_f32 %5, %2, %1\nv_sub_f32 %5, %3, %5\nv_add_f32 %0, %5, %6\nv_mul_f32 %6, %2, %1\nv_sub_f32 %6, %3, %6\n") : "+v"(z), "+v"(t) : "v"(p), "v"(x), "v"(xin), "v"(u), "v"(w));
    }
    unsigned long long t1 = now();
    out[1 + (threadIdx.x & 63)] = z + t + u + w + (float)zd;
    if ((threadIdx.x & 63) == 0) ((unsigned long long*)out)[40 + (threadIdx.x >> 6)] = t1 - t0;
}

// The one-pole block loop as the code generator would emit it: 512 frames from an LDS slot to an LDS slot, 4 frames per
// ds_read_b128 / ds_write_b128, reads PF chunks ahead. VARIANT picks the wait placement.
template <int VARIANT>
__global__ void pole_lds(float* out, int reps, float p) {
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0.001f * (i & 511);
    __syncthreads();
    if ((threadIdx.x >> 6) != 0) return;
    float z = out[0];
    unsigned long long t0 = now();
    const unsigned inB = 16, outB = 16 + 4 * 1024;   // byte addresses
    for (int r = 0; r < reps; ++r) {
        if (VARIANT == 0) {
            // registers: x chunks in v[10:13], v[14:17], v[18:21] (ring of 3), outputs v[22:25]
            asm volatile(
                "v_mov_b32 v8, %2\n v_mov_b32 v9, %3\n"
                "ds_read_b128 v[10:13], v8\n ds_read_b128 v[14:17], v8 offset:16\n ds_read_b128 v[18:21], v8 offset:32\n"
                "s_movk_i32 s20, 42\n"          // 42 iterations of 12 frames + tail = 504 + 8
                "1:\n"
                "s_waitcnt lgkmcnt(2)\n"
                "v_mul_f32 v26, %1, %0\n v_add_f32 v22, v10, v26\n v_mul_f32 v26, %1, v22\n v_add_f32 v23, v11, v26\n"
                "v_mul_f32 v26, %1, v23\n v_add_f32 v24, v12, v26\n v_mul_f32 v26, %1, v24\n v_add_f32 v25, v13, v26\n"
                "ds_write_b128 v9, v[22:25]\n ds_read_b128 v[10:13], v8 offset:48\n"
                "s_waitcnt lgkmcnt(3)\n"
                "v_mul_f32 v26, %1, v25\n v_add_f32 v22, v14, v26\n v_mul_f32 v26, %1, v22\n v_add_f32 v23, v15, v26\n"
                "v_mul_f32 v26, %1, v23\n v_add_f32 v24, v16, v26\n v_mul_f32 v26, %1, v24\n v_add_f32 v25, v17, v26\n"
                "ds_write_b128 v9, v[22:25] offset:16\n ds_read_b128 v[14:17], v8 offset:64\n"
                "s_waitcnt lgkmcnt(4)\n"
                "v_mul_f32 v26, %1, v25\n v_add_f32 v22, v18, v26\n v_mul_f32 v26, %1, v22\n v_add_f32 v23, v19, v26\n"
                "v_mul_f32 v26, %1, v23\n v_add_f32 v24, v20, v26\n v_mul_f32 v26, %1, v24\n v_add_f32 %0, v21, v26\n"
                "v_mov_b32 v25, %0\n"
                "ds_write_b128 v9, v[22:25] offset:32\n ds_read_b128 v[18:21], v8 offset:80\n"
                "v_add_u32 v8, 48, v8\n v_add_u32 v9, 48, v9\n"
                "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "+v"(z) : "v"(p), "v"(inB), "v"(outB)
                : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "s20", "scc", "memory");
        }
        if (VARIANT == 1) {
            // same, but the output quad is the add's destination and state stays in the last output register: no v_mov;
            // writes issued right after the 4th add of a chunk, reads issued in the shadow of the first mul
            asm volatile(
                "v_mov_b32 v8, %2\n v_mov_b32 v9, %3\n v_mov_b32 v25, %0\n"
                "ds_read_b128 v[10:13], v8\n ds_read_b128 v[14:17], v8 offset:16\n"
                "s_movk_i32 s20, 64\n"          // 64 iterations of 8 frames
                "1:\n"
                "s_waitcnt lgkmcnt(1)\n"
                "v_mul_f32 v26, %1, v25\n ds_read_b128 v[18:21], v8 offset:32\n v_add_f32 v22, v10, v26\n v_mul_f32 v26, %1, v22\n v_add_f32 v23, v11, v26\n"
                "v_mul_f32 v26, %1, v23\n v_add_f32 v24, v12, v26\n v_mul_f32 v26, %1, v24\n v_add_f32 v25, v13, v26\n"
                "ds_write_b128 v9, v[22:25]\n"
                "s_waitcnt lgkmcnt(2)\n"
                "v_mul_f32 v26, %1, v25\n ds_read_b128 v[10:13], v8 offset:48\n v_add_f32 v22, v14, v26\n v_mul_f32 v26, %1, v22\n v_add_f32 v23, v15, v26\n"
                "v_mul_f32 v26, %1, v23\n v_add_f32 v24, v16, v26\n v_mul_f32 v26, %1, v24\n v_add_f32 v25, v17, v26\n"
                "ds_write_b128 v9, v[22:25] offset:16\n"
                "v_add_u32 v8, 32, v8\n v_add_u32 v9, 32, v9\n"
                "v_mov_b32 v14, v18\n v_mov_b32 v15, v19\n v_mov_b32 v16, v20\n v_mov_b32 v17, v21\n"
                "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
                "s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v25\n"
                : "+v"(z) : "v"(p), "v"(inB), "v"(outB)
                : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "s20", "scc", "memory");
        }
        if (VARIANT == 2) {
            // everything loaded up front (512 frames = 128 ds_read_b128 would not fit): 64 frames at a time into 64 VGPRs,
            // pure VALU chain of 64 steps, then 16 writes
            for (int c = 0; c < 8; ++c) {
                const unsigned ib = inB + c * 256, ob = outB + c * 256;
                asm volatile(
                    "v_mov_b32 v8, %2\n v_mov_b32 v9, %3\n"
                    "ds_read_b128 v[30:33], v8\n ds_read_b128 v[34:37], v8 offset:16\n ds_read_b128 v[38:41], v8 offset:32\n ds_read_b128 v[42:45], v8 offset:48\n"
                    "ds_read_b128 v[46:49], v8 offset:64\n ds_read_b128 v[50:53], v8 offset:80\n ds_read_b128 v[54:57], v8 offset:96\n ds_read_b128 v[58:61], v8 offset:112\n"
                    "ds_read_b128 v[62:65], v8 offset:128\n ds_read_b128 v[66:69], v8 offset:144\n ds_read_b128 v[70:73], v8 offset:160\n ds_read_b128 v[74:77], v8 offset:176\n"
                    "ds_read_b128 v[78:81], v8 offset:192\n ds_read_b128 v[82:85], v8 offset:208\n ds_read_b128 v[86:89], v8 offset:224\n ds_read_b128 v[90:93], v8 offset:240\n"
                    "s_waitcnt lgkmcnt(0)\n"
#define ST(k) "v_mul_f32 v26, %1, %0\n v_add_f32 %0, v" #k ", v26\n v_mov_b32 v" #k ", %0\n"
                    ST(30) ST(31) ST(32) ST(33) ST(34) ST(35) ST(36) ST(37) ST(38) ST(39) ST(40) ST(41) ST(42) ST(43) ST(44) ST(45)
                    ST(46) ST(47) ST(48) ST(49) ST(50) ST(51) ST(52) ST(53) ST(54) ST(55) ST(56) ST(57) ST(58) ST(59) ST(60) ST(61)
                    ST(62) ST(63) ST(64) ST(65) ST(66) ST(67) ST(68) ST(69) ST(70) ST(71) ST(72) ST(73) ST(74) ST(75) ST(76) ST(77)
                    ST(78) ST(79) ST(80) ST(81) ST(82) ST(83) ST(84) ST(85) ST(86) ST(87) ST(88) ST(89) ST(90) ST(91) ST(92) ST(93)
#undef ST
                    "ds_write_b128 v9, v[30:33]\n ds_write_b128 v9, v[34:37] offset:16\n ds_write_b128 v9, v[38:41] offset:32\n ds_write_b128 v9, v[42:45] offset:48\n"
                    "ds_write_b128 v9, v[46:49] offset:64\n ds_write_b128 v9, v[50:53] offset:80\n ds_write_b128 v9, v[54:57] offset:96\n ds_write_b128 v9, v[58:61] offset:112\n"
                    "ds_write_b128 v9, v[62:65] offset:128\n ds_write_b128 v9, v[66:69] offset:144\n ds_write_b128 v9, v[70:73] offset:160\n ds_write_b128 v9, v[74:77] offset:176\n"
                    "ds_write_b128 v9, v[78:81] offset:192\n ds_write_b128 v9, v[82:85] offset:208\n ds_write_b128 v9, v[86:89] offset:224\n ds_write_b128 v9, v[90:93] offset:240\n"
                    : "+v"(z) : "v"(p), "v"(ib), "v"(ob)
                    : "v8", "v9", "v26", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45",
                      "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61",
                      "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77",
                      "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "memory");
            }
        }
        if (VARIANT == 3) {
            // 64 frames per group like variant 2, but the chain writes its result IN PLACE (v_add dst = the x register):
            // exactly 2 VALU per frame, nothing else on the chain
            for (int c = 0; c < 8; ++c) {
                const unsigned ib = inB + c * 256, ob = outB + c * 256;
                asm volatile(
                    "v_mov_b32 v8, %2\n v_mov_b32 v9, %3\n v_mov_b32 v27, %0\n"
                    "ds_read_b128 v[30:33], v8\n ds_read_b128 v[34:37], v8 offset:16\n ds_read_b128 v[38:41], v8 offset:32\n ds_read_b128 v[42:45], v8 offset:48\n"
                    "ds_read_b128 v[46:49], v8 offset:64\n ds_read_b128 v[50:53], v8 offset:80\n ds_read_b128 v[54:57], v8 offset:96\n ds_read_b128 v[58:61], v8 offset:112\n"
                    "ds_read_b128 v[62:65], v8 offset:128\n ds_read_b128 v[66:69], v8 offset:144\n ds_read_b128 v[70:73], v8 offset:160\n ds_read_b128 v[74:77], v8 offset:176\n"
                    "ds_read_b128 v[78:81], v8 offset:192\n ds_read_b128 v[82:85], v8 offset:208\n ds_read_b128 v[86:89], v8 offset:224\n ds_read_b128 v[90:93], v8 offset:240\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "v_mul_f32 v26, %1, v27\n v_add_f32 v30, v30, v26\n"
#define ST(k, j) "v_mul_f32 v26, %1, v" #j "\n v_add_f32 v" #k ", v" #k ", v26\n"
                    ST(31,30) ST(32,31) ST(33,32) ST(34,33) ST(35,34) ST(36,35) ST(37,36) ST(38,37) ST(39,38) ST(40,39) ST(41,40) ST(42,41) ST(43,42) ST(44,43) ST(45,44)
                    ST(46,45) ST(47,46) ST(48,47) ST(49,48) ST(50,49) ST(51,50) ST(52,51) ST(53,52) ST(54,53) ST(55,54) ST(56,55) ST(57,56) ST(58,57) ST(59,58) ST(60,59) ST(61,60)
                    ST(62,61) ST(63,62) ST(64,63) ST(65,64) ST(66,65) ST(67,66) ST(68,67) ST(69,68) ST(70,69) ST(71,70) ST(72,71) ST(73,72) ST(74,73) ST(75,74) ST(76,75) ST(77,76)
                    ST(78,77) ST(79,78) ST(80,79) ST(81,80) ST(82,81) ST(83,82) ST(84,83) ST(85,84) ST(86,85) ST(87,86) ST(88,87) ST(89,88) ST(90,89) ST(91,90) ST(92,91) ST(93,92)
#undef ST
                    "v_mov_b32 %0, v93\n"
                    "ds_write_b128 v9, v[30:33]\n ds_write_b128 v9, v[34:37] offset:16\n ds_write_b128 v9, v[38:41] offset:32\n ds_write_b128 v9, v[42:45] offset:48\n"
                    "ds_write_b128 v9, v[46:49] offset:64\n ds_write_b128 v9, v[50:53] offset:80\n ds_write_b128 v9, v[54:57] offset:96\n ds_write_b128 v9, v[58:61] offset:112\n"
                    "ds_write_b128 v9, v[62:65] offset:128\n ds_write_b128 v9, v[66:69] offset:144\n ds_write_b128 v9, v[70:73] offset:160\n ds_write_b128 v9, v[74:77] offset:176\n"
                    "ds_write_b128 v9, v[78:81] offset:192\n ds_write_b128 v9, v[82:85] offset:208\n ds_write_b128 v9, v[86:89] offset:224\n ds_write_b128 v9, v[90:93] offset:240\n"
                    : "+v"(z) : "v"(p), "v"(ib), "v"(ob)
                    : "v8", "v9", "v26", "v27", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45",
                      "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61",
                      "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77",
                      "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "memory");
            }
        }
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) { out[1] = z + lds[16 / 4 + 1024 + 5]; ((unsigned long long*)out)[40] = t1 - t0; }
}

static float* d;
static unsigned long long h[64];
template <int MODE> void runv(const char* what, int opsPerRep, int threads = 64, int lanes = 64) {
    for (int k = 0; k < 2; ++k) { hipLaunchKernelGGL(valu<MODE>, dim3(1), dim3(threads), 0, 0, d, 16, 0.9995f, 0.25f, lanes); hipDeviceSynchronize(); }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-86s %7.2f cycles/%s", what, (double)h[40] / (16.0 * 256.0), opsPerRep == 1 ? "op  " : "step");
    for (int w = 1; w < threads / 64; ++w) printf("  w%d %.2f", w, (double)h[40 + w] / (16.0 * 256.0));
    printf("\n");
}
template <int V> void runp(const char* what) {
    for (int k = 0; k < 2; ++k) { hipLaunchKernelGGL(pole_lds<V>, dim3(1), dim3(64), 32768, 0, d, 16, 0.9995f); hipDeviceSynchronize(); }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-86s %7.2f cycles/frame  (%.0f per 512-frame block)\n", what, (double)h[40] / (16.0 * 512.0), (double)h[40] / 16.0);
}
int main() {
    hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    runv<0>("dependent v_add_f32", 1);
    runv<12>("dependent v_fma_f32", 1);
    runv<1>("pole step: v_mul -> v_add (dependent pair)", 2);
    runv<10>("pole step, coefficient in an SGPR", 2);
    runv<11>("pole step, VOP3 encodings", 2);
    runv<2>("pole step + 1 independent v_mov after each op", 2);
    runv<3>("pole step + 2 independent v_mov after each op", 2);
    runv<13>("pole step + 3 independent v_mov after each op", 2);
    runv<4>("pole step + s_nop 0 after each op", 2);
    runv<8>("pole step + 1 s_mov after each op", 2);
    runv<9>("two interleaved pole chains (cycles per step of BOTH)", 2);
    runv<5>("phasor step: v_add -> v_fract", 2);
    runv<14>("biquad critical cycle (add, mul, sub, add) + 2 side ops", 2);
    runv<6>("dependent v_fma_f64", 1);
    runv<7>("dependent v_mul_f64 -> v_add_f64", 2);
    runv<1>("pole step, 1 active lane", 2, 64, 1);
    runv<1>("pole step, 32 active lanes", 2, 64, 32);
    runv<1>("pole step, 8 waves (2 per SIMD) all chaining", 2, 512, 64);
    runv<1>("pole step, 16 waves (4 per SIMD) all chaining", 2, 1024, 64);
    runp<0>("pole over LDS, 12 frames/iter, v_mov for the state");
    runp<1>("pole over LDS, 8 frames/iter, reads in the mul shadow");
    runp<2>("pole over LDS, 64 frames per group: load all, chain (+v_mov per frame), store all");
    runp<3>("pole over LDS, 64 frames per group, in-place chain: 2 VALU per frame only");
    return 0;
}
