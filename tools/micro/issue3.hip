// Third single-wave micro-benchmark: the lane-parallel block I/O of a one-member recurrence (island_ops.inc,
// chain_loop_uni). A one-pole chain (v_mul, v_add per frame, fully dependent) whose input reaches it through
// v_readlane_b32 (SGPR operand) and whose results are parked in lane k/4 by v_cndmask_b32 under (lane == k/4).
// Variants isolate what each piece costs a lone wavefront and whether software pipelining the cross-lane reads helps.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/issue3_bin tools/micro/issue3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))

#define CLOB "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v16", "v20", "v21", "v22", "v23", "v24", "v25", "s6", "s20", "s24", "s25", "s26", "s27", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "memory", "vcc", "scc"

// one group = 4 frames
#define G_CHAIN "v_mul_f32 v21, v20, v22\n v_add_f32 v22, v23, v21\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, v23, v21\n" \
                "v_mul_f32 v21, v20, v22\n v_add_f32 v22, v23, v21\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, v23, v21\n"
// readlane right before use
#define G_RL    "v_readlane_b32 s32, v6, s6\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s32, v21\n" \
                "v_readlane_b32 s33, v7, s6\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s33, v21\n" \
                "v_readlane_b32 s34, v8, s6\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s34, v21\n" \
                "v_readlane_b32 s35, v9, s6\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s35, v21\n s_add_u32 s6, s6, 1\n s_and_b32 s6, s6, 63\n"
// readlanes of the NEXT group issued between this group's chain ops (two register sets alternate: A uses s32-35 and fetches s36-39)
#define G_RLP_A "v_mul_f32 v21, v20, v22\n v_readlane_b32 s36, v6, s6\n v_add_f32 v22, s32, v21\n" \
                "v_mul_f32 v21, v20, v22\n v_readlane_b32 s37, v7, s6\n v_add_f32 v22, s33, v21\n" \
                "v_mul_f32 v21, v20, v22\n v_readlane_b32 s38, v8, s6\n v_add_f32 v22, s34, v21\n" \
                "v_mul_f32 v21, v20, v22\n v_readlane_b32 s39, v9, s6\n v_add_f32 v22, s35, v21\n s_add_u32 s6, s6, 1\n s_and_b32 s6, s6, 63\n"
#define G_RLP_B "v_mul_f32 v21, v20, v22\n v_readlane_b32 s32, v6, s6\n v_add_f32 v22, s36, v21\n" \
                "v_mul_f32 v21, v20, v22\n v_readlane_b32 s33, v7, s6\n v_add_f32 v22, s37, v21\n" \
                "v_mul_f32 v21, v20, v22\n v_readlane_b32 s34, v8, s6\n v_add_f32 v22, s38, v21\n" \
                "v_mul_f32 v21, v20, v22\n v_readlane_b32 s35, v9, s6\n v_add_f32 v22, s39, v21\n s_add_u32 s6, s6, 1\n s_and_b32 s6, s6, 63\n"
// chain + results parked with cndmask (distinct result registers v22..v25 per frame so the cndmasks are off the chain)
#define G_CND   "v_cmp_eq_u32 vcc, s6, v16\n" \
                "v_mul_f32 v21, v20, v25\n v_add_f32 v22, v23, v21\n v_cndmask_b32 v10, v10, v22, vcc\n" \
                "v_mul_f32 v21, v20, v22\n v_add_f32 v24, v23, v21\n v_cndmask_b32 v11, v11, v24, vcc\n" \
                "v_mul_f32 v21, v20, v24\n v_add_f32 v22, v23, v21\n v_cndmask_b32 v12, v12, v22, vcc\n" \
                "v_mul_f32 v21, v20, v22\n v_add_f32 v25, v23, v21\n v_cndmask_b32 v13, v13, v25, vcc\n s_add_u32 s6, s6, 1\n s_and_b32 s6, s6, 63\n"
// everything, readlane right before use (what the compiler emits for chain_loop_uni)
#define G_ALL   "v_cmp_eq_u32 vcc, s6, v16\n" \
                "v_readlane_b32 s32, v6, s6\n v_mul_f32 v21, v20, v25\n v_add_f32 v22, s32, v21\n v_cndmask_b32 v10, v10, v22, vcc\n" \
                "v_readlane_b32 s33, v7, s6\n v_mul_f32 v21, v20, v22\n v_add_f32 v24, s33, v21\n v_cndmask_b32 v11, v11, v24, vcc\n" \
                "v_readlane_b32 s34, v8, s6\n v_mul_f32 v21, v20, v24\n v_add_f32 v22, s34, v21\n v_cndmask_b32 v12, v12, v22, vcc\n" \
                "v_readlane_b32 s35, v9, s6\n v_mul_f32 v21, v20, v22\n v_add_f32 v25, s35, v21\n v_cndmask_b32 v13, v13, v25, vcc\n s_add_u32 s6, s6, 1\n s_and_b32 s6, s6, 63\n"
// everything, readlanes one group ahead
#define G_ALLP_A "v_cmp_eq_u32 vcc, s6, v16\n" \
                "v_mul_f32 v21, v20, v25\n v_readlane_b32 s36, v6, s6\n v_add_f32 v22, s32, v21\n v_cndmask_b32 v10, v10, v22, vcc\n" \
                "v_mul_f32 v21, v20, v22\n v_readlane_b32 s37, v7, s6\n v_add_f32 v24, s33, v21\n v_cndmask_b32 v11, v11, v24, vcc\n" \
                "v_mul_f32 v21, v20, v24\n v_readlane_b32 s38, v8, s6\n v_add_f32 v22, s34, v21\n v_cndmask_b32 v12, v12, v22, vcc\n" \
                "v_mul_f32 v21, v20, v22\n v_readlane_b32 s39, v9, s6\n v_add_f32 v25, s35, v21\n v_cndmask_b32 v13, v13, v25, vcc\n s_add_u32 s6, s6, 1\n s_and_b32 s6, s6, 63\n"
#define G_ALLP_B "v_cmp_eq_u32 vcc, s6, v16\n" \
                "v_mul_f32 v21, v20, v25\n v_readlane_b32 s32, v6, s6\n v_add_f32 v22, s36, v21\n v_cndmask_b32 v10, v10, v22, vcc\n" \
                "v_mul_f32 v21, v20, v22\n v_readlane_b32 s33, v7, s6\n v_add_f32 v24, s37, v21\n v_cndmask_b32 v11, v11, v24, vcc\n" \
                "v_mul_f32 v21, v20, v24\n v_readlane_b32 s34, v8, s6\n v_add_f32 v22, s38, v21\n v_cndmask_b32 v12, v12, v22, vcc\n" \
                "v_mul_f32 v21, v20, v22\n v_readlane_b32 s35, v9, s6\n v_add_f32 v25, s39, v21\n v_cndmask_b32 v13, v13, v25, vcc\n s_add_u32 s6, s6, 1\n s_and_b32 s6, s6, 63\n"
// which half of "+ v_readlane" costs: the SGPR lane select, or the VALU op that reads the SGPR the readlane wrote?
#define G_RL_IMM "v_readlane_b32 s32, v6, 5\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s32, v21\n" \
                "v_readlane_b32 s33, v7, 5\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s33, v21\n" \
                "v_readlane_b32 s34, v8, 5\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s34, v21\n" \
                "v_readlane_b32 s35, v9, 5\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s35, v21\n s_add_u32 s6, s6, 1\n s_and_b32 s6, s6, 63\n"
#define G_RL_UNUSED "v_readlane_b32 s32, v6, s6\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, v23, v21\n" \
                "v_readlane_b32 s33, v7, s6\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, v23, v21\n" \
                "v_readlane_b32 s34, v8, s6\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, v23, v21\n" \
                "v_readlane_b32 s35, v9, s6\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, v23, v21\n s_add_u32 s6, s6, 1\n s_and_b32 s6, s6, 63\n"
#define G_RFL   "v_readfirstlane_b32 s32, v6\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s32, v21\n" \
                "v_readfirstlane_b32 s33, v7\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s33, v21\n" \
                "v_readfirstlane_b32 s34, v8\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s34, v21\n" \
                "v_readfirstlane_b32 s35, v9\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s35, v21\n s_add_u32 s6, s6, 1\n s_and_b32 s6, s6, 63\n"
// the chain reading a loop-invariant SGPR operand (no readlane at all)
#define G_SGPR  "v_mul_f32 v21, v20, v22\n v_add_f32 v22, s32, v21\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s33, v21\n" \
                "v_mul_f32 v21, v20, v22\n v_add_f32 v22, s34, v21\n v_mul_f32 v21, v20, v22\n v_add_f32 v22, s35, v21\n"
// broadcast through the VALU only: v_mov_b32_dpp row_bcast / quad_perm cannot cross rows, but lane 0's value can be
// spread with v_readlane -> v_mov (SGPR -> VGPR) off the chain, so the chain reads a VGPR
#define G_RL_MOV "v_readlane_b32 s32, v6, s6\n v_mul_f32 v21, v20, v22\n v_mov_b32 v24, s32\n v_add_f32 v22, v24, v21\n" \
                "v_readlane_b32 s33, v7, s6\n v_mul_f32 v21, v20, v22\n v_mov_b32 v24, s33\n v_add_f32 v22, v24, v21\n" \
                "v_readlane_b32 s34, v8, s6\n v_mul_f32 v21, v20, v22\n v_mov_b32 v24, s34\n v_add_f32 v22, v24, v21\n" \
                "v_readlane_b32 s35, v9, s6\n v_mul_f32 v21, v20, v22\n v_mov_b32 v24, s35\n v_add_f32 v22, v24, v21\n s_add_u32 s6, s6, 1\n s_and_b32 s6, s6, 63\n"

template <int V>
__global__ void probe(float* out, int reps) {
    unsigned t0 = 0, t1 = 0;
    float z = out[0];
    unsigned lane = threadIdx.x & 63;
    asm volatile(
        "v_mov_b32 v20, 0x3f7fdf3b\n v_mov_b32 v22, %2\n v_mov_b32 v25, %2\n v_mov_b32 v24, %2\n v_mov_b32 v23, 0.5\n v_mov_b32 v16, %3\n"
        "v_mov_b32 v6, 0.5\n v_mov_b32 v7, 0.5\n v_mov_b32 v8, 0.5\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n"
        "s_mov_b32 s6, 0\n s_mov_b32 s32, 0.5\n s_mov_b32 s33, 0.5\n s_mov_b32 s34, 0.5\n s_mov_b32 s35, 0.5\n s_mov_b32 s36, 0.5\n s_mov_b32 s37, 0.5\n s_mov_b32 s38, 0.5\n s_mov_b32 s39, 0.5\n"
        "s_memtime s[24:25]\n s_waitcnt lgkmcnt(0)\n"
        "s_mov_b32 s20, %4\n"
        "1:\n"
        ".if %c5 == 0\n" R16(G_CHAIN) ".endif\n"
        ".if %c5 == 1\n" R16(G_RL) ".endif\n"
        ".if %c5 == 2\n" R4(R4(G_RLP_A G_RLP_B)) ".endif\n"
        ".if %c5 == 3\n" R16(G_CND) ".endif\n"
        ".if %c5 == 4\n" R16(G_ALL) ".endif\n"
        ".if %c5 == 5\n" R4(R4(G_ALLP_A G_ALLP_B)) ".endif\n"
        ".if %c5 == 6\n" R16(G_RL_IMM) ".endif\n"
        ".if %c5 == 7\n" R16(G_RL_UNUSED) ".endif\n"
        ".if %c5 == 8\n" R16(G_RFL) ".endif\n"
        ".if %c5 == 9\n" R16(G_SGPR) ".endif\n"
        ".if %c5 == 10\n" R16(G_RL_MOV) ".endif\n"
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
        "s_memtime s[26:27]\n s_waitcnt lgkmcnt(0)\n"
        "v_mov_b32 %0, s24\n v_mov_b32 %1, s26\n v_add_f32 v22, v22, v10\n v_add_f32 v22, v22, v25\n v_mov_b32 %2, v22\n"
        : "=v"(t0), "=v"(t1), "+v"(z)
        : "v"(lane), "s"(reps), "n"(V)
        : CLOB);
    if (threadIdx.x == 0) { out[1] = z; ((unsigned*)out)[2] = t1 - t0; }
}

template <int V>
static void run(const char* name, int groupsPerRep) {
    float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    const int reps = 2000;
    probe<V><<<1, 64>>>(d, reps); hipDeviceSynchronize();
    probe<V><<<1, 64>>>(d, reps); hipDeviceSynchronize();
    unsigned h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-72s %8.2f cycles per 4-frame group\n", name, (double)h[2] / ((double)reps * groupsPerRep));
    hipFree(d);
}

int main() {
    run<0>("chain only (4 x mul, add)", 16);
    run<1>("+ 4 v_readlane right before use", 16);
    run<2>("+ 4 v_readlane one group ahead", 32);
    run<3>("+ v_cmp + 4 v_cndmask", 16);
    run<4>("+ readlane before use + cmp + cndmask  (chain_loop_uni as compiled)", 16);
    run<5>("+ readlane one group ahead + cmp + cndmask", 32);
    run<6>("+ 4 v_readlane, IMMEDIATE lane, result used by the chain", 16);
    run<7>("+ 4 v_readlane, SGPR lane, result NOT used", 16);
    run<8>("+ 4 v_readfirstlane, result used by the chain", 16);
    run<9>("chain with a loop-invariant SGPR operand, no cross-lane op", 16);
    run<10>("+ 4 x (v_readlane, v_mov to a VGPR the chain reads)", 16);
    return 0;
}
