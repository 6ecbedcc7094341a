// Micro-benchmark: the one-pole recurrence loop of the island kernel in isolation (z = x + p*z over 512 LDS samples,
// 16-byte LDS reads one chunk ahead, 16-byte writes), to separate the hardware's dependent-op latency from what the
// interpreter context adds. Variants: active lanes (64 / 1), waves per workgroup (1 / 8, the others idle at a barrier or
// spinning on an LDS word), VGPR budget (launch bounds 64 vs 512 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ __attribute__((aligned(16))) float lds[];
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f ld4(unsigned w) { return *reinterpret_cast<const v4f*>(__builtin_assume_aligned(&lds[w], 16)); }
__device__ __forceinline__ void st4(unsigned w, v4f v) { *reinterpret_cast<v4f*>(__builtin_assume_aligned(&lds[w], 16)) = v; }

template <int MODE>   // bit4: the coefficient lives in a VGPR (loaded per lane) instead of an SGPR; bit0: only lane 0 active; bit1: other waves spin-poll LDS instead of waiting at the barrier; bit2: they run VALU streams; bit3: only wave 4 does
__global__ void pole(float* out, float p_, int reps) {
    float p = p_;
    if (MODE & 16) p = out[8 + (threadIdx.x & 63)];   // host wrote 0.95 there: same value, but the compiler must keep it per lane
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = 0.001f * i;
    if (threadIdx.x == 0) lds[4096] = 0.0f;
    __syncthreads();
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long t0 = 0, t1 = 0;
    if (wave == 0) {
        if (!(MODE & 1) || lane == 0) {
            float z = 0.0f;
            t0 = clock64();
            for (int r = 0; r < reps; ++r) {
                const unsigned in = 0, outw = 1024;
                v4f a = ld4(in), b = ld4(in + 4);
                for (unsigned t = 0; t < 512; t += 8) {
                    const unsigned nx = t + 8 < 512 ? t + 8 : t;
                    const v4f na = ld4(in + nx), nb = ld4(in + nx + 4);
                    v4f ya, yb;
                    z = a.x + p * z; ya.x = z; z = a.y + p * z; ya.y = z; z = a.z + p * z; ya.z = z; z = a.w + p * z; ya.w = z;
                    z = b.x + p * z; yb.x = z; z = b.y + p * z; yb.y = z; z = b.z + p * z; yb.z = z; z = b.w + p * z; yb.w = z;
                    st4(outw + t, ya); st4(outw + t + 4, yb);
                    a = na; b = nb;
                }
            }
            t1 = clock64();
            if (lane == 0) { out[0] = z; ((unsigned*)out)[1] = (unsigned)(t1 - t0); }
            __atomic_store_n((unsigned*)&lds[4096], 1u, __ATOMIC_RELEASE);
        }
    } else if (MODE & 4) {   // busy neighbours: independent multiply-add streams until wave 0 is done (wave 4 shares wave 0's SIMD)
        if ((MODE & 8) && wave != 4) { /* only the SIMD mate works */ }
        else {
            float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
            while (__hip_atomic_load((unsigned*)&lds[4096], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u)
                for (int i = 0; i < 64; ++i) { x0 = x0 * p + 1.0f; x1 = x1 * p + 1.0f; x2 = x2 * p + 1.0f; x3 = x3 * p + 1.0f; }
            if (x0 + x1 + x2 + x3 == 123.0f) out[2] = x0;
        }
    } else if (MODE & 2) {
        while (__hip_atomic_load((unsigned*)&lds[4096], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}
template <int MODE>
void run(const char* what, int threads, float* d) {
    unsigned h[2];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(pole<MODE>, dim3(1), dim3(threads), 20000, 0, d, 0.95f, 8); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("%-70s %.1f cycles / 512-frame block, %.2f per frame\n", what, h[1] / 8.0, h[1] / 8.0 / 512.0);
}
int main() {
    float* d; hipMalloc(&d, 4096);
    { float h[128]; for (int i = 0; i < 128; ++i) h[i] = 0.95f; hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice); }
    run<0>("1 wave, 64 lanes active", 64, d);
    run<1>("1 wave, lane 0 only", 64, d);
    run<1>("8 waves, lane 0 of wave 0, others at the barrier", 512, d);
    run<3>("8 waves, lane 0 of wave 0, others polling an LDS word (s_sleep 2)", 512, d);
    run<2>("8 waves, wave 0 all lanes, others polling", 512, d);
    run<17>("1 wave, lane 0 only, coefficient in a VGPR", 64, d);
    run<16>("1 wave, 64 lanes, coefficient in a VGPR", 64, d);
    run<5>("8 waves, lane 0 of wave 0, the other 7 running independent mul/add streams", 512, d);
    run<13>("8 waves, lane 0 of wave 0, only wave 4 (same SIMD) running mul/add streams", 512, d);
    run<5>("4 waves, lane 0 of wave 0, the other 3 (other SIMDs) running mul/add streams", 256, d);
    return 0;
}
