// Micro-benchmark: do MFMA and VALU instructions of one SIMD overlap?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o ab_libs/issue tools/micro/issue.hip ; run on the GPU box
// Each test runs one workgroup per CU slot (grid 256 x waves) and reports wall time per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)

template <int MODE>   // 5: as 3 with s_nop padding after each MFMA, 6: one wave, MFMA / 8 VALU interleaved, 7: as 5 with s_sleep; 0: MFMA only, 1: VALU only, 2: interleaved in one wave, 3: even waves MFMA / odd waves VALU, 4: permlane16_swap
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(lane * 0.002f - i); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, v4 = lane + 4, v5 = lane + 5, v6 = lane + 6, v7 = lane + 7;
    unsigned p0 = lane, p1 = lane * 3;
    // waves are placed round-robin on the CU's four SIMDs (wave w and w + 4 share one): the role is bit 2 of the wave id
    const int role = (wave >> 2) & 1;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && role == 0);
    const bool do_v = MODE == 1 || MODE == 2 || ((MODE == 3 || MODE == 5 || MODE == 7 || MODE == 8) && role == 1);
    const bool do_mn = (MODE == 5 || MODE == 7 || MODE == 8) && role == 0;
    for (int it = 0; it < iters; ++it) {
        if (do_m) { MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); }                       // 4 x 32 cycles
        if (do_v) {                                                                  // 32 x 4 cycles
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
            }
        }
        if (do_mn) {
#define PAD() do { if (MODE == 5) { asm volatile("s_nop 15"); asm volatile("s_nop 9"); } else if (MODE == 7) { __builtin_amdgcn_s_sleep(1); } else { asm volatile("s_nop 15"); asm volatile("s_nop 15"); asm volatile("s_nop 3");} } while (0)
            MFMA(c0); PAD(); MFMA(c1); PAD(); MFMA(c2); PAD(); MFMA(c3); PAD();
        }
        if (MODE == 6) {
#define V8() do { v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f); \
                  v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f); } while (0)
            MFMA(c0); __builtin_amdgcn_sched_barrier(0); V8(); __builtin_amdgcn_sched_barrier(0);
            MFMA(c1); __builtin_amdgcn_sched_barrier(0); V8(); __builtin_amdgcn_sched_barrier(0);
            MFMA(c2); __builtin_amdgcn_sched_barrier(0); V8(); __builtin_amdgcn_sched_barrier(0);
            MFMA(c3); __builtin_amdgcn_sched_barrier(0); V8(); __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 4) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { auto r = __builtin_amdgcn_permlane16_swap(p0, p1, false, false); p0 = r[0] + 1; p1 = r[1] + 1; }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)p0 + (float)p1;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int waves_per_simd, float* out) {
    const int iters = 20000, threads = 64 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s waves/SIMD %d: %8.1f ns per iteration\n", name, waves_per_simd, ms * 1e6 / iters);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0>("MFMA only (4 x 32x32x16 bf16 per iteration)", 1, out);
    run<0>("MFMA only", 2, out);
    run<1>("VALU only (32 fma per iteration)", 1, out);
    run<1>("VALU only", 2, out);
    run<2>("MFMA + VALU in the same wave", 1, out);
    run<2>("MFMA + VALU in the same wave", 2, out);
    run<3>("one MFMA wave + one VALU wave per SIMD", 2, out);
    run<4>("16 x permlane16_swap (dependent)", 1, out);
    run<5>("MFMA wave (+ s_nop 26) + VALU wave per SIMD", 2, out);
    run<8>("MFMA wave (+ s_nop 36) + VALU wave per SIMD", 2, out);
    run<7>("MFMA wave (+ s_sleep 1) + VALU wave per SIMD", 2, out);
    run<6>("one wave: MFMA, 8 VALU, MFMA, 8 VALU ...", 1, out);
    run<6>("same, two waves", 2, out);
    return 0;
}
