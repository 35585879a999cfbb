// MFMA / VALU co-issue on one gfx950 SIMD, instruction streams pinned with inline asm (the compiler neither reorders nor
// pads them).  Answers, against /opt/skills/guides/MI355X_MICROARCH.md ("<= 5 single-issue instructions hidden per
// v_mfma_f32_32x32x16_bf16 gap", "VALU issue is arbitrated between the two waves of a SIMD by priority, then age"):
//   1. one wave per SIMD: cycles per MFMA with n = 0..8 independent VALU fillers behind every MFMA;
//   2. two waves per SIMD running the same MFMA + n filler stream;
//   3. an MFMA (+ n fillers) wave next to a VALU-only / VALU + LDS "epilogue" wave, with s_setprio on either;
//   4. the same with v_mfma_f32_16x16x32_bf16.
// Role A is measured over its whole run (s_memtime); role B (waves 4..7 of a 512-thread workgroup: wave w and w + 4
// share a SIMD) keeps running until A is done and reports how many of ITS iterations it completed per A iteration.
//   hipcc --offload-arch=gfx950 -O3 -o ab_libs/issue2 tools/micro/issue2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// bodies: 0..8 = 8 x (MFMA 32x32x16 + n fillers); 10..14 = 8 x (MFMA 16x16x32 + (n - 10) fillers);
// 20 = 40 v_fma (VALU only); 21 = 32 VALU + 8 ds_write_b16 (epilogue-like); 22 = 24 VALU + 8 v_cvt_pk + 8 ds_write_b16
template <int BODY>
__device__ __forceinline__ void body(f32x16 (&c)[4], f32x4 (&c4)[4], f32x4 a, f32x4 b, float (&f)[8], unsigned lds_w) {
    const float k1 = 1.0001f, k2 = 0.5f;
    if constexpr (BODY <= 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[q & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < BODY; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(q * BODY + k) & 7]) : "v"(k1), "v"(k2));
        }
    } else if constexpr (BODY >= 10 && BODY <= 14) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c4[q & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < BODY - 10; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(q * (BODY - 10) + k) & 7]) : "v"(k1), "v"(k2));
        }
    } else if constexpr (BODY == 20) {
#pragma unroll
        for (int k = 0; k < 40; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k & 7]) : "v"(k1), "v"(k2));
    } else if constexpr (BODY == 21 || BODY == 22) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int k = 0; k < (BODY == 21 ? 4 : 3); ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(g * 4 + k) & 7]) : "v"(k1), "v"(k2));
            unsigned pk;
            if (BODY == 22) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(f[g & 7]), "v"(f[(g + 1) & 7]));
            else pk = __float_as_uint(f[g & 7]);
            asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"(lds_w), "v"(pk), "n"(0));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int BODY_A, int BODY_B, int PRIO_A, int PRIO_B>
__global__ __launch_bounds__(512) void k(unsigned long long* res, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    volatile int* flag = (volatile int*)lds;
    if (threadIdx.x == 0) { flag[0] = 0; }
    __syncthreads();
    f32x4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = __uint_as_float(0x3c003c00u + lane); b[i] = __uint_as_float(0x3b803b80u + i); }
    f32x16 c[4];
    f32x4 c4[4];
    for (int q = 0; q < 4; ++q) { for (int i = 0; i < 16; ++i) c[q][i] = 0.f; c4[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float f[8];
    for (int q = 0; q < 8; ++q) f[q] = lane + q;
    const unsigned lds_w = 1024 + wave * 2048 + lane * 4;   // byte address in LDS (no static LDS: dynamic starts at 0)
    const bool role_b = wave >= 4;
    unsigned long long t0 = 0, t1 = 0;
    long done = 0;
    if (!role_b) {
        if (PRIO_A) __builtin_amdgcn_s_setprio(PRIO_A);
        t0 = clock64();
        for (int it = 0; it < iters; ++it) body<BODY_A>(c, c4, a, b, f, lds_w);
        t1 = clock64();
        if (lane == 0) atomicAdd((int*)flag, 1);
    } else {
        if (PRIO_B) __builtin_amdgcn_s_setprio(PRIO_B);
        t0 = clock64();
        while (flag[0] < 4) { body<BODY_B>(c, c4, a, b, f, lds_w); ++done; }
        t1 = clock64();
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float s = 0.f;
    for (int q = 0; q < 4; ++q) { for (int i = 0; i < 16; ++i) s += c[q][i]; s += c4[q][0] + c4[q][2]; }
    for (int q = 0; q < 8; ++q) s += f[q];
    if (s == 123.456f) res[100] = 1;                       // keep everything alive
    if (lane == 0) {
        atomicAdd(&res[role_b ? 2 : 0], t1 - t0);
        atomicAdd(&res[role_b ? 3 : 1], role_b ? (unsigned long long)done : (unsigned long long)iters);
    }
}

template <int BODY_A, int BODY_B, int PRIO_A, int PRIO_B>
static void run(const char* name, int waves_per_simd, unsigned long long* res) {
    const int iters = 4000;
    unsigned long long h[4];
    hipMemset(res, 0, 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BODY_A, BODY_B, PRIO_A, PRIO_B>), dim3(256), dim3(256 * waves_per_simd), 100 * 1024, 0, res, 50);
    hipMemset(res, 0, 1024);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BODY_A, BODY_B, PRIO_A, PRIO_B>), dim3(256), dim3(256 * waves_per_simd), 100 * 1024, 0, res, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, res, 32, hipMemcpyDeviceToHost);
    const double cyc_a = (double)h[0] / (double)h[1];                        // cycles per A iteration (8 MFMAs for MFMA bodies)
    const int n_m = BODY_A <= 14 ? 8 : 1;
    printf("%-58s A: %7.1f cyc/iter (%6.2f per MFMA)", name, cyc_a, cyc_a / n_m);
    if (waves_per_simd == 2) {
        const double iters_b_per_a = (double)h[3] / (double)h[1];            // B iterations completed per A iteration
        printf("  B: %5.2f iter per A iter (%7.1f cyc per B iter)", iters_b_per_a, (double)h[2] / (double)(h[3] ? h[3] : 1));
    }
    printf("   [%.3f ms]\n", ms);
}

int main() {
    unsigned long long* res; hipMalloc(&res, 1024);
    printf("-- one wave per SIMD: 8 x (v_mfma_f32_32x32x16_bf16 + n v_fma_f32) per iteration\n");
    run<0, 0, 0, 0>("n = 0", 1, res);  run<1, 0, 0, 0>("n = 1", 1, res);  run<2, 0, 0, 0>("n = 2", 1, res);
    run<3, 0, 0, 0>("n = 3", 1, res);  run<4, 0, 0, 0>("n = 4", 1, res);  run<5, 0, 0, 0>("n = 5", 1, res);
    run<6, 0, 0, 0>("n = 6", 1, res);  run<7, 0, 0, 0>("n = 7", 1, res);  run<8, 0, 0, 0>("n = 8", 1, res);
    printf("-- one wave per SIMD: 8 x (v_mfma_f32_16x16x32_bf16 + n v_fma_f32)\n");
    run<10, 0, 0, 0>("n = 0", 1, res); run<11, 0, 0, 0>("n = 1", 1, res); run<12, 0, 0, 0>("n = 2", 1, res);
    run<13, 0, 0, 0>("n = 3", 1, res); run<14, 0, 0, 0>("n = 4", 1, res);
    printf("-- one wave per SIMD: VALU only / epilogue-like (40 instructions per iteration)\n");
    run<20, 0, 0, 0>("40 v_fma", 1, res); run<21, 0, 0, 0>("32 v_fma + 8 ds_write_b16", 1, res); run<22, 0, 0, 0>("24 v_fma + 8 cvt_pk + 8 ds_write_b16", 1, res);
    printf("-- two waves per SIMD, same stream in both (A = waves 0-3, B = waves 4-7 until A is done)\n");
    run<0, 0, 0, 0>("MFMA n=0 | MFMA n=0", 2, res);   run<2, 2, 0, 0>("MFMA n=2 | MFMA n=2", 2, res);
    run<4, 4, 0, 0>("MFMA n=4 | MFMA n=4", 2, res);   run<20, 20, 0, 0>("40 v_fma | 40 v_fma", 2, res);
    printf("-- two waves per SIMD: MFMA stream (A) next to a VALU stream (B)\n");
    run<0, 20, 0, 0>("MFMA n=0 | 40 v_fma", 2, res);            run<0, 20, 1, 0>("MFMA n=0 prio 1 | 40 v_fma", 2, res);
    run<0, 20, 0, 1>("MFMA n=0 | 40 v_fma prio 1", 2, res);     run<2, 20, 0, 0>("MFMA n=2 | 40 v_fma", 2, res);
    run<2, 20, 1, 0>("MFMA n=2 prio 1 | 40 v_fma", 2, res);     run<2, 20, 0, 1>("MFMA n=2 | 40 v_fma prio 1", 2, res);
    run<4, 20, 0, 0>("MFMA n=4 | 40 v_fma", 2, res);            run<4, 20, 1, 0>("MFMA n=4 prio 1 | 40 v_fma", 2, res);
    run<2, 22, 0, 0>("MFMA n=2 | epilogue-like (cvt + ds_write)", 2, res);
    run<2, 22, 1, 0>("MFMA n=2 prio 1 | epilogue-like", 2, res); run<2, 22, 0, 1>("MFMA n=2 | epilogue-like prio 1", 2, res);
    printf("-- two waves per SIMD: VALU stream measured (A) next to an MFMA stream (B)\n");
    run<20, 0, 0, 0>("40 v_fma | MFMA n=0", 2, res);            run<20, 2, 0, 0>("40 v_fma | MFMA n=2", 2, res);
    run<20, 2, 1, 0>("40 v_fma prio 1 | MFMA n=2", 2, res);     run<20, 2, 0, 1>("40 v_fma | MFMA n=2 prio 1", 2, res);
    run<22, 2, 0, 0>("epilogue-like | MFMA n=2", 2, res);
    printf("-- 16x16x32 next to VALU\n");
    run<10, 20, 0, 0>("MFMA16 n=0 | 40 v_fma", 2, res);         run<12, 20, 0, 0>("MFMA16 n=2 | 40 v_fma", 2, res);
    run<12, 20, 1, 0>("MFMA16 n=2 prio 1 | 40 v_fma", 2, res);
    return 0;
}
