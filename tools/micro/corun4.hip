// Characterisation of the co-run corruption found by corun3.hip: WHICH registers and WHICH lanes of a VALU-only wave go
// wrong while 16-bit-input MFMA waves of another kernel share its SIMD.  Kernel F<R, OP> keeps R live VGPRs and updates
// them with ONE pinned instruction form (inline asm: 0 v_fma_f32, 1 v_pk_fma_f32 on register pairs, 2 v_mul_f32 + v_add_f32,
// 3 v_mov_b32 only (no arithmetic: pure register traffic)); every register of every thread is written out and compared
// with the solo run.  Kernel M = bf16 32x32x16 MFMAs with B operands streamed from global memory (the strongest trigger).
//   hipcc --offload-arch=gfx950 -O3 -o ab_libs/corun4 tools/micro/corun4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));

template <int R, int OP>
__global__ __launch_bounds__(256) void kern_f(float* out, unsigned* hw, int iters, int nthreads) {
    const int lane = threadIdx.x & 63;
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = 0.25f + 0.001f * lane + 0.01f * i;
    const float k1 = 0.75f, k2 = 0.125f;
    const f32x2 k1p = {0.75f, 0.75f}, k2p = {0.125f, 0.125f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(k1), "v"(k2));
            if (OP == 2) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(k1)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(k2)); }
            if (OP == 3) { float t; asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(v[i])); asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(t)); }
            if (OP == 1 && (i & 1) == 0 && i + 1 < R) {
                f32x2 p = {v[i], v[i + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(k1p), "v"(k2p));
                v[i] = p[0]; v[i + 1] = p[1];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) out[(size_t)i * nthreads + tid] = v[i];
    if (lane == 0) hw[tid >> 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
}

__global__ __launch_bounds__(256, 2) void kern_m(float* out, int iters, const f32x4* __restrict__ wsrc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) ((float*)lds)[i] = 0.001f * i;
    __syncthreads();
    const f32x4* src = (const f32x4*)lds;
    f32x16 c[8];
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 16; ++i) c[q][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const f32x4 a = src[(lane + it) & 1023];
        f32x4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = wsrc[((size_t)(it * 4 + q) * 64 + lane) & 65535];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, bq[q & 3]), c[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 16; ++i) s += c[q][i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int R, int OP>
static void test(const char* name, float* df, unsigned* dhw, float* dm, const f32x4* dw, hipStream_t s1, hipStream_t s2) {
    const int fblocks = 512, n = fblocks * 256, iters = 3000;
    std::vector<float> r0((size_t)R * n), r1((size_t)R * n);
    std::vector<unsigned> hw(n / 64);
    hipLaunchKernelGGL((kern_f<R, OP>), dim3(fblocks), dim3(256), 0, s1, df, dhw, iters, n);
    hipDeviceSynchronize();
    hipMemcpy(r0.data(), df, r0.size() * 4, hipMemcpyDeviceToHost);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)kern_f<R, OP>);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemsetAsync(df, 0, r0.size() * 4, s1);
        hipDeviceSynchronize();
        if (rep) hipLaunchKernelGGL(kern_m, dim3(4096), dim3(256), 78848, s2, dm, 3000, dw);
        hipLaunchKernelGGL((kern_f<R, OP>), dim3(fblocks), dim3(256), 0, s1, df, dhw, iters, n);
        hipDeviceSynchronize();
        hipMemcpy(r1.data(), df, r1.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hw.data(), dhw, hw.size() * 4, hipMemcpyDeviceToHost);
        long bad = 0, by_reg[128] = {}, by_lane[64] = {}, by_slot[16] = {}, bad_waves = 0;
        std::vector<char> wbad(n / 64, 0);
        for (int i = 0; i < R; ++i)
            for (int t = 0; t < n; ++t)
                if (memcmp(&r0[(size_t)i * n + t], &r1[(size_t)i * n + t], 4)) { ++bad; ++by_reg[i]; ++by_lane[t & 63]; wbad[t >> 6] = 1; }
        for (int w = 0; w < n / 64; ++w) if (wbad[w]) { ++bad_waves; ++by_slot[hw[w] & 15]; }
        printf("%-26s R=%3d (%3d VGPRs alloc) %s: bad values %7ld, bad waves %4ld of %d", name, R, fa.numRegs, rep ? "next to M" : "alone    ", bad, bad_waves, n / 64);
        if (bad) {
            printf("\n    by register:");
            for (int i = 0; i < R; ++i) if (by_reg[i]) printf(" r%d:%ld", i, by_reg[i]);
            printf("\n    by lane:");
            for (int l = 0; l < 64; ++l) if (by_lane[l]) printf(" %d:%ld", l, by_lane[l]);
            printf("\n    bad waves by hardware wave slot:");
            for (int s = 0; s < 16; ++s) if (by_slot[s]) printf(" %d:%ld", s, by_slot[s]);
        }
        printf("\n");
    }
}

int main() {
    float *df, *dm; unsigned* dhw;
    hipMalloc(&df, (size_t)128 * 512 * 256 * 4); hipMalloc(&dm, (size_t)4096 * 256 * 4); hipMalloc(&dhw, 2048 * 4);
    f32x4* dw; hipMalloc(&dw, 65536 * 16); hipMemset(dw, 0x3c, 65536 * 16);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    test<16, 0>("v_fma_f32", df, dhw, dm, dw, s1, s2);
    test<28, 0>("v_fma_f32", df, dhw, dm, dw, s1, s2);
    test<40, 0>("v_fma_f32", df, dhw, dm, dw, s1, s2);
    test<56, 0>("v_fma_f32", df, dhw, dm, dw, s1, s2);
    test<88, 0>("v_fma_f32", df, dhw, dm, dw, s1, s2);
    test<120, 0>("v_fma_f32", df, dhw, dm, dw, s1, s2);
    test<28, 1>("v_pk_fma_f32", df, dhw, dm, dw, s1, s2);
    test<56, 1>("v_pk_fma_f32", df, dhw, dm, dw, s1, s2);
    test<56, 2>("v_mul_f32 + v_add_f32", df, dhw, dm, dw, s1, s2);
    test<56, 3>("v_mov_b32 only", df, dhw, dm, dw, s1, s2);
    return 0;
}
