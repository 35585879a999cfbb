// Two KERNELS on two streams, shaped like the real pair (tools/probe_concurrency.py): kernel F = the 512-point FFT of
// mel_frame_kernel in 4-wave workgroups with ~230 VGPRs and 67 KB of LDS (two workgroups per CU); kernel M = bf16 MFMA
// spam in 4-wave workgroups with ~230 VGPRs and 78 KB of LDS.  F's output alone vs next to M, bit for bit.
#include "../../nisqa_amd/csrc/mel.hip"
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void kern_f(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* exch = lds + 16128 + wave * 12928;
    mel_twiddles tw;
    for (int r = 0; r < 4; ++r) { tw.a[r] = cmk(__cosf(0.01f * lane * r), -__sinf(0.01f * lane * r)); tw.d[r] = tw.a[r]; }
    for (int p = 0; p < 8; ++p) { tw.b[p] = cmk(__cosf(0.02f * lane * p), -__sinf(0.02f * lane * p)); tw.c[p] = cmk(__cosf(0.3f * (lane & 7) * p), -__sinf(0.3f * (lane & 7) * p)); }
    c32 z[8], u[8], keep[56];
    for (int a = 0; a < 8; ++a) z[a] = cmk(0.001f * (lane + 64 * a), 0.5f - 0.002f * lane);
    for (int a = 0; a < 56; ++a) keep[a] = cmk(0.01f * a, 0.02f * lane);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        fft512<1>(u, z, tw, exch, lane);
#pragma unroll
        for (int a = 0; a < 8; ++a) { acc += u[a].x - u[a].y; z[a] = u[a] * 0.04f + keep[a + 8 * (it & 1)] * 1e-3f; }
#pragma unroll
        for (int a = 0; a < 56; ++a) keep[a] = keep[a] * 0.999f + u[a & 7] * 1e-6f;
    }
    for (int a = 0; a < 56; ++a) acc += keep[a].x + keep[a].y;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256, 2) void kern_m(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) ((float*)lds)[i] = 0.001f * i;
    __syncthreads();
    const f32x4* src = (const f32x4*)lds;
    f32x16 c[12];
    for (int q = 0; q < 12; ++q) c[q] = zero16();
    for (int it = 0; it < iters; ++it) {
        const f32x4 a = src[(lane + it) & 1023], b = src[(lane * 3 + it) & 1023];
#pragma unroll
        for (int q = 0; q < 12; ++q)
            c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b), c[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 12; ++q) for (int i = 0; i < 16; ++i) s += c[q][i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int blocks = 512, n = blocks * 256;
    float *df, *dm; hipMalloc(&df, n * 4); hipMalloc(&dm, 4096 * 256 * 4);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    std::vector<float> r0(n), r1(n);
    long worst = 0;
    hipLaunchKernelGGL(kern_f, dim3(blocks), dim3(256), 67840, s1, df, 2000);
    hipDeviceSynchronize();
    hipMemcpy(r0.data(), df, n * 4, hipMemcpyDeviceToHost);
    for (int rep = 0; rep < 6; ++rep) {
        hipMemsetAsync(df, 0, n * 4, s1);
        hipDeviceSynchronize();
        if (rep) hipLaunchKernelGGL(kern_m, dim3(4096), dim3(256), 78848, s2, dm, 3000);
        hipLaunchKernelGGL(kern_f, dim3(blocks), dim3(256), 67840, s1, df, 2000);
        hipDeviceSynchronize();
        hipMemcpy(r1.data(), df, n * 4, hipMemcpyDeviceToHost);
        long d = 0;
        for (int i = 0; i < n; ++i) d += memcmp(&r0[i], &r1[i], 4) != 0;
        printf("%s: FFT kernel outputs differing from the solo run: %ld of %d\n", rep ? "next to the bf16-MFMA kernel" : "alone again", d, n);
    }
    return 0;
}
