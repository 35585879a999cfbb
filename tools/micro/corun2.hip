// Two KERNELS on two streams, shaped like the real pair (tools/probe_concurrency.py): kernel F = the 512-point FFT of
// mel_frame_kernel in 4-wave workgroups with ~230 VGPRs and 67 KB of LDS (two workgroups per CU); kernel M = bf16 MFMA
// spam in 4-wave workgroups with ~230 VGPRs and 78 KB of LDS.  F's output alone vs next to M, bit for bit.
#include "../../nisqa_amd/csrc/mel.hip"
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));

template <int MODE>
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
        if (MODE == 0) fft512<1>(u, z, tw, exch, lane);
        if (MODE == 1 || MODE == 3) {                                   // the FFT's arithmetic without its LDS transposes
#pragma unroll
            for (int a = 0; a < 8; ++a) u[a] = z[a];
            dft8(u);
#pragma unroll
            for (int p = 1; p < 8; ++p) u[p] = cmul(cmul(u[p], tw.b[p]), tw.a[1]);
            dft8(u);
#pragma unroll
            for (int p = 1; p < 8; ++p) u[p] = cmul(u[p], tw.c[p]);
            dft8(u);
        }
        if (MODE == 4) {                                   // plain v_fma_f32 / v_add_f32 arithmetic, no packed instructions
            float f[16];
#pragma unroll
            for (int a = 0; a < 8; ++a) { f[2 * a] = z[a].x; f[2 * a + 1] = z[a].y; }
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int a = 0; a < 16; ++a) f[a] = fmaf(f[(a + 5) & 15], 0.37f, f[a] * 0.61f) + 0.01f * tw.b[a & 7].x;
#pragma unroll
            for (int a = 0; a < 8; ++a) u[a] = cmk(f[2 * a], f[2 * a + 1]);
        }
        if (MODE == 2) {                                   // the FFT's LDS transposes without its arithmetic
            c32* b1 = (c32*)exch;
#pragma unroll
            for (int p = 0; p < 8; ++p) b1[p * 72 + lane] = z[p];
            __builtin_amdgcn_wave_barrier();
            const int pq = lane >> 3, j1 = lane & 7;
#pragma unroll
            for (int j2 = 0; j2 < 8; ++j2) u[j2] = b1[pq * 72 + j1 + 8 * j2];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q1 = 0; q1 < 8; ++q1) *(c32*)(exch + (pq + 8 * q1) * 80 + j1 * 8) = u[q1];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t4 = *(const f32x4*)(exch + lane * 80 + q * 16);
                u[2 * q] = cmk(t4[0], t4[1]);
                u[2 * q + 1] = cmk(t4[2], t4[3]);
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) { acc += u[a].x - u[a].y; z[a] = u[a] * (MODE == 2 ? 0.999f : MODE == 4 ? 0.5f : 0.04f) + (MODE == 3 || MODE == 4 ? cmk(1e-3f * a, 1e-4f) : keep[a + 8 * (it & 1)] * 1e-3f); }
        if (MODE != 3 && MODE != 4) {
#pragma unroll
            for (int a = 0; a < 56; ++a) keep[a] = keep[a] * 0.999f + u[a & 7] * 1e-6f;
        }
    }
    if (MODE != 3 && MODE != 4) for (int a = 0; a < 56; ++a) acc += keep[a].x + keep[a].y;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int F32>
__global__ __launch_bounds__(256, 2) void kern_m(float* out, int iters, const f32x4* __restrict__ wsrc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) ((float*)lds)[i] = 0.001f * i;
    __syncthreads();
    const f32x4* src = (const f32x4*)lds;
    f32x16 c[12];
    for (int q = 0; q < 12; ++q) c[q] = zero16();
    for (int it = 0; it < iters; ++it) {
        const f32x4 a = src[(lane + it) & 1023];
        // B operands stream from global memory like the conv kernels' weight fragments (VMEM returns land between the MFMAs)
        f32x4 b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = wsrc ? wsrc[((size_t)(it * 4 + q) * 64 + lane) & 65535] : src[(lane * 3 + it + q) & 1023];
#pragma unroll
        for (int q = 0; q < 12; ++q)
            c[q] = F32 ? mfma32(a[q & 3], b[q & 3][q & 3], c[q])
                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b[q & 3]), c[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 12; ++q) for (int i = 0; i < 16; ++i) s += c[q][i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int blocks = 512, n = blocks * 256;
    float *df, *dm; hipMalloc(&df, n * 4); hipMalloc(&dm, 4096 * 256 * 4);
    f32x4* dw; hipMalloc(&dw, 65536 * 16); hipMemset(dw, 0x3c, 65536 * 16);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    std::vector<float> r0(n), r1(n);
    auto test = [&](auto kf, const char* name, bool f32 = false) {
        hipLaunchKernelGGL(kf, dim3(blocks), dim3(256), 67840, s1, df, 2000);
        hipDeviceSynchronize();
        hipMemcpy(r0.data(), df, n * 4, hipMemcpyDeviceToHost);
        for (int rep = 0; rep < 4; ++rep) {
            hipMemsetAsync(df, 0, n * 4, s1);
            hipDeviceSynchronize();
            if (rep && !f32) hipLaunchKernelGGL(kern_m<0>, dim3(4096), dim3(256), 78848, s2, dm, 3000, (const f32x4*)dw);
            if (rep && f32) hipLaunchKernelGGL(kern_m<1>, dim3(4096), dim3(256), 78848, s2, dm, 1500, (const f32x4*)dw);
            hipLaunchKernelGGL(kf, dim3(blocks), dim3(256), 67840, s1, df, 2000);
            hipDeviceSynchronize();
            hipMemcpy(r1.data(), df, n * 4, hipMemcpyDeviceToHost);
            long d = 0;
            for (int i = 0; i < n; ++i) d += memcmp(&r0[i], &r1[i], 4) != 0;
            printf("%-28s %s: outputs differing from the solo run: %ld of %d\n", name, !rep ? "alone again" : f32 ? "next to the fp32-MFMA kernel" : "next to the bf16-MFMA kernel", d, n);
        }
    };
    test(kern_f<1>, "packed VALU part only");
    test(kern_f<2>, "LDS transposes only");
    test(kern_f<3>, "packed VALU, few registers");
    test(kern_f<3>, "packed VALU, few registers", true);
    test(kern_f<4>, "plain fp32 VALU");
    return 0;
}
