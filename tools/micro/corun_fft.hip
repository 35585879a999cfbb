// Follow-up of tools/probe_concurrency.py: the 512-point FFT of mel_frame_kernel (packed-f32 VALU + two wave-private LDS
// transposes, taken from csrc/mel.hip as is) on waves 4-7 of each workgroup, next to waves 0-3 that spam split-bf16 style
// MFMAs (v_mfma_f32_32x32x16_bf16) fed by ds_read_b128 -- or fp32 MFMAs -- on the same SIMDs.  The FFT waves' outputs are
// compared bit for bit between an idle and a busy neighbourhood.
#include "../../nisqa_amd/csrc/mel.hip"
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void corun_fft(float* out, int iters, int busy) {
    __shared__ __attribute__((aligned(16))) char lds[4 * MEL_EXCH_BYTES + 16384];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!busy) return;
        const f32x4* src = (const f32x4*)(lds + 4 * MEL_EXCH_BYTES);
        for (int i = threadIdx.x; i < 4096; i += 256) ((float*)(lds + 4 * MEL_EXCH_BYTES))[i] = 0.001f * i;
        f32x16 c0 = zero16(), c1 = zero16();
        for (int it = 0; it < iters * 8; ++it) {
            const f32x4 a = src[(lane + it) & 1023], b = src[(lane * 3 + it) & 1023];
            if (busy == 1) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, b), __builtin_bit_cast(bfx8, a), c1, 0, 0, 0);
            } else {
                c0 = mfma32(a[0], b[0], c0); c1 = mfma32(a[1], b[1], c1);
                c0 = mfma32(a[2], b[2], c0); c1 = mfma32(a[3], b[3], c1);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
        if (s == 12345.f) out[0] = s;
        return;
    }
    char* exch = lds + (wave - 4) * MEL_EXCH_BYTES;
    mel_twiddles tw;
    for (int r = 0; r < 4; ++r) { tw.a[r] = cmk(__cosf(0.01f * lane * r), -__sinf(0.01f * lane * r)); tw.d[r] = tw.a[r]; }
    for (int p = 0; p < 8; ++p) { tw.b[p] = cmk(__cosf(0.02f * lane * p), -__sinf(0.02f * lane * p)); tw.c[p] = cmk(__cosf(0.3f * (lane & 7) * p), -__sinf(0.3f * (lane & 7) * p)); }
    c32 z[8], u[8];
    for (int a = 0; a < 8; ++a) z[a] = cmk(0.001f * (lane + 64 * a), 0.5f - 0.002f * lane);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        fft512<1>(u, z, tw, exch, lane);
        for (int a = 0; a < 8; ++a) { acc += u[a].x - u[a].y; z[a] = u[a] * 0.04f + cmk(0.001f * a, 0.f); }
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
    const int blocks = 512, n = blocks * 512;
    float* d; hipMalloc(&d, n * 4);
    std::vector<float> r0(n), r1(n), r2(n), r3(n);
    auto run = [&](int busy, std::vector<float>& r) {
        hipMemset(d, 0, n * 4);
        hipLaunchKernelGGL(corun_fft, dim3(blocks), dim3(512), 0, 0, d, 3000, busy);
        hipDeviceSynchronize();
        hipMemcpy(r.data(), d, n * 4, hipMemcpyDeviceToHost);
    };
    run(0, r0); run(0, r1); run(1, r2); run(2, r3);
    long d01 = 0, d02 = 0, d03 = 0;
    for (int i = 0; i < n; ++i) {
        if ((i & 511) < 256) continue;
        d01 += memcmp(&r0[i], &r1[i], 4) != 0; d02 += memcmp(&r0[i], &r2[i], 4) != 0; d03 += memcmp(&r0[i], &r3[i], 4) != 0;
    }
    printf("FFT-wave results differing: idle vs idle %ld, idle vs bf16-MFMA neighbours %ld, idle vs fp32-MFMA neighbours %ld (of %d)\n", d01, d02, d03, n / 2);
    return 0;
}
