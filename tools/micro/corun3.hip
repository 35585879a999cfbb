// Root-cause probe for DESIGN.md 7.1 ("fp32 VALU results of one kernel differ while bf16-MFMA waves of ANOTHER kernel
// run next to it"):  kernel F = plain fp32 VALU chains (no LDS, no packed math, no transcendental), kernel M = a matrix /
// vector spam variant on a second stream.  For every F wave the probe records WHERE it ran (XCC, SE, CU, SIMD) and WHEN
// (s_memrealtime), the same for every M wave, and then reports
//   * how many F outputs differ from the solo run, and HOW (ULP distance histogram, sample pairs),
//   * whether the differing waves shared a SIMD / a CU with an M wave at the time, or nothing at all,
//   * the same for several M variants (bf16 32x32x16 back to back, half duty, 16x16x32, f16, fp32 MFMA, VALU only).
//   hipcc --offload-arch=gfx950 -O3 -o ab_libs/corun3 tools/micro/corun3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 hfx8 __attribute__((ext_vector_type(8)));

struct wave_rec { unsigned hwid, xcc; unsigned long long t0, t1; };

__device__ __forceinline__ void stamp(wave_rec* r, bool first) {
    if ((threadIdx.x & 63) == 0) {
        if (first) {
            r->hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID, all 32 bits
            r->xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
            r->t0 = wall_clock64();
        } else
            r->t1 = wall_clock64();
    }
}

__global__ __launch_bounds__(256, 2) void kern_f(float* out, wave_rec* rec, int iters) {
    const int lane = threadIdx.x & 63;
    wave_rec* r = rec + (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    stamp(r, true);
    float f[16];
    float acc = 0.f;
#ifndef F_VARIANT
#pragma unroll
    for (int a = 0; a < 16; ++a) f[a] = 0.001f * (lane + 64 * a) - 0.03f * a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rr = 0; rr < 6; ++rr)
#pragma unroll
            for (int a = 0; a < 16; ++a) f[a] = fmaf(f[(a + 5) & 15], 0.37f, f[a] * 0.61f) + 0.01f * (float)(a & 7);
#pragma unroll
        for (int a = 0; a < 16; ++a) acc += f[a];
    }
#else
    // -DF_VARIANT: the arithmetic of tools/micro/corun2.hip kern_f<4> (constants from v_cos_f32 at kernel start, state
    // re-seeded through z every iteration)
    float zx[8], zy[8], tb[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) { zx[a] = 0.001f * (lane + 64 * a); zy[a] = 0.5f - 0.002f * lane; tb[a] = F_VARIANT == 2 ? 1.f - 0.0002f * lane * a : __cosf(0.02f * lane * a); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 8; ++a) { f[2 * a] = zx[a]; f[2 * a + 1] = zy[a]; }
#pragma unroll
        for (int rr = 0; rr < 6; ++rr)
#pragma unroll
            for (int a = 0; a < 16; ++a) f[a] = fmaf(f[(a + 5) & 15], 0.37f, f[a] * 0.61f) + 0.01f * tb[a & 7];
#pragma unroll
        for (int a = 0; a < 8; ++a) { acc += f[2 * a] - f[2 * a + 1]; zx[a] = f[2 * a] * 0.5f + 1e-3f * a; zy[a] = f[2 * a + 1] * 0.5f + 1e-4f; }
    }
#endif
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
    stamp(r, false);
}

// M variants: 0 bf16 32x32x16 back to back, 1 the same at ~half duty (s_nop padding), 2 bf16 16x16x32, 3 f16 32x32x16,
// 4 fp32 32x32x2, 5 VALU only (no matrix instruction); with wsrc != nullptr the B operands stream from GLOBAL memory
// (like the conv kernels' weight fragments: VMEM returns land between the MFMAs); 6 = 0 with the loads waited for
// (s_waitcnt vmcnt(0)) before the MFMAs issue
template <int V>
__global__ __launch_bounds__(256, 2) void kern_m(float* out, wave_rec* rec, int iters, const f32x4* __restrict__ wsrc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    wave_rec* r = rec + (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    stamp(r, true);
    for (int i = threadIdx.x; i < 4096; i += 256) ((float*)lds)[i] = 0.001f * i;
    __syncthreads();
    const f32x4* src = (const f32x4*)lds;
    f32x16 c[8];
    f32x4 c4[8];
    for (int q = 0; q < 8; ++q) { for (int i = 0; i < 16; ++i) c[q][i] = 0.f; c4[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = lane + q;
    for (int it = 0; it < iters; ++it) {
        const f32x4 a = src[(lane + it) & 1023];
        f32x4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = wsrc ? wsrc[((size_t)(it * 4 + q) * 64 + lane) & 65535] : src[(lane * 3 + it + q) & 1023];
        if (V == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 b = bq[q & 3];
            if (V == 0 || V == 1 || V == 6) c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b), c[q], 0, 0, 0);
            if (V == 1) { asm volatile("s_nop 15"); asm volatile("s_nop 15"); }
            if (V == 2) c4[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b), c4[q], 0, 0, 0);
            if (V == 3) c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hfx8, a), __builtin_bit_cast(hfx8, b), c[q], 0, 0, 0);
            if (V == 4) c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 3], b[q & 3], c[q], 0, 0, 0);
            if (V == 5) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], 1.0001f, a[k & 3]);
            }
        }
    }
    float s = 0.f;
    for (int q = 0; q < 8; ++q) { for (int i = 0; i < 16; ++i) s += c[q][i]; s += c4[q][0] + c4[q][3] + v[q]; }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    stamp(r, false);
}

static unsigned simd_key(const wave_rec& w) {            // gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    return ((w.xcc & 0xf) << 16) | (w.hwid & 0xff30 & ~0xc0u);
}
static unsigned cu_key(const wave_rec& w) { return ((w.xcc & 0xf) << 16) | (w.hwid & 0xff00); }

int main(int argc, char** argv) {
    const int fblocks = 512, n = fblocks * 256, mblocks_max = 4096;
    const int fiters = argc > 1 ? atoi(argv[1]) : 2000;
    float *df, *dm;
    wave_rec *rf, *rm;
    hipMalloc(&df, n * 4); hipMalloc(&dm, (size_t)mblocks_max * 256 * 4);
    hipMalloc(&rf, fblocks * 4 * sizeof(wave_rec)); hipMalloc(&rm, (size_t)mblocks_max * 4 * sizeof(wave_rec));
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    std::vector<float> r0(n), r1(n);
    std::vector<wave_rec> wf(fblocks * 4), wm((size_t)mblocks_max * 4);
    hipLaunchKernelGGL(kern_f, dim3(fblocks), dim3(256), 0, s1, df, rf, fiters);
    hipDeviceSynchronize();
    hipMemcpy(r0.data(), df, n * 4, hipMemcpyDeviceToHost);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)kern_f); printf("kern_f: %d regs, %zu B scratch\n", fa.numRegs, fa.localSizeBytes);
    hipFuncGetAttributes(&fa, (const void*)kern_m<0>); printf("kern_m<0>: %d regs, %zu B scratch\n", fa.numRegs, fa.localSizeBytes);

    f32x4* dw; hipMalloc(&dw, 65536 * 16); hipMemset(dw, 0x3c, 65536 * 16);
    auto run = [&](int variant, int mblocks, int miters, const char* name, bool gl = false, int flds = 0) {
        const f32x4* ws = gl ? dw : nullptr;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemsetAsync(df, 0, n * 4, s1);
            hipMemsetAsync(rm, 0, (size_t)mblocks_max * 4 * sizeof(wave_rec), s1);
            hipDeviceSynchronize();
            if (mblocks > 0) {
                switch (variant) {
                    case 0: hipLaunchKernelGGL(kern_m<0>, dim3(mblocks), dim3(256), 78848, s2, dm, rm, miters, ws); break;
                    case 1: hipLaunchKernelGGL(kern_m<1>, dim3(mblocks), dim3(256), 78848, s2, dm, rm, miters / 2, ws); break;
                    case 2: hipLaunchKernelGGL(kern_m<2>, dim3(mblocks), dim3(256), 78848, s2, dm, rm, miters * 2, ws); break;
                    case 3: hipLaunchKernelGGL(kern_m<3>, dim3(mblocks), dim3(256), 78848, s2, dm, rm, miters, ws); break;
                    case 4: hipLaunchKernelGGL(kern_m<4>, dim3(mblocks), dim3(256), 78848, s2, dm, rm, miters / 2, ws); break;
                    case 6: hipLaunchKernelGGL(kern_m<6>, dim3(mblocks), dim3(256), 78848, s2, dm, rm, miters, ws); break;
                    case 5: hipLaunchKernelGGL(kern_m<5>, dim3(mblocks), dim3(256), 78848, s2, dm, rm, miters / 2, ws); break;
                }
            }
            hipLaunchKernelGGL(kern_f, dim3(fblocks), dim3(256), flds, s1, df, rf, fiters);
            hipDeviceSynchronize();
            hipMemcpy(r1.data(), df, n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(wf.data(), rf, wf.size() * sizeof(wave_rec), hipMemcpyDeviceToHost);
            hipMemcpy(wm.data(), rm, wm.size() * sizeof(wave_rec), hipMemcpyDeviceToHost);
            // co-residency: M waves per SIMD / CU as (t0, t1) intervals
            std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> by_simd, by_cu;
            for (int i = 0; i < mblocks * 4; ++i) if (wm[i].t1) { by_simd[simd_key(wm[i])].push_back({wm[i].t0, wm[i].t1}); by_cu[cu_key(wm[i])].push_back({wm[i].t0, wm[i].t1}); }
            auto overlaps = [](const std::vector<std::pair<unsigned long long, unsigned long long>>* v, const wave_rec& w) {
                if (!v) return false;
                for (auto& p : *v) if (p.first < w.t1 && w.t0 < p.second) return true;
                return false;
            };
            long waves[3] = {0, 0, 0}, bad_waves[3] = {0, 0, 0}, bad_threads = 0;
            long ulp_hist[6] = {0, 0, 0, 0, 0, 0};   // 1, 2-3, 4-15, 16-255, 256-65535, more
            int shown = 0;
            for (int w = 0; w < fblocks * 4; ++w) {
                auto is = by_simd.find(simd_key(wf[w])); auto ic = by_cu.find(cu_key(wf[w]));
                const int cls = overlaps(is == by_simd.end() ? nullptr : &is->second, wf[w]) ? 0 : overlaps(ic == by_cu.end() ? nullptr : &ic->second, wf[w]) ? 1 : 2;
                ++waves[cls];
                bool bad = false;
                for (int l = 0; l < 64; ++l) {
                    const int i = w * 64 + l;
                    unsigned a, b; memcpy(&a, &r0[i], 4); memcpy(&b, &r1[i], 4);
                    if (a != b) {
                        bad = true; ++bad_threads;
                        const long d = labs((long)(int)a - (long)(int)b);
                        ++ulp_hist[d <= 1 ? 0 : d <= 3 ? 1 : d <= 15 ? 2 : d <= 255 ? 3 : d <= 65535 ? 4 : 5];
                        if (shown < 4 && rep == 1) { printf("    thread %d: solo %08x (%g)  co-run %08x (%g)\n", i, a, r0[i], b, r1[i]); ++shown; }
                    }
                }
                if (bad) ++bad_waves[cls];
            }
            printf("%-34s M blocks %4d rep %d: differing threads %6ld | F waves sharing a SIMD with M: %ld of %ld bad, same CU other SIMD: %ld of %ld, no M on the CU: %ld of %ld | ulp 1:%ld 2-3:%ld 4-15:%ld 16-255:%ld <65536:%ld more:%ld\n",
                   name, mblocks, rep, bad_threads, bad_waves[0], waves[0], bad_waves[1], waves[1], bad_waves[2], waves[2],
                   ulp_hist[0], ulp_hist[1], ulp_hist[2], ulp_hist[3], ulp_hist[4], ulp_hist[5]);
        }
    };
    run(0, 0, 0, "F alone again");
    run(0, 4096, 3000, "bf16 32x32x16 MFMA, whole chip");
    run(0, 128, 60000, "bf16 32x32x16 MFMA, 128 blocks");
    run(1, 4096, 3000, "bf16 32x32x16 half duty");
    run(2, 4096, 3000, "bf16 16x16x32 MFMA");
    run(3, 4096, 3000, "f16 32x32x16 MFMA");
    run(4, 4096, 3000, "fp32 32x32x2 MFMA");
    run(5, 4096, 3000, "VALU only");
    printf("-- B operands of M streamed from global memory (L2-resident 1 MiB), as in tools/micro/corun2.hip\n");
    run(0, 4096, 3000, "bf16 32x32x16 + global B loads", true);
    run(0, 128, 60000, "bf16 + global loads, 128 blocks", true);
    run(6, 4096, 3000, "bf16 + global loads, vmcnt(0)", true);
    run(4, 4096, 3000, "fp32 MFMA + global B loads", true);
    run(5, 4096, 3000, "VALU only + global loads", true);
    run(2, 4096, 3000, "bf16 16x16x32 + global loads", true);
    printf("-- F with 67840 B of (unused) dynamic LDS: one F and one M workgroup per CU\n");
    run(0, 4096, 3000, "bf16 32x32x16, F with LDS", false, 67840);
    run(0, 4096, 3000, "bf16 + global loads, F with LDS", true, 67840);
    return 0;
}
