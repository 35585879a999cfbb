// Can the fp32-grade CNN run on f16 MFMA with TWO terms per operand instead of three bf16 terms?  (round 5)
//   x = hi + lo,  hi = f16(x), lo = f16(x - hi): 11 + 11 significand bits and lo's sign = 23 bits in the worst case, all 24 in
//   ~75 % of the values (the residual is a multiple of ulp32(x) of magnitude <= 4096 ulp32; f16 holds every such integer up to
//   2048 and every even one up to 4096).  All FOUR products hh + hl + lh + ll are formed: 4 MFMAs instead of bf16x6's 6.
// Three questions, answered on the GPU:
//   1. does v_mfma_f32_32x32x16_f16 honour f16 subnormal inputs?            (the lo terms of small values are subnormal)
//   2. shader clock / rate under sustained f16 MFMA load on random operands vs bf16 (the power envelope sets the K-loop rate)
//   3. the error of a K = 576 dot product (conv4's shape) against float64:  fp32 MFMA | bf16x6 | f16x4 | f16x3, operands scaled
//      by powers of two into f16's range the way the kernel does (max -> [2^14, 2^15)), and f16x4 unscaled
//   hipcc --offload-arch=gfx950 -O3 -o ab_libs/f16probe tools/micro/f16probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 hfx8 __attribute__((ext_vector_type(8)));

// ---- 2. clock under load ---------------------------------------------------------------------------------------------------
template <int F16>
__global__ __launch_bounds__(256, 2) void clk_kern(const f32x4* __restrict__ src, float* out, unsigned long long* clk, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    f32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 65535]; b[i] = src[(tid * 8 + 4 + i) & 65535]; }
    f32x16 c[4];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) c[q][i] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                c[q] = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hfx8, a[i]), __builtin_bit_cast(hfx8, b[(i + q) & 3]), c[q], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, a[i]), __builtin_bit_cast(bfx8, b[(i + q) & 3]), c[q], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += c[q][i];
    out[tid] = s;
    if ((threadIdx.x & 63) == 0) { clk[(tid >> 6) * 2] = t1 - t0; clk[(tid >> 6) * 2 + 1] = r1 - r0; }
}

// ---- 1. + 3. one wave, C[32][32] = sum over products of term planes ----------------------------------------------------------
// planes: A terms [TA][32][K] and B terms [TB][K][32] as 16-bit patterns; products (i, j) with i + j <= ORDER, smallest first per
// K-step of 16 (mma_terms' order)
template <int F16>
__global__ void dot_kern(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, int K, int TA, int TB, int order, float* C) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int o = order; o >= 0; --o)
            for (int ta = o; ta >= 0; --ta) {
                const int tb = o - ta;
                if (ta >= TA || tb >= TB) continue;
                unsigned short av[8], bv[8];
                for (int e = 0; e < 8; ++e) {
                    av[e] = A[((size_t)ta * 32 + i) * K + k0 + 8 * h + e];
                    bv[e] = B[((size_t)tb * K + k0 + 8 * h + e) * 32 + i];
                }
                f32x4 a4, b4;
                __builtin_memcpy(&a4, av, 16);
                __builtin_memcpy(&b4, bv, 16);
                acc = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hfx8, a4), __builtin_bit_cast(hfx8, b4), acc, 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, a4), __builtin_bit_cast(bfx8, b4), acc, 0, 0, 0);
            }
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}
__global__ void dot_f32_kern(const float* __restrict__ A, const float* __restrict__ B, int K, float* C) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)i * K + k + h], B[(size_t)(k + h) * 32 + i], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

static unsigned short bf16_rn(float v) { unsigned u; __builtin_memcpy(&u, &v, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf16_f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }
static unsigned short f16_rn(float v) { _Float16 hv = (_Float16)v; unsigned short b; __builtin_memcpy(&b, &hv, 2); return b; }
static float f16_f(unsigned short b) { _Float16 hv; __builtin_memcpy(&hv, &b, 2); return (float)hv; }

// x -> T terms, round to nearest at each step; f16: x is scaled by `scale` (a power of two) first
static void split(const std::vector<float>& x, int T, bool f16, float scale, std::vector<unsigned short>& out) {
    const size_t n = x.size();
    out.assign(n * T, 0);
    for (size_t q = 0; q < n; ++q) {
        float r = x[q] * scale;
        for (int t = 0; t < T; ++t) {
            const unsigned short b = f16 ? f16_rn(r) : bf16_rn(r);
            out[t * n + q] = b;
            r -= f16 ? f16_f(b) : bf16_f(b);
        }
    }
}
static float pow2_to(float mx, int target) { int e; frexpf(mx, &e); return ldexpf(1.f, target - e); }   // max -> [2^(target-1), 2^target)

int main() {
    // ---- 1. subnormal inputs
    {
        const int K = 16;
        std::vector<float> a(32 * K, ldexpf(1.f, -20)), b(K * 32, 1024.f);
        std::vector<unsigned short> ta, tb;
        split(a, 1, true, 1.f, ta); split(b, 1, true, 1.f, tb);
        unsigned short *dA, *dB; float* dC;
        hipMalloc(&dA, ta.size() * 2); hipMalloc(&dB, tb.size() * 2); hipMalloc(&dC, 4096);
        hipMemcpy(dA, ta.data(), ta.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, tb.data(), tb.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(dot_kern<1>, dim3(1), dim3(64), 0, 0, dA, dB, K, 1, 1, 0, dC);
        float c[1024]; hipMemcpy(c, dC, 4096, hipMemcpyDeviceToHost);
        printf("subnormal f16 inputs: A = 2^-20 (f16 bits 0x%04x), B = 1024, K = 16: got %.9g, expected %.9g (0 = inputs flushed)\n", ta[0], c[0], 16 * ldexp(1.0, -10));
    }
    // ---- 2. clocks
    {
        const int blocks = 512 * 8, iters = 20000;
        f32x4* src; float* out; unsigned long long* clk;
        hipMalloc(&src, 65536 * 16); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 4 * 16);
        std::vector<unsigned> h(65536 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep)
            for (int mode = 0; mode < 4; ++mode) {
                const bool f16 = mode & 1, rnd = mode >> 1;
                for (auto& v : h) {
                    if (!rnd) v = 0;
                    else {
                        unsigned r = (unsigned)rand() ^ ((unsigned)rand() << 16);
                        // random sign and mantissa, exponents around 1 (bf16: 0x3f00 | 7 mantissa bits; f16: 0x3800..0x3fff)
                        v = f16 ? ((r & 0x83ff83ffu) | 0x38003800u | ((r >> 3) & 0x04000400u))
                                : ((r & 0x807f807fu) | 0x3f003f00u | ((r >> 3) & 0x00800080u));
                    }
                }
                hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
                hipEventRecord(e0);
                if (f16) hipLaunchKernelGGL(clk_kern<1>, dim3(blocks), dim3(256), 0, 0, src, out, clk, iters);
                else hipLaunchKernelGGL(clk_kern<0>, dim3(blocks), dim3(256), 0, 0, src, out, clk, iters);
                hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> c(blocks * 4 * 2);
                hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
                double st = 0, rt = 0; for (int w = 0; w < blocks * 4; ++w) { st += c[2 * w]; rt += c[2 * w + 1]; }
                printf("%-5s %-7s wall %8.3f ms  %7.1f TFLOP/s  cycles/MFMA/wave %6.2f  shader clock %7.1f MHz\n", f16 ? "f16" : "bf16", rnd ? "random" : "zeros",
                       ms, (double)blocks * 4 * iters * 16 * 32768.0 / ms * 1e-9, st / (blocks * 4) / (iters * 16.0), st / rt * 100.0);
            }
    }
    // ---- 3. dot-product error against float64
    {
        const int K = 576, NTRIAL = 24;
        std::mt19937 rng(12345);
        std::normal_distribution<float> nd(0.f, 1.f);
        const char* names[6] = {"fp32 MFMA 32x32x2", "bf16x6 (3 terms, 6 products)", "f16x4 scaled (2 terms, 4 products)", "f16x3 scaled (ll dropped)", "f16x4 unscaled",
                                "bf16x3 (2 terms, 3 products)"};
        double maxe[6] = {0}, sse[6] = {0}, ref_ss = 0;
        size_t cnt = 0;
        unsigned short *dA, *dB; float *dC, *dAf, *dBf;
        hipMalloc(&dA, 3 * 32 * K * 2); hipMalloc(&dB, 3 * K * 32 * 2); hipMalloc(&dC, 4096); hipMalloc(&dAf, 32 * K * 4); hipMalloc(&dBf, K * 32 * 4);
        for (int trial = 0; trial < NTRIAL; ++trial) {
            std::vector<float> a(32 * K), b(K * 32);
            // activations: post-ReLU-like (half zeros), log-normal magnitudes over ~3 decades; weights: N(0, 0.05) with a few large ones
            const float amag = trial % 3 == 0 ? 1.f : trial % 3 == 1 ? 40.f : 0.02f;
            for (auto& v : a) { const float g = nd(rng); v = g > 0 ? amag * expf(1.5f * nd(rng)) * g : 0.f; }
            for (auto& v : b) v = 0.05f * nd(rng) * (rng() % 50 == 0 ? 8.f : 1.f);
            std::vector<double> ref(1024, 0.0);
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)a[i * K + k] * (double)b[k * 32 + j]; ref[i * 32 + j] = s; }
            float amax = 0, bmax = 0;
            for (float v : a) amax = fmaxf(amax, fabsf(v));
            for (float v : b) bmax = fmaxf(bmax, fabsf(v));
            const float sa = pow2_to(amax, 15), sb = pow2_to(bmax, 15);
            float c[1024];
            for (int m = 0; m < 6; ++m) {
                float post = 1.f;
                if (m == 0) {
                    hipMemcpy(dAf, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dBf, b.data(), b.size() * 4, hipMemcpyHostToDevice);
                    hipLaunchKernelGGL(dot_f32_kern, dim3(1), dim3(64), 0, 0, dAf, dBf, K, dC);
                } else {
                    const bool f16 = m >= 2 && m <= 4;
                    const int T = m == 1 ? 3 : 2, order = m == 1 ? 2 : m == 3 || m == 5 ? 1 : 2;
                    const float s1 = (m == 2 || m == 3) ? sa : 1.f, s2 = (m == 2 || m == 3) ? sb : 1.f;
                    post = 1.f / (s1 * s2);
                    std::vector<unsigned short> ta, tb;
                    split(a, T, f16, s1, ta); split(b, T, f16, s2, tb);
                    hipMemcpy(dA, ta.data(), ta.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, tb.data(), tb.size() * 2, hipMemcpyHostToDevice);
                    if (f16) hipLaunchKernelGGL(dot_kern<1>, dim3(1), dim3(64), 0, 0, dA, dB, K, T, T, order, dC);
                    else hipLaunchKernelGGL(dot_kern<0>, dim3(1), dim3(64), 0, 0, dA, dB, K, T, T, order, dC);
                }
                hipMemcpy(c, dC, 4096, hipMemcpyDeviceToHost);
                for (int q = 0; q < 1024; ++q) {
                    const double e = (double)c[q] * post - ref[q];
                    // errors in units of the row's scale: sqrt(sum a^2 b^2) would be the natural unit; use |ref| rms per trial instead
                    maxe[m] = fmax(maxe[m], fabs(e) / (amag * 1.0));
                    sse[m] += e * e / ((double)amag * amag);
                }
            }
            for (int q = 0; q < 1024; ++q) ref_ss += ref[q] * ref[q] / ((double)amag * amag);
            cnt += 1024;
        }
        printf("K = %d dot products, %zu outputs; errors against float64 in units of the activation scale (rms |ref| = %.4g)\n", K, cnt, sqrt(ref_ss / cnt));
        for (int m = 0; m < 6; ++m) printf("  %-36s rms %.4g  max %.4g  (rms / fp32-MFMA rms = %.3f)\n", names[m], sqrt(sse[m] / cnt), maxe[m], sqrt(sse[m] / sse[0]));
    }
    return 0;
}
