// Operators of the training step (include/nisqa_train.h; SURVEY.md section 8f-3, BASELINE config 5): what
// model.train(); model(x, n_wins); loss.backward(); opt.step() executes in the reference
// (nisqa/NISQA_model.py:131-152, NISQA_lib.py:688-710, 988-1040, 1171-1183, 1880-1950).
//
// Train-mode BatchNorm couples every valid segment of the batch between a convolution and its activation, so the
// wave-owns-a-segment fusion of the inference kernels does not apply: activations live in HBM ([S][H*W][C], channels
// contiguous) and a layer is a short sequence of grid-wide kernels.  conv1 reads the spectrogram directly, conv2..6
// are implicit GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32) whose loaders gather the 3x3 patches; reductions that
// feed normalisation statistics accumulate in float64.  The same GEMM core serves every Linear, the attention
// products (ragged, one group per clip) and all their gradients (transposed operands; split-K with atomics where K
// is the row count of the batch).
#include "common.hpp"
#include "conv_bf16.hpp"
#include "../../include/nisqa_hip.h"
#include "../../include/nisqa_train.h"

#define GB_K 32

struct gemm_one { int64_t v[10]; };                       // the descriptor of a single-group call, passed by value

// One operand tile [R rows][GB_K] from global memory into registers, then into LDS as T[k][r] (k-major, so that a
// lane's MFMA fragment element -- row/col = lane & 31, k = lane >> 5 -- is a conflict-free ds_read_b32).
//   KC = true : k is the contiguous index in memory, element (r, k) at g[r * ld + k]   (A as stored, B = a weight [N][K])
//   KC = false: r is the contiguous index,           element (r, k) at g[k * ld + r]   (A stored [K][M], B stored [K][N])
// Each thread owns NV groups of four consecutive elements along the contiguous index: one 128-bit load when the
// group is aligned and fully inside the matrix, four guarded scalar loads otherwise.
template <int R, bool KC, int NTH>
struct tile_loader {
    static constexpr int NV = R * GB_K / 4 / NTH;
    static_assert(NV >= 1 && NV * NTH * 4 == R * GB_K, "tile must split evenly over the workgroup");
    f32x4 v[NV];
    // interior tile with aligned rows: straight-line 128-bit loads, no per-element guards (wave-uniform choice)
    __device__ __forceinline__ void load_fast(const float* __restrict__ g, int64_t ld, int r0, int k0, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + NTH * j;
            if (KC) v[j] = *(const f32x4*)(g + (int64_t)(r0 + i / (GB_K / 4)) * ld + k0 + 4 * (i % (GB_K / 4)));
            else v[j] = *(const f32x4*)(g + (int64_t)(k0 + i / (R / 4)) * ld + r0 + 4 * (i % (R / 4)));
        }
    }
    __device__ __forceinline__ void load(const float* __restrict__ g, int64_t ld, int r0, int r_lim, int k0, int k_lim, bool vec_ok, int tid) {
        if (vec_ok && r0 + R <= r_lim && k0 + GB_K <= k_lim) { load_fast(g, ld, r0, k0, tid); return; }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + NTH * j;
            int r, k;
            if (KC) { r = i / (GB_K / 4); k = 4 * (i % (GB_K / 4)); } else { k = i / (R / 4); r = 4 * (i % (R / 4)); }
            const int gr = r0 + r, gk = k0 + k;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (KC) {
                if (gr < r_lim) {
                    const float* p = g + (int64_t)gr * ld + gk;
                    if (vec_ok && gk + 3 < k_lim) x = *(const f32x4*)p;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (gk + e < k_lim) x[e] = p[e];
                    }
                }
            } else {
                if (gk < k_lim) {
                    const float* p = g + (int64_t)gk * ld + gr;
                    if (vec_ok && gr + 3 < r_lim) x = *(const f32x4*)p;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (gr + e < r_lim) x[e] = p[e];
                    }
                }
            }
            v[j] = x;
        }
    }
    __device__ __forceinline__ void store(float (*T)[R + 4], int tid) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + NTH * j;
            if (KC) {
                const int r = i / (GB_K / 4), k = 4 * (i % (GB_K / 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) T[k + e][r] = v[j][e];
            } else {
                const int k = i / (R / 4), r = 4 * (i % (R / 4));
                *(f32x4*)&T[k][r] = v[j];
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------------
// grouped GEMM: 256 threads = 4 waves, one 32 x 32 MFMA block each (2 x 2 waves on a 64 x 64 tile, or 4 x 1 on a
// 128 x 32 tile for narrow outputs); the next K-tile travels global -> registers while the current one is multiplied
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN, int MT, int NT, bool TA, bool TB>
__global__ __launch_bounds__((BM / (32 * MT)) * (BN / (32 * NT)) * 64) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, const int64_t* __restrict__ desc,
                                                       gemm_one one, int n_groups, int ksplit, float alpha,
                                                       const float* __restrict__ bias, int relu) {
    __shared__ __attribute__((aligned(16))) float As[GB_K][BM + 4];
    __shared__ __attribute__((aligned(16))) float Bs[GB_K][BN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int lo = 0, hi = n_groups;                            // group of this tile: desc[g][9] <= tile < desc[g+1][9]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (desc[(int64_t)mid * 10 + 9] <= (int64_t)blockIdx.x) lo = mid; else hi = mid;
    }
    const int64_t* d = n_groups > 0 ? desc + (int64_t)lo * 10 : one.v;
    const float* Ag = A + d[0];
    const float* Bg = B + d[1];
    float* Cg = C + d[2];
    const int M = (int)d[3], N = (int)d[4], K = (int)d[5];
    const int64_t lda = d[6], ldb = d[7], ldc = d[8];
    const int t = (int)((int64_t)blockIdx.x - d[9]);
    const int tiles_n = (N + BN - 1) / BN;
    const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
    int kc = (K + ksplit - 1) / ksplit;
    kc = (kc + GB_K - 1) / GB_K * GB_K;
    const int k_begin = blockIdx.y * kc, k_end = min(K, k_begin + kc);
    if (k_begin >= k_end) return;
    const bool va = ((((uintptr_t)Ag) & 15) == 0) && ((lda & 3) == 0);
    const bool vb = ((((uintptr_t)Bg) & 15) == 0) && ((ldb & 3) == 0);
    constexpr int WN = BN / (32 * NT);                     // waves along N; each wave owns MT x NT blocks of 32 x 32
    constexpr int NTH = (BM / (32 * MT)) * WN * 64;        // 4 or 8 waves
    const int wm = (wave / WN) * 32 * MT, wn = (wave % WN) * 32 * NT;
    tile_loader<BM, !TA, NTH> la;
    tile_loader<BN, TB, NTH> lb;
    la.load(Ag, lda, m0, M, k_begin, k_end, va, tid);
    lb.load(Bg, ldb, n0, N, k_begin, k_end, vb, tid);
    la.store(As, tid);
    lb.store(Bs, tid);
    __syncthreads();
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero16();
    for (int k0 = k_begin; k0 < k_end; k0 += GB_K) {
        const bool more = k0 + GB_K < k_end;
        if (more) {
            la.load(Ag, lda, m0, M, k0 + GB_K, k_end, va, tid);
            lb.load(Bg, ldb, n0, N, k0 + GB_K, k_end, vb, tid);
        }
#pragma unroll
        for (int s = 0; s < GB_K / 2; ++s) {
            float av[MT], bv[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = As[2 * s + (lane >> 5)][wm + 32 * i + (lane & 31)];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = Bs[2 * s + (lane >> 5)][wn + 32 * j + (lane & 31)];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma32(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
        if (more) {
            la.store(As, tid);
            lb.store(Bs, tid);
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn + 32 * j + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + NQ_DROW(r, lane >> 5);
                if (row < M && col < N) {
                    float v = alpha * acc[i][j][r];
                    if (ksplit > 1) atomicAdd(Cg + (int64_t)row * ldc + col, v);
                    else {
                        if (bias) v += bias[col];
                        if (relu) v = fmaxf(v, 0.f);
                        Cg[(int64_t)row * ldc + col] = v;
                    }
                }
            }
        }
}

template <int BM, int BN, int MT, int NT>
static void gemm_launch(dim3 grid, hipStream_t st, const float* a, const float* b, float* c, const int64_t* desc, gemm_one one,
                        int n_groups, int ta, int tb, int ksplit, float alpha, const float* bias = nullptr, int relu = 0) {
    if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, MT, NT, false, false>), grid, dim3((BM / (32 * MT)) * (BN / (32 * NT)) * 64), 0, st, a, b, c, desc, one, n_groups, ksplit, alpha, bias, relu);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, MT, NT, false, true>), grid, dim3((BM / (32 * MT)) * (BN / (32 * NT)) * 64), 0, st, a, b, c, desc, one, n_groups, ksplit, alpha, bias, relu);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, MT, NT, true, false>), grid, dim3((BM / (32 * MT)) * (BN / (32 * NT)) * 64), 0, st, a, b, c, desc, one, n_groups, ksplit, alpha, bias, relu);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, MT, NT, true, true>), grid, dim3((BM / (32 * MT)) * (BN / (32 * NT)) * 64), 0, st, a, b, c, desc, one, n_groups, ksplit, alpha, bias, relu);
}

extern "C" int nisqa_gemm_f32(const float* a, const float* b, float* c, const int64_t* desc, int32_t n_groups,
                              int32_t total_tiles, int32_t trans_a, int32_t trans_b, int32_t ksplit, float alpha,
                              void* stream) {
    if (!a || !b || !c || !desc || n_groups <= 0 || total_tiles < 0 || ksplit < 1 || ksplit > 65535) return NISQA_ERR_ARG;
    if (total_tiles == 0) return NISQA_OK;
    NQ_LAUNCH_BEGIN();
    gemm_launch<64, 64, 1, 1>(dim3(total_tiles, ksplit), (hipStream_t)stream, a, b, c, desc, gemm_one{}, n_groups, trans_a,
                              trans_b, ksplit, alpha);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_gemm_f32_one(const float* a, const float* b, float* c, int64_t m, int64_t n, int64_t k, int64_t lda,
                                  int64_t ldb, int64_t ldc, int32_t trans_a, int32_t trans_b, int32_t ksplit, float alpha,
                                  const float* bias, int32_t relu, void* stream) {
    if (!a || !b || !c || m < 0 || n < 0 || k <= 0 || ksplit < 1 || ksplit > 65535 || (ksplit > 1 && (bias || relu)))
        return NISQA_ERR_ARG;
    // tile shape by problem shape: (BM, BN) and the 32 x 32 blocks per wave (more blocks = fewer LDS reads and
    // barriers per MFMA): conv1/2 outputs are narrow, conv/Linear forward is tall (N = 64), the patch gradient is
    // tall and wide, weight gradients are short and wide with K = the row count of the batch
    int cfg = 0;
    int64_t bm = 64, bn = 64;
    if (n <= 32 && m > 64) { cfg = 1; bm = 128; bn = 32; }
    // (the big tiles only where they still give every CU a workgroup: a 7 904 x 64 product is 31 tiles of 256 x 64)
    else if (m >= 256 && n >= 128 && ((m + 127) / 128) * ((n + 127) / 128) >= 192) { cfg = 2; bm = 128; bn = 128; }
    else if (m >= 512 && n > 32 && ((m + 255) / 256) * ((n + 63) / 64) >= 192) { cfg = 3; bm = 256; bn = 64; }
    else if (m <= 64 && n >= 256) { cfg = 4; bm = 64; bn = 256; }
    const int64_t tiles = ((m + bm - 1) / bm) * ((n + bn - 1) / bn);
    if (tiles == 0) return NISQA_OK;
    if (tiles > 0x7fffffff) return NISQA_ERR_ARG;
    gemm_one one;
    const int64_t v[10] = {0, 0, 0, m, n, k, lda, ldb, ldc, 0};
    for (int i = 0; i < 10; ++i) one.v[i] = v[i];
    const dim3 grid((unsigned)tiles, ksplit);
    hipStream_t st = (hipStream_t)stream;
    NQ_LAUNCH_BEGIN();
    switch (cfg) {
        case 1: gemm_launch<128, 32, 1, 1>(grid, st, a, b, c, nullptr, one, 0, trans_a, trans_b, ksplit, alpha, bias, relu); break;
        case 2: gemm_launch<128, 128, 2, 1>(grid, st, a, b, c, nullptr, one, 0, trans_a, trans_b, ksplit, alpha, bias, relu); break;
        case 3: gemm_launch<256, 64, 2, 1>(grid, st, a, b, c, nullptr, one, 0, trans_a, trans_b, ksplit, alpha, bias, relu); break;
        case 4: gemm_launch<64, 256, 2, 1>(grid, st, a, b, c, nullptr, one, 0, trans_a, trans_b, ksplit, alpha, bias, relu); break;
        default: gemm_launch<64, 64, 1, 1>(grid, st, a, b, c, nullptr, one, 0, trans_a, trans_b, ksplit, alpha, bias, relu); break;
    }
    return NQ_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------------
// 3 x 3 convolutions as IMPLICIT GEMMs: the patch matrix of im2col is never written -- the operand loaders gather it
// from the activation tensor ([S][H*W][C], channels contiguous: a group of four k's is one 128-bit load inside one
// tap).  Same tiles, LDS staging and MFMA core as gemm_f32_kernel.
//   MODE 0 forward : z[(s,yo,xo)][co]   = sum_{tap,ci} x[s][(yo+dy-1, xo+dx-pad)][ci] * w[co][tap*CI+ci] (+ bias)
//   MODE 1 dgrad   : dx[(s,y,x)][ci]    = sum_{tap,co} dz[s][(y-dy+1, x-dx+pad)][co]   * w[co][tap*CI+ci]
//   MODE 2 wgrad   : dw[co][tap*CI+ci] += sum_{(s,yo,xo)} dz[(s,yo,xo)][co] * x[s][(yo+dy-1, xo+dx-pad)][ci]   (split-K, atomics)
// ---------------------------------------------------------------------------------------------------------
struct nq_fdiv { uint32_t m, s1, s2; };                       // q = (t + ((x - t) >> s1)) >> s2 with t = umulhi(m, x)
NQ_DEV uint32_t fdiv_q(uint32_t x, nq_fdiv d) {
    const uint32_t t = __umulhi(d.m, x);
    return (t + ((x - t) >> d.s1)) >> d.s2;
}
static nq_fdiv fdiv_make(uint32_t d) {
    nq_fdiv r = {0u, 0u, 0u};
    if (d <= 1) return r;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    r.m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    r.s1 = 1;
    r.s2 = l - 1;
    return r;
}

struct conv_geom {
    int hr, wr;            // spatial size the GEMM rows (pixels) run over
    int hs, ws, lcs;       // spatial size and log2(channels) of the gathered tensor
    int pad, sgn;          // padding along x; +1: source pixel = row pixel + (tap offset), -1: minus (dgrad)
    int ci, co;            // channels of the layer
    nq_fdiv d_img, d_row;     // division by hr*wr and by wr
};

NQ_DEV f32x4 conv_gather(const float* __restrict__ src, const conv_geom& g, int s, int yr, int xr, int tap, int c) {
    const int ty = tap / 3, tx = tap - 3 * ty;
    const int ys = yr + g.sgn * (ty - 1), xs = xr + g.sgn * (tx - g.pad);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)ys < (unsigned)g.hs && (unsigned)xs < (unsigned)g.ws)
        v = *(const f32x4*)(src + ((((int64_t)s * g.hs + ys) * g.ws + xs) << g.lcs) + c);
    return v;
}

template <int BM, int BN, int MT, int NT, int MODE>
__global__ __launch_bounds__((BM / (32 * MT)) * (BN / (32 * NT)) * 64) void conv_gemm_kernel(
    const float* __restrict__ G, const float* __restrict__ O, float* __restrict__ C, conv_geom g, int M, int N, int K, int ksplit,
    const float* __restrict__ bias, double* __restrict__ stats) {
    // G: the gathered tensor (x for forward / wgrad, dz for dgrad); O: the other operand (weights, or dz for wgrad)
    __shared__ __attribute__((aligned(16))) float As[GB_K][BM + 4];
    __shared__ __attribute__((aligned(16))) float Bs[GB_K][BN + 4];
    constexpr int WN = BN / (32 * NT);
    constexpr int NTH = (BM / (32 * MT)) * WN * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (N + BN - 1) / BN;
    const int m0 = ((int)blockIdx.x / tiles_n) * BM, n0 = ((int)blockIdx.x % tiles_n) * BN;
    int kc = (K + ksplit - 1) / ksplit;
    kc = (kc + GB_K - 1) / GB_K * GB_K;
    const int k_begin = blockIdx.y * kc, k_end = min(K, k_begin + kc);
    if (k_begin >= k_end) return;
    const int wm = (wave / WN) * 32 * MT, wn = (wave % WN) * 32 * NT;
    tile_loader<BM, MODE != 2, NTH> la;                     // forward / dgrad: A rows are pixels, k contiguous; wgrad: A = dz [K][M]
    tile_loader<BN, MODE == 0, NTH> lb;                     // forward: B = w [N][K]; dgrad / wgrad: n contiguous
    const int csm = (1 << g.lcs) - 1;

    // A rows of this thread (forward / dgrad): pixel coordinates once, they do not change along K
    int a_s[la.NV], a_y[la.NV], a_x[la.NV];
    if (MODE != 2) {
#pragma unroll
        for (int j = 0; j < la.NV; ++j) {
            const int row = m0 + (tid + NTH * j) / (GB_K / 4);
            const uint32_t s = fdiv_q((uint32_t)row, g.d_img);
            const uint32_t rem = (uint32_t)row - s * (uint32_t)(g.hr * g.wr);
            const uint32_t y = fdiv_q(rem, g.d_row);
            a_s[j] = row < M ? (int)s : -1;
            a_y[j] = (int)y;
            a_x[j] = (int)(rem - y * (uint32_t)g.wr);
        }
    }
    auto load_a = [&](int k0) {
#pragma unroll
        for (int j = 0; j < la.NV; ++j) {
            const int i = tid + NTH * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (MODE != 2) {                                 // gather, k = tap * cs + c
                const int kk = k0 + 4 * (i % (GB_K / 4));
                if (a_s[j] >= 0 && kk < k_end) v = conv_gather(G, g, a_s[j], a_y[j], a_x[j], kk >> g.lcs, kk & csm);
            } else {                                         // dz stored [K = pixel rows][M = co]
                const int kk = k0 + i / (BM / 4), m = m0 + 4 * (i % (BM / 4));
                if (kk < k_end && m < M) v = *(const f32x4*)(O + (int64_t)kk * g.co + m);
            }
            la.v[j] = v;
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int j = 0; j < lb.NV; ++j) {
            const int i = tid + NTH * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 0) {                                 // w [N = co][K = 9 ci], k contiguous
                const int n = n0 + i / (GB_K / 4), kk = k0 + 4 * (i % (GB_K / 4));
                if (n < N && kk < k_end) v = *(const f32x4*)(O + (int64_t)n * K + kk);
            } else if (MODE == 1) {                          // w viewed as [k = tap * co_n + co][n = ci]
                const int kk = k0 + i / (BN / 4), n = n0 + 4 * (i % (BN / 4));
                if (kk < k_end && n < N) {
                    const int tap = kk >> g.lcs, co = kk & csm;
                    v = *(const f32x4*)(O + (int64_t)co * (9 * g.ci) + tap * g.ci + n);
                }
            } else {                                         // gather x at pixel row kk, column n = tap * ci + c
                const int kk = k0 + i / (BN / 4), n = n0 + 4 * (i % (BN / 4));
                if (kk < k_end && n < N) {
                    const uint32_t s = fdiv_q((uint32_t)kk, g.d_img);
                    const uint32_t rem = (uint32_t)kk - s * (uint32_t)(g.hr * g.wr);
                    const uint32_t y = fdiv_q(rem, g.d_row);
                    v = conv_gather(G, g, (int)s, (int)y, (int)(rem - y * (uint32_t)g.wr), n >> g.lcs, n & csm);
                }
            }
            lb.v[j] = v;
        }
    };

    load_a(k_begin);
    load_b(k_begin);
    la.store(As, tid);
    lb.store(Bs, tid);
    __syncthreads();
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero16();
    for (int k0 = k_begin; k0 < k_end; k0 += GB_K) {
        const bool more = k0 + GB_K < k_end;
        if (more) {
            load_a(k0 + GB_K);
            load_b(k0 + GB_K);
        }
#pragma unroll
        for (int s = 0; s < GB_K / 2; ++s) {
            float av[MT], bv[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = As[2 * s + (lane >> 5)][wm + 32 * i + (lane & 31)];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = Bs[2 * s + (lane >> 5)][wn + 32 * j + (lane & 31)];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma32(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
        if (more) {
            la.store(As, tid);
            lb.store(Bs, tid);
            __syncthreads();
        }
    }
    // forward with stats != NULL: the per-channel sums of z and z^2 over all rows (the BatchNorm batch statistics) ride
    // along in float64 -- lane pairs, then the workgroup's waves through LDS, then one atomic per channel and workgroup
    __shared__ double red[MODE == 0 ? 2 * BN : 1];
    const bool with_stats = MODE == 0 && stats != nullptr;
    if (with_stats) {
        for (int q = tid; q < 2 * BN; q += NTH) red[q] = 0.0;
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn + 32 * j + (lane & 31);
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + NQ_DROW(r, lane >> 5);
                if (row < M && col < N) {
                    const float v = acc[i][j][r];
                    if (MODE == 2) atomicAdd(C + (int64_t)row * N + col, v);
                    else {
                        const float zv = bias ? v + bias[col] : v;
                        C[(int64_t)row * N + col] = zv;
                        if (MODE == 0) { s1 += (double)zv; s2 += (double)zv * (double)zv; }
                    }
                }
            }
            if (with_stats) {
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (lane < 32 && col < N) {
                    atomicAdd(&red[wn + 32 * j + lane], s1);
                    atomicAdd(&red[BN + wn + 32 * j + lane], s2);
                }
            }
        }
    if (with_stats) {
        __syncthreads();
        for (int q = tid; q < BN; q += NTH)
            if (n0 + q < N) {
                atomicAdd(stats + n0 + q, red[q]);
                atomicAdd(stats + N + n0 + q, red[BN + q]);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same three products on split-bf16 MFMA (v_mfma_f32_32x32x16_bf16; hi * hi + hi * lo + lo * hi, fp32 accumulate:
// 16 operand bits like the inference path, 5.3x the fp32-MFMA rate).  Loaders and tiling as above; what changes is
// the LDS image and how fragments come out of it.  A tile is two bf16 planes (hi, lo) written by the loader threads, which
// split their four fp32 values on the way (v_cvt_pk_bf16_f32):
//   * operand whose CONTIGUOUS index in memory is k (forward / dgrad patches, forward weights): image [row][32 k] with
//     80-byte rows; a lane's fragment (row lane & 31, k = 8 * (lane >> 5) .. + 7) is one ds_read_b128;
//   * operand whose contiguous index is the row (dgrad weights; dz and the patches of wgrad, where k is the pixel row of
//     the batch): image [32 k][rows] as loaded (row stride 2 * rows + 64 bytes: the four k-rows of a transpose read land
//     in different bank quarters), and the fragment comes out of ds_read_b64_tr_b16, gfx950's transposing LDS read:
//     within a 16-lane group lane s points at row (s >> 2), columns 4 * (s & 3) .. + 3 of a [4 k][16 rows] block and
//     lane i receives column i of that block (tools/micro/trread.hip prints the mapping from the hardware).
// ---------------------------------------------------------------------------------------------------------
typedef short nq_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned nq_u32x2 __attribute__((ext_vector_type(2)));

template <int R, bool KC>
struct bf_tile {
    static constexpr int RS = KC ? 80 : 2 * R + ((R / 32) % 2 ? 128 : 64);   // row stride in bytes: an odd multiple of 64
    static constexpr int PLANE = KC ? R * 80 : GB_K * RS;     // one bf16 plane
    static constexpr int BYTES = 2 * PLANE;
    // the loader thread's groups of four values -> both planes
    template <int NV, int NTH>
    static __device__ __forceinline__ void store(const f32x4 (&v)[NV], unsigned base, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + NTH * j;
            unsigned a;
            if (KC) a = base + (i / (GB_K / 4)) * RS + 8 * (i % (GB_K / 4));
            else a = base + (i / (R / 4)) * RS + 8 * (i % (R / 4));
            const unsigned h0 = cvt_pk_bf16(v[j][0], v[j][1]), h1 = cvt_pk_bf16(v[j][2], v[j][3]);
            const unsigned l0 = cvt_pk_bf16(v[j][0] - __uint_as_float(h0 << 16), v[j][1] - __uint_as_float(h0 & 0xffff0000u));
            const unsigned l1 = cvt_pk_bf16(v[j][2] - __uint_as_float(h1 << 16), v[j][3] - __uint_as_float(h1 & 0xffff0000u));
            *(NQ_AS3 nq_u32x2*)(a) = nq_u32x2{h0, h1};
            *(NQ_AS3 nq_u32x2*)(a + PLANE) = nq_u32x2{l0, l1};
        }
    }
    // lane-dependent part of a fragment address for the 32 rows starting at row0
    static __device__ __forceinline__ unsigned lane_base(unsigned base, int row0, int lane) {
        if (KC) return base + (row0 + (lane & 31)) * RS + 16 * (lane >> 5);
        return base + (8 * (lane >> 5) + ((lane & 15) >> 2)) * RS + 2 * (row0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
    }
    // fragment of k-substep s (k = 16 s .. 16 s + 15) of one plane
    static __device__ __forceinline__ f32x4 frag(unsigned lb, int s) {
        if (KC) return lds_ld128(lb + 32 * s);
        const unsigned a = lb + 16 * s * RS;
        const nq_s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((NQ_AS3 nq_s16x4*)(a));
        const nq_s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((NQ_AS3 nq_s16x4*)(a + 4 * RS));
        struct { nq_s16x4 a, b; } both = {x0, x1};
        return __builtin_bit_cast(f32x4, both);
    }
};

template <int BM, int BN, int MT, int NT, int MODE>
__global__ __launch_bounds__((BM / (32 * MT)) * (BN / (32 * NT)) * 64) void conv_gemm_bf16_kernel(
    const float* __restrict__ G, const float* __restrict__ O, float* __restrict__ C, conv_geom g, int M, int N, int K, int ksplit,
    const float* __restrict__ bias, double* __restrict__ stats) {
    typedef bf_tile<BM, MODE != 2> TA;
    typedef bf_tile<BN, MODE == 0> TB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TA::BYTES + TB::BYTES];
    constexpr int WN = BN / (32 * NT);
    constexpr int NTH = (BM / (32 * MT)) * WN * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (N + BN - 1) / BN;
    const int m0 = ((int)blockIdx.x / tiles_n) * BM, n0 = ((int)blockIdx.x % tiles_n) * BN;
    int kc = (K + ksplit - 1) / ksplit;
    kc = (kc + GB_K - 1) / GB_K * GB_K;
    const int k_begin = blockIdx.y * kc, k_end = min(K, k_begin + kc);
    if (k_begin >= k_end) return;
    const int wm = (wave / WN) * 32 * MT, wn = (wave % WN) * 32 * NT;
    tile_loader<BM, MODE != 2, NTH> la;
    tile_loader<BN, MODE == 0, NTH> lb;
    const int csm = (1 << g.lcs) - 1;
    const unsigned sa = (unsigned)(size_t)(NQ_AS3 unsigned char*)smem, sb = sa + TA::BYTES;
    smem[TA::BYTES + TB::BYTES - 1] = 0;                    // a pad byte: the kernel must be seen to use its LDS through the symbol

    int a_s[la.NV], a_y[la.NV], a_x[la.NV];
    if (MODE != 2) {
#pragma unroll
        for (int j = 0; j < la.NV; ++j) {
            const int row = m0 + (tid + NTH * j) / (GB_K / 4);
            const uint32_t s = fdiv_q((uint32_t)row, g.d_img);
            const uint32_t rem = (uint32_t)row - s * (uint32_t)(g.hr * g.wr);
            const uint32_t y = fdiv_q(rem, g.d_row);
            a_s[j] = row < M ? (int)s : -1;
            a_y[j] = (int)y;
            a_x[j] = (int)(rem - y * (uint32_t)g.wr);
        }
    }
    auto load_a = [&](int k0) {
#pragma unroll
        for (int j = 0; j < la.NV; ++j) {
            const int i = tid + NTH * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (MODE != 2) {
                const int kk = k0 + 4 * (i % (GB_K / 4));
                if (a_s[j] >= 0 && kk < k_end) v = conv_gather(G, g, a_s[j], a_y[j], a_x[j], kk >> g.lcs, kk & csm);
            } else {
                const int kk = k0 + i / (BM / 4), m = m0 + 4 * (i % (BM / 4));
                if (kk < k_end && m < M) v = *(const f32x4*)(O + (int64_t)kk * g.co + m);
            }
            la.v[j] = v;
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int j = 0; j < lb.NV; ++j) {
            const int i = tid + NTH * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 0) {
                const int n = n0 + i / (GB_K / 4), kk = k0 + 4 * (i % (GB_K / 4));
                if (n < N && kk < k_end) v = *(const f32x4*)(O + (int64_t)n * K + kk);
            } else if (MODE == 1) {
                const int kk = k0 + i / (BN / 4), n = n0 + 4 * (i % (BN / 4));
                if (kk < k_end && n < N) {
                    const int tap = kk >> g.lcs, co = kk & csm;
                    v = *(const f32x4*)(O + (int64_t)co * (9 * g.ci) + tap * g.ci + n);
                }
            } else {
                const int kk = k0 + i / (BN / 4), n = n0 + 4 * (i % (BN / 4));
                if (kk < k_end && n < N) {
                    const uint32_t s = fdiv_q((uint32_t)kk, g.d_img);
                    const uint32_t rem = (uint32_t)kk - s * (uint32_t)(g.hr * g.wr);
                    const uint32_t y = fdiv_q(rem, g.d_row);
                    v = conv_gather(G, g, (int)s, (int)y, (int)(rem - y * (uint32_t)g.wr), n >> g.lcs, n & csm);
                }
            }
            lb.v[j] = v;
        }
    };

    unsigned fa[MT], fb[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[i] = TA::lane_base(sa, wm + 32 * i, lane);
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = TB::lane_base(sb, wn + 32 * j, lane);

    load_a(k_begin);
    load_b(k_begin);
    TA::template store<la.NV, NTH>(la.v, sa, tid);
    TB::template store<lb.NV, NTH>(lb.v, sb, tid);
    __syncthreads();
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero16();
    for (int k0 = k_begin; k0 < k_end; k0 += GB_K) {
        const bool more = k0 + GB_K < k_end;
        if (more) {
            load_a(k0 + GB_K);
            load_b(k0 + GB_K);
        }
#pragma unroll
        for (int s = 0; s < GB_K / 16; ++s) {
            f32x4 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) { ah[i] = TA::frag(fa[i], s); al[i] = TA::frag(fa[i] + TA::PLANE, s); }
#pragma unroll
            for (int j = 0; j < NT; ++j) { bh[j] = TB::frag(fb[j], s); bl[j] = TB::frag(fb[j] + TB::PLANE, s); }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = mfma_bf(ah[i], bh[j], acc[i][j]);
                    acc[i][j] = mfma_bf(ah[i], bl[j], acc[i][j]);
                    acc[i][j] = mfma_bf(al[i], bh[j], acc[i][j]);
                }
        }
        __syncthreads();
        if (more) {
            TA::template store<la.NV, NTH>(la.v, sa, tid);
            TB::template store<lb.NV, NTH>(lb.v, sb, tid);
            __syncthreads();
        }
    }
    // forward with stats != NULL: the per-channel sums of z and z^2 over all rows (the BatchNorm batch statistics) ride
    // along in float64 -- lane pairs, then the workgroup's waves through LDS, then one atomic per channel and workgroup
    __shared__ double red[MODE == 0 ? 2 * BN : 1];
    const bool with_stats = MODE == 0 && stats != nullptr;
    if (with_stats) {
        for (int q = tid; q < 2 * BN; q += NTH) red[q] = 0.0;
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn + 32 * j + (lane & 31);
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + NQ_DROW(r, lane >> 5);
                if (row < M && col < N) {
                    const float v = acc[i][j][r];
                    if (MODE == 2) atomicAdd(C + (int64_t)row * N + col, v);
                    else {
                        const float zv = bias ? v + bias[col] : v;
                        C[(int64_t)row * N + col] = zv;
                        if (MODE == 0) { s1 += (double)zv; s2 += (double)zv * (double)zv; }
                    }
                }
            }
            if (with_stats) {
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (lane < 32 && col < N) {
                    atomicAdd(&red[wn + 32 * j + lane], s1);
                    atomicAdd(&red[BN + wn + 32 * j + lane], s2);
                }
            }
        }
    if (with_stats) {
        __syncthreads();
        for (int q = tid; q < BN; q += NTH)
            if (n0 + q < N) {
                atomicAdd(stats + n0 + q, red[q]);
                atomicAdd(stats + N + n0 + q, red[BN + q]);
            }
    }
}

template <int BM, int BN, int MT, int NT, int MODE>
static void conv_launch(hipStream_t st, const float* gsrc, const float* other, float* c, const conv_geom& g, int M, int N, int K,
                        int ksplit, const float* bias, bool bf16, double* stats = nullptr) {
    const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const dim3 grid((unsigned)tiles, ksplit), block((BM / (32 * MT)) * (BN / (32 * NT)) * 64);
    if (bf16) hipLaunchKernelGGL((conv_gemm_bf16_kernel<BM, BN, MT, NT, MODE>), grid, block, 0, st, gsrc, other, c, g, M, N, K, ksplit, bias, stats);
    else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, MT, NT, MODE>), grid, block, 0, st, gsrc, other, c, g, M, N, K, ksplit, bias, stats);
}

static int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// mode 0: x -> z (bias may be NULL); mode 1: dz -> dx; mode 2: (x, dz) -> dw += (dw zeroed by the caller), ksplit chunks
static int conv3x3_gemm(bool bf16, int32_t mode, const float* x_or_dz, const float* w_or_dz, float* out, int32_t n_segments,
                        int32_t h, int32_t w, int32_t ci, int32_t co, int32_t pad_w, const float* bias,
                        int32_t ksplit, void* stream, double* stats = nullptr) {
    const int wo = w + 2 * pad_w - 2;
    if (mode < 0 || mode > 2 || !x_or_dz || !w_or_dz || !out || n_segments <= 0 || h <= 0 || w <= 0 || wo <= 0 || pad_w < 0 ||
        pad_w > 1 || ilog2_exact(ci) < 2 || ilog2_exact(co) < 2 || ksplit < 1 || ksplit > 65535 || (mode != 2 && ksplit != 1) ||
        (mode != 0 && bias))
        return NISQA_ERR_ARG;
    const int64_t rows_out = (int64_t)n_segments * h * wo, rows_in = (int64_t)n_segments * h * w;
    if (rows_out > 0x7fffffff || rows_in > 0x7fffffff) return NISQA_ERR_ARG;
    conv_geom g;
    g.ci = ci;
    g.co = co;
    g.pad = pad_w;
    hipStream_t st = (hipStream_t)stream;
    NQ_LAUNCH_BEGIN();
    if (mode == 1) {                                        // rows = input pixels, gather dz [H][Wo][co]
        g.hr = h; g.wr = w; g.hs = h; g.ws = wo; g.lcs = ilog2_exact(co); g.sgn = -1;
        g.d_img = fdiv_make((uint32_t)(h * w)); g.d_row = fdiv_make((uint32_t)w);
        const int M = (int)rows_in, N = ci, K = 9 * co;
        if (N > 32) conv_launch<256, 64, 2, 1, 1>(st, x_or_dz, w_or_dz, out, g, M, N, K, 1, nullptr, bf16);
        else conv_launch<128, 32, 1, 1, 1>(st, x_or_dz, w_or_dz, out, g, M, N, K, 1, nullptr, bf16);
    } else {                                                // rows = output pixels, gather x [H][W][ci]
        g.hr = h; g.wr = wo; g.hs = h; g.ws = w; g.lcs = ilog2_exact(ci); g.sgn = 1;
        g.d_img = fdiv_make((uint32_t)(h * wo)); g.d_row = fdiv_make((uint32_t)wo);
        if (mode == 0) {
            const int M = (int)rows_out, N = co, K = 9 * ci;
            if (N > 32) conv_launch<256, 64, 2, 1, 0>(st, x_or_dz, w_or_dz, out, g, M, N, K, 1, bias, bf16, stats);
            else conv_launch<128, 32, 1, 1, 0>(st, x_or_dz, w_or_dz, out, g, M, N, K, 1, bias, bf16, stats);
        } else {
            const int M = co, N = 9 * ci, K = (int)rows_out;
            if (N >= 256) conv_launch<64, 256, 2, 1, 2>(st, x_or_dz, w_or_dz, out, g, M, N, K, ksplit, nullptr, bf16);
            else conv_launch<64, 64, 1, 1, 2>(st, x_or_dz, w_or_dz, out, g, M, N, K, ksplit, nullptr, bf16);
        }
    }
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_conv3x3_gemm(int32_t mode, const float* x_or_dz, const float* w_or_dz, float* out, int32_t n_segments,
                                  int32_t h, int32_t w, int32_t ci, int32_t co, int32_t pad_w, const float* bias,
                                  int32_t ksplit, void* stream) {
    return conv3x3_gemm(false, mode, x_or_dz, w_or_dz, out, n_segments, h, w, ci, co, pad_w, bias, ksplit, stream);
}
// forward convolution that also leaves sum z, sum z^2 per output channel in stats2c [dev, float64, zeroed by the caller]
extern "C" int nisqa_conv3x3_fwd_stats(int32_t split_bf16, const float* x, const float* w_, float* z, int32_t n_segments, int32_t h,
                                       int32_t w, int32_t ci, int32_t co, int32_t pad_w, const float* bias, double* stats2c,
                                       void* stream) {
    if (!stats2c) return NISQA_ERR_ARG;
    return conv3x3_gemm(split_bf16 != 0, 0, x, w_, z, n_segments, h, w, ci, co, pad_w, bias, 1, stream, stats2c);
}
extern "C" int nisqa_conv3x3_gemm_bf16(int32_t mode, const float* x_or_dz, const float* w_or_dz, float* out, int32_t n_segments,
                                       int32_t h, int32_t w, int32_t ci, int32_t co, int32_t pad_w, const float* bias,
                                       int32_t ksplit, void* stream) {
    return conv3x3_gemm(true, mode, x_or_dz, w_or_dz, out, n_segments, h, w, ci, co, pad_w, bias, ksplit, stream);
}

// ---------------------------------------------------------------------------------------------------------
// patches
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_mel_kernel(const float* __restrict__ mel_tm,
                                                         const int32_t* __restrict__ frame_off,
                                                         const int32_t* __restrict__ seg_off,
                                                         const float* __restrict__ clip_floor, int n_clips, int seg_hop,
                                                         float* __restrict__ col) {
    const int s = blockIdx.x;
    const int b = find_segment(seg_off, n_clips, s);
    const int k = s - seg_off[b];
    const float fl = clip_floor[b];
    const float* src = mel_tm + (int64_t)(frame_off[b] + k * seg_hop) * 48;      // [15 frames][48 bands]
    float* dst = col + (int64_t)s * (720 * 9);
    for (int i = threadIdx.x; i < 720 * 9; i += 256) {
        const int p = i / 9, tap = i - 9 * p;
        const int m = p / 15 + tap / 3 - 1, j = p % 15 + tap % 3 - 1;
        dst[i] = ((unsigned)m < 48u && (unsigned)j < 15u) ? fmaxf(src[j * 48 + m], fl) : 0.f;
    }
}

// conv1 (1 -> 16 channels, 9 taps) is memory-bound and its K = 9 makes a poor GEMM: forward and weight gradient read
// the 15 x 48 segment patch from the spectrogram directly (zero-bordered copy in LDS), no patch matrix in HBM.
template <int STRIDE = 256>
__device__ __forceinline__ void stage_patch(float (*patch)[50], const float* __restrict__ src, float fl, int tid) {
    for (int i = tid; i < 17 * 50; i += STRIDE) {
        const int j = i / 50 - 1, m = i % 50 - 1;           // frame, mel band
        patch[0][i] = ((unsigned)j < 15u && (unsigned)m < 48u) ? fmaxf(src[j * 48 + m], fl) : 0.f;
    }
}

__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off,
                                                        const int32_t* __restrict__ seg_off, const float* __restrict__ clip_floor,
                                                        int n_clips, int seg_hop, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ z) {
    __shared__ float patch[17][50];
    __shared__ float ws[16 * 9 + 16];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int b = find_segment(seg_off, n_clips, s);
    const int k = s - seg_off[b];
    stage_patch(patch, mel_tm + (int64_t)(frame_off[b] + k * seg_hop) * 48, clip_floor[b], tid);
    if (tid < 144) ws[tid] = w[tid];
    else if (tid < 160) ws[tid] = bias[tid - 144];
    __syncthreads();
    for (int p = tid; p < 720; p += 256) {
        const int m = p / 15, j = p % 15;                   // pixel (y = mel band, x = frame)
        float x[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) x[t] = patch[j + t % 3][m + t / 3];      // tap = dy*3+dx: band m+dy-1, frame j+dx-1
        f32x4* dst = (f32x4*)(z + ((int64_t)s * 720 + p) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = ws[144 + 4 * q + e];
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = fmaf(x[t], ws[(4 * q + e) * 9 + t], acc);
                o[e] = acc;
            }
            dst[q] = o;
        }
    }
}

// dW1[co][tap] += sum over this workgroup's segments and all pixels of dz[s][p][co] * x[s][p + tap].
// Thread (tap, pixel lane) keeps all 16 channels of its tap in registers: per pixel one LDS read of x and one 64-byte
// row of dz (four 128-bit loads) feed 16 FMAs; the 28 pixel lanes of a tap are reduced through LDS atomics at the end.
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off,
                                                          const int32_t* __restrict__ seg_off, const float* __restrict__ clip_floor,
                                                          int n_clips, int n_segments, int seg_hop, const float* __restrict__ dz,
                                                          float* __restrict__ dw) {
    __shared__ float patch[17][50];
    __shared__ float red[144];
    const int tid = threadIdx.x;
    const int tap = tid / 28, pl = tid % 28;               // 9 taps x 28 pixel lanes = 252 threads
    const int dy = tap / 3, dx = tap % 3;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    if (tid < 144) red[tid] = 0.f;
    for (int s = blockIdx.x; s < n_segments; s += gridDim.x) {
        const int b = find_segment(seg_off, n_clips, s);
        const int k = s - seg_off[b];
        __syncthreads();
        stage_patch(patch, mel_tm + (int64_t)(frame_off[b] + k * seg_hop) * 48, clip_floor[b], tid);
        __syncthreads();
        if (tap < 9) {
            const f32x4* d = (const f32x4*)(dz + (int64_t)s * 720 * 16);
            for (int p = pl; p < 720; p += 28) {
                const float x = patch[p % 15 + dx][p / 15 + dy];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = d[p * 4 + q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * q + e] = fmaf(v[e], x, acc[4 * q + e]);
                }
            }
        }
    }
    if (tap < 9) {
#pragma unroll
        for (int c = 0; c < 16; ++c) atomicAdd(&red[c * 9 + tap], acc[c]);
    }
    __syncthreads();
    if (tid < 144) atomicAdd(dw + tid, red[tid]);
}

extern "C" int nisqa_conv1_fwd(const float* mel_tm, const int32_t* frame_off, const int32_t* seg_off, const float* clip_floor,
                               int32_t n_clips, int32_t n_segments, int32_t seg_hop, const float* w, const float* bias,
                               float* z, void* stream) {
    if (!mel_tm || !frame_off || !seg_off || !clip_floor || !w || !bias || !z || n_clips <= 0 || n_segments <= 0 || seg_hop <= 0)
        return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(conv1_fwd_kernel, dim3(n_segments), dim3(256), 0, (hipStream_t)stream, mel_tm, frame_off, seg_off,
                       clip_floor, n_clips, seg_hop, w, bias, z);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_conv1_wgrad(const float* mel_tm, const int32_t* frame_off, const int32_t* seg_off, const float* clip_floor,
                                 int32_t n_clips, int32_t n_segments, int32_t seg_hop, const float* dz, float* dw, void* stream) {
    if (!mel_tm || !frame_off || !seg_off || !clip_floor || !dz || !dw || n_clips <= 0 || n_segments <= 0 || seg_hop <= 0)
        return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(conv1_wgrad_kernel, dim3(n_segments < 2048 ? n_segments : 2048), dim3(256), 0, (hipStream_t)stream, mel_tm,
                       frame_off, seg_off, clip_floor, n_clips, n_segments, seg_hop, dz, dw);
    return NQ_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------------
// Layer 1 without its activations.  z1 = conv1(patch) has S x 720 x 16 entries (364 MB at 32 x 10 s clips) and the
// unfused step makes nine passes over tensors of that size (z, dyb, dz).  But z is AFFINE in nine numbers per pixel, so
//   * the BatchNorm statistics follow from the first and second moments of the patches, P1[t] = sum patch[t] and
//     P2[t][u] = sum patch[t] patch[u] over all pixels of all valid segments (54 float64 numbers, one pass over the
//     spectrogram): sum z_c = w_c . P1 + N b_c, sum z_c^2 = w_c' P2 w_c + 2 b_c w_c . P1 + N b_c^2;
//   * forward = recompute the six pixels of every pooling window from the LDS copy of the segment, normalise, ReLU, max;
//   * the gradient arriving from the pool is non-zero at ONE pixel per pooled value: the reductions of the BatchNorm
//     backward and the data-dependent part of the weight gradient are sums over the 168 x 16 pooled values of a segment
//     (z at the argmax pixel is recomputed), and the dense parts of dz = gamma rstd (dyb - mean(dyb) - xhat mean(dyb xhat))
//     contribute -mean(dyb) P1[t] - mean(dyb xhat) sum xhat patch[t], again moments.  The bias gradient is exactly zero.
// No tensor of 720 pixels per segment is ever written.
// ---------------------------------------------------------------------------------------------------------
#define C1_MOM 54                                            /* P1[9], then P2 upper triangle [t <= u] */
typedef int i32x4 __attribute__((ext_vector_type(4)));
NQ_DEV int win_lo(int i, int n_in, int n_out);
NQ_DEV int win_hi(int i, int n_in, int n_out);
__global__ void bn_finalize_kernel(const double* __restrict__ sums, int c, int64_t m_rows, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ mean_rstd);
NQ_DEV int c1_p2(int t, int u) { return t <= u ? 9 + t * 9 - t * (t - 1) / 2 + (u - t) : 9 + u * 9 - u * (u - 1) / 2 + (t - u); }

// every WAVE walks over its own segments with its own LDS copy: no workgroup barrier inside the loop (with 256 threads per
// segment a thread had three pixels between two barriers and the kernel was latency-bound: 89 us for 8 us of arithmetic)
#define C1_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
__global__ __launch_bounds__(256) void conv1_moments_kernel(const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off,
                                                            const int32_t* __restrict__ seg_off, const float* __restrict__ clip_floor,
                                                            int n_clips, int n_segments, int seg_hop, double* __restrict__ mom) {
    __shared__ float patches[4][17][50];
    __shared__ double red[C1_MOM];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float (*patch)[50] = patches[wave];
    double acc[C1_MOM];
#pragma unroll
    for (int i = 0; i < C1_MOM; ++i) acc[i] = 0.0;
    if (tid < C1_MOM) red[tid] = 0.0;
    for (int s = blockIdx.x * 4 + wave; s < n_segments; s += gridDim.x * 4) {
        const int b = find_segment(seg_off, n_clips, s);
        C1_WSYNC();
        stage_patch<64>(patch, mel_tm + (int64_t)(frame_off[b] + (s - seg_off[b]) * seg_hop) * 48, clip_floor[b], lane);
        C1_WSYNC();
        for (int p = lane; p < 720; p += 64) {
            const int m = p / 15, j = p % 15;
            double x[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) x[t] = (double)patch[j + t % 3][m + t / 3];
            int q = 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                acc[t] += x[t];
#pragma unroll
                for (int u = t; u < 9; ++u) acc[q++] += x[t] * x[u];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < C1_MOM; ++i) {
        double v = acc[i];
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) atomicAdd(&red[i], v);
    }
    __syncthreads();
    if (tid < C1_MOM) atomicAdd(mom + tid, red[tid]);
}

// sums[c] = sum z_c, sums[16 + c] = sum z_c^2 from the moments (the layout col_dot(z, z) produces)
__global__ void conv1_sums_kernel(const double* __restrict__ mom, const float* __restrict__ w, const float* __restrict__ bias,
                                  int64_t n_rows, double* __restrict__ sums) {
    const int c = threadIdx.x;
    if (c >= 16) return;
    double wp = 0.0, wpw = 0.0;
    for (int t = 0; t < 9; ++t) {
        wp += (double)w[c * 9 + t] * mom[t];
        for (int u = 0; u < 9; ++u) wpw += (double)w[c * 9 + t] * (double)w[c * 9 + u] * mom[c1_p2(t, u)];
    }
    const double b = bias[c], n = (double)n_rows;
    sums[c] = wp + n * b;
    sums[16 + c] = wpw + 2.0 * b * wp + n * b * b;
}

// one thread = four channels of one pooled value: y[s][oy * 7 + ox][c], arg = pixel index (band * 15 + frame) of the maximum
__global__ __launch_bounds__(256) void conv1_bn_act_pool_fwd_kernel(
    const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off, const int32_t* __restrict__ seg_off,
    const float* __restrict__ clip_floor, int n_clips, int seg_hop, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean_rstd,
    const float* __restrict__ drop, float* __restrict__ y, int32_t* __restrict__ arg) {
    __shared__ float patch[17][50];
    __shared__ float ws[16 * 9 + 16 + 32];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int b = find_segment(seg_off, n_clips, s);
    stage_patch(patch, mel_tm + (int64_t)(frame_off[b] + (s - seg_off[b]) * seg_hop) * 48, clip_floor[b], tid);
    if (tid < 144) ws[tid] = w[tid];
    else if (tid < 160) ws[tid] = bias[tid - 144];
    else if (tid < 176) {                                   // the affine form of the normalisation: y = z * g + sh
        const int c = tid - 160;
        const float g = gamma[c] * mean_rstd[16 + c];
        ws[160 + c] = g;
        ws[176 + c] = beta[c] - mean_rstd[c] * g;
    }
    __syncthreads();
    for (int i = tid; i < 168 * 4; i += 256) {
        const int o = i >> 2, ch = 4 * (i & 3);
        const int oy = o / 7, ox = o % 7;
        f32x4 best = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        i32x4 bp = {0, 0, 0, 0};
        for (int m = win_lo(oy, 48, 24); m < win_hi(oy, 48, 24); ++m)
            for (int j = win_lo(ox, 15, 7); j < win_hi(ox, 15, 7); ++j) {
                float x[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) x[t] = patch[j + t % 3][m + t / 3];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float acc = ws[144 + ch + e];
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc = fmaf(x[t], ws[(ch + e) * 9 + t], acc);
                    const float r = fmaxf(acc * ws[160 + ch + e] + ws[176 + ch + e], 0.f);
                    if (r > best[e]) { best[e] = r; bp[e] = m * 15 + j; }
                }
            }
        if (drop) best *= *(const f32x4*)(drop + (int64_t)s * 16 + ch);
        *(f32x4*)(y + ((int64_t)s * 168 + o) * 16 + ch) = best;
        *(i32x4*)(arg + ((int64_t)s * 168 + o) * 16 + ch) = bp;
    }
}

// acc[c][0] = sum dyb, [1] = sum dyb * z, [2 + t] = sum dyb * patch[t] over all pooled values whose ReLU is open
__global__ __launch_bounds__(256) void conv1_bn_act_pool_bwd_kernel(
    const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off, const int32_t* __restrict__ seg_off,
    const float* __restrict__ clip_floor, int n_clips, int n_segments, int seg_hop, const float* __restrict__ w,
    const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean_rstd, const float* __restrict__ drop, const float* __restrict__ dy,
    const int32_t* __restrict__ arg, double* __restrict__ out) {
    __shared__ float patches[4][17][50];
    __shared__ float ws[16 * 9 + 16 + 32];
    __shared__ double red[16 * 11];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ch = 4 * (tid & 3);                            // items stride by 64: a lane always meets the same four channels
    float (*patch)[50] = patches[wave];
    if (tid < 144) ws[tid] = w[tid];
    else if (tid < 160) ws[tid] = bias[tid - 144];
    else if (tid < 176) {
        const int c = tid - 160;
        const float g = gamma[c] * mean_rstd[16 + c];
        ws[160 + c] = g;
        ws[176 + c] = beta[c] - mean_rstd[c] * g;
    }
    if (tid < 176) red[tid] = 0.0;
    double a[4][11];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < 11; ++q) a[e][q] = 0.0;
    __syncthreads();                                        // ws
    for (int s = blockIdx.x * 4 + wave; s < n_segments; s += gridDim.x * 4) {     // a wave per segment, no workgroup barrier
        const int b = find_segment(seg_off, n_clips, s);
        C1_WSYNC();
        stage_patch<64>(patch, mel_tm + (int64_t)(frame_off[b] + (s - seg_off[b]) * seg_hop) * 48, clip_floor[b], lane);
        C1_WSYNC();
        f32x4 dr = {1.f, 1.f, 1.f, 1.f};
        if (drop) dr = *(const f32x4*)(drop + (int64_t)s * 16 + ch);
        for (int i = lane; i < 168 * 4; i += 64) {
            const int64_t at = ((int64_t)s * 168 + (i >> 2)) * 16 + ch;
            const f32x4 g = *(const f32x4*)(dy + at) * dr;
            const i32x4 ap = *(const i32x4*)(arg + at);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = ap[e] / 15, j = ap[e] % 15;
                float x[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) x[t] = patch[j + t % 3][m + t / 3];
                float z = ws[144 + ch + e];
#pragma unroll
                for (int t = 0; t < 9; ++t) z = fmaf(x[t], ws[(ch + e) * 9 + t], z);
                if (z * ws[160 + ch + e] + ws[176 + ch + e] > 0.f) {                  // ReLU gate
                    const double gd = (double)g[e];
                    a[e][0] += gd;
                    a[e][1] += gd * (double)z;
#pragma unroll
                    for (int t = 0; t < 9; ++t) a[e][2 + t] += gd * (double)x[t];
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            double v = a[e][q];                             // lanes with equal (lane & 3) hold the same channels
            for (int o = 32; o >= 4; o >>= 1) v += __shfl_xor(v, o);
            if ((tid & 63) < 4) atomicAdd(&red[(ch + e) * 11 + q], v);
        }
    __syncthreads();
    if (tid < 176) atomicAdd(out + tid, red[tid]);
}

// dgamma, dbeta, dw[16][9] of layer 1 from the sparse sums and the patch moments (one thread per (channel, tap))
__global__ void conv1_bwd_finalize_kernel(const double* __restrict__ acc, const double* __restrict__ mom, const float* __restrict__ w,
                                          const float* __restrict__ bias, const float* __restrict__ gamma,
                                          const float* __restrict__ mean_rstd, int64_t n_rows, float* __restrict__ dgamma,
                                          float* __restrict__ dbeta, float* __restrict__ dw) {
    const int c = threadIdx.x / 9, t = threadIdx.x % 9;
    if (c >= 16) return;
    const double mean = mean_rstd[c], rstd = mean_rstd[16 + c], n = (double)n_rows;
    const double m1 = acc[c * 11], m2 = rstd * (acc[c * 11 + 1] - mean * acc[c * 11]);      // sum dyb, sum dyb * xhat
    if (t == 0) {
        dbeta[c] = (float)m1;
        dgamma[c] = (float)m2;
    }
    double zp = (double)bias[c] * mom[t];                                              // sum z * patch[t]
    for (int u = 0; u < 9; ++u) zp += (double)w[c * 9 + u] * mom[c1_p2(u, t)];
    const double xp = rstd * (zp - mean * mom[t]);                                      // sum xhat * patch[t]
    dw[c * 9 + t] = (float)((double)gamma[c] * rstd * (acc[c * 11 + 2 + t] - m1 / n * mom[t] - m2 / n * xp));
}

extern "C" int nisqa_conv1_moments(const float* mel_tm, const int32_t* frame_off, const int32_t* seg_off, const float* clip_floor,
                                   int32_t n_clips, int32_t n_segments, int32_t seg_hop, double* mom54, void* stream) {
    if (!mel_tm || !frame_off || !seg_off || !clip_floor || !mom54 || n_clips <= 0 || n_segments <= 0 || seg_hop <= 0) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(conv1_moments_kernel, dim3((n_segments + 3) / 4 < 512 ? (n_segments + 3) / 4 : 512), dim3(256), 0, (hipStream_t)stream, mel_tm,
                       frame_off, seg_off, clip_floor, n_clips, n_segments, seg_hop, mom54);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_conv1_bn_act_pool_fwd(const float* mel_tm, const int32_t* frame_off, const int32_t* seg_off, const float* clip_floor,
                                           int32_t n_clips, int32_t n_segments, int32_t seg_hop, const float* w, const float* bias,
                                           const double* mom54, const float* gamma, const float* beta, float* running_mean,
                                           float* running_var, double* sums32, float* mean_rstd, const float* drop, float* y,
                                           int32_t* arg, void* stream) {
    if (!mel_tm || !frame_off || !seg_off || !clip_floor || !w || !bias || !mom54 || !gamma || !beta || !running_mean || !running_var ||
        !sums32 || !mean_rstd || !y || !arg || n_clips <= 0 || n_segments <= 0 || seg_hop <= 0)
        return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)n_segments * 720;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(conv1_sums_kernel, dim3(1), dim3(64), 0, st, mom54, w, bias, rows, sums32);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)sums32, 16, rows, running_mean, running_var, mean_rstd);
    hipLaunchKernelGGL(conv1_bn_act_pool_fwd_kernel, dim3(n_segments), dim3(256), 0, st, mel_tm, frame_off, seg_off, clip_floor, n_clips,
                       seg_hop, w, bias, gamma, beta, (const float*)mean_rstd, drop, y, arg);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_conv1_bn_act_pool_bwd(const float* mel_tm, const int32_t* frame_off, const int32_t* seg_off, const float* clip_floor,
                                           int32_t n_clips, int32_t n_segments, int32_t seg_hop, const float* w, const float* bias,
                                           const double* mom54, const float* gamma, const float* beta, const float* mean_rstd,
                                           const float* drop, const float* dy, const int32_t* arg, double* acc176, float* dgamma,
                                           float* dbeta, float* dw, void* stream) {
    if (!mel_tm || !frame_off || !seg_off || !clip_floor || !w || !bias || !mom54 || !gamma || !beta || !mean_rstd || !dy || !arg ||
        !acc176 || !dgamma || !dbeta || !dw || n_clips <= 0 || n_segments <= 0 || seg_hop <= 0)
        return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(conv1_bn_act_pool_bwd_kernel, dim3((n_segments + 3) / 4 < 512 ? (n_segments + 3) / 4 : 512), dim3(256), 0, st, mel_tm, frame_off,
                       seg_off, clip_floor, n_clips, n_segments, seg_hop, w, bias, gamma, beta, mean_rstd, drop, dy, arg, acc176);
    hipLaunchKernelGGL(conv1_bwd_finalize_kernel, dim3(1), dim3(144), 0, st, (const double*)acc176, mom54, w, bias, gamma, mean_rstd,
                       (int64_t)n_segments * 720, dgamma, dbeta, dw);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_im2col_mel(const float* mel_tm, const int32_t* frame_off, const int32_t* seg_off,
                                const float* clip_floor, int32_t n_clips, int32_t n_segments, int32_t seg_hop,
                                float* col, void* stream) {
    if (!mel_tm || !frame_off || !seg_off || !clip_floor || !col || n_clips <= 0 || n_segments <= 0 || seg_hop <= 0)
        return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(im2col_mel_kernel, dim3(n_segments), dim3(256), 0, (hipStream_t)stream, mel_tm, frame_off,
                       seg_off, clip_floor, n_clips, seg_hop, col);
    return NQ_LAUNCH_STATUS();
}

// The geometry template arguments make the index arithmetic divisions by constants (0 = use the run-time values):
// with run-time divisors these kernels spent ~10 integer divisions per element moved -- arithmetic, not HBM, was their cost.
// One thread moves four consecutive channels (c % 4 == 0: 128-bit accesses, a quarter of the index arithmetic).
template <int H, int W, int C, int PW>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x, int64_t total4, int h_, int w_, int c_,
                                                        int pad_, float* __restrict__ col) {
    const int h = H ? H : h_, w = W ? W : w_, c = C ? C : c_, pad_w = H ? PW : pad_;
    const int wo = w + 2 * pad_w - 2, kc4 = 9 * c / 4, c4 = c / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / kc4;
        const int k = (int)(i - row * kc4);
        const int tap = k / c4, ch = 4 * (k - tap * c4);
        const int64_t s = row / (h * wo);
        const int po = (int)(row - s * (h * wo));
        const int y = po / wo + tap / 3 - 1, xx = po % wo + tap % 3 - pad_w;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)y < (unsigned)h && (unsigned)xx < (unsigned)w) v = *(const f32x4*)(x + (s * (h * w) + y * w + xx) * c + ch);
        ((f32x4*)col)[i] = v;
    }
}

template <int H, int W, int C, int PW>
__global__ __launch_bounds__(256) void col2im3x3_kernel(const float* __restrict__ dcol, int64_t total4, int h_, int w_, int c_,
                                                        int pad_, float* __restrict__ dx) {
    const int h = H ? H : h_, w = W ? W : w_, c = C ? C : c_, pad_w = H ? PW : pad_;
    const int wo = w + 2 * pad_w - 2, kc = 9 * c, c4 = c / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int ch = 4 * (int)(i % c4);
        const int64_t pix = i / c4;
        const int64_t s = pix / (h * w);
        const int p = (int)(pix - s * (h * w));
        const int y = p / w, xx = p % w;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yo = y - tap / 3 + 1, xo = xx - tap % 3 + pad_w;
            if ((unsigned)yo < (unsigned)h && (unsigned)xo < (unsigned)wo)
                acc += *(const f32x4*)(dcol + (s * (h * wo) + yo * wo + xo) * kc + tap * c + ch);
        }
        ((f32x4*)dx)[i] = acc;
    }
}

static int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

// grid-stride kernels whose threads keep per-channel constants across iterations: one resident round of workgroups (256 CUs x
// 8 blocks of 256 threads x 2), so that a thread sees ~10+ elements instead of one and the constants are worth keeping
static int grid_resident(int64_t total) {
    const int64_t g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// geometries of the CNN-SA-AP model get their own instantiation, anything else the generic one
#define NQ_GEOM_DISPATCH(KERNEL, GRID, ST, ...)                                                                   \
    do {                                                                                                          \
        if (h == 24 && w == 7 && c == 16 && pad_w == 1) hipLaunchKernelGGL((KERNEL<24, 7, 16, 1>), GRID, dim3(256), 0, ST, __VA_ARGS__);      \
        else if (h == 12 && w == 5 && c == 32 && pad_w == 1) hipLaunchKernelGGL((KERNEL<12, 5, 32, 1>), GRID, dim3(256), 0, ST, __VA_ARGS__); \
        else if (h == 12 && w == 5 && c == 64 && pad_w == 1) hipLaunchKernelGGL((KERNEL<12, 5, 64, 1>), GRID, dim3(256), 0, ST, __VA_ARGS__); \
        else if (h == 6 && w == 3 && c == 64 && pad_w == 1) hipLaunchKernelGGL((KERNEL<6, 3, 64, 1>), GRID, dim3(256), 0, ST, __VA_ARGS__);   \
        else if (h == 6 && w == 3 && c == 64 && pad_w == 0) hipLaunchKernelGGL((KERNEL<6, 3, 64, 0>), GRID, dim3(256), 0, ST, __VA_ARGS__);   \
        else hipLaunchKernelGGL((KERNEL<0, 0, 0, 0>), GRID, dim3(256), 0, ST, __VA_ARGS__);                        \
    } while (0)

extern "C" int nisqa_im2col3x3(const float* x, int32_t n_segments, int32_t h, int32_t w, int32_t c, int32_t pad_w,
                               float* col, void* stream) {
    const int wo = w + 2 * pad_w - 2;
    if (!x || !col || n_segments <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || pad_w < 0 || pad_w > 1 || wo <= 0 ||
        ((((uintptr_t)x) | ((uintptr_t)col)) & 15))
        return NISQA_ERR_ARG;
    const int64_t total = (int64_t)n_segments * h * wo * 9 * c / 4;
    NQ_LAUNCH_BEGIN();
    NQ_GEOM_DISPATCH(im2col3x3_kernel, dim3(grid_for(total)), (hipStream_t)stream, x, total, h, w, c, pad_w, col);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_col2im3x3(const float* dcol, int32_t n_segments, int32_t h, int32_t w, int32_t c, int32_t pad_w,
                               float* dx, void* stream) {
    const int wo = w + 2 * pad_w - 2;
    if (!dcol || !dx || n_segments <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || pad_w < 0 || pad_w > 1 || wo <= 0 ||
        ((((uintptr_t)dcol) | ((uintptr_t)dx)) & 15))
        return NISQA_ERR_ARG;
    const int64_t total = (int64_t)n_segments * h * w * c / 4;
    NQ_LAUNCH_BEGIN();
    NQ_GEOM_DISPATCH(col2im3x3_kernel, dim3(grid_for(total)), (hipStream_t)stream, dcol, total, h, w, c, pad_w, dx);
    return NQ_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------------
// column reductions in float64: thread (channel = tid % c, row lane = tid / c) walks rows, LDS tree, atomics
// ---------------------------------------------------------------------------------------------------------
template <int V>                                          // V = 4: c % 4 == 0, 128-bit loads; V = 1: any c
__global__ __launch_bounds__(256) void col_dot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      int64_t rows, int c, int64_t rows_per_block, double* __restrict__ out) {
    __shared__ double s1[256 * V], s2[256 * V];
    const int tid = threadIdx.x;
    const int cg = c / V;                                 // column groups of V channels
    const int rl = 256 / cg;                              // row lanes per block
    const int g = tid % cg, r0 = tid / cg;
    double x1[V], x2[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { x1[e] = 0.0; x2[e] = 0.0; }
    if (r0 < rl) {
        const int64_t begin = (int64_t)blockIdx.x * rows_per_block, end = min(rows, begin + rows_per_block);
        const bool same = a == b;
        int64_t r = begin + r0;
        for (; r + rl < end; r += 2 * rl) {                // two rows in flight per thread
            float av[2][V], bv[2][V];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if constexpr (V == 4) {
                    const f32x4 t = *(const f32x4*)(a + (r + q * rl) * c + 4 * g);
                    av[q][0] = t[0]; av[q][1] = t[1]; av[q][2] = t[2]; av[q][3] = t[3];
                } else av[q][0] = a[(r + q * rl) * c + g];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (same) {
#pragma unroll
                    for (int e = 0; e < V; ++e) bv[q][e] = av[q][e];
                } else if constexpr (V == 4) {
                    const f32x4 t = *(const f32x4*)(b + (r + q * rl) * c + 4 * g);
                    bv[q][0] = t[0]; bv[q][1] = t[1]; bv[q][2] = t[2]; bv[q][3] = t[3];
                } else bv[q][0] = b[(r + q * rl) * c + g];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < V; ++e) { x1[e] += (double)av[q][e]; x2[e] += (double)av[q][e] * (double)bv[q][e]; }
        }
        for (; r < end; r += rl)
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float av = a[r * c + V * g + e], bv = same ? av : b[r * c + V * g + e];
                x1[e] += (double)av;
                x2[e] += (double)av * (double)bv;
            }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) { s1[tid * V + e] = x1[e]; s2[tid * V + e] = x2[e]; }
    __syncthreads();
    if (tid < c) {                                         // channel tid = V * g + e of row lane q sits at (q * cg + g) * V + e
        double t1 = 0.0, t2 = 0.0;
        for (int q = 0; q < rl; ++q) { t1 += s1[q * c + tid]; t2 += s2[q * c + tid]; }
        atomicAdd(out + tid, t1);
        atomicAdd(out + c + tid, t2);
    }
}

extern "C" int nisqa_col_dot(const float* a, const float* b, int64_t rows, int32_t c, double* out, void* stream) {
    if (!a || !b || !out || rows <= 0 || c <= 0 || c > 256) return NISQA_ERR_ARG;
    int64_t blocks = (rows * c + 256 * 64 - 1) / (256 * 64);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    const int64_t rpb = (rows + blocks - 1) / blocks;
    NQ_LAUNCH_BEGIN();
    const bool vec = (c & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0;
    if (vec) hipLaunchKernelGGL(col_dot_kernel<4>, dim3((int)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, a, b,
                                rows, c, rpb, out);
    else hipLaunchKernelGGL(col_dot_kernel<1>, dim3((int)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, a, b, rows,
                            c, rpb, out);
    return NQ_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm (batch statistics) + ReLU + adaptive max pool + Dropout2d
// ---------------------------------------------------------------------------------------------------------
NQ_DEV int win_lo(int i, int n_in, int n_out) { return (i * n_in) / n_out; }
NQ_DEV int win_hi(int i, int n_in, int n_out) { return ((i + 1) * n_in + n_out - 1) / n_out; }

NQ_DEV void bn_stats(const double* __restrict__ sums, int c, int ch, double m_rows, float& mean, float& rstd, double& var) {
    const double mu = sums[ch] / m_rows;
    var = sums[c + ch] / m_rows - mu * mu;
    if (var < 0.0) var = 0.0;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + 1e-5));
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, int c, int64_t m_rows, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ mean_rstd) {
    const int ch = threadIdx.x;
    if (ch >= c) return;
    float mean, rstd;
    double var;
    bn_stats(sums, c, ch, (double)m_rows, mean, rstd, var);
    mean_rstd[ch] = mean;
    mean_rstd[c + ch] = rstd;
    const double unb = m_rows > 1 ? var * ((double)m_rows / (double)(m_rows - 1)) : var;
    running_mean[ch] = 0.9f * running_mean[ch] + 0.1f * mean;
    running_var[ch] = 0.9f * running_var[ch] + 0.1f * (float)unb;
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// per-channel affine form of the normalisation for four consecutive channels: y = z * g + b
NQ_DEV void bn_affine4(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean_rstd, int c,
                       int ch, f32x4& g, f32x4& b) {
    const f32x4 ga = *(const f32x4*)(gamma + ch), be = *(const f32x4*)(beta + ch);
    const f32x4 mu = *(const f32x4*)(mean_rstd + ch), rs = *(const f32x4*)(mean_rstd + c + ch);
    g = ga * rs;
    b = be - mu * g;
}

// the same constants straight from the float64 column sums (sum z, sum z^2 over m_rows rows): what bn_finalize_kernel would
// have left in mean_rstd, bit for bit
NQ_DEV void bn_affine4_sums(const float* __restrict__ gamma, const float* __restrict__ beta, const double* __restrict__ sums, int c,
                            double m_rows, int ch, f32x4& g, f32x4& b) {
    const f32x4 ga = *(const f32x4*)(gamma + ch), be = *(const f32x4*)(beta + ch);
    f32x4 mu, rs;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double var;
        float m_, r_;
        bn_stats(sums, c, ch + e, m_rows, m_, r_, var);
        mu[e] = m_;
        rs[e] = r_;
    }
    g = ga * rs;
    b = be - mu * g;
}

// One thread = four consecutive channels of one output pixel (c % 4 == 0): 128-bit accesses throughout.  The batch statistics
// are finalised HERE (every thread derives its channels' mean / rstd from the float64 sums; block 0 also writes mean_rstd for
// the backward kernels and updates the running statistics): no separate bn_finalize launch in front of each layer.
template <int H, int W, int C, int HO, int WO>
__global__ __launch_bounds__(256) void bn_act_pool_fwd_kernel(
    const float* __restrict__ z, const float* __restrict__ gamma, const float* __restrict__ beta,
    const double* __restrict__ sums, float* __restrict__ running_mean, float* __restrict__ running_var,
    float* __restrict__ mean_rstd, int64_t total4, int h_, int w_, int c_, int ho_, int wo_,
    const float* __restrict__ drop, float* __restrict__ y, int32_t* __restrict__ arg) {
    const int h = H ? H : h_, w = H ? W : w_, c = H ? C : c_, ho = H ? HO : ho_, wo = H ? WO : wo_;
    const int c4 = c / 4;
    const int64_t m_rows = (total4 / c4) / (ho * wo) * (h * w);
    if (blockIdx.x == 0 && (int)threadIdx.x < c) {            // what bn_finalize_kernel did
        const int ch = threadIdx.x;
        float mean, rstd;
        double var;
        bn_stats(sums, c, ch, (double)m_rows, mean, rstd, var);
        mean_rstd[ch] = mean;
        mean_rstd[c + ch] = rstd;
        const double unb = m_rows > 1 ? var * ((double)m_rows / (double)(m_rows - 1)) : var;
        running_mean[ch] = 0.9f * running_mean[ch] + 0.1f * mean;
        running_var[ch] = 0.9f * running_var[ch] + 0.1f * (float)unb;
    }
    // compiled shapes: c / 4 divides the 256-thread stride, so a thread meets the SAME four channels in every iteration and
    // their constants are computed once (the generic instantiation recomputes them per element)
    constexpr bool INV = H != 0 && 256 % ((C ? C : 4) / 4) == 0;
    // identity pooling (layers 3, 5, 6): the winning pixel is the cell itself -- no backward kernel reads arg there, and not
    // writing it saves a tensor-sized store
    const bool identity = h == ho && w == wo;
    f32x4 g, bsh;
    if (INV) bn_affine4_sums(gamma, beta, sums, c, (double)m_rows, 4 * ((int)threadIdx.x % c4), g, bsh);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int ch = 4 * (int)(i % c4);
        const int64_t op = i / c4;
        const int64_t s = op / (ho * wo);
        const int o = (int)(op - s * (ho * wo));
        const int oy = o / wo, ox = o % wo;
        if (!INV) bn_affine4_sums(gamma, beta, sums, c, (double)m_rows, ch, g, bsh);
        f32x4 best = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        i32x4 bp = {0, 0, 0, 0};
        for (int yy = win_lo(oy, h, ho); yy < win_hi(oy, h, ho); ++yy)
            for (int xx = win_lo(ox, w, wo); xx < win_hi(ox, w, wo); ++xx) {
                const int p = yy * w + xx;
                const f32x4 v = *(const f32x4*)(z + (s * (h * w) + p) * c + ch) * g + bsh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = fmaxf(v[e], 0.f);
                    if (r > best[e]) { best[e] = r; bp[e] = p; }
                }
            }
        if (drop) best *= *(const f32x4*)(drop + s * c + ch);
        ((f32x4*)y)[i] = best;
        if (!identity) ((i32x4*)arg)[i] = bp;
    }
}

#define NQ_POOL_DISPATCH(KERNEL, GRID, ST, ...)                                                                   \
    do {                                                                                                          \
        if (h == 48 && w == 15 && c == 16 && ho == 24 && wo == 7) hipLaunchKernelGGL((KERNEL<48, 15, 16, 24, 7>), GRID, dim3(256), 0, ST, __VA_ARGS__);   \
        else if (h == 24 && w == 7 && c == 32 && ho == 12 && wo == 5) hipLaunchKernelGGL((KERNEL<24, 7, 32, 12, 5>), GRID, dim3(256), 0, ST, __VA_ARGS__); \
        else if (h == 12 && w == 5 && c == 64 && ho == 12 && wo == 5) hipLaunchKernelGGL((KERNEL<12, 5, 64, 12, 5>), GRID, dim3(256), 0, ST, __VA_ARGS__); \
        else if (h == 12 && w == 5 && c == 64 && ho == 6 && wo == 3) hipLaunchKernelGGL((KERNEL<12, 5, 64, 6, 3>), GRID, dim3(256), 0, ST, __VA_ARGS__);   \
        else if (h == 6 && w == 3 && c == 64 && ho == 6 && wo == 3) hipLaunchKernelGGL((KERNEL<6, 3, 64, 6, 3>), GRID, dim3(256), 0, ST, __VA_ARGS__);     \
        else if (h == 6 && w == 1 && c == 64 && ho == 6 && wo == 1) hipLaunchKernelGGL((KERNEL<6, 1, 64, 6, 1>), GRID, dim3(256), 0, ST, __VA_ARGS__);     \
        else hipLaunchKernelGGL((KERNEL<0, 0, 0, 0, 0>), GRID, dim3(256), 0, ST, __VA_ARGS__);                     \
    } while (0)

extern "C" int nisqa_bn_act_pool_fwd(const float* z, const double* sums, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float* mean_rstd, int32_t n_segments,
                                     int32_t h, int32_t w, int32_t c, int32_t ho, int32_t wo, const float* drop, float* y,
                                     int32_t* arg, void* stream) {
    if (!z || !sums || !gamma || !beta || !running_mean || !running_var || !mean_rstd || !y || !arg || n_segments <= 0 ||
        h <= 0 || w <= 0 || c <= 0 || c > 256 || (c & 3) || ho <= 0 || wo <= 0 || ho > h || wo > w)
        return NISQA_ERR_ARG;
    const int64_t total = (int64_t)n_segments * ho * wo * c / 4;
    NQ_LAUNCH_BEGIN();
    NQ_POOL_DISPATCH(bn_act_pool_fwd_kernel, dim3(grid_resident(total)), (hipStream_t)stream, z, gamma, beta, sums, running_mean,
                     running_var, mean_rstd, total, h, w, c, ho, wo, drop, y, arg);
    return NQ_LAUNCH_STATUS();
}

// Grid-stride kernels over [rows][c] in which a thread owns four consecutive channels and 1024 % c == 0 always meet the
// SAME four channels (4 * (tid % (c/4))): eight float64 partial sums per thread are reduced over the block in LDS and
// added to out[0..c) / out[c..2c) with atomics.
NQ_DEV void block_channel_sums4(const double (&v1)[4], const double (&v2)[4], int c, double* __restrict__ out) {
    __shared__ double r1[1024], r2[1024];
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r1[4 * tid + e] = v1[e]; r2[4 * tid + e] = v2[e]; }
    __syncthreads();
    if (tid < c) {                                         // channel tid lives at positions tid, tid + c, ... of the 1024
        double t1 = 0.0, t2 = 0.0;
        for (int q = tid; q < 1024; q += c) { t1 += r1[q]; t2 += r2[q]; }
        atomicAdd(out + tid, t1);
        atomicAdd(out + c + tid, t2);
    }
}

template <int H, int W, int C, int HO, int WO>
__global__ __launch_bounds__(256) void bn_act_pool_bwd1_kernel(
    const float* __restrict__ dy, const int32_t* __restrict__ arg, const float* __restrict__ drop,
    const float* __restrict__ z, const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, int64_t total4, int h_, int w_, int c_, int ho_, int wo_, float* __restrict__ dyb,
    double* __restrict__ sums2) {
    const int h = H ? H : h_, w = H ? W : w_, c = H ? C : c_, ho = H ? HO : ho_, wo = H ? WO : wo_;
    const int c4 = c / 4;
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};     // sums2 != NULL: sum(dyb), sum(dyb * z)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int ch = 4 * (int)(i % c4);
        const int64_t pix = i / c4;
        const int64_t s = pix / (h * w);
        const int p = (int)(pix - s * (h * w));
        const int yy = p / w, xx = p % w;
        f32x4 g, bsh;
        bn_affine4(gamma, beta, mean_rstd, c, ch, g, bsh);
        const f32x4 zi = ((const f32x4*)z)[i];
        const f32x4 yb = zi * g + bsh;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bool live = yb[0] > 0.f || yb[1] > 0.f || yb[2] > 0.f || yb[3] > 0.f;
        if (h == ho && w == wo) {                           // identity pooling: the only window of pixel p is p itself
            acc = ((const f32x4*)dy)[i];
        } else if (live) {
            const int oy0 = max(0, (yy * ho) / h - 1), oy1 = min(ho - 1, ((yy + 1) * ho) / h + 1);
            const int ox0 = max(0, (xx * wo) / w - 1), ox1 = min(wo - 1, ((xx + 1) * wo) / w + 1);
            for (int oy = oy0; oy <= oy1; ++oy) {
                if (yy < win_lo(oy, h, ho) || yy >= win_hi(oy, h, ho)) continue;
                for (int ox = ox0; ox <= ox1; ++ox) {
                    if (xx < win_lo(ox, w, wo) || xx >= win_hi(ox, w, wo)) continue;
                    const int64_t o = (s * (ho * wo) + oy * wo + ox) * c + ch;
                    const i32x4 ag = *(const i32x4*)(arg + o);
                    const f32x4 d = *(const f32x4*)(dy + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (ag[e] == p) acc[e] += d[e];
                }
            }
        }
        if (drop) acc *= *(const f32x4*)(drop + s * c + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!(yb[e] > 0.f)) acc[e] = 0.f;               // ReLU gate
            a1[e] += (double)acc[e];
            a2[e] += (double)acc[e] * (double)zi[e];
        }
        ((f32x4*)dyb)[i] = acc;
    }
    if (sums2) block_channel_sums4(a1, a2, c, sums2);
}

extern "C" int nisqa_bn_act_pool_bwd1(const float* dy, const int32_t* arg, const float* drop, const float* z,
                                      const float* mean_rstd, const float* gamma, const float* beta, int32_t n_segments,
                                      int32_t h, int32_t w, int32_t c, int32_t ho, int32_t wo, float* dyb, double* sums2_opt,
                                      void* stream) {
    if (!dy || !arg || !z || !mean_rstd || !gamma || !beta || !dyb || n_segments <= 0 || h <= 0 || w <= 0 || c <= 0 ||
        ho <= 0 || wo <= 0 || ho > h || wo > w || (c & 3) || (sums2_opt && (1024 % c) != 0))
        return NISQA_ERR_ARG;
    const int64_t total = (int64_t)n_segments * h * w * c / 4;
    NQ_LAUNCH_BEGIN();
    // with the reductions riding along every block ends in 2c float64 atomics on the same addresses: 2048 blocks fill the
    // chip (8 per CU) and keep that tail short
    const int grid = sums2_opt ? (grid_for(total) < 2048 ? grid_for(total) : 2048) : grid_for(total);
    NQ_POOL_DISPATCH(bn_act_pool_bwd1_kernel, dim3(grid), (hipStream_t)stream, dy, arg, drop, z, mean_rstd, gamma,
                     beta, total, h, w, c, ho, wo, dyb, sums2_opt);
    return NQ_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------------
// The same backward in two launches that touch the dense tensors ONCE (read z, write dz) instead of twice each way:
// the reductions of the BatchNorm backward only see pixels that won a pooling window, so they are sums over the POOLED
// values (one z gather per value); with them known, dz follows from dy / arg / z in a single dense pass.
// ---------------------------------------------------------------------------------------------------------
template <int H, int W, int C, int HO, int WO>
__global__ __launch_bounds__(256) void pool_bwd_sums_kernel(
    const float* __restrict__ dy, const int32_t* __restrict__ arg, const float* __restrict__ drop, const float* __restrict__ z,
    const float* __restrict__ mean_rstd, const float* __restrict__ gamma, const float* __restrict__ beta, int64_t total4,
    int h_, int w_, int c_, int ho_, int wo_, double* __restrict__ sums2) {
    const int h = H ? H : h_, w = H ? W : w_, c = H ? C : c_, ho = H ? HO : ho_, wo = H ? WO : wo_;
    const int c4 = c / 4;
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
    constexpr bool INV = H != 0 && 256 % ((C ? C : 4) / 4) == 0;        // see bn_act_pool_fwd_kernel
    f32x4 g, bsh;
    if (INV) bn_affine4(gamma, beta, mean_rstd, c, 4 * ((int)threadIdx.x % c4), g, bsh);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int ch = 4 * (int)(i % c4);
        const int64_t s = (i / c4) / (ho * wo);
        if (!INV) bn_affine4(gamma, beta, mean_rstd, c, ch, g, bsh);
        f32x4 d = ((const f32x4*)dy)[i];
        if (drop) d *= *(const f32x4*)(drop + s * c + ch);
        f32x4 zw;
        if (h == ho && w == wo) zw = ((const f32x4*)z)[i];  // identity pooling: the winning pixel is the cell itself
        else {
            const i32x4 ap = ((const i32x4*)arg)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) zw[e] = z[(s * (h * w) + ap[e]) * c + ch + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float zv = zw[e];
            asm volatile("" : "+v"(zv));                    // keep the four gates scalar: packed, hipcc emits the op_sel form of DESIGN.md 7.1
            if (fmaf(zv, g[e], bsh[e]) > 0.f) {             // ReLU gate at the winning pixel
                a1[e] += (double)d[e];
                a2[e] += (double)d[e] * (double)zv;
            }
        }
    }
    block_channel_sums4(a1, a2, c, sums2);
}

template <int H, int W, int C, int HO, int WO>
__global__ __launch_bounds__(256) void bn_act_pool_bwd_dense_kernel(
    const float* __restrict__ dy, const int32_t* __restrict__ arg, const float* __restrict__ drop, const float* __restrict__ z,
    const float* __restrict__ mean_rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const double* __restrict__ sums2, int64_t total4, int h_, int w_, int c_, int ho_, int wo_, float* __restrict__ dz,
    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int h = H ? H : h_, w = H ? W : w_, c = H ? C : c_, ho = H ? HO : ho_, wo = H ? WO : wo_;
    const int c4 = c / 4;
    const double inv = 1.0 / ((double)(total4 / c4));
    if (blockIdx.x == 0 && (int)threadIdx.x < c) {
        const int ch = threadIdx.x;
        const double mean = mean_rstd[ch], rstd = mean_rstd[c + ch];
        dbeta[ch] = (float)sums2[ch];
        dgamma[ch] = (float)(rstd * (sums2[c + ch] - mean * sums2[ch]));
    }
    constexpr bool INV = H != 0 && 256 % ((C ? C : 4) / 4) == 0;        // see bn_act_pool_fwd_kernel
    f32x4 g, bsh, mean, rstd, m1, m2;
    auto channel_constants = [&](int ch) {                  // eight float64 operations per channel: once per thread where INV
        bn_affine4(gamma, beta, mean_rstd, c, ch, g, bsh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mean[e] = mean_rstd[ch + e];
            rstd[e] = mean_rstd[c + ch + e];
            m1[e] = (float)(sums2[ch + e] * inv);
            m2[e] = (float)((double)rstd[e] * (sums2[c + ch + e] - (double)mean[e] * sums2[ch + e]) * inv);
        }
    };
    if (INV) channel_constants(4 * ((int)threadIdx.x % c4));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int ch = 4 * (int)(i % c4);
        const int64_t pix = i / c4;
        const int64_t s = pix / (h * w);
        const int p = (int)(pix - s * (h * w));
        const int yy = p / w, xx = p % w;
        if (!INV) channel_constants(ch);
        const f32x4 zi = ((const f32x4*)z)[i];
        const f32x4 yb = zi * g + bsh;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bool live = yb[0] > 0.f || yb[1] > 0.f || yb[2] > 0.f || yb[3] > 0.f;
        if (h == ho && w == wo) {
            acc = *(const f32x4*)(dy + (s * (ho * wo) + p) * c + ch);
        } else if (live) {
            const int oy0 = max(0, (yy * ho) / h - 1), oy1 = min(ho - 1, ((yy + 1) * ho) / h + 1);
            const int ox0 = max(0, (xx * wo) / w - 1), ox1 = min(wo - 1, ((xx + 1) * wo) / w + 1);
            for (int oy = oy0; oy <= oy1; ++oy) {
                if (yy < win_lo(oy, h, ho) || yy >= win_hi(oy, h, ho)) continue;
                for (int ox = ox0; ox <= ox1; ++ox) {
                    if (xx < win_lo(ox, w, wo) || xx >= win_hi(ox, w, wo)) continue;
                    const int64_t o = (s * (ho * wo) + oy * wo + ox) * c + ch;
                    const i32x4 ag = *(const i32x4*)(arg + o);
                    const f32x4 d = *(const f32x4*)(dy + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (ag[e] == p) acc[e] += d[e];
                }
            }
        }
        if (drop) acc *= *(const f32x4*)(drop + s * c + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (!(yb[e] > 0.f)) acc[e] = 0.f;
        const f32x4 xh = (zi - mean) * rstd;
        ((f32x4*)dz)[i] = (g) * (acc - m1 - xh * m2);        // g = gamma * rstd
    }
}

// dy [S][HO*WO][C] -> dz [S][H*W][C], dgamma, dbeta; sums2 [2c] float64 scratch zeroed by the caller
extern "C" int nisqa_bn_act_pool_bwd(const float* dy, const int32_t* arg, const float* drop, const float* z, const float* mean_rstd,
                                     const float* gamma, const float* beta, int32_t n_segments, int32_t h, int32_t w, int32_t c,
                                     int32_t ho, int32_t wo, double* sums2, float* dz, float* dgamma, float* dbeta, void* stream) {
    if (!dy || !arg || !z || !mean_rstd || !gamma || !beta || !sums2 || !dz || !dgamma || !dbeta || n_segments <= 0 || h <= 0 ||
        w <= 0 || c <= 0 || ho <= 0 || wo <= 0 || ho > h || wo > w || (c & 3) || (1024 % c) != 0)
        return NISQA_ERR_ARG;
    const int64_t cells = (int64_t)n_segments * ho * wo * c / 4, total = (int64_t)n_segments * h * w * c / 4;
    hipStream_t st = (hipStream_t)stream;
    NQ_LAUNCH_BEGIN();
    // the sums kernel ends in 2c float64 atomics per block on the same addresses: blocks of >= 2048 items, at most 1024 of them
    int64_t g1 = (cells + 2047) / 2048;
    g1 = g1 < 1 ? 1 : (g1 > 1024 ? 1024 : g1);
    NQ_POOL_DISPATCH(pool_bwd_sums_kernel, dim3((unsigned)g1), st, dy, arg, drop, z, mean_rstd, gamma, beta, cells, h, w, c, ho, wo, sums2);
    NQ_POOL_DISPATCH(bn_act_pool_bwd_dense_kernel, dim3(grid_resident(total)), st, dy, arg, drop, z, mean_rstd, gamma, beta,
                     (const double*)sums2, total, h, w, c, ho, wo, dz, dgamma, dbeta);
    return NQ_LAUNCH_STATUS();
}

// only the first of the two launches above: sums2 [2c] += sum dyb, sum dyb z over the pooled values -- for callers that fold the
// dense pass into a consumer of dz (nisqa_segconv_wgrad_bn_bf16)
extern "C" int nisqa_bn_pool_bwd_sums(const float* dy, const int32_t* arg, const float* drop, const float* z, const float* mean_rstd,
                                      const float* gamma, const float* beta, int32_t n_segments, int32_t h, int32_t w, int32_t c,
                                      int32_t ho, int32_t wo, double* sums2, void* stream) {
    if (!dy || !arg || !z || !mean_rstd || !gamma || !beta || !sums2 || n_segments <= 0 || h <= 0 || w <= 0 || c <= 0 || ho <= 0 ||
        wo <= 0 || ho > h || wo > w || (c & 3) || (1024 % c) != 0)
        return NISQA_ERR_ARG;
    const int64_t cells = (int64_t)n_segments * ho * wo * c / 4;
    hipStream_t st = (hipStream_t)stream;
    NQ_LAUNCH_BEGIN();
    int64_t g1 = (cells + 2047) / 2048;
    g1 = g1 < 1 ? 1 : (g1 > 1024 ? 1024 : g1);
    NQ_POOL_DISPATCH(pool_bwd_sums_kernel, dim3((unsigned)g1), st, dy, arg, drop, z, mean_rstd, gamma, beta, cells, h, w, c, ho, wo, sums2);
    return NQ_LAUNCH_STATUS();
}

__global__ __launch_bounds__(256) void bn_bwd2_kernel(float* __restrict__ d, const float* __restrict__ z,
                                                      const double* __restrict__ sums2, const float* __restrict__ mean_rstd,
                                                      const float* __restrict__ gamma, int64_t rows, int c,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      double* __restrict__ sum_dz) {
    if (blockIdx.x == 0 && (int)threadIdx.x < c) {
        const int ch = threadIdx.x;
        const double mean = mean_rstd[ch], rstd = mean_rstd[c + ch];
        dbeta[ch] = (float)sums2[ch];
        dgamma[ch] = (float)(rstd * (sums2[c + ch] - mean * sums2[ch]));
    }
    const int c4 = c / 4;
    const int64_t total4 = rows * c4;
    const double inv = 1.0 / (double)rows;
    // this thread's four channels never change (grid stride 256 * gridDim is a multiple of c / 4): per-channel constants once
    const int ch = 4 * (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % c4);
    f32x4 mean, rstd, m1, m2, gr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        mean[e] = mean_rstd[ch + e];
        rstd[e] = mean_rstd[c + ch + e];
        m1[e] = (float)(sums2[ch + e] * inv);                                                            // mean(dyb)
        m2[e] = (float)((double)rstd[e] * (sums2[c + ch + e] - (double)mean[e] * sums2[ch + e]) * inv);   // mean(dyb * xhat)
        gr[e] = gamma[ch + e] * rstd[e];
    }
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a0[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const f32x4 xh = (((const f32x4*)z)[i] - mean) * rstd;
        const f32x4 o = gr * (((f32x4*)d)[i] - m1 - xh * m2);
        ((f32x4*)d)[i] = o;
#pragma unroll
        for (int e = 0; e < 4; ++e) a1[e] += (double)o[e];
    }
    if (sum_dz) block_channel_sums4(a1, a0, c, sum_dz);    // column sums of dz = the conv bias gradient (second half unused)
}

extern "C" int nisqa_bn_bwd2(float* dyb_to_dz, const float* z, const double* sums2, const float* mean_rstd,
                             const float* gamma, int64_t rows, int32_t c, float* dgamma, float* dbeta, double* sum_dz_opt,
                             void* stream) {
    if (!dyb_to_dz || !z || !sums2 || !mean_rstd || !gamma || !dgamma || !dbeta || rows <= 0 || c <= 0 || c > 256 ||
        (1024 % c) != 0)
        return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    // gridDim * 256 must be a multiple of c / 4 (the kernel hoists its per-channel constants): any gridDim is, as c / 4 | 256
    const int grid = grid_for(rows * c / 4) < 2048 ? grid_for(rows * c / 4) : 2048;
    hipLaunchKernelGGL(bn_bwd2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dyb_to_dz, z, sums2,
                       mean_rstd, gamma, rows, c, dgamma, dbeta, sum_dz_opt);
    return NQ_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm over rows of 64: one wave per row
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int64_t rows,
                                                            float* __restrict__ y, float* __restrict__ xhat,
                                                            float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float v = x[r * 64 + lane];
    const float mean = wave_sum(v) * (1.0f / 64.0f);
    const float dv = v - mean;
    const float var = wave_sum(dv * dv) * (1.0f / 64.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float xh = dv * rstd;
    xhat[r * 64 + lane] = xh;
    y[r * 64 + lane] = fmaf(xh, gamma[lane], beta[lane]);
    if (lane == 0) rstd_out[r] = rstd;
}

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            int64_t rows, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float g = dy[r * 64 + lane] * gamma[lane], xh = xhat[r * 64 + lane];
    const float m1 = wave_sum(g) * (1.0f / 64.0f), m2 = wave_sum(g * xh) * (1.0f / 64.0f);
    dx[r * 64 + lane] = rstd[r] * (g - m1 - xh * m2);
}

extern "C" int nisqa_layernorm_fwd(const float* x, const float* gamma, const float* beta, int64_t rows, float* y,
                                   float* xhat, float* rstd, void* stream) {
    if (!x || !gamma || !beta || !y || !xhat || !rstd || rows <= 0) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                       rows, y, xhat, rstd);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, int64_t rows,
                                   float* dx, void* stream) {
    if (!dy || !xhat || !rstd || !gamma || !dx || rows <= 0) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, xhat, rstd,
                       gamma, rows, dx);
    return NQ_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------------
// ragged row softmax: one wave per row
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ off,
                                                               const int32_t* __restrict__ len, int64_t rows, float scale,
                                                               float* __restrict__ p) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + off[r];
    float* pr = p + off[r];
    const int n = len[r];
    float mx = -3.0e38f;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, xr[j] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) sum += expf(xr[j] * scale - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < n; j += 64) pr[j] = expf(xr[j] * scale - mx) * inv;
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                               const int64_t* __restrict__ off, const int32_t* __restrict__ len,
                                                               int64_t rows, float scale, float* __restrict__ ds) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* pr = p + off[r];
    const float* dr = dp + off[r];
    float* sr = ds + off[r];
    const int n = len[r];
    float dot = 0.f;
    for (int j = lane; j < n; j += 64) dot += pr[j] * dr[j];
    dot = wave_sum(dot);
    for (int j = lane; j < n; j += 64) sr[j] = scale * pr[j] * (dr[j] - dot);
}

extern "C" int nisqa_softmax_rows_fwd(const float* x, const int64_t* off, const int32_t* len, int64_t rows, float scale,
                                      float* p, void* stream) {
    if (!x || !off || !len || !p || rows <= 0) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, off, len,
                       rows, scale, p);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_softmax_rows_bwd(const float* p, const float* dp, const int64_t* off, const int32_t* len, int64_t rows,
                                      float scale, float* ds, void* stream) {
    if (!p || !dp || !off || !len || !ds || rows <= 0) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, dp, off,
                       len, rows, scale, ds);
    return NQ_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------------
// elementwise, loss, optimiser
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void elementwise_kernel(int op, const float* __restrict__ x, const float* __restrict__ aux,
                                                          const float* __restrict__ bias, int64_t total, int cols,
                                                          float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        float o;
        switch (op) {
            case 0: o = v + bias[i % cols]; break;
            case 1: o = fmaxf(v + bias[i % cols], 0.f); break;
            case 2: o = aux[i] > 0.f ? v : 0.f; break;
            case 3: o = v * aux[i]; break;
            case 4: o = v + aux[i]; break;
            default: o = v * bias[i % cols]; break;
        }
        y[i] = o;
    }
}

extern "C" int nisqa_elementwise(int32_t op, const float* x, const float* aux, const float* bias, int64_t rows,
                                 int32_t cols, float* y, void* stream) {
    if (op < 0 || op > 5 || !x || !y || rows <= 0 || cols <= 0) return NISQA_ERR_ARG;
    if ((op == 0 || op == 1 || op == 5) && !bias) return NISQA_ERR_ARG;
    if ((op == 2 || op == 3 || op == 4) && !aux) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(elementwise_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, op, x, aux, bias,
                       rows * cols, cols, y);
    return NQ_LAUNCH_STATUS();
}

__global__ __launch_bounds__(64) void mse_loss_kernel(const float* __restrict__ y_hat, const float* __restrict__ y,
                                                      const float* __restrict__ bias, int n_clips, int n_heads,
                                                      float* __restrict__ loss, float* __restrict__ dy_hat) {
    __shared__ float part[64];
    const int hd = threadIdx.x;
    float l = 0.f;
    if (hd < n_heads) {
        int cnt = 0;
        for (int b = 0; b < n_clips; ++b) cnt += !isnan(y[b * n_heads + hd]);
        const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
        for (int b = 0; b < n_clips; ++b) {
            const float t = y[b * n_heads + hd], v = y_hat[b * n_heads + hd];
            float mapped = v, slope = 1.f;
            if (bias) {
                const float* q = bias + b * 4;
                mapped = q[0] + v * (q[1] + v * (q[2] + v * q[3]));
                slope = q[1] + v * (2.f * q[2] + 3.f * v * q[3]);
            }
            float g = 0.f;
            if (!isnan(t)) {
                const float e = mapped - t;
                l += e * e * inv;
                g = 2.f * e * inv * slope;
            }
            dy_hat[b * n_heads + hd] = g;
        }
    }
    part[hd] = l;
    if (hd < n_heads) loss[1 + hd] = l;
    __syncthreads();
    if (hd == 0) {
        float s = 0.f;
        for (int q = 0; q < n_heads; ++q) s += part[q];
        loss[0] = s;
    }
}

extern "C" int nisqa_mse_loss(const float* y_hat, const float* y, const float* bias, int32_t n_clips, int32_t n_heads,
                              float* loss, float* dy_hat, void* stream) {
    if (!y_hat || !y || !loss || !dy_hat || n_clips <= 0 || n_heads <= 0 || n_heads > 64) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, y_hat, y, bias, n_clips, n_heads, loss,
                       dy_hat);
    return NQ_LAUNCH_STATUS();
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float step_size, float bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = 0.9f * m[i] + 0.1f * gi;
        const float vi = 0.999f * v[i] + 0.001f * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) / bc2_sqrt + 1e-8f);
    }
}

// ---------------------------------------------------------------------------------------------------------
// dropout multipliers from a counter-based generator (Philox-4x32-10): value i of the stream (seed, offset) does not
// depend on the launch geometry, so a step's masks are reproducible from (seed, offset) alone
// ---------------------------------------------------------------------------------------------------------
NQ_DEV void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint64_t seed, uint64_t offset, float p, float keep_scale, int64_t n,
                                                           float* __restrict__ out) {
    const int64_t quads = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (int64_t)gridDim.x * 256) {
        const uint64_t ctr = offset + (uint64_t)q;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t i = 4 * q + e;
            if (i < n) out[i] = ((float)c[e] * 2.3283064365386963e-10f >= p) ? keep_scale : 0.f;      // u in [0, 1)
        }
    }
}

extern "C" int nisqa_dropout_mask(uint64_t seed, uint64_t offset, float p, int64_t n, float* out, void* stream) {
    if (!out || n <= 0 || !(p >= 0.f) || !(p < 1.f)) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, seed, offset, p,
                       1.0f / (1.0f - p), n, out);
    return NQ_LAUNCH_STATUS();
}

// float64 reduction results -> float32 gradient slots, all pending copies of a step in one launch
__global__ __launch_bounds__(256) void cast_scatter_kernel(const double* __restrict__ src, const int32_t* __restrict__ table,
                                                           float* __restrict__ dst) {
    const int32_t* e = table + 3 * blockIdx.x;
    for (int t = threadIdx.x; t < e[2]; t += 256) dst[e[1] + t] = (float)src[e[0] + t];
}

extern "C" int nisqa_cast_scatter(const double* src, const int32_t* table, int32_t n_entries, float* dst, void* stream) {
    if (!src || !table || !dst || n_entries <= 0) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(cast_scatter_kernel, dim3(n_entries), dim3(256), 0, (hipStream_t)stream, src, table, dst);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr, int32_t t,
                               void* stream) {
    if (!param || !grad || !m || !v || n <= 0 || t < 1) return NISQA_ERR_ARG;
    const double bc1 = 1.0 - pow(0.9, (double)t), bc2 = 1.0 - pow(0.999, (double)t);
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n,
                       (float)(lr / bc1), (float)sqrt(bc2));
    return NQ_LAUNCH_STATUS();
}
