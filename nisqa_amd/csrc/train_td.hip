// Self-attention block + attention-pooling heads + loss of the TRAINING step, forward and backward, as a dozen
// launches (include/nisqa_train.h "nisqa_tdtrain_*"; SURVEY.md section 8f-3, BASELINE config 5).  Replaces what the
// reference runs per batch in SelfAttention.forward / SelfAttentionLayer.forward (nisqa/NISQA_lib.py:988-996,
// 1025-1040), PoolAttFF.forward (NISQA_lib.py:1171-1183), biasLoss.get_loss (NISQA_lib.py:1880-1892, 1946-1950) and in
// autograd's backward of all of them (NISQA_model.py:142-143).  Round 3 drove this part of the step as 116 launches
// (49 grouped 64 x 64 GEMMs, 30 element-wise kernels, 15 column sums, ...) that took 0.9 ms of a 3.9 ms step for
// 7 GFLOP of work; here it is
//   tdt_pack            weight fragments of the step (forward order and transposed), once
//   tdt_proj_fwd        Linear 384->64 + LayerNorm + layer-0 QKV                               per 32-token tile
//   tdt_layer_fwd  x L  attention (flash-style, dropout on the probabilities inside) + out-proj + dropout + residual
//                       + LayerNorm + FFN (ReLU, dropouts) + residual + LayerNorm + next QKV (or the pooling heads'
//                       hidden layer and scores)                                                per 32-token tile
//   tdt_pool_clip       per (clip, head): masked softmax, pooled vector, y_hat, the loss term and its gradient, and
//                       straight on into the backward of the pooling (d score, d pooled, linear3 gradients)
//   tdt_bwd_tail        pooling heads' token-wise backward + the token-wise backward of the last layer down to d ctx
//   tdt_attn_bwd   x L  recomputes P from q, k and the saved log-sum-exp: d q per query tile, d k / d v per key tile
//   tdt_bwd_mid         d qkv -> d x of the layer below, then that layer's token-wise backward (or, below layer 0,
//                       LayerNorm + the 64->384 input gradient for the CNN)
//   nisqa_gemm_f32      ONE grouped split-K launch for every weight gradient of the block (d Y^T X)
//   tdt_colsum          ONE launch for every bias / LayerNorm-parameter gradient (column sums)
// Same "features x tokens" register chaining as the inference kernels (td.hip): a wave owns 32 tokens (MFMA columns),
// keeps their 64 features in registers (rows), weights are the A operand as pre-packed fragments, the D fragment of one
// product is the B operand of the next.  v_mfma_f32_32x32x2_f32 throughout: exact fp32, the reference's arithmetic.
// Tokens live in a PADDED space inside the block (32 per tile, clips start on tile boundaries; rows of padding tokens
// hold zeros) so that attention tiles never straddle clips and feature-major operands load as aligned float4s.
#include "common.hpp"
#include "../../include/nisqa_hip.h"
#include "../../include/nisqa_train.h"

#define LN_EPS 1e-5f
#define TDT_MAX_LAYERS 4
#define TDT_MAX_HEADS 8

// ---- D-layout helpers (lane l: token j = l & 31, half hf = l >> 5; register r of tile mt <-> feature 32 mt + DROW(r, hf)) ----
template <int MT>
NQ_DEV void ld_vec(const float* __restrict__ base, f32x16 (&out)[MT], int hf) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4*)(base + 32 * mt + 8 * g + 4 * hf);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[mt][4 * g + e] = v[e];
        }
}
NQ_DEV f32x16 ld_row16(const float* __restrict__ base, int hf) {      // 32 consecutive row values -> the D rows of this lane half
    f32x16 o[1];
    ld_vec<1>(base, o, hf);
    return o[0];
}
template <int MT>
NQ_DEV void st_vec(float* __restrict__ rowp, const f32x16 (&v)[MT], int hf, float scale) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[mt][4 * g + e] * scale;
            *(f32x4*)(rowp + 32 * mt + 8 * g + 4 * hf) = o;
        }
}
// feature-major copy: base[feature][np tokens]
NQ_DEV void st_fm(float* __restrict__ base, int np, int tok, const f32x16 (&v)[2], int hf, float scale) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) base[(size_t)(32 * mt + NQ_DROW(r, hf)) * np + tok] = v[mt][r] * scale;
}
template <int MT>
NQ_DEV void zero_t(f32x16 (&v)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) v[mt] = zero16();
}
template <int MT>
NQ_DEV void mul_t(f32x16 (&v)[MT], const f32x16 (&m)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[mt][r] *= m[mt][r];
}

// out[mt] += A (fragments, [4 KT steps][MT][64 lanes][4]) * in   (in: KT tiles of 32 features x 32 tokens in D layout).
// One wave per SIMD runs these kernels (a chain of dependent products per 32-token tile), so nobody else hides the L2 round
// trip of a fragment: they are requested two steps ahead into a three-slot ring, held in place by scheduling fences.
template <int KT, int MT>
NQ_DEV void chain_gemm(const f32x4* __restrict__ af, const f32x16 (&in)[KT], f32x16 (&out)[MT], int lane) {
    constexpr int STEPS = 4 * KT;
    f32x4 a[3][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        a[0][mt] = af[mt * 64 + lane];
        a[1][mt] = af[(MT + mt) * 64 + lane];
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s + 2 < STEPS) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[(s + 2) % 3][mt] = af[((s + 2) * MT + mt) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) out[mt] = mfma32(a[s % 3][mt][kk], in[s >> 2][4 * (s & 3) + kk], out[mt]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));   // four consecutive floats at any 4-byte boundary

// the 16 dropout multipliers of this lane's row for the 32 columns col0 .. col0 + 31 of a [.. x n] mask, in D-row order
// (register 4 g + e <-> column 8 g + 4 hf + e): four 16-byte reads; in a row's last tile (columns beyond n) element-wise, 1 there
NQ_DEV f32x16 ld_mask_row(const float* __restrict__ row, int col0, int n, int hf) {
    f32x16 m;
    if (col0 + 32 <= n) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_u v = *(const f32x4_u*)(row + col0 + 8 * g + 4 * hf);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[4 * g + e] = v[e];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = col0 + NQ_DROW(r, hf);
            m[r] = col < n ? row[col] : 1.f;
        }
    }
    return m;
}

// LayerNorm over the 64 features of a token (32 in this lane, 32 in lane ^ 32); x -> gamma * xhat + beta, xhat and rstd kept
NQ_DEV void ln_fwd(f32x16 (&x)[2], const float* __restrict__ gamma, const float* __restrict__ beta, int hf, f32x16 (&xh)[2],
                   float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += x[mt][r];
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = x[mt][r] - mean;
            q = fmaf(d, d, q);
        }
    q += __shfl_xor(q, 32);
    rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + LN_EPS);
    f32x16 g[2], bt[2];
    ld_vec<2>(gamma, g, hf);
    ld_vec<2>(beta, bt, hf);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            xh[mt][r] = (x[mt][r] - mean) * rstd;
            x[mt][r] = xh[mt][r] * g[mt][r] + bt[mt][r];
        }
}
// d: d loss / d (LayerNorm output) in, d loss / d (LayerNorm input) out: rstd * (g - mean(g) - xhat * mean(g * xhat)), g = d * gamma
NQ_DEV void ln_bwd(f32x16 (&d)[2], const f32x16 (&xh)[2], float rstd, const float* __restrict__ gamma, int hf) {
    f32x16 g[2];
    ld_vec<2>(gamma, g, hf);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            g[mt][r] *= d[mt][r];
            s1 += g[mt][r];
            s2 = fmaf(g[mt][r], xh[mt][r], s2);
        }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    const float m1 = s1 * (1.0f / 64.0f), m2 = s2 * (1.0f / 64.0f);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[mt][r] = rstd * (g[mt][r] - m1 - xh[mt][r] * m2);
}

// ---- kernel arguments ------------------------------------------------------------------------------------------
struct tdt_common {
    const int32_t* seg_off;      // [B + 1] tokens (segments) per clip, exclusive prefix sum: the caller's unpadded token space
    const int32_t* ptok_off;     // [B + 1] the same in the padded space (multiples of 32)
    const int32_t* tile_clip;    // [np / 32] clip of every 32-token tile
    const int64_t* sq_off;       // [B + 1] prefix sum of L^2: the attention-probability dropout masks are [sum L^2], row-major per clip
    int np;                      // padded tokens
};
struct tdt_layer_p {
    const float *f_qkv, *f_out, *f_ff1, *f_ff2, *t_qkv, *t_out, *t_ff1, *t_ff2;     // fragments: forward order, transposed
    const float *b_qkv, *b_out, *g1, *be1, *b_ff1, *b_ff2, *g2, *be2;               // vectors in the flat parameter buffer
    float *qs, *qsT, *k, *kT, *v, *vT, *lse, *ctx, *x1, *xh1, *rs1, *hd, *xh2, *rs2;  // kept by the forward pass
    float *dx, *df, *dh, *dx1, *datt, *dctx, *dctxT, *dd, *dr1, *dqkv, *dqkvp;       // written by the backward pass (dqkvp: shares 1.. of d qkv)
    const float *mP, *m1, *mf, *m2;                                                  // dropout multipliers (NULL: none)
};
struct tdt_head_p {
    const float *f_p1, *t_p1, *b1, *w2, *b2, *w3, *b3;
    float *u, *sc, *att, *dsc, *du, *dpooled;
    float *g_w3, *g_b3, *g_b2;                                                       // gradients written by tdt_pool_clip
};
struct tdt_heads { tdt_head_p h[TDT_MAX_HEADS]; int n; };

// q (pre-scaled by 1/sqrt(64)), k, v of the next attention: token-major and feature-major copies
NQ_DEV void qkv_store(const tdt_layer_p& L, const f32x16 (&x)[2], int ptok, int np, int lane, float vm) {
    const int hf = lane >> 5;
    f32x16 acc[6];
    ld_vec<6>(L.b_qkv, acc, hf);
    chain_gemm<2, 6>((const f32x4*)L.f_qkv, x, acc, lane);
    f32x16 t2[2];
    t2[0] = acc[0]; t2[1] = acc[1];
    st_vec<2>(L.qs + (size_t)ptok * 64, t2, hf, 0.125f * vm);
    st_fm(L.qsT, np, ptok, t2, hf, 0.125f * vm);
    t2[0] = acc[2]; t2[1] = acc[3];
    st_vec<2>(L.k + (size_t)ptok * 64, t2, hf, vm);
    st_fm(L.kT, np, ptok, t2, hf, vm);
    t2[0] = acc[4]; t2[1] = acc[5];
    st_vec<2>(L.v + (size_t)ptok * 64, t2, hf, vm);
    st_fm(L.vT, np, ptok, t2, hf, vm);
}

// ---------------------------------------------------------------------------------------------------------
// Linear 384 -> 64 + LayerNorm (NISQA_lib.py:989-991) + layer-0 QKV
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tdt_proj_fwd_kernel(tdt_common c, const float* __restrict__ feat, const float* __restrict__ f_w0,
                                                          const float* __restrict__ b0, const float* __restrict__ g0,
                                                          const float* __restrict__ be0, float* __restrict__ x0,
                                                          float* __restrict__ xh0, float* __restrict__ rs0, tdt_layer_p L0) {
    const int lane = threadIdx.x, j = lane & 31, hf = lane >> 5;
    const int b = c.tile_clip[blockIdx.x];
    const int n = c.seg_off[b + 1] - c.seg_off[b];
    const int ptok = blockIdx.x * 32 + j, kq = ptok - c.ptok_off[b];
    const bool valid = kq < n;
    const float vm = valid ? 1.f : 0.f;
    const int utok = c.seg_off[b] + (valid ? kq : 0);
    const f32x4* frow = (const f32x4*)(feat + (size_t)utok * 384);
    const f32x4* af = (const f32x4*)f_w0;
    f32x16 acc[2];
    ld_vec<2>(b0, acc, hf);
    // this lane's half of its token's 384 features: 48 independent 16-byte reads requested together (a lane walks its own
    // row, 1.5 KB from its neighbour's: nothing coalesces, so the reads are latency, not bandwidth)
    f32x4 bv[48];
#pragma unroll
    for (int s = 0; s < 48; ++s) {
        bv[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (valid) bv[s] = frow[2 * s + hf];
    }
    f32x4 a[3][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) { a[0][mt] = af[mt * 64 + lane]; a[1][mt] = af[(2 + mt) * 64 + lane]; }
#pragma unroll
    for (int s = 0; s < 48; ++s) {
        if (s + 2 < 48) {
            a[(s + 2) % 3][0] = af[((s + 2) * 2 + 0) * 64 + lane];
            a[(s + 2) % 3][1] = af[((s + 2) * 2 + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc[0] = mfma32(a[s % 3][0][kk], bv[s][kk], acc[0]);
            acc[1] = mfma32(a[s % 3][1][kk], bv[s][kk], acc[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x16 xh[2];
    float rstd;
    ln_fwd(acc, g0, be0, hf, xh, rstd);
    st_vec<2>(xh0 + (size_t)ptok * 64, xh, hf, vm);
    st_vec<2>(x0 + (size_t)ptok * 64, acc, hf, vm);
    if (hf == 0) rs0[ptok] = rstd * vm;
    qkv_store(L0, acc, ptok, c.np, lane, vm);
}

// ---------------------------------------------------------------------------------------------------------
// One SelfAttentionLayer in train mode (NISQA_lib.py:1025-1040) for a 32-query tile, then the next layer's QKV or -- behind
// the last layer -- the pooling heads' hidden layer and scores (NISQA_lib.py:1173-1174)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tdt_layer_fwd_kernel(tdt_common c, tdt_layer_p L, tdt_layer_p Ln, int has_next,
                                                           const float* __restrict__ x_in, float* __restrict__ x_out, tdt_heads hs) {
    const int lane = threadIdx.x, j = lane & 31, hf = lane >> 5;
    const int b = c.tile_clip[blockIdx.x];
    const int n = c.seg_off[b + 1] - c.seg_off[b], c0 = c.ptok_off[b];
    const int ptok = blockIdx.x * 32 + j, kq = ptok - c0;
    const bool valid = kq < n;
    const float vm = valid ? 1.f : 0.f;
    const int utok = c.seg_off[b] + (valid ? kq : 0);
    const int np = c.np;

    f32x4 qf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *(const f32x4*)(L.qs + (size_t)ptok * 64 + 8 * s + 4 * hf);
    const float* mrow = L.mP ? L.mP + c.sq_off[b] + (int64_t)(valid ? kq : 0) * n : nullptr;

    f32x16 o[2];
    zero_t<2>(o);
    float m = -INFINITY, l = 0.f;
    const int nkt = (n + 31) >> 5;
    const bool use_mask = mrow != nullptr && valid;
    // K rows of tile kt + 1 are requested while tile kt is multiplied; V^T and the dropout multipliers of tile kt are
    // requested at its start and used behind its 32 QK^T MFMAs
    f32x4 kA[8], kB[8];
    {
        const float* krow = L.k + (size_t)(c0 + j) * 64 + 4 * hf;
#pragma unroll
        for (int s = 0; s < 8; ++s) kA[s] = *(const f32x4*)(krow + 8 * s);
    }
    auto tile = [&](int kt, const f32x4 (&kcur)[8], f32x4 (&knext)[8]) {
        const int key0 = c0 + 32 * kt;
        f32x4 vf[2][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            vf[0][g] = *(const f32x4*)(L.vT + (size_t)j * np + key0 + 8 * g + 4 * hf);
            vf[1][g] = *(const f32x4*)(L.vT + (size_t)(j + 32) * np + key0 + 8 * g + 4 * hf);
        }
        f32x16 mk;
        if (use_mask) mk = ld_mask_row(mrow, 32 * kt, n, hf);
        if (kt + 1 < nkt) {
            const float* krow = L.k + (size_t)(key0 + 32 + j) * 64 + 4 * hf;
#pragma unroll
            for (int s = 0; s < 8; ++s) knext[s] = *(const f32x4*)(krow + 8 * s);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 sacc = zero16();                                   // S^T tile: rows = keys, columns = queries
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) sacc = mfma32(kcur[s][kk], qf[s][kk], sacc);
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (32 * kt + NQ_DROW(r, hf) >= n) sacc[r] = -INFINITY;   // key_padding_mask (NISQA_lib.py:1028)
            mx = fmaxf(mx, sacc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);                         // finite: key 32 kt is always valid
        const float alpha = expf(m - m_new);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = expf(sacc[r] - m_new);
            rs += sacc[r];
        }
        rs += __shfl_xor(rs, 32);
        l = l * alpha + rs;
        m = m_new;
        if (use_mask) {                                           // dropout on the probabilities (nn.MultiheadAttention)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] *= mk[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                o[0] = mfma32(vf[0][g][kk], sacc[4 * g + kk], o[0]);
                o[1] = mfma32(vf[1][g][kk], sacc[4 * g + kk], o[1]);
            }
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile(kt, kA, kB);
        if (kt + 1 < nkt) tile(kt + 1, kB, kA);
    }
    const float inv_l = 1.0f / l;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= inv_l; o[1][r] *= inv_l; }
    if (hf == 0) L.lse[ptok] = valid ? m + logf(l) : 0.f;
    st_vec<2>(L.ctx + (size_t)ptok * 64, o, hf, vm);

    // out_proj, dropout1, residual, LayerNorm1
    f32x16 y[2], t[2], xh[2];
    float rstd;
    ld_vec<2>(L.b_out, y, hf);
    chain_gemm<2, 2>((const f32x4*)L.f_out, o, y, lane);
    if (L.m1 && valid) {
        ld_vec<2>(L.m1 + (size_t)utok * 64, t, hf);
        mul_t<2>(y, t);
    }
    ld_vec<2>(x_in + (size_t)ptok * 64, t, hf);
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[0][r] += t[0][r]; y[1][r] += t[1][r]; }
    ln_fwd(y, L.g1, L.be1, hf, xh, rstd);
    st_vec<2>(L.xh1 + (size_t)ptok * 64, xh, hf, vm);
    st_vec<2>(L.x1 + (size_t)ptok * 64, y, hf, vm);
    if (hf == 0) L.rs1[ptok] = rstd * vm;
    // FFN: relu(linear1), dropout, linear2, dropout2, residual, LayerNorm2
    f32x16 h1[2], h2[2];
    ld_vec<2>(L.b_ff1, h1, hf);
    chain_gemm<2, 2>((const f32x4*)L.f_ff1, y, h1, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) { h1[0][r] = fmaxf(h1[0][r], 0.f); h1[1][r] = fmaxf(h1[1][r], 0.f); }
    if (L.mf && valid) {
        ld_vec<2>(L.mf + (size_t)utok * 64, t, hf);
        mul_t<2>(h1, t);
    }
    st_vec<2>(L.hd + (size_t)ptok * 64, h1, hf, vm);
    ld_vec<2>(L.b_ff2, h2, hf);
    chain_gemm<2, 2>((const f32x4*)L.f_ff2, h1, h2, lane);
    if (L.m2 && valid) {
        ld_vec<2>(L.m2 + (size_t)utok * 64, t, hf);
        mul_t<2>(h2, t);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[0][r] += h2[0][r]; y[1][r] += h2[1][r]; }
    ln_fwd(y, L.g2, L.be2, hf, xh, rstd);
    st_vec<2>(L.xh2 + (size_t)ptok * 64, xh, hf, vm);
    st_vec<2>(x_out + (size_t)ptok * 64, y, hf, vm);
    if (hf == 0) L.rs2[ptok] = rstd * vm;
    if (has_next) {
        qkv_store(Ln, y, ptok, np, lane, vm);
        return;
    }
    // pooling heads, token-wise part: u = relu(linear1 x), score = linear2 u   (NISQA_lib.py:1173-1174; pool_att_dropout = 0)
    for (int hd = 0; hd < hs.n; ++hd) {
        const tdt_head_p& H = hs.h[hd];
        f32x16 u[4], w2[4];
        ld_vec<4>(H.b1, u, hf);
        chain_gemm<2, 4>((const f32x4*)H.f_p1, y, u, lane);
        ld_vec<4>(H.w2, w2, hf);
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                u[mt][r] = fmaxf(u[mt][r], 0.f);
                s = fmaf(w2[mt][r], u[mt][r], s);
            }
        s += __shfl_xor(s, 32);
        st_vec<4>(H.u + (size_t)ptok * 128, u, hf, vm);
        if (hf == 0) H.sc[ptok] = (s + H.b2[0]) * vm;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Per (clip, head): masked softmax over the clip's scores, pooled = att x, y_hat = linear3(pooled) (NISQA_lib.py:1176-1183),
// this clip's term of the loss and d loss / d y_hat (biasLoss._nan_mse with the optional cubic mapping, NISQA_lib.py:
// 1880-1892, 1946-1950; inv_cnt[h] = 1 / number of labelled clips of the WHOLE batch, counted by the host), and on into
// the backward pass: d pooled, linear3's gradients, d att -> d score.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tdt_pool_clip_kernel(tdt_common c, tdt_heads hs, const float* __restrict__ x,
                                                            const float* __restrict__ labels, const float* __restrict__ bias_map,
                                                            const float* __restrict__ inv_cnt, float* __restrict__ y_hat,
                                                            float* __restrict__ loss) {
    __shared__ float red[256];
    __shared__ float vec[64];
    __shared__ float bc[2];
    const int tid = threadIdx.x, b = blockIdx.x, hd = blockIdx.y, nh = hs.n;
    const tdt_head_p& H = hs.h[hd];
    const int n = c.seg_off[b + 1] - c.seg_off[b], c0 = c.ptok_off[b];
    const int npad = (n + 31) & ~31;
    auto block_reduce = [&](float v, bool is_max) {
        red[tid] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] = is_max ? fmaxf(red[tid], red[tid + s]) : red[tid] + red[tid + s];
            __syncthreads();
        }
        const float r = red[0];
        __syncthreads();
        return r;
    };
    float mx = -INFINITY;
    for (int t = tid; t < n; t += 256) mx = fmaxf(mx, H.sc[c0 + t]);
    mx = block_reduce(mx, true);
    float den = 0.f;
    for (int t = tid; t < n; t += 256) den += expf(H.sc[c0 + t] - mx);
    den = block_reduce(den, false);
    const float inv_den = 1.0f / den;
    for (int t = tid; t < npad; t += 256) H.att[c0 + t] = t < n ? expf(H.sc[c0 + t] - mx) * inv_den : 0.f;
    __syncthreads();
    // pooled[f] = sum_t att_t x_t[f]: thread (w, f) takes tokens w, w + 4, ...
    const int f = tid & 63, w = tid >> 6;
    float acc = 0.f;
    for (int t = w; t < n; t += 4) acc = fmaf(H.att[c0 + t], x[(size_t)(c0 + t) * 64 + f], acc);
    red[tid] = acc;
    __syncthreads();
    if (tid < 64) vec[tid] = red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192];
    __syncthreads();
    if (tid < 64) {
        const float pooled = vec[tid];
        const float yv = wave_sum(pooled * H.w3[tid]) + H.b3[0];
        float dy = 0.f;
        if (tid == 0) {
            y_hat[b * nh + hd] = yv;
            const float tgt = labels[b * nh + hd];
            if (!isnan(tgt)) {
                float mapped = yv, slope = 1.f;
                if (bias_map) {
                    const float* q = bias_map + b * 4;
                    mapped = q[0] + yv * (q[1] + yv * (q[2] + yv * q[3]));
                    slope = q[1] + yv * (2.f * q[2] + 3.f * yv * q[3]);
                }
                const float e = mapped - tgt, ic = inv_cnt[hd];
                atomicAdd(loss, e * e * ic);
                atomicAdd(loss + 1 + hd, e * e * ic);
                dy = 2.f * e * ic * slope;
            }
            bc[0] = dy;
            atomicAdd(H.g_b3, dy);
        }
        dy = __shfl(dy, 0);
        atomicAdd(H.g_w3 + tid, dy * pooled);                      // linear3.weight gradient
        const float dp = dy * H.w3[tid];                           // d loss / d pooled
        vec[tid] = dp;
        H.dpooled[(size_t)b * 64 + tid] = dp;
    }
    __syncthreads();
    // d att_t = d pooled . x_t, then the softmax backward: d score_t = att_t (d att_t - sum att d att)
    float part = 0.f;
    for (int t = tid; t < n; t += 256) {
        const f32x4* xr = (const f32x4*)(x + (size_t)(c0 + t) * 64);
        float d = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x4 xv = xr[q];
            d = fmaf(xv[0], vec[4 * q], d); d = fmaf(xv[1], vec[4 * q + 1], d);
            d = fmaf(xv[2], vec[4 * q + 2], d); d = fmaf(xv[3], vec[4 * q + 3], d);
        }
        H.dsc[c0 + t] = d;
        part = fmaf(H.att[c0 + t], d, part);
    }
    const float dot = block_reduce(part, false);
    float sb = 0.f;
    for (int t = tid; t < npad; t += 256) {
        const float ds = t < n ? H.att[c0 + t] * (H.dsc[c0 + t] - dot) : 0.f;
        H.dsc[c0 + t] = ds;
        sb += ds;
    }
    sb = block_reduce(sb, false);
    if (tid == 0) atomicAdd(H.g_b2, sb);                           // linear2.bias gradient (zero up to rounding under the softmax)
}

// ---------------------------------------------------------------------------------------------------------
// Token-wise backward of one layer from d loss / d (layer output) down to d ctx (the attention's output gradient)
// ---------------------------------------------------------------------------------------------------------
NQ_DEV void bwd_part_a(const tdt_layer_p& L, f32x16 (&dx)[2], int ptok, int utok, bool valid, float vm, int np, int lane) {
    const int hf = lane >> 5;
    f32x16 t[2], xh[2];
    st_vec<2>(L.dx + (size_t)ptok * 64, dx, hf, vm);               // LayerNorm2 parameter gradients: column sums of dx (* xhat2)
    ld_vec<2>(L.xh2 + (size_t)ptok * 64, xh, hf);
    ln_bwd(dx, xh, L.rs2[ptok], L.g2, hf);                         // dx := d r2
    f32x16 df[2];
    df[0] = dx[0]; df[1] = dx[1];
    if (L.m2 && valid) {
        ld_vec<2>(L.m2 + (size_t)utok * 64, t, hf);
        mul_t<2>(df, t);
    }
    st_vec<2>(L.df + (size_t)ptok * 64, df, hf, vm);
    f32x16 dh[2];
    zero_t<2>(dh);
    chain_gemm<2, 2>((const f32x4*)L.t_ff2, df, dh, lane);
    ld_vec<2>(L.hd + (size_t)ptok * 64, t, hf);                    // hd = relu(.) * mask: gate and mask in one comparison
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[mt][r] = t[mt][r] > 0.f ? dh[mt][r] : 0.f;
    if (L.mf && valid) {
        ld_vec<2>(L.mf + (size_t)utok * 64, t, hf);
        mul_t<2>(dh, t);
    }
    st_vec<2>(L.dh + (size_t)ptok * 64, dh, hf, vm);
    chain_gemm<2, 2>((const f32x4*)L.t_ff1, dh, dx, lane);         // dx := d x1 = W1^T dh + d r2
    st_vec<2>(L.dx1 + (size_t)ptok * 64, dx, hf, vm);
    ld_vec<2>(L.xh1 + (size_t)ptok * 64, xh, hf);
    ln_bwd(dx, xh, L.rs1[ptok], L.g1, hf);                         // dx := d r1
    st_vec<2>(L.dr1 + (size_t)ptok * 64, dx, hf, vm);
    if (L.m1 && valid) {
        ld_vec<2>(L.m1 + (size_t)utok * 64, t, hf);
        mul_t<2>(dx, t);
    }
    st_vec<2>(L.datt + (size_t)ptok * 64, dx, hf, vm);
    f32x16 dc[2];
    zero_t<2>(dc);
    chain_gemm<2, 2>((const f32x4*)L.t_out, dx, dc, lane);
    st_vec<2>(L.dctx + (size_t)ptok * 64, dc, hf, vm);
    st_fm(L.dctxT, np, ptok, dc, hf, vm);
    ld_vec<2>(L.ctx + (size_t)ptok * 64, t, hf);
    float dd = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dd = fmaf(dc[mt][r], t[mt][r], dd);
    dd += __shfl_xor(dd, 32);
    if (hf == 0) L.dd[ptok] = dd * vm;                             // D_i = sum_j P_ij dP_ij = d ctx_i . ctx_i
}

__global__ __launch_bounds__(64) void tdt_bwd_tail_kernel(tdt_common c, tdt_layer_p L, tdt_heads hs, const float* __restrict__ x) {
    const int lane = threadIdx.x, j = lane & 31, hf = lane >> 5;
    const int b = c.tile_clip[blockIdx.x];
    const int n = c.seg_off[b + 1] - c.seg_off[b];
    const int ptok = blockIdx.x * 32 + j, kq = ptok - c.ptok_off[b];
    const bool valid = kq < n;
    const float vm = valid ? 1.f : 0.f;
    const int utok = c.seg_off[b] + (valid ? kq : 0);
    f32x16 dx[2];
    zero_t<2>(dx);
    for (int hd = 0; hd < hs.n; ++hd) {
        const tdt_head_p& H = hs.h[hd];
        f32x16 u[4], w2[4], dp[2];
        ld_vec<4>(H.u + (size_t)ptok * 128, u, hf);
        ld_vec<4>(H.w2, w2, hf);
        const float dsc = H.dsc[ptok], att = H.att[ptok];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) u[mt][r] = u[mt][r] > 0.f ? dsc * w2[mt][r] : 0.f;    // d u through the ReLU
        st_vec<4>(H.du + (size_t)ptok * 128, u, hf, vm);
        chain_gemm<4, 2>((const f32x4*)H.t_p1, u, dx, lane);       // += linear1^T d u
        ld_vec<2>(H.dpooled + (size_t)b * 64, dp, hf);
#pragma unroll
        for (int r = 0; r < 16; ++r) { dx[0][r] = fmaf(att, dp[0][r], dx[0][r]); dx[1][r] = fmaf(att, dp[1][r], dx[1][r]); }
    }
    (void)x;
    bwd_part_a(L, dx, ptok, utok, valid, vm, c.np, lane);
}

// ---------------------------------------------------------------------------------------------------------
// Attention backward (the gradient of softmax(q k^T / 8) with dropout, times v), P recomputed from q, k and the saved
// log-sum-exp.  blockIdx.y = 0: a 32-QUERY tile in the lanes, key tiles in a loop -> d q;
// blockIdx.y = 1: a 32-KEY tile in the lanes, query tiles in a loop -> d k, d v.
// blockIdx.z = one of TDT_ASPLIT interleaved shares of the loop's tiles: a wave is a serial chain of ~128 MFMAs per tile, and
// with one wave per (tile, pass) only 512 waves ran on 1 024 SIMDs for 8 dependent tiles each.  Share 0 writes its partial
// d q / d k / d v into d qkv, share s > 0 into plane s - 1 of dqkvp; tdt_bwd_mid sums the planes and stores the total where the
// parameter-gradient launches read it.  (A first version added the shares with fp32 atomics: a lane's row is 768 bytes from its
// neighbour's, 64 separate L2 transactions per instruction -- 261 us instead of 56.)
// ---------------------------------------------------------------------------------------------------------
#define TDT_ASPLIT 2
__global__ __launch_bounds__(64) void tdt_attn_bwd_kernel(tdt_common c, tdt_layer_p L) {
    const int lane = threadIdx.x, j = lane & 31, hf = lane >> 5;
    const int share = blockIdx.z, nshare = gridDim.z;
    float* __restrict__ part = share == 0 ? L.dqkv : L.dqkvp + (size_t)(share - 1) * c.np * 192;
    const int b = c.tile_clip[blockIdx.x];
    const int n = c.seg_off[b + 1] - c.seg_off[b], c0 = c.ptok_off[b];
    const int ptok = blockIdx.x * 32 + j, kq = ptok - c0;           // this lane's token: a query (y = 0) or a key (y = 1)
    const bool valid = kq < n;
    const int np = c.np, ntile = (n + 31) >> 5;
    const float* mbase = L.mP ? L.mP + c.sq_off[b] : nullptr;
    if (blockIdx.y == 0) {
        f32x4 qf[8], cf[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            qf[s] = *(const f32x4*)(L.qs + (size_t)ptok * 64 + 8 * s + 4 * hf);
            cf[s] = *(const f32x4*)(L.dctx + (size_t)ptok * 64 + 8 * s + 4 * hf);
        }
        const float lse = L.lse[ptok], dd = L.dd[ptok];
        const bool use_mask = mbase != nullptr && valid;
        const float* mrow = use_mask ? mbase + (int64_t)kq * n : nullptr;
        f32x16 dq[2];
        zero_t<2>(dq);
        // K / V rows of tile kt + 1 are requested while tile kt is multiplied; K^T and the dropout multipliers of tile kt at
        // its start, used behind its 64 MFMAs
        f32x4 kA[8], vA[8], kB[8], vB[8];
        if (share >= ntile) {                                    // fewer key tiles than shares: this share's plane holds zeros
            st_vec<2>(part + (size_t)ptok * 192, dq, hf, 0.f);
            return;
        }
        {
            const float* krow = L.k + (size_t)(c0 + 32 * share + j) * 64 + 4 * hf;
            const float* vrow = L.v + (size_t)(c0 + 32 * share + j) * 64 + 4 * hf;
#pragma unroll
            for (int s = 0; s < 8; ++s) { kA[s] = *(const f32x4*)(krow + 8 * s); vA[s] = *(const f32x4*)(vrow + 8 * s); }
        }
        auto tile = [&](int kt, const f32x4 (&kc)[8], const f32x4 (&vc)[8], f32x4 (&kn)[8], f32x4 (&vn)[8]) {
            const int key0 = c0 + 32 * kt;
            f32x4 kf[2][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                kf[0][g] = *(const f32x4*)(L.kT + (size_t)j * np + key0 + 8 * g + 4 * hf);
                kf[1][g] = *(const f32x4*)(L.kT + (size_t)(j + 32) * np + key0 + 8 * g + 4 * hf);
            }
            f32x16 mk;
            if (use_mask) mk = ld_mask_row(mrow, 32 * kt, n, hf);
            if (kt + nshare < ntile) {
                const float* krow = L.k + (size_t)(key0 + 32 * nshare + j) * 64 + 4 * hf;
                const float* vrow = L.v + (size_t)(key0 + 32 * nshare + j) * 64 + 4 * hf;
#pragma unroll
                for (int s = 0; s < 8; ++s) { kn[s] = *(const f32x4*)(krow + 8 * s); vn[s] = *(const f32x4*)(vrow + 8 * s); }
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 sT = zero16(), dpT = zero16();                  // rows = keys, columns = queries
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    sT = mfma32(kc[s][kk], qf[s][kk], sT);
                    dpT = mfma32(vc[s][kk], cf[s][kk], dpT);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * kt + NQ_DROW(r, hf);
                const float p = key < n ? expf(sT[r] - lse) : 0.f;
                float dp = dpT[r];
                if (use_mask) dp *= mk[r];
                sT[r] = p * (dp - dd);                               // d S^T
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    dq[0] = mfma32(kf[0][g][kk], sT[4 * g + kk], dq[0]);
                    dq[1] = mfma32(kf[1][g][kk], sT[4 * g + kk], dq[1]);
                }
        };
        for (int kt = share; kt < ntile; kt += 2 * nshare) {
            tile(kt, kA, vA, kB, vB);
            if (kt + nshare < ntile) tile(kt + nshare, kB, vB, kA, vA);
        }
        st_vec<2>(part + (size_t)ptok * 192, dq, hf, valid ? 0.125f : 0.f);      // q entered the scores as q / 8
    } else {
        f32x4 kf[8], vf[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            kf[s] = *(const f32x4*)(L.k + (size_t)ptok * 64 + 8 * s + 4 * hf);
            vf[s] = *(const f32x4*)(L.v + (size_t)ptok * 64 + 8 * s + 4 * hf);
        }
        const bool use_mask = mbase != nullptr && valid;
        f32x16 dk[2], dv[2];
        zero_t<2>(dk);
        zero_t<2>(dv);
        f32x4 qA[8], cA[8], qB[8], cB[8];
        if (share >= ntile) {
            st_vec<2>(part + (size_t)ptok * 192 + 64, dk, hf, 0.f);
            st_vec<2>(part + (size_t)ptok * 192 + 128, dv, hf, 0.f);
            return;
        }
        {
            const float* qrow = L.qs + (size_t)(c0 + 32 * share + j) * 64 + 4 * hf;
            const float* crow = L.dctx + (size_t)(c0 + 32 * share + j) * 64 + 4 * hf;
#pragma unroll
            for (int s = 0; s < 8; ++s) { qA[s] = *(const f32x4*)(qrow + 8 * s); cA[s] = *(const f32x4*)(crow + 8 * s); }
        }
        auto tile = [&](int qt, const f32x4 (&qc)[8], const f32x4 (&cc)[8], f32x4 (&qn)[8], f32x4 (&cn)[8]) {
            const int q0 = c0 + 32 * qt;
            f32x4 qT[2][4], cT[2][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                qT[0][g] = *(const f32x4*)(L.qsT + (size_t)j * np + q0 + 8 * g + 4 * hf);
                qT[1][g] = *(const f32x4*)(L.qsT + (size_t)(j + 32) * np + q0 + 8 * g + 4 * hf);
                cT[0][g] = *(const f32x4*)(L.dctxT + (size_t)j * np + q0 + 8 * g + 4 * hf);
                cT[1][g] = *(const f32x4*)(L.dctxT + (size_t)(j + 32) * np + q0 + 8 * g + 4 * hf);
            }
            const f32x16 lse = ld_row16(L.lse + q0, hf), dd = ld_row16(L.dd + q0, hf);
            f32x16 mk;
            if (use_mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qi = 32 * qt + NQ_DROW(r, hf);
                    mk[r] = qi < n ? mbase[(int64_t)qi * n + kq] : 1.f;
                }
            }
            if (qt + nshare < ntile) {
                const float* qrow = L.qs + (size_t)(q0 + 32 * nshare + j) * 64 + 4 * hf;
                const float* crow = L.dctx + (size_t)(q0 + 32 * nshare + j) * 64 + 4 * hf;
#pragma unroll
                for (int s = 0; s < 8; ++s) { qn[s] = *(const f32x4*)(qrow + 8 * s); cn[s] = *(const f32x4*)(crow + 8 * s); }
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 s_ = zero16(), dp = zero16();                   // rows = queries, columns = keys
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    s_ = mfma32(qc[s][kk], kf[s][kk], s_);
                    dp = mfma32(cc[s][kk], vf[s][kk], dp);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = 32 * qt + NQ_DROW(r, hf);
                const float p = (valid && qi < n) ? expf(s_[r] - lse[r]) : 0.f;
                const float mkr = use_mask ? mk[r] : 1.f;
                s_[r] = p * (dp[r] * mkr - dd[r]);                   // d S
                dp[r] = p * mkr;                                     // P after dropout
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    dk[0] = mfma32(qT[0][g][kk], s_[4 * g + kk], dk[0]);
                    dk[1] = mfma32(qT[1][g][kk], s_[4 * g + kk], dk[1]);
                    dv[0] = mfma32(cT[0][g][kk], dp[4 * g + kk], dv[0]);
                    dv[1] = mfma32(cT[1][g][kk], dp[4 * g + kk], dv[1]);
                }
        };
        for (int qt = share; qt < ntile; qt += 2 * nshare) {
            tile(qt, qA, cA, qB, cB);
            if (qt + nshare < ntile) tile(qt + nshare, qB, cB, qA, cA);
        }
        st_vec<2>(part + (size_t)ptok * 192 + 64, dk, hf, 1.f);
        st_vec<2>(part + (size_t)ptok * 192 + 128, dv, hf, 1.f);
    }
}

// ---------------------------------------------------------------------------------------------------------
// d qkv of layer l + 1 -> d (its input) = in_proj^T d qkv + d r1, then the token-wise backward of layer l; below layer 0:
// LayerNorm backward and the 64 -> 384 input gradient that goes back into the CNN (unpadded token order)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tdt_bwd_mid_kernel(tdt_common c, tdt_layer_p Lup, tdt_layer_p Llow, int has_lower,
                                                         const float* __restrict__ xh0, const float* __restrict__ rs0,
                                                         const float* __restrict__ g0, const float* __restrict__ t_w0,
                                                         float* __restrict__ dxl0, float* __restrict__ dx0,
                                                         float* __restrict__ dx0u, float* __restrict__ dfeat) {
    const int lane = threadIdx.x, j = lane & 31, hf = lane >> 5;
    const int b = c.tile_clip[blockIdx.x];
    const int n = c.seg_off[b + 1] - c.seg_off[b];
    const int ptok = blockIdx.x * 32 + j, kq = ptok - c.ptok_off[b];
    const bool valid = kq < n;
    const float vm = valid ? 1.f : 0.f;
    const int utok = c.seg_off[b] + (valid ? kq : 0);
    f32x16 dqkv[6], dx[2];
    ld_vec<6>(Lup.dqkv + (size_t)ptok * 192, dqkv, hf);
    {                                                            // + the other shares of the attention backward; the total goes back
        f32x16 pp[6];                                            // to d qkv for the weight-gradient GEMM and the bias column sums
#pragma unroll 1
        for (int sh = 1; sh < TDT_ASPLIT; ++sh) {
            ld_vec<6>(Lup.dqkvp + ((size_t)(sh - 1) * c.np + ptok) * 192, pp, hf);
#pragma unroll
            for (int mt = 0; mt < 6; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dqkv[mt][r] += pp[mt][r];
        }
        st_vec<6>(Lup.dqkv + (size_t)ptok * 192, dqkv, hf, 1.f);
    }
    ld_vec<2>(Lup.dr1 + (size_t)ptok * 64, dx, hf);
    chain_gemm<6, 2>((const f32x4*)Lup.t_qkv, dqkv, dx, lane);
    if (has_lower) {
        bwd_part_a(Llow, dx, ptok, utok, valid, vm, c.np, lane);
        return;
    }
    st_vec<2>(dxl0 + (size_t)ptok * 64, dx, hf, vm);
    f32x16 xh[2];
    ld_vec<2>(xh0 + (size_t)ptok * 64, xh, hf);
    ln_bwd(dx, xh, rs0[ptok], g0, hf);
    st_vec<2>(dx0 + (size_t)ptok * 64, dx, hf, vm);
    if (valid) st_vec<2>(dx0u + (size_t)utok * 64, dx, hf, 1.f);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        f32x16 acc[6];
        zero_t<6>(acc);
        chain_gemm<2, 6>((const f32x4*)(t_w0 + (size_t)half * (8 * 6 * 256)), dx, acc, lane);
        if (valid) st_vec<6>(dfeat + (size_t)utok * 384 + 192 * half, acc, hf, 1.f);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Weight fragments of the step from the flat parameter buffer: A operand value(s, mt, lane, kk) = M[row0 + 32 mt + (lane & 31)]
// [8 s + 4 (lane >> 5) + kk], M = W (row length ld) or W^T
// ---------------------------------------------------------------------------------------------------------
struct tdt_pack_job { int32_t src, ld, transposed, row0, mt, steps, dst, pad; };
struct tdt_pack_jobs { tdt_pack_job j[56]; int n; float* zero; int n_zero; };

__global__ __launch_bounds__(256) void tdt_pack_kernel(tdt_pack_jobs jobs, const float* __restrict__ params, float* __restrict__ frags) {
    if ((int)blockIdx.x == jobs.n) {                               // one extra block clears the loss accumulators
        for (int i = threadIdx.x; i < jobs.n_zero; i += 256) jobs.zero[i] = 0.f;
        return;
    }
    const tdt_pack_job J = jobs.j[blockIdx.x];
    const int total = J.steps * J.mt * 256;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < total; i += gridDim.y * 256) {
        const int kk = i & 3, lane = (i >> 2) & 63, sm = i >> 8;
        const int mt = sm % J.mt, s = sm / J.mt;
        const int row = J.row0 + 32 * mt + (lane & 31), k = 8 * s + 4 * (lane >> 5) + kk;
        frags[J.dst + i] = J.transposed ? params[J.src + (size_t)k * J.ld + row] : params[J.src + (size_t)row * J.ld + k];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Column sums: job = (a_off, b_off or -1, cols, rows, dst_sum or -1, dst_dot or -1): grads[dst_sum + c] += sum_r a[r][c],
// grads[dst_dot + c] += sum_r a[r][c] b[r][c]   (bias and LayerNorm-parameter gradients; rows of padding tokens are zero)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tdt_colsum_kernel(const int64_t* __restrict__ jobs, const float* __restrict__ ws,
                                                         float* __restrict__ grads) {
    __shared__ double red[2][256][4];
    const int64_t* J = jobs + (int64_t)blockIdx.x * 6;
    const float* a = ws + J[0];
    const float* bq = J[1] >= 0 ? ws + J[1] : nullptr;
    const int cols = (int)J[2], rows = (int)J[3];
    const int c4 = cols >> 2, ngrp = 256 / c4, tid = threadIdx.x;
    const int ci = tid % c4, grp = tid / c4;
    const int per = (rows + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
    if (grp < ngrp)
        for (int r = r0 + grp; r < r1; r += ngrp) {
            const f32x4 av = *(const f32x4*)(a + (size_t)r * cols + 4 * ci);
            s += av;
            if (bq) d += av * *(const f32x4*)(bq + (size_t)r * cols + 4 * ci);
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][tid][e] = s[e]; red[1][tid][e] = d[e]; }
    __syncthreads();
    if (tid < c4) {
        double ts[4] = {0, 0, 0, 0}, td[4] = {0, 0, 0, 0};
        for (int g = 0; g < ngrp; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) { ts[e] += red[0][g * c4 + tid][e]; td[e] += red[1][g * c4 + tid][e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (J[4] >= 0) atomicAdd(grads + J[4] + 4 * tid + e, (float)ts[e]);
            if (J[5] >= 0) atomicAdd(grads + J[5] + 4 * tid + e, (float)td[e]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side: workspace layout, plan (descriptor tables for the caller to upload), step
// ---------------------------------------------------------------------------------------------------------
struct tdt_layout {
    int64_t feat, dfeat, yhat, loss, x[TDT_MAX_LAYERS + 1], xh0, rs0, dxl0, dx0, dx0u;
    struct { int64_t qs, qsT, k, kT, v, vT, lse, ctx, x1, xh1, rs1, hd, xh2, rs2, dx, df, dh, dx1, datt, dctx, dctxT, dd, dr1, dqkv, dqkvp; } L[TDT_MAX_LAYERS];
    struct { int64_t u, sc, att, dsc, du, dpooled; } H[TDT_MAX_HEADS];
    int64_t total;
    // fragments
    int64_t f_w0, t_w0;
    struct { int64_t f_qkv, f_out, f_ff1, f_ff2, t_qkv, t_out, t_ff1, t_ff2; } FL[TDT_MAX_LAYERS];
    struct { int64_t f_p1, t_p1; } FH[TDT_MAX_HEADS];
    int64_t frag_total;
};

static bool tdt_dims_ok(int B, int S, int NP, int nl, int nh) {
    return B > 0 && S > 0 && NP >= S && (NP & 31) == 0 && nl >= 1 && nl <= TDT_MAX_LAYERS && nh >= 1 && nh <= TDT_MAX_HEADS;
}

static tdt_layout tdt_make_layout(int B, int S, int NP, int nl, int nh) {
    tdt_layout y;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += (n + 3) / 4 * 4; return at; };
    const int64_t t64 = (int64_t)NP * 64;
    y.feat = take((int64_t)S * 384);
    y.dfeat = take((int64_t)S * 384);
    y.yhat = take((int64_t)B * nh);
    y.loss = take(1 + TDT_MAX_HEADS);
    for (int l = 0; l <= nl; ++l) y.x[l] = take(t64);
    y.xh0 = take(t64); y.rs0 = take(NP); y.dxl0 = take(t64); y.dx0 = take(t64); y.dx0u = take((int64_t)S * 64);
    for (int l = 0; l < nl; ++l) {
        auto& L = y.L[l];
        L.qs = take(t64); L.qsT = take(t64); L.k = take(t64); L.kT = take(t64); L.v = take(t64); L.vT = take(t64);
        L.lse = take(NP); L.ctx = take(t64); L.x1 = take(t64); L.xh1 = take(t64); L.rs1 = take(NP); L.hd = take(t64);
        L.xh2 = take(t64); L.rs2 = take(NP); L.dx = take(t64); L.df = take(t64); L.dh = take(t64); L.dx1 = take(t64);
        L.datt = take(t64); L.dctx = take(t64); L.dctxT = take(t64); L.dd = take(NP); L.dr1 = take(t64); L.dqkv = take(3 * t64);
        L.dqkvp = take((TDT_ASPLIT - 1) * 3 * t64);
    }
    for (int h = 0; h < nh; ++h) {
        auto& H = y.H[h];
        H.u = take(2 * t64); H.sc = take(NP); H.att = take(NP); H.dsc = take(NP); H.du = take(2 * t64); H.dpooled = take((int64_t)B * 64);
    }
    y.total = o;
    int64_t f = 0;
    auto ftake = [&](int64_t n) { const int64_t at = f; f += n; return at; };
    y.f_w0 = ftake(48 * 2 * 256);
    y.t_w0 = ftake(2 * 8 * 6 * 256);
    for (int l = 0; l < nl; ++l) {
        auto& F = y.FL[l];
        F.f_qkv = ftake(8 * 6 * 256); F.f_out = ftake(8 * 2 * 256); F.f_ff1 = ftake(8 * 2 * 256); F.f_ff2 = ftake(8 * 2 * 256);
        F.t_qkv = ftake(24 * 2 * 256); F.t_out = ftake(8 * 2 * 256); F.t_ff1 = ftake(8 * 2 * 256); F.t_ff2 = ftake(8 * 2 * 256);
    }
    for (int h = 0; h < nh; ++h) { y.FH[h].f_p1 = ftake(8 * 4 * 256); y.FH[h].t_p1 = ftake(16 * 2 * 256); }
    y.frag_total = f;
    return y;
}

// poff: offsets (floats) of the block's parameters in the flat parameter / gradient buffers, in the order documented in
// nisqa_train.h: proj.W, proj.b, ln0.g, ln0.b; per layer qkv.W, qkv.b, out.W, out.b, ln1.g, ln1.b, ff1.W, ff1.b, ff2.W,
// ff2.b, ln2.g, ln2.b; per head p1.W, p1.b, p2.w, p2.b, p3.w, p3.b
#define PO_LAYER(l) (4 + 12 * (l))
#define PO_HEAD(nl, h) (4 + 12 * (nl) + 6 * (h))

extern "C" int nisqa_tdtrain_plan(int32_t n_clips, int32_t n_tokens, int32_t n_tokens_padded, int32_t n_layers, int32_t n_heads,
                                  const int32_t* poff, int64_t* out, int64_t cap) {
    if (!tdt_dims_ok(n_clips, n_tokens, n_tokens_padded, n_layers, n_heads) || !poff || !out) return NISQA_ERR_ARG;
    const int B = n_clips, S = n_tokens, NP = n_tokens_padded, nl = n_layers, nh = n_heads;
    const tdt_layout y = tdt_make_layout(B, S, NP, nl, nh);
    const int n_groups = 1 + 4 * nl + 2 * nh, n_jobs = 2 + 6 * nl + nh;
    const int64_t need = 8 + (int64_t)n_groups * 10 + (int64_t)n_jobs * 6;
    if (cap < need) return NISQA_ERR_ARG;
    int64_t* d = out + 8;
    int64_t tiles = 0;
    auto group = [&](int64_t a, int64_t b, int64_t c_, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc) {
        const int64_t v[10] = {a, b, c_, M, N, K, lda, ldb, ldc, tiles};
        for (int i = 0; i < 10; ++i) d[i] = v[i];
        d += 10;
        tiles += ((M + 63) / 64) * ((N + 63) / 64);
    };
    // every weight gradient of the block is d Y^T X: A = d Y stored [rows][M], B = X stored [rows][N]
    group(y.dx0u, y.feat, poff[0], 64, 384, S, 64, 384, 384);
    for (int l = 0; l < nl; ++l) {
        const int32_t* p = poff + PO_LAYER(l);
        group(y.L[l].dqkv, y.x[l], p[0], 192, 64, NP, 192, 64, 64);
        group(y.L[l].datt, y.L[l].ctx, p[2], 64, 64, NP, 64, 64, 64);
        group(y.L[l].dh, y.L[l].x1, p[6], 64, 64, NP, 64, 64, 64);
        group(y.L[l].df, y.L[l].hd, p[8], 64, 64, NP, 64, 64, 64);
    }
    for (int h = 0; h < nh; ++h) {
        const int32_t* p = poff + PO_HEAD(nl, h);
        group(y.H[h].du, y.x[nl], p[0], 128, 64, NP, 128, 64, 64);
        group(y.H[h].dsc, y.H[h].u, p[2], 1, 128, NP, 1, 128, 128);
    }
    int64_t* q = d;
    auto job = [&](int64_t a, int64_t b, int64_t cols, int64_t rows, int64_t dsum, int64_t ddot) {
        const int64_t v[6] = {a, b, cols, rows, dsum, ddot};
        for (int i = 0; i < 6; ++i) q[i] = v[i];
        q += 6;
    };
    job(y.dx0, -1, 64, NP, poff[1], -1);                           // proj bias
    job(y.dxl0, y.xh0, 64, NP, poff[3], poff[2]);                  // LayerNorm0 beta, gamma
    for (int l = 0; l < nl; ++l) {
        const int32_t* p = poff + PO_LAYER(l);
        job(y.L[l].dqkv, -1, 192, NP, p[1], -1);
        job(y.L[l].datt, -1, 64, NP, p[3], -1);
        job(y.L[l].dx1, y.L[l].xh1, 64, NP, p[5], p[4]);
        job(y.L[l].dh, -1, 64, NP, p[7], -1);
        job(y.L[l].df, -1, 64, NP, p[9], -1);
        job(y.L[l].dx, y.L[l].xh2, 64, NP, p[11], p[10]);
    }
    for (int h = 0; h < nh; ++h) job(y.H[h].du, -1, 128, NP, poff[PO_HEAD(nl, h) + 1], -1);
    out[0] = y.total; out[1] = y.frag_total; out[2] = n_groups; out[3] = tiles; out[4] = n_jobs;
    out[5] = y.feat; out[6] = y.dfeat; out[7] = y.yhat;            // (the loss follows y_hat: out[7] + round_up(B * heads, 4))
    return NISQA_OK;
}

extern "C" int nisqa_tdtrain_step(const nisqa_tdtrain_args* a, void* stream) {
    if (!a || !tdt_dims_ok(a->n_clips, a->n_tokens, a->n_tokens_padded, a->n_layers, a->n_heads) || !a->seg_off || !a->ptok_off ||
        !a->tile_clip || !a->sq_off || !a->params || !a->grads || !a->poff || !a->ws || !a->frags || !a->labels || !a->inv_count ||
        !a->wgrad_desc || !a->colsum_jobs)
        return NISQA_ERR_ARG;
    const int B = a->n_clips, S = a->n_tokens, NP = a->n_tokens_padded, nl = a->n_layers, nh = a->n_heads;
    const tdt_layout y = tdt_make_layout(B, S, NP, nl, nh);
    const int32_t* po = a->poff;
    const float* P = a->params;
    float* G = a->grads;
    float* ws = a->ws;
    float* fr = a->frags;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = NP / 32;
    tdt_common c = {a->seg_off, a->ptok_off, a->tile_clip, a->sq_off, NP};

    tdt_layer_p Lp[TDT_MAX_LAYERS];
    for (int l = 0; l < nl; ++l) {
        const int32_t* p = po + PO_LAYER(l);
        const auto& F = y.FL[l];
        const auto& W = y.L[l];
        tdt_layer_p& L = Lp[l];
        L.f_qkv = fr + F.f_qkv; L.f_out = fr + F.f_out; L.f_ff1 = fr + F.f_ff1; L.f_ff2 = fr + F.f_ff2;
        L.t_qkv = fr + F.t_qkv; L.t_out = fr + F.t_out; L.t_ff1 = fr + F.t_ff1; L.t_ff2 = fr + F.t_ff2;
        L.b_qkv = P + p[1]; L.b_out = P + p[3]; L.g1 = P + p[4]; L.be1 = P + p[5]; L.b_ff1 = P + p[7]; L.b_ff2 = P + p[9];
        L.g2 = P + p[10]; L.be2 = P + p[11];
        L.qs = ws + W.qs; L.qsT = ws + W.qsT; L.k = ws + W.k; L.kT = ws + W.kT; L.v = ws + W.v; L.vT = ws + W.vT;
        L.lse = ws + W.lse; L.ctx = ws + W.ctx; L.x1 = ws + W.x1; L.xh1 = ws + W.xh1; L.rs1 = ws + W.rs1; L.hd = ws + W.hd;
        L.xh2 = ws + W.xh2; L.rs2 = ws + W.rs2; L.dx = ws + W.dx; L.df = ws + W.df; L.dh = ws + W.dh; L.dx1 = ws + W.dx1;
        L.datt = ws + W.datt; L.dctx = ws + W.dctx; L.dctxT = ws + W.dctxT; L.dd = ws + W.dd; L.dr1 = ws + W.dr1;
        L.dqkv = ws + W.dqkv;
        L.dqkvp = ws + W.dqkvp;
        L.mP = a->mask_p[l]; L.m1 = a->mask_1[l]; L.mf = a->mask_f[l]; L.m2 = a->mask_2[l];
    }
    tdt_heads hs;
    hs.n = nh;
    for (int h = 0; h < nh; ++h) {
        const int32_t* p = po + PO_HEAD(nl, h);
        tdt_head_p& H = hs.h[h];
        H.f_p1 = fr + y.FH[h].f_p1; H.t_p1 = fr + y.FH[h].t_p1;
        H.b1 = P + p[1]; H.w2 = P + p[2]; H.b2 = P + p[3]; H.w3 = P + p[4]; H.b3 = P + p[5];
        H.u = ws + y.H[h].u; H.sc = ws + y.H[h].sc; H.att = ws + y.H[h].att; H.dsc = ws + y.H[h].dsc; H.du = ws + y.H[h].du;
        H.dpooled = ws + y.H[h].dpooled;
        H.g_w3 = G + p[4]; H.g_b3 = G + p[5]; H.g_b2 = G + p[3];
    }
    for (int h = nh; h < TDT_MAX_HEADS; ++h) hs.h[h] = hs.h[0];

    // fragments of this step's weights
    tdt_pack_jobs jobs;
    int nj = 0;
    auto pj = [&](int32_t src, int32_t ld, int32_t tr, int32_t row0, int32_t mt, int32_t steps, int64_t dst) {
        tdt_pack_job J = {src, ld, tr, row0, mt, steps, (int32_t)dst, 0};
        jobs.j[nj++] = J;
    };
    pj(po[0], 384, 0, 0, 2, 48, y.f_w0);
    pj(po[0], 384, 1, 0, 6, 8, y.t_w0);
    pj(po[0], 384, 1, 192, 6, 8, y.t_w0 + 8 * 6 * 256);
    for (int l = 0; l < nl; ++l) {
        const int32_t* p = po + PO_LAYER(l);
        const auto& F = y.FL[l];
        pj(p[0], 64, 0, 0, 6, 8, F.f_qkv); pj(p[2], 64, 0, 0, 2, 8, F.f_out); pj(p[6], 64, 0, 0, 2, 8, F.f_ff1); pj(p[8], 64, 0, 0, 2, 8, F.f_ff2);
        pj(p[0], 64, 1, 0, 2, 24, F.t_qkv); pj(p[2], 64, 1, 0, 2, 8, F.t_out); pj(p[6], 64, 1, 0, 2, 8, F.t_ff1); pj(p[8], 64, 1, 0, 2, 8, F.t_ff2);
    }
    for (int h = 0; h < nh; ++h) {
        const int32_t* p = po + PO_HEAD(nl, h);
        pj(p[0], 64, 0, 0, 4, 8, y.FH[h].f_p1);
        pj(p[0], 64, 1, 0, 2, 16, y.FH[h].t_p1);
    }
    jobs.n = nj;
    jobs.zero = ws + y.loss;
    jobs.n_zero = 1 + TDT_MAX_HEADS;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(tdt_pack_kernel, dim3(nj + 1, 4), dim3(256), 0, st, jobs, P, fr);

    // forward
    hipLaunchKernelGGL(tdt_proj_fwd_kernel, dim3(tiles), dim3(64), 0, st, c, (const float*)(ws + y.feat), (const float*)(fr + y.f_w0),
                       P + po[1], P + po[2], P + po[3], ws + y.x[0], ws + y.xh0, ws + y.rs0, Lp[0]);
    for (int l = 0; l < nl; ++l) {
        const int has_next = l + 1 < nl;
        hipLaunchKernelGGL(tdt_layer_fwd_kernel, dim3(tiles), dim3(64), 0, st, c, Lp[l], Lp[has_next ? l + 1 : l], has_next,
                           (const float*)(ws + y.x[l]), ws + y.x[l + 1], hs);
    }
    hipLaunchKernelGGL(tdt_pool_clip_kernel, dim3(B, nh), dim3(256), 0, st, c, hs, (const float*)(ws + y.x[nl]), a->labels, a->bias_map,
                       a->inv_count, ws + y.yhat, ws + y.loss);
    // backward
    hipLaunchKernelGGL(tdt_bwd_tail_kernel, dim3(tiles), dim3(64), 0, st, c, Lp[nl - 1], hs, (const float*)(ws + y.x[nl]));
    for (int l = nl - 1; l >= 0; --l) {
        hipLaunchKernelGGL(tdt_attn_bwd_kernel, dim3(tiles, 2, TDT_ASPLIT), dim3(64), 0, st, c, Lp[l]);
        const int has_lower = l > 0;
        hipLaunchKernelGGL(tdt_bwd_mid_kernel, dim3(tiles), dim3(64), 0, st, c, Lp[l], Lp[has_lower ? l - 1 : l], has_lower,
                           (const float*)(ws + y.xh0), (const float*)(ws + y.rs0), P + po[2], (const float*)(fr + y.t_w0), ws + y.dxl0,
                           ws + y.dx0, ws + y.dx0u, ws + y.dfeat);
    }
    int rc = NQ_LAUNCH_STATUS();
    if (rc) return rc;
    // parameter gradients: one grouped split-K GEMM (d Y^T X) and one column-sum launch
    const int ksplit = NP >= 4096 ? 64 : (NP >= 1024 ? 16 : (NP >= 256 ? 4 : 1));
    rc = nisqa_gemm_f32(ws, ws, G, a->wgrad_desc, a->n_wgrad_groups, a->n_wgrad_tiles, 1, 0, ksplit, 1.0f, stream);
    if (rc) return rc;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(tdt_colsum_kernel, dim3(a->n_colsum_jobs, NP >= 2048 ? 32 : (NP >= 256 ? 4 : 1)), dim3(256), 0, st, a->colsum_jobs,
                       (const float*)ws, G);
    return NQ_LAUNCH_STATUS();
}
