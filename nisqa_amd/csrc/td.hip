// Time-dependency (self-attention) and attention-pooling heads for gfx950 -- replaces
// SelfAttention.forward / SelfAttentionLayer.forward (reference nisqa/NISQA_lib.py:988-996,
// 1025-1040) and PoolAttFF.forward x n_heads (NISQA_lib.py:1171-1183).
//
// Everything is "features x tokens": a wave owns a tile of 32 tokens (MFMA columns = lanes) and
// keeps each token's 64 features in registers (MFMA rows), so
//   * every weight matrix is the A operand, streamed as pre-packed fragments from L2;
//   * the D fragment of one GEMM IS the B operand of the next (k-pairs are chosen as the two lane
//     halves' rows of the same register), so Linear -> LayerNorm -> QKV, and attention -> out-proj
//     -> residual+LN -> FFN -> residual+LN -> next layer's QKV chain through registers only;
//   * LayerNorm / softmax reductions are in-lane sums over registers plus ONE lane^32 exchange;
//   * S^T = K Q^T is computed "swapped" (keys on rows) so that the softmax P^T feeds the
//     O^T = V^T P^T MFMAs directly from the accumulator registers (flash-style online softmax,
//     never materialising the L x L score matrix; keys >= n_wins are masked to -inf exactly like
//     the reference's key_padding_mask).
// Tokens are stored padded to 32 per clip (tok_off), V is stored feature-major (vT[64][NP]) so the
// P.V operand loads are aligned float4s along the key axis.
#include "common.hpp"
#include "layout.hpp"
#include "../../include/nisqa_hip.h"

#define LN_EPS 1e-5f

// vec[f] for this lane's D-layout features f = 32*mt + (r&3) + 8*(r>>2) + 4*hf
template <int MT>
NQ_DEV void load_dvec(const float* __restrict__ base, f32x16 (&out)[MT], int hf) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4*)(base + 32 * mt + 8 * g + 4 * hf);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[mt][4 * g + e] = v[e];
        }
}

// token-major [tok][64] rows <-> D layout
NQ_DEV void load_dtok(const float* __restrict__ rowp, f32x16 (&out)[2], int hf) { load_dvec<2>(rowp, out, hf); }
NQ_DEV void store_dtok(float* __restrict__ rowp, const f32x16 (&v)[2], int hf, float scale) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[mt][4 * g + e] * scale;
            *(f32x4*)(rowp + 32 * mt + 8 * g + 4 * hf) = o;
        }
}

// out[mt] += W[64*... rows][64] * in   (in = D layout of a 64-feature x 32-token tile)
template <int MT>
NQ_DEV void chain_gemm64(const f32x4* __restrict__ af, const f32x16 (&in)[2], f32x16 (&out)[MT], int lane) {
    // weight fragments stream from L2: request step s+1 before issuing the MFMAs of step s
    f32x4 a[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[0][mt] = af[mt * 64 + lane];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s + 1 < 8) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[(s + 1) & 1][mt] = af[((s + 1) * MT + mt) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                out[mt] = mfma32(a[s & 1][mt][kk], in[s >> 2][4 * (s & 3) + kk], out[mt]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// LayerNorm over the 64 features of each token (32 in this lane, 32 in lane^32)
NQ_DEV void layernorm64(f32x16 (&x)[2], const float* __restrict__ gamma, const float* __restrict__ beta, int hf) {
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
    const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + LN_EPS);
    f32x16 g[2], bt[2];
    load_dvec<2>(gamma, g, hf);
    load_dvec<2>(beta, bt, hf);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[mt][r] = (x[mt][r] - mean) * rstd * g[mt][r] + bt[mt][r];
}

// QKV projection of a register-resident x tile; writes q (pre-scaled by 1/sqrt(64)), k, v^T
NQ_DEV void qkv_store(const float* __restrict__ lw, const f32x16 (&x)[2], float* __restrict__ q, float* __restrict__ k,
                      float* __restrict__ vT, int tok, int np, int lane) {
    const int hf = lane >> 5;
    f32x16 acc[6];
    load_dvec<6>(lw + TDL_QKV_B, acc, hf);             // bias as accumulator init
    chain_gemm64<6>((const f32x4*)(lw + TDL_QKV_AF), x, acc, lane);
    f32x16 t2[2];
    t2[0] = acc[0]; t2[1] = acc[1];
    store_dtok(q + (size_t)tok * 64, t2, hf, 0.125f);  // (q W + b) * head_dim^-0.5, head_dim = 64
    t2[0] = acc[2]; t2[1] = acc[3];
    store_dtok(k + (size_t)tok * 64, t2, hf, 1.0f);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) vT[(size_t)(32 * mt + NQ_DROW(r, hf)) * np + tok] = acc[4 + mt][r];
}

// ---------------------------------------------------------------------------------------------
// Linear 384->64 + LayerNorm (NISQA_lib.py:989-991) + layer-0 QKV.  One wave per 32-token tile.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void td_proj_kernel(const float* __restrict__ feat, const int32_t* __restrict__ tok_off,
                                                     const int32_t* __restrict__ n_wins, int n_clips, int np,
                                                     const float* __restrict__ tw, float* __restrict__ x,
                                                     float* __restrict__ q, float* __restrict__ k, float* __restrict__ vT) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int tile0 = blockIdx.x * 32;
    const int b = find_segment(tok_off, n_clips, tile0);
    const int n = n_wins[b], k0 = tile0 - tok_off[b];
    if (k0 >= n) return;                                   // tile is all padding
    const int tok = tile0 + j;
    const bool valid = k0 + j < n;
    const f32x4* frow = (const f32x4*)(feat + (size_t)tok * 384);
    const f32x4* af = (const f32x4*)(tw + TD_PROJ_AF);
    f32x16 acc[2];
    load_dvec<2>(tw + TD_PROJ_B, acc, h);
#pragma unroll 4
    for (int s = 0; s < 48; ++s) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (valid) bv = frow[2 * s + h];                  // padding tokens project a zero row
        const f32x4 a0 = af[(s * 2 + 0) * 64 + lane], a1 = af[(s * 2 + 1) * 64 + lane];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc[0] = mfma32(a0[kk], bv[kk], acc[0]);
            acc[1] = mfma32(a1[kk], bv[kk], acc[1]);
        }
    }
    layernorm64(acc, tw + TD_LN0_G, tw + TD_LN0_B, h);
    store_dtok(x + (size_t)tok * 64, acc, h, 1.0f);
    qkv_store(tw + TD_LAYER0, acc, q, k, vT, tok, np, lane);
}

// ---------------------------------------------------------------------------------------------
// One SelfAttentionLayer (NISQA_lib.py:1025-1040) for a 32-query tile, plus the next layer's QKV.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void td_layer_kernel(const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
                                                      int n_clips, int np, const float* __restrict__ lw,
                                                      const float* __restrict__ lw_next, const float* x_in,
                                                      const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ vT, float* x_out,
                                                      float* __restrict__ q_n, float* __restrict__ k_n,
                                                      float* __restrict__ vT_n) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int tile0 = blockIdx.x * 32;
    const int b = find_segment(tok_off, n_clips, tile0);
    const int n = n_wins[b], c0 = tok_off[b];
    if (tile0 - c0 >= n) return;
    const int tok = tile0 + j;

    f32x4 qf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *(const f32x4*)(q + (size_t)tok * 64 + 8 * s + 4 * h);

    f32x16 o[2];
    o[0] = zero16(); o[1] = zero16();
    float m = -INFINITY, l = 0.f;
    const int nkt = (n + 31) >> 5;
    // K rows (A operand of S^T = K Q^T) and V^T rows (A operand of O^T = V^T P^T) are register-prefetched:
    // the next tile's K and this tile's V are requested before this tile's QK^T MFMAs are issued
    f32x4 kA[8], kB[8];
    {
        const float* krow = k + (size_t)(c0 + j) * 64 + 4 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) kA[s] = *(const f32x4*)(krow + 8 * s);
    }
    auto tile = [&](int kt, const f32x4 (&kcur)[8], f32x4 (&knext)[8]) {
        const int key0 = c0 + 32 * kt;
        f32x4 vf[2][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            vf[0][g] = *(const f32x4*)(vT + (size_t)j * np + key0 + 8 * g + 4 * h);
            vf[1][g] = *(const f32x4*)(vT + (size_t)(j + 32) * np + key0 + 8 * g + 4 * h);
        }
        if (kt + 1 < nkt) {
            const float* krow = k + (size_t)(key0 + 32 + j) * 64 + 4 * h;
#pragma unroll
            for (int s = 0; s < 8; ++s) knext[s] = *(const f32x4*)(krow + 8 * s);
        }
        __builtin_amdgcn_sched_barrier(0);
        // S^T tile: rows = keys, cols = queries
        f32x16 sacc = zero16();
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) sacc = mfma32(kcur[s][kk], qf[s][kk], sacc);
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (32 * kt + NQ_DROW(r, h) >= n) sacc[r] = -INFINITY;   // key_padding_mask
            mx = fmaxf(mx, sacc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);                 // finite: key 32*kt is always valid
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
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        // O^T += V^T P^T : A = vT[feature][keys], B = P^T straight from the accumulator
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

    // out_proj + residual + LayerNorm1
    f32x16 y[2], xr[2];
    load_dvec<2>(lw + TDL_OUT_B, y, h);
    chain_gemm64<2>((const f32x4*)(lw + TDL_OUT_AF), o, y, lane);
    load_dtok(x_in + (size_t)tok * 64, xr, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[0][r] += xr[0][r]; y[1][r] += xr[1][r]; }
    layernorm64(y, lw + TDL_LN1_G, lw + TDL_LN1_B, h);
    // FFN (ReLU) + residual + LayerNorm2
    f32x16 h1[2], h2[2];
    load_dvec<2>(lw + TDL_FF1_B, h1, h);
    chain_gemm64<2>((const f32x4*)(lw + TDL_FF1_AF), y, h1, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) { h1[0][r] = fmaxf(h1[0][r], 0.f); h1[1][r] = fmaxf(h1[1][r], 0.f); }
    load_dvec<2>(lw + TDL_FF2_B, h2, h);
    chain_gemm64<2>((const f32x4*)(lw + TDL_FF2_AF), h1, h2, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[0][r] += h2[0][r]; y[1][r] += h2[1][r]; }
    layernorm64(y, lw + TDL_LN2_G, lw + TDL_LN2_B, h);
    store_dtok(x_out + (size_t)tok * 64, y, h, 1.0f);
    if (lw_next) qkv_store(lw_next, y, q_n, k_n, vT_n, tok, np, lane);
}

// ---------------------------------------------------------------------------------------------
// Attention pooling (NISQA_lib.py:1171-1183), two passes:
//   1) per 32-token tile and head: score = w2 . relu(W1 x + b1) + b2 and yv = w3 . x + b3
//   2) per clip and head: softmax over the valid tokens, out = sum_t a_t * yv_t
//      ( = linear3(sum_t a_t x_t) because sum_t a_t = 1 )
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pool_score_kernel(const float* __restrict__ x, const int32_t* __restrict__ tok_off,
                                                        const int32_t* __restrict__ n_wins, int n_clips, int n_heads,
                                                        const float* __restrict__ pw, float* __restrict__ sc,
                                                        float* __restrict__ yv) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int tile0 = blockIdx.x * 32;
    const int b = find_segment(tok_off, n_clips, tile0);
    if (tile0 - tok_off[b] >= n_wins[b]) return;
    const int tok = tile0 + j;
    f32x16 xr[2];
    load_dtok(x + (size_t)tok * 64, xr, h);
    for (int hd = 0; hd < n_heads; ++hd) {
        const float* w = pw + (size_t)hd * PL_FLOATS;
        f32x16 hid[4], w2[4], w3[2];
        load_dvec<4>(w + PL_B1, hid, h);
        chain_gemm64<4>((const f32x4*)(w + PL_W1_AF), xr, hid, lane);
        load_dvec<4>(w + PL_W2, w2, h);
        load_dvec<2>(w + PL_W3, w3, h);
        float s = 0.f, v = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s = fmaf(w2[mt][r], fmaxf(hid[mt][r], 0.f), s);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) v = fmaf(w3[mt][r], xr[mt][r], v);
        s += __shfl_xor(s, 32);
        v += __shfl_xor(v, 32);
        if (h == 0) {
            sc[(size_t)tok * 8 + hd] = s + w[PL_B2];
            yv[(size_t)tok * 8 + hd] = v + w[PL_B2 + 1];
        }
    }
}

__global__ __launch_bounds__(64) void pool_final_kernel(const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
                                                        int n_heads, const float* __restrict__ sc,
                                                        const float* __restrict__ yv, float* __restrict__ out) {
    const int lane = threadIdx.x, b = blockIdx.x;
    const int n = n_wins[b], c0 = tok_off[b];
    {
        const int hd = blockIdx.y;                           // one wave per (clip, head)
        float mx = -INFINITY;
        for (int t = lane; t < n; t += 64) mx = fmaxf(mx, sc[(size_t)(c0 + t) * 8 + hd]);
        mx = wave_max(mx);
        float den = 0.f, num = 0.f;
        for (int t = lane; t < n; t += 64) {
            const float e = expf(sc[(size_t)(c0 + t) * 8 + hd] - mx);
            den += e;
            num = fmaf(e, yv[(size_t)(c0 + t) * 8 + hd], num);
        }
        den = wave_sum(den);
        num = wave_sum(num);
        if (lane == 0) out[(size_t)b * n_heads + hd] = num / den;
    }
}

extern "C" int nisqa_td_selfatt(const float* feat, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                                int32_t total_tok_padded, int32_t n_layers, const float* td_w, float* ws,
                                float* x_out, void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31) || n_layers < 1) return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int np = total_tok_padded;
    const size_t sz = (size_t)np * 64;
    float* qb[2] = {ws, ws + 3 * sz};
    float* kb[2] = {ws + sz, ws + 4 * sz};
    float* vb[2] = {ws + 2 * sz, ws + 5 * sz};
    const int tiles = np / 32;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(td_proj_kernel, dim3(tiles), dim3(64), 0, st, feat, tok_off, n_wins, n_clips, np, td_w, x_out,
                       qb[0], kb[0], vb[0]);
    for (int l = 0; l < n_layers; ++l) {
        const float* lw = td_w + TD_LAYER0 + (size_t)l * TDL_FLOATS;
        const float* lwn = (l + 1 < n_layers) ? lw + TDL_FLOATS : nullptr;
        const int c = l & 1, nx = c ^ 1;
        hipLaunchKernelGGL(td_layer_kernel, dim3(tiles), dim3(64), 0, st, tok_off, n_wins, n_clips, np, lw, lwn,
                           (const float*)x_out, (const float*)qb[c], (const float*)kb[c], (const float*)vb[c], x_out,
                           qb[nx], kb[nx], vb[nx]);
    }
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_pool_final(const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips, int32_t total_tok_padded,
                                int32_t n_heads, const float* ws, float* out, void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || n_heads < 1 || n_heads > 8) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(pool_final_kernel, dim3(n_clips, n_heads), dim3(64), 0, (hipStream_t)stream, tok_off, n_wins, n_heads, ws,
                       ws + (size_t)total_tok_padded * 8, out);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_pool_att(const float* x, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                              int32_t total_tok_padded, int32_t n_heads, const float* pool_w, float* ws, float* out,
                              void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31) || n_heads < 1 || n_heads > 8)
        return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* sc = ws;
    float* yv = ws + (size_t)total_tok_padded * 8;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(pool_score_kernel, dim3(total_tok_padded / 32), dim3(64), 0, st, x, tok_off, n_wins, n_clips,
                       n_heads, pool_w, sc, yv);
    hipLaunchKernelGGL(pool_final_kernel, dim3(n_clips, n_heads), dim3(64), 0, st, tok_off, n_wins, n_heads,
                       (const float*)sc, (const float*)yv, out);
    return NQ_LAUNCH_STATUS();
}
