// Self-attention and attention-pooling at fp32 OPERAND precision on the bf16 matrix pipe ("bf16x6") -- the kernels, layouts and
// maths of td_bf16.hip (reference nisqa/NISQA_lib.py:988-996, 1025-1040, 1171-1183) with every GEMM operand carried as THREE bf16
// terms (hi + mid + lo: an exact split of the fp32 value) and six v_mfma_f32_32x32x16_bf16 products per term pair (hh, hm, mh, hl,
// lh, mm; fp32 accumulate).  What is dropped is of the size of an fp32 multiply-add's own rounding of the product (cnn_bf16x6.hip,
// DESIGN.md 4.5 "bf16x6"), so the outputs sit where the exact-fp32 kernels of td.hip sit; the matrix-pipe time of a tile's chain
// is 2.7 x shorter than theirs (v_mfma_f32_32x32x2_f32).
//
// What changes against td_bf16.hip: every fragment set has three terms ([..][3][64 lanes][8] in the weight blobs), q / k / v are
// stored as three bf16 planes each (nine planes of np * 64 per layer buffer), the D -> B register chaining splits into three terms.
// softmax, LayerNorm, residuals, biases stay fp32 in registers.
#include "common.hpp"
#include "layout.hpp"
#include "../../include/nisqa_hip.h"

#define XT 3
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define LN_EPS 1e-5f

NQ_DEV f32x16 mfma_bf(f32x4 a, f32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
NQ_DEV unsigned cvt_pk_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// (v0, v1) -> XT packed bf16 pairs, each term rounded to nearest: v = t[0] + t[1] + t[2] exactly
NQ_DEV void split2t(float v0, float v1, unsigned (&t)[XT]) {
    f32x2_t r = {v0, v1};
#pragma unroll
    for (int q = 0; q < XT; ++q) {
        t[q] = cvt_pk_bf16(r[0], r[1]);
        if (q + 1 < XT) r = r - f32x2_t{__uint_as_float(t[q] << 16), __uint_as_float(t[q] & 0xffff0000u)};
    }
}
// 8 consecutive registers of a D fragment -> the XT B-operand terms of one K=16 step
NQ_DEV void split8t(const f32x16& a, int base, f32x4 (&b)[XT]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned t[XT];
        split2t(a[base + 2 * q], a[base + 2 * q + 1], t);
#pragma unroll
        for (int k = 0; k < XT; ++k) b[k][q] = __uint_as_float(t[k]);
    }
}

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
NQ_DEV void store_dtok(float* __restrict__ rowp, const f32x16 (&v)[2], int hf) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[mt][4 * g + e];
            *(f32x4*)(rowp + 32 * mt + 8 * g + 4 * hf) = o;
        }
}
// a 64-feature D tile as XT bf16 rows [64] (token-major planes, `plane` elements apart)
NQ_DEV void store_dtok_terms(u16* __restrict__ row0, size_t plane, const f32x16 (&v)[2], int hf, float scale) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned a[XT], b[XT];
            split2t(v[mt][4 * g] * scale, v[mt][4 * g + 1] * scale, a);
            split2t(v[mt][4 * g + 2] * scale, v[mt][4 * g + 3] * scale, b);
#pragma unroll
            for (int k = 0; k < XT; ++k) *(u32x2*)(row0 + k * plane + 32 * mt + 8 * g + 4 * hf) = u32x2{a[k], b[k]};
        }
}

// out[mt] += W * in  (W: chain-order fragments [4 steps][MTT][XT][64][8], in: D layout of a 64 x 32 tile); the fragments of a
// whole GEMM are requested by chain_load a phase ahead of chain_mma (one wave per SIMD: every load is an exposed round trip)
template <int MT>
struct chain_frags { f32x4 t[4][MT][XT]; };

template <int MT, int MTT>
NQ_DEV void chain_load(const u16* __restrict__ wb, int mt0, chain_frags<MT>& f, int lane) {
    const f32x4* af = (const f32x4*)wb + lane;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int k = 0; k < XT; ++k) f.t[s][mt][k] = af[((s * MTT + mt0 + mt) * XT + k) * 64];
}

// the six term products (weight term i, activation term j), i + j <= 2, smallest first; consecutive MFMAs on different tiles
template <int MT>
NQ_DEV void mma_terms(const f32x4 (&a)[MT][XT], const f32x4 (&b)[XT], f32x16 (&out)[MT]) {
#pragma unroll
    for (int order = XT - 1; order >= 0; --order)
#pragma unroll
        for (int i = order; i >= 0; --i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) out[mt] = mfma_bf(a[mt][i], b[order - i], out[mt]);
}

template <int MT>
NQ_DEV void chain_mma(const chain_frags<MT>& f, const f32x16 (&in)[2], f32x16 (&out)[MT]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        f32x4 b[XT];
        split8t(in[s >> 1], 8 * (s & 1), b);
        mma_terms<MT>(f.t[s], b, out);
    }
}

// LayerNorm over the 64 features of each token (gamma / beta preloaded in D layout)
NQ_DEV void layernorm64(f32x16 (&x)[2], const f32x16 (&g)[2], const f32x16 (&bt)[2]) {
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
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[mt][r] = (x[mt][r] - mean) * rstd * g[mt][r] + bt[mt][r];
}

// nine bf16 planes of np * 64 per layer buffer: q[XT], k[XT] as [NP][64]; v[XT] as [64][NP]  (= 4.5 * np * 64 floats)
struct qkv_planes { u16 *q, *k, *v; size_t plane; };
NQ_DEV qkv_planes planes_of(float* base, size_t np64) {
    u16* p = (u16*)base;
    qkv_planes r;
    r.q = p; r.k = p + XT * np64; r.v = p + 2 * XT * np64; r.plane = np64;
    return r;
}

// Q, K, V of a layer as three 64-wide chain GEMMs; qkv_prefetch (the q fragments) is issued by the caller a phase ahead
struct qkv_pre { chain_frags<2> f; f32x16 bias[2]; };
NQ_DEV void qkv_prefetch(const float* __restrict__ lw, const u16* __restrict__ lwx, qkv_pre& p, int lane) {
    chain_load<2, 6>(lwx + TDXL_QKV, 0, p.f, lane);
    load_dvec<2>(lw + TDL_QKV_B, p.bias, lane >> 5);
}
NQ_DEV void qkv_store(const float* __restrict__ lw, const u16* __restrict__ lwx, const f32x16 (&x)[2], const qkv_planes& P,
                      int tok, int np, int lane, qkv_pre& q) {
    const int hf = lane >> 5;
    qkv_pre k;
    chain_load<2, 6>(lwx + TDXL_QKV, 2, k.f, lane);
    load_dvec<2>(lw + TDL_QKV_B + 64, k.bias, hf);
    chain_mma<2>(q.f, x, q.bias);
    qkv_pre v;
    chain_load<2, 6>(lwx + TDXL_QKV, 4, v.f, lane);
    load_dvec<2>(lw + TDL_QKV_B + 128, v.bias, hf);
    store_dtok_terms(P.q + (size_t)tok * 64, P.plane, q.bias, hf, 0.125f);       // (a power of two: the split of q / 8 is the split of q)
    chain_mma<2>(k.f, x, k.bias);
    store_dtok_terms(P.k + (size_t)tok * 64, P.plane, k.bias, hf, 1.0f);
    chain_mma<2>(v.f, x, v.bias);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            unsigned t[XT];
            split2t(v.bias[mt][r], v.bias[mt][r + 1], t);
            const size_t f0 = (size_t)(32 * mt + NQ_DROW(r, hf)) * np + tok, f1 = f0 + np;   // rows r, r+1 are adjacent features
#pragma unroll
            for (int kk = 0; kk < XT; ++kk) {
                P.v[kk * P.plane + f0] = (u16)t[kk];
                P.v[kk * P.plane + f1] = (u16)(t[kk] >> 16);
            }
        }
}

// Linear 384 -> 64 + LayerNorm + layer-0 QKV, one wave per 32-token tile
__global__ __launch_bounds__(64) void td_proj_bf16x6_kernel(const float* __restrict__ feat, const int32_t* __restrict__ tok_off,
                                                            const int32_t* __restrict__ n_wins, int n_clips, int np,
                                                            const float* __restrict__ tw, const u16* __restrict__ twx,
                                                            float* __restrict__ x, float* __restrict__ qkv) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int tile0 = xcd_tile(blockIdx.x, gridDim.x) * 32;
    const int b = find_segment_wave(tok_off, n_clips, tile0, lane);
    const int n = n_wins[b], k0 = tile0 - tok_off[b];
    if (k0 >= n) return;
    const int tok = tile0 + j;
    const bool valid = k0 + j < n;
    const f32x4* frow = (const f32x4*)(feat + (size_t)tok * 384);
    const f32x4* af = (const f32x4*)(twx + TDX_PROJ) + lane;
    f32x16 acc[2], g0[2], b0[2];
    load_dvec<2>(tw + TD_PROJ_B, acc, h);
    load_dvec<2>(tw + TD_LN0_G, g0, h);
    load_dvec<2>(tw + TD_LN0_B, b0, h);
    // feature rows 8 steps and weight fragments 4 steps ahead of their MFMAs (td_bf16.hip)
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 fr[8][2], wr[4][2][XT];
    auto load_f = [&](int s) {
        fr[s & 7][0] = valid ? frow[4 * s + 2 * h] : z4;
        fr[s & 7][1] = valid ? frow[4 * s + 2 * h + 1] : z4;
    };
    auto load_w = [&](int s) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int k = 0; k < XT; ++k) wr[s & 3][mt][k] = af[((s * 2 + mt) * XT + k) * 64];
    };
#pragma unroll
    for (int s = 0; s < 4; ++s) load_w(s);
#pragma unroll
    for (int s = 0; s < 8; ++s) load_f(s);
#pragma unroll
    for (int s = 0; s < 24; ++s) {
        const f32x4 f0 = fr[s & 7][0], f1 = fr[s & 7][1];
        f32x4 bt[XT];
        unsigned t[XT];
        split2t(f0[0], f0[1], t);
#pragma unroll
        for (int k = 0; k < XT; ++k) bt[k][0] = __uint_as_float(t[k]);
        split2t(f0[2], f0[3], t);
#pragma unroll
        for (int k = 0; k < XT; ++k) bt[k][1] = __uint_as_float(t[k]);
        split2t(f1[0], f1[1], t);
#pragma unroll
        for (int k = 0; k < XT; ++k) bt[k][2] = __uint_as_float(t[k]);
        split2t(f1[2], f1[3], t);
#pragma unroll
        for (int k = 0; k < XT; ++k) bt[k][3] = __uint_as_float(t[k]);
        mma_terms<2>(wr[s & 3], bt, acc);
        if (s + 4 < 24) load_w(s + 4);
        if (s + 8 < 24) load_f(s + 8);
    }
    qkv_pre qp;
    qkv_prefetch(tw + TD_LAYER0, twx + TDX_LAYER0, qp, lane);
    layernorm64(acc, g0, b0);
    store_dtok(x + (size_t)tok * 64, acc, h);
    qkv_store(tw + TD_LAYER0, twx + TDX_LAYER0, acc, planes_of(qkv, (size_t)np * 64), tok, np, lane, qp);
}

__global__ __launch_bounds__(64) void td_layer_bf16x6_kernel(const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
                                                             int n_clips, int np, const float* __restrict__ lw,
                                                             const u16* __restrict__ lwx, const float* __restrict__ lw_next,
                                                             const u16* __restrict__ lwx_next, const float* x_in, float* qkv_cur,
                                                             float* x_out, float* qkv_next) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int tile0 = xcd_tile(blockIdx.x, gridDim.x) * 32;
    const int b = find_segment_wave(tok_off, n_clips, tile0, lane);
    const int n = n_wins[b], c0 = tok_off[b];
    if (tile0 - c0 >= n) return;
    const int tok = tile0 + j;
    const qkv_planes P = planes_of(qkv_cur, (size_t)np * 64);

    f32x4 q[4][XT];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k = 0; k < XT; ++k) q[s][k] = *(const f32x4*)(P.q + k * P.plane + (size_t)tok * 64 + 16 * s + 8 * h);
    f32x16 o[2];
    o[0] = zero16(); o[1] = zero16();
    float m = -INFINITY, l = 0.f;
    const int nkt = (n + 31) >> 5;
    f32x4 kA[4][1][XT], kB[4][1][XT];                      // ([step][one M tile][term]: the A operand of mma_terms<1>)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k = 0; k < XT; ++k) kA[s][0][k] = *(const f32x4*)(P.k + k * P.plane + (size_t)(c0 + j) * 64 + 16 * s + 8 * h);
    auto tile = [&](int kt, const f32x4 (&kc)[4][1][XT], f32x4 (&kn)[4][1][XT]) {
        const int key0 = c0 + 32 * kt;
        // V^T fragments: element e of lane half h <-> key 16 s + (e&3) + 8 (e>>2) + 4 h (the rows of P this half owns)
        f32x4 vt[2][2][XT];                                // [K step s][feature tile ft][term]
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int k = 0; k < XT; ++k) {
                    const size_t base = k * P.plane + (size_t)(j + 32 * ft) * np + key0 + 16 * s + 4 * h;
                    const u32x2 a0 = *(const u32x2*)(P.v + base), a1 = *(const u32x2*)(P.v + base + 8);
                    vt[s][ft][k] = f32x4{__uint_as_float(a0[0]), __uint_as_float(a0[1]), __uint_as_float(a1[0]), __uint_as_float(a1[1])};
                }
        if (kt + 1 < nkt) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int k = 0; k < XT; ++k) kn[s][0][k] = *(const f32x4*)(P.k + k * P.plane + (size_t)(key0 + 32 + j) * 64 + 16 * s + 8 * h);
        }
        __builtin_amdgcn_sched_barrier(0);     // K / V requests of the next tile stay ahead of this tile's MFMAs
        // S^T = K Q^T: one accumulator per product ORDER (independent chains), added smallest first
        f32x16 sa[XT];
#pragma unroll
        for (int k = 0; k < XT; ++k) sa[k] = zero16();
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int order = XT - 1; order >= 0; --order)
#pragma unroll
                for (int i = order; i >= 0; --i) sa[order] = mfma_bf(kc[s][0][i], q[s][order - i], sa[order]);
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = (sa[2][r] + sa[1][r]) + sa[0][r];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (32 * kt + NQ_DROW(r, h) >= n) sacc[r] = -INFINITY;
            mx = fmaxf(mx, sacc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f((m - m_new) * 1.44269504088896341f);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = __builtin_amdgcn_exp2f((sacc[r] - m_new) * 1.44269504088896341f);
            rs += sacc[r];
        }
        rs += __shfl_xor(rs, 32);
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 p[XT];
            split8t(sacc, 8 * s, p);
            mma_terms<2>(vt[s], p, o);
        }
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile(kt, kA, kB);
        if (kt + 1 < nkt) tile(kt + 1, kB, kA);
    }
    // ---- out-projection, residual + LN1, feed-forward, residual + LN2, next layer's QKV: every parameter block is
    //      requested one phase before its use
    chain_frags<2> fa, fb;
    f32x16 y[2], xr[2], g[2], bt[2], h1[2], h2[2];
    chain_load<2, 2>(lwx + TDXL_OUT, 0, fa, lane);
    load_dvec<2>(lw + TDL_OUT_B, y, h);
    load_dvec<2>(x_in + (size_t)tok * 64, xr, h);
    const float inv_l = 1.0f / l;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= inv_l; o[1][r] *= inv_l; }
    chain_mma<2>(fa, o, y);
    chain_load<2, 2>(lwx + TDXL_FF1, 0, fb, lane);
    load_dvec<2>(lw + TDL_LN1_G, g, h);
    load_dvec<2>(lw + TDL_LN1_B, bt, h);
    load_dvec<2>(lw + TDL_FF1_B, h1, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[0][r] += xr[0][r]; y[1][r] += xr[1][r]; }
    layernorm64(y, g, bt);
    chain_mma<2>(fb, y, h1);
    chain_load<2, 2>(lwx + TDXL_FF2, 0, fa, lane);
    load_dvec<2>(lw + TDL_FF2_B, h2, h);
    load_dvec<2>(lw + TDL_LN2_G, g, h);
    load_dvec<2>(lw + TDL_LN2_B, bt, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) { h1[0][r] = fmaxf(h1[0][r], 0.f); h1[1][r] = fmaxf(h1[1][r], 0.f); }
    chain_mma<2>(fa, h1, h2);
    qkv_pre qp;
    if (lw_next) qkv_prefetch(lw_next, lwx_next, qp, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[0][r] += h2[0][r]; y[1][r] += h2[1][r]; }
    layernorm64(y, g, bt);
    store_dtok(x_out + (size_t)tok * 64, y, h);
    if (lw_next) qkv_store(lw_next, lwx_next, y, planes_of(qkv_next, (size_t)np * 64), tok, np, lane, qp);
}

__global__ __launch_bounds__(64) void pool_score_bf16x6_kernel(const float* __restrict__ x, const int32_t* __restrict__ tok_off,
                                                               const int32_t* __restrict__ n_wins, int n_clips, int n_heads,
                                                               const float* __restrict__ pw, const u16* __restrict__ pwx,
                                                               float* __restrict__ sc, float* __restrict__ yv) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int tile0 = xcd_tile(blockIdx.x, gridDim.x) * 32;
    const int b = find_segment_wave(tok_off, n_clips, tile0, lane);
    if (tile0 - tok_off[b] >= n_wins[b]) return;
    const int tok = tile0 + j;
    f32x16 xr[2];
    load_dvec<2>(x + (size_t)tok * 64, xr, h);
    const int hd = blockIdx.y;                               // one wave per (tile, head): the heads are independent
    const float* w = pw + (size_t)hd * PL_FLOATS;
    f32x16 hid[4], w2[4], w3[2];
    chain_frags<4> f;
    chain_load<4, 4>(pwx + (size_t)hd * PLX_U16S, 0, f, lane);
    load_dvec<4>(w + PL_B1, hid, h);
    load_dvec<4>(w + PL_W2, w2, h);
    load_dvec<2>(w + PL_W3, w3, h);
    chain_mma<4>(f, xr, hid);
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

// ws: 9 * np * 64 floats (two layer buffers of nine bf16 planes each)
extern "C" int nisqa_td_selfatt_bf16x6(const float* feat, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                                       int32_t total_tok_padded, int32_t n_layers, const float* td_w, const uint16_t* td_wx,
                                       float* ws, float* x_out, void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31) || n_layers < 1 || !td_wx) return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int np = total_tok_padded;
    const size_t sz = (size_t)np * 64;
    float* qkv[2] = {ws, ws + (9 * sz + 1) / 2};         // each: nine bf16 planes of np * 64 = 4.5 * sz floats
    const int tiles = np / 32;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(td_proj_bf16x6_kernel, dim3(tiles), dim3(64), 0, st, feat, tok_off, n_wins, n_clips, np, td_w, td_wx,
                       x_out, qkv[0]);
    for (int l = 0; l < n_layers; ++l) {
        const float* lw = td_w + TD_LAYER0 + (size_t)l * TDL_FLOATS;
        const uint16_t* lwx = td_wx + TDX_LAYER0 + (size_t)l * TDXL_U16S;
        const bool more = l + 1 < n_layers;
        hipLaunchKernelGGL(td_layer_bf16x6_kernel, dim3(tiles), dim3(64), 0, st, tok_off, n_wins, n_clips, np, lw, lwx,
                           more ? lw + TDL_FLOATS : (const float*)nullptr, more ? lwx + TDXL_U16S : (const uint16_t*)nullptr,
                           (const float*)x_out, qkv[l & 1], x_out, qkv[(l & 1) ^ 1]);
    }
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_pool_score_bf16x6(const float* x, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                                       int32_t total_tok_padded, int32_t n_heads, const float* pool_w, const uint16_t* pool_wx,
                                       float* ws, void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31) || n_heads < 1 || n_heads > 8 || !pool_wx)
        return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(pool_score_bf16x6_kernel, dim3(total_tok_padded / 32, n_heads), dim3(64), 0, (hipStream_t)stream, x, tok_off,
                       n_wins, n_clips, n_heads, pool_w, pool_wx, ws, ws + (size_t)total_tok_padded * 8);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_pool_att_bf16x6(const float* x, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                                     int32_t total_tok_padded, int32_t n_heads, const float* pool_w, const uint16_t* pool_wx,
                                     float* ws, float* out, void* stream) {
    const int rc = nisqa_pool_score_bf16x6(x, tok_off, n_wins, n_clips, total_tok_padded, n_heads, pool_w, pool_wx, ws, stream);
    if (rc) return rc;
    return nisqa_pool_final(tok_off, n_wins, n_clips, total_tok_padded, n_heads, ws, out, stream);
}
