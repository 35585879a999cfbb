// Self-attention and attention-pooling on split-bf16 MFMA -- same kernels, layouts and maths as td.hip
// (reference nisqa/NISQA_lib.py:988-996, 1025-1040, 1171-1183) with every GEMM as three
// v_mfma_f32_32x32x16_bf16 products of bf16 hi/lo operands (fp32 accumulate; |dMOS| ~ 1e-5).
//
// What changes against the fp32 version:
//   * a K-step is 16 wide; lane half h supplies 8 k-slots.  For operands that come from MEMORY the slots are
//     8 consecutive indices (16 B of bf16).  For the register-chained GEMMs the slots of lane half h are the
//     rows this half already owns in the previous D fragment, 16s + (e&3) + 8(e>>2) + 4h, so the accumulator is
//     split into (hi, lo) in place with v_cvt_pk_bf16_f32 and fed back as the B operand -- no data movement;
//     the weight fragments are packed in the matching order on the host;
//   * q, k are stored as bf16 hi/lo planes [tok][64], v as hi/lo planes [64][tok] (same bytes as fp32);
//   * softmax, LayerNorm, residuals, biases stay fp32 in registers.
#include "common.hpp"
#include "layout.hpp"
#include "../../include/nisqa_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define LN_EPS 1e-5f

NQ_DEV f32x16 mfma_bf(f32x4 a, f32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// fp32 -> bf16 (round to nearest even), two values per instruction: the compiler selects v_cvt_pk_bf16_f32 for
// this conversion, and -- unlike an inline-asm statement -- tracks its hazards and schedules around it
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
NQ_DEV unsigned cvt_pk_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// (v0, v1) -> packed bf16 hi pair and packed bf16 lo pair
NQ_DEV void split2(float v0, float v1, unsigned& hi, unsigned& lo) {
    hi = cvt_pk_bf16(v0, v1);
    lo = cvt_pk_bf16(v0 - __uint_as_float(hi << 16), v1 - __uint_as_float(hi & 0xffff0000u));
}
// 8 consecutive registers of a D fragment -> B-operand (hi, lo) of one K=16 step
NQ_DEV void split8(const f32x16& a, int base, f32x4& hi, f32x4& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned h_, l_;
        split2(a[base + 2 * q], a[base + 2 * q + 1], h_, l_);
        hi[q] = __uint_as_float(h_);
        lo[q] = __uint_as_float(l_);
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
// a 64-feature D tile as bf16 hi / lo rows [64] (token-major planes)
NQ_DEV void store_dtok_split(u16* __restrict__ hi_row, u16* __restrict__ lo_row, const f32x16 (&v)[2], int hf, float scale) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned h0, l0, h1, l1;
            split2(v[mt][4 * g] * scale, v[mt][4 * g + 1] * scale, h0, l0);
            split2(v[mt][4 * g + 2] * scale, v[mt][4 * g + 3] * scale, h1, l1);
            *(u32x2*)(hi_row + 32 * mt + 8 * g + 4 * hf) = u32x2{h0, h1};
            *(u32x2*)(lo_row + 32 * mt + 8 * g + 4 * hf) = u32x2{l0, l1};
        }
}

// out[mt] += W * in  (W: chain-order bf16 fragments [4 steps][MTT][hl][64][8], in: D layout of a 64 x 32 tile).
// One wave runs per SIMD, so every global load is an exposed round trip unless it is requested a phase ahead: the
// fragments of a whole GEMM (tiles mt0 .. mt0+MT-1 of MTT) are loaded by chain_load -- which the caller issues before
// the VALU work that precedes the GEMM -- and consumed by chain_mma.
template <int MT>
struct chain_frags { f32x4 h[4][MT], l[4][MT]; };

template <int MT, int MTT>
NQ_DEV void chain_load(const u16* __restrict__ wb, int mt0, chain_frags<MT>& f, int lane) {
    const f32x4* af = (const f32x4*)wb + lane;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f.h[s][mt] = af[((s * MTT + mt0 + mt) * 2 + 0) * 64];
            f.l[s][mt] = af[((s * MTT + mt0 + mt) * 2 + 1) * 64];
        }
}

template <int MT>
NQ_DEV void chain_mma(const chain_frags<MT>& f, const f32x16 (&in)[2], f32x16 (&out)[MT]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        f32x4 bh, bl;
        split8(in[s >> 1], 8 * (s & 1), bh, bl);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) out[mt] = mfma_bf(f.h[s][mt], bl, out[mt]);   // product-major: consecutive
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) out[mt] = mfma_bf(f.l[s][mt], bh, out[mt]);   // MFMAs on different
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) out[mt] = mfma_bf(f.h[s][mt], bh, out[mt]);   // accumulators
    }
}

template <int MT>
NQ_DEV void chain_gemm_bf(const u16* __restrict__ wb, const f32x16 (&in)[2], f32x16 (&out)[MT], int lane) {
    chain_frags<MT> f;
    chain_load<MT, MT>(wb, 0, f, lane);
    chain_mma<MT>(f, in, out);
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
struct qkv_planes { u16 *qh, *ql, *kh, *kl, *vh, *vl; };   // q,k: [NP][64]; v: [64][NP]
NQ_DEV qkv_planes planes_of(float* base, size_t np64) {
    u16* p = (u16*)base;                                    // 6 planes of np*64 bf16 = 3 * np64 floats
    qkv_planes r;
    r.qh = p; r.ql = p + np64; r.kh = p + 2 * np64; r.kl = p + 3 * np64; r.vh = p + 4 * np64; r.vl = p + 5 * np64;
    return r;
}

// Q, K, V of a layer as three 64-wide chain GEMMs: the fragments and biases of the next one are in flight while the
// previous one's results are split and stored.  qkv_prefetch (the q fragments) is issued by the caller a phase ahead.
struct qkv_pre { chain_frags<2> f; f32x16 bias[2]; };
NQ_DEV void qkv_prefetch(const float* __restrict__ lw, const u16* __restrict__ lwb, qkv_pre& p, int lane) {
    chain_load<2, 6>(lwb + TDBL_QKV, 0, p.f, lane);
    load_dvec<2>(lw + TDL_QKV_B, p.bias, lane >> 5);
}
NQ_DEV void qkv_store_bf(const float* __restrict__ lw, const u16* __restrict__ lwb, const f32x16 (&x)[2],
                         const qkv_planes& P, int tok, int np, int lane, qkv_pre& q) {
    const int hf = lane >> 5;
    qkv_pre k;
    chain_load<2, 6>(lwb + TDBL_QKV, 2, k.f, lane);
    load_dvec<2>(lw + TDL_QKV_B + 64, k.bias, hf);
    chain_mma<2>(q.f, x, q.bias);
    qkv_pre v;
    chain_load<2, 6>(lwb + TDBL_QKV, 4, v.f, lane);
    load_dvec<2>(lw + TDL_QKV_B + 128, v.bias, hf);
    store_dtok_split(P.qh + (size_t)tok * 64, P.ql + (size_t)tok * 64, q.bias, hf, 0.125f);
    chain_mma<2>(k.f, x, k.bias);
    store_dtok_split(P.kh + (size_t)tok * 64, P.kl + (size_t)tok * 64, k.bias, hf, 1.0f);
    chain_mma<2>(v.f, x, v.bias);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            unsigned h_, l_;
            split2(v.bias[mt][r], v.bias[mt][r + 1], h_, l_);
            const size_t f0 = (size_t)(32 * mt + NQ_DROW(r, hf)) * np + tok, f1 = f0 + np;   // rows r, r+1 are adjacent features
            P.vh[f0] = (u16)h_; P.vh[f1] = (u16)(h_ >> 16);
            P.vl[f0] = (u16)l_; P.vl[f1] = (u16)(l_ >> 16);
        }
}

// Linear 384 -> 64 + LayerNorm + layer-0 QKV, one wave per 32-token tile
__global__ __launch_bounds__(64) void td_proj_bf16_kernel(const float* __restrict__ feat, const int32_t* __restrict__ tok_off,
                                                          const int32_t* __restrict__ n_wins, int n_clips, int np,
                                                          const float* __restrict__ tw, const u16* __restrict__ twb,
                                                          float* __restrict__ x, float* __restrict__ qkv) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int tile0 = xcd_tile(blockIdx.x, gridDim.x) * 32;
    const int b = find_segment_wave(tok_off, n_clips, tile0, lane);
    const int n = n_wins[b], k0 = tile0 - tok_off[b];
    if (k0 >= n) return;
    const int tok = tile0 + j;
    const bool valid = k0 + j < n;
    const f32x4* frow = (const f32x4*)(feat + (size_t)tok * 384);
    const f32x4* af = (const f32x4*)(twb + TDB_PROJ) + lane;
    f32x16 acc[2], g0[2], b0[2];
    load_dvec<2>(tw + TD_PROJ_B, acc, h);
    load_dvec<2>(tw + TD_LN0_G, g0, h);
    load_dvec<2>(tw + TD_LN0_B, b0, h);
    // One wave per tile and nothing else on its SIMD: the 24 K-steps are a pure latency chain unless the feature rows
    // (ring of 8 steps) and the weight fragments (ring of 4 steps) are requested far ahead of their MFMAs.
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 fr[8][2], wr[4][4];
    auto load_f = [&](int s) {
        fr[s & 7][0] = valid ? frow[4 * s + 2 * h] : z4;
        fr[s & 7][1] = valid ? frow[4 * s + 2 * h + 1] : z4;
    };
    auto load_w = [&](int s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) wr[s & 3][q] = af[(s * 4 + q) * 64];       // (mt 0 hi, mt 0 lo, mt 1 hi, mt 1 lo)
    };
#pragma unroll
    for (int s = 0; s < 4; ++s) load_w(s);
#pragma unroll
    for (int s = 0; s < 8; ++s) load_f(s);
#pragma unroll
    for (int s = 0; s < 24; ++s) {
        const f32x4 f0 = fr[s & 7][0], f1 = fr[s & 7][1];
        const f32x4 ah0 = wr[s & 3][0], al0 = wr[s & 3][1], ah1 = wr[s & 3][2], al1 = wr[s & 3][3];
        f32x4 bh, bl;
        unsigned hh, ll;
        split2(f0[0], f0[1], hh, ll); bh[0] = __uint_as_float(hh); bl[0] = __uint_as_float(ll);
        split2(f0[2], f0[3], hh, ll); bh[1] = __uint_as_float(hh); bl[1] = __uint_as_float(ll);
        split2(f1[0], f1[1], hh, ll); bh[2] = __uint_as_float(hh); bl[2] = __uint_as_float(ll);
        split2(f1[2], f1[3], hh, ll); bh[3] = __uint_as_float(hh); bl[3] = __uint_as_float(ll);
        acc[0] = mfma_bf(ah0, bl, acc[0]); acc[1] = mfma_bf(ah1, bl, acc[1]);
        acc[0] = mfma_bf(al0, bh, acc[0]); acc[1] = mfma_bf(al1, bh, acc[1]);
        acc[0] = mfma_bf(ah0, bh, acc[0]); acc[1] = mfma_bf(ah1, bh, acc[1]);
        if (s + 4 < 24) load_w(s + 4);
        if (s + 8 < 24) load_f(s + 8);
    }
    qkv_pre qp;
    qkv_prefetch(tw + TD_LAYER0, twb + TDB_LAYER0, qp, lane);
    layernorm64(acc, g0, b0);
    store_dtok(x + (size_t)tok * 64, acc, h);
    qkv_store_bf(tw + TD_LAYER0, twb + TDB_LAYER0, acc, planes_of(qkv, (size_t)np * 64), tok, np, lane, qp);
}

__global__ __launch_bounds__(64) void td_layer_bf16_kernel(const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
                                                           int n_clips, int np, const float* __restrict__ lw,
                                                           const u16* __restrict__ lwb, const float* __restrict__ lw_next,
                                                           const u16* __restrict__ lwb_next, const float* x_in, float* qkv_cur,
                                                           float* x_out, float* qkv_next) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int tile0 = xcd_tile(blockIdx.x, gridDim.x) * 32;
    const int b = find_segment_wave(tok_off, n_clips, tile0, lane);
    const int n = n_wins[b], c0 = tok_off[b];
    if (tile0 - c0 >= n) return;
    const int tok = tile0 + j;
    const qkv_planes P = planes_of(qkv_cur, (size_t)np * 64);

    f32x4 qh[4], ql[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qh[s] = *(const f32x4*)(P.qh + (size_t)tok * 64 + 16 * s + 8 * h);
        ql[s] = *(const f32x4*)(P.ql + (size_t)tok * 64 + 16 * s + 8 * h);
    }
    f32x16 o[2];
    o[0] = zero16(); o[1] = zero16();
    float m = -INFINITY, l = 0.f;
    const int nkt = (n + 31) >> 5;
    f32x4 kAh[4], kAl[4], kBh[4], kBl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kAh[s] = *(const f32x4*)(P.kh + (size_t)(c0 + j) * 64 + 16 * s + 8 * h);
        kAl[s] = *(const f32x4*)(P.kl + (size_t)(c0 + j) * 64 + 16 * s + 8 * h);
    }
    auto tile = [&](int kt, const f32x4 (&kh_)[4], const f32x4 (&kl_)[4], f32x4 (&nh_)[4], f32x4 (&nl_)[4]) {
        const int key0 = c0 + 32 * kt;
        // V^T fragments: element e of lane half h <-> key 16 s + (e&3) + 8 (e>>2) + 4 h (the rows of P this half owns)
        f32x4 vh_[2][2], vl_[2][2];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const size_t base = (size_t)(j + 32 * ft) * np + key0 + 16 * s + 4 * h;
                const u32x2 a0 = *(const u32x2*)(P.vh + base), a1 = *(const u32x2*)(P.vh + base + 8);
                const u32x2 b0 = *(const u32x2*)(P.vl + base), b1 = *(const u32x2*)(P.vl + base + 8);
                vh_[ft][s] = f32x4{__uint_as_float(a0[0]), __uint_as_float(a0[1]), __uint_as_float(a1[0]), __uint_as_float(a1[1])};
                vl_[ft][s] = f32x4{__uint_as_float(b0[0]), __uint_as_float(b0[1]), __uint_as_float(b1[0]), __uint_as_float(b1[1])};
            }
        if (kt + 1 < nkt) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                nh_[s] = *(const f32x4*)(P.kh + (size_t)(key0 + 32 + j) * 64 + 16 * s + 8 * h);
                nl_[s] = *(const f32x4*)(P.kl + (size_t)(key0 + 32 + j) * 64 + 16 * s + 8 * h);
            }
        }
        __builtin_amdgcn_sched_barrier(0);     // K / V requests of the next tile stay ahead of this tile's MFMAs
        f32x16 sacc = zero16(), sacc1 = zero16(), sacc2 = zero16();    // one accumulator per product: independent chains
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            sacc1 = mfma_bf(kh_[s], ql[s], sacc1);
            sacc2 = mfma_bf(kl_[s], qh[s], sacc2);
            sacc = mfma_bf(kh_[s], qh[s], sacc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] += sacc1[r] + sacc2[r];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (32 * kt + NQ_DROW(r, h) >= n) sacc[r] = -INFINITY;
            mx = fmaxf(mx, sacc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);
        // exp(x) = 2^(x log2 e) on v_exp_f32: x <= 0 here, |x log2 e| < 150, so the argument's rounding moves the
        // result by < 2e-5 relative at the underflow edge and ~1e-7 where the weights matter
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
            f32x4 ph, pl;
            split8(sacc, 8 * s, ph, pl);
            o[0] = mfma_bf(vh_[0][s], pl, o[0]); o[1] = mfma_bf(vh_[1][s], pl, o[1]);
            o[0] = mfma_bf(vl_[0][s], ph, o[0]); o[1] = mfma_bf(vl_[1][s], ph, o[1]);
            o[0] = mfma_bf(vh_[0][s], ph, o[0]); o[1] = mfma_bf(vh_[1][s], ph, o[1]);
        }
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile(kt, kAh, kAl, kBh, kBl);
        if (kt + 1 < nkt) tile(kt + 1, kBh, kBl, kAh, kAl);
    }
    // ---- out-projection, residual + LN1, feed-forward, residual + LN2, next layer's QKV: every parameter block is
    //      requested one phase before its use
    chain_frags<2> fa, fb;
    f32x16 y[2], xr[2], g[2], bt[2], h1[2], h2[2];
    chain_load<2, 2>(lwb + TDBL_OUT, 0, fa, lane);
    load_dvec<2>(lw + TDL_OUT_B, y, h);
    load_dvec<2>(x_in + (size_t)tok * 64, xr, h);
    const float inv_l = 1.0f / l;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= inv_l; o[1][r] *= inv_l; }
    chain_mma<2>(fa, o, y);
    chain_load<2, 2>(lwb + TDBL_FF1, 0, fb, lane);
    load_dvec<2>(lw + TDL_LN1_G, g, h);
    load_dvec<2>(lw + TDL_LN1_B, bt, h);
    load_dvec<2>(lw + TDL_FF1_B, h1, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[0][r] += xr[0][r]; y[1][r] += xr[1][r]; }
    layernorm64(y, g, bt);
    chain_mma<2>(fb, y, h1);
    chain_load<2, 2>(lwb + TDBL_FF2, 0, fa, lane);
    load_dvec<2>(lw + TDL_FF2_B, h2, h);
    load_dvec<2>(lw + TDL_LN2_G, g, h);
    load_dvec<2>(lw + TDL_LN2_B, bt, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) { h1[0][r] = fmaxf(h1[0][r], 0.f); h1[1][r] = fmaxf(h1[1][r], 0.f); }
    chain_mma<2>(fa, h1, h2);
    qkv_pre qp;
    if (lw_next) qkv_prefetch(lw_next, lwb_next, qp, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[0][r] += h2[0][r]; y[1][r] += h2[1][r]; }
    layernorm64(y, g, bt);
    store_dtok(x_out + (size_t)tok * 64, y, h);
    if (lw_next) qkv_store_bf(lw_next, lwb_next, y, planes_of(qkv_next, (size_t)np * 64), tok, np, lane, qp);
}

__global__ __launch_bounds__(64) void pool_score_bf16_kernel(const float* __restrict__ x, const int32_t* __restrict__ tok_off,
                                                             const int32_t* __restrict__ n_wins, int n_clips, int n_heads,
                                                             const float* __restrict__ pw, const u16* __restrict__ pwb,
                                                             float* __restrict__ sc, float* __restrict__ yv) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int tile0 = xcd_tile(blockIdx.x, gridDim.x) * 32;
    const int b = find_segment_wave(tok_off, n_clips, tile0, lane);
    if (tile0 - tok_off[b] >= n_wins[b]) return;
    const int tok = tile0 + j;
    f32x16 xr[2];
    load_dvec<2>(x + (size_t)tok * 64, xr, h);
    {
        const int hd = blockIdx.y;                           // one wave per (tile, head): the heads are independent
        const float* w = pw + (size_t)hd * PL_FLOATS;
        f32x16 hid[4], w2[4], w3[2];
        chain_frags<4> f;
        chain_load<4, 4>(pwb + (size_t)hd * PLB_U16S, 0, f, lane);
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
}

extern "C" int nisqa_td_selfatt_bf16(const float* feat, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                                     int32_t total_tok_padded, int32_t n_layers, const float* td_w, const uint16_t* td_wb,
                                     float* ws, float* x_out, void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31) || n_layers < 1 || !td_wb) return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int np = total_tok_padded;
    const size_t sz = (size_t)np * 64;
    float* qkv[2] = {ws, ws + 3 * sz};                  // each: six bf16 planes of np*64 = 3*sz floats
    const int tiles = np / 32;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(td_proj_bf16_kernel, dim3(tiles), dim3(64), 0, st, feat, tok_off, n_wins, n_clips, np, td_w, td_wb,
                       x_out, qkv[0]);
    for (int l = 0; l < n_layers; ++l) {
        const float* lw = td_w + TD_LAYER0 + (size_t)l * TDL_FLOATS;
        const uint16_t* lwb = td_wb + TDB_LAYER0 + (size_t)l * TDBL_U16S;
        const bool more = l + 1 < n_layers;
        hipLaunchKernelGGL(td_layer_bf16_kernel, dim3(tiles), dim3(64), 0, st, tok_off, n_wins, n_clips, np, lw, lwb,
                           more ? lw + TDL_FLOATS : (const float*)nullptr, more ? lwb + TDBL_U16S : (const uint16_t*)nullptr,
                           (const float*)x_out, qkv[l & 1], x_out, qkv[(l & 1) ^ 1]);
    }
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_pool_score_bf16(const float* x, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                                     int32_t total_tok_padded, int32_t n_heads, const float* pool_w, const uint16_t* pool_wb,
                                     float* ws, void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31) || n_heads < 1 || n_heads > 8 || !pool_wb)
        return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(pool_score_bf16_kernel, dim3(total_tok_padded / 32, n_heads), dim3(64), 0, (hipStream_t)stream, x, tok_off,
                       n_wins, n_clips, n_heads, pool_w, pool_wb, ws, ws + (size_t)total_tok_padded * 8);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_pool_att_bf16(const float* x, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                                   int32_t total_tok_padded, int32_t n_heads, const float* pool_w, const uint16_t* pool_wb,
                                   float* ws, float* out, void* stream) {
    const int rc = nisqa_pool_score_bf16(x, tok_off, n_wins, n_clips, total_tok_padded, n_heads, pool_w, pool_wb, ws, stream);
    if (rc) return rc;
    return nisqa_pool_final(tok_off, n_wins, n_clips, total_tok_padded, n_heads, ws, out, stream);
}
