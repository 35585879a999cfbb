// Attention pooling (5 x PoolAttFF; reference nisqa/NISQA_lib.py:1171-1183) at fp32 OPERAND precision on the bf16 matrix pipe
// ("bf16x6"): every GEMM operand as THREE bf16 terms (hi + mid + lo: an exact split of the fp32 value) and six
// v_mfma_f32_32x32x16_bf16 products per term pair (hh, hm, mh, hl, lh, mm; fp32 accumulate) -- cnn_bf16x6.hip, DESIGN.md 4.5.
// One wave per (32-token tile, head); the self-attention kernels of the same precision are td16_bf16x6.hip.
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
