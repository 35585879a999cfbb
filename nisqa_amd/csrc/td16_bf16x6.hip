// Self-attention and attention pooling (Linear 384 -> 64 + LayerNorm, the TransformerEncoder layers, 5 x PoolAttFF; reference
// nisqa/NISQA_lib.py:988-996, 1025-1040, 1171-1183) at fp32 OPERAND precision on the bf16 matrix pipe, round-6 form: 16-token tiles on
// v_mfma_f32_16x16x32_bf16, FOUR waves = 64 tokens of ONE clip per workgroup, one workgroup per CU, n_layers + 1 launches:
//   td16_proj_kernel          projection + LayerNorm + layer-0 Q / K / V
//   td16_layer_kernel<false>  attention, out-projection, LayerNorm, feed-forward, LayerNorm, the NEXT layer's Q / K / V
//   td16_layer_kernel<true>   the last layer, then every pooling head's scores for its own tokens; the clip's last workgroup to arrive
//                             does the softmax over the clip's tokens (device-scope stores / loads, no fences)
//
// Why: the round 1-5 kernels ran one wave per 32-token tile -- 512 waves for 64 x 10 s on 1 024 SIMDs, a
// dependent chain of ~700 32x32x16 MFMAs per layer and wave, every wave streaming the layer's 144 KB of weight fragments and its
// clip's 196 KB of K / V through the vector-memory path, Q / K / V round trips through a [tokens][64] x 9 workspace, six launches.
// Here a tile is half as wide (all 1 024 SIMDs work); the weight fragments the four waves share are fetched from L2 once per
// workgroup (LDS-DMA) and read from LDS as 1 KB conflict-free ds_read_b128 fragments; what only ONE wave needs -- a key block of the
// attention (the clip's key blocks are split over the waves, each wave attends with all four query tiles and the partial results
// are merged through LDS), a weight block of the pooling (split over the waves the same way) -- goes from L2 straight into that
// wave's registers.
//
// Operands: every fp32 operand as THREE bf16 terms (hi + mid + lo, an exact split) and the six products hh hm mh hl lh mm, fp32
// accumulate, smallest products first -- the arithmetic of td_bf16x6.hip's round-5 kernels (softmax, LayerNorm, biases, residuals
// fp32 in registers).
//
// Fragment maps (wave64, lane l: c = l & 15, g = l >> 4):
//   A (16 x 32): A[row c][k-slot 8 g + e]    B (32 x 16): B[k-slot 8 g + e][col c]    D (16 x 16): D[row 4 g + r][col c], r = 0..3
// A token tile's activations live FEATURE-major in D layout: x[mt][r] = X[token c][feature 16 mt + 4 g + r] (four D tiles).  The
// contraction order of a chain GEMM is free as long as both operands agree, so k-slot (s, g, e) of K-step s means
//   feature(s, g, e) = 16 (2 s + (e >> 2)) + 4 g + (e & 3)
// -- the B fragment of step s is exactly {x[2 s][0..3], x[2 s + 1][0..3]}, no lane exchange between GEMMs; the weight fragments are
// packed in that order on the host (nisqa_amd/weights.py: linear_a_fragments_bf16_t16).  The same registers serve as the A operand
// when the roles are swapped (V is computed token-major so that its D tile IS the V^T A-fragment of the P V product).
//
// Q / K / V of a layer never exist as matrices: the producing wave stores them as the MFMA fragments the consumer needs (per layer
// buffer, bf16 units):  Q  [tile][s 2][term 3][64 lanes][8]          (B fragments of S^T = K Q^T, pre-scaled by 1/8)
//                       KV [32-key block][ K: [jt 2][s 2][term 3][64][8] | V: [ft 4][term 3][64][8] ]   (24 KB per block)
// K fragments are the producer's own D registers; a V fragment interleaves the two 16-token tiles of its block (8 bytes per lane
// each).  The consumer reads a fragment as one 16-byte load per lane (from LDS for Q, from L2 for K / V).
#include <type_traits>
#include "common.hpp"
#include "layout.hpp"
#include "../../include/nisqa_hip.h"

#define XT 3
#define LN_EPS 1e-5f
#define NQ_AS3 __attribute__((address_space(3)))
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define T16_FRAG 1024                                 /* bytes of one fragment: 64 lanes x 8 bf16 */
#define T16_GEMM (2 * 4 * XT * T16_FRAG)              /* 64 x 64 GEMM: [s 2][mt 4][term][1 KB] = 24 KB */
#define T16_KVBLK (24 * T16_FRAG)                     /* one 32-key block: K 12 fragments, V 12 fragments */
#define T16_QTILE_U16 (2 * XT * 512)                  /* Q fragments of a 16-token tile, bf16 units */
#define T16_KVBLK_U16 (24 * 512)

namespace {

NQ_DEV f32x4 mfma16(f32x4 a, f32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// LDS fragment reads are plain loads (the compiler tracks them with lgkmcnt).  What it must NOT see is the LDS-DMA staging of the layer
// kernel: hipcc orders every LDS read behind ALL outstanding LDS-DMA of the wave (s_waitcnt vmcnt(0): it cannot tell the areas
// apart), i.e. the next layer's fragments, requested early to travel under this layer's GEMMs, would have to land before those GEMMs
// read their own.  dma16<true> below issues the request from inline asm, invisible to that bookkeeping; the protocol that makes a read
// safe is explicit (the issuing wave's vmcnt + a workgroup barrier).  The compiler's own vmcnt waits for its global loads then count
// too few younger requests: they wait for more than they need, never for less.
template <int OFF>
NQ_DEV f32x4 lds_rd(unsigned a) { return *(NQ_AS3 const f32x4*)(a + OFF); }
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
// the 8 k-slots of one K-step (two D tiles of 4 registers) -> XT operand fragments
NQ_DEV void split8(const f32x4& lo, const f32x4& hi, f32x4 (&b)[XT], float scale = 1.0f) {
    unsigned t0[XT], t1[XT], t2[XT], t3[XT];
    split2t(lo[0] * scale, lo[1] * scale, t0);
    split2t(lo[2] * scale, lo[3] * scale, t1);
    split2t(hi[0] * scale, hi[1] * scale, t2);
    split2t(hi[2] * scale, hi[3] * scale, t3);
#pragma unroll
    for (int k = 0; k < XT; ++k) b[k] = f32x4{__uint_as_float(t0[k]), __uint_as_float(t1[k]), __uint_as_float(t2[k]), __uint_as_float(t3[k])};
}

struct tile16 { f32x4 v[4]; };                         // [mt][r]: feature 16 mt + 4 g + r of token c (or the swapped role, see V)

// a 64-vector (bias, LayerNorm weight) in D layout: v[mt][r] = p[16 mt + 4 g + r]
NQ_DEV void load_dvec(const float* __restrict__ p, tile16& o, int g) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) o.v[mt] = *(const f32x4*)(p + 16 * mt + 4 * g);
}
// reductions over the four lanes (g = 0..3: the four 16-lane rows) that share a token: v_permlane16_swap / v_permlane32_swap
// exchange rows inside the VALU (a __shfl_xor is a ds_bpermute: an LDS round trip in the middle of every softmax step)
NQ_DEV float swap16(float v) {                             // rows (0,1) and (2,3) exchanged against a copy: [v0 v0 v2 v2] / [v1 v1 v3 v3]
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0] ^ r[1] ^ __float_as_uint(v));   // the OTHER row's value (one of r[0], r[1] is this lane's own)
}
NQ_DEV float swap32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0] ^ r[1] ^ __float_as_uint(v));
}
NQ_DEV float sum_g(float v) { v += swap16(v); v += swap32(v); return v; }
NQ_DEV float max_g(float v) { v = fmaxf(v, swap16(v)); v = fmaxf(v, swap32(v)); return v; }

NQ_DEV void layernorm64(tile16& x, const tile16& gm, const tile16& bt) {
    float s = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += x.v[mt][r];
    const float mean = sum_g(s) * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = x.v[mt][r] - mean;
            q = fmaf(d, d, q);
        }
    const float rstd = 1.0f / sqrtf(sum_g(q) * (1.0f / 64.0f) + LN_EPS);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) x.v[mt][r] = (x.v[mt][r] - mean) * rstd * gm.v[mt][r] + bt.v[mt][r];
}

// the six term products (weight term i, activation term j), i + j <= 2, smallest first; consecutive MFMAs go to the four
// different output tiles.  SWAP: the activation fragment is the A operand (token-major result: rows = tokens, cols = features).
template <bool SWAP>
NQ_DEV void mma_terms(const f32x4 (&w)[4][XT], const f32x4 (&x)[XT], tile16& out) {
#pragma unroll
    for (int order = XT - 1; order >= 0; --order)
#pragma unroll
        for (int i = order; i >= 0; --i)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                out.v[mt] = SWAP ? mfma16(x[order - i], w[mt][i], out.v[mt]) : mfma16(w[mt][i], x[order - i], out.v[mt]);
}

// the fragments of one 64 x 64 GEMM ([s 2][mt 4][term][1 KB] in LDS) in registers: requested a phase ahead of their MFMAs
struct gemm_frags { f32x4 w[2][4 * XT]; };            // [s][mt * XT + term]
template <int I>
NQ_DEV void frags_rd12(f32x4 (&w)[12], unsigned a) {
    if constexpr (I < 12) {
        w[I] = lds_rd<I * T16_FRAG>(a);
        frags_rd12<I + 1>(w, a);
    }
}
NQ_DEV void frags_load(gemm_frags& f, unsigned wbase, unsigned lane16) {
    frags_rd12<0>(f.w[0], wbase + lane16);
    frags_rd12<0>(f.w[1], wbase + 12 * T16_FRAG + lane16);
}
// six products of one K-step from 12 fragments [mt * XT + term] (see mma_terms)
template <bool SWAP>
NQ_DEV void mma_terms12(const f32x4 (&w)[12], const f32x4 (&x)[XT], tile16& out) {
#pragma unroll
    for (int order = XT - 1; order >= 0; --order)
#pragma unroll
        for (int i = order; i >= 0; --i)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                out.v[mt] = SWAP ? mfma16(x[order - i], w[mt * XT + i], out.v[mt]) : mfma16(w[mt * XT + i], x[order - i], out.v[mt]);
}
// out += W in
template <bool SWAP = false>
NQ_DEV void chain_mma(gemm_frags& f, const tile16& in, tile16& out) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f32x4 b[XT];
        split8(in.v[2 * s], in.v[2 * s + 1], b);
        mma_terms12<SWAP>(f.w[s], b, out);
    }
}

// ---- global -> LDS without registers: global_load_lds_dwordx4 writes the wave's 64 x 16 bytes to LDS base (M0) + lane * 16, i.e.
// one 1 KB fragment per instruction.  Fragment f of a block is taken by wave f & 3; `rot` rotates the order per workgroup so that
// the 256 workgroups, which all want the SAME weight fragments at the same moment, do not walk the L2 channels in step.
// Completion is the issuing wave's vmcnt (in order) plus a workgroup barrier for the other waves' reads.
#define NQ_AS1 __attribute__((address_space(1)))
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"           /* "m0 is a reserved register": named so that the compiler reloads it after us */
template <bool ASM>
NQ_DEV void dma16(const u16* src_lane, unsigned lds_uniform) {
    if constexpr (ASM)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src_lane), "s"(lds_uniform) : "memory", "m0");
    else
        __builtin_amdgcn_global_load_lds((const NQ_AS1 void*)src_lane, (NQ_AS3 void*)lds_uniform, 16, 0, 0);
}
#pragma clang diagnostic pop
template <int N, bool ASM = false>                          // N fragments per wave, 4 N per block
NQ_DEV void dma_frags(const u16* __restrict__ src, unsigned dst, int wave, int lane, int rot) {
    int ii = rot % N;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int f = 4 * ii + wave;
        dma16<ASM>(src + (size_t)f * 512 + lane * 8, __builtin_amdgcn_readfirstlane(dst + f * T16_FRAG));
        ii = ii + 1 == N ? 0 : ii + 1;
    }
}
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define FENCE() asm volatile("" ::: "memory")

// ---- Q / K / V of the NEXT layer from this tile's activations, stored as the consumer's fragments --------------------------------
// lds_q / lds_k / lds_v: the three GEMMs' weight fragments in LDS; lw: the layer's fp32 parameter block (biases)
NQ_DEV void qkv_store(unsigned lds_q, unsigned lds_k, unsigned lds_v, const float* __restrict__ lw, const tile16& x,
                      u16* __restrict__ qbuf, u16* __restrict__ kvbuf, int tile, int lane, int c, int g, unsigned lane16) {
    gemm_frags fa, fb;
    frags_load(fa, lds_q, lane16);
    frags_load(fb, lds_k, lane16);
    tile16 q, k, v;
    load_dvec(lw + TDL_QKV_B, q, g);
    load_dvec(lw + TDL_QKV_B + 64, k, g);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {                       // V token-major: the bias is per column = per lane
        const float bv = lw[TDL_QKV_B + 128 + 16 * nt + c];
        v.v[nt] = f32x4{bv, bv, bv, bv};
    }
    chain_mma(fa, x, q);
    frags_load(fa, lds_v, lane16);
    chain_mma(fb, x, k);
    u16* qd = qbuf + (size_t)tile * T16_QTILE_U16 + lane * 8;
    u16* kd = kvbuf + (size_t)(tile >> 1) * T16_KVBLK_U16 + (tile & 1) * (2 * XT * 512) + lane * 8;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f32x4 fq[XT];
        split8(q.v[2 * s], q.v[2 * s + 1], fq, 0.125f);           // (a power of two: the split of q / 8 is the split of q)
#pragma unroll
        for (int t = 0; t < XT; ++t) *(f32x4*)(qd + (s * XT + t) * 512) = fq[t];
    }
    // V token-major: D[row = token 4 g + r][col = feature 16 nt + c]
    chain_mma<true>(fa, x, v);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f32x4 fk[XT];
        split8(k.v[2 * s], k.v[2 * s + 1], fk);
#pragma unroll
        for (int t = 0; t < XT; ++t) *(f32x4*)(kd + (s * XT + t) * 512) = fk[t];
    }
    u16* vd = kvbuf + (size_t)(tile >> 1) * T16_KVBLK_U16 + 12 * 512 + lane * 8 + 4 * (tile & 1);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        unsigned a[XT], b[XT];
        split2t(v.v[nt][0], v.v[nt][1], a);
        split2t(v.v[nt][2], v.v[nt][3], b);
#pragma unroll
        for (int t = 0; t < XT; ++t) *(u32x2*)(vd + (nt * XT + t) * 512) = u32x2{a[t], b[t]};
    }
}

// LDS plan of the projection kernel: the 384 -> 64 fragments in three 48 KB chunks (4 K-steps each); the layer-0 Q / K / V
// fragments overwrite chunks 0 and 1 once those are consumed
#define PJ_CHUNK (4 * 4 * XT * T16_FRAG)               /* 48 KB */
#define PJ_LDS (3 * PJ_CHUNK)                          /* 144 KB */

}  // namespace

// phase stamps of the two kernels (tools/td16_clock.py; empty macros unless the unit is built with -DNQ_EXPERIMENTAL)
NQ_CLK_EXPORT(g_td16_proj_clk, nisqa_debug_td16_proj_clock)
NQ_CLK_EXPORT(g_td16_layer_clk, nisqa_debug_td16_layer_clock)
NQ_CLK_EXPORT(g_td16_loop_clk, nisqa_debug_td16_loop_clock)       // sum clock inside the attention loop

// Linear 384 -> 64 + LayerNorm + layer-0 Q / K / V
__global__ __launch_bounds__(256, 1) void td16_proj_kernel(const float* __restrict__ feat, const int32_t* __restrict__ tok_off,
                                                           const int32_t* __restrict__ n_wins, int n_clips, int np,
                                                           const float* __restrict__ tw, const u16* __restrict__ twx,
                                                           float* __restrict__ x_out, u16* __restrict__ qbuf, u16* __restrict__ kvbuf,
                                                           int32_t* __restrict__ arrivals) {
    NQ_STAMP_BEGIN();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane16 = lane * 16;
    const int wg = xcd_tile(blockIdx.x, gridDim.x), rot = blockIdx.x;
    const int tok0 = wg * 64;
    const int tile = wg * 4 + wave, tok = tile * 16 + c;
    // Everything the workgroup needs is requested up front, in the order of its use: this token's feature row (K-step s needs floats
    // 32 s + 8 g .. + 7; rows of padding tokens exist, they are zeroed behind the load) and the three weight chunks (LDS-DMA).  The
    // order is pinned (vmcnt counts in order): a chunk's wait below names exactly the requests issued behind it.
    const f32x4* frow = (const f32x4*)(feat + (size_t)tok * 384) + 2 * g;
    tile16 acc, g0, b0;
    load_dvec(tw + TD_PROJ_B, acc, g);
    load_dvec(tw + TD_LN0_G, g0, g);
    load_dvec(tw + TD_LN0_B, b0, g);
    FENCE();
    f32x4 fr[12][2];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
        for (int s = 4 * ch; s < 4 * ch + 4; ++s) { fr[s][0] = frow[8 * s]; fr[s][1] = frow[8 * s + 1]; }
        FENCE();
        dma_frags<12>(twx + TDX_PROJ + (size_t)ch * (PJ_CHUNK / 2), ch * PJ_CHUNK, wave, lane, rot);
        FENCE();
    }
    // the clip of this workgroup by scalar loads (their counter is not the vector memory's: nothing above is waited for)
    const int b = find_segment(tok_off, n_clips, tok0);
    const int n = n_wins[b], c0 = tok_off[b];
    const bool active = ((tok0 - c0) >> 4) + wave < 2 * ((n + 31) >> 5);   // this tile lies in a 32-key block some query reads
    const bool valid = tok - c0 < n;                      // padding tokens enter as zero rows: everything behind stays finite
    if (arrivals && tok0 == c0 && threadIdx.x == 0) arrivals[b] = 0;      // the pooling tail of the last layer counts the clip's workgroups
    NQ_STAMP(0);                                          // requests issued, clip lookup
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        if (ch == 0) VMCNT(40); else if (ch == 1) VMCNT(20); else VMCNT(0);    // chunk ch landed (behind it: the 20 requests of every later chunk)
        __syncthreads();                                 // ... in every wave; ch = 2: chunks 0 / 1 are consumed
        NQ_STAMP(1 + 2 * ch);
        if (ch == 2) dma_frags<18>(twx + TDX_LAYER0 + TDXL_QKV, 0, wave, lane, rot);      // layer-0 Q, K, V fragments over chunks 0 / 1
        if (active) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int s = 4 * ch + s4;
                f32x4 w[12];
                frags_rd12<0>(w, ch * PJ_CHUNK + s4 * 12 * T16_FRAG + lane16);
                f32x4 bt[XT];
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                split8(valid ? fr[s][0] : z4, valid ? fr[s][1] : z4, bt);
                mma_terms12<false>(w, bt, acc);
            }
        }
        NQ_STAMP(2 + 2 * ch);
    }
    if (active) {
        layernorm64(acc, g0, b0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *(f32x4*)(x_out + (size_t)tok * 64 + 16 * mt + 4 * g) = acc.v[mt];
    }
    VMCNT(0);
    __syncthreads();
    NQ_STAMP(7);                                          // LayerNorm, x stored, Q / K / V fragments landed
    if (active) qkv_store(0, T16_GEMM, 2 * T16_GEMM, tw + TD_LAYER0, acc, qbuf, kvbuf, tile, lane, c, g, lane16);
    NQ_STAMP(8); NQ_STAMP(9); NQ_STAMP(10); NQ_STAMP(11);
    NQ_STAMP_END(g_td16_proj_clk, blockIdx.x * 4 + wave);
}

// LDS plan of the layer kernel: a 72 KB work area (the four tiles' Q fragments during the attention loop, the waves' partial results
// behind it, then the next layer's Q / K / V fragments or the pooling tail's exchange), then the layer's own three GEMMs
#define LY_RING 0u
#define LY_WOUT (LY_RING + 3 * T16_KVBLK)               /* out-projection, feed-forward 1, feed-forward 2 */
#define LY_WFF1 (LY_WOUT + T16_GEMM)
#define LY_WFF2 (LY_WFF1 + T16_GEMM)
#define LY_LDS (LY_WFF2 + T16_GEMM)                     /* 144 KB */
#define LY_QT LY_RING                                   /* during the attention loop: the four tiles' Q fragments [tile][s 2][term][1 KB] = 24 KB */
#define LY_PO LY_RING                                   /* behind it: the waves' partial outputs [wave][tile][ft 4][1 KB] = 64 KB ... */
#define LY_PM (LY_RING + 64 * T16_FRAG)                 /* ... and (max, sum) pairs [wave][tile][64 lanes] = 8 KB */
// Pooling tail of the LAST layer (5 x PoolAttFF, NISQA_lib.py:1171-1183): a head's 64 -> 128 linear is two 64-row blocks of the same
// fragment form, read from L2 straight into the registers of the ONE wave that multiplies them; LDS holds the fp32 vectors of every
// block (b1 | w2 | w3 | b2, b3: 1 KB each, behind the layer's own areas), the four tiles' activation terms and the partial scores
#define PL16_PAR LY_LDS
#define PL16_BLK_U16 (24 * 512)
#define PL16_XT LY_RING                                 /* the four tiles' activation terms: [tile 4][s 2][term][1 KB] = 24 KB */
#define PL16_SP (LY_RING + 4 * 2 * XT * T16_FRAG)       /* partial scores [block][64 tokens] */
struct pool_args {
    int n_heads;
    const u16* wx;                                      // [2 n_heads][24 fragments] then [2 n_heads][256 floats]
    float* sc;                                          // [np][8] scores, then [np][8] values
    int32_t* arrivals;                                  // per clip: workgroups done (zeroed by td16_proj_kernel)
    float* out;                                         // [n_clips][n_heads]
};

// one encoder layer for 64 tokens: attention over the clip's K / V blocks (LDS ring filled by LDS-DMA), out-projection, residual +
// LayerNorm, feed-forward, residual + LayerNorm, and the next layer's Q / K / V
template <bool POOL>
__global__ __launch_bounds__(256, 1) void td16_layer_kernel(const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
                                                            int n_clips, int np, const float* __restrict__ lw,
                                                            const u16* __restrict__ lwx, const float* __restrict__ lw_next,
                                                            const u16* __restrict__ lwx_next, const float* x_in,
                                                            const u16* __restrict__ qbuf, const u16* __restrict__ kvbuf,
                                                            float* x_out, u16* __restrict__ qbuf_next, u16* __restrict__ kvbuf_next,
                                                            pool_args pl) {
    NQ_STAMP_BEGIN();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane16 = lane * 16;
    const int wg = xcd_tile(blockIdx.x, gridDim.x), rot = blockIdx.x;
    const int tok0 = wg * 64;
    const int tile = wg * 4 + wave, tok = tile * 16 + c;
    // the clip of this workgroup FIRST, by one vector load and a ballot per 64 clips (a binary search by scalar loads is a chain of
    // log2(n_clips) trips to L2 at the head of everything: 2.5 k cycles of an 11 us kernel)
    const int b = find_segment_wave(tok_off, n_clips, tok0, lane);
    const int n = n_wins[b], c0 = tok_off[b];
    tile16 xr;                                            // this tile's residual rows
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xr.v[mt] = *(const f32x4*)(x_in + (size_t)tok * 64 + 16 * mt + 4 * g);
    FENCE();
    const int nkb = (n + 31) >> 5;                        // 32-key blocks of this clip
    const bool active = ((tok0 - c0) >> 4) + wave < 2 * nkb;
    const int nact = min(4, 2 * nkb - ((tok0 - c0) >> 4));             // active tiles of this workgroup (wave-uniform)
    const u16* kvg = kvbuf + (size_t)(c0 >> 5) * T16_KVBLK_U16;
    // The clip's key blocks are SPLIT OVER THE FOUR WAVES (wave w takes blocks w, w + 4, ...), every wave attends with all four query
    // tiles of the workgroup: a K / V fragment is then needed by one wave only and comes from L2 straight into its registers (no ring,
    // no barrier per block), the four tiles' Q fragments sit in LDS (24 KB, read per use), and the waves' partial (max, sum, output)
    // triples are merged through LDS behind the loop.  With one tile per wave and every wave reading every block from LDS a block
    // cost 2.8 k cycles for 0.77 k of MFMA work, nothing in a wave overlapping with anything.
    dma_frags<6, true>(qbuf + (size_t)wg * 4 * T16_QTILE_U16, LY_QT, wave, lane, 0);
    f32x4 kf[12], vf[12];                                  // K: [(jt * 2 + s) * XT + term], V: [ft * XT + term]
    auto kv_global = [&](f32x4 (&w)[12], int kb, int half) {
        const f32x4* src = (const f32x4*)(kvg + (size_t)kb * T16_KVBLK_U16) + lane + half * 12 * 64;
#pragma unroll
        for (int q = 0; q < 12; ++q) w[q] = src[q * 64];
    };
    if (wave < nkb) { kv_global(kf, wave, 0); kv_global(vf, wave, 1); }
    FENCE();
    // the layer's own weights (out-projection, feed-forward 1 and 2) behind the first block: they are not needed before the loop ends
    for (int wb = 0; wb < 3; ++wb) dma_frags<6, true>(lwx + TDXL_OUT + (size_t)wb * (T16_GEMM / 2), LY_WOUT + wb * T16_GEMM, wave, lane, rot);
    VMCNT(18);                                            // everything older than the 18 weight requests: Q fragments, first K / V block, xr
    __syncthreads();                                     // Q fragments of every wave's share
    NQ_STAMP(0);                                          // clip lookup, Q and the first K / V block landed

    tile16 o4[4];
    float m4[4], l4[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
        m4[t4] = -INFINITY;
        l4[t4] = 0.f;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) o4[t4].v[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float L2E = 1.44269504088896341f;
    NQ_SUM_BEGIN();
    // FULL: all four tiles active (every workgroup but a clip's last) -- straight-line code, the four tiles' chains interleave
    auto attend = [&](auto full) {
    constexpr bool FULL = decltype(full)::value;
    for (int kb = wave; kb < nkb; kb += 4) {
        // S^T = K Q^T of the two 16-key sub-tiles for every active query tile: one accumulator per product ORDER (independent chains),
        // added smallest first
        f32x4 st[4][2];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            if (FULL || t4 < nact) {
                f32x4 q[2][XT];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int t = 0; t < XT; ++t) q[s][t] = *(NQ_AS3 const f32x4*)(LY_QT + ((t4 * 2 + s) * XT + t) * T16_FRAG + lane16);
                f32x4 sa[2][XT];
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int k = 0; k < XT; ++k) sa[jt][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int order = XT - 1; order >= 0; --order)
#pragma unroll
                        for (int i = order; i >= 0; --i)
#pragma unroll
                            for (int jt = 0; jt < 2; ++jt) sa[jt][order] = mfma16(kf[(jt * 2 + s) * XT + i], q[s][order - i], sa[jt][order]);
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[t4][jt][r] = (sa[jt][2][r] + sa[jt][1][r]) + sa[jt][0][r];
            }
        }
        if (kb + 4 < nkb) kv_global(kf, kb + 4, 0);       // (the K fragments are consumed: the next block's take their place)
        NQ_SUM(1);                                       // S of the four tiles
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            if (FULL || t4 < nact) {
                // online softmax over the keys (rows: key 32 kb + 16 jt + 4 g + r; a query's keys sit in the four lanes g = 0..3)
                float mx = -INFINITY;
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (32 * kb + 16 * jt + 4 * g + r >= n) st[t4][jt][r] = -INFINITY;
                        mx = fmaxf(mx, st[t4][jt][r]);
                    }
                mx = max_g(mx);
                const float m_new = fmaxf(m4[t4], mx);
                const float alpha = __builtin_amdgcn_exp2f((m4[t4] - m_new) * L2E);
                float rs = 0.f;
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        st[t4][jt][r] = __builtin_amdgcn_exp2f((st[t4][jt][r] - m_new) * L2E);
                        rs += st[t4][jt][r];
                    }
                l4[t4] = l4[t4] * alpha + rs;            // (per-lane partial sums: alpha is the same in the four lanes of a query)
                m4[t4] = m_new;
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) o4[t4].v[ft] *= alpha;
                f32x4 p[XT];
                split8(st[t4][0], st[t4][1], p);
                mma_terms12<false>(vf, p, o4[t4]);
            }
        }
        if (kb + 4 < nkb) kv_global(vf, kb + 4, 1);
        NQ_SUM(3);                                       // softmax, rescale, split, P V of the four tiles
        NQ_SUM_COUNT(8, 1);
    }
    };
    if (nact == 4) attend(std::true_type{}); else attend(std::false_type{});
    NQ_SUM_END(g_td16_loop_clk, blockIdx.x * 4 + wave, lane == 0);
    // merge: every wave leaves its (max, sum, output) of every tile in LDS, the wave that owns a tile combines the four
    __syncthreads();                                     // (every wave is done with the Q fragments: the partials overlay them)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
        if (t4 < nact) {
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) *(NQ_AS3 f32x4*)(LY_PO + ((wave * 4 + t4) * 4 + ft) * T16_FRAG + lane16) = o4[t4].v[ft];
            *(NQ_AS3 f32x2_t*)(LY_PM + (wave * 4 + t4) * 512 + lane * 8) = f32x2_t{m4[t4], l4[t4]};
        }
    }
    __syncthreads();
    tile16 o;
    float l = 0.f;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) o.v[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
        f32x2_t ml[4];
        float mm = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            ml[w] = *(NQ_AS3 const f32x2_t*)(LY_PM + (w * 4 + wave) * 512 + lane * 8);
            mm = fmaxf(mm, ml[w][0]);
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = __builtin_amdgcn_exp2f((ml[w][0] - mm) * L2E);            // a wave without key blocks left max = -inf: weight 0
            l = fmaf(ml[w][1], a, l);
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const f32x4 po = *(NQ_AS3 const f32x4*)(LY_PO + ((w * 4 + wave) * 4 + ft) * T16_FRAG + lane16);
                o.v[ft] += po * a;
            }
        }
    }
    __syncthreads();                                     // every wave has read the partials: the area is free for the next fragments
    NQ_STAMP(1);                                          // attention over the clip's key blocks
    const int nblk = POOL ? 2 * pl.n_heads : 0;
    if constexpr (POOL) {                                 // the blocks' fp32 vectors: requested first, so that the wait below covers them
        for (int f = wave; f < nblk; f += 4) dma16<true>(pl.wx + (size_t)nblk * PL16_BLK_U16 + (size_t)f * 512 + lane * 8, PL16_PAR + f * T16_FRAG);
    }
    FENCE();
    // the next layer's Q / K / V fragments go where the partials were (every wave is behind the merge's last barrier)
    if (POOL) {
        VMCNT(0);
    } else if (lw_next) { dma_frags<18, true>(lwx_next + TDXL_QKV, LY_RING, wave, lane, rot); VMCNT(18); } else { VMCNT(0); }
    __syncthreads();                                     // out / ff1 / ff2 fragments landed in every wave
    NQ_STAMP(2);
    if (active) {
        gemm_frags fa, fb;
        frags_load(fa, LY_WOUT, lane16);
        tile16 y, gm, bt, h1, h2;
        load_dvec(lw + TDL_OUT_B, y, g);
        load_dvec(lw + TDL_LN1_G, gm, g);
        load_dvec(lw + TDL_LN1_B, bt, g);
        load_dvec(lw + TDL_FF1_B, h1, g);
        const float inv_l = 1.0f / sum_g(l);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) o.v[ft] *= inv_l;
        frags_load(fb, LY_WFF1, lane16);
        chain_mma(fa, o, y);
        frags_load(fa, LY_WFF2, lane16);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) y.v[mt] += xr.v[mt];
        layernorm64(y, gm, bt);
        NQ_STAMP(3);                                      // out-projection, residual, LayerNorm
        load_dvec(lw + TDL_FF2_B, h2, g);
        load_dvec(lw + TDL_LN2_G, gm, g);
        load_dvec(lw + TDL_LN2_B, bt, g);
        chain_mma(fb, y, h1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h1.v[mt][r] = fmaxf(h1.v[mt][r], 0.f);
        chain_mma(fa, h1, h2);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) y.v[mt] += h2.v[mt];
        layernorm64(y, gm, bt);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *(f32x4*)(x_out + (size_t)tok * 64 + 16 * mt + 4 * g) = y.v[mt];
        o = y;
    } else {
        NQ_STAMP(3);
    }
    NQ_STAMP(4);                                          // feed-forward, residual, LayerNorm, x stored
    if constexpr (POOL) {
        // The 2 n_heads blocks are split over the four WAVES, every wave multiplies its blocks with all four token tiles of the
        // workgroup: a weight fragment is then needed by ONE wave, so it comes from L2 straight into registers, and LDS carries only
        // the tiles' activation terms (24 KB read once per wave).  With the blocks staged in LDS and every wave reading all of them
        // (240 KB per wave) the LDS port was the limit: 26 k cycles for 7.7 k of MFMA work.
        if (active) {                                    // (the ring is free since the loop's last barrier; the blocks' fp32 vectors landed with the weights)
            f32x4 xb[2][XT];
            split8(o.v[0], o.v[1], xb[0]);
            split8(o.v[2], o.v[3], xb[1]);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < XT; ++t) *(NQ_AS3 f32x4*)(PL16_XT + ((wave * 2 + s) * XT + t) * T16_FRAG + lane16) = xb[s][t];
            // linear3 of every head on this tile's tokens (the "value" the pooled softmax weights)
            for (int hd = 0; hd < pl.n_heads; ++hd) {
                const unsigned par = PL16_PAR + 2 * hd * T16_FRAG;
                float vpart = 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const f32x4 w3 = *(NQ_AS3 const f32x4*)(par + 512 + 64 * mt + 16 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) vpart = fmaf(w3[r], o.v[mt][r], vpart);
                }
                const float vv = sum_g(vpart) + *(NQ_AS3 const float*)(par + 772);
                if (g == 0) __hip_atomic_store(pl.sc + (size_t)np * 8 + (size_t)tok * 8 + hd, vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();                                 // the tiles' terms are in LDS
        NQ_STAMP(5);                                      // this tile's terms and values
        if (nact > 0) {
            // one fragment buffer: a K-step's 12 fragments are re-requested for the wave's NEXT block as soon as the four tiles have
            // consumed them (half a block of MFMA work ahead of their use); the tiles' terms are read from LDS where they are used
            gemm_frags f;
            auto frags_global = [&](f32x4 (&w)[12], int blk, int s) {
                const f32x4* src = (const f32x4*)(pl.wx + (size_t)blk * PL16_BLK_U16) + lane + s * 12 * 64;
#pragma unroll
                for (int q = 0; q < 12; ++q) w[q] = src[q * 64];
            };
            if (wave < nblk) { frags_global(f.w[0], wave, 0); frags_global(f.w[1], wave, 1); }
            // FULL: all four tiles active (every workgroup but a clip's last) -- straight-line code, the terms of a K-step's four
            // tiles are requested together ahead of their products
            auto blocks = [&](auto full) {
                constexpr bool FULL = decltype(full)::value;
                for (int blk = wave; blk < nblk; blk += 4) {
                    const unsigned par = PL16_PAR + blk * T16_FRAG + 16 * g;
                    tile16 h[4], w2;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const f32x4 b1 = *(NQ_AS3 const f32x4*)(par + 64 * mt);
                        w2.v[mt] = *(NQ_AS3 const f32x4*)(par + 256 + 64 * mt);
#pragma unroll
                        for (int t4 = 0; t4 < 4; ++t4) h[t4].v[mt] = b1;
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        f32x4 xt[4][XT];
#pragma unroll
                        for (int t4 = 0; t4 < 4; ++t4)
                            if (FULL || t4 < nact)
#pragma unroll
                                for (int t = 0; t < XT; ++t) xt[t4][t] = *(NQ_AS3 const f32x4*)(PL16_XT + ((t4 * 2 + s) * XT + t) * T16_FRAG + lane16);
#pragma unroll
                        for (int t4 = 0; t4 < 4; ++t4)
                            if (FULL || t4 < nact) mma_terms12<false>(f.w[s], xt[t4], h[t4]);
                        if (blk + 4 < nblk) frags_global(f.w[s], blk + 4, s);
                    }
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        if (FULL || t4 < nact) {
                            float spart = 0.f;
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                                for (int r = 0; r < 4; ++r) spart = fmaf(w2.v[mt][r], fmaxf(h[t4].v[mt][r], 0.f), spart);
                            spart = sum_g(spart);
                            if (g == 0) *(NQ_AS3 float*)(PL16_SP + (blk * 64 + 16 * t4 + c) * 4) = spart;    // this half's share of the score
                        }
                    }
                }
            };
            if (nact == 4) blocks(std::true_type{}); else blocks(std::false_type{});
        }
        NQ_STAMP(6);                                      // this wave's blocks x four tiles
        __syncthreads();                                 // both halves of every head's hidden layer are in
        for (int i = threadIdx.x; i < 64 * pl.n_heads; i += 256) {
            const int hd = i >> 6, t = i & 63;
            if ((t >> 4) < nact) {
                const float sv = *(NQ_AS3 const float*)(PL16_SP + ((2 * hd) * 64 + t) * 4) + *(NQ_AS3 const float*)(PL16_SP + ((2 * hd + 1) * 64 + t) * 4) +
                                 *(NQ_AS3 const float*)(PL16_PAR + 2 * hd * T16_FRAG + 768);
                __hip_atomic_store(pl.sc + (size_t)(tok0 + t) * 8 + hd, sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // device scope: the reader may sit on another XCD
            }
        }
        NQ_STAMP(7);                                      // barrier, scores stored
        // The clip's last workgroup to arrive pools over the clip's tokens (softmax of the scores, weighted sum of the values).  No
        // fences (a device-scope release / acquire writes back and invalidates the XCD's whole L2: measured 26 us per launch):
        // scores and values travel as device-scope stores and loads, ordered by "stores complete -> count -> loads".
        VMCNT(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int nwg = (tok_off[b + 1] - c0) >> 6;
            *(NQ_AS3 int*)(0u) = __hip_atomic_fetch_add(pl.arrivals + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1;
        }
        __syncthreads();
        if (*(NQ_AS3 const int*)(0u)) {
            // every wave takes a quarter of the tokens and ALL heads: one round trip to memory for the 2 n_heads loads of a token
            // (a loop over heads with a loop over tokens inside is a chain of dependent trips: measured 15 us), an online softmax
            // per lane, then the lanes and the four waves merge their (max, denominator, numerator) triples
            float m[8], d[8], u[8];
#pragma unroll
            for (int hd = 0; hd < 8; ++hd) { m[hd] = -INFINITY; d[hd] = 0.f; u[hd] = 0.f; }
            for (int t = 64 * wave + lane; t < n; t += 256) {
                const float* sc = pl.sc + (size_t)(c0 + t) * 8;
                float sv[8], vv[8];
#pragma unroll
                for (int hd = 0; hd < 8; ++hd)
                    if (hd < pl.n_heads) {
                        sv[hd] = __hip_atomic_load(sc + hd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        vv[hd] = __hip_atomic_load(sc + (size_t)np * 8 + hd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                for (int hd = 0; hd < 8; ++hd)
                    if (hd < pl.n_heads) {
                        const float mn = fmaxf(m[hd], sv[hd]);
                        const float a = expf(m[hd] - mn), e = expf(sv[hd] - mn);
                        d[hd] = fmaf(d[hd], a, e);
                        u[hd] = fmaf(u[hd], a, e * vv[hd]);
                        m[hd] = mn;
                    }
            }
            __syncthreads();                             // (the flag at LDS 0 has been read by every wave)
#pragma unroll
            for (int hd = 0; hd < 8; ++hd)
                if (hd < pl.n_heads) {
                    const float mw = wave_max(m[hd]);
                    const float a = m[hd] == -INFINITY ? 0.f : expf(m[hd] - mw);
                    const float dw = wave_sum(d[hd] * a), uw = wave_sum(u[hd] * a);
                    if (lane == 0) *(NQ_AS3 f32x4*)(16u * (wave * 8 + hd)) = f32x4{mw, dw, uw, 0.f};
                }
            __syncthreads();
            if (threadIdx.x < pl.n_heads) {
                float mm = -INFINITY;
                f32x4 p[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) { p[w] = *(NQ_AS3 const f32x4*)(16u * (w * 8 + threadIdx.x)); mm = fmaxf(mm, p[w][0]); }
                float den = 0.f, num = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float a = p[w][0] == -INFINITY ? 0.f : expf(p[w][0] - mm);
                    den = fmaf(p[w][1], a, den);
                    num = fmaf(p[w][2], a, num);
                }
                pl.out[(size_t)b * pl.n_heads + threadIdx.x] = num / den;
            }
        }
    } else if (lw_next) {
        VMCNT(0);
        __syncthreads();
        NQ_STAMP(5);                                      // next layer's Q / K / V fragments landed
        if (active) qkv_store(LY_RING, LY_RING + T16_GEMM, LY_RING + 2 * T16_GEMM, lw_next, o, qbuf_next, kvbuf_next, tile, lane, c, g, lane16);
    } else {
        NQ_STAMP(5);
    }
    if constexpr (!POOL) { NQ_STAMP(6); NQ_STAMP(7); }
    NQ_STAMP(8); NQ_STAMP(9); NQ_STAMP(10); NQ_STAMP(11);
    NQ_STAMP_END(g_td16_layer_clk, (lw_next ? 0 : 16384) + blockIdx.x * 4 + wave);
}

// ws: 9 * np * 64 floats = two layer buffers of np * 576 bf16 each (Q: np * 192, K / V blocks: np * 384)
// pool_wx16 != null: the last layer's launch carries the attention pooling as well (ws_pool: np * 16 floats + n_clips counters)
static int td16_launch(const float* feat, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips, int32_t total_tok_padded,
                       int32_t n_layers, const float* td_w, const uint16_t* td_wx, int32_t n_heads, const uint16_t* pool_wx16,
                       float* ws, float* x_out, float* ws_pool, float* out, void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 63) || n_layers < 1 || !td_wx) return NISQA_ERR_ARG;
    if (pool_wx16 && (n_heads < 1 || n_heads > 8 || !ws_pool || !out)) return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int np = total_tok_padded;
    u16* lbuf[2] = {(u16*)ws, (u16*)ws + (size_t)np * 576};
    const int wgs = np / 64;
    static std::atomic<bool> ok_p[64], ok_l[64], ok_lp[64];
    if (nq_lds_opt_in((const void*)td16_proj_kernel, PJ_LDS, ok_p) || nq_lds_opt_in((const void*)td16_layer_kernel<false>, LY_LDS, ok_l) ||
        nq_lds_opt_in((const void*)td16_layer_kernel<true>, LY_LDS + 16 * T16_FRAG, ok_lp))
        return NISQA_ERR_LAUNCH;
    pool_args pl = {};
    if (pool_wx16) pl = pool_args{n_heads, pool_wx16, ws_pool, (int32_t*)(ws_pool + (size_t)np * 16), out};
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(td16_proj_kernel, dim3(wgs), dim3(256), PJ_LDS, st, feat, tok_off, n_wins, n_clips, np, td_w, td_wx, x_out,
                       lbuf[0], lbuf[0] + (size_t)np * 192, pl.arrivals);
    for (int l = 0; l < n_layers; ++l) {
        const float* lw = td_w + TD_LAYER0 + (size_t)l * TDL_FLOATS;
        const uint16_t* lwx = td_wx + TDX_LAYER0 + (size_t)l * TDXL_U16S;
        const bool more = l + 1 < n_layers;
        u16 *cur = lbuf[l & 1], *nxt = lbuf[(l & 1) ^ 1];
        if (!more && pool_wx16)
            hipLaunchKernelGGL(td16_layer_kernel<true>, dim3(wgs), dim3(256), LY_LDS + 2 * n_heads * T16_FRAG, st, tok_off, n_wins, n_clips,
                               np, lw, lwx, (const float*)nullptr, (const uint16_t*)nullptr, (const float*)x_out, (const u16*)cur,
                               (const u16*)(cur + (size_t)np * 192), x_out, nxt, nxt + (size_t)np * 192, pl);
        else
            hipLaunchKernelGGL(td16_layer_kernel<false>, dim3(wgs), dim3(256), LY_LDS, st, tok_off, n_wins, n_clips, np, lw, lwx,
                               more ? lw + TDL_FLOATS : (const float*)nullptr, more ? lwx + TDXL_U16S : (const uint16_t*)nullptr,
                               (const float*)x_out, (const u16*)cur, (const u16*)(cur + (size_t)np * 192), x_out, nxt,
                               nxt + (size_t)np * 192, pl);
    }
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_td_selfatt_bf16x6(const float* feat, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                                       int32_t total_tok_padded, int32_t n_layers, const float* td_w, const uint16_t* td_wx,
                                       float* ws, float* x_out, void* stream) {
    return td16_launch(feat, tok_off, n_wins, n_clips, total_tok_padded, n_layers, td_w, td_wx, 0, nullptr, ws, x_out, nullptr, nullptr,
                       stream);
}

// self-attention and attention pooling of a batch in n_layers + 1 launches: pool_wx = the three-term pooling blob, whose 16-token
// part (nisqa_amd/weights.py: pack_pool_att_t16) lies behind the n_heads blocks of nisqa_pool_att_bf16x6
extern "C" int nisqa_td_pool_bf16x6(const float* feat, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                                    int32_t total_tok_padded, int32_t n_layers, const float* td_w, const uint16_t* td_wx,
                                    int32_t n_heads, const uint16_t* pool_wx, float* ws, float* x_out, float* ws_pool, float* out,
                                    void* stream) {
    if (!pool_wx || n_heads < 1 || n_heads > 8) return NISQA_ERR_ARG;
    return td16_launch(feat, tok_off, n_wins, n_clips, total_tok_padded, n_layers, td_w, td_wx, n_heads,
                       pool_wx + (size_t)n_heads * PLX_U16S, ws, x_out, ws_pool, out, stream);
}
