// The whole AdaptCNN (conv1..conv6, BatchNorm folded, adaptive max-pools) on split-bf16 MFMA ("bf16x3") --
// same role, inputs and outputs as cnn_front_kernel + cnn_back_kernel in cnn.hip (reference
// nisqa/NISQA_lib.py:2239-2282, 487-502, 688-710).
//
// Every fp32 operand x is carried as x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa
// bits) and each product is formed as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation: 3 MFMAs at 16x the fp32-MFMA rate = 5.3x the exact-fp32 throughput.  Dropped term:
// lo*lo ~ 2^-18.  conv1 sees dB values up to |80|: two terms resolve them to 6e-4 dB, the size of the mel
// stage's own deviation from the oracle.  Measured effect on the outputs with the real nisqa.tar
// weights: |dMOS| <= 5e-5 (pure bf16: 4.5e-3, pure f16: 5e-4; bar 1e-3) -- DESIGN.md 4.5.
//
// Structure:
//   * a workgroup is FOUR waves = four consecutive segments of one clip; conv1..conv4 are wave-private and barrier-free:
//     each wave streams its weight fragments from L2 into a 3-deep register ring (conv_k_bf16) and keeps its
//     activations in its own LDS region as two bf16 planes (hi, lo), pixel-major with pixel rows padded by 16 bytes, with
//     row -> pixel maps chosen so that every adaptive-max-pool window is in-lane;
//   * conv1 runs on the matrix pipe too, two mel-adjacent output pixels per MFMA row (operands are dword reads from two
//     zero-bordered bf16 planes of the input patch, no packing);
//   * conv5/conv6 (18 / 6 output pixels per segment) are batched over the workgroup's four segments with the output
//     channels split over the waves (16x16x32 MFMA tiles), so no tile is mostly padding.
// What bounds it (tools/micro/issue2.hip): a wave hides <= 5 other instructions behind a 32x32x16 MFMA
// (<= 2 behind a 16x16x32) and a VALU-only stream issues one instruction per ~5 cycles, so every instruction outside the
// MFMA shadows counts.  Hence: LDS by 32-bit addresses (no 64-bit pointer arithmetic), lane-static tap masks instead of
// per-tap bounds arithmetic, weight fragments through a buffer descriptor (no per-load address VALU), epilogue addresses
// as lane base + immediates, and the conv1 epilogue finalises each pooled value in ONE lane of the mel pair, not both.
#include "common.hpp"
#include "layout.hpp"
#include <stdlib.h>
#include <atomic>
#include "conv_bf16.hpp"
#include "internal.hpp"
#include "../../include/nisqa_hip.h"
// weight-fragment ring depths of the conv2 / conv3-4 K loops (slots; RING - 1 K-steps are in flight)
#define NQ_RING2 3
#define NQ_RING34 3

// LDS plan (byte addresses; the kernel has no static LDS, so the dynamic segment starts at 0 and addresses are used as
// plain 32-bit numbers): one 128-byte zero block shared by the four waves ABOVE the largest tap offset (so "zero block
// address - tap offset" never goes negative), then one region per wave.
//   activation planes: pixel rows of C bf16 padded by 16 bytes (row stride 2 C + 16, NOT swizzled): consecutive pixels
//   land 4 banks apart, and every LDS address is lane base + compile-time offset.
#define FB_RS1 48                          /* A1: 168 px x 16 ch */
#define FB_P1 (168 * FB_RS1)
#define FB_RS2 80                          /* A2: 60 px x 32 ch */
#define FB_P2 (60 * FB_RS2)
#define FB_RS3 144                         /* A3: 60 px x 64 ch; S4 / S5: 72 rows x 64 ch */
#define FB_P3 (60 * FB_RS3)
#define FB_PS (72 * FB_RS3)
#define FB_PATCH (2 * FB_P1)               /* conv1 input: two zero-bordered bf16 planes [17][50] behind the A1 planes */
#define FB_PPLANE 1700                     /* bytes per patch plane (850 bf16) */
#define FB_ZADDR 2048u                     /* the shared zero block */
#define FB_BASE 2176u                      /* first wave region */
#define FB_WAVE 19584u
#define FB_WGS 2
#define FB_LDS (FB_BASE + 4 * FB_WAVE)     /* 80512 B -> two workgroups (8 waves) per CU */
static_assert(FB_PATCH + 2 * FB_PPLANE <= FB_WAVE && 2 * FB_P3 <= FB_WAVE && 2 * FB_PS <= 2 * FB_WAVE, "LDS plan");
static_assert(FB_WGS * FB_LDS <= 160 * 1024, "workgroups per CU");

__device__ constexpr int bwin75_lo(int b) { return b == 0 ? 0 : b == 1 ? 1 : b == 2 ? 2 : b == 3 ? 4 : 5; }
__device__ constexpr int bwin75_hi(int b) { return b == 0 ? 2 : b == 1 ? 3 : b == 2 ? 5 : b == 3 ? 6 : 7; }
__device__ constexpr int bwin53_lo(int b) { return b == 0 ? 0 : b == 1 ? 1 : 3; }
__device__ constexpr int bwin53_hi(int b) { return b == 0 ? 2 : b == 1 ? 4 : 5; }

// layer-boundary stamps of the phase clock (tools/phase_clock.py; empty macros unless the unit is built with -DNQ_EXPERIMENTAL)
NQ_CLK_EXPORT(g_phase_clk, nisqa_debug_phase_clock)

// the per-row scale tables of the f16 formats' conv5 / conv6 epilogues (the LDS below the zero block is otherwise unused)
#define FB_TAB5 0u                         /* [72 rows] {2^(e5 - e4 - kw5), 2^e5} of the row's segment */
#define FB_TAB6 1024u                      /* [24 rows] 2^-(e5 + kw6) */
NQ_DEV f32x2_t lds_ld64(unsigned a) { return *(NQ_AS3 const f32x2_t*)(a); }
// bias + ReLU of an epilogue value; the f16 formats fold the power-of-two scales in: relu(v * c + t) with c = 2^(e_out - e_in - kw),
// t = shift * 2^e_out -- exact scalings, so the result is 2^e_out times what the unscaled arithmetic rounds to
template <int FMT>
NQ_DEV float epi_fmt(float v, float c, float t) { return FMT == NQ_FMT_BF16X3 ? fmaxf(v + t, 0.f) : fmaxf(fmaf(v, c, t), 0.f); }

// FMT: operand format (conv_bf16.hpp): bf16 hi + lo / three products, or f16 hi + lo of the power-of-two-scaled tensors with three or
//      all four products.  For the f16 formats wb is the CNNH_ blob (fragments of W * 2^kw + per-layer constants) and every activation
//      tensor is stored as y * 2^e with e = 15 - ceil(log2(m_in * G + T)): m_in the MEASURED maximum of the layer's input for this
//      segment, G = max_c sum |W_c| and T = max |shift| of the layer -- |y| <= m_in * G + T, so the scaled tensor stays below 2^15 for
//      any finite input and any weights (no calibration, no clamping), 3-5 bits below it for the shipped weights.
// SEGX: the input is the reference's segment tensor x[B][L][1][48][15] (inner-operator mode) instead of the spectrogram
// P3: also write the pooled conv4 output as fp32 (debug / parity callers of nisqa_cnn_adapt_bf16 that pass p3_opt)
template <int FMT, bool SEGX, bool P3>
NQ_DEV void cnn_front_split_body(
    const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off,
    const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
    const float* __restrict__ clip_floor, int n_clips, int seg_hop,
    const float* __restrict__ cw, const unsigned short* __restrict__ wb, float* __restrict__ p3,
    float* __restrict__ feat, const float* __restrict__ seg_x, int seg_L,
    const uint32_t* __restrict__ clip_max_enc, float top_db) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool F16 = FMT != NQ_FMT_BF16X3;
    static_assert(!(F16 && P3), "the fp32 copy of the pooled conv4 tensor is a bf16x3 debug output");
    // F16: per-layer constants behind the fragments (layout.hpp CNNH_META): kw[l], G[l], T[l] for l = 1..6 at index l - 1
    const int* __restrict__ meta_i = (const int*)(wb + CNNH_META);
    const float* __restrict__ meta_f = (const float*)(wb + CNNH_META);
    float dummy_mx = 0.f;
    NQ_STAMP_BEGIN();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p0 = blockIdx.x * 4;                      // tok_off is a multiple of 32: no clip straddling (an XCD-grouped order measured 1.7 % slower)
    const int b = __builtin_amdgcn_readfirstlane(find_segment_wave(tok_off, n_clips, p0, lane));   // one vector load + ballot, not a chain of log2(n) scalar loads
    const int k0 = p0 - tok_off[b];
    const int nvalid = min(4, n_wins[b] - k0);
    if (nvalid <= 0) return;                             // whole workgroup is padding
    const bool valid = wave < nvalid;                    // padding waves still walk the barriers (on zeros)
    const int p = p0 + wave, k = k0 + wave;
    const unsigned R = FB_BASE + wave * FB_WAVE;         // this wave's LDS region
    const unsigned lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, CNNB_U16S * 2, 0x00020000);
    NQ_STAMP(0);

    // ---- stage the 15-frame window as two zero-bordered bf16 planes (hi, lo) [frame j + 1][mel m + 1]:
    //      the 3x3 taps of any output pixel are then at constant offsets from it, no bounds checks.
    //      All 12 global loads of the window (and the per-channel shifts of every layer) are requested up front: one
    //      memory latency instead of twelve.
    // top_db floor of the clip: precomputed (nisqa_mel_finalize), or taken here from the encoded running maximum the mel
    // kernel published (whole-forward path: one tiny kernel launch less)
    const float fl = SEGX ? -3.0e38f : clip_max_enc ? dec_ordered(clip_max_enc[b]) - top_db : clip_floor[b];
    const float* src = SEGX ? seg_x + ((size_t)b * seg_L + k) * 720
                            : mel_tm + (size_t)(frame_off[b] + k * seg_hop) * 48;
    float vraw[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const int i0 = lane + 64 * q;
        vraw[q] = (valid && (q < 11 || lane < 16)) ? src[i0] : 0.f;
    }
    const float tn1 = cw[CNN_T1 + (lane & 15)], tn2 = cw[CNN_T2 + (lane & 31)];
    const float tn3[2] = {cw[CNN_T3 + (lane & 31)], cw[CNN_T3 + 32 + (lane & 31)]};
    const float tn4[2] = {cw[CNN_T4 + (lane & 31)], cw[CNN_T4 + 32 + (lane & 31)]};
    const float tn5 = cw[CNN_T5 + 16 * wave + (lane & 15)], tn6 = cw[CNN_T6 + 16 * wave + (lane & 15)];
    // the window's values (floored at the clip's top_db level); F16: scaled by 2^e0, e0 from the window's own largest magnitude
    float vin[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) vin[q] = !valid ? 0.f : SEGX ? vraw[q] : fmaxf(vraw[q], fl);
    int e_in = 0;                                        // scale exponent of the tensor the next layer reads (F16)
    float m_in = 0.f;                                    // its largest magnitude, unscaled
    float s0 = 1.f;
    if (F16) {
        float mr = 0.f;
#pragma unroll
        for (int q = 0; q < 12; ++q) mr = fmaxf(mr, __builtin_fabsf(vin[q]));
        m_in = wave_max_nonneg(mr);
        e_in = f16_scale_exp(m_in);
        s0 = pow2_f32(e_in);
    }
    {
        const unsigned pb = R + FB_PATCH;
        // zero the patch planes (213 x 16 bytes) and the shared zero block (every wave writes the same zeros)
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (it < 3 || lane < (2 * FB_PPLANE + 15) / 16 - 192) lds_st128(pb + (lane + 64 * it) * 16, f32x4{0.f, 0.f, 0.f, 0.f});
        // (this store goes through the dynamic-LDS symbol on purpose: a kernel that only touches LDS through integer
        // addresses is compiled as one that uses no LDS at all, and then computes garbage)
        if (lane < 32) ((unsigned*)(smem + FB_ZADDR))[lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (!SEGX) {
            // element i0 = lane + 64 q of the [15][48] window is (frame j, mel m) = divmod(i0, 48); with q = 3 t + u that
            // is j = q + t + (lane + 16 u) / 48, m = (lane + 16 u) % 48: three lane-dependent store bases, the rest are
            // immediates
            unsigned ob[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int e = lane + 16 * u, j0 = e >= 48 ? 1 : 0, m = e - 48 * j0;
                ob[u] = pb + ((j0 + 1) * 50 + m + 1) * 2;
            }
#pragma unroll
            for (int q = 0; q < 12; q += 2) {
                const float v0 = vin[q] * s0, v1 = vin[q + 1] * s0;
                lds_store_pair_fmt<FMT>(ob[q % 3] + (q + q / 3) * 100, ob[(q + 1) % 3] + (q + 1 + (q + 1) / 3) * 100, FB_PPLANE, v0, v1,
                                               dummy_mx, true, q + 1 < 11 || lane < 16);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const int i0 = lane + 64 * q;
                const int m = i0 / 15, j = i0 - 15 * m;
                if (i0 < 720) lds_store_one_fmt<FMT>(pb + ((j + 1) * 50 + (m + 1)) * 2, FB_PPLANE, vin[q] * s0, dummy_mx);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    NQ_STAMP(1);

    const int i = lane & 31, hfi = (i >> 2) & 1, qi = (i & 3) + 4 * (i >> 3);
    const int n = lane & 31, hf = lane >> 5, h = lane >> 5;

    // ---- conv1 1->16 + pool 48x15 -> 24x7 on the matrix pipe, two output pixels per row.  A row = the mel pair
    //      (m0, m0 + 1) of one frame x; N = 32 = (channel c, pair member dm); K = 12 of 16 = (frame tap kx, mel m0 - 1 +
    //      dmm), dmm = 0..3: the four mels of one kx are CONTIGUOUS in the bordered patch, so the 8 k-slots of a lane are
    //      two pairs of dwords per plane and no packing.  B[k][n] = w[c][dmm - dm][kx] (zero outside the kernel), packed
    //      by weights.py.  Each lane half owns 12 pooled rows gl = mel pairs; the 16 rows of its tile are the frames.
    {
        f32x4 w1[2];                                      // weights hi and the first residual term (16 mantissa bits)
#pragma unroll
        for (int t = 0; t < 2; ++t) w1[t] = wfrag_load(wrs, lane16, (CNNB_W1 + t * 512) * 2);
        // F16: |y1| <= m0 * G1 + T1 fixes the scale of conv1's output before its first store
        float tn = tn1, c1 = 1.f, ms1 = 0.f;
        int e1 = 0;
        if (F16) {
            e1 = f16_scale_exp(fmaf(m_in, meta_f[8], meta_f[16]));
            c1 = pow2_f32(e1 - e_in - meta_i[0]);
            tn = tn1 * pow2_f32(e1);
        }
        // lane half 0: k-slots 0..7 = (kx 0, kx 1); half 1: k-slots 8..11 = kx 2 (12..15 meet zero weights: kx 2 again)
        const int xq = min(qi, 14);                       // row 15 of a tile is padding (result unused)
        unsigned rd_a = R + FB_PATCH + ((xq + (h ? 2 : 0)) * 50 + 24 * hfi) * 2;
        unsigned rd_b = R + FB_PATCH + ((xq + (h ? 2 : 1)) * 50 + 24 * hfi) * 2;
        // The pooled value of a mel pair needs both pair members, which sit 16 lanes apart (columns n and n ^ 16).  Of the
        // 14 pooled values of an iteration (2 mel pairs x 7 frame windows, consecutive pixels v = 7 tt + bb of A1), lane
        // group dm = 0 finalises the even ones and dm = 1 the odd ones: each sends the partner the values it does not own
        // (ONE ds_swizzle per value pair), takes the maximum, splits it into hi + lo and stores both planes.
        const bool is_b = (n & 16) != 0;
        const unsigned mb = is_b ? ~0u : 0u;             // bit select (v_bfi_b32): a ternary on r[] becomes an indexed stack array
        unsigned wr = R + (12 * hf * 7) * FB_RS1 + (n & 15) * 2 + (is_b ? FB_RS1 : 0);
        for (int g2 = 0; g2 < 6; ++g2) {
            f32x16 acc[2];
            f32x4 xa[2][2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int t = 0; t < 2; ++t) {              // dword reads (ds_read2_b32): the pairs are only 4-byte aligned
                    const unsigned pa = rd_a + 4 * tt + t * FB_PPLANE, pq = rd_b + 4 * tt + t * FB_PPLANE;
                    xa[tt][t] = f32x4{__uint_as_float(lds_ld32(pa)), __uint_as_float(lds_ld32(pa + 4)),
                                      __uint_as_float(lds_ld32(pq)), __uint_as_float(lds_ld32(pq + 4))};
                }
            acc[0] = zero16();
            acc[1] = zero16();
            // x = hi + lo carries 16 mantissa bits: 6e-4 dB at |80| dB, the size of the mel stage's own deviation from
            // the oracle (2.6e-4 dB) and three orders below what moves a MOS by 1e-3; smallest products first
            if (FMT == NQ_FMT_F16X4) { acc[0] = mfma32_fmt<FMT>(xa[0][1], w1[1], acc[0]); acc[1] = mfma32_fmt<FMT>(xa[1][1], w1[1], acc[1]); }
            acc[0] = mfma32_fmt<FMT>(xa[0][1], w1[0], acc[0]); acc[1] = mfma32_fmt<FMT>(xa[1][1], w1[0], acc[1]);
            acc[0] = mfma32_fmt<FMT>(xa[0][0], w1[1], acc[0]); acc[1] = mfma32_fmt<FMT>(xa[1][0], w1[1], acc[1]);
            acc[0] = mfma32_fmt<FMT>(xa[0][0], w1[0], acc[0]); acc[1] = mfma32_fmt<FMT>(xa[1][0], w1[0], acc[1]);
            // ReLU(. + shift) is monotone, so it is applied before the pair maximum, which is then taken on non-negative
            // floats -- as unsigned integers
            unsigned r[14];
#pragma unroll
            for (int v = 0; v < 14; ++v) {
                const int tt = v / 7, bb = v - 7 * tt;
                const float mx = fmaxf(fmaxf(acc[tt][2 * bb], acc[tt][2 * bb + 1]), acc[tt][2 * bb + 2]);   // frames
                r[v] = __float_as_uint(epi_fmt<FMT>(mx, c1, tn));
            }
            unsigned got[7];
#pragma unroll
            for (int kk = 0; kk < 7; ++kk)                 // (lanes ^ 16 on the LDS pipe; v_permlane16_swap costs ~20 VALU cycles)
                got[kk] = (unsigned)__builtin_amdgcn_ds_swizzle((int)((r[2 * kk] & mb) | (r[2 * kk + 1] & ~mb)), 0x401F);
            float fin[7];
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) {
                const unsigned own = (r[2 * kk + 1] & mb) | (r[2 * kk] & ~mb);
                fin[kk] = __uint_as_float(max(own, got[kk]));
            }
#pragma unroll
            for (int kk = 0; kk < 6; kk += 2) lds_store_pair_fmt<FMT>(wr + 2 * kk * FB_RS1, wr + 2 * (kk + 1) * FB_RS1, FB_P1, fin[kk], fin[kk + 1], ms1);
            lds_store_one_fmt<FMT>(wr + 12 * FB_RS1, FB_P1, fin[6], ms1);
            rd_a += 8; rd_b += 8;                          // mel m0 = 2 gl, gl = 12 hfi + 2 g2 + tt
            wr += 14 * FB_RS1;
        }
        if (F16) { m_in = wave_max_nonneg(ms1) * pow2_f32(-e1); e_in = e1; }
    }

    NQ_STAMP(2);
    // ---- conv2 16->32 on 24x7, pool -> 12x5 (row maps as in cnn.hip)
    {
        f32x16 acc[6][1];
        unsigned base[6], m9[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            acc[t][0] = zero16();
            const int u = 16 * t + qi;
            const int gl = u / 14, w = u % 14, yy = w / 7;
            const int py = 2 * (6 * hfi + gl) + yy, px = w - 7 * yy;
            m9[t] = tap_mask(u < 84, py, px, 24, 7);
            base[t] = R + ((py - 1) * 7 + (px - 1)) * FB_RS1 + (h << 4);
        }
        conv_k_bf16<16, 6, 1, 7, FB_RS1, FB_P1, FB_ZADDR, false, NQ_RING2, FMT>(acc, wrs, CNNB_W2 * 2, lane16, base, m9);
        NQ_STAMP(3);
        float tn = tn2, c2 = 1.f, ms2 = 0.f;
        int e2 = 0;
        if (F16) {
            e2 = f16_scale_exp(fmaf(m_in, meta_f[9], meta_f[17]));
            c2 = pow2_f32(e2 - e_in - meta_i[1]);
            tn = tn2 * pow2_f32(e2);
        }
        const unsigned wr = R + (6 * hf * 5) * FB_RS2 + n * 2;
#pragma unroll
        for (int k2 = 0; k2 < 30; k2 += 2) {              // pooled pixel k = gl * 5 + bb, two per packed split
            float pv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int gl = (k2 + e) / 5, bb = (k2 + e) % 5;
                float mx = -3.0e38f;
#pragma unroll
                for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                    for (int x = bwin75_lo(bb); x < bwin75_hi(bb); ++x) {
                        const int u = 14 * gl + 7 * yy + x;
                        mx = fmaxf(mx, acc[u >> 4][0][u & 15]);
                    }
                pv[e] = epi_fmt<FMT>(mx, c2, tn);
            }
            lds_store_pair_fmt<FMT>(wr + k2 * FB_RS2, wr + (k2 + 1) * FB_RS2, FB_P2, pv[0], pv[1], ms2);
        }
        if (F16) { m_in = wave_max_nonneg(ms2) * pow2_f32(-e2); e_in = e2; }
    }

    NQ_STAMP(4);
    unsigned base34[2], m34[2];                           // conv3 and conv4 share the 12 x 5 geometry
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int u = 16 * t + qi;
        const int gl = u / 10, w = u % 10, yy = w / 5;
        const int py = 2 * (3 * hfi + gl) + yy, px = w - 5 * yy;
        m34[t] = tap_mask(u < 30, py, px, 12, 5);
        base34[t] = (py - 1) * 5 + (px - 1);              // pixel index of tap (-1, -1)
    }

    // ---- conv3 32->64 on 12x5
    {
        f32x16 acc[2][2];
        unsigned base[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[t][nt] = zero16();
            base[t] = R + base34[t] * FB_RS2 + (h << 4);
        }
        conv_k_bf16<32, 2, 2, 5, FB_RS2, FB_P2, FB_ZADDR, true, NQ_RING34, FMT>(acc, wrs, CNNB_W3 * 2, lane16, base, m34);
        NQ_STAMP(5);
        float tn[2] = {tn3[0], tn3[1]}, c3 = 1.f, ms3 = 0.f;
        int e3 = 0;
        if (F16) {
            e3 = f16_scale_exp(fmaf(m_in, meta_f[10], meta_f[18]));
            c3 = pow2_f32(e3 - e_in - meta_i[2]);
            tn[0] *= pow2_f32(e3);
            tn[1] *= pow2_f32(e3);
        }
        const unsigned wr = R + (6 * hf * 5) * FB_RS3 + n * 2;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int u = 16 * t + r;                  // rows u, u + 1: pixel (2 gl + yy) * 5 + x = u (same order)
                    if (u < 30)
                        lds_store_pair_fmt<FMT>(wr + u * FB_RS3 + 64 * nt, wr + (u + 1) * FB_RS3 + 64 * nt, FB_P3,
                                                epi_fmt<FMT>(acc[t][nt][r], c3, tn[nt]), epi_fmt<FMT>(acc[t][nt][r + 1], c3, tn[nt]), ms3);
                }
        if (F16) { m_in = wave_max_nonneg(ms3) * pow2_f32(-e3); e_in = e3; }
    }

    // ---- conv4 64->64 on 12x5, pool -> 6x3.  The pooled outputs of the workgroup's four segments go to a
    //      SHARED pair of bf16 planes S4[72 px][64 ch] (row = 18 * wave + pixel) for the N-split conv5/conv6.
    NQ_STAMP(6);
    const unsigned S4 = FB_BASE;                          // 2 planes x FB_PS (wave 0/1 regions; their A3 is dead by then)
    const unsigned S5 = FB_BASE + 2 * FB_WAVE;            // conv5 output, same shape (wave 2/3 regions)
    // conv5 / conv6 weight fragments of this wave (its 16 output channels), [step][hi,lo][lane][8]: rings of 4 / 8
    // K-steps, requested 3 / 7 steps ahead -- a step of conv5 (conv6) is only 15 (6) short MFMAs, an L2 round trip
    // several steps long.  The first requests go out before the previous layer's epilogue.
    const int w5b = __builtin_amdgcn_readfirstlane((CNNB_W5 + wave * (18 * 2 * 512)) * 2);
    const int w6b = __builtin_amdgcn_readfirstlane((CNNB_W6 + wave * (18 * 2 * 512)) * 2);
    f32x4 b5[4][2], b6[8][2];
    {
        f32x16 acc[2][2];
        unsigned base[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[t][nt] = zero16();
            base[t] = R + base34[t] * FB_RS3 + (h << 4);
        }
        conv_k_bf16<64, 2, 2, 5, FB_RS3, FB_P3, FB_ZADDR, true, NQ_RING34, FMT>(acc, wrs, CNNB_W4 * 2, lane16, base, m34);
        NQ_STAMP(7);
        float tn[2] = {tn4[0], tn4[1]}, c4 = 1.f, ms4 = 0.f;
        int e4 = 0;
        if (F16) {
            e4 = f16_scale_exp(fmaf(m_in, meta_f[11], meta_f[19]));
            c4 = pow2_f32(e4 - e_in - meta_i[3]);
            tn[0] *= pow2_f32(e4);
            tn[1] *= pow2_f32(e4);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) { b5[g][0] = wfrag_load(wrs, lane16, w5b + g * 2048); b5[g][1] = wfrag_load(wrs, lane16, w5b + g * 2048 + 1024); }
        __syncthreads();                   // every wave has consumed its A3: the regions may be re-used
        const unsigned wr = S4 + (18 * wave + 9 * hf) * FB_RS3 + n * 2;
        float* dst = P3 ? p3 + (size_t)p * (18 * 64) : nullptr;
#pragma unroll
        for (int gl = 0; gl < 3; ++gl)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) {
                float pv[2];                                // the pooled pixel's channels n and n + 32
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float mx = -3.0e38f;
#pragma unroll
                    for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                        for (int x = bwin53_lo(bb); x < bwin53_hi(bb); ++x) {
                            const int u = 10 * gl + 5 * yy + x;
                            mx = fmaxf(mx, acc[u >> 4][nt][u & 15]);
                        }
                    pv[nt] = epi_fmt<FMT>(mx, c4, tn[nt]);
                    if (P3 && valid) dst[((3 * hf + gl) * 3 + bb) * 64 + n + 32 * nt] = pv[nt];   // optional fp32 copy (debug / parity)
                }
                lds_store_pair_fmt<FMT>(wr + (gl * 3 + bb) * FB_RS3, wr + (gl * 3 + bb) * FB_RS3 + 64, FB_PS, pv[0], pv[1], ms4);
            }
        if (F16) {
            // conv5 / conv6 run over the four segments' rows at once: every row carries its own segment's scales.  conv5's output
            // scale comes from this segment's measured conv4 maximum; conv6 writes fp32 features and needs only its input's scale.
            const float m4 = wave_max_nonneg(ms4) * pow2_f32(-e4);
            const int e5 = f16_scale_exp(fmaf(m4, meta_f[12], meta_f[20]));
            if (lane < 18) {
                lds_st32(FB_TAB5 + (18 * wave + lane) * 8, __float_as_uint(pow2_f32(e5 - e4 - meta_i[4])));
                lds_st32(FB_TAB5 + (18 * wave + lane) * 8 + 4, __float_as_uint(pow2_f32(e5)));
            }
            if (lane < 6) lds_st32(FB_TAB6 + (6 * wave + lane) * 4, __float_as_uint(pow2_f32(-(e5 + meta_i[5]))));
        }
    }
    __syncthreads();
    NQ_STAMP(8);

    // ---- conv5 / conv6 with N split over the waves: wave w owns output channels 16w..16w+15 of ALL four
    //      segments (72 / 24 output rows in 16-row tiles of v_mfma_f32_16x16x32_bf16) and streams its private
    //      quarter of the weights from L2 (no staging, no barriers); fragments [wave][step][hi,lo][lane][8].
    {
        const int i16 = lane & 15, kg = lane >> 4;
        const int ch = 16 * wave + i16;                    // D-fragment column = output channel
        // conv5: rows rho = 16 t + i16 <-> (slot = rho / 18, pixel = rho % 18), 6 x 3 image per slot
        f32x4 acc5[5];
        unsigned base5[5], m5[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            acc5[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int rho = 16 * t + i16;
            const int slot = rho / 18, pix = rho - 18 * slot, ry = pix / 3, rx = pix - 3 * ry;
            m5[t] = tap_mask(rho < 72, ry, rx, 6, 3);
            base5[t] = S4 + (slot * 18 + (ry - 1) * 3 + (rx - 1)) * FB_RS3 + (kg << 4);
        }
        unsigned a5h[5], a5l[5];
        f32x4 a5[2][5][2];                                  // A rows one step ahead: [buffer][tile][hi, lo]
        auto load_a5 = [&](int g) {
            const int tap = g >> 1, s = g & 1;
            const int tapoff = ((tap / 3) * 3 + tap % 3) * FB_RS3;
            if (s == 0) {
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const bool ok = (m5[t] >> tap) & 1u;
                    a5h[t] = ok ? base5[t] : FB_ZADDR - tapoff;
                    a5l[t] = ok ? base5[t] + FB_PS : FB_ZADDR - tapoff;
                }
            }
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                a5[g & 1][t][0] = lds_ld128(a5h[t] + tapoff + 64 * s);
                a5[g & 1][t][1] = lds_ld128(a5l[t] + tapoff + 64 * s);
            }
        };
        load_a5(0);
#pragma unroll
        for (int g = 0; g < 18; ++g) {
            if (g + 3 < 18) { b5[(g + 3) & 3][0] = wfrag_load(wrs, lane16, w5b + (g + 3) * 2048); b5[(g + 3) & 3][1] = wfrag_load(wrs, lane16, w5b + (g + 3) * 2048 + 1024); }
            if (g + 1 < 18) load_a5(g + 1);
            mma16_pair_fmt<FMT, 5>(acc5, a5[g & 1], b5[g & 3]);
        }
        NQ_STAMP(9);
#pragma unroll
        for (int g = 0; g < 7; ++g) { b6[g][0] = wfrag_load(wrs, lane16, w6b + g * 2048); b6[g][1] = wfrag_load(wrs, lane16, w6b + g * 2048 + 1024); }
        {
            const unsigned wr = S5 + (4 * kg) * FB_RS3 + ch * 2;
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int r = 0; r < 4; r += 2)
                    if (t < 4 || kg < 2) {                     // rho = 16 t + 4 kg + r < 72
                        f32x2_t cs0 = {1.f, 1.f}, cs1 = {1.f, 1.f};      // F16: {2^(e5 - e4 - kw5), 2^e5} of the rows' segments
                        if (F16) { cs0 = lds_ld64(FB_TAB5 + (16 * t + r) * 8 + kg * 32); cs1 = lds_ld64(FB_TAB5 + (16 * t + r + 1) * 8 + kg * 32); }
                        lds_store_pair_fmt<FMT>(wr + (16 * t + r) * FB_RS3, wr + (16 * t + r + 1) * FB_RS3, FB_PS,
                                                epi_fmt<FMT>(acc5[t][r], cs0[0], tn5 * cs0[1]), epi_fmt<FMT>(acc5[t][r + 1], cs1[0], tn5 * cs1[1]), dummy_mx);
                    }
        }
        __syncthreads();
        NQ_STAMP(10);

        // conv6 (3 x 3 kernel, padding (1,0)) = padding-1 conv at the centre column: rows (slot, y), 24 of 32
        f32x4 acc6[2], acc6b[2];            // even / odd K-steps accumulate separately: four independent chains
        unsigned base6[2], m6[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            acc6[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc6b[t] = acc6[t];
            const int rho = 16 * t + i16;
            const int slot = rho / 6, y = rho - 6 * slot;
            m6[t] = tap_mask(rho < 24, y, 1, 6, 3);          // output column x = 1: input columns 0..2 are all inside
            base6[t] = S5 + (slot * 18 + (y - 1) * 3) * FB_RS3 + (kg << 4);
        }
        unsigned a6h[2], a6l[2];
        f32x4 a6[2][2][2];
        auto load_a6 = [&](int g) {
            const int tap = g >> 1, s = g & 1;
            const int tapoff = ((tap / 3) * 3 + tap % 3) * FB_RS3;
            if (s == 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bool ok = (m6[t] >> tap) & 1u;
                    a6h[t] = ok ? base6[t] : FB_ZADDR - tapoff;
                    a6l[t] = ok ? base6[t] + FB_PS : FB_ZADDR - tapoff;
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a6[g & 1][t][0] = lds_ld128(a6h[t] + tapoff + 64 * s);
                a6[g & 1][t][1] = lds_ld128(a6l[t] + tapoff + 64 * s);
            }
        };
        load_a6(0);
#pragma unroll
        for (int g = 0; g < 18; ++g) {
            if (g + 7 < 18) { b6[(g + 7) & 7][0] = wfrag_load(wrs, lane16, w6b + (g + 7) * 2048); b6[(g + 7) & 7][1] = wfrag_load(wrs, lane16, w6b + (g + 7) * 2048 + 1024); }
            if (g + 1 < 18) load_a6(g + 1);
            if (g & 1) mma16_pair_fmt<FMT, 2>(acc6b, a6[1], b6[g & 7]);
            else mma16_pair_fmt<FMT, 2>(acc6, a6[0], b6[g & 7]);
        }
        NQ_STAMP(11);
        // this wave's 4 x 96 outputs (slot, channel * 6 + y) go through S4 (dead since the barrier above) so that the
        // feature rows leave as 16-byte stores: 384 contiguous bytes per slot
        const unsigned fo = S4 + wave * 2048;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rho = 16 * t + 4 * kg + r;
                const int slot = rho / 6, y = rho - 6 * slot;
                if (rho < 24) {
                    const float c6 = F16 ? __uint_as_float(lds_ld32(FB_TAB6 + (16 * t + r) * 4 + kg * 16)) : 1.f;   // 2^-(e5 + kw6) of the row's segment
                    lds_st32(fo + (slot * 96 + i16 * 6 + y) * 4, __float_as_uint(epi_fmt<FMT>(acc6[t][r] + acc6b[t][r], c6, tn6)));
                }
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q0 = 0; q0 < 96; q0 += 64) {
            const int q = q0 + lane;                                         // float4 index: slot = q / 24
            const int slot = q / 24;
            if (q < 96 && slot < nvalid)
                *(f32x4*)(feat + (size_t)(p0 + slot) * 384 + 96 * wave + 4 * (q - 24 * slot)) = lds_ld128(fo + 16 * q);
        }
    }
    NQ_STAMP_END(g_phase_clk, ((blockIdx.y * gridDim.x + blockIdx.x) << 2) + wave);
}

// ---- kernels: one body, three operand formats -------------------------------------------------------------------------------------
template <bool SEGX, bool P3>
__global__ __launch_bounds__(256, FB_WGS) void cnn_front_bf16_kernel(
    const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off, const int32_t* __restrict__ tok_off,
    const int32_t* __restrict__ n_wins, const float* __restrict__ clip_floor, int n_clips, int seg_hop,
    const float* __restrict__ cw, const unsigned short* __restrict__ wb, float* __restrict__ p3, float* __restrict__ feat,
    const float* __restrict__ seg_x, int seg_L, const uint32_t* __restrict__ clip_max_enc, float top_db) {
    cnn_front_split_body<NQ_FMT_BF16X3, SEGX, P3>(mel_tm, frame_off, tok_off, n_wins, clip_floor, n_clips, seg_hop, cw, wb, p3, feat, seg_x, seg_L,
                                                  clip_max_enc, top_db);
}
// fp32 operands as two f16 terms of the power-of-two-scaled tensors; P4: all four term products ('f16x4'), else hh + hl + lh ('f16x3')
template <bool P4, bool SEGX>
__global__ __launch_bounds__(256, FB_WGS) void cnn_front_f16_kernel(
    const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off, const int32_t* __restrict__ tok_off,
    const int32_t* __restrict__ n_wins, const float* __restrict__ clip_floor, int n_clips, int seg_hop,
    const float* __restrict__ cw, const unsigned short* __restrict__ wb, float* __restrict__ p3, float* __restrict__ feat,
    const float* __restrict__ seg_x, int seg_L, const uint32_t* __restrict__ clip_max_enc, float top_db) {
    cnn_front_split_body<P4 ? NQ_FMT_F16X4 : NQ_FMT_F16X3, SEGX, false>(mel_tm, frame_off, tok_off, n_wins, clip_floor, n_clips, seg_hop, cw, wb, p3,
                                                                       feat, seg_x, seg_L, clip_max_enc, top_db);
}

typedef void (*fb_kernel_t)(const float*, const int32_t*, const int32_t*, const int32_t*, const float*, int, int, const float*,
                            const unsigned short*, float*, float*, const float*, int, const uint32_t*, float);
// fmt: 0 bf16x3, 1 f16x3, 2 f16x4.  80.5 KB of dynamic LDS is above the 64 KB default: every instantiation is opted in once per
// device ordinal (a process may drive several GPUs), like the three-term kernel's launcher
static int fb_launch(int fmt, bool segx, bool p3, const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                     const int32_t* n_wins, const float* clip_floor, const uint32_t* clip_max_enc, float top_db, int32_t n_clips,
                     int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w, const uint16_t* cnn_wb, float* p3_opt, float* feat,
                     const float* seg_x, int32_t seg_L, void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31) || seg_hop <= 0 || !cnn_wb || !feat || fmt < 0 || fmt > 2 ||
        (p3 && fmt != 0) || (segx && (!seg_x || seg_L <= 0)) || (!segx && !clip_floor && !clip_max_enc))
        return NISQA_ERR_ARG;
    static const fb_kernel_t kernels[7] = {
        cnn_front_bf16_kernel<false, false>, cnn_front_bf16_kernel<false, true>, cnn_front_bf16_kernel<true, false>,
        cnn_front_f16_kernel<false, false>, cnn_front_f16_kernel<false, true>, cnn_front_f16_kernel<true, false>, cnn_front_f16_kernel<true, true>};
    const int which = fmt == 0 ? (segx ? 2 : p3 ? 1 : 0) : 3 + 2 * (fmt - 1) + (segx ? 1 : 0);
    NQ_LAUNCH_BEGIN();
    static std::atomic<bool> lds_ok[7][64];
    if (nq_lds_opt_in((const void*)kernels[which], (int)FB_LDS, lds_ok[which])) return 2;
    hipLaunchKernelGGL(kernels[which], dim3(total_tok_padded / 4), dim3(256), FB_LDS, (hipStream_t)stream, mel_tm, frame_off, tok_off, n_wins,
                       clip_floor, n_clips, seg_hop, cnn_w, cnn_wb, p3_opt, feat, seg_x, seg_L, clip_max_enc, top_db);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_cnn_adapt_bf16(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                                    const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                                    int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w,
                                    const uint16_t* cnn_wb, float* p3_opt, float* feat, void* stream) {
    if (!clip_floor) return NISQA_ERR_ARG;
    return fb_launch(0, false, p3_opt != nullptr, mel_tm, frame_off, tok_off, n_wins, clip_floor, nullptr, 0.f, n_clips, total_tok_padded, seg_hop,
                     cnn_w, cnn_wb, p3_opt, feat, nullptr, 0, stream);
}

// nisqa_cnn_adapt_bf16 with the per-clip floor derived in the kernel from the mel kernel's clip_max_enc (internal.hpp)
int nq_cnn_adapt_bf16_from_max(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                               const int32_t* n_wins, const uint32_t* clip_max_enc, float top_db, int32_t n_clips,
                               int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w, const uint16_t* cnn_wb,
                               float* feat, void* stream) {
    if (!clip_max_enc) return NISQA_ERR_ARG;
    return fb_launch(0, false, false, mel_tm, frame_off, tok_off, n_wins, nullptr, clip_max_enc, top_db, n_clips, total_tok_padded, seg_hop, cnn_w,
                     cnn_wb, nullptr, feat, nullptr, 0, stream);
}

extern "C" int nisqa_cnn_adapt_segments_bf16(const float* x, int32_t seg_len_padded, const int32_t* tok_off,
                                             const int32_t* n_wins, int32_t n_clips, int32_t total_tok_padded,
                                             const float* cnn_w, const uint16_t* cnn_wb, float* feat, void* stream) {
    return fb_launch(0, true, false, nullptr, nullptr, tok_off, n_wins, nullptr, nullptr, 0.f, n_clips, total_tok_padded, 1, cnn_w, cnn_wb, nullptr,
                     feat, x, seg_len_padded, stream);
}

// ---- the f16 formats (cnn_wh: nisqa_amd.weights.pack_adapt_cnn_f16, CNNH_U16S uint16) -------------------------------------------------
extern "C" int nisqa_cnn_adapt_f16(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off, const int32_t* n_wins,
                                   const float* clip_floor, int32_t n_clips, int32_t total_tok_padded, int32_t seg_hop,
                                   const float* cnn_w, const uint16_t* cnn_wh, int32_t products, float* feat, void* stream) {
    if (!clip_floor || (products != 3 && products != 4)) return NISQA_ERR_ARG;
    return fb_launch(products - 2, false, false, mel_tm, frame_off, tok_off, n_wins, clip_floor, nullptr, 0.f, n_clips, total_tok_padded, seg_hop,
                     cnn_w, cnn_wh, nullptr, feat, nullptr, 0, stream);
}
int nq_cnn_adapt_f16_from_max(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off, const int32_t* n_wins,
                              const uint32_t* clip_max_enc, float top_db, int32_t n_clips, int32_t total_tok_padded, int32_t seg_hop,
                              const float* cnn_w, const uint16_t* cnn_wh, int32_t products, float* feat, void* stream) {
    if (!clip_max_enc || (products != 3 && products != 4)) return NISQA_ERR_ARG;
    return fb_launch(products - 2, false, false, mel_tm, frame_off, tok_off, n_wins, nullptr, clip_max_enc, top_db, n_clips, total_tok_padded,
                     seg_hop, cnn_w, cnn_wh, nullptr, feat, nullptr, 0, stream);
}
extern "C" int nisqa_cnn_adapt_segments_f16(const float* x, int32_t seg_len_padded, const int32_t* tok_off, const int32_t* n_wins,
                                            int32_t n_clips, int32_t total_tok_padded, const float* cnn_w, const uint16_t* cnn_wh,
                                            int32_t products, float* feat, void* stream) {
    if (products != 3 && products != 4) return NISQA_ERR_ARG;
    return fb_launch(products - 2, true, false, nullptr, nullptr, tok_off, n_wins, nullptr, nullptr, 0.f, n_clips, total_tok_padded, 1, cnn_w, cnn_wh,
                     nullptr, feat, x, seg_len_padded, stream);
}
