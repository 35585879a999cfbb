// The whole AdaptCNN (conv1..conv6, BatchNorm folded, adaptive max-pools) at fp32 operand precision on the bf16 matrix pipe
// ("bf16x6") -- same role, inputs and outputs as cnn_front_kernel + cnn_back_kernel in cnn.hip (reference
// nisqa/NISQA_lib.py:2239-2282, 487-502, 688-710).
//
// Every fp32 operand x is carried as THREE bf16 terms x = hi + mid + lo (8 + 8 + 8 significant bits, each rounded to nearest:
// the split is EXACT, no bit of the fp32 value is lost) and a product is formed as the six MFMA products hh + hm + mh + hl +
// lh + mm on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The three dropped products (ml, lm, ll) are at most 2 x 2^-24 of
// the product and typically 0.5 x 2^-24 rms -- the size of the rounding an fp32 multiply-add applies to the same product.
// Against float64 the results are as close as the exact-fp32 MFMA kernels' (tests/test_gpu_parity.py:
// test_rounding_error_of_the_precision_modes_against_float64; DESIGN.md 4.5 "bf16x6"), at 16 / 6 = 2.7 x their matrix-pipe rate.
//
// Structure: cnn_bf16.hip's (a workgroup is four waves = four consecutive segments; conv1..conv4 wave-private and
// barrier-free over LDS planes, pixel-major, rows padded by 16 bytes; conv1 on the matrix pipe with two mel-adjacent output
// pixels per MFMA row; conv5 / conv6 batched over the four segments with the output channels split over the waves) with
// three planes per activation tensor.  117 KB of LDS per workgroup: ONE workgroup per CU, one wave per SIMD on the
// 512-register budget -- the K loops keep the A rows of the next step and two steps of weight fragments in flight themselves
// (conv_k_terms; tools/micro/klx6.hip: 96 % matrix-pipe duty at 1.79 GHz in the conv3 + conv4 loops, the chip's power envelope).
#include "common.hpp"
#include "layout.hpp"
#include <stdlib.h>
#include <atomic>
#include "conv_bf16.hpp"
#include "internal.hpp"
#include "../../include/nisqa_hip.h"

#define XT 3                               /* terms per operand */
#define X_RS1 48                           /* A1: 168 px x 16 ch */
#define X_P1 (168 * X_RS1)
#define X_RS2 80                           /* A2: 60 px x 32 ch */
#define X_P2 (60 * X_RS2)
#define X_RS3 144                          /* A3: 60 px x 64 ch; S4 / S5: 72 rows x 64 ch */
#define X_P3 (60 * X_RS3)
#define X_PS (72 * X_RS3)
#define X_PATCH (XT * X_P1)                /* conv1 input: XT zero-bordered bf16 planes [17][50] behind the A1 planes */
#define X_PPLANE 1700
#define X_ZADDR 2048u
#define X_BASE 2176u
#define X_WAVE 29312u
#define X_LDS (X_BASE + 4 * X_WAVE)        /* 119 424 B: one workgroup per CU */
static_assert(X_PATCH + XT * X_PPLANE <= X_WAVE && XT * X_P3 <= X_WAVE && XT * X_PS <= 2 * X_WAVE, "LDS plan");
static_assert(X_LDS <= 160 * 1024, "LDS");

__device__ constexpr int xwin75_lo(int b) { return b == 0 ? 0 : b == 1 ? 1 : b == 2 ? 2 : b == 3 ? 4 : 5; }
__device__ constexpr int xwin75_hi(int b) { return b == 0 ? 2 : b == 1 ? 3 : b == 2 ? 5 : b == 3 ? 6 : 7; }
__device__ constexpr int xwin53_lo(int b) { return b == 0 ? 0 : b == 1 ? 1 : 3; }
__device__ constexpr int xwin53_hi(int b) { return b == 0 ? 2 : b == 1 ? 4 : 5; }

// layer-boundary stamps of the phase clock (tools/phase_clock.py with NQ_PRECISION=bf16x6; empty macros unless built with -DNQ_EXPERIMENTAL)
NQ_CLK_EXPORT(g_phase_clk6, nisqa_debug_phase_clock6)

// 16x16x32 products of T-term operands for conv5 / conv6 (smallest first)
template <int MT>
NQ_DEV void mma16_terms(f32x4 (&acc)[MT], const f32x4 (&a)[MT][XT], const f32x4 (&b)[XT]) {
#pragma unroll
    for (int order = XT - 1; order >= 0; --order)
#pragma unroll
        for (int i = order; i >= 0; --i) {
            const int j = order - i;
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = mfma_bf16x16(a[t][i], b[j], acc[t]);
        }
}

// SEGX: the input is the reference's segment tensor x[B][L][1][48][15] (inner-operator mode, NISQA_lib.py:260-268) instead of the
// spectrogram; no dB floor is applied (x is already clamped)
template <bool SEGX>
__global__ __launch_bounds__(256, 1) void cnn_front_bf16x6_kernel(
    const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off,
    const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
    const float* __restrict__ clip_floor, int n_clips, int seg_hop,
    const float* __restrict__ cw, const unsigned short* __restrict__ wb, float* __restrict__ feat,
    const uint32_t* __restrict__ clip_max_enc, float top_db, const float* __restrict__ seg_x, int seg_L) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NQ_STAMP_BEGIN();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p0 = blockIdx.x * 4;                      // tok_off is a multiple of 32: no clip straddling
    const int b = __builtin_amdgcn_readfirstlane(find_segment_wave(tok_off, n_clips, p0, lane));
    const int k0 = p0 - tok_off[b];
    const int nvalid = min(4, n_wins[b] - k0);
    if (nvalid <= 0) return;                             // whole workgroup is padding
    const bool valid = wave < nvalid;                    // padding waves still walk the barriers (on zeros)
    const int k = k0 + wave;
    NQ_STAMP(0);
    const unsigned R = X_BASE + wave * X_WAVE;           // this wave's LDS region
    const unsigned lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, CNNX_U16S * 2, 0x00020000);

    // ---- stage the 15-frame window as XT zero-bordered bf16 planes [frame j + 1][mel m + 1]
    const float fl = SEGX ? -3.0e38f : clip_max_enc ? dec_ordered(clip_max_enc[b]) - top_db : clip_floor[b];
    const float* src = SEGX ? seg_x + ((size_t)b * seg_L + k) * 720 : mel_tm + (size_t)(frame_off[b] + k * seg_hop) * 48;
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
    {
        const unsigned pb = R + X_PATCH;
        // zero the patch planes (319 x 16 bytes) and the shared zero block (every wave writes the same zeros)
#pragma unroll
        for (int it = 0; it < 5; ++it)
            if (lane + 64 * it < (XT * X_PPLANE + 15) / 16) lds_st128(pb + (lane + 64 * it) * 16, f32x4{0.f, 0.f, 0.f, 0.f});
        // (through the dynamic-LDS symbol on purpose: a kernel that only touches LDS through integer addresses is compiled
        // as one that uses no LDS at all)
        if (lane < 32) ((unsigned*)(smem + X_ZADDR))[lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (!SEGX) {
            // element i0 = lane + 64 q of the [15][48] window is (frame j, mel m) = divmod(i0, 48); with q = 3 t + u that is
            // j = q + t + (lane + 16 u) / 48, m = (lane + 16 u) % 48: three lane-dependent store bases, the rest are immediates
            unsigned ob[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int e = lane + 16 * u, j0 = e >= 48 ? 1 : 0, m = e - 48 * j0;
                ob[u] = pb + ((j0 + 1) * 50 + m + 1) * 2;
            }
#pragma unroll
            for (int q = 0; q < 12; q += 2) {
                const float v0 = valid ? fmaxf(vraw[q], fl) : 0.f, v1 = valid ? fmaxf(vraw[q + 1], fl) : 0.f;
                lds_store_terms2<XT>(ob[q % 3] + (q + q / 3) * 100, ob[(q + 1) % 3] + (q + 1 + (q + 1) / 3) * 100, X_PPLANE, v0, v1,
                                     true, q + 1 < 11 || lane < 16);
            }
        } else {
            // the segment tensor is [mel m][frame j]: element i0 = (m, j) = divmod(i0, 15)
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const int i0 = lane + 64 * q;
                const int m = i0 / 15, j = i0 - 15 * m;
                if (i0 < 720) lds_store_terms<XT>(pb + ((j + 1) * 50 + (m + 1)) * 2, X_PPLANE, valid ? vraw[q] : 0.f);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    NQ_STAMP(1);

    const int i = lane & 31, hfi = (i >> 2) & 1, qi = (i & 3) + 4 * (i >> 3);
    const int n = lane & 31, hf = lane >> 5, h = lane >> 5;

    // ---- conv1 1->16 + pool 48x15 -> 24x7 on the matrix pipe, two output pixels per row (cnn_bf16.hip: same row / column
    //      / k-slot maps); the dB input and the weights are exact in their three terms
    // the first two K steps of conv2's fragments travel while conv1 runs (one wave per SIMD: nobody else covers the L2 round trip
    // a K loop otherwise opens with); conv3's are requested above conv2's epilogue, conv4's above conv3's
    conv_k_ring<XT, 1, 3> ring2;
    conv_k_preload(ring2, wrs, CNNX_W2 * 2, lane16);
    {
        f32x4 w1[1][XT];
#pragma unroll
        for (int t = 0; t < XT; ++t) w1[0][t] = wfrag_load(wrs, lane16, (CNNX_W1 + t * 512) * 2);
        const float tn = tn1;
        const int xq = min(qi, 14);                       // row 15 of a tile is padding (result unused)
        unsigned rd_a = R + X_PATCH + ((xq + (h ? 2 : 0)) * 50 + 24 * hfi) * 2;
        unsigned rd_b = R + X_PATCH + ((xq + (h ? 2 : 1)) * 50 + 24 * hfi) * 2;
        const bool is_b = (n & 16) != 0;
        const unsigned mb = is_b ? ~0u : 0u;
        unsigned wr = R + (12 * hf * 7) * X_RS1 + (n & 15) * 2 + (is_b ? X_RS1 : 0);
        for (int g2 = 0; g2 < 6; ++g2) {
            f32x16 acc[2][1];
            f32x4 xa[2][XT];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int t = 0; t < XT; ++t) {             // dword reads: the pairs are only 4-byte aligned
                    const unsigned pa = rd_a + 4 * tt + t * X_PPLANE, pq = rd_b + 4 * tt + t * X_PPLANE;
                    xa[tt][t] = f32x4{__uint_as_float(lds_ld32(pa)), __uint_as_float(lds_ld32(pa + 4)),
                                      __uint_as_float(lds_ld32(pq)), __uint_as_float(lds_ld32(pq + 4))};
                }
            acc[0][0] = zero16();
            acc[1][0] = zero16();
            mma_terms<XT, 2, 1>(acc, xa, w1);
            unsigned r[14];
#pragma unroll
            for (int v = 0; v < 14; ++v) {
                const int tt = v / 7, bb = v - 7 * tt;
                const float mx = fmaxf(fmaxf(acc[tt][0][2 * bb], acc[tt][0][2 * bb + 1]), acc[tt][0][2 * bb + 2]);   // frames
                r[v] = __float_as_uint(fmaxf(mx + tn, 0.f));
            }
            unsigned got[7];
#pragma unroll
            for (int kk = 0; kk < 7; ++kk)
                got[kk] = (unsigned)__builtin_amdgcn_ds_swizzle((int)((r[2 * kk] & mb) | (r[2 * kk + 1] & ~mb)), 0x401F);
            float fin[7];
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) {
                const unsigned own = (r[2 * kk + 1] & mb) | (r[2 * kk] & ~mb);
                fin[kk] = __uint_as_float(max(own, got[kk]));
            }
#pragma unroll
            for (int kk = 0; kk < 6; kk += 2) lds_store_terms2<XT>(wr + 2 * kk * X_RS1, wr + 2 * (kk + 1) * X_RS1, X_P1, fin[kk], fin[kk + 1]);
            lds_store_terms<XT>(wr + 12 * X_RS1, X_P1, fin[6]);
            rd_a += 8; rd_b += 8;
            wr += 14 * X_RS1;
        }
    }

    NQ_STAMP(2);
    // ---- conv2 16->32 on 24x7, pool -> 12x5
    conv_k_ring<XT, 2, 3> ring34;                         // conv3's, then conv4's first fragments
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
            base[t] = R + ((py - 1) * 7 + (px - 1)) * X_RS1 + (h << 4);
        }
        conv_k_terms_ring<XT, 16, 6, 1, 7, X_RS1, X_P1, X_ZADDR, 3, true, true>(acc, wrs, CNNX_W2 * 2, lane16, base, m9, ring2);
        NQ_STAMP(3);
        conv_k_preload(ring34, wrs, CNNX_W3 * 2, lane16);
        __builtin_amdgcn_sched_barrier(0);                // (hipcc would sink the requests below the epilogue, to their use)
        const unsigned wr = R + (6 * hf * 5) * X_RS2 + n * 2;
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
                    for (int x = xwin75_lo(bb); x < xwin75_hi(bb); ++x) {
                        const int u = 14 * gl + 7 * yy + x;
                        mx = fmaxf(mx, acc[u >> 4][0][u & 15]);
                    }
                pv[e] = fmaxf(mx + tn2, 0.f);
            }
            lds_store_terms2<XT>(wr + k2 * X_RS2, wr + (k2 + 1) * X_RS2, X_P2, pv[0], pv[1]);
        }
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
            base[t] = R + base34[t] * X_RS2 + (h << 4);
        }
        conv_k_terms_ring<XT, 32, 2, 2, 5, X_RS2, X_P2, X_ZADDR, 3, true, true>(acc, wrs, CNNX_W3 * 2, lane16, base, m34, ring34);
        NQ_STAMP(5);
        conv_k_preload(ring34, wrs, CNNX_W4 * 2, lane16);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned wr = R + (6 * hf * 5) * X_RS3 + n * 2;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int u = 16 * t + r;                  // rows u, u + 1: pixel (2 gl + yy) * 5 + x = u (same order)
                    if (u < 30)
                        lds_store_terms2<XT>(wr + u * X_RS3 + 64 * nt, wr + (u + 1) * X_RS3 + 64 * nt, X_P3,
                                             fmaxf(acc[t][nt][r] + tn3[nt], 0.f), fmaxf(acc[t][nt][r + 1] + tn3[nt], 0.f));
                }
    }

    // ---- conv4 64->64 on 12x5, pool -> 6x3.  The pooled outputs of the workgroup's four segments go to SHARED planes
    //      S4[72 px][64 ch] (row = 18 * wave + pixel) for the N-split conv5 / conv6.
    NQ_STAMP(6);
    const unsigned S4 = X_BASE;                           // XT planes x X_PS (wave 0/1 regions; their A3 is dead by then)
    const unsigned S5 = X_BASE + 2 * X_WAVE;              // conv5 output, same shape (wave 2/3 regions)
    const int w5b = __builtin_amdgcn_readfirstlane((CNNX_W5 + wave * (18 * XT * 512)) * 2);
    const int w6b = __builtin_amdgcn_readfirstlane((CNNX_W6 + wave * (18 * XT * 512)) * 2);
    f32x4 b5[4][XT], b6[8][XT];
    {
        f32x16 acc[2][2];
        unsigned base[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[t][nt] = zero16();
            base[t] = R + base34[t] * X_RS3 + (h << 4);
        }
        conv_k_terms_ring<XT, 64, 2, 2, 5, X_RS3, X_P3, X_ZADDR, 3, true, true>(acc, wrs, CNNX_W4 * 2, lane16, base, m34, ring34);
        NQ_STAMP(7);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int t = 0; t < XT; ++t) b5[g][t] = wfrag_load(wrs, lane16, w5b + (g * XT + t) * 1024);
        __syncthreads();             // every wave has consumed its A3: the regions may be re-used
        const unsigned wr = S4 + (18 * wave + 9 * hf) * X_RS3 + n * 2;
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
                        for (int x = xwin53_lo(bb); x < xwin53_hi(bb); ++x) {
                            const int u = 10 * gl + 5 * yy + x;
                            mx = fmaxf(mx, acc[u >> 4][nt][u & 15]);
                        }
                    pv[nt] = fmaxf(mx + tn4[nt], 0.f);
                }
                lds_store_terms2<XT>(wr + (gl * 3 + bb) * X_RS3, wr + (gl * 3 + bb) * X_RS3 + 64, X_PS, pv[0], pv[1]);
            }
    }
    __syncthreads();
    NQ_STAMP(8);

    // ---- conv5 / conv6 with N split over the waves: wave w owns output channels 16w..16w+15 of ALL four segments (72 / 24
    //      output rows in 16-row tiles of v_mfma_f32_16x16x32_bf16); fragments [wave][step][term][lane][8]
    {
        const int i16 = lane & 15, kg = lane >> 4;
        const int ch = 16 * wave + i16;                    // D-fragment column = output channel
        f32x4 acc5[5];
        unsigned base5[5], m5[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            acc5[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int rho = 16 * t + i16;
            const int slot = rho / 18, pix = rho - 18 * slot, ry = pix / 3, rx = pix - 3 * ry;
            m5[t] = tap_mask(rho < 72, ry, rx, 6, 3);
            base5[t] = S4 + (slot * 18 + (ry - 1) * 3 + (rx - 1)) * X_RS3 + (kg << 4);
        }
        unsigned a5ad[5][XT];
        f32x4 a5[2][5][XT];                                 // A rows one step ahead: [buffer][tile][term]
        auto load_a5 = [&](int g) {
            const int tap = g >> 1, s = g & 1;
            const int tapoff = ((tap / 3) * 3 + tap % 3) * X_RS3;
            if (s == 0) {
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const bool ok = (m5[t] >> tap) & 1u;
#pragma unroll
                    for (int q = 0; q < XT; ++q) a5ad[t][q] = ok ? base5[t] + q * X_PS : X_ZADDR - tapoff;
                }
            }
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int q = 0; q < XT; ++q) a5[g & 1][t][q] = lds_ld128(a5ad[t][q] + tapoff + 64 * s);
        };
        load_a5(0);
#pragma unroll
        for (int g = 0; g < 18; ++g) {
            if (g + 3 < 18) {
#pragma unroll
                for (int q = 0; q < XT; ++q) b5[(g + 3) & 3][q] = wfrag_load(wrs, lane16, w5b + ((g + 3) * XT + q) * 1024);
            }
            if (g + 1 < 18) load_a5(g + 1);
            __builtin_amdgcn_sched_barrier(0);             // requests stay ahead of the step's MFMAs (conv_k_terms: FENCE)
            mma16_terms<5>(acc5, a5[g & 1], b5[g & 3]);
        }
        NQ_STAMP(9);
#pragma unroll
        for (int g = 0; g < 7; ++g)
#pragma unroll
            for (int q = 0; q < XT; ++q) b6[g][q] = wfrag_load(wrs, lane16, w6b + (g * XT + q) * 1024);
        {
            const unsigned wr = S5 + (4 * kg) * X_RS3 + ch * 2;
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int r = 0; r < 4; r += 2)
                    if (t < 4 || kg < 2)                       // rho = 16 t + 4 kg + r < 72
                        lds_store_terms2<XT>(wr + (16 * t + r) * X_RS3, wr + (16 * t + r + 1) * X_RS3, X_PS,
                                             fmaxf(acc5[t][r] + tn5, 0.f), fmaxf(acc5[t][r + 1] + tn5, 0.f));
        }
        __syncthreads();
        NQ_STAMP(10);

        // conv6 (3 x 3 kernel, padding (1,0)) = padding-1 conv at the centre column: rows (slot, y), 24 of 32
        f32x4 acc6[2], acc6b[2];            // even / odd K-steps accumulate separately
        unsigned base6[2], m6[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            acc6[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc6b[t] = acc6[t];
            const int rho = 16 * t + i16;
            const int slot = rho / 6, y = rho - 6 * slot;
            m6[t] = tap_mask(rho < 24, y, 1, 6, 3);          // output column x = 1: input columns 0..2 are all inside
            base6[t] = S5 + (slot * 18 + (y - 1) * 3) * X_RS3 + (kg << 4);
        }
        unsigned a6ad[2][XT];
        f32x4 a6[2][2][XT];
        auto load_a6 = [&](int g) {
            const int tap = g >> 1, s = g & 1;
            const int tapoff = ((tap / 3) * 3 + tap % 3) * X_RS3;
            if (s == 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bool ok = (m6[t] >> tap) & 1u;
#pragma unroll
                    for (int q = 0; q < XT; ++q) a6ad[t][q] = ok ? base6[t] + q * X_PS : X_ZADDR - tapoff;
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < XT; ++q) a6[g & 1][t][q] = lds_ld128(a6ad[t][q] + tapoff + 64 * s);
        };
        load_a6(0);
#pragma unroll
        for (int g = 0; g < 18; ++g) {
            if (g + 7 < 18) {
#pragma unroll
                for (int q = 0; q < XT; ++q) b6[(g + 7) & 7][q] = wfrag_load(wrs, lane16, w6b + ((g + 7) * XT + q) * 1024);
            }
            if (g + 1 < 18) load_a6(g + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (g & 1) mma16_terms<2>(acc6b, a6[1], b6[g & 7]);
            else mma16_terms<2>(acc6, a6[0], b6[g & 7]);
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
                if (rho < 24) lds_st32(fo + (slot * 96 + i16 * 6 + y) * 4, __float_as_uint(fmaxf(acc6[t][r] + acc6b[t][r] + tn6, 0.f)));
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
    NQ_STAMP_END(g_phase_clk6, ((blockIdx.y * gridDim.x + blockIdx.x) << 2) + wave);
}

template <bool SEGX>
static int x6_launch(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off, const int32_t* n_wins,
                     const float* clip_floor, const uint32_t* clip_max_enc, float top_db, int32_t n_clips,
                     int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w, const uint16_t* cnn_wx, float* feat,
                     void* stream, const float* seg_x = nullptr, int32_t seg_L = 0) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31) || seg_hop <= 0 || !cnn_wx || !feat ||
        (!SEGX && !clip_floor && !clip_max_enc) || (SEGX && (!seg_x || seg_L <= 0)))
        return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    // 117 KB of dynamic LDS is above the 64 KB default: opted in once per device ordinal (a process may drive several GPUs)
    static std::atomic<bool> lds_ok[64];                  // (one array per SEGX instantiation)
    if (nq_lds_opt_in((const void*)cnn_front_bf16x6_kernel<SEGX>, (int)X_LDS, lds_ok)) return 2;
    hipLaunchKernelGGL(cnn_front_bf16x6_kernel<SEGX>, dim3(total_tok_padded / 4), dim3(256), X_LDS, (hipStream_t)stream, mel_tm,
                       frame_off, tok_off, n_wins, clip_floor, n_clips, seg_hop, cnn_w, cnn_wx, feat, clip_max_enc, top_db, seg_x, seg_L);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_cnn_adapt_bf16x6(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                                      const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                                      int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w,
                                      const uint16_t* cnn_wx, float* feat, void* stream) {
    return x6_launch<false>(mel_tm, frame_off, tok_off, n_wins, clip_floor, nullptr, 0.f, n_clips, total_tok_padded, seg_hop, cnn_w,
                            cnn_wx, feat, stream);
}

// nisqa_cnn_adapt_bf16x6 with the per-clip floor derived in the kernel from the mel kernel's clip_max_enc (internal.hpp)
int nq_cnn_adapt_bf16x6_from_max(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                                 const int32_t* n_wins, const uint32_t* clip_max_enc, float top_db, int32_t n_clips,
                                 int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w, const uint16_t* cnn_wx,
                                 float* feat, void* stream) {
    if (!clip_max_enc) return NISQA_ERR_ARG;
    return x6_launch<false>(mel_tm, frame_off, tok_off, n_wins, nullptr, clip_max_enc, top_db, n_clips, total_tok_padded, seg_hop,
                            cnn_w, cnn_wx, feat, stream);
}

// segment-tensor input mode (nisqa_cnn_adapt_segments_bf16's contract)
extern "C" int nisqa_cnn_adapt_segments_bf16x6(const float* x, int32_t seg_len_padded, const int32_t* tok_off,
                                               const int32_t* n_wins, int32_t n_clips, int32_t total_tok_padded,
                                               const float* cnn_w, const uint16_t* cnn_wx, float* feat, void* stream) {
    return x6_launch<true>(nullptr, nullptr, tok_off, n_wins, nullptr, nullptr, 0.f, n_clips, total_tok_padded, 1, cnn_w, cnn_wx, feat,
                           stream, x, seg_len_padded);
}
