// ONE source for the four 16-bit operand formats of this kernel (conv_bf16.hpp): 'bf16x3' (two bf16 terms, three products), 'f16x3' /
// 'f16x4' (two f16 terms of the scaled tensors, three / four products) and 'bf16x6' (three bf16 terms -- an exact split -- and six
// products: three planes per tensor, 23.7 KB of LDS per wave, ONE workgroup per CU, the K loops of conv_k_terms with their scheduling
// fence; until round 5 a separate file, cnn_std_bf16x6.hip).  The text below was written for the first form.
//
// StandardCNN (fixed 2x2 max-pools) + fc_out on split-bf16 MFMA ("bf16x3") for the nisqa_tts.tar
// architecture -- same role, inputs and outputs as cnn_std_front_kernel + cnn_std_back_kernel in cnn_std.hip
// (reference nisqa/NISQA_lib.py:2239-2282, 487-502, 811-836), built from the blocks of conv_bf16.hpp exactly
// like cnn_front_bf16_kernel (cnn_bf16.hip); only the geometry differs:
//   48x15 -conv1-> pool 2x2 pad (0,1) -> 24x8 -conv2-> pool -> 12x4 -conv3,conv4-> pool -> 6x2 -conv5,conv6->
//   64 x 6 x 2 = 768 -fc_out-> 20.
// conv2 fills its 32-row tiles completely (a pooled row of a lane half is one 16-row half tile), conv3/4 use 24
// of 32 rows; conv5/conv6 batch the workgroup's four segments (48 rows = three full 16-row tiles per wave, N
// split over the waves).  conv6 leaves its output in LDS as fp32 and fc_out (768 -> 20, 0.2 % of the FLOPs)
// runs on the VALU, one wave per segment.
#include "common.hpp"
#include "layout.hpp"
#include "conv_bf16.hpp"
#include "../../include/nisqa_hip.h"

#define SS_A1PLANE 6144                    /* 192 px x 16 ch bf16 */
#define SS_PPLANE 1700
/* the per-wave region depends on the planes per tensor (T = 2, or 3 for 'bf16x6'): A1 / A2 / A3 planes, then the conv1 input patch (up to
   three zero-bordered planes [17][50], 5 120 B), then 128 B of zeros (the zero block of wave 3 serves conv5 / conv6) */
#define SS_PATCH_T(T) ((T) * SS_A1PLANE)
#define SS_ZERO_T(T) (SS_PATCH_T(T) + 5120)
#define SS_WAVE_T(T) (SS_ZERO_T(T) + 128)  /* 17 536 (T = 2) / 23 680 (T = 3) */
#define SS_LDS_T(T) (SS_BASE + 4 * SS_WAVE_T(T))   /* 72 320 B -> two workgroups per CU / 96 896 B -> one */
#define SS_ZADDR 2048u                     /* a zero block shared by the workgroup, above the largest tap offset (conv_k_bf16) */
#define SS_BASE 2176u                      /* first wave region */
#define SS_PLANE 6144                      /* S4 / S5: 48 rows x 64 ch bf16, chunk-swizzled */
/* conv2 -> conv3 -> conv4 activations: pixel rows padded by 16 bytes, NOT swizzled (every address = lane base + immediate) */
#define SS_RS2 80                          /* A2: 48 px x 32 ch */
#define SS_P2 (48 * SS_RS2)
#define SS_RS3 144                         /* A3: 48 px x 64 ch */
#define SS_P3 (48 * SS_RS3)
static_assert(2 * SS_P3 <= SS_ZERO_T(2) && 2 * SS_LDS_T(2) <= 160 * 1024, "LDS plan, two planes per tensor");
static_assert(3 * SS_P3 <= SS_ZERO_T(3) && 3 * SS_PLANE <= SS_WAVE_T(3) && SS_LDS_T(3) <= 160 * 1024 && 3 * SS_PPLANE <= 5120, "LDS plan, three planes");

// sum over the 16 lanes of a DPP row, result in every lane of the row
NQ_DEV float row16_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// conv2's K loop, second generation (round 4; before: the generic-pointer conv3x3_bf16 with per-tap bounds / swizzle / row
// arithmetic on every tile).  Its input planes are the chunk-swizzled 16-channel A1 planes conv1 writes (32 bytes per pixel,
// 16-byte chunk (c >> 3) ^ (image row & 1) at W = 8), so a tap's address is one of TWO lane-static bases -- the centre
// pixel's chunk for taps in its own image row, the flipped chunk for the rows above and below -- plus a compile-time tap
// offset; validity is a lane-static 9-bit mask, out-of-image taps read the workgroup's zero block, fragments come through
// the buffer descriptor.  One K-step per tap (16 channels), A rows one tap ahead, B fragments two.
template <int MT, int FMT = NQ_FMT_BF16X3>
NQ_DEV void conv_k_bf16_c16(f32x16 (&acc)[MT][1], __amdgpu_buffer_rsrc_t rsrc, int wbyte, unsigned lane16,
                            const unsigned (&a_same)[MT], const unsigned (&a_flip)[MT], const unsigned (&m9)[MT]) {
    f32x4 bh[3][1], bl[3][1], ah[2][MT], al[2][MT];
    auto load_b = [&](int g, int slot) {
        bh[slot][0] = wfrag_load(rsrc, lane16, wbyte + (g * 2 + 0) * 1024);
        bl[slot][0] = wfrag_load(rsrc, lane16, wbyte + (g * 2 + 1) * 1024);
    };
    auto load_a = [&](int g, int slot) {
        const int dy = g / 3, dx = g % 3;
        const int tapoff = ((dy - 1) * 8 + (dx - 1)) * 32;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const bool ok = (m9[t] >> g) & 1u;
            const unsigned a = (unsigned)((int)(dy == 1 ? a_same[t] : a_flip[t]) + tapoff);
            ah[slot][t] = lds_ld128(ok ? a : SS_ZADDR);
            al[slot][t] = lds_ld128(ok ? a + SS_A1PLANE : SS_ZADDR);
        }
    };
    load_b(0, 0);
    load_b(1, 1);
    load_a(0, 0);
#pragma unroll
    for (int g = 0; g < 9; ++g) {
        if (g + 2 < 9) load_b(g + 2, (g + 2) % 3);
        if (g + 1 < 9) load_a(g + 1, (g + 1) & 1);
        const int sa = g & 1, sb = g % 3;
        mma_pair_fmt<FMT, MT, 1>(acc, ah[sa], al[sa], bh[sb], bl[sb]);
    }
}

// conv2's K loop with T terms per operand and the products of mma_terms (T = 3: six), one wave per SIMD: A rows one tap ahead, fragments
// two, a scheduling fence behind each step's requests (conv_k_terms)
template <int MT, int T>
NQ_DEV void conv_k_terms_c16(f32x16 (&acc)[MT][1], __amdgpu_buffer_rsrc_t rsrc, int wbyte, unsigned lane16,
                             const unsigned (&a_same)[MT], const unsigned (&a_flip)[MT], const unsigned (&m9)[MT]) {
    f32x4 b[3][1][T], a[2][MT][T];
    auto load_b = [&](int g, int slot) {
#pragma unroll
        for (int t = 0; t < T; ++t) b[slot][0][t] = wfrag_load(rsrc, lane16, wbyte + (g * T + t) * 1024);
    };
    auto load_a = [&](int g, int slot) {
        const int dy = g / 3, dx = g % 3;
        const int tapoff = ((dy - 1) * 8 + (dx - 1)) * 32;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bool ok = (m9[m] >> g) & 1u;
            const unsigned ad = (unsigned)((int)(dy == 1 ? a_same[m] : a_flip[m]) + tapoff);
#pragma unroll
            for (int t = 0; t < T; ++t) a[slot][m][t] = lds_ld128(ok ? ad + t * SS_A1PLANE : SS_ZADDR);
        }
    };
    load_b(0, 0);
    load_b(1, 1);
    load_a(0, 0);
#pragma unroll
    for (int g = 0; g < 9; ++g) {
        if (g + 2 < 9) load_b(g + 2, (g + 2) % 3);
        if (g + 1 < 9) load_a(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_terms<T, MT, 1>(acc, a[g & 1], b[g % 3]);
    }
}

// the per-row scale tables of the f16 formats' conv5 / conv6 epilogues (the LDS below the zero block is otherwise unused)
#define SS_TAB5 0u                         /* [48 rows] {2^(e5 - e4 - kw5), 2^e5} of the row's segment */
#define SS_TAB6 1024u                      /* [48 rows] 2^-(e5 + kw6) */
NQ_DEV f32x2_t ss_ld64(unsigned a) { return *(NQ_AS3 const f32x2_t*)(a); }
template <int FMT>
NQ_DEV float ss_epi(float v, float c, float t) { return (FMT == NQ_FMT_BF16X3 || FMT == NQ_FMT_BF16X6) ? fmaxf(v + t, 0.f) : fmaxf(fmaf(v, c, t), 0.f); }

// FMT (conv_bf16.hpp): NQ_FMT_BF16X3 -- bf16 hi + lo, three products (the input keeps a third term) -- or the f16 formats: every
// tensor as f16 hi + lo of y * 2^e, e from the measured maximum of the layer's input and the layer's weight norm (cnn_bf16.hip's
// scheme: |y| <= m_in * G + T), three or four products; wb is then the CNNH_ blob (weights.pack_adapt_cnn_f16)
template <int FMT>
NQ_DEV void cnn_std_split_body(
    const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off,
    const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
    const float* __restrict__ clip_floor, int n_clips, int seg_hop,
    const float* __restrict__ cw, const unsigned short* __restrict__ wb, float* __restrict__ feat20) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool X6 = FMT == NQ_FMT_BF16X6;
    constexpr bool F16 = FMT == NQ_FMT_F16X3 || FMT == NQ_FMT_F16X4;
    constexpr int NT_ = X6 ? 3 : 2;                                         // planes per activation tensor / terms per weight fragment
    constexpr unsigned SS_PATCH = SS_PATCH_T(NT_), SS_ZERO = SS_ZERO_T(NT_), SS_WAVE = SS_WAVE_T(NT_);
    // fragment blob of the term count: CNNB_ (two terms, also the CNNH_ blob of the f16 formats) or CNNX_ (three)
    constexpr int W1_ = X6 ? CNNX_W1 : CNNB_W1, W2_ = X6 ? CNNX_W2 : CNNB_W2, W3_ = X6 ? CNNX_W3 : CNNB_W3, W4_ = X6 ? CNNX_W4 : CNNB_W4,
                  W5_ = X6 ? CNNX_W5 : CNNB_W5, W6_ = X6 ? CNNX_W6 : CNNB_W6, WU16_ = X6 ? CNNX_U16S : CNNB_U16S;
    const int* __restrict__ meta_i = (const int*)(wb + CNNH_META);         // F16: kw[l], G[l] at +8, T[l] at +16 (l = layer - 1)
    const float* __restrict__ meta_f = (const float*)(wb + CNNH_META);
    float dummy_mx = 0.f;
    int e_in = 0;                                                           // F16: scale exponent / largest magnitude of the tensor
    float m_in = 0.f;                                                       //      the next layer reads
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p0 = blockIdx.x * 4;                      // tok_off is a multiple of 32: no clip straddling
    const int b = __builtin_amdgcn_readfirstlane(find_segment_wave(tok_off, n_clips, p0, lane));   // one vector load + ballot, not a chain of log2(n) scalar loads
    const int k0 = p0 - tok_off[b];
    const int nvalid = min(4, n_wins[b] - k0);
    if (nvalid <= 0) return;
    const bool valid = wave < nvalid;                    // padding waves still walk the barriers (on zeros)
    const int k = k0 + wave;
    char* act = smem + SS_BASE + wave * SS_WAVE;
    char* zero = act + SS_ZERO;

    // ---- the 15-frame window as three zero-bordered bf16 planes (hi, mid, lo) [frame j + 1][mel m + 1]; all 12 global loads
    //      of the window are requested up front (one memory latency), the plane addresses are three lane-dependent bases +
    //      immediates (element i0 = lane + 64 q = (frame, mel) = divmod(i0, 48); q = 3 t + u: frame q + t + (lane + 16 u) / 48)
    const unsigned R = SS_BASE + (unsigned)wave * SS_WAVE; // this wave's region (the kernel has no static LDS: addresses start at 0)
    {
        const float fl = clip_floor[b];
        const float* src = mel_tm + (size_t)(frame_off[b] + k * seg_hop) * 48;
        float vraw[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) vraw[q] = (valid && (q < 11 || lane < 16)) ? src[lane + 64 * q] : 0.f;
        const unsigned pb = R + SS_PATCH;
#pragma unroll
        for (int it = 0; it < 5; ++it)                      // 319 x 16 bytes of zeros (three planes)
            if (it < 4 || lane < (3 * SS_PPLANE + 15) / 16 - 256) lds_st128(pb + (lane + 64 * it) * 16, f32x4{0.f, 0.f, 0.f, 0.f});
        if (lane < 32) { ((float*)zero)[lane] = 0.f; ((unsigned*)(smem + SS_ZADDR))[lane] = 0u; }   // (through the dynamic-LDS symbol: see cnn_bf16.hip)
        __builtin_amdgcn_wave_barrier();
        unsigned ob[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int e = lane + 16 * u, j0 = e >= 48 ? 1 : 0, m = e - 48 * j0;
            ob[u] = pb + ((j0 + 1) * 50 + m + 1) * 2;
        }
        float s0 = 1.f;
        if (F16) {                                          // the window's own largest magnitude fixes its scale
            float mr = 0.f;
#pragma unroll
            for (int q = 0; q < 12; ++q) mr = fmaxf(mr, valid ? __builtin_fabsf(fmaxf(vraw[q], fl)) : 0.f);
            m_in = wave_max_nonneg(mr);
            e_in = f16_scale_exp(m_in);
            s0 = pow2_f32(e_in);
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const float v = valid ? fmaxf(vraw[q], fl) : 0.f;
            const unsigned a = ob[q % 3] + (q + q / 3) * 100;
            if (F16) {
                if (q < 11 || lane < 16) lds_store_one_fmt<FMT>(a, SS_PPLANE, v * s0, dummy_mx);
            } else {
                const unsigned hi = cvt_pk_bf16(v, 0.f);
                const float r1 = v - __uint_as_float(hi << 16);
                const unsigned mid = cvt_pk_bf16(r1, 0.f);
                const unsigned lo = cvt_pk_bf16(r1 - __uint_as_float(mid << 16), 0.f);
                if (q < 11 || lane < 16) {
                    lds_st16(a, hi);
                    lds_st16(a + SS_PPLANE, mid);
                    lds_st16(a + 2 * SS_PPLANE, lo);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    const int i = lane & 31, hfi = (i >> 2) & 1, qi = (i & 3) + 4 * (i >> 3);
    const int n = lane & 31, hf = lane >> 5, h = lane >> 5;

    // ---- conv1 1->16 on the matrix pipe + MaxPool2d(2, stride 2, padding (0,1)): 48x15 -> 24x8, TWO output pixels per MFMA
    //      row like cnn_front_bf16_kernel (round 3; before: one pixel per row, k-slots gathered with 48 16-bit LDS reads and 24
    //      shift-ors per tile pair).  A row = the mel pair (m0, m0 + 1) of one frame; N = 32 = (channel c, pair member dm);
    //      K = 12 of 16 = (frame tap kx, the four mels m0 - 1 .. m0 + 2 the pair touches): contiguous in the bordered patch, so
    //      a lane's 8 k-slots are two pairs of dwords per plane.  B[k][n] = w[c][dmm - dm][kx] (weights.py, conv1_pairs).
    //      The input keeps THREE bf16 terms (six products): the BiLSTM head amplifies input error ten times more than the
    //      attention head.  Pooled column bb covers frames {2 bb - 1, 2 bb} (column -1 is pool padding): in-lane; the mel
    //      pair's maximum needs the partner 16 lanes away: lane group dm = 0 finalises the even pooled pixels of an
    //      iteration, dm = 1 the odd ones (one ds_swizzle per pixel pair); ReLU first, maxima on non-negative floats as uints.
    {
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, WU16_ * 2, 0x00020000);
        f32x4 w1[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) w1[t] = wfrag_load(wrs, lane * 16, (W1_ + t * 512) * 2);
        float tn = cw[CNN_T1 + (n & 15)], c1 = 1.f, ms1 = 0.f;
        int e1 = 0;
        if (F16) {
            e1 = f16_scale_exp(fmaf(m_in, meta_f[8], meta_f[16]));
            c1 = pow2_f32(e1 - e_in - meta_i[0]);
            tn *= pow2_f32(e1);
        }
        const int xq = min(qi, 14);                       // row 15 of a tile is padding (result unused)
        unsigned rd_a = R + SS_PATCH + ((xq + (h ? 2 : 0)) * 50 + 24 * hfi) * 2;
        unsigned rd_b = R + SS_PATCH + ((xq + (h ? 2 : 1)) * 50 + 24 * hfi) * 2;
        const bool is_b = (n & 16) != 0;
        const unsigned mb = is_b ? ~0u : 0u;
        // A1 is chunk-swizzled (16-byte chunk (c >> 3) ^ ((pixel >> 3) & 1)); an iteration's 16 pixels start at a multiple of
        // 16, so the swizzle bit of pixel 2 kk (+ 1) is kk >> 2: two lane bases, the rest are immediates
        const int c = n & 15;
        const unsigned wr0 = R + (12 * hf * 8) * 32 + (is_b ? 32 : 0) + (c & 7) * 2;
        const unsigned wrA = wr0 + ((c >> 3) << 4), wrB = wr0 + (((c >> 3) ^ 1) << 4);
        for (int g2 = 0; g2 < 6; ++g2) {
            f32x16 acc[2];
            f32x4 xa[2][3];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int t = 0; t < (F16 ? 2 : 3); ++t) {  // dword reads (ds_read2_b32): the pairs are only 4-byte aligned
                    const unsigned pa = rd_a + 4 * tt + t * SS_PPLANE, pq = rd_b + 4 * tt + t * SS_PPLANE;
                    xa[tt][t] = f32x4{__uint_as_float(lds_ld32(pa)), __uint_as_float(lds_ld32(pa + 4)),
                                      __uint_as_float(lds_ld32(pq)), __uint_as_float(lds_ld32(pq + 4))};
                }
            acc[0] = zero16();
            acc[1] = zero16();
            if (F16) {
                if (FMT == NQ_FMT_F16X4) { acc[0] = mfma32_fmt<FMT>(xa[0][1], w1[1], acc[0]); acc[1] = mfma32_fmt<FMT>(xa[1][1], w1[1], acc[1]); }
                acc[0] = mfma32_fmt<FMT>(xa[0][1], w1[0], acc[0]); acc[1] = mfma32_fmt<FMT>(xa[1][1], w1[0], acc[1]);
                acc[0] = mfma32_fmt<FMT>(xa[0][0], w1[1], acc[0]); acc[1] = mfma32_fmt<FMT>(xa[1][0], w1[1], acc[1]);
                acc[0] = mfma32_fmt<FMT>(xa[0][0], w1[0], acc[0]); acc[1] = mfma32_fmt<FMT>(xa[1][0], w1[0], acc[1]);
            } else {
            acc[0] = mfma_bf(xa[0][2], w1[0], acc[0]); acc[1] = mfma_bf(xa[1][2], w1[0], acc[1]);   // smallest products first
            acc[0] = mfma_bf(xa[0][1], w1[1], acc[0]); acc[1] = mfma_bf(xa[1][1], w1[1], acc[1]);
            acc[0] = mfma_bf(xa[0][0], w1[2], acc[0]); acc[1] = mfma_bf(xa[1][0], w1[2], acc[1]);
            acc[0] = mfma_bf(xa[0][1], w1[0], acc[0]); acc[1] = mfma_bf(xa[1][1], w1[0], acc[1]);
            acc[0] = mfma_bf(xa[0][0], w1[1], acc[0]); acc[1] = mfma_bf(xa[1][0], w1[1], acc[1]);
            acc[0] = mfma_bf(xa[0][0], w1[0], acc[0]); acc[1] = mfma_bf(xa[1][0], w1[0], acc[1]);
            }
            unsigned r[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int tt = v >> 3, bb = v & 7;
                const float mx = bb ? fmaxf(acc[tt][2 * bb - 1], acc[tt][2 * bb]) : acc[tt][0];   // frames
                r[v] = __float_as_uint(ss_epi<FMT>(mx, c1, tn));
            }
            unsigned got[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                got[kk] = (unsigned)__builtin_amdgcn_ds_swizzle((int)((r[2 * kk] & mb) | (r[2 * kk + 1] & ~mb)), 0x401F);
            float fin[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const unsigned own = (r[2 * kk + 1] & mb) | (r[2 * kk] & ~mb);
                fin[kk] = __uint_as_float(max(own, got[kk]));
            }
#pragma unroll
            for (int kk = 0; kk < 8; kk += 2) {             // pixels 16 g2 + 2 kk (+ 1 in the dm = 1 lanes) and two further
                const unsigned w_ = (kk < 4 ? wrA : wrB) + 512 * g2 + 64 * kk;
                lds_store_pair_fmt<FMT>(w_, w_ + 64, SS_A1PLANE, fin[kk], fin[kk + 1], ms1);
            }
            rd_a += 8; rd_b += 8;                          // mel m0 = 2 gl, gl = 12 hfi + 2 g2 + tt
        }
        if (F16) { m_in = wave_max_nonneg(ms1) * pow2_f32(-e1); e_in = e1; }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- conv2 16->32 on 24x8, pool 2x2 -> 12x4: tile t of a lane half = pooled row 6*half + t (2 rows x 8 cols)
    {
        f32x16 acc[6][1];
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[t][0] = zero16();
        unsigned a_same[6], a_flip[6], m2[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int py = 2 * (6 * hfi + t) + (qi >> 3), px = qi & 7;
            m2[t] = tap_mask(true, py, px, 24, 8);
            const unsigned row = R + (unsigned)((py * 8 + px) * 32);
            a_same[t] = row + (unsigned)((h ^ (py & 1)) << 4);
            a_flip[t] = row + (unsigned)((h ^ (~py & 1)) << 4);
        }
        const __amdgpu_buffer_rsrc_t wrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, WU16_ * 2, 0x00020000);
        if constexpr (X6) conv_k_terms_c16<6, 3>(acc, wrs2, W2_ * 2, lane * 16, a_same, a_flip, m2);
        else conv_k_bf16_c16<6, FMT>(acc, wrs2, W2_ * 2, lane * 16, a_same, a_flip, m2);
        float tn = cw[CNN_T2 + n], c2 = 1.f, ms2 = 0.f;
        int e2 = 0;
        if (F16) {
            e2 = f16_scale_exp(fmaf(m_in, meta_f[9], meta_f[17]));
            c2 = pow2_f32(e2 - e_in - meta_i[1]);
            tn *= pow2_f32(e2);
        }
        const unsigned wr = R + (6 * hf * 4) * SS_RS2 + n * 2;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int bb = 0; bb < 4; bb += 2) {             // pooled pixels (6 hf + t) * 4 + bb, + 1: one packed split
                float pvv[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int q = 2 * (bb + e);
                    const float mx = fmaxf(fmaxf(acc[t][0][q], acc[t][0][q + 1]), fmaxf(acc[t][0][8 + q], acc[t][0][8 + q + 1]));
                    pvv[e] = ss_epi<FMT>(mx, c2, tn);
                }
                lds_store_pair_fmt<FMT>(wr + (4 * t + bb) * SS_RS2, wr + (4 * t + bb + 1) * SS_RS2, SS_P2, pvv[0], pvv[1], ms2);
            }
        if (F16) { m_in = wave_max_nonneg(ms2) * pow2_f32(-e2); e_in = e2; }
    }
    __builtin_amdgcn_wave_barrier();

    // conv3 / conv4 on 12x4: a lane half owns 3 pooled rows = 3 groups of 8 pixels; u = 8*gl + 4*yy + x.  Second-generation
    // K loop (conv_bf16.hpp: lane-static tap masks, one select per tap and tile, fragments through a buffer descriptor)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, WU16_ * 2, 0x00020000);
    const unsigned lane16 = lane * 16;
    unsigned base34[2], m34[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int u = 16 * t + qi;
        const int py = 2 * (3 * hfi + (u >> 3)) + ((u >> 2) & 1), px = u & 3;
        m34[t] = tap_mask(u < 24, py, px, 12, 4);
        base34[t] = (py - 1) * 4 + (px - 1);              // pixel index of tap (-1, -1)
    }
    {
        f32x16 acc[2][2];
        unsigned base[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[t][nt] = zero16();
            base[t] = R + base34[t] * SS_RS2 + (h << 4);
        }
        if constexpr (X6) conv_k_terms<3, 32, 2, 2, 4, SS_RS2, SS_P2, SS_ZADDR, 3>(acc, wrs, W3_ * 2, lane16, base, m34);
        else conv_k_bf16<32, 2, 2, 4, SS_RS2, SS_P2, SS_ZADDR, true, 3, FMT>(acc, wrs, W3_ * 2, lane16, base, m34);
        float c3 = 1.f, s3 = 1.f, ms3 = 0.f;
        int e3 = 0;
        if (F16) {
            e3 = f16_scale_exp(fmaf(m_in, meta_f[10], meta_f[18]));
            c3 = pow2_f32(e3 - e_in - meta_i[2]);
            s3 = pow2_f32(e3);
        }
        const unsigned wr = R + (24 * hf) * SS_RS3 + n * 2; // pixel (2 (3 hf + (u >> 3)) + ((u >> 2) & 1)) * 4 + (u & 3) = 24 hf + u
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float tn = F16 ? cw[CNN_T3 + n + 32 * nt] * s3 : cw[CNN_T3 + n + 32 * nt];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int u = 16 * t + r;
                    if (u < 24)
                        lds_store_pair_fmt<FMT>(wr + u * SS_RS3 + 64 * nt, wr + (u + 1) * SS_RS3 + 64 * nt, SS_P3,
                                                ss_epi<FMT>(acc[t][nt][r], c3, tn), ss_epi<FMT>(acc[t][nt][r + 1], c3, tn), ms3);
                }
        }
        if (F16) { m_in = wave_max_nonneg(ms3) * pow2_f32(-e3); e_in = e3; }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- conv4 64->64 on 12x4, pool -> 6x2.  The pooled outputs of the four segments go to a SHARED pair of
    //      bf16 planes S4[48 rows][64 ch] (row = 12 * wave + pixel) for the N-split conv5 / conv6.
    char* s4 = smem + SS_BASE;             // wave 0's region (its A3 is dead by then)
    char* s5 = smem + SS_BASE + SS_WAVE;   // wave 1's region
    float* s6 = (float*)(smem + SS_BASE);  // conv6 output, fp32 [48][64], over S4
    {
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[t][nt] = zero16();
        unsigned base[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) base[t] = R + base34[t] * SS_RS3 + (h << 4);
        if constexpr (X6) conv_k_terms<3, 64, 2, 2, 4, SS_RS3, SS_P3, SS_ZADDR, 3>(acc, wrs, W4_ * 2, lane16, base, m34);
        else conv_k_bf16<64, 2, 2, 4, SS_RS3, SS_P3, SS_ZADDR, true, 3, FMT>(acc, wrs, W4_ * 2, lane16, base, m34);
        float c4 = 1.f, s4s = 1.f, ms4 = 0.f;
        int e4 = 0;
        if (F16) {
            e4 = f16_scale_exp(fmaf(m_in, meta_f[11], meta_f[19]));
            c4 = pow2_f32(e4 - e_in - meta_i[3]);
            s4s = pow2_f32(e4);
        }
        __syncthreads();                   // every wave has consumed its A3: the regions may be re-used
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = n + 32 * nt;
            const float tn = F16 ? cw[CNN_T4 + c] * s4s : cw[CNN_T4 + c];
#pragma unroll
            for (int gl = 0; gl < 3; ++gl)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    float mx = -3.0e38f;
#pragma unroll
                    for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                        for (int xx = 0; xx < 2; ++xx) {
                            const int u = 8 * gl + 4 * yy + 2 * bb + xx;
                            mx = fmaxf(mx, acc[u >> 4][nt][u & 15]);
                        }
                    const int pp = 12 * wave + (3 * hf + gl) * 2 + bb;
                    const int off4 = pp * 128 + (((c >> 3) ^ ((pp >> 1) & 7)) << 4) + (c & 7) * 2;
                    if (F16 || X6) lds_store_one_fmt<FMT>(SS_BASE + off4, SS_PLANE, ss_epi<FMT>(mx, c4, tn), ms4);
                    else store_split(s4, SS_PLANE, off4, fmaxf(mx + tn, 0.f));
                }
        }
        if (F16) {
            // conv5 / conv6 run over the four segments' rows at once: every row carries its segment's scales (cnn_bf16.hip)
            const float m4 = wave_max_nonneg(ms4) * pow2_f32(-e4);
            const int e5 = f16_scale_exp(fmaf(m4, meta_f[12], meta_f[20]));
            if (lane < 12) {
                lds_st32(SS_TAB5 + (12 * wave + lane) * 8, __float_as_uint(pow2_f32(e5 - e4 - meta_i[4])));
                lds_st32(SS_TAB5 + (12 * wave + lane) * 8 + 4, __float_as_uint(pow2_f32(e5)));
                lds_st32(SS_TAB6 + (12 * wave + lane) * 4, __float_as_uint(pow2_f32(-(e5 + meta_i[5]))));
            }
        }
    }
    __syncthreads();

    // fc_out (768 -> 20) is split over the waves like conv5 / conv6: wave w computes outputs 5 w .. 5 w + 4 of all four segments
    // (a quarter of the weights per wave: 60 floats per lane instead of 240), and those 60 floats are requested NOW, so that
    // they arrive under conv5 / conv6 (before: twelve exposed L2 round trips behind the last barrier)
    struct __attribute__((packed, aligned(4))) fc5_t { float v[5]; };
    fc5_t wfc_pre[12];
    __builtin_amdgcn_sched_barrier(0);                     // not earlier: conv2..conv4 need the registers
#pragma unroll
    for (int m = 0; m < 12; ++m) wfc_pre[m] = *(const fc5_t*)(cw + CNNS_FC_W + (size_t)(m * 64 + lane) * 20 + 5 * wave);
    // ---- conv5 / conv6 (3x3, padding 1, on 6x2) with N split over the waves: wave w owns output channels
    //      16w..16w+15 of all four segments; rows rho = 16 t + i16 <-> (slot = rho / 12, pixel = rho % 12)
    {
        const int i16 = lane & 15, kg = lane >> 4;
        const int ch = 16 * wave + i16;
        const char* zero3 = smem + SS_BASE + 3 * SS_WAVE + SS_ZERO;  // wave 3's zero block: S4 / S5 / S6 never cover it
        int ry[3], rx[3], rb[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int rho = 16 * t + i16;
            const int slot = rho / 12, pix = rho - 12 * slot;
            ry[t] = pix >> 1;
            rx[t] = pix & 1;
            rb[t] = slot * 12;
        }
        // the 9 x 3 tap addresses of a lane are static (row -> pixel map, image bounds, chunk swizzle): computed once for
        // both layers as offsets into the S4 / S5 planes (round 3; before: per tap, tile, step and layer); K-step s = 1 is
        // the address ^ 64 (the chunk index is (4 s + kg) ^ swizzle and the zero block is 128-byte aligned)
        unsigned poff[9][3];
        unsigned pok = 0;                                   // bit 3 tap + t: the tap is inside the image
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int y = ry[t] + tap / 3 - 1, x = rx[t] + tap % 3 - 1;
                const bool ok = (unsigned)y < 6u && (unsigned)x < 2u;
                const int pix = rb[t] + y * 2 + x;
                poff[tap][t] = (unsigned)(pix * 128 + ((kg ^ ((pix >> 1) & 7)) << 4));
                pok |= ok ? (1u << (3 * tap + t)) : 0u;
            }
        const unsigned Z3 = SS_BASE + 3 * SS_WAVE + SS_ZERO;
        const __amdgpu_buffer_rsrc_t wrs5 = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, WU16_ * 2, 0x00020000);
#pragma unroll 1
        for (int layer = 0; layer < 2; ++layer) {
            const unsigned srcA = layer ? SS_BASE + SS_WAVE : SS_BASE;
            f32x4 acc5[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) acc5[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int wbyte = ((layer ? W6_ : W5_) + wave * (18 * NT_ * 512)) * 2;
            f32x4 bq[4][NT_], aq[2][3][NT_];               // fragments three K-steps ahead, A rows one step ahead
            auto load_a = [&](int g) {
                const int tap = g >> 1, sx = (g & 1) << 6;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const bool ok = (pok >> (3 * tap + t)) & 1u;
                    const unsigned ah_ = ok ? (srcA + poff[tap][t]) ^ sx : Z3;
#pragma unroll
                    for (int q = 0; q < NT_; ++q) aq[g & 1][t][q] = lds_ld128(ok ? ah_ + q * SS_PLANE : Z3);
                }
            };
            auto load_bq = [&](int g) {
#pragma unroll
                for (int q = 0; q < NT_; ++q) bq[g & 3][q] = wfrag_load(wrs5, lane * 16, wbyte + (g * NT_ + q) * 1024);
            };
#pragma unroll
            for (int g = 0; g < 3; ++g) load_bq(g);
            load_a(0);
#pragma unroll
            for (int g = 0; g < 18; ++g) {
                if (g + 3 < 18) load_bq(g + 3);
                if (g + 1 < 18) load_a(g + 1);
                if constexpr (X6) {
                    __builtin_amdgcn_sched_barrier(0);             // requests stay ahead of the step's MFMAs (one wave per SIMD)
#pragma unroll
                    for (int order = 2; order >= 0; --order)      // the six products, smallest first
#pragma unroll
                        for (int i_ = order; i_ >= 0; --i_)
#pragma unroll
                            for (int t = 0; t < 3; ++t) acc5[t] = mfma_bf16x16(aq[g & 1][t][i_], bq[g & 3][order - i_], acc5[t]);
                } else {
                    mma16_pair_fmt<FMT, 3>(acc5, aq[g & 1], bq[g & 3]);
                }
            }
            const float tn = cw[(layer ? CNN_T6 : CNN_T5) + ch];
            // conv6's fp32 output goes over S4, which every wave finished reading before the barrier that ended conv5
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rho = 16 * t + 4 * kg + r;
                    const int off5 = rho * 128 + (((ch >> 3) ^ ((rho >> 1) & 7)) << 4) + (ch & 7) * 2;
                    if (F16) {
                        // conv5: {2^(e5 - e4 - kw5), 2^e5} of the row's segment; conv6 (fp32 output): 2^-(e5 + kw6)
                        if (layer) s6[rho * 64 + ch] = ss_epi<FMT>(acc5[t][r], __uint_as_float(lds_ld32(SS_TAB6 + (16 * t + r) * 4 + kg * 16)), tn);
                        else {
                            const f32x2_t cs = ss_ld64(SS_TAB5 + (16 * t + r) * 8 + kg * 32);
                            lds_store_one_fmt<FMT>(SS_BASE + SS_WAVE + off5, SS_PLANE, ss_epi<FMT>(acc5[t][r], cs[0], tn * cs[1]), dummy_mx);
                        }
                    } else {
                        const float v = fmaxf(acc5[t][r] + tn, 0.f);
                        if (layer) s6[rho * 64 + ch] = v;
                        else if (X6) lds_store_terms<3>(SS_BASE + SS_WAVE + off5, SS_PLANE, v);
                        else store_split(s5, SS_PLANE, off5, v);
                    }
                }
            __syncthreads();
        }
    }

    // ---- fc_out 768 -> 20, N split over the waves: lane takes k' = lane + 64 m (k' = pixel * 64 + c) of ALL four segments
    //      (48 LDS reads of conv6's fp32 output) against its wave's five output columns (wfc_pre).  The 20 sums over the
    //      wave (4 segments x 5 outputs) are a reduce-scatter -- 10 + 5 exchanges, then four DPP steps on 5 values -- instead
    //      of 20 six-step butterflies; row r of the wave ends up holding segment r.
    {
        float o[4][5];
#pragma unroll
        for (int sg = 0; sg < 4; ++sg)
#pragma unroll
            for (int j = 0; j < 5; ++j) o[sg][j] = 0.f;
#pragma unroll
        for (int m = 0; m < 12; ++m)
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                const float a = s6[(12 * sg + m) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 5; ++j) o[sg][j] = fmaf(a, wfc_pre[m].v[j], o[sg][j]);
            }
        // lanes l, l ^ 32: the lower half keeps segments 0, 1, the upper half segments 2, 3
        float p10[2][5];
        const bool up = lane >= 32;
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float give = up ? o[sg][j] : o[2 + sg][j], keep = up ? o[2 + sg][j] : o[sg][j];
                p10[sg][j] = keep + __shfl_xor(give, 32);
            }
        // lanes l, l ^ 16: bit 4 clear keeps the first segment of its two, bit 4 set the second
        float p5[5];
        const bool hi16 = (lane & 16) != 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float give = hi16 ? p10[0][j] : p10[1][j], keep = hi16 ? p10[1][j] : p10[0][j];
            p5[j] = keep + __shfl_xor(give, 16);
        }
        float mine = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float t = row16_sum_dpp(p5[j]);
            if ((lane & 15) == j) mine = t;
        }
        const int sg = lane >> 4, jo = 5 * wave + (lane & 15);
        if ((lane & 15) < 5 && sg < nvalid) feat20[(size_t)(p0 + sg) * 20 + jo] = mine + cw[CNNS_FC_B + jo];
    }
}

__global__ __launch_bounds__(256, 2) void cnn_std_bf16_kernel(
    const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off, const int32_t* __restrict__ tok_off,
    const int32_t* __restrict__ n_wins, const float* __restrict__ clip_floor, int n_clips, int seg_hop,
    const float* __restrict__ cw, const unsigned short* __restrict__ wb, float* __restrict__ feat20) {
    cnn_std_split_body<NQ_FMT_BF16X3>(mel_tm, frame_off, tok_off, n_wins, clip_floor, n_clips, seg_hop, cw, wb, feat20);
}
// three exact bf16 terms per operand, six products: three planes per tensor, one workgroup per CU
__global__ __launch_bounds__(256, 1) void cnn_std_bf16x6_kernel(
    const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off, const int32_t* __restrict__ tok_off,
    const int32_t* __restrict__ n_wins, const float* __restrict__ clip_floor, int n_clips, int seg_hop,
    const float* __restrict__ cw, const unsigned short* __restrict__ wb, float* __restrict__ feat20) {
    cnn_std_split_body<NQ_FMT_BF16X6>(mel_tm, frame_off, tok_off, n_wins, clip_floor, n_clips, seg_hop, cw, wb, feat20);
}
// fp32 operands as two f16 terms of the power-of-two-scaled tensors; P4: all four term products ('f16x4'), else three ('f16x3')
template <bool P4>
__global__ __launch_bounds__(256, 2) void cnn_std_f16_kernel(
    const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off, const int32_t* __restrict__ tok_off,
    const int32_t* __restrict__ n_wins, const float* __restrict__ clip_floor, int n_clips, int seg_hop,
    const float* __restrict__ cw, const unsigned short* __restrict__ wb, float* __restrict__ feat20) {
    cnn_std_split_body<P4 ? NQ_FMT_F16X4 : NQ_FMT_F16X3>(mel_tm, frame_off, tok_off, n_wins, clip_floor, n_clips, seg_hop, cw, wb, feat20);
}

typedef void (*ss_kernel_t)(const float*, const int32_t*, const int32_t*, const int32_t*, const float*, int, int, const float*,
                            const unsigned short*, float*);
// fmt: 0 bf16x3, 1 f16x3, 2 f16x4, 3 bf16x6; 70.6 / 94.6 KB of dynamic LDS: above the 64 KB default, opted in once per kernel and device
static int ss_launch(int fmt, const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off, const int32_t* n_wins,
                     const float* clip_floor, int32_t n_clips, int32_t total_tok_padded, int32_t seg_hop, const float* cnn_std_w,
                     const uint16_t* cnn_wb, float* feat20, void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31) || seg_hop <= 0 || !cnn_wb || !feat20 || fmt < 0 || fmt > 3)
        return NISQA_ERR_ARG;
    static const ss_kernel_t kernels[4] = {cnn_std_bf16_kernel, cnn_std_f16_kernel<false>, cnn_std_f16_kernel<true>, cnn_std_bf16x6_kernel};
    const int lds = fmt == 3 ? SS_LDS_T(3) : SS_LDS_T(2);
    NQ_LAUNCH_BEGIN();
    static std::atomic<bool> lds_ok[4][64];
    if (nq_lds_opt_in((const void*)kernels[fmt], lds, lds_ok[fmt])) return 2;
    hipLaunchKernelGGL(kernels[fmt], dim3(total_tok_padded / 4), dim3(256), lds, (hipStream_t)stream, mel_tm, frame_off, tok_off, n_wins,
                       clip_floor, n_clips, seg_hop, cnn_std_w, cnn_wb, feat20);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_cnn_standard_bf16(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                                       const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                                       int32_t total_tok_padded, int32_t seg_hop, const float* cnn_std_w,
                                       const uint16_t* cnn_wb, float* feat20, void* stream) {
    return ss_launch(0, mel_tm, frame_off, tok_off, n_wins, clip_floor, n_clips, total_tok_padded, seg_hop, cnn_std_w, cnn_wb, feat20, stream);
}

// the f16 formats (cnn_wh: nisqa_amd.weights.pack_adapt_cnn_f16 of the StandardCNN's convolutions; products = 3 or 4)
extern "C" int nisqa_cnn_standard_f16(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                                      const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                                      int32_t total_tok_padded, int32_t seg_hop, const float* cnn_std_w,
                                      const uint16_t* cnn_wh, int32_t products, float* feat20, void* stream) {
    if (products != 3 && products != 4) return NISQA_ERR_ARG;
    return ss_launch(products - 2, mel_tm, frame_off, tok_off, n_wins, clip_floor, n_clips, total_tok_padded, seg_hop, cnn_std_w, cnn_wh, feat20,
                     stream);
}

// the three-term form (cnn_wx: nisqa_amd.weights.pack_adapt_cnn_bf16(terms=3) of the StandardCNN's convolutions)
extern "C" int nisqa_cnn_standard_bf16x6(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                                         const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                                         int32_t total_tok_padded, int32_t seg_hop, const float* cnn_std_w,
                                         const uint16_t* cnn_wx, float* feat20, void* stream) {
    return ss_launch(3, mel_tm, frame_off, tok_off, n_wins, clip_floor, n_clips, total_tok_padded, seg_hop, cnn_std_w, cnn_wx, feat20, stream);
}
