// Split-bf16 ("bf16x3") building blocks shared by the AdaptCNN and StandardCNN kernels: MFMA wrappers, the
// compiler-visible fp32 -> bf16 split, and the barrier-free 3x3 conv layer over wave-private LDS activations.
#pragma once
#include "common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

NQ_DEV f32x16 mfma_bf(f32x4 a, f32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
NQ_DEV f32x4 mfma_bf16x16(f32x4 a, f32x4 b, f32x4 c) {     // 16x16x32: A[i = l&15][k = 8*(l>>4)+e], D row 4*(l>>4)+r
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// ---- operand formats of the two-term kernels (cnn_bf16.hip) ---------------------------------------------------------------------
//   NQ_FMT_BF16X3: x = hi + lo in bf16 (8 + 8 significand bits), products hh + hl + lh: 16 of an fp32 operand's 24 bits
//   NQ_FMT_F16X3 / NQ_FMT_F16X4: x * 2^e = hi + lo in f16 (11 + 11 significand bits and lo's sign: the residual x - hi is a
//     multiple of ulp32(x) of magnitude <= 4096 ulp32, and f16 holds every such integer up to 2048 and every even one up to 4096 --
//     the pair is the fp32 value itself for ~75 % of the values and one fp32 ulp off for the rest), e a power-of-two scale that keeps
//     the tensor inside f16's range (cnn_bf16.hip: from the measured maximum of the layer's input and the layer's weight norm);
//     products hh + hl + lh (+ ll for F16X4).  v_mfma_*_f16 honours f16 subnormals (tools/micro/f16probe.hip), so small values lose
//     absolute, not relative, precision: 2^-25 of the scaled range.  Measured against float64 on K = 576 dot products
//     (profiles/r05_micro_f16probe.txt): rms error 0.85 x (F16X4) / 0.87 x (F16X3) the fp32-MFMA kernels' -- fewer accumulator
//     roundings than their 288 K-steps -- against 0.97 x for bf16x6 and 14.6 x for BF16X3.
#define NQ_FMT_BF16X3 0
#define NQ_FMT_F16X3 1
#define NQ_FMT_F16X4 2
#define NQ_FMT_BF16X6 3                    /* three bf16 terms, six products (exact operands): the kernels that template over it keep conv_k_terms' loops */
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <int FMT>
NQ_DEV f32x16 mfma32_fmt(f32x4 a, f32x4 b, f32x16 c) {
    if (FMT == NQ_FMT_BF16X3) return mfma_bf(a, b, c);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int FMT>
NQ_DEV f32x4 mfma16_fmt(f32x4 a, f32x4 b, f32x4 c) {
    if (FMT == NQ_FMT_BF16X3) return mfma_bf16x16(a, b, c);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the term products of one (A, B) fragment pair, smallest first: ll (F16X4 only), hl, lh, hh
template <int FMT, int MT, int NT>
NQ_DEV void mma_pair_fmt(f32x16 (&acc)[MT][NT], const f32x4 (&ah)[MT], const f32x4 (&al)[MT], const f32x4 (&bh)[NT], const f32x4 (&bl)[NT]) {
    if (FMT == NQ_FMT_F16X4) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = mfma32_fmt<FMT>(al[t], bl[nt], acc[t][nt]);
    }
    // product-major: consecutive MFMAs go to DIFFERENT accumulators (no dependent-accumulate bubbles)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = mfma32_fmt<FMT>(ah[t], bl[nt], acc[t][nt]);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = mfma32_fmt<FMT>(al[t], bh[nt], acc[t][nt]);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = mfma32_fmt<FMT>(ah[t], bh[nt], acc[t][nt]);
}
template <int FMT, int MT>
NQ_DEV void mma16_pair_fmt(f32x4 (&acc)[MT], const f32x4 (&a)[MT][2], const f32x4 (&b)[2]) {
    if (FMT == NQ_FMT_F16X4) {
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = mfma16_fmt<FMT>(a[t][1], b[1], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = mfma16_fmt<FMT>(a[t][0], b[1], acc[t]);
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = mfma16_fmt<FMT>(a[t][1], b[0], acc[t]);
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = mfma16_fmt<FMT>(a[t][0], b[0], acc[t]);
}
// ---- power-of-two scales of the f16 formats (wave-uniform integer exponents) -------------------------------------------------
// e with bound * 2^e in [2^14, 2^15) (f16's largest finite value is 2^16 - 32), clamped to +-60; bound >= 0
NQ_DEV int f16_scale_exp(float bound) {
    const int be = (int)((__float_as_uint(bound) >> 23) & 0xffu);       // bound = f * 2^(be - 126), f in [0.5, 1)
    return min(max(15 - (be - 126), -60), 60);
}
NQ_DEV float pow2_f32(int e) { return __uint_as_float((unsigned)(min(max(e, -126), 127) + 127) << 23); }
// maximum of a non-negative float over the wave (four DPP steps inside the rows of 16 lanes, then the four row results through SGPRs)
NQ_DEV float wave_max_nonneg(float v) {
    int x = (int)__float_as_uint(v);
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));      // quad_perm [1, 0, 3, 2]
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));      // quad_perm [2, 3, 0, 1]
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false));     // row_half_mirror
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false));     // row_mirror
    const unsigned m = max(max((unsigned)__builtin_amdgcn_readlane(x, 0), (unsigned)__builtin_amdgcn_readlane(x, 16)),
                           max((unsigned)__builtin_amdgcn_readlane(x, 32), (unsigned)__builtin_amdgcn_readlane(x, 48)));
    return __uint_as_float(m);
}

// round-to-nearest-even fp32 -> bf16 (finite inputs)
NQ_DEV unsigned bf16_bits(float v) {
    const unsigned u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
NQ_DEV float bf16_val(unsigned b) { return __uint_as_float(b << 16); }

// fp32 -> bf16 (round to nearest even), two values per instruction: the compiler selects v_cvt_pk_bf16_f32 for
// this conversion, and -- unlike an inline-asm statement -- tracks its hazards and schedules around it
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
NQ_DEV unsigned cvt_pk_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// store v = hi + lo into the two bf16 planes at byte offset `off` of the hi plane
NQ_DEV void store_split(char* plane_hi, int plane_bytes, int off, float v) {
    const unsigned hi = cvt_pk_bf16(v, 0.f);
    const unsigned lo = cvt_pk_bf16(v - __uint_as_float(hi << 16), 0.f);
    *(unsigned short*)(plane_hi + off) = (unsigned short)hi;
    *(unsigned short*)(plane_hi + plane_bytes + off) = (unsigned short)lo;
}

// One conv layer (3x3, padding 1) for this wave's segment, barrier-free.
//   act_in : this wave's input planes (hi at +0, lo at +PLANE), pixel rows of CIN bf16, swizzled chunks
//   wb     : layer fragments [TOTAL steps][NT][2][64][8] bf16, streamed from L2: one contiguous 1 KiB
//            global_load_dwordx4 per fragment, requested TWO K-steps ahead into a 3-deep register ring
//            (an L2 round trip is ~600 clk, a step of MFMAs 200-800 clk); ~10 TB/s of L2 reads chip-wide
//   APF    : also double-buffer the A rows from LDS one step ahead (off for conv2: 6 M-tiles of registers)
//   PAD    : pixel rows are CIN*2 + 16 bytes apart and NOT swizzled (the 16-byte pad spreads consecutive pixels over the
//            banks like the XOR swizzle does, and every address becomes lane base + compile-time offset)
template <int CIN, int MT, int NT, int H, int W, bool APF, bool PAD = false>
NQ_DEV void conv3x3_bf16(f32x16 (&acc)[MT][NT], const char* act_in, const char* zero,
                         const unsigned short* __restrict__ wb, const int (&py)[MT], const int (&px)[MT],
                         const bool (&pvalid)[MT], int lane) {
    constexpr int S16 = CIN / 16;             // K=16 steps per tap
    constexpr int TOTAL = 9 * S16;
    constexpr int Cc = CIN / 8;               // 16-byte chunks per pixel row (per plane)
    constexpr int RS = CIN * 2 + (PAD ? 16 : 0);   // bytes between pixel rows
    constexpr int PLANE = H * W * RS;         // bytes per plane
    constexpr int AB = APF ? 2 : 1;
    const int h = lane >> 5;
    const f32x4* wl = (const f32x4*)wb + lane;
    f32x4 bh[3][NT], bl[3][NT], ah[AB][MT], al[AB][MT];

    auto load_b = [&](int g, int slot) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bh[slot][nt] = wl[((g * NT + nt) * 2 + 0) * 64];
            bl[slot][nt] = wl[((g * NT + nt) * 2 + 1) * 64];
        }
    };
    // A operand of tile t at K-step (tap, s): 16 bytes at  pixel row + (((2s + h) ^ swz) << 4)  of each plane.  With
    // t = h ^ swz this is  (pixel row | t << 4) ^ (32 s)  (pixel rows are aligned to their size, 32 s stays inside a
    // row), so the bounds check, the swizzle and the row address are per TAP; a K-step costs one XOR per plane.
    // Out-of-image taps point both planes at the wave's 128-byte zero block (aligned, so the XOR stays inside it).
    int a_hi[MT], a_lo[MT];
    const int zoff = (int)(zero - act_in);
    auto tap_a = [&](int tap) {
        const int dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int y = py[t] + dy, x = px[t] + dx;
            const bool ok = pvalid[t] && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const int pix = y * W + x;
            const int swz = PAD ? 0 : ((pix * Cc) >> 4) & (Cc - 1);
            const int row = pix * RS + ((h ^ swz) << 4);
            a_hi[t] = ok ? row : zoff;
            a_lo[t] = ok ? row + PLANE : zoff;
        }
    };
    auto load_a = [&](int g, int slot) {
        const int tap = g / S16, s = g - tap * S16;
        if (s == 0) tap_a(tap);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            // PAD: plain + 32 s (an immediate offset of the LDS read); swizzled: ^ 32 s
            ah[slot][t] = *(const f32x4*)(act_in + (PAD ? a_hi[t] + 32 * s : a_hi[t] ^ (32 * s)));
            al[slot][t] = *(const f32x4*)(act_in + (PAD ? a_lo[t] + 32 * s : a_lo[t] ^ (32 * s)));
        }
    };

    load_b(0, 0);
    load_b(1, 1);
    if (APF) load_a(0, 0);
#pragma unroll
    for (int g = 0; g < TOTAL; ++g) {
        if (g + 2 < TOTAL) load_b(g + 2, (g + 2) % 3);
        if (APF) { if (g + 1 < TOTAL) load_a(g + 1, (g + 1) & 1); } else load_a(g, 0);
        const int sa = APF ? (g & 1) : 0, sb = g % 3;
        // product-major: consecutive MFMAs go to DIFFERENT accumulators (no dependent-accumulate bubbles)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = mfma_bf(ah[sa][t], bl[sb][nt], acc[t][nt]);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = mfma_bf(al[sa][t], bh[sb][nt], acc[t][nt]);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = mfma_bf(ah[sa][t], bh[sb][nt], acc[t][nt]);

    }
}



// ======================================================================================================================
// Second-generation K loop (AdaptCNN kernel): LDS addressed by 32-bit byte addresses in address space 3 (no 64-bit
// pointer arithmetic), tap validity as one precomputed 9-bit mask per M tile (lane-static: the row -> pixel maps do not
// depend on the data), out-of-image taps redirected to a zero block with ONE select per tap and tile, weight fragments
// through a buffer resource (address = SGPR descriptor + constant lane offset + immediate).  DESIGN.md 4.5.
// ======================================================================================================================
#define NQ_AS3 __attribute__((address_space(3)))
// Scheduling fences (experiment -DNQ_SB=1|3, DESIGN.md 4.5 "round 3"): hipcc's machine scheduler sinks the ring's
// prefetches to just before their use (register-pressure heuristics: the ISA shows s_waitcnt vmcnt(1) two MFMAs behind a
// request).  sched_barrier(0) is a wall no instruction is moved across; with one per K-step (bit 1) and one behind the
// step's requests (bit 2) the loop comes out exactly as written (requests for step g + 2 / g + 1 first, waits of
// vmcnt(10) / lgkmcnt(7), no stall on either).  Measured 0.401 ms against 0.391 without: the waves are not short of
// prefetch distance, the other wave of the SIMD fills those stalls; left off.
#ifndef NQ_SB
#define NQ_SB 0
#endif
#define NQ_STEP_FENCE() do { if (NQ_SB & 1) __builtin_amdgcn_sched_barrier(0); } while (0)
#define NQ_ISSUE_FENCE() do { if (NQ_SB & 2) __builtin_amdgcn_sched_barrier(0); } while (0)
NQ_DEV f32x4 lds_ld128(unsigned a) { return *(NQ_AS3 const f32x4*)(a); }
NQ_DEV unsigned lds_ld32(unsigned a) { return *(NQ_AS3 const unsigned*)(a); }
NQ_DEV void lds_st16(unsigned a, unsigned v) { *(NQ_AS3 unsigned short*)(a) = (unsigned short)v; }
NQ_DEV void lds_st32(unsigned a, unsigned v) { *(NQ_AS3 unsigned*)(a) = v; }
NQ_DEV void lds_st128(unsigned a, f32x4 v) { *(NQ_AS3 f32x4*)(a) = v; }
// v = hi + lo into the two bf16 planes (lo plane `plane` bytes behind the hi plane)
NQ_DEV void lds_store_split(unsigned a, int plane, float v) {
    const unsigned hi = cvt_pk_bf16(v, 0.f);
    const unsigned lo = cvt_pk_bf16(v - __uint_as_float(hi << 16), 0.f);
    lds_st16(a, hi);
    lds_st16(a + plane, lo);
}
// two values at once: ONE v_cvt_pk_bf16_f32 per plane for the pair, the residual as a packed subtraction, the halves of the
// packed results stored with ds_write_b16 / ds_write_b16_d16_hi (2.5 VALU instructions per value instead of 4; the bits are
// those of lds_store_split)
NQ_DEV void lds_st16_hi(unsigned a, unsigned v) { *(NQ_AS3 unsigned short*)(a) = (unsigned short)(v >> 16); }
NQ_DEV void lds_store_split2(unsigned a0, unsigned a1, int plane, float v0, float v1, bool st0 = true, bool st1 = true) {
    const unsigned hi2 = cvt_pk_bf16(v0, v1);
    const f32x2_t vv = {v0, v1};
    const f32x2_t hf = {__uint_as_float(hi2 << 16), __uint_as_float(hi2 & 0xffff0000u)};
    const f32x2_t r = vv - hf;
    const unsigned lo2 = cvt_pk_bf16(r[0], r[1]);
    if (st0) { lds_st16(a0, hi2); lds_st16(a0 + plane, lo2); }
    if (st1) { lds_st16_hi(a1, hi2); lds_st16_hi(a1 + plane, lo2); }
}
// fp32 pair -> packed f16 pair (v_cvt_pk_f16_f32, round to nearest even)
NQ_DEV unsigned cvt_pk_f16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
// lds_store_split2 for either format; `mx` (F16 formats) gathers the maximum of the stored values (they are >= 0 behind a ReLU; the
// conv1 input passes |v|): the next layer's scale comes from it
template <int T> NQ_DEV void lds_store_terms2(unsigned a0, unsigned a1, int plane, float v0, float v1, bool st0 = true, bool st1 = true);
template <int T> NQ_DEV void lds_store_terms(unsigned a, int plane, float v);
template <int FMT>
NQ_DEV void lds_store_pair_fmt(unsigned a0, unsigned a1, int plane, float v0, float v1, float& mx, bool st0 = true, bool st1 = true) {
    if (FMT == NQ_FMT_BF16X3) { lds_store_split2(a0, a1, plane, v0, v1, st0, st1); return; }
    if (FMT == NQ_FMT_BF16X6) { lds_store_terms2<3>(a0, a1, plane, v0, v1, st0, st1); return; }
    const unsigned hi2 = cvt_pk_f16(v0, v1);
    const f32x2_t vv = {v0, v1};
    const f32x2_t hf = __builtin_convertvector(__builtin_bit_cast(f16x2_t, hi2), f32x2_t);
    const f32x2_t r = vv - hf;
    const unsigned lo2 = cvt_pk_f16(r[0], r[1]);
    mx = fmaxf(fmaxf(mx, __builtin_fabsf(v0)), __builtin_fabsf(v1));
    if (st0) { lds_st16(a0, hi2); lds_st16(a0 + plane, lo2); }
    if (st1) { lds_st16_hi(a1, hi2); lds_st16_hi(a1 + plane, lo2); }
}
template <int FMT>
NQ_DEV void lds_store_one_fmt(unsigned a, int plane, float v, float& mx) {
    if (FMT == NQ_FMT_BF16X3) { lds_store_split(a, plane, v); return; }
    if (FMT == NQ_FMT_BF16X6) { lds_store_terms<3>(a, plane, v); return; }
    const unsigned hi = cvt_pk_f16(v, 0.f);
    const float hf = (float)__builtin_bit_cast(f16x2_t, hi)[0];
    const unsigned lo = cvt_pk_f16(v - hf, 0.f);
    mx = fmaxf(mx, __builtin_fabsf(v));
    lds_st16(a, hi);
    lds_st16(a + plane, lo);
}
NQ_DEV f32x16 splat16(float v) { f32x16 r; for (int q = 0; q < 16; ++q) r[q] = v; return r; }
typedef int nq_i32x4 __attribute__((ext_vector_type(4)));
NQ_DEV f32x4 wfrag_load(__amdgpu_buffer_rsrc_t rsrc, unsigned lane16, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16, byte_off, 0));
}
// 9-bit tap mask of output pixel (y, x) of an H x W image: bit dy * 3 + dx set iff (y + dy - 1, x + dx - 1) is inside
NQ_DEV unsigned tap_mask(bool valid, int y, int x, int H, int W) {
    unsigned m = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        m |= (valid && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? (1u << tap) : 0u;
    }
    return m;
}

// acc[t][nt] += conv over K = 9 taps x CIN channels for this wave's MT row tiles.
//   base[t] : LDS address of the 16-byte chunk (K half h of channel step 0) of the pixel at tap (-1, -1) of tile t's row,
//             hi plane; the lo plane is PLANE bytes behind.  Tap (dy, dx) sits TAPOFF = (dy * W + dx) * RS bytes further
//             (dy, dx = 0..2), channel step s 32 s bytes further: both ride on the ds_read offset field.
//   m9[t]   : tap_mask of the row; a cleared bit sends both reads to the zero block at ZADDR (a kernel-wide constant).
//   wbyte   : byte offset of the layer's fragments [step][NT][hi, lo][64 lanes][8 bf16] in the weight blob
template <int CIN, int MT, int NT, int W, int RS, int PLANE, unsigned ZADDR, bool APF, int RING = 3, int FMT = NQ_FMT_BF16X3>
NQ_DEV void conv_k_bf16(f32x16 (&acc)[MT][NT], __amdgpu_buffer_rsrc_t rsrc, int wbyte, unsigned lane16,
                        const unsigned (&base)[MT], const unsigned (&m9)[MT]) {
    constexpr int S16 = CIN / 16, TOTAL = 9 * S16, AB = APF ? 2 : 1;
    static_assert(ZADDR >= (2 * W + 2) * RS + 32 * S16, "zero block must sit above the largest tap offset");
    f32x4 bh[RING][NT], bl[RING][NT], ah[AB][MT], al[AB][MT];
    unsigned a_hi[MT], a_lo[MT];
    auto load_b = [&](int g, int slot) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bh[slot][nt] = wfrag_load(rsrc, lane16, wbyte + ((g * NT + nt) * 2 + 0) * 1024);
            bl[slot][nt] = wfrag_load(rsrc, lane16, wbyte + ((g * NT + nt) * 2 + 1) * 1024);
        }
    };
    auto load_a = [&](int g, int slot) {
        const int tap = g / S16, s = g - tap * S16;
        const int tapoff = ((tap / 3) * W + tap % 3) * RS;
        if (s == 0) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const bool ok = (m9[t] >> tap) & 1u;
                a_hi[t] = ok ? base[t] : ZADDR - tapoff;
                a_lo[t] = ok ? base[t] + PLANE : ZADDR - tapoff;
            }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            ah[slot][t] = lds_ld128(a_hi[t] + tapoff + 32 * s);
            al[slot][t] = lds_ld128(a_lo[t] + tapoff + 32 * s);
        }
    };
#pragma unroll
    for (int g = 0; g < RING - 1; ++g) load_b(g, g);
    if (APF) load_a(0, 0);
#pragma unroll
    for (int g = 0; g < TOTAL; ++g) {
        NQ_STEP_FENCE();
        if (g + RING - 1 < TOTAL) load_b(g + RING - 1, (g + RING - 1) % RING);
        if (APF) { if (g + 1 < TOTAL) load_a(g + 1, (g + 1) & 1); } else load_a(g, 0);
        NQ_ISSUE_FENCE();
        const int sa = APF ? (g & 1) : 0, sb = g % RING;
        mma_pair_fmt<FMT, MT, NT>(acc, ah[sa], al[sa], bh[sb], bl[sb]);
    }
}


// ======================================================================================================================
// T terms per operand (cnn_bf16x6.hip: T = 3, bf16 hi + mid + lo = the fp32 operand EXACTLY, 24 mantissa bits): the products
// (i, j) with i + j <= T - 1 are formed, smallest first -- for T = 3 the six products hh, hm, mh, hl, lh, mm; what is dropped
// (ml, lm, ll) is at most 2 x 2^-24 of the product, typically 0.5 x 2^-24 rms: the size of an fp32 multiply-add's own rounding of
// that product.  T = 2 is the shipped hi/lo form (three products).
// ======================================================================================================================
// v0, v1 -> T bf16 terms each (round to nearest: |term t+1| <= 2^-8 |term t|), plane t `t * plane` bytes behind the first; the halves of the packed conversions are stored with
// ds_write_b16 / ds_write_b16_d16_hi as in lds_store_split2.
template <int T>
NQ_DEV void lds_store_terms2(unsigned a0, unsigned a1, int plane, float v0, float v1, bool st0, bool st1) {
    f32x2_t r = {v0, v1};
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const unsigned pk = cvt_pk_bf16(r[0], r[1]);
        if (st0) lds_st16(a0 + t * plane, pk);
        if (st1) lds_st16_hi(a1 + t * plane, pk);
        if (t + 1 < T) {
            const f32x2_t part = {__uint_as_float(pk << 16), __uint_as_float(pk & 0xffff0000u)};
            r = r - part;
        }
    }
}
template <int T>
NQ_DEV void lds_store_terms(unsigned a, int plane, float v) {
    float r = v;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const unsigned pk = cvt_pk_bf16(r, 0.f);
        lds_st16(a + t * plane, pk);
        if (t + 1 < T) r -= __uint_as_float(pk << 16);
    }
}
// acc[m][nt] += sum over the kept term products of a[m][i] x b[nt][j]; smallest products first, consecutive MFMAs on
// different accumulators
template <int T, int MT, int NT>
NQ_DEV void mma_terms(f32x16 (&acc)[MT][NT], const f32x4 (&a)[MT][T], const f32x4 (&b)[NT][T]) {
#pragma unroll
    for (int order = T - 1; order >= 0; --order)
#pragma unroll
        for (int i = order; i >= 0; --i) {                 // (i, order - i): within an order the term with the smaller A part first
            const int j = order - i;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma_bf(a[m][i], b[nt][j], acc[m][nt]);
        }
}
// conv_k_bf16 for T terms: fragments [step][NT][T][64 lanes][8 bf16], activation planes PLANE bytes apart; A rows always
// one step ahead (this form runs one wave per SIMD on the 512-register budget).  FENCE: a sched_barrier behind the step's
// requests -- hipcc's scheduler otherwise sinks them to just before their use, and with ONE wave per SIMD nobody fills the
// stall (tools/micro/klx6.hip: matrix-pipe duty 0.82 -> 0.96 in the conv3 + conv4 loops; with two waves per SIMD the same
// fence made the two-term kernel slower)
// the first RING - 1 steps of a layer's fragments, requested by the caller a phase ahead (behind the previous layer's K loop,
// above its epilogue): with one wave per SIMD nobody else covers the L2 round trip a K loop otherwise opens with
template <int T, int NT, int RING = 3>
struct conv_k_ring { f32x4 b[RING][NT][T]; };
template <int T, int NT, int RING>
NQ_DEV void conv_k_preload(conv_k_ring<T, NT, RING>& r, __amdgpu_buffer_rsrc_t rsrc, int wbyte, unsigned lane16) {
#pragma unroll
    for (int g = 0; g < RING - 1; ++g)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int t = 0; t < T; ++t) r.b[g][nt][t] = wfrag_load(rsrc, lane16, wbyte + ((g * NT + nt) * T + t) * 1024);
}
// PRE: `ring` already holds the first RING - 1 steps (conv_k_preload)
template <int T, int CIN, int MT, int NT, int W, int RS, int PLANE, unsigned ZADDR, int RING, bool FENCE, bool PRE>
NQ_DEV void conv_k_terms_ring(f32x16 (&acc)[MT][NT], __amdgpu_buffer_rsrc_t rsrc, int wbyte, unsigned lane16,
                              const unsigned (&base)[MT], const unsigned (&m9)[MT], conv_k_ring<T, NT, RING>& ring) {
    constexpr int S16 = CIN / 16, TOTAL = 9 * S16;
    static_assert(ZADDR >= (2 * W + 2) * RS + 32 * S16, "zero block must sit above the largest tap offset");
    f32x4 a[2][MT][T];
    unsigned a_ad[MT][T];
    auto load_b = [&](int g, int slot) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int t = 0; t < T; ++t) ring.b[slot][nt][t] = wfrag_load(rsrc, lane16, wbyte + ((g * NT + nt) * T + t) * 1024);
    };
    auto load_a = [&](int g, int slot) {
        const int tap = g / S16, s = g - tap * S16;
        const int tapoff = ((tap / 3) * W + tap % 3) * RS;
        if (s == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool ok = (m9[m] >> tap) & 1u;
#pragma unroll
                for (int t = 0; t < T; ++t) a_ad[m][t] = ok ? base[m] + t * PLANE : ZADDR - tapoff;
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < T; ++t) a[slot][m][t] = lds_ld128(a_ad[m][t] + tapoff + 32 * s);
    };
    if (!PRE) {
#pragma unroll
        for (int g = 0; g < RING - 1; ++g) load_b(g, g);
    }
    load_a(0, 0);
#pragma unroll
    for (int g = 0; g < TOTAL; ++g) {
        if (g + RING - 1 < TOTAL) load_b(g + RING - 1, (g + RING - 1) % RING);
        if (g + 1 < TOTAL) load_a(g + 1, (g + 1) & 1);
        if (FENCE) __builtin_amdgcn_sched_barrier(0);
        mma_terms<T, MT, NT>(acc, a[g & 1], ring.b[g % RING]);
    }
}
template <int T, int CIN, int MT, int NT, int W, int RS, int PLANE, unsigned ZADDR, int RING = 3, bool FENCE = true>
NQ_DEV void conv_k_terms(f32x16 (&acc)[MT][NT], __amdgpu_buffer_rsrc_t rsrc, int wbyte, unsigned lane16,
                         const unsigned (&base)[MT], const unsigned (&m9)[MT]) {
    conv_k_ring<T, NT, RING> ring;
    conv_k_terms_ring<T, CIN, MT, NT, W, RS, PLANE, ZADDR, RING, FENCE, false>(acc, rsrc, wbyte, lane16, base, m9, ring);
}
