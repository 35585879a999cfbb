// Mel front end for gfx950 -- replaces get_librosa_melspec after lb.load
// (reference nisqa/NISQA_lib.py:2308-2331; arithmetic of librosa 0.8.1 stft / filters.mel /
// amplitude_to_db restated in oracle/mel.py).
//
// One wave computes one STFT frame end to end (window -> FFT -> |.| -> mel -> dB) and walks over
// FRAMES_PER_WAVE consecutive frames with all per-lane constants resident in registers:
//   * pruned FFT: at sr <= 51.2 kHz the hann window has `win` <= 1024 non-zero taps inside the 4096-sample frame,
//     so after a (magnitude-preserving) circular shift the real sequence is supported on [0, 1024) (longer windows,
//     e.g. 1920 taps at 96 kHz, take the NQ = 2 / 4 instantiations that fold the extra quarters in first).
//     Packed as z[n] = x[2n] + i x[2n+1] (n < 512) the 2048-point complex FFT collapses to FOUR
//     512-point FFTs of z[n] * W2048^(r n), r = 0..3, giving Z[4m + r]: the three outer radix-4
//     stages of a 4096-point transform are never executed;
//   * each 512-point FFT is radix 8x8x8 with 8 complex values per lane and two wave-private LDS
//     transposes (row strides 72 complex / 80 B: conflict free for ds_write_b64 / ds_read_b128);
//     every twiddle is either a per-lane register loaded once per wave or a compile-time constant;
//   * real-input recombination X[K] = (Z[K] + conj Z[2048-K])/2 - i W4096^K (Z[K] - conj Z[2048-K])/2:
//     the partner of Z_r[k] is Z_(4-r)[511-k] (Z_0[512-k] for r = 0), i.e. the SAME register index
//     mirrored in lane 63-l (64-l): a lane shuffle, no spectrum buffer;
//   * slaney filterbank in sparse form (each bin feeds <= 2 triangles): 4 bands per pass, one per
//     16-lane row, weights shared in LDS by the workgroup, row totals by DPP butterflies;
//   * 10*log10(max(amin^2, S^2)), coalesced 192-byte store per frame, running per-clip maximum kept
//     in a register and published with one order-preserving atomicMax per wave and clip;
//   * the next frame's samples are prefetched while the current frame is transformed.
// The top_db clamp needs the per-clip maximum, i.e. a reduction over every frame of the clip; it
// is applied by the consumer (the CNN loads max(x, floor)) or by nisqa_mel_finalize in place.
#include <stdlib.h>
#include "common.hpp"
#include "../../include/nisqa_hip.h"

// complex numbers as 64-bit register pairs: add / sub / multiply map onto v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32 (two flops per lane and instruction; swaps and sign flips ride on op_sel / neg modifiers)
typedef float c32 __attribute__((ext_vector_type(2)));
NQ_DEV c32 cmk(float x, float y) { return c32{x, y}; }
NQ_DEV c32 cadd(c32 a, c32 b) { return a + b; }
NQ_DEV c32 csub(c32 a, c32 b) { return a - b; }
NQ_DEV c32 cmul(c32 a, c32 b) {                        // (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
    return __builtin_elementwise_fma(c32{a.y, a.y}, c32{-b.y, b.x}, c32{a.x, a.x} * b);
}
NQ_DEV c32 cnegi(c32 a) { return c32{a.y, -a.x}; }     // a * (-i)

// a +- (-i) b and a +- conj(b) in ONE packed add: the half swap and the sign ride on the op_sel / neg modifiers of
// v_pk_add_f32 (the compiler materialises cnegi / conj as v_xor + v_mov first; same IEEE adds, same bits).
// The half-swapped operand must be SRC0: on gfx950 a packed-f32 instruction whose LOW result reads the HIGH half of a
// VGPR src1 (op_sel:[x,1]) returns wrong values in lanes 48..63 while 16-bit-input MFMA waves of ANOTHER kernel share
// the SIMD (tools/micro/corun6.hip); the same swap on src0 (or src2 of an fma) is exact.
// tests/test_host.py scans the ISA of every kernel for the bad form.
NQ_DEV c32 cadd_mi(c32 a, c32 b) {                     // (a.x + b.y, a.y - b.x)
    c32 r;
    asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
NQ_DEV c32 csub_mi(c32 a, c32 b) {                     // (a.x - b.y, a.y + b.x)
    c32 r;
    asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
NQ_DEV c32 cadd_conj(c32 a, c32 b) {                   // (a.x + b.x, a.y - b.y)
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
NQ_DEV c32 csub_conj(c32 a, c32 b) {                   // (a.x - b.x, a.y + b.y)
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

NQ_DEV void dft4(c32& a0, c32& a1, c32& a2, c32& a3) {
    const c32 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
    a0 = cadd(t0, t2); a1 = cadd_mi(t1, d); a2 = csub(t0, t2); a3 = csub_mi(t1, d);
}

// v[p] <- sum_a v[a] * exp(-2 pi i a p / 8), natural order in and out
NQ_DEV void dft8(c32 (&v)[8]) {
    c32 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    c32 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    const float r = 0.70710678118654752440f;
    const c32 s1 = cadd_mi(o1, o1);                       // o1 * W8^1 = r (x + y, y - x) = r s1
    const c32 s3 = csub_mi(o3, o3);                       // o3 * W8^3 = r (y - x, -x - y) = -r s3
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = e1 + s1 * r; v[5] = e1 - s1 * r;
    v[2] = cadd_mi(e2, o2); v[6] = csub_mi(e2, o2);       // o2 * W8^2 = -i o2
    v[3] = e3 - s3 * r; v[7] = e3 + s3 * r;
}

// exp(-2 pi i m / 32) and exp(-2 pi i m / 16) as (cos, -sin)
__device__ constexpr float W32C[32] = {1.000000000e+00f, 9.807852804e-01f, 9.238795325e-01f, 8.314696123e-01f, 7.071067812e-01f, 5.555702330e-01f, 3.826834324e-01f, 1.950903220e-01f, 6.123233996e-17f, -1.950903220e-01f, -3.826834324e-01f, -5.555702330e-01f, -7.071067812e-01f, -8.314696123e-01f, -9.238795325e-01f, -9.807852804e-01f, -1.000000000e+00f, -9.807852804e-01f, -9.238795325e-01f, -8.314696123e-01f, -7.071067812e-01f, -5.555702330e-01f, -3.826834324e-01f, -1.950903220e-01f, -1.836970199e-16f, 1.950903220e-01f, 3.826834324e-01f, 5.555702330e-01f, 7.071067812e-01f, 8.314696123e-01f, 9.238795325e-01f, 9.807852804e-01f};
__device__ constexpr float W32S[32] = {-0.000000000e+00f, -1.950903220e-01f, -3.826834324e-01f, -5.555702330e-01f, -7.071067812e-01f, -8.314696123e-01f, -9.238795325e-01f, -9.807852804e-01f, -1.000000000e+00f, -9.807852804e-01f, -9.238795325e-01f, -8.314696123e-01f, -7.071067812e-01f, -5.555702330e-01f, -3.826834324e-01f, -1.950903220e-01f, -1.224646799e-16f, 1.950903220e-01f, 3.826834324e-01f, 5.555702330e-01f, 7.071067812e-01f, 8.314696123e-01f, 9.238795325e-01f, 9.807852804e-01f, 1.000000000e+00f, 9.807852804e-01f, 9.238795325e-01f, 8.314696123e-01f, 7.071067812e-01f, 5.555702330e-01f, 3.826834324e-01f, 1.950903220e-01f};
__device__ constexpr float W16C[16] = {1.000000000e+00f, 9.238795325e-01f, 7.071067812e-01f, 3.826834324e-01f, 6.123233996e-17f, -3.826834324e-01f, -7.071067812e-01f, -9.238795325e-01f, -1.000000000e+00f, -9.238795325e-01f, -7.071067812e-01f, -3.826834324e-01f, -1.836970199e-16f, 3.826834324e-01f, 7.071067812e-01f, 9.238795325e-01f};
__device__ constexpr float W16S[16] = {-0.000000000e+00f, -3.826834324e-01f, -7.071067812e-01f, -9.238795325e-01f, -1.000000000e+00f, -9.238795325e-01f, -7.071067812e-01f, -3.826834324e-01f, -1.224646799e-16f, 3.826834324e-01f, 7.071067812e-01f, 9.238795325e-01f, 1.000000000e+00f, 9.238795325e-01f, 7.071067812e-01f, 3.826834324e-01f};

// wave-private LDS round trips: a compiler barrier (the hardware keeps one wave's LDS operations in order)
#define MEL_WBAR() __builtin_amdgcn_wave_barrier()
// waves per workgroup: the 48 kHz-class instantiation (NQ = 1) fits 168 registers and 9.3 KB of LDS per wave -> ONE workgroup of
// twelve waves per CU = three per SIMD (the kernel is short of resident waves); longer windows keep four
#define MEL_WAVES_OF(NQ) ((NQ) == 1 ? 12 : 4)
// per-frame phase clock (tools/mel_clock.py; empty macros unless the unit is built with -DNQ_EXPERIMENTAL): phases 0..6, [8] = frames
NQ_CLK_EXPORT(g_mel_clk, nisqa_debug_mel_clock)
#define MEL_TAB_BYTES (4096 + 2048 + 1536 + 6144)   /* window taps, W4096 / W2048 twiddles, per-(pass, lane) filter-bank offsets */
#define MEL_EXCH_BYTES 5120            /* exchange 1 [8][72] complex (4608 B) and exchange 2 64 x 80 B alias */

// sum over the 16 lanes of a DPP row, result in every lane of the row
NQ_DEV float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

NQ_DEV c32 shfl_c(c32 v, int src) { return cmk(__shfl((float)v.x, src), __shfl((float)v.y, src)); }

// |2 X[K]| for K = 4k + r from za = Z[K], zb = Z[2048-K], W4096^K = wl * wc (per-lane x constant part;
// applied one after the other so that nothing loop-invariant can be hoisted into 64 extra registers)
NQ_DEV float xmag(c32 za, c32 zb, c32 wl, c32 wc) {
    const c32 a = cadd_conj(za, zb);                     // Za + conj(Zb)
    const c32 w = cmul(cmul(csub_conj(za, zb), wl), wc); // W4096^K (Za - conj(Zb))
    const c32 x = cadd_mi(a, w);                         // 2 X[K] = a - i w
    const c32 x2 = x * x;
    return __builtin_amdgcn_sqrtf(x2.x + x2.y);          // |2 X[K]|; v_sqrt_f32 (1 ulp): 5e-7 dB, saves ~10 VALU per bin
}

// two bins at once, their dependent chains interleaved link by link: on gfx950 a packed-f32 operation that reads the result
// of the packed operation right before it costs an s_nop; hipcc's scheduler does not model that and leaves chains as chains
NQ_DEV void xmag2(c32 za0, c32 zb0, c32 wl0, c32 wc0, c32 za1, c32 zb1, c32 wl1, c32 wc1, float& m0, float& m1) {
    const c32 a0 = cadd_conj(za0, zb0), a1 = cadd_conj(za1, zb1);
    const c32 d0 = csub_conj(za0, zb0), d1 = csub_conj(za1, zb1);
    c32 p0 = c32{d0.x, d0.x} * wl0, p1 = c32{d1.x, d1.x} * wl1;                       // cmul(d, wl), link by link
    p0 = __builtin_elementwise_fma(c32{d0.y, d0.y}, c32{-wl0.y, wl0.x}, p0);
    p1 = __builtin_elementwise_fma(c32{d1.y, d1.y}, c32{-wl1.y, wl1.x}, p1);
    c32 q0 = c32{p0.x, p0.x} * wc0, q1 = c32{p1.x, p1.x} * wc1;                       // cmul(., wc)
    q0 = __builtin_elementwise_fma(c32{p0.y, p0.y}, c32{-wc0.y, wc0.x}, q0);
    q1 = __builtin_elementwise_fma(c32{p1.y, p1.y}, c32{-wc1.y, wc1.x}, q1);
    const c32 x0 = cadd_mi(a0, q0), x1 = cadd_mi(a1, q1);
    const c32 s0 = x0 * x0, s1 = x1 * x1;
    m0 = __builtin_amdgcn_sqrtf(s0.x + s0.y);
    m1 = __builtin_amdgcn_sqrtf(s1.x + s1.y);
}

struct mel_twiddles {
    c32 b[8];   // W512^(l p)
    c32 c[8];   // W64^((l&7) q1)
};

// 512-point FFT of u[a] = z[l + 64 a] * W2048^(r (l + 64 a)); returns Z_r[l + 64 q2] in u[q2]
template <int R>
NQ_DEV void fft512(c32 (&u)[8], const c32 (&z)[8], const mel_twiddles& tw, char* exch, int lane, const c32* tab_a) {
    const c32 aR = R ? tab_a[(R - 1) * 64 + lane] : cmk(1.f, 0.f);
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        if (R == 0 || a == 0) u[a] = z[a];
        else u[a] = cmul(z[a], cmk(W32C[(R * a) & 31], W32S[(R * a) & 31]));     // W2048^(64 R a) = W32^(R a)
    }
    dft8(u);                                                // over a -> p
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        if (p != 0) u[p] = cmul(u[p], tw.b[p]);               // W512^(l p)
        if (R != 0) u[p] = cmul(u[p], aR);                    // W2048^(R l): per-lane part of the pre-twiddle
    }
    c32* b1 = (c32*)exch;
#pragma unroll
    for (int p = 0; p < 8; ++p) b1[p * 72 + lane] = u[p];
    MEL_WBAR();
    const int pq = lane >> 3, j1 = lane & 7;
#pragma unroll
    for (int j2 = 0; j2 < 8; ++j2) u[j2] = b1[pq * 72 + j1 + 8 * j2];
    MEL_WBAR();
    dft8(u);                                                // over j2 -> q1
#pragma unroll
    for (int q1 = 1; q1 < 8; ++q1) u[q1] = cmul(u[q1], tw.c[q1]);
#pragma unroll
    for (int q1 = 0; q1 < 8; ++q1) *(c32*)(exch + (pq + 8 * q1) * 80 + j1 * 8) = u[q1];
    MEL_WBAR();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t4 = *(const f32x4*)(exch + lane * 80 + q * 16);
        u[2 * q] = cmk(t4[0], t4[1]);
        u[2 * q + 1] = cmk(t4[2], t4[3]);
    }
    MEL_WBAR();
    dft8(u);                                                // over j1 -> q2 ; k = lane + 64 q2
}

// NQ = ceil(win / 1024) quarters of the frame carry non-zero window taps (1 for sr <= 51.2 kHz, 2 for 96 kHz, 4 up to the
// full 4096): z[n] for n >= 512 q folds onto n - 512 q with the radix-4 factor (-i)^(q r) before the 512-point FFTs
// T = float (samples as lb.load returns them) or int16_t (PCM16 as it sits in the file: the x / 32768 of soundfile is
// folded into the window taps -- a power of two, so both instantiations produce the same bits)
// FB: filter-bank trip counts known at compile time (round 3).  The twelve passes of the sparse bank are chains of dependent
// LDS round trips (run-time trip count -> no overlap between passes); with the counts of the two shipped configurations as
// constants every pass is unrolled, its reads issue together and the twelve row sums interleave.  0: generic (any sample
// rate / fmax); 1: 48 kHz, fmax 20 kHz (nisqa.tar, nisqa_mos_only.tar); 2: 48 kHz, fmax 8 kHz (nisqa_tts.tar).  The kernel
// checks the table it was given against the constants once per wave and falls back to the generic loop on a mismatch.
__device__ constexpr int MEL_FB_NIT[3][12] = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
                                              {1, 1, 1, 2, 2, 3, 4, 5, 6, 9, 12, 17},
                                              {1, 1, 1, 1, 1, 2, 2, 2, 3, 4, 4, 6}};
template <int NQ, typename T, int FB = 0>
__global__ __launch_bounds__(64 * MEL_WAVES_OF(NQ), NQ == 2 ? 2 : 1) void mel_frame_kernel(
    const T* __restrict__ pcm, const int64_t* __restrict__ clip_off,
    const int32_t* __restrict__ frame_off, int n_clips, int total_frames, int frames_per_wave,
    nisqa_mel_cfg cfg, int mag_stride, int w_floats,
    const float* __restrict__ window, const float2* __restrict__ twg,
    const int32_t* __restrict__ band_start, const int32_t* __restrict__ band_len,
    const int32_t* __restrict__ band_woff, const float* __restrict__ band_w,
    float* __restrict__ mel_tm, uint32_t* __restrict__ clip_max_enc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MEL_WAVES = MEL_WAVES_OF(NQ);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* wlds = (float*)smem;                                   // shared sparse filterbank weights
    const int w_bytes = (w_floats * 4 + 15) & ~15;
    // per wave: the four magnitude planes in the order r = 0, 2, 1, 3, then what the FFT exchange (5120 B) needs beyond planes
    // 1 and 3, which it OVERLAYS (their magnitudes are written after the last transform of a frame), then slack
    const int exch_extra = max(0, MEL_EXCH_BYTES - 8 * mag_stride);
    const int per_wave = mag_stride * 16 + exch_extra + 512;
    float* tab_w = (float*)(smem + w_bytes);                      // window taps (pre-scaled), 1024 floats
    c32* tab_d = (c32*)(smem + w_bytes + 4096);                   // W4096^(4 l + r), [4][64]
    c32* tab_a = (c32*)(smem + w_bytes + 4096 + 2048);            // W2048^(r l), r = 1..3, [3][64]
    int* tab_band = (int*)(smem + w_bytes + 4096 + 2048 + 1536);  // per (pass, lane): byte offsets of the first magnitude and the first weight
    float* mag = (float*)(smem + w_bytes + MEL_TAB_BYTES + wave * per_wave);   // |X[K]| at plane(K&3)*mag_stride + (K>>2), plane(r) = 0, 2, 1, 3
    char* exch = (char*)(mag + 2 * mag_stride);                   // = planes 1, 3 (+ extra)
    // the 1/2 of |X[K]| = |2 X[K]| / 2 rides on the band weights (a power of two: same bits), not on every bin
    for (int i = tid; i < w_floats; i += 64 * MEL_WAVES) wlds[i] = 0.5f * band_w[i];
    for (int i = lane; i < per_wave / 4; i += 64) mag[i] = 0.f;              // planes + slack start finite
    {
        const float sc_ = sizeof(T) == 2 ? 1.0f / 32768.0f : 1.0f;
        for (int i = tid; i < 1024; i += 64 * MEL_WAVES) tab_w[i] = (NQ == 1 && i < cfg.win) ? window[i] * sc_ : 0.f;
        for (int i = tid; i < 256; i += 64 * MEL_WAVES) { const float2 w = twg[4 * (i & 63) + (i >> 6)]; tab_d[i] = cmk(w.x, w.y); }
        for (int i = tid; i < 192; i += 64 * MEL_WAVES) { const float2 w = twg[(2 * (i / 64 + 1) * (i & 63)) & 4095]; tab_a[i] = cmk(w.x, w.y); }
        for (int i = tid; i < 12 * 64; i += 64 * MEL_WAVES) {           // pass ps, lane l: band 4 ps + (l >> 4), first bin K0 + (l & 15)
            const int bnd = 4 * (i >> 6) + ((i & 63) >> 4), K0_ = band_start[bnd] + (i & 15);
            tab_band[2 * i] = 4 * ((((K0_ & 1) << 1) | ((K0_ >> 1) & 1)) * mag_stride + (K0_ >> 2));   // plane order 0, 2, 1, 3
            tab_band[2 * i + 1] = 4 * (band_woff[bnd] + (i & 15));
        }
    }
    __syncthreads();

    // ---- per-lane constants, loaded once per wave
    mel_twiddles tw;
#pragma unroll
    for (int p = 0; p < 8; ++p) { const float2 w = twg[(8 * lane * p) & 4095]; tw.b[p] = cmk(w.x, w.y); }
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float2 w = twg[(64 * (lane & 7) * q) & 4095]; tw.c[q] = cmk(w.x, w.y); }
    constexpr bool PCM16 = sizeof(T) == 2;
    const float scale = PCM16 ? 1.0f / 32768.0f : 1.0f;
    // band tables of this lane's DPP row: pass ps handles band 4*ps + row
    const int row = lane >> 4, l16 = lane & 15;
    // per pass: this lane's first magnitude index (plane (K&3), entry K>>2; K advances by 16 = 4 entries per
    // iteration, so the plane never changes) and first weight index.  Reads past a band's support meet zero
    // weights; they stay inside the wave's magnitude planes + 128 floats of slack that are kept finite.
    const int f_begin = (blockIdx.x * MEL_WAVES + wave) * frames_per_wave;
    const int f_end = min(f_begin + frames_per_wave, total_frames);
    if (f_begin >= f_end) return;
    const int start0 = -cfg.n_fft / 2 + (cfg.n_fft - cfg.win) / 2;   // first windowed sample of frame 0
    // magnitudes are needed for K < n_bins (the last bin any band weights), i.e. plane entries k = K >> 2 < kmax; entries
    // kmax .. mag_stride - 1 keep whatever finite values they hold (zeros / FFT exchange data) and meet zero weights only
    // (a compile-time constant in the FB instantiations, which the launcher selects BY n_bins: the per-group guards below
    // then fold away and the eight magnitude groups of a transform become one basic block the scheduler can interleave)
    const int kmax = FB == 1 ? (1707 + 3) / 4 : FB == 2 ? (683 + 3) / 4 : (cfg.n_bins + 3) >> 2;
    constexpr int ZG = FB != 0 ? 8 : 2, MG = FB != 0 ? 4 : 1;                   // magnitude groups whose partners are fetched ahead (registers: the generic
                                                         // instantiation sits at the 168-register limit of three waves per SIMD)

    int b = find_segment(frame_off, n_clips, f_begin);
    float runmax = -3.0e38f;
    bool fb_const = FB != 0;                                // wave-uniform: the table matches the compile-time trip counts
    if (FB != 0) {
#pragma unroll
        for (int ps = 0; ps < 12; ++ps) fb_const = fb_const && (__builtin_amdgcn_readfirstlane(band_len[4 * ps]) >> 4) == MEL_FB_NIT[FB][ps];
    }

    // raw (unwindowed) samples of frame f, reflect-padded like np.pad(mode='reflect')
    auto load_frame = [&](int f, int bb, int quarter, float (&raw)[8][2]) {
        const int64_t c0 = clip_off[bb];
        const int L = (int)(clip_off[bb + 1] - c0);
        const T* y = pcm + c0;
        const int s0 = (f - frame_off[bb]) * cfg.hop + start0 + 1024 * quarter;
        if (s0 >= 0 && s0 + 1024 <= L) {              // interior frame (wave-uniform): plain coalesced loads
            const T* q = y + s0 + 2 * lane;
            if (PCM16 && ((c0 + s0) & 1) == 0) {      // sample pairs as one aligned dword
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const int v = *(const int*)(q + 128 * a);
                    raw[a][0] = (float)(short)(v & 0xffff);
                    raw[a][1] = (float)(v >> 16);
                }
            } else {
#pragma unroll
                for (int a = 0; a < 8; ++a) { raw[a][0] = (float)q[128 * a]; raw[a][1] = (float)q[128 * a + 1]; }
            }
        } else {
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    int i = s0 + 2 * (lane + 64 * a) + e;
                    i = i < 0 ? -i : i;
                    i = i >= L ? 2 * (L - 1) - i : i;
                    i = min(max(i, 0), L - 1);
                    raw[a][e] = (float)y[i];
                }
        }
    };

    float raw[8][2];
    if (NQ == 1) load_frame(f_begin, b, 0, raw);
    NQ_SUM_BEGIN();
    for (int f = f_begin; f < f_end; ++f) {
        c32 zq[NQ][8];
        const int fn = f + 1;
        int bn = b;
        if (fn < f_end)
            while (fn >= frame_off[bn + 1]) ++bn;
        if (NQ == 1) {
#pragma unroll
            for (int a = 0; a < 8; ++a) { const c32 wv = *(const c32*)(tab_w + 2 * (lane + 64 * a)); zq[0][a] = cmk(raw[a][0] * wv.x, raw[a][1] * wv.y); }
            if (fn < f_end) load_frame(fn, bn, 0, raw);          // prefetch the next frame (clip may change)
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {                        // long windows: no prefetch, taps read through L1
                load_frame(f, b, q, raw);
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const int m = 1024 * q + 2 * (lane + 64 * a);
                    zq[q][a] = cmk(m < cfg.win ? raw[a][0] * (window[m] * scale) : 0.f,
                                   m + 1 < cfg.win ? raw[a][1] * (window[m + 1] * scale) : 0.f);
                }
            }
        }
        // fold the quarters: z_r[n] = sum_q (-i)^(q r) z[n + 512 q]
        auto fold = [&](int r, c32 (&o)[8]) {
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                c32 acc = zq[0][a];
#pragma unroll
                for (int q = 1; q < NQ; ++q) {
                    const int e = (q * r) & 3;
                    const c32 v = zq[q][a];
                    acc = e == 0 ? cadd(acc, v) : e == 1 ? cadd_mi(acc, v) : e == 2 ? csub(acc, v) : csub_mi(acc, v);
                }
                o[a] = acc;
            }
        };
        c32 z[8];

        c32 u[8], u1[8];
        const int mir = 63 - lane;
        NQ_SUM(0);                                   // window + prefetch
        // r = 0: partner Z_0[512 - k] = lane (64 - l) & 63, register 7 - q2 (lane 0: register (8 - q2) & 7)
        fold(0, z);
        fft512<0>(u, z, tw, exch, lane, tab_a);
        {
            const int src = (64 - lane) & 63;
            // partners first (ZG groups ahead), then the magnitude chains two groups per call: the dependent chains of packed
            // operations interleave instead of running one behind the other with s_nop between their links (round 3)
#pragma unroll
            for (int q0 = 0; q0 < 8; q0 += ZG) {
                c32 zb[ZG];
#pragma unroll
                for (int e = 0; e < ZG; ++e) {
                    const int q2 = q0 + e;
                    if (64 * q2 >= kmax) continue;         // wave-uniform: bins above fmax are never produced
                    zb[e] = shfl_c(u[7 - q2], src);
                    if (lane == 0) zb[e] = u[(8 - q2) & 7];
                }
#pragma unroll
                for (int e = 0; e < ZG; e += 2) {
                    const int q2 = q0 + e;
                    if (64 * q2 >= kmax) continue;
                    const c32 wl = tab_d[0 * 64 + lane];
                    if (64 * (q2 + 1) < kmax) {
                        float m0, m1;
                        xmag2(u[q2], zb[e], wl, cmk(W16C[q2], W16S[q2]), u[q2 + 1], zb[e + 1], wl, cmk(W16C[q2 + 1], W16S[q2 + 1]), m0, m1);
                        if (lane + 64 * q2 < kmax) mag[0 * mag_stride + lane + 64 * q2] = m0;
                        if (lane + 64 * (q2 + 1) < kmax) mag[0 * mag_stride + lane + 64 * (q2 + 1)] = m1;
                    } else if (lane + 64 * q2 < kmax)
                        mag[0 * mag_stride + lane + 64 * q2] = xmag(u[q2], zb[e], wl, cmk(W16C[q2], W16S[q2]));
                }
            }
            if (lane == 0 && 512 < mag_stride) mag[512] = 2.0f * fabsf(u[0].x - u[0].y);   // Nyquist bin: 2 X[2048] = 2 (Re Z0 - Im Z0)

        }
        NQ_SUM(1);                                   // FFT r = 0 + magnitudes
        // r = 2: partner Z_2[511 - k] = lane 63 - l, register 7 - q2
        fold(2, z);
        fft512<2>(u, z, tw, exch, lane, tab_a);
#pragma unroll
        for (int q0 = 0; q0 < 8; q0 += ZG) {
            c32 zb[ZG];
#pragma unroll
            for (int e = 0; e < ZG; ++e)
                if (64 * (q0 + e) < kmax) zb[e] = shfl_c(u[7 - q0 - e], mir);
#pragma unroll
            for (int e = 0; e < ZG; e += 2) {
                const int q2 = q0 + e;
                if (64 * q2 >= kmax) continue;
                const c32 wl = tab_d[2 * 64 + lane];
                if (64 * (q2 + 1) < kmax) {
                    float m0, m1;
                    xmag2(u[q2], zb[e], wl, cmk(W16C[q2], W16S[q2]), u[q2 + 1], zb[e + 1], wl, cmk(W16C[q2 + 1], W16S[q2 + 1]), m0, m1);
                    if (lane + 64 * q2 < kmax) mag[1 * mag_stride + lane + 64 * q2] = m0;
                    if (lane + 64 * (q2 + 1) < kmax) mag[1 * mag_stride + lane + 64 * (q2 + 1)] = m1;
                } else if (lane + 64 * q2 < kmax)
                    mag[1 * mag_stride + lane + 64 * q2] = xmag(u[q2], zb[e], wl, cmk(W16C[q2], W16S[q2]));
            }
        }
        NQ_SUM(2);                                   // FFT r = 2 + magnitudes
        // r = 1 and r = 3 are each other's partners
        fold(1, z);
        fft512<1>(u1, z, tw, exch, lane, tab_a);
        fold(3, z);
        fft512<3>(u, z, tw, exch, lane, tab_a);
        NQ_SUM(3);                                   // FFTs r = 1, 3
#pragma unroll
        for (int q0 = 0; q0 < 8; q0 += MG) {               // MG groups at a time: 2 MG partners, then 2 MG chains
            c32 z3m[MG], z1m[MG];
#pragma unroll
            for (int e = 0; e < MG; ++e)
                if (64 * (q0 + e) < kmax) { z3m[e] = shfl_c(u[7 - q0 - e], mir); z1m[e] = shfl_c(u1[7 - q0 - e], mir); }
#pragma unroll
            for (int e = 0; e < MG; ++e) {
                const int q2 = q0 + e;
                if (64 * q2 >= kmax) continue;
                const c32 w16 = cmk(W16C[q2], W16S[q2]);
                float m1_, m3_;
                xmag2(u1[q2], z3m[e], tab_d[1 * 64 + lane], w16, u[q2], z1m[e], tab_d[3 * 64 + lane], w16, m1_, m3_);
                if (lane + 64 * q2 < kmax) {
                    mag[2 * mag_stride + lane + 64 * q2] = m1_;
                    mag[3 * mag_stride + lane + 64 * q2] = m3_;
                }
            }
        }
        MEL_WBAR();
        NQ_SUM(4);                                   // magnitudes r = 1, 3

        // ---- sparse slaney filterbank: 4 bands per pass (one per 16-lane row)
        float mine = 0.f;
        if (FB != 0 && fb_const) {
            float part[12];
#pragma unroll
            for (int ps = 0; ps < 12; ++ps) {
                const int2 bo = *(const int2*)(tab_band + 2 * (64 * ps + lane));
                const float* wp = (const float*)((const char*)wlds + bo.y);
                const float* mp = (const float*)((const char*)mag + bo.x);
                float acc_ = 0.f;
#pragma unroll
                for (int it = 0; it < MEL_FB_NIT[FB][ps]; ++it) acc_ = fmaf(wp[16 * it], mp[4 * it], acc_);
                part[ps] = acc_;
            }
#pragma unroll
            for (int ps = 0; ps < 12; ++ps) part[ps] = row16_sum(part[ps]);
#pragma unroll
            for (int ps = 0; ps < 12; ++ps)
                if (l16 == ps) mine = part[ps];             // lane 16*row + ps holds band 4*ps + row
        } else
#pragma unroll
        for (int ps = 0; ps < 12; ++ps) {
            // padded length is the same for the 4 bands of a pass (zero weights beyond a band's support)
            const int nit = __builtin_amdgcn_readfirstlane(band_len[4 * ps]) >> 4;
            float part = 0.f;
            const int2 bo = *(const int2*)(tab_band + 2 * (64 * ps + lane));
            const float* wp = (const float*)((const char*)wlds + bo.y);
            const float* mp = (const float*)((const char*)mag + bo.x);
#pragma unroll 4
            for (int it = 0; it < nit; ++it) part = fmaf(wp[16 * it], mp[4 * it], part);
            part = row16_sum(part);
            if (l16 == ps) mine = part;                 // lane 16*row + ps holds band 4*ps + row
        }
        MEL_WBAR();
        NQ_SUM(5);                                   // filterbank
        // ---- amplitude_to_db(ref=1, amin=1e-4): 10*log10(max(amin^2, S^2)); running per-clip max
        if (l16 < 12) {
            const float db = 10.0f * log10f(fmaxf(cfg.amin_sq, mine * mine));
            mel_tm[(size_t)f * NISQA_N_MELS + 4 * l16 + row] = db;
            runmax = fmaxf(runmax, db);
        }
        if (bn != b || fn >= f_end) {                   // wave-uniform: publish this clip's maximum
            const float m = wave_max(runmax);
            if (lane == 0) atomicMax(clip_max_enc + b, enc_ordered(m));
            runmax = -3.0e38f;
            b = bn;
        }
        NQ_SUM(6);                                   // dB, store, clip maximum
    }
    NQ_SUM_COUNT(8, f_end - f_begin);
    NQ_SUM_END(g_mel_clk, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane == 0);
}

__global__ void mel_floor_kernel(const uint32_t* __restrict__ clip_max_enc, float top_db, int n_clips,
                                 float* __restrict__ clip_floor) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_clips) clip_floor[b] = dec_ordered(clip_max_enc[b]) - top_db;
}

__global__ void mel_clamp_kernel(float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off, int n_clips,
                                 int total_frames, const float* __restrict__ clip_floor) {
    const int f = blockIdx.x;
    const int b = find_segment(frame_off, n_clips, f);
    const float fl = clip_floor[b];
    if (threadIdx.x < NISQA_N_MELS) {
        float* p = mel_tm + (size_t)f * NISQA_N_MELS + threadIdx.x;
        *p = fmaxf(*p, fl);
    }
}

__global__ void pcm16_kernel(const int16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (float)in[i] * (1.0f / 32768.0f);
}

template <typename T>
static int mel_db_launch(const T* pcm, const int64_t* clip_off, const int32_t* frame_off,
                         int32_t n_clips, int32_t total_frames, const nisqa_mel_cfg* cfg,
                         const float* window, const float* twiddle, const int32_t* band_start,
                         const int32_t* band_len, const int32_t* band_woff, const float* band_w,
                         float* mel_tm, uint32_t* clip_max_enc, void* stream) {
    if (!cfg || cfg->n_fft != NISQA_N_FFT || cfg->n_mels != NISQA_N_MELS || cfg->win < 2 || cfg->win > 4096 ||
        cfg->hop < 1 || cfg->n_bins < 1 || cfg->n_bins > 2049 || cfg->w_floats < 1 || cfg->w_floats > 8192 ||
        n_clips <= 0 || total_frames <= 0)
        return NISQA_ERR_ARG;
    const int w_floats = cfg->w_floats;
    NQ_LAUNCH_BEGIN();
    // LDS: shared band weights + per wave (FFT exchange 5120 B + magnitudes 4 planes x mag_stride floats);
    // mag_stride >= ceil(n_bins/4) and == 8 (mod 32) keeps the band-sum reads bank-conflict free
    int mag_stride = (cfg->n_bins + 3) / 4;
    mag_stride += (8 - (mag_stride & 31) + 32) & 31;
    const int w_bytes = (w_floats * 4 + 15) & ~15;
    const int nq = cfg->win <= 1024 ? 1 : (cfg->win <= 2048 ? 2 : 4);
    const int waves = MEL_WAVES_OF(nq);
    const int exch_extra = MEL_EXCH_BYTES - 8 * mag_stride > 0 ? MEL_EXCH_BYTES - 8 * mag_stride : 0;
    const size_t lds = (size_t)w_bytes + MEL_TAB_BYTES + waves * (mag_stride * 16 + exch_extra + 512);
    if (lds > 160 * 1024) return NISQA_ERR_ARG;      // the slaney bank of any (sr, fmax) needs <= 150.5 KB; a denser table does not fit
    // frames a wave walks over: more frames amortise its per-lane twiddle loads and the workgroup's table fills, but the
    // chip holds 256 x (12, 8 or 4) waves of this kernel: enough frames per wave to cover the batch in one round, 4 to 32
    static const int fpw_env = [] {
        const char* e = getenv("NISQA_MEL_FPW");
        return e && atoi(e) > 0 ? atoi(e) : 0;
    }();
    const int resident = 256 * (nq == 1 ? 12 : (nq == 2 ? 8 : 4));
    // whole rounds of the resident waves: a batch that needs 1.3 rounds at 32 frames per wave takes as long as two
    // (measured: 64 x 10 s at 16 frames per wave = 1.3 rounds: 0.41 ms against 0.28 at 21 = one round)
    int frames_per_wave = fpw_env;
    if (!fpw_env) {
        const int64_t rounds = ((int64_t)total_frames + (int64_t)resident * 32 - 1) / ((int64_t)resident * 32);
        frames_per_wave = (int)(((int64_t)total_frames + resident * rounds - 1) / (resident * rounds));
        frames_per_wave = frames_per_wave < 4 ? 4 : (frames_per_wave > 32 ? 32 : frames_per_wave);
    }
    const int per_wg = waves * frames_per_wave;
    const dim3 grid((total_frames + per_wg - 1) / per_wg), block(64 * waves);
    // up to 150.5 KB of dynamic LDS (the band table + per-wave planes): every instantiation is opted in for the whole 160 KB once per device
    static std::atomic<bool> lds_ok[5][64];
    int which = 0, rc_attr = 0;
    auto go = [&](auto kernel) {
        if ((rc_attr = nq_lds_opt_in((const void*)kernel, 160 * 1024, lds_ok[which])) != 0) return;
        hipLaunchKernelGGL(kernel, grid, block, lds, (hipStream_t)stream, pcm, clip_off, frame_off, n_clips, total_frames,
                           frames_per_wave, *cfg, mag_stride, w_floats, window, (const float2*)twiddle, band_start,
                           band_len, band_woff, band_w, mel_tm, clip_max_enc);
    };
    // the two shipped front ends at 48 kHz get the instantiations with compile-time filter-bank trip counts (checked in the kernel)
    const int fb = getenv("NISQA_MEL_FB_GENERIC") ? 0 : (cfg->hop == 480 && cfg->win == 960 && cfg->n_bins == 1707) ? 1
                   : (cfg->hop == 480 && cfg->win == 960 && cfg->n_bins == 683) ? 2 : 0;
    if (cfg->win <= 1024 && fb == 1) { which = 0; go(mel_frame_kernel<1, T, 1>); }
    else if (cfg->win <= 1024 && fb == 2) { which = 1; go(mel_frame_kernel<1, T, 2>); }
    else if (cfg->win <= 1024) { which = 2; go(mel_frame_kernel<1, T>); }
    else if (cfg->win <= 2048) { which = 3; go(mel_frame_kernel<2, T>); }
    else { which = 4; go(mel_frame_kernel<4, T>); }
    if (rc_attr) return rc_attr;
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_mel_db(const float* pcm, const int64_t* clip_off, const int32_t* frame_off,
                            int32_t n_clips, int32_t total_frames, const nisqa_mel_cfg* cfg,
                            const float* window, const float* twiddle, const int32_t* band_start,
                            const int32_t* band_len, const int32_t* band_woff, const float* band_w,
                            float* mel_tm, uint32_t* clip_max_enc, void* stream) {
    return mel_db_launch(pcm, clip_off, frame_off, n_clips, total_frames, cfg, window, twiddle, band_start, band_len,
                         band_woff, band_w, mel_tm, clip_max_enc, stream);
}

extern "C" int nisqa_mel_db_pcm16(const int16_t* pcm, const int64_t* clip_off, const int32_t* frame_off,
                                  int32_t n_clips, int32_t total_frames, const nisqa_mel_cfg* cfg,
                                  const float* window, const float* twiddle, const int32_t* band_start,
                                  const int32_t* band_len, const int32_t* band_woff, const float* band_w,
                                  float* mel_tm, uint32_t* clip_max_enc, void* stream) {
    return mel_db_launch(pcm, clip_off, frame_off, n_clips, total_frames, cfg, window, twiddle, band_start, band_len,
                         band_woff, band_w, mel_tm, clip_max_enc, stream);
}

extern "C" int nisqa_mel_finalize(float* mel_tm, const int32_t* frame_off, int32_t n_clips, int32_t total_frames,
                                  const uint32_t* clip_max_enc, float top_db, float* clip_floor,
                                  int32_t clamp_in_place, void* stream) {
    if (n_clips <= 0 || total_frames <= 0) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(mel_floor_kernel, dim3((n_clips + 63) / 64), dim3(64), 0, (hipStream_t)stream, clip_max_enc,
                       top_db, n_clips, clip_floor);
    if (clamp_in_place)
        hipLaunchKernelGGL(mel_clamp_kernel, dim3(total_frames), dim3(64), 0, (hipStream_t)stream, mel_tm, frame_off,
                           n_clips, total_frames, clip_floor);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_pcm16_to_f32(const int16_t* pcm16, float* pcm, int64_t n, void* stream) {
    if (n <= 0) return NISQA_ERR_ARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(pcm16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pcm16, pcm, n);
    return NQ_LAUNCH_STATUS();
}
