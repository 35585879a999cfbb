// Segment-resident 3 x 3 convolutions of the training step on split-bf16 MFMA (include/nisqa_train.h: nisqa_segconv_*):
// forward z = conv(x, w) + b and input gradient dx = conv^T(dz, w) of the AdaptCNN layers 2..6 (reference
// nisqa/NISQA_lib.py:688-710 in train mode; NISQA_model.py:131-152 drives forward + backward).
//
// The implicit GEMMs of train.hip gather every K-tile of the patch matrix from HBM / L2 with index arithmetic, split it
// into bf16 hi + lo on the way into LDS and pay two workgroup barriers per 32 k: each activation is fetched and split nine
// times (once per tap), and the matrix pipe waits for the gathers.  Train-mode BatchNorm forbids carrying a
// segment through several layers, but ONE layer can still be done the way the inference kernel does it (cnn_bf16.hip):
//   * a workgroup of four waves owns SEGS consecutive segments: their activations ([S][H*W][C] fp32, contiguous) are read
//     ONCE with 128-bit loads, split once, and kept as two bf16 planes (hi, lo), pixel rows padded by 16 bytes;
//   * the output pixels of those segments are the M rows, 32 per tile, MT tiles per wave; the K loop is conv_k_bf16
//     (conv_bf16.hpp): A fragments are ds_read_b128 at lane base + (tap, channel-step) immediates, out-of-image taps go to
//     a shared zero block by a lane-static 9-bit mask, no barrier inside the loop;
//   * weight fragments stream from L2 through a buffer descriptor into a 3-deep register ring; they are packed (and split)
//     once per optimiser step by segconv_pack_kernel -- 0.4 MB for all five layers and both directions;
//   * the input gradient is the same kernel: a convolution of dz with the taps mirrored and the weight matrix transposed
//     (both folded into the packing), and horizontal padding 2 - pad_w.
// Forward epilogue: + bias, z to HBM, and the BatchNorm batch statistics (sum z, sum z^2 per channel, float64) ride along
// exactly as in conv_gemm_bf16_kernel.
#include <stdlib.h>
#include "common.hpp"
// one scheduling fence per K-step of conv_k_bf16 in THIS unit: the two workgroups of a CU run in step here, nobody fills the
// stalls of a sunk prefetch (measured -9 % forward, -15 % input gradient; the inference kernel, whose waves are out of step,
// is faster without: conv_bf16.hpp)
#define NQ_SB 1
#include "conv_bf16.hpp"
#include "../../include/nisqa_hip.h"
#include "../../include/nisqa_train.h"

#define SC_ZADDR 2048u                     /* 128-byte zero block above the largest tap offset */
#define SC_BASE 2176u                      /* hi plane; the lo plane follows */
// per-phase clock of the segment-conv kernels (tools/bench_segconv.py; empty macros unless built with -DNQ_EXPERIMENTAL): phases 0..5, [6] = groups, [7] = waves
NQ_CLK_EXPORT(g_sc_clk, nisqa_debug_segconv_clock)
#define SC_W_MSPLIT 1         /* weight gradient of the 64 -> 64 layers: 1 = a wave takes both M tiles and every 8th N tile, 2 = one M tile, every 4th */
#define SC_SEGS60 4            /* segments per workgroup of the 12 x 5 layers (two 32-row tiles per wave at 4, one at 2) */
#define SC_WGS 2
#define SC_RING 3            /* weight-fragment ring of the K loop (slots) */
#define SC_FENCE() __builtin_amdgcn_sched_barrier(0)

// ---- fragment packing: [step g][NT][hi, lo][64 lanes][8 bf16];  step g = (tap, 16-channel group), lane l holds column
//      n = 32 nt + (l & 31) and reduction channels 16 (g % S16) + 8 (l >> 5) .. + 7 of that tap
//   mode 0 (forward):  B[(tap, ci)][n = co] = w[co][tap * CI + ci]
//   mode 1 (dgrad):    B[(tap, co)][n = ci] = w[co][(8 - tap) * CI + ci]
__global__ __launch_bounds__(256) void segconv_pack_kernel(const float* __restrict__ w, int ci, int co, int mode,
                                                           unsigned short* __restrict__ out) {
    const int kc = mode ? co : ci, n_real = mode ? ci : co;
    const int s16 = kc / 16, nt_n = (n_real + 31) / 32;
    const int total = 9 * s16 * nt_n * 64 * 8;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int e = i & 7, lane = (i >> 3) & 63, nt = (i >> 9) % nt_n, g = (i >> 9) / nt_n;
        const int tap = g / s16, c = 16 * (g % s16) + 8 * (lane >> 5) + e, n = 32 * nt + (lane & 31);
        float v = 0.f;
        if (n < n_real) v = mode ? w[(size_t)c * (9 * ci) + (8 - tap) * ci + n] : w[(size_t)n * (9 * ci) + tap * ci + c];
        const unsigned hi = bf16_bits(v);
        const unsigned lo = bf16_bits(v - bf16_val(hi));
        const size_t o = (((size_t)(g * nt_n + nt) * 2) * 64 + lane) * 8 + e;
        out[o] = (unsigned short)hi;
        out[o + 512] = (unsigned short)lo;
    }
}

// the same for up to ten (layer, mode) jobs in ONE launch (blockIdx.y = job): the training step packs five layers x two
// directions per optimiser step, and ten launches of ~5 us each were 1.3 % of it
// (terms = 3: hi / mid / lo fragments [step g][NT][3][64 lanes][8 bf16] of the three-term kernels -- an fp32 weight exactly)
struct segconv_pack_jobs { const float* w[10]; unsigned short* out[10]; int ci[10], co[10], mode[10]; int terms; };
__global__ __launch_bounds__(256) void segconv_pack_many_kernel(segconv_pack_jobs jobs) {
    const int j = blockIdx.y;
    const float* __restrict__ w = jobs.w[j];
    unsigned short* __restrict__ out = jobs.out[j];
    const int ci = jobs.ci[j], co = jobs.co[j], mode = jobs.mode[j], terms = jobs.terms;
    const int kc = mode ? co : ci, n_real = mode ? ci : co;
    const int s16 = kc / 16, nt_n = (n_real + 31) / 32;
    const int total = 9 * s16 * nt_n * 64 * 8;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int e = i & 7, lane = (i >> 3) & 63, nt = (i >> 9) % nt_n, g = (i >> 9) / nt_n;
        const int tap = g / s16, c = 16 * (g % s16) + 8 * (lane >> 5) + e, n = 32 * nt + (lane & 31);
        float v = 0.f;
        if (n < n_real) v = mode ? w[(size_t)c * (9 * ci) + (8 - tap) * ci + n] : w[(size_t)n * (9 * ci) + tap * ci + c];
        const size_t o = (((size_t)(g * nt_n + nt) * terms) * 64 + lane) * 8 + e;
        for (int t = 0; t < terms; ++t) {
            const unsigned b = bf16_bits(v);
            out[o + 512 * t] = (unsigned short)b;
            v -= bf16_val(b);
        }
    }
}

// ---- f16 formats (precision mode 'f16x4'): fragments of W * 2^kw as f16 hi + lo in the two-term layout, kw from the layer's largest
//      |W| of THIS optimiser step (the largest lands in [2^14, 2^15)), stored as one int32 behind the fragments (the buffer is 16 bytes
//      longer): segconv_wmax_kernel writes it, the packer and the convolution read it
__global__ __launch_bounds__(256) void segconv_wmax_kernel(segconv_pack_jobs jobs, int frag_u16_unused) {
    const int j = blockIdx.x;
    const float* __restrict__ w = jobs.w[j];
    const int n = jobs.ci[j] * jobs.co[j] * 9;
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
    m = wave_max(m);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const int kc = jobs.mode[j] ? jobs.co[j] : jobs.ci[j], n_real = jobs.mode[j] ? jobs.ci[j] : jobs.co[j];
        const size_t u16s = (size_t)9 * (kc / 16) * ((n_real + 31) / 32) * 2 * 512;
        *(int*)(jobs.out[j] + u16s) = (m > 0.f && m < 3.0e38f) ? f16_scale_exp(m) : 0;
    }
}
__global__ __launch_bounds__(256) void segconv_pack_f16_many_kernel(segconv_pack_jobs jobs) {
    const int j = blockIdx.y;
    const float* __restrict__ w = jobs.w[j];
    unsigned short* __restrict__ out = jobs.out[j];
    const int ci = jobs.ci[j], co = jobs.co[j], mode = jobs.mode[j];
    const int kc = mode ? co : ci, n_real = mode ? ci : co;
    const int s16 = kc / 16, nt_n = (n_real + 31) / 32;
    const int total = 9 * s16 * nt_n * 64 * 8;
    const float sc = pow2_f32(*(const int*)(out + (size_t)total * 2));
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int e = i & 7, lane = (i >> 3) & 63, nt = (i >> 9) % nt_n, g = (i >> 9) / nt_n;
        const int tap = g / s16, c = 16 * (g % s16) + 8 * (lane >> 5) + e, n = 32 * nt + (lane & 31);
        float v = 0.f;
        if (n < n_real) v = mode ? w[(size_t)c * (9 * ci) + (8 - tap) * ci + n] : w[(size_t)n * (9 * ci) + tap * ci + c];
        v *= sc;
        const size_t o = (((size_t)(g * nt_n + nt) * 2) * 64 + lane) * 8 + e;
        const unsigned hi = cvt_pk_f16(v, 0.f) & 0xffffu;
        const float hf = (float)__builtin_bit_cast(f16x2_t, hi)[0];
        out[o] = (unsigned short)hi;
        out[o + 512] = (unsigned short)(cvt_pk_f16(v - hf, 0.f) & 0xffffu);
    }
}

extern "C" int64_t nisqa_segconv_frag_bytes(int32_t mode, int32_t ci, int32_t co) {
    if (mode < 0 || mode > 1 || ci < 16 || co < 16 || (ci & 15) || (co & 15)) return -1;
    const int kc = mode ? co : ci, n_real = mode ? ci : co;
    return (int64_t)9 * (kc / 16) * ((n_real + 31) / 32) * 2 * 1024;
}

extern "C" int nisqa_segconv_pack(int32_t mode, const float* w, int32_t ci, int32_t co, uint16_t* frags, void* stream) {
    if (!w || !frags || nisqa_segconv_frag_bytes(mode, ci, co) < 0) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    const int total = (int)(nisqa_segconv_frag_bytes(mode, ci, co) / 4);          // elements of one plane pair / 2
    hipLaunchKernelGGL(segconv_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, ci, co, mode, frags);
    return NQ_LAUNCH_STATUS();
}

extern "C" int64_t nisqa_segconv_frag_bytes_f16(int32_t mode, int32_t ci, int32_t co) {
    const int64_t b = nisqa_segconv_frag_bytes(mode, ci, co);
    return b < 0 ? b : b + 16;                                  // + the layer's scale exponent (int32) behind the fragments
}
extern "C" int nisqa_segconv_pack_f16_many(int32_t n_jobs, const int32_t* modes, const float* const* w, const int32_t* ci,
                                           const int32_t* co, uint16_t* const* frags, void* stream) {
    if (n_jobs < 1 || n_jobs > 10 || !modes || !w || !ci || !co || !frags) return NISQA_ERR_ARG;
    segconv_pack_jobs jobs = {};
    jobs.terms = 2;
    for (int j = 0; j < n_jobs; ++j) {
        if (!w[j] || !frags[j] || nisqa_segconv_frag_bytes(modes[j], ci[j], co[j]) < 0) return NISQA_ERR_ARG;
        jobs.w[j] = w[j]; jobs.out[j] = frags[j]; jobs.ci[j] = ci[j]; jobs.co[j] = co[j]; jobs.mode[j] = modes[j];
    }
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(segconv_wmax_kernel, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, jobs, 0);
    hipLaunchKernelGGL(segconv_pack_f16_many_kernel, dim3(36, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_segconv_pack_many(int32_t n_jobs, const int32_t* modes, const float* const* w, const int32_t* ci,
                                       const int32_t* co, uint16_t* const* frags, void* stream) {
    if (n_jobs < 1 || n_jobs > 10 || !modes || !w || !ci || !co || !frags) return NISQA_ERR_ARG;
    segconv_pack_jobs jobs = {};
    jobs.terms = 2;
    for (int j = 0; j < n_jobs; ++j) {
        if (!w[j] || !frags[j] || nisqa_segconv_frag_bytes(modes[j], ci[j], co[j]) < 0) return NISQA_ERR_ARG;
        jobs.w[j] = w[j]; jobs.out[j] = frags[j]; jobs.ci[j] = ci[j]; jobs.co[j] = co[j]; jobs.mode[j] = modes[j];
    }
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(segconv_pack_many_kernel, dim3(36, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return NQ_LAUNCH_STATUS();
}

// three-term fragments (nisqa_segconv_bf16x6): 1.5 x the bytes
extern "C" int64_t nisqa_segconv_frag_bytes_x6(int32_t mode, int32_t ci, int32_t co) {
    const int64_t b = nisqa_segconv_frag_bytes(mode, ci, co);
    return b < 0 ? b : b / 2 * 3;
}
extern "C" int nisqa_segconv_pack_x6_many(int32_t n_jobs, const int32_t* modes, const float* const* w, const int32_t* ci,
                                          const int32_t* co, uint16_t* const* frags, void* stream) {
    if (n_jobs < 1 || n_jobs > 10 || !modes || !w || !ci || !co || !frags) return NISQA_ERR_ARG;
    segconv_pack_jobs jobs = {};
    jobs.terms = 3;
    for (int j = 0; j < n_jobs; ++j) {
        if (!w[j] || !frags[j] || nisqa_segconv_frag_bytes(modes[j], ci[j], co[j]) < 0) return NISQA_ERR_ARG;
        jobs.w[j] = w[j]; jobs.out[j] = frags[j]; jobs.ci[j] = ci[j]; jobs.co[j] = co[j]; jobs.mode[j] = modes[j];
    }
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(segconv_pack_many_kernel, dim3(36, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return NQ_LAUNCH_STATUS();
}

// ---- exact fp32 variant (precision modes 'f32' and 'mixed' forward, 'f32' input gradient): the same kernel with fp32 planes
// (pixel rows of CIN floats padded by 16 bytes: a lane group of a ds_read_b128 lands on 16 distinct 16-byte slots for CIN =
// 16, 32, 64), v_mfma_f32_32x32x2_f32 and fp32 fragments [step g = (tap, 8-channel group)][NT][64 lanes][4]: lane l holds
// column n = 32 nt + (l & 31) and reduction channels 8 (g % S8) + 4 (l >> 5) .. + 3 of that tap.  A K-step is 8 channels =
// 4 MFMAs per (M tile, N tile): one ds_read_b128 per M tile, one 16-byte buffer load per N tile.
#define SCF_ZADDR 4096u                    /* 256-byte zero block above the largest tap offset ((2 W + 2) rows of 272 bytes) */
#define SCF_BASE 4352u
template <int CIN, int MT, int NT, int W, int RS, unsigned ZADDR, int RING>
NQ_DEV void conv_k_f32(f32x16 (&acc)[MT][NT], __amdgpu_buffer_rsrc_t rsrc, unsigned lane16, const unsigned (&base)[MT],
                       const unsigned (&m9)[MT]) {
    constexpr int S8 = CIN / 8, TOTAL = 9 * S8;
    static_assert(ZADDR >= (2 * W + 2) * RS + 32 * S8, "zero block must sit above the largest tap offset");
    f32x4 b[RING][NT], a[2][MT];
    unsigned a_ad[MT];
    auto load_b = [&](int g, int slot) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[slot][nt] = wfrag_load(rsrc, lane16, (g * NT + nt) * 1024);
    };
    auto load_a = [&](int g, int slot) {
        const int tap = g / S8, s = g - tap * S8;
        const int tapoff = ((tap / 3) * W + tap % 3) * RS;
        if (s == 0) {
#pragma unroll
            for (int t = 0; t < MT; ++t) a_ad[t] = ((m9[t] >> tap) & 1u) ? base[t] : ZADDR - tapoff;
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) a[slot][t] = lds_ld128(a_ad[t] + tapoff + 32 * s);
    };
#pragma unroll
    for (int g = 0; g < RING - 1; ++g) load_b(g, g);
    load_a(0, 0);
#pragma unroll
    for (int g = 0; g < TOTAL; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + RING - 1 < TOTAL) load_b(g + RING - 1, (g + RING - 1) % RING);
        if (g + 1 < TOTAL) load_a(g + 1, (g + 1) & 1);
        const int sa = g & 1, sb = g % RING;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[t][nt] = mfma32(a[sa][t][kk], b[sb][nt][kk], acc[t][nt]);
    }
}

struct segconv_pack_f32_jobs { const float* w[10]; float* out[10]; int ci[10], co[10], mode[10]; };
__global__ __launch_bounds__(256) void segconv_pack_f32_many_kernel(segconv_pack_f32_jobs jobs) {
    const int j = blockIdx.y;
    const float* __restrict__ w = jobs.w[j];
    float* __restrict__ out = jobs.out[j];
    const int ci = jobs.ci[j], co = jobs.co[j], mode = jobs.mode[j];
    const int kc = mode ? co : ci, n_real = mode ? ci : co;
    const int s8 = kc / 8, nt_n = (n_real + 31) / 32;
    const int total = 9 * s8 * nt_n * 256;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int kk = i & 3, lane = (i >> 2) & 63, nt = (i >> 8) % nt_n, g = (i >> 8) / nt_n;
        const int tap = g / s8, c = 8 * (g % s8) + 4 * (lane >> 5) + kk, n = 32 * nt + (lane & 31);
        float v = 0.f;
        if (n < n_real) v = mode ? w[(size_t)c * (9 * ci) + (8 - tap) * ci + n] : w[(size_t)n * (9 * ci) + tap * ci + c];
        out[i] = v;
    }
}
extern "C" int64_t nisqa_segconv_frag_bytes_f32(int32_t mode, int32_t ci, int32_t co) {
    if (mode < 0 || mode > 1 || ci < 16 || co < 16 || (ci & 15) || (co & 15)) return -1;
    const int kc = mode ? co : ci, n_real = mode ? ci : co;
    return (int64_t)9 * (kc / 8) * ((n_real + 31) / 32) * 1024;
}
extern "C" int nisqa_segconv_pack_f32_many(int32_t n_jobs, const int32_t* modes, const float* const* w, const int32_t* ci,
                                           const int32_t* co, float* const* frags, void* stream) {
    if (n_jobs < 1 || n_jobs > 10 || !modes || !w || !ci || !co || !frags) return NISQA_ERR_ARG;
    segconv_pack_f32_jobs jobs = {};
    for (int j = 0; j < n_jobs; ++j) {
        if (!w[j] || !frags[j] || nisqa_segconv_frag_bytes_f32(modes[j], ci[j], co[j]) < 0) return NISQA_ERR_ARG;
        jobs.w[j] = w[j]; jobs.out[j] = frags[j]; jobs.ci[j] = ci[j]; jobs.co[j] = co[j]; jobs.mode[j] = modes[j];
    }
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(segconv_pack_f32_many_kernel, dim3(36, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return NQ_LAUNCH_STATUS();
}

// four consecutive channels of one pixel -> T bf16 terms each (round to nearest), 8 bytes per plane
template <int T>
NQ_DEV void sc_store_terms4(unsigned a, int plane, f32x4 v) {
    f32x2_t r0 = {v[0], v[1]}, r1 = {v[2], v[3]};
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const unsigned p0 = cvt_pk_bf16(r0[0], r0[1]), p1 = cvt_pk_bf16(r1[0], r1[1]);
        lds_st32(a + t * plane, p0);
        lds_st32(a + t * plane + 4, p1);
        if (t + 1 < T) {
            r0 = r0 - f32x2_t{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u)};
            r1 = r1 - f32x2_t{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
        }
    }
}
#define SC_X6_FENCE true     /* sched_barrier behind a K step's requests in the three-term loops (conv_k_terms) */

// CIN: channels of the staged tensor (the reduction runs over 9 x CIN); NT: 32-column tiles of the output channels
// HR x WR: output pixels of a segment (the rows); HS x WS: pixels of the staged tensor; source pixel of row (y, x) and
// tap (ty, tx) is (y + ty - 1, x + tx - PADX)
// TERMS (split-bf16 only): 2 = hi / lo planes and three products per term pair; 3 = hi / mid / lo planes, an exact split of the
// fp32 activations, and six products (conv_k_terms: the accuracy of the fp32 variant at 2.7 x its matrix-pipe rate)
// FMT (TERMS = 2 only): NQ_FMT_BF16X3 (default) or NQ_FMT_F16X4 -- the staged tensor of a workgroup's GROUP of segments as f16 hi + lo of
// x * 2^e, e from the group's own largest magnitude (measured while the values are in registers, exchanged between the four waves
// through the barrier the staging has anyway), all four term products; the output leaves as fp32 (acc * 2^-(e + kw) + bias), so no
// bound on the next tensor is needed here
#define SC_MAXSLOT 1024u                   /* four floats below the zero block: the waves' maxima of the group being staged */
template <int CIN, int NT, int NOUT, int HR, int WR, int HS, int WS, int PADX, int SEGS, int MT, bool FWD, bool F32 = false, int TERMS = 2, int FMT = NQ_FMT_BF16X3>
struct segconv_cfg {
    static constexpr int RS = F32 ? 4 * CIN + 16 : 2 * CIN + 16;
    static constexpr int PXS = HS * WS, PXR = HR * WR;
    static constexpr int PLANE = SEGS * PXS * RS;
    static constexpr unsigned BASE = F32 ? SCF_BASE : SC_BASE, ZADDR = F32 ? SCF_ZADDR : SC_ZADDR;
    static constexpr unsigned LDS = F32 ? BASE + PLANE : BASE + TERMS * PLANE;
    static constexpr int ROWS = SEGS * PXR;
    static constexpr int F4 = SEGS * PXS * CIN / 4;          // 128-bit groups of the workgroup's activations
    static constexpr int NV = (F4 + 255) / 256;
    static_assert(ROWS <= 4 * MT * 32, "rows of the workgroup's segments must fit its tiles");
    static_assert(BASE >= (unsigned)((WS + PADX) * RS), "tap (-1, -PADX) of pixel 0 must not address below 0");
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
};

template <int CIN, int NT, int NOUT, int HR, int WR, int HS, int WS, int PADX, int SEGS, int MT, bool FWD, bool F32 = false, int TERMS = 2, int FMT = NQ_FMT_BF16X3>
__global__ __launch_bounds__(256, SC_WGS) void segconv_bf16_kernel(
    const float* __restrict__ src, const unsigned short* __restrict__ frags, float* __restrict__ out, int n_segments,
    const float* __restrict__ bias, double* __restrict__ stats) {
    typedef segconv_cfg<CIN, NT, NOUT, HR, WR, HS, WS, PADX, SEGS, MT, FWD, F32, TERMS, FMT> C;
    constexpr bool F16 = FMT == NQ_FMT_F16X4;
    static_assert(!F16 || (!F32 && TERMS == 2), "the f16 format is a two-term form");
    // F16: the layer's weight-scale exponent sits behind the fragments (nisqa_segconv_pack_f16_many)
    const int kw = F16 ? *(const int*)(frags + (size_t)9 * (CIN / 16) * NT * 2 * 512) : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid0 = threadIdx.x, lane0 = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int n_groups = (n_segments + SEGS - 1) / SEGS;

    // ---- rows of this wave's tiles: lane-static geometry (a short last group is staged as zeros and masked in the epilogue)
    const bool active = (wave * MT) * 32 < C::ROWS;              // otherwise this wave's tiles are all padding
    unsigned base[MT], m9[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int r = (wave * MT + t) * 32 + (lane0 & 31);
        const bool valid = r < C::ROWS;
        const int sg = r / C::PXR, pix = r - sg * C::PXR, y = pix / WR, x = pix - y * WR;
        unsigned m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ys = y + tap / 3 - 1, xs = x + tap % 3 - PADX;
            m |= (valid && (unsigned)ys < (unsigned)HS && (unsigned)xs < (unsigned)WS) ? (1u << tap) : 0u;
        }
        m9[t] = m;
        base[t] = valid ? C::BASE + (unsigned)(((sg * HS + y - 1) * WS + (x - PADX)) * C::RS) + 16u * (lane0 >> 5)
                        : C::BASE + 16u * (lane0 >> 5);
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)frags, 0, F32 ? 9 * (CIN / 8) * NT * 1024 : 9 * (CIN / 16) * NT * TERMS * 1024, 0x00020000);
    if (tid0 < (F32 ? 64 : 32)) *(unsigned*)(smem + C::ZADDR + 4 * tid0) = 0u;  // through the symbol: the kernel must be seen to use LDS
    const bool with_stats = FWD && stats != nullptr;
    double s1[NT], s2[NT];                                       // this lane's column sums over all groups of the workgroup
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s1[nt] = s2[nt] = 0.0;
        const int col = 32 * nt + (lane0 & 31);
        bv[nt] = (FWD && bias && col < NOUT) ? bias[col] : 0.f;
    }

    // The workgroup walks over groups of SEGS segments.  The two workgroups of a CU run in step, so the loads, the K loop and
    // the stores of a group would overlap with nothing (measured: 120 us for a layer whose K loops need 50).  Registers for a
    // whole group of raw activations across the K loop are not to be had (60 more: the compiler spills), so the NEXT group is
    // only TOUCHED before the K loop -- one dword per 128-byte line, 2 registers -- which brings it from HBM into L2 while
    // the matrix pipe works; its real loads are issued right behind the K loop, ahead of the output stores, and hit L2.
    // Output stores are never waited for.
    constexpr int LINES = SEGS * C::PXS * CIN / 32;             // 128-byte lines of a group
    constexpr int NTOUCH = (LINES + 255) / 256;
    NQ_SUM_BEGIN();
    f32x4 v[C::NV];
    auto request = [&](int grp, int tid) {                      // the 128-bit loads of a group, all in flight together
        const int seg0 = grp * SEGS;
        const f32x4* g = (const f32x4*)(src + (size_t)seg0 * C::PXS * CIN);
        const int lim = grp < n_groups ? min(SEGS, n_segments - seg0) * C::PXS * CIN / 4 : 0;
#pragma unroll
        for (int j = 0; j < C::NV; ++j) {
            const int i = tid + 256 * j;
            v[j] = i < lim ? g[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    request(blockIdx.x, tid0);
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        // (the thread's indices are re-defined opaquely per iteration: otherwise every staging address and every output offset
        // of the epilogue is computed once before the loop and kept in registers across it -- 100+ of them)
        int tid = tid0, lane = lane0;
        asm volatile("" : "+v"(tid), "+v"(lane));
        float gscale = 1.f, cinv = 1.f;                       // F16: 2^e of this group's staged tensor, 2^-(e + kw) for its outputs
        {
            NQ_SUM(0);
            if constexpr (F16) {
                float mr = 0.f;
#pragma unroll
                for (int j = 0; j < C::NV; ++j)                  // (values beyond the group's end were loaded as zeros)
                    mr = fmaxf(fmaxf(mr, fmaxf(__builtin_fabsf(v[j][0]), __builtin_fabsf(v[j][1]))), fmaxf(__builtin_fabsf(v[j][2]), __builtin_fabsf(v[j][3])));
                mr = wave_max_nonneg(mr);
                if (lane == 0) lds_st32(SC_MAXSLOT + 4 * wave, __float_as_uint(mr));
            }
            __syncthreads();                                    // every wave has left the K loop over the previous planes
            if constexpr (F16) {
                // (the slot is rewritten only behind the NEXT group's first barrier's predecessor: the barrier below separates)
                const f32x4 m4 = lds_ld128(SC_MAXSLOT);
                const int ge = f16_scale_exp(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])));
                gscale = pow2_f32(ge);
                cinv = pow2_f32(-(ge + kw));
            }
            NQ_SUM(1);
#pragma unroll
            for (int j = 0; j < C::NV; ++j) {
                const int i = tid + 256 * j;
                if (C::F4 % 256 == 0 || i < C::F4) {
                    const int pix = (4 * i) / CIN, c = (4 * i) % CIN;
                    if (F32) {
                        lds_st128(C::BASE + pix * C::RS + 4 * c, v[j]);
                        continue;
                    }
                    const unsigned a = SC_BASE + pix * C::RS + 2 * c;
                    if constexpr (TERMS == 3) {
                        sc_store_terms4<3>(a, C::PLANE, v[j]);
                        continue;
                    }
                    if constexpr (F16) {
                        const f32x4 sv = v[j] * gscale;
                        const unsigned h0 = cvt_pk_f16(sv[0], sv[1]), h1 = cvt_pk_f16(sv[2], sv[3]);
                        const f32x2_t f0 = __builtin_convertvector(__builtin_bit_cast(f16x2_t, h0), f32x2_t);
                        const f32x2_t f1 = __builtin_convertvector(__builtin_bit_cast(f16x2_t, h1), f32x2_t);
                        lds_st32(a, h0); lds_st32(a + 4, h1);
                        lds_st32(a + C::PLANE, cvt_pk_f16(sv[0] - f0[0], sv[1] - f0[1]));
                        lds_st32(a + C::PLANE + 4, cvt_pk_f16(sv[2] - f1[0], sv[3] - f1[1]));
                        continue;
                    }
                    const unsigned h0 = cvt_pk_bf16(v[j][0], v[j][1]), h1 = cvt_pk_bf16(v[j][2], v[j][3]);
                    const unsigned l0 = cvt_pk_bf16(v[j][0] - __uint_as_float(h0 << 16), v[j][1] - __uint_as_float(h0 & 0xffff0000u));
                    const unsigned l1 = cvt_pk_bf16(v[j][2] - __uint_as_float(h1 << 16), v[j][3] - __uint_as_float(h1 & 0xffff0000u));
                    lds_st32(a, h0); lds_st32(a + 4, h1);
                    lds_st32(a + C::PLANE, l0); lds_st32(a + C::PLANE + 4, l1);
                }
            }
        }
        NQ_SUM(2);                                              // split + stored (includes the wait for the loads)
        __syncthreads();
        NQ_SUM(3);
        float touch[NTOUCH];
        {
            const int nxt = grp + (int)gridDim.x;
            const float* g = src + (size_t)nxt * SEGS * C::PXS * CIN;
            const int lim = nxt < n_groups ? min(SEGS, n_segments - nxt * SEGS) * C::PXS * CIN / 32 : 0;
#pragma unroll
            for (int j = 0; j < NTOUCH; ++j) {
                const int i = tid + 256 * j;
                touch[j] = i < lim ? g[32 * i] : 0.f;
            }
        }
        SC_FENCE();                                             // keep the requests above the K loop (hipcc sinks loads to their use)
        // loop-invariant code motion would lift the per-tap address selects of every tile out of the group loop (72 registers
        // live across everything: spills); opaque re-definitions keep them inside
#pragma unroll
        for (int t = 0; t < MT; ++t) asm volatile("" : "+v"(base[t]), "+v"(m9[t]));

        f32x16 acc[MT][NT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = zero16();
        if (active) {
            if constexpr (F32) conv_k_f32<CIN, MT, NT, WS, C::RS, C::ZADDR, SC_RING>(acc, rsrc, lane0 * 16, base, m9);
            else if constexpr (TERMS == 3) conv_k_terms<3, CIN, MT, NT, WS, C::RS, C::PLANE, SC_ZADDR, SC_RING, SC_X6_FENCE>(acc, rsrc, 0, lane0 * 16, base, m9);
            else conv_k_bf16<CIN, MT, NT, WS, C::RS, C::PLANE, SC_ZADDR, (MT * NT <= 4), SC_RING, FMT>(acc, rsrc, 0, lane0 * 16, base, m9);
        }

        NQ_SUM(4);                                              // K loop
        // the next group's real loads go out BEFORE this group's output stores (they would queue behind 64 stores per lane
        // otherwise: 13 % of a group's time); its lines were touched into L2 before the K loop, the accumulators are the only
        // other large live set here
        request(grp + (int)gridDim.x, tid);
        __builtin_amdgcn_sched_barrier(0);
        const int seg0 = grp * SEGS;
        const int rows = min(SEGS, n_segments - seg0) * C::PXR;
        float* o = out + (size_t)seg0 * C::PXR * NOUT;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = 32 * nt + (lane & 31);
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wave * MT + t) * 32 + NQ_DROW(r, lane >> 5);
                    if (row < rows && col < NOUT) {
                        const float zv = F16 ? fmaf(acc[t][nt][r], cinv, bv[nt]) : acc[t][nt][r] + bv[nt];
                        o[(size_t)row * NOUT + col] = zv;
                        if (FWD) { s1[nt] += (double)zv; s2[nt] += (double)zv * (double)zv; }
                    }
                }
        }
#pragma unroll
        for (int j = 0; j < NTOUCH; ++j) asm volatile("" ::"v"(touch[j]));      // the touches end here, not before
        NQ_SUM(5);                                              // epilogue
        NQ_SUM_COUNT(6, 1);
    }
    NQ_SUM_COUNT(7, 1);
    NQ_SUM_END(g_sc_clk, blockIdx.x * 4 + wave, lane0 == 0);

    // ---- BatchNorm statistics of the workgroup's rows: lane pairs, the four waves through LDS, one atomic per channel
    if (with_stats) {
        double* red = (double*)(smem + C::BASE);
        __syncthreads();                                        // the planes are dead
        for (int q = tid0; q < 2 * 32 * NT; q += 256) red[q] = 0.0;
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = 32 * nt + (lane0 & 31);
            const double a1 = s1[nt] + __shfl_xor(s1[nt], 32), a2 = s2[nt] + __shfl_xor(s2[nt], 32);
            if (lane0 < 32 && col < NOUT) {
                atomicAdd(&red[col], a1);
                atomicAdd(&red[32 * NT + col], a2);
            }
        }
        __syncthreads();
        for (int q = tid0; q < 32 * NT; q += 256)
            if (q < NOUT) {
                atomicAdd(stats + q, red[q]);
                atomicAdd(stats + NOUT + q, red[32 * NT + q]);
            }
    }
}

// Launch state that depends on the DEVICE (CU count, resident workgroups per CU, the > 64 KB dynamic-LDS opt-in) is cached
// per device ordinal: a process that drives a second GPU asks again for that GPU.  Atomics make the caches safe to fill
// from several host threads (the worst case is two threads asking the runtime the same question once).
#include <atomic>
static constexpr int SC_MAX_DEV = 64;
static int sc_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SC_MAX_DEV) dev = 0;
    return dev;
}
static int sc_cu_count() {
    static std::atomic<int> n[SC_MAX_DEV];
    const int dev = sc_device();
    int v = n[dev].load(std::memory_order_relaxed);
    if (!v) {
        hipDeviceProp_t p;
        v = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
        n[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

template <int CIN, int NT, int NOUT, int HR, int WR, int HS, int WS, int PADX, int SEGS, int MT, bool FWD, bool F32 = false, int TERMS = 2, int FMT = NQ_FMT_BF16X3>
static void segconv_launch(hipStream_t st, const float* src, const uint16_t* frags, float* out, int n_segments,
                           const float* bias, double* stats) {
    typedef segconv_cfg<CIN, NT, NOUT, HR, WR, HS, WS, PADX, SEGS, MT, FWD, F32, TERMS, FMT> C;
    static std::atomic<int> per_cu_dev[SC_MAX_DEV];             // resident workgroups per CU (registers and LDS), asked once per device
    const int dev = sc_device();
    int per_cu = per_cu_dev[dev].load(std::memory_order_relaxed);
    if (!per_cu) {
        // 50-80 KB of dynamic LDS: opt in explicitly (a runtime that enforces the 64 KB default would refuse the launch)
        (void)hipFuncSetAttribute((const void*)segconv_bf16_kernel<CIN, NT, NOUT, HR, WR, HS, WS, PADX, SEGS, MT, FWD, F32, TERMS, FMT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, segconv_bf16_kernel<CIN, NT, NOUT, HR, WR, HS, WS, PADX, SEGS, MT, FWD, F32, TERMS, FMT>, 256,
                                                         C::LDS) != hipSuccess || nb < 1)
            nb = 2;
        per_cu = nb;
        per_cu_dev[dev].store(nb, std::memory_order_relaxed);
    }
    const int n_groups = (n_segments + SEGS - 1) / SEGS;
    const int grid = n_groups < per_cu * sc_cu_count() ? n_groups : per_cu * sc_cu_count();
    hipLaunchKernelGGL((segconv_bf16_kernel<CIN, NT, NOUT, HR, WR, HS, WS, PADX, SEGS, MT, FWD, F32, TERMS, FMT>), dim3(grid), dim3(256), C::LDS, st, src,
                       frags, out, n_segments, bias, stats);
}

// the five layer shapes of the AdaptCNN at the reference's pooling sizes (config/train_nisqa_cnn_sa_ap.yaml:
// cnn_pool_1 [24, 7], cnn_pool_2 [12, 5], cnn_pool_3 [6, 3]); for anything else nisqa_segconv_supported says 0 and the
// caller keeps nisqa_conv3x3_gemm_bf16
#define SC_KEY(h_, w_, ci_, co_) ((((h_) * 100 + (w_)) * 100 + (ci_)) * 100 + (co_))
extern "C" int nisqa_segconv_supported(int32_t h, int32_t w, int32_t ci, int32_t co, int32_t pad_w) {
    const int key = SC_KEY(h, w, ci, co);
    if (pad_w == 0) return key == SC_KEY(6, 3, 64, 64);
    return pad_w == 1 && (key == SC_KEY(24, 7, 16, 32) || key == SC_KEY(12, 5, 32, 64) || key == SC_KEY(12, 5, 64, 64) ||
                          key == SC_KEY(6, 3, 64, 64));
}
extern "C" int nisqa_segconv_bf16(int32_t mode, const float* src, const uint16_t* frags, float* out, int32_t n_segments,
                                  int32_t h, int32_t w, int32_t ci, int32_t co, int32_t pad_w, const float* bias,
                                  double* stats2c, void* stream) {
    if (mode < 0 || mode > 1 || !src || !frags || !out || n_segments <= 0 || (mode == 1 && (bias || stats2c)) ||
        !nisqa_segconv_supported(h, w, ci, co, pad_w))
        return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    NQ_LAUNCH_BEGIN();
    const int key = ((h * 100 + w) * 100 + ci) * 100 + co;
    const int n = n_segments;
    if (mode == 0) {
        if (key == SC_KEY(24, 7, 16, 32) && pad_w == 1) segconv_launch<16, 1, 32, 24, 7, 24, 7, 1, 3, 4, true>(st, src, frags, out, n, bias, stats2c);
        else if (key == SC_KEY(12, 5, 32, 64) && pad_w == 1) segconv_launch<32, 2, 64, 12, 5, 12, 5, 1, SC_SEGS60, SC_SEGS60 / 2, true>(st, src, frags, out, n, bias, stats2c);
        else if (key == SC_KEY(12, 5, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 12, 5, 12, 5, 1, SC_SEGS60, SC_SEGS60 / 2, true>(st, src, frags, out, n, bias, stats2c);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 6, 3, 6, 3, 1, 14, 2, true>(st, src, frags, out, n, bias, stats2c);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 0) segconv_launch<64, 2, 64, 6, 1, 6, 3, 0, 15, 1, true>(st, src, frags, out, n, bias, stats2c);
        else return NISQA_ERR_ARG;
    } else {                                                  // staged tensor = dz [S][h * wo][co], rows = the h * w input pixels
        if (key == SC_KEY(24, 7, 16, 32) && pad_w == 1) segconv_launch<32, 1, 16, 24, 7, 24, 7, 1, 2, 3, false>(st, src, frags, out, n, nullptr, nullptr);
        else if (key == SC_KEY(12, 5, 32, 64) && pad_w == 1) segconv_launch<64, 1, 32, 12, 5, 12, 5, 1, SC_SEGS60, SC_SEGS60 / 2, false>(st, src, frags, out, n, nullptr, nullptr);
        else if (key == SC_KEY(12, 5, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 12, 5, 12, 5, 1, SC_SEGS60, SC_SEGS60 / 2, false>(st, src, frags, out, n, nullptr, nullptr);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 6, 3, 6, 3, 1, 14, 2, false>(st, src, frags, out, n, nullptr, nullptr);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 0) segconv_launch<64, 2, 64, 6, 3, 6, 1, 2, 14, 2, false>(st, src, frags, out, n, nullptr, nullptr);
        else return NISQA_ERR_ARG;
    }
    return NQ_LAUNCH_STATUS();
}

// the same two products on f16 hi + lo of the per-group scaled tensors, four term products (precision mode 'f16x4'; frags =
// nisqa_segconv_pack_f16_many of the same mode, nisqa_segconv_frag_bytes_f16 bytes)
extern "C" int nisqa_segconv_f16(int32_t mode, const float* src, const uint16_t* frags, float* out, int32_t n_segments,
                                  int32_t h, int32_t w, int32_t ci, int32_t co, int32_t pad_w, const float* bias,
                                  double* stats2c, void* stream) {
    if (mode < 0 || mode > 1 || !src || !frags || !out || n_segments <= 0 || (mode == 1 && (bias || stats2c)) ||
        !nisqa_segconv_supported(h, w, ci, co, pad_w))
        return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    NQ_LAUNCH_BEGIN();
    const int key = ((h * 100 + w) * 100 + ci) * 100 + co;
    const int n = n_segments;
    if (mode == 0) {
        if (key == SC_KEY(24, 7, 16, 32) && pad_w == 1) segconv_launch<16, 1, 32, 24, 7, 24, 7, 1, 3, 4, true, false, 2, NQ_FMT_F16X4>(st, src, frags, out, n, bias, stats2c);
        else if (key == SC_KEY(12, 5, 32, 64) && pad_w == 1) segconv_launch<32, 2, 64, 12, 5, 12, 5, 1, SC_SEGS60, SC_SEGS60 / 2, true, false, 2, NQ_FMT_F16X4>(st, src, frags, out, n, bias, stats2c);
        else if (key == SC_KEY(12, 5, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 12, 5, 12, 5, 1, SC_SEGS60, SC_SEGS60 / 2, true, false, 2, NQ_FMT_F16X4>(st, src, frags, out, n, bias, stats2c);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 6, 3, 6, 3, 1, 14, 2, true, false, 2, NQ_FMT_F16X4>(st, src, frags, out, n, bias, stats2c);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 0) segconv_launch<64, 2, 64, 6, 1, 6, 3, 0, 15, 1, true, false, 2, NQ_FMT_F16X4>(st, src, frags, out, n, bias, stats2c);
        else return NISQA_ERR_ARG;
    } else {                                                  // staged tensor = dz [S][h * wo][co], rows = the h * w input pixels
        if (key == SC_KEY(24, 7, 16, 32) && pad_w == 1) segconv_launch<32, 1, 16, 24, 7, 24, 7, 1, 2, 3, false, false, 2, NQ_FMT_F16X4>(st, src, frags, out, n, nullptr, nullptr);
        else if (key == SC_KEY(12, 5, 32, 64) && pad_w == 1) segconv_launch<64, 1, 32, 12, 5, 12, 5, 1, SC_SEGS60, SC_SEGS60 / 2, false, false, 2, NQ_FMT_F16X4>(st, src, frags, out, n, nullptr, nullptr);
        else if (key == SC_KEY(12, 5, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 12, 5, 12, 5, 1, SC_SEGS60, SC_SEGS60 / 2, false, false, 2, NQ_FMT_F16X4>(st, src, frags, out, n, nullptr, nullptr);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 6, 3, 6, 3, 1, 14, 2, false, false, 2, NQ_FMT_F16X4>(st, src, frags, out, n, nullptr, nullptr);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 0) segconv_launch<64, 2, 64, 6, 3, 6, 1, 2, 14, 2, false, false, 2, NQ_FMT_F16X4>(st, src, frags, out, n, nullptr, nullptr);
        else return NISQA_ERR_ARG;
    }
    return NQ_LAUNCH_STATUS();
}

// the same two products in exact fp32 (frags = nisqa_segconv_pack_f32_many of the same mode)
extern "C" int nisqa_segconv_f32(int32_t mode, const float* src, const float* frags, float* out, int32_t n_segments, int32_t h,
                                 int32_t w, int32_t ci, int32_t co, int32_t pad_w, const float* bias, double* stats2c, void* stream) {
    if (mode < 0 || mode > 1 || !src || !frags || !out || n_segments <= 0 || (mode == 1 && (bias || stats2c)) ||
        !nisqa_segconv_supported(h, w, ci, co, pad_w))
        return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const uint16_t* fr = (const uint16_t*)frags;
    NQ_LAUNCH_BEGIN();
    const int key = ((h * 100 + w) * 100 + ci) * 100 + co;
    const int n = n_segments;
    if (mode == 0) {
        if (key == SC_KEY(24, 7, 16, 32) && pad_w == 1) segconv_launch<16, 1, 32, 24, 7, 24, 7, 1, 3, 4, true, true>(st, src, fr, out, n, bias, stats2c);
        else if (key == SC_KEY(12, 5, 32, 64) && pad_w == 1) segconv_launch<32, 2, 64, 12, 5, 12, 5, 1, SC_SEGS60, SC_SEGS60 / 2, true, true>(st, src, fr, out, n, bias, stats2c);
        else if (key == SC_KEY(12, 5, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 12, 5, 12, 5, 1, SC_SEGS60, SC_SEGS60 / 2, true, true>(st, src, fr, out, n, bias, stats2c);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 6, 3, 6, 3, 1, 14, 2, true, true>(st, src, fr, out, n, bias, stats2c);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 0) segconv_launch<64, 2, 64, 6, 1, 6, 3, 0, 15, 1, true, true>(st, src, fr, out, n, bias, stats2c);
        else return NISQA_ERR_ARG;
    } else {
        if (key == SC_KEY(24, 7, 16, 32) && pad_w == 1) segconv_launch<32, 1, 16, 24, 7, 24, 7, 1, 2, 3, false, true>(st, src, fr, out, n, nullptr, nullptr);
        else if (key == SC_KEY(12, 5, 32, 64) && pad_w == 1) segconv_launch<64, 1, 32, 12, 5, 12, 5, 1, SC_SEGS60, SC_SEGS60 / 2, false, true>(st, src, fr, out, n, nullptr, nullptr);
        else if (key == SC_KEY(12, 5, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 12, 5, 12, 5, 1, SC_SEGS60, SC_SEGS60 / 2, false, true>(st, src, fr, out, n, nullptr, nullptr);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 6, 3, 6, 3, 1, 14, 2, false, true>(st, src, fr, out, n, nullptr, nullptr);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 0) segconv_launch<64, 2, 64, 6, 3, 6, 1, 2, 14, 2, false, true>(st, src, fr, out, n, nullptr, nullptr);
        else return NISQA_ERR_ARG;
    }
    return NQ_LAUNCH_STATUS();
}

// the same two products at fp32 OPERAND precision on the bf16 matrix pipe (precision mode 'bf16x6'): activations and weights
// as three exact bf16 terms, six products per term pair (conv_k_terms; frags = nisqa_segconv_pack_x6_many of the same mode).
// Three planes per staged tensor: fewer segments per workgroup than the two-term table above, still two workgroups per CU.
extern "C" int nisqa_segconv_bf16x6(int32_t mode, const float* src, const uint16_t* frags, float* out, int32_t n_segments, int32_t h,
                                    int32_t w, int32_t ci, int32_t co, int32_t pad_w, const float* bias, double* stats2c, void* stream) {
    if (mode < 0 || mode > 1 || !src || !frags || !out || n_segments <= 0 || (mode == 1 && (bias || stats2c)) ||
        !nisqa_segconv_supported(h, w, ci, co, pad_w))
        return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    NQ_LAUNCH_BEGIN();
    const int key = ((h * 100 + w) * 100 + ci) * 100 + co;
    const int n = n_segments;
    if (mode == 0) {
        if (key == SC_KEY(24, 7, 16, 32) && pad_w == 1) segconv_launch<16, 1, 32, 24, 7, 24, 7, 1, 3, 4, true, false, 3>(st, src, frags, out, n, bias, stats2c);
        else if (key == SC_KEY(12, 5, 32, 64) && pad_w == 1) segconv_launch<32, 2, 64, 12, 5, 12, 5, 1, 4, 2, true, false, 3>(st, src, frags, out, n, bias, stats2c);
        else if (key == SC_KEY(12, 5, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 12, 5, 12, 5, 1, 2, 1, true, false, 3>(st, src, frags, out, n, bias, stats2c);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 6, 3, 6, 3, 1, 7, 1, true, false, 3>(st, src, frags, out, n, bias, stats2c);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 0) segconv_launch<64, 2, 64, 6, 1, 6, 3, 0, 7, 1, true, false, 3>(st, src, frags, out, n, bias, stats2c);
        else return NISQA_ERR_ARG;
    } else {
        if (key == SC_KEY(24, 7, 16, 32) && pad_w == 1) segconv_launch<32, 1, 16, 24, 7, 24, 7, 1, 1, 2, false, false, 3>(st, src, frags, out, n, nullptr, nullptr);
        else if (key == SC_KEY(12, 5, 32, 64) && pad_w == 1) segconv_launch<64, 1, 32, 12, 5, 12, 5, 1, 2, 1, false, false, 3>(st, src, frags, out, n, nullptr, nullptr);
        else if (key == SC_KEY(12, 5, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 12, 5, 12, 5, 1, 2, 1, false, false, 3>(st, src, frags, out, n, nullptr, nullptr);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 1) segconv_launch<64, 2, 64, 6, 3, 6, 3, 1, 7, 1, false, false, 3>(st, src, frags, out, n, nullptr, nullptr);
        else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 0) segconv_launch<64, 2, 64, 6, 3, 6, 1, 2, 14, 2, false, false, 3>(st, src, frags, out, n, nullptr, nullptr);
        else return NISQA_ERR_ARG;
    }
    return NQ_LAUNCH_STATUS();
}

// ======================================================================================================================
// Weight gradient, segment-resident: dw[co][tap * CI + ci] += sum over the pixels of every segment of
// dz[px][co] * x[px + tap][ci]  -- a GEMM with M = co, N = (tap, ci), K = pixels.  The implicit GEMM (train.hip, mode 2)
// gathers both operands per K-tile from HBM with index arithmetic and restarts its accumulators per row chunk (split-K,
// atomics per chunk).  Here a workgroup stages the x and dz of SEGS whole segments once (x with a ZERO BORDER, so a tap is
// a constant address offset and needs no mask), its four waves split the N tiles and keep their part of dw in registers
// across ALL the groups the workgroup walks over (one set of atomics per workgroup at the end).
//   * k is the pixel, i.e. the planes' ROW index: fragments come out of ds_read_b64_tr_b16, gfx950's transposing LDS read
//     (in a 16-lane group lane s points at row s >> 2, columns 4 (s & 3) .. + 3 of a [4 rows][16 columns] block and lane i
//     receives column i; two reads = the 8 k of a lane; mapping read off the hardware by tools/micro/trread.hip);
//   * eight waves; wave w owns the N tiles w, w + 8, ... (or, MSPLIT = 2, one of the two M tiles and every fourth N tile); a
//     tile's (tap, channel) offset is one add per read; A (dz) fragments are shared by a wave's tiles;
//   * two LDS buffers: the next group is requested before the K loop and split into the other buffer behind it, one barrier
//     per group.
// ======================================================================================================================
typedef short sc_s16x4 __attribute__((ext_vector_type(4)));
NQ_DEV f32x4 sc_tr_frag(unsigned a0, unsigned a1) {
    const sc_s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((NQ_AS3 sc_s16x4*)(a0));
    const sc_s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((NQ_AS3 sc_s16x4*)(a1));
    struct { sc_s16x4 a, b; } both = {x0, x1};
    return __builtin_bit_cast(f32x4, both);
}

// TERMS: 2 = hi / lo planes of x and dz, three products; 3 = hi / mid / lo (exact splits), six products: fp32-grade ('bf16x6')
template <int CI, int CO, int H, int W, int WO, int PADW, int SEGS, int MSPLIT, int TERMS = 2>
struct segwgrad_cfg {
    static constexpr int T = TERMS;
    static constexpr int RSX = 2 * CI + 16, RSZ = 2 * CO + 16;
    static constexpr int XPW = W + 2 * PADW, XPH = H + 2;
    static constexpr int PXX = XPH * XPW, PXI = H * W, PXZ = H * WO;
    static constexpr int PLX = SEGS * PXX * RSX;
    static constexpr int KROWS = SEGS * PXZ;
    static constexpr int PLZ = (KROWS + 1) * RSZ;            // + one zero row: the k rows behind the last pixel
    static constexpr unsigned ZB = (unsigned)TERMS * PLX;    // inside a buffer: the x planes (hi, [mid,] lo), then the dz planes
    static constexpr unsigned BUF = (unsigned)TERMS * PLX + (unsigned)TERMS * PLZ;
    static constexpr unsigned LDS = 2u * BUF;                // two buffers: group g + 1 is staged while group g is multiplied
    static constexpr int KSTEPS = (KROWS + 15) / 16;
    static constexpr int MT = CO / 32, MTW = MT / MSPLIT;    // M tiles of the layer / of a wave
    static constexpr int NSPLIT = 8 / MSPLIT;                // waves along N
    static constexpr int NTILES = (9 * CI + 31) / 32;
    static constexpr int NTW = (NTILES + NSPLIT - 1) / NSPLIT;
    static constexpr int FX = SEGS * PXI * CI / 4, FZ = SEGS * PXZ * CO / 4;     // 128-bit groups of a group's x / dz
    static constexpr int NVX = (FX + 511) / 512, NVZ = (FZ + 511) / 512;
    static_assert(LDS <= 160 * 1024, "LDS");
    static_assert(BUF % 16 == 0 && CI % 16 == 0 && CO % 32 == 0 && MT % MSPLIT == 0, "shapes");
};

// K loop of a wave: its N tiles are wq, wq + NSPLIT, ...; noff[j] = byte offset of tile j (tap, channels) in an x plane
template <typename C>
NQ_DEV void segwgrad_kloop(f32x16 (&acc)[C::MTW][C::NTW], const unsigned (&za)[C::KSTEPS][2], const unsigned (&xa)[C::KSTEPS][2],
                           const unsigned (&noff)[C::NTW], int n_own) {
#pragma unroll
    for (int st = 0; st < C::KSTEPS; ++st) {
        __builtin_amdgcn_sched_barrier(0);                     // a step's reads stay in their step (register pressure)
        f32x4 a[C::MTW][C::T];
#pragma unroll
        for (int m = 0; m < C::MTW; ++m)
#pragma unroll
            for (int t = 0; t < C::T; ++t) a[m][t] = sc_tr_frag(za[st][0] + 64 * m + t * C::PLZ, za[st][1] + 64 * m + t * C::PLZ);
#pragma unroll
        for (int j = 0; j < C::NTW; ++j) {
            if (j < n_own) {                                   // wave-uniform
                const unsigned a0 = xa[st][0] + noff[j], a1 = xa[st][1] + noff[j];
                f32x4 b[C::T];
#pragma unroll
                for (int t = 0; t < C::T; ++t) b[t] = sc_tr_frag(a0 + t * C::PLX, a1 + t * C::PLX);
                // the term products (i, j2) with i + j2 <= T - 1, smallest first (T = 2: hl, lh, hh as before)
#pragma unroll
                for (int order = C::T - 1; order >= 0; --order)
#pragma unroll
                    for (int i = 0; i <= order; ++i) {
                        const int j2 = order - i;
#pragma unroll
                        for (int m = 0; m < C::MTW; ++m) acc[m][j] = mfma_bf(a[m][i], b[j2], acc[m][j]);
                    }
            }
        }
    }
}

// BatchNorm backward folded into the staging (PH > 0): the kernel is handed z (the layer's convolution output) and the POOLED
// gradient dy [S][PH * PW][CO] with the arg-max pixels of the forward pass instead of dz, computes
//   dz = gamma rstd (gate (sum of the <= 2 pooling windows that chose this pixel) drop - mean(dyb) - xhat mean(dyb xhat))
// per element while it deposits a group into LDS -- the arithmetic of bn_act_pool_bwd_dense_kernel (train.hip), whose dense
// pass over z and dz (0.29 ms per step for layers 2-6, memory-bound) disappears -- and writes dz out once for the input
// gradient kernel that runs next.  sums2 = the float64 sums over the pooled values (nisqa_bn_pool_bwd_sums).
struct segw_bn {
    const float* z; const float* dy; const int32_t* arg; const float* drop; const float* mean_rstd; const float* gamma;
    const float* beta; const double* sums2; float* dz_out; float* dgamma; float* dbeta;
};
NQ_DEV int sc_win_lo(int i, int n_in, int n_out) { return (i * n_in) / n_out; }
NQ_DEV int sc_win_hi(int i, int n_in, int n_out) { return ((i + 1) * n_in + n_out - 1) / n_out; }
typedef int sc_i32x4 __attribute__((ext_vector_type(4)));

template <int CI, int CO, int H, int W, int WO, int PADW, int SEGS, int MSPLIT, int PH, int PW, int TERMS = 2>
__global__ __launch_bounds__(512, 1) void segwgrad_bf16_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                               float* __restrict__ dw, int n_segments, segw_bn bn) {
    typedef segwgrad_cfg<CI, CO, H, W, WO, PADW, SEGS, MSPLIT, TERMS> C;
    constexpr bool BN = PH > 0;
    constexpr bool IDENT = PH == H && PW == WO;
    static_assert(!BN || (H % PH == 0 && 512 % (CO / 4) == 0), "pooling rows are disjoint; a thread keeps its four channels");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid0 = threadIdx.x, lane0 = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int wq = wave / MSPLIT, wm = wave % MSPLIT;           // this wave's N residue and M part
    const int n_groups = (n_segments + SEGS - 1) / SEGS;
    // zero both buffers once: the borders of the x planes and the zero rows of dz stay zero
    for (unsigned a = 16u * tid0; a < C::LDS; a += 16u * 512u) *(f32x4*)(smem + a) = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane-static read addresses (buffer 0): k row of (step, read) -> dz row / x pixel at tap (0, 0) in padded coordinates
    const int g16 = (lane0 >> 4) & 1;
    unsigned za0[C::KSTEPS][2], xa0[C::KSTEPS][2];
#pragma unroll
    for (int st = 0; st < C::KSTEPS; ++st)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int r = 16 * st + 8 * (lane0 >> 5) + 4 * rd + ((lane0 & 15) >> 2);
            const bool valid = r < C::KROWS;
            const int sg = r / C::PXZ, p = r - sg * C::PXZ, y = p / WO, xo = p - y * WO;
            za0[st][rd] = C::ZB + (unsigned)((valid ? r : C::KROWS) * C::RSZ) + 2u * (16 * g16 + 4 * (lane0 & 3)) + 64u * C::MTW * wm;
            xa0[st][rd] = (unsigned)((valid ? sg * C::PXX + y * C::XPW + xo : 0) * C::RSX) + 2u * (4 * (lane0 & 3));
        }
    const int n_own = (C::NTILES - wq + C::NSPLIT - 1) / C::NSPLIT;      // N tiles of this wave
    unsigned noff[C::NTW];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) {
        const int n0 = 32 * (wq + C::NSPLIT * j) + (CI == 16 ? 16 * g16 : 0);
        const int tap = min(n0 / CI, 8);                        // columns behind 9 * CI: computed on valid memory, never stored
        noff[j] = (unsigned)(((tap / 3) * C::XPW + tap % 3) * C::RSX + 2 * (n0 % CI)) + (CI == 16 ? 0u : 32u * g16);
    }
    f32x16 acc[C::MTW][C::NTW];
#pragma unroll
    for (int m = 0; m < C::MTW; ++m)
#pragma unroll
        for (int j = 0; j < C::NTW; ++j) acc[m][j] = zero16();

    f32x4 vx[C::NVX], vz[C::NVZ];
    // BN: the pooled gradients (and arg-max pixels) of the <= 2 windows that contain the element's pixel
    f32x4 vd0[BN ? C::NVZ : 1], vd1[(BN && !IDENT) ? C::NVZ : 1];
    sc_i32x4 va0[(BN && !IDENT) ? C::NVZ : 1], va1[(BN && !IDENT) ? C::NVZ : 1];
    // this thread's four channels are the same in every element it stages (512 * 4 % CO == 0): their constants once
    f32x4 bn_g, bn_b, bn_mu, bn_rs, bn_m1, bn_m2;
    f32x4 vdr[BN ? C::NVZ : 1];                                  // Dropout2d multipliers of the element's (segment, channels)
    if (BN) {
        const int ch = (4 * tid0) % CO;
        const double inv = 1.0 / ((double)n_segments * C::PXZ);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float mu = bn.mean_rstd[ch + e], rs = bn.mean_rstd[CO + ch + e];
            bn_mu[e] = mu;
            bn_rs[e] = rs;
            bn_g[e] = bn.gamma[ch + e] * rs;
            bn_b[e] = bn.beta[ch + e] - mu * bn_g[e];
            bn_m1[e] = (float)(bn.sums2[ch + e] * inv);
            bn_m2[e] = (float)((double)rs * (bn.sums2[CO + ch + e] - (double)mu * bn.sums2[ch + e]) * inv);
        }
        if (blockIdx.x == 0 && tid0 < CO) {                     // the BatchNorm parameter gradients (what the dense kernel's block 0 wrote)
            const double mean = bn.mean_rstd[tid0], rstd = bn.mean_rstd[CO + tid0];
            bn.dbeta[tid0] = (float)bn.sums2[tid0];
            bn.dgamma[tid0] = (float)(rstd * (bn.sums2[CO + tid0] - mean * bn.sums2[tid0]));
        }
    }
    auto request = [&](int grp, int tid) {
        const int seg0 = grp * SEGS;
        const int nseg = grp < n_groups ? min(SEGS, n_segments - seg0) : 0;
        const f32x4* gx = (const f32x4*)(x + (size_t)seg0 * C::PXI * CI);
        const f32x4* gz = (const f32x4*)((BN ? bn.z : dz) + (size_t)seg0 * C::PXZ * CO);
#pragma unroll
        for (int j = 0; j < C::NVX; ++j) {
            const int i = tid + 512 * j;
            vx[j] = i < nseg * C::PXI * CI / 4 ? gx[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < C::NVZ; ++j) {
            const int i = tid + 512 * j;
            const bool ok = i < nseg * C::PXZ * CO / 4;
            vz[j] = ok ? gz[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (BN) {
                const int pix = (4 * i) / CO, c = (4 * i) % CO;
                const int sg = pix / C::PXZ, p = pix - sg * C::PXZ;
                vdr[j] = (ok && bn.drop) ? *(const f32x4*)(bn.drop + (size_t)(seg0 + sg) * CO + c) : f32x4{1.f, 1.f, 1.f, 1.f};
                if (IDENT) {
                    vd0[j] = ok ? *(const f32x4*)(bn.dy + ((size_t)(seg0 + sg) * C::PXZ + p) * CO + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    const int yy = p / WO, xx = p - yy * WO;
                    const int oy = yy / (H / PH);
                    int oa = (xx * PW) / WO;                    // the window whose floor bound is at or below xx ...
                    if (oa > 0 && sc_win_hi(oa - 1, WO, PW) > xx) --oa;      // ... or the one before it when that still covers xx
                    const int ob = oa + 1;
                    const bool has_b = ob < PW && sc_win_lo(ob, WO, PW) <= xx;
                    const size_t o0 = ((size_t)(seg0 + sg) * (PH * PW) + oy * PW + oa) * CO + c;
                    const size_t o1 = ((size_t)(seg0 + sg) * (PH * PW) + oy * PW + ob) * CO + c;
                    vd0[j] = ok ? *(const f32x4*)(bn.dy + o0) : f32x4{0.f, 0.f, 0.f, 0.f};
                    va0[j] = ok ? *(const sc_i32x4*)(bn.arg + o0) : sc_i32x4{-1, -1, -1, -1};
                    vd1[j] = (ok && has_b) ? *(const f32x4*)(bn.dy + o1) : f32x4{0.f, 0.f, 0.f, 0.f};
                    va1[j] = (ok && has_b) ? *(const sc_i32x4*)(bn.arg + o1) : sc_i32x4{-1, -1, -1, -1};
                }
            }
        }
    };
    auto deposit = [&](unsigned buf, int tid, int grp) {         // split and store the requested group into buffer `buf`
#pragma unroll
        for (int j = 0; j < C::NVX; ++j) {
            const int i = tid + 512 * j;
            if (C::FX % 512 == 0 || i < C::FX) {
                const int pix = (4 * i) / CI, c = (4 * i) % CI;
                const int sg = pix / C::PXI, p = pix - sg * C::PXI, y = p / W, xx = p - y * W;
                const unsigned a = buf + (unsigned)((sg * C::PXX + (y + 1) * C::XPW + xx + PADW) * C::RSX) + 2 * c;
                sc_store_terms4<TERMS>(a, C::PLX, vx[j]);
            }
        }
        const int seg0 = grp * SEGS;
        const int nseg_d = grp < n_groups ? min(SEGS, n_segments - seg0) : 0;
#pragma unroll
        for (int j = 0; j < C::NVZ; ++j) {
            const int i = tid + 512 * j;
            if (C::FZ % 512 == 0 || i < C::FZ) {
                const int pix = (4 * i) / CO, c = (4 * i) % CO;
                f32x4 dzv = vz[j];
                if (BN) {
                    const bool ok = i < nseg_d * C::PXZ * CO / 4;
                    const int sg = pix / C::PXZ, p = pix - sg * C::PXZ;
                    const f32x4 zi = vz[j];
                    f32x4 acc;
                    if (IDENT) acc = vd0[j];
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = (va0[j][e] == p ? vd0[j][e] : 0.f) + (va1[j][e] == p ? vd1[j][e] : 0.f);
                    }
                    acc *= vdr[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float zv = zi[e];
                        asm volatile("" : "+v"(zv));              // scalar gates: packed, hipcc emits the op_sel form of DESIGN.md 7.1
                        const float a_ = fmaf(zv, bn_g[e], bn_b[e]) > 0.f ? acc[e] : 0.f;
                        const float xh = (zv - bn_mu[e]) * bn_rs[e];
                        dzv[e] = ok ? bn_g[e] * (a_ - bn_m1[e] - xh * bn_m2[e]) : 0.f;
                    }
                    if (ok) *(f32x4*)(bn.dz_out + (size_t)seg0 * C::PXZ * CO + (size_t)4 * i) = dzv;
                }
                const unsigned a = buf + C::ZB + (unsigned)(pix * C::RSZ) + 2 * c;
                sc_store_terms4<TERMS>(a, C::PLZ, dzv);
            }
        }
    };

    int grp = blockIdx.x;
    request(grp, tid0);
    __syncthreads();                                            // the zero fill is complete
    deposit(0u, tid0, grp);
    __syncthreads();
    unsigned cur = 0u;
    NQ_SUM_BEGIN();
    for (; grp < n_groups; grp += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));                          // keep the staging addresses inside the loop
        request(grp + gridDim.x, tid);                          // the next group travels while this one is multiplied
        __builtin_amdgcn_sched_barrier(0);                      // (hipcc would sink these loads to their use behind the K loop)
        NQ_SUM(0);                                              // requests issued
        unsigned za[C::KSTEPS][2], xa[C::KSTEPS][2];
#pragma unroll
        for (int st = 0; st < C::KSTEPS; ++st)
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                za[st][rd] = za0[st][rd] + cur;
                xa[st][rd] = xa0[st][rd] + cur;
            }
        segwgrad_kloop<C>(acc, za, xa, noff, n_own);
        __builtin_amdgcn_sched_barrier(0);
        NQ_SUM(4);                                              // K loop
        deposit(C::BUF - cur, tid, grp + (int)gridDim.x);       // the other buffer: last read before the previous barrier
        NQ_SUM(2);                                              // wait for the loads + split + store
        __syncthreads();
        NQ_SUM(3);
        cur = C::BUF - cur;
        NQ_SUM_COUNT(6, 1);
    }
    NQ_SUM_RESTART();
    // ---- this workgroup's share of dw
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) {
        const int nt = wq + C::NSPLIT * j;
        const int col = 32 * nt + (lane0 & 31);
        if (nt < C::NTILES && col < 9 * CI) {
#pragma unroll
            for (int m = 0; m < C::MTW; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * (C::MTW * wm + m) + NQ_DROW(r, lane0 >> 5);
                    atomicAdd(dw + (size_t)row * (9 * CI) + col, acc[m][j][r]);
                }
        }
    }
    NQ_SUM(5);                                                  // the atomics (issue only)
    NQ_SUM_COUNT(7, 1);
    NQ_SUM_END(g_sc_clk, blockIdx.x * 8 + wave, lane0 == 0);
}

template <int CI, int CO, int H, int W, int WO, int PADW, int SEGS, int MSPLIT, int PH = 0, int PW = 0, int TERMS = 2>
static void segwgrad_launch(hipStream_t st, const float* x, const float* dz, float* dw, int n_segments, segw_bn bn = segw_bn{}) {
    typedef segwgrad_cfg<CI, CO, H, W, WO, PADW, SEGS, MSPLIT, TERMS> C;
    static std::atomic<bool> attr[SC_MAX_DEV];
    const int dev = sc_device();
    if (!attr[dev].load(std::memory_order_relaxed)) {           // more than 64 KB of dynamic LDS, per device
        (void)hipFuncSetAttribute((const void*)segwgrad_bf16_kernel<CI, CO, H, W, WO, PADW, SEGS, MSPLIT, PH, PW, TERMS>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
        attr[dev].store(true, std::memory_order_relaxed);
    }
    const int n_groups = (n_segments + SEGS - 1) / SEGS;
    const int grid = n_groups < sc_cu_count() ? n_groups : sc_cu_count();
    hipLaunchKernelGGL((segwgrad_bf16_kernel<CI, CO, H, W, WO, PADW, SEGS, MSPLIT, PH, PW, TERMS>), dim3(grid), dim3(512), C::LDS, st, x, dz, dw,
                       n_segments, bn);
}

// dw[co][9 * ci] += dz^T * patches(x)  (dw zeroed by the caller, as for nisqa_conv3x3_gemm mode 2); same five shapes
extern "C" int nisqa_segconv_wgrad_bf16(const float* x, const float* dz, float* dw, int32_t n_segments, int32_t h, int32_t w,
                                        int32_t ci, int32_t co, int32_t pad_w, void* stream) {
    if (!x || !dz || !dw || n_segments <= 0 || !nisqa_segconv_supported(h, w, ci, co, pad_w)) return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    NQ_LAUNCH_BEGIN();
    const int key = SC_KEY(h, w, ci, co);
    if (key == SC_KEY(24, 7, 16, 32)) segwgrad_launch<16, 32, 24, 7, 7, 1, 1, 1>(st, x, dz, dw, n_segments);
    else if (key == SC_KEY(12, 5, 32, 64)) segwgrad_launch<32, 64, 12, 5, 5, 1, 1, 2>(st, x, dz, dw, n_segments);
    else if (key == SC_KEY(12, 5, 64, 64)) segwgrad_launch<64, 64, 12, 5, 5, 1, 1, SC_W_MSPLIT>(st, x, dz, dw, n_segments);
    else if (pad_w == 1) segwgrad_launch<64, 64, 6, 3, 3, 1, 4, SC_W_MSPLIT>(st, x, dz, dw, n_segments);
    else segwgrad_launch<64, 64, 6, 3, 1, 0, 8, SC_W_MSPLIT>(st, x, dz, dw, n_segments);
    return NQ_LAUNCH_STATUS();
}

// The same with the BatchNorm / ReLU / max-pool / Dropout2d backward of the layer folded into the staging (see segw_bn): for
// the five layers of the reference configuration with their pooling sizes (ho, wo) = (12, 5) (12, 5) (6, 3) (6, 3) (6, 1).
// Returns NISQA_ERR_ARG for anything else (callers then run nisqa_bn_act_pool_bwd + nisqa_segconv_wgrad_bf16).
extern "C" int nisqa_segconv_wgrad_bn_bf16(const float* x, const float* z, const float* dy, const int32_t* arg, const float* drop,
                                           const float* mean_rstd, const float* gamma, const float* beta, const double* sums2,
                                           float* dz_out, float* dgamma, float* dbeta, float* dw, int32_t n_segments, int32_t h,
                                           int32_t w, int32_t ci, int32_t co, int32_t pad_w, int32_t ho, int32_t wo, void* stream) {
    if (!x || !z || !dy || !arg || !mean_rstd || !gamma || !beta || !sums2 || !dz_out || !dgamma || !dbeta || !dw || n_segments <= 0 ||
        !nisqa_segconv_supported(h, w, ci, co, pad_w))
        return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const segw_bn bn = {z, dy, arg, drop, mean_rstd, gamma, beta, sums2, dz_out, dgamma, dbeta};
    const int key = SC_KEY(h, w, ci, co);
    NQ_LAUNCH_BEGIN();
    if (key == SC_KEY(24, 7, 16, 32) && ho == 12 && wo == 5) segwgrad_launch<16, 32, 24, 7, 7, 1, 1, 1, 12, 5>(st, x, nullptr, dw, n_segments, bn);
    else if (key == SC_KEY(12, 5, 32, 64) && ho == 12 && wo == 5) segwgrad_launch<32, 64, 12, 5, 5, 1, 1, 2, 12, 5>(st, x, nullptr, dw, n_segments, bn);
    else if (key == SC_KEY(12, 5, 64, 64) && ho == 6 && wo == 3) segwgrad_launch<64, 64, 12, 5, 5, 1, 1, SC_W_MSPLIT, 6, 3>(st, x, nullptr, dw, n_segments, bn);
    else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 1 && ho == 6 && wo == 3) segwgrad_launch<64, 64, 6, 3, 3, 1, 4, SC_W_MSPLIT, 6, 3>(st, x, nullptr, dw, n_segments, bn);
    else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 0 && ho == 6 && wo == 1) segwgrad_launch<64, 64, 6, 3, 1, 0, 8, SC_W_MSPLIT, 6, 1>(st, x, nullptr, dw, n_segments, bn);
    else return NISQA_ERR_ARG;
    return NQ_LAUNCH_STATUS();
}


// The weight gradient at fp32 OPERAND precision on the bf16 matrix pipe (precision mode 'bf16x6'): x and dz as three exact bf16
// terms, six products; the contract of nisqa_segconv_wgrad_f32 (z == NULL: dz_out holds dz on entry and nothing is folded;
// otherwise the BatchNorm backward runs inside the staging and dz_out, dgamma, dbeta are written).  Three planes per tensor
// and two buffers: half the segments per group of the two-term table where that does not fit 160 KB.
extern "C" int nisqa_segconv_wgrad_bf16x6(const float* x, const float* z, const float* dy, const int32_t* arg, const float* drop,
                                          const float* mean_rstd, const float* gamma, const float* beta, const double* sums2,
                                          float* dz_out, float* dgamma, float* dbeta, float* dw, int32_t n_segments, int32_t h, int32_t w,
                                          int32_t ci, int32_t co, int32_t pad_w, int32_t ho, int32_t wo, void* stream) {
    if (!x || !dz_out || !dw || n_segments <= 0 || !nisqa_segconv_supported(h, w, ci, co, pad_w)) return NISQA_ERR_ARG;
    const bool fold = z != nullptr;
    if (fold && (!dy || !arg || !mean_rstd || !gamma || !beta || !sums2 || !dgamma || !dbeta)) return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const segw_bn bn = {z, dy, arg, drop, mean_rstd, gamma, beta, sums2, dz_out, dgamma, dbeta};
    const int key = SC_KEY(h, w, ci, co);
    NQ_LAUNCH_BEGIN();
    if (!fold) {
        if (key == SC_KEY(24, 7, 16, 32)) segwgrad_launch<16, 32, 24, 7, 7, 1, 1, 1, 0, 0, 3>(st, x, dz_out, dw, n_segments);
        else if (key == SC_KEY(12, 5, 32, 64)) segwgrad_launch<32, 64, 12, 5, 5, 1, 1, 2, 0, 0, 3>(st, x, dz_out, dw, n_segments);
        else if (key == SC_KEY(12, 5, 64, 64)) segwgrad_launch<64, 64, 12, 5, 5, 1, 1, SC_W_MSPLIT, 0, 0, 3>(st, x, dz_out, dw, n_segments);
        else if (pad_w == 1) segwgrad_launch<64, 64, 6, 3, 3, 1, 2, SC_W_MSPLIT, 0, 0, 3>(st, x, dz_out, dw, n_segments);
        else segwgrad_launch<64, 64, 6, 3, 1, 0, 4, SC_W_MSPLIT, 0, 0, 3>(st, x, dz_out, dw, n_segments);
        return NQ_LAUNCH_STATUS();
    }
    if (key == SC_KEY(24, 7, 16, 32) && ho == 12 && wo == 5) segwgrad_launch<16, 32, 24, 7, 7, 1, 1, 1, 12, 5, 3>(st, x, nullptr, dw, n_segments, bn);
    else if (key == SC_KEY(12, 5, 32, 64) && ho == 12 && wo == 5) segwgrad_launch<32, 64, 12, 5, 5, 1, 1, 2, 12, 5, 3>(st, x, nullptr, dw, n_segments, bn);
    else if (key == SC_KEY(12, 5, 64, 64) && ho == 6 && wo == 3) segwgrad_launch<64, 64, 12, 5, 5, 1, 1, SC_W_MSPLIT, 6, 3, 3>(st, x, nullptr, dw, n_segments, bn);
    else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 1 && ho == 6 && wo == 3) segwgrad_launch<64, 64, 6, 3, 3, 1, 2, SC_W_MSPLIT, 6, 3, 3>(st, x, nullptr, dw, n_segments, bn);
    else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 0 && ho == 6 && wo == 1) segwgrad_launch<64, 64, 6, 3, 1, 0, 4, SC_W_MSPLIT, 6, 1, 3>(st, x, nullptr, dw, n_segments, bn);
    else return NISQA_ERR_ARG;
    return NQ_LAUNCH_STATUS();
}

// ======================================================================================================================
// The same weight gradient in EXACT fp32 (precision mode 'f32': the reference's arithmetic) on v_mfma_f32_32x32x2_f32:
// M = co, N = (tap, ci), K = pixels.  The implicit GEMM of train.hip (mode 2) needs 1.57 ms per step for the five layers where
// the forward convolutions -- the same FLOPs -- take 1.14: its K-tiles are gathered from HBM with index arithmetic and its
// accumulators restart per 128-row chunk.  Here, as in the split-bf16 kernel above, a workgroup of eight waves stages the x
// (zero-bordered: a tap is a constant address offset) and dz of SEGS whole segments once per group as fp32 planes
// [pixel][channel], keeps its share of dw in registers over all the groups it walks, and adds it to dw once.  An MFMA takes
// one dword per lane and operand: A = dz[pixel 2 s + (lane >> 5)][co], B = x[that pixel + tap][ci], both ds_read_b32 on 32
// consecutive words per lane half (conflict-free at any row stride).  The eight waves split (M tiles) x (N tiles) x (K steps):
// every wave holds NTW accumulator tiles of one M tile and walks every KSPLIT-th K step.  PH > 0 folds the BatchNorm backward
// into the staging exactly like the split-bf16 kernel (segw_bn).
// ======================================================================================================================
template <int CI, int CO, int H, int W, int WO, int PADW, int SEGS, int MSPLIT, int NSPLIT, int KSPLIT>
struct segwf_cfg {
    static constexpr int RSX = CI, RSZ = CO;                 // words per pixel row
    static constexpr int XPW = W + 2 * PADW, XPH = H + 2;
    static constexpr int PXX = XPH * XPW, PXI = H * W, PXZ = H * WO;
    static constexpr int KROWS = SEGS * PXZ;
    static constexpr int KSTEPS = (KROWS + 1) / 2;
    static constexpr unsigned XB = 4u * SEGS * PXX * RSX;      // bytes of the x plane
    static constexpr unsigned ZBYTES = 4u * (2 * KSTEPS) * RSZ;  // dz plane incl. the zero row behind an odd last pixel
    static constexpr unsigned BUF = XB + ZBYTES;
    static constexpr unsigned LDS = 2u * BUF;
    static constexpr int MT = CO / 32, NTILES = (9 * CI + 31) / 32;
    static constexpr int NTW = (NTILES + NSPLIT - 1) / NSPLIT;
    static constexpr int FX = SEGS * PXI * CI / 4, FZ = SEGS * PXZ * CO / 4;
    static constexpr int NVX = (FX + 511) / 512, NVZ = (FZ + 511) / 512;
    static_assert(MSPLIT * NSPLIT * KSPLIT == 8 && MT == MSPLIT, "eight waves; one M tile per wave");
    static_assert(LDS <= 160 * 1024 && BUF % 16 == 0, "LDS");
};

template <int CI, int CO, int H, int W, int WO, int PADW, int SEGS, int MSPLIT, int NSPLIT, int KSPLIT, int PH, int PW>
__global__ __launch_bounds__(512, 1) void segwgrad_f32_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                              float* __restrict__ dw, int n_segments, segw_bn bn) {
    typedef segwf_cfg<CI, CO, H, W, WO, PADW, SEGS, MSPLIT, NSPLIT, KSPLIT> C;
    constexpr bool BN = PH > 0;
    constexpr bool IDENT = PH == H && PW == WO;
    static_assert(!BN || (H % PH == 0 && 512 % (CO / 4) == 0), "pooling rows are disjoint; a thread keeps its four channels");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid0 = threadIdx.x, lane0 = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int wm = wave % MSPLIT, wn = (wave / MSPLIT) % NSPLIT, wk = wave / (MSPLIT * NSPLIT);
    const int n_groups = (n_segments + SEGS - 1) / SEGS;
    for (unsigned a = 16u * tid0; a < C::LDS; a += 16u * 512u) *(f32x4*)(smem + a) = f32x4{0.f, 0.f, 0.f, 0.f};   // borders, zero rows
    const int hf = lane0 >> 5, l31 = lane0 & 31;
    // this wave's N tiles: wn, wn + NSPLIT, ...; byte offset of the lane's column (tap, ci) inside an x plane
    const int n_own = (C::NTILES - wn + NSPLIT - 1) / NSPLIT;
    unsigned noff[C::NTW];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) {
        const int n = 32 * (wn + NSPLIT * j) + l31;
        const int tap = min(n / CI, 8);                        // columns behind 9 * CI read valid memory and are never stored
        noff[j] = 4u * (unsigned)(((tap / 3) * C::XPW + tap % 3) * C::RSX + n % CI);
    }
    const unsigned zlane = C::XB + 4u * (unsigned)(hf * C::RSZ + 32 * wm + l31);
    f32x16 acc[C::NTW];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) acc[j] = zero16();

    f32x4 vx[C::NVX], vz[C::NVZ];
    f32x4 vd0[BN ? C::NVZ : 1], vd1[(BN && !IDENT) ? C::NVZ : 1], vdr[BN ? C::NVZ : 1];
    sc_i32x4 va0[(BN && !IDENT) ? C::NVZ : 1], va1[(BN && !IDENT) ? C::NVZ : 1];
    f32x4 bn_g, bn_b, bn_mu, bn_rs, bn_m1, bn_m2;
    if (BN) {
        const int ch = (4 * tid0) % CO;
        const double inv = 1.0 / ((double)n_segments * C::PXZ);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float mu = bn.mean_rstd[ch + e], rs = bn.mean_rstd[CO + ch + e];
            bn_mu[e] = mu;
            bn_rs[e] = rs;
            bn_g[e] = bn.gamma[ch + e] * rs;
            bn_b[e] = bn.beta[ch + e] - mu * bn_g[e];
            bn_m1[e] = (float)(bn.sums2[ch + e] * inv);
            bn_m2[e] = (float)((double)rs * (bn.sums2[CO + ch + e] - (double)mu * bn.sums2[ch + e]) * inv);
        }
        if (blockIdx.x == 0 && tid0 < CO) {
            const double mean = bn.mean_rstd[tid0], rstd = bn.mean_rstd[CO + tid0];
            bn.dbeta[tid0] = (float)bn.sums2[tid0];
            bn.dgamma[tid0] = (float)(rstd * (bn.sums2[CO + tid0] - mean * bn.sums2[tid0]));
        }
    }
    auto request = [&](int grp, int tid) {
        const int seg0 = grp * SEGS;
        const int nseg = grp < n_groups ? min(SEGS, n_segments - seg0) : 0;
        const f32x4* gx = (const f32x4*)(x + (size_t)seg0 * C::PXI * CI);
        const f32x4* gz = (const f32x4*)((BN ? bn.z : dz) + (size_t)seg0 * C::PXZ * CO);
#pragma unroll
        for (int j = 0; j < C::NVX; ++j) {
            const int i = tid + 512 * j;
            vx[j] = i < nseg * C::PXI * CI / 4 ? gx[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < C::NVZ; ++j) {
            const int i = tid + 512 * j;
            const bool ok = i < nseg * C::PXZ * CO / 4;
            vz[j] = ok ? gz[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (BN) {
                const int pix = (4 * i) / CO, c = (4 * i) % CO;
                const int sg = pix / C::PXZ, p = pix - sg * C::PXZ;
                vdr[j] = (ok && bn.drop) ? *(const f32x4*)(bn.drop + (size_t)(seg0 + sg) * CO + c) : f32x4{1.f, 1.f, 1.f, 1.f};
                if (IDENT) {
                    vd0[j] = ok ? *(const f32x4*)(bn.dy + ((size_t)(seg0 + sg) * C::PXZ + p) * CO + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    const int yy = p / WO, xx = p - yy * WO;
                    const int oy = yy / (H / PH);
                    int oa = (xx * PW) / WO;
                    if (oa > 0 && sc_win_hi(oa - 1, WO, PW) > xx) --oa;
                    const int ob = oa + 1;
                    const bool has_b = ob < PW && sc_win_lo(ob, WO, PW) <= xx;
                    const size_t o0 = ((size_t)(seg0 + sg) * (PH * PW) + oy * PW + oa) * CO + c;
                    const size_t o1 = ((size_t)(seg0 + sg) * (PH * PW) + oy * PW + ob) * CO + c;
                    vd0[j] = ok ? *(const f32x4*)(bn.dy + o0) : f32x4{0.f, 0.f, 0.f, 0.f};
                    va0[j] = ok ? *(const sc_i32x4*)(bn.arg + o0) : sc_i32x4{-1, -1, -1, -1};
                    vd1[j] = (ok && has_b) ? *(const f32x4*)(bn.dy + o1) : f32x4{0.f, 0.f, 0.f, 0.f};
                    va1[j] = (ok && has_b) ? *(const sc_i32x4*)(bn.arg + o1) : sc_i32x4{-1, -1, -1, -1};
                }
            }
        }
    };
    auto deposit = [&](unsigned buf, int tid, int grp) {
#pragma unroll
        for (int j = 0; j < C::NVX; ++j) {
            const int i = tid + 512 * j;
            if (C::FX % 512 == 0 || i < C::FX) {
                const int pix = (4 * i) / CI, c = (4 * i) % CI;
                const int sg = pix / C::PXI, p = pix - sg * C::PXI, y = p / W, xx = p - y * W;
                lds_st128(buf + 4u * (unsigned)((sg * C::PXX + (y + 1) * C::XPW + xx + PADW) * C::RSX + c), vx[j]);
            }
        }
        const int seg0 = grp * SEGS;
        const int nseg_d = grp < n_groups ? min(SEGS, n_segments - seg0) : 0;
#pragma unroll
        for (int j = 0; j < C::NVZ; ++j) {
            const int i = tid + 512 * j;
            if (C::FZ % 512 == 0 || i < C::FZ) {
                const int pix = (4 * i) / CO, c = (4 * i) % CO;
                f32x4 dzv = vz[j];
                if (BN) {
                    const bool ok = i < nseg_d * C::PXZ * CO / 4;
                    const int sg = pix / C::PXZ, p = pix - sg * C::PXZ;
                    const f32x4 zi = vz[j];
                    f32x4 a4;
                    if (IDENT) a4 = vd0[j];
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) a4[e] = (va0[j][e] == p ? vd0[j][e] : 0.f) + (va1[j][e] == p ? vd1[j][e] : 0.f);
                    }
                    a4 *= vdr[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float zv = zi[e];
                        asm volatile("" : "+v"(zv));
                        const float a_ = fmaf(zv, bn_g[e], bn_b[e]) > 0.f ? a4[e] : 0.f;
                        const float xh = (zv - bn_mu[e]) * bn_rs[e];
                        dzv[e] = ok ? bn_g[e] * (a_ - bn_m1[e] - xh * bn_m2[e]) : 0.f;
                    }
                    if (ok) *(f32x4*)(bn.dz_out + (size_t)seg0 * C::PXZ * CO + (size_t)4 * i) = dzv;
                }
                lds_st128(buf + C::XB + 4u * (unsigned)(pix * C::RSZ + c), dzv);
            }
        }
    };

    int grp = blockIdx.x;
    request(grp, tid0);
    __syncthreads();                                            // the zero fill is complete
    deposit(0u, tid0, grp);
    __syncthreads();
    unsigned cur = 0u;
    for (; grp < n_groups; grp += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        request(grp + gridDim.x, tid);                          // the next group travels while this one is multiplied
        __builtin_amdgcn_sched_barrier(0);
        // K steps wk, wk + KSPLIT, ...: the pixel pair 2 s + (lane >> 5); its dz row is linear in s, its x row is not (rows of
        // the padded plane are XPW pixels apart): two constant divisions per step
        constexpr int NFULL = C::NTILES / NSPLIT;               // tiles every wave owns: no guard around their MFMAs (only a last, odd one)
        float av[2], bv[2][C::NTW];
        auto fetch = [&](int s, int slot) {                      // operands of K step s (one dword per lane and MFMA)
            const int r = min(2 * s + hf, C::KROWS - 1);        // (the row behind an odd last pixel: dz reads zeros, x anything valid)
            const int sg = r / C::PXZ, p = r - sg * C::PXZ, y = p / WO, xo = p - y * WO;
            const unsigned xa = cur + 4u * (unsigned)((sg * C::PXX + y * C::XPW + xo) * C::RSX);
            av[slot] = __uint_as_float(lds_ld32(cur + zlane + 4u * (unsigned)(2 * s * C::RSZ)));
#pragma unroll
            for (int j = 0; j < C::NTW; ++j) bv[slot][j] = (j < NFULL || j < n_own) ? __uint_as_float(lds_ld32(xa + noff[j])) : 0.f;
        };
        fetch(wk, 0);
        int s = wk;
        // two steps per trip so that the operand slots are compile-time; the next step's reads are issued ahead of this step's MFMAs
        for (; s + KSPLIT < C::KSTEPS; s += 2 * KSPLIT) {
            fetch(s + KSPLIT, 1);
#pragma unroll
            for (int j = 0; j < C::NTW; ++j)
                if (j < NFULL || j < n_own) acc[j] = mfma32(av[0], bv[0][j], acc[j]);
            if (s + 2 * KSPLIT < C::KSTEPS) fetch(s + 2 * KSPLIT, 0);
#pragma unroll
            for (int j = 0; j < C::NTW; ++j)
                if (j < NFULL || j < n_own) acc[j] = mfma32(av[1], bv[1][j], acc[j]);
        }
        if (s < C::KSTEPS) {                                     // an odd number of steps: the last one sits in slot 0
#pragma unroll
            for (int j = 0; j < C::NTW; ++j)
                if (j < NFULL || j < n_own) acc[j] = mfma32(av[0], bv[0][j], acc[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        deposit(C::BUF - cur, tid, grp + (int)gridDim.x);
        __syncthreads();
        cur = C::BUF - cur;
    }
    // ---- this workgroup's share of dw
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) {
        const int nt = wn + NSPLIT * j;
        const int col = 32 * nt + l31;
        if (nt < C::NTILES && col < 9 * CI) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * wm + NQ_DROW(r, hf);
                atomicAdd(dw + (size_t)row * (9 * CI) + col, acc[j][r]);
            }
        }
    }
}

template <int CI, int CO, int H, int W, int WO, int PADW, int SEGS, int MSPLIT, int NSPLIT, int KSPLIT, int PH = 0, int PW = 0>
static void segwgrad_f32_launch(hipStream_t st, const float* x, const float* dz, float* dw, int n_segments, segw_bn bn = segw_bn{}) {
    typedef segwf_cfg<CI, CO, H, W, WO, PADW, SEGS, MSPLIT, NSPLIT, KSPLIT> C;
    static std::atomic<bool> attr[SC_MAX_DEV];
    const int dev = sc_device();
    if (!attr[dev].load(std::memory_order_relaxed)) {
        (void)hipFuncSetAttribute((const void*)segwgrad_f32_kernel<CI, CO, H, W, WO, PADW, SEGS, MSPLIT, NSPLIT, KSPLIT, PH, PW>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
        attr[dev].store(true, std::memory_order_relaxed);
    }
    const int n_groups = (n_segments + SEGS - 1) / SEGS;
    const int grid = n_groups < sc_cu_count() ? n_groups : sc_cu_count();
    hipLaunchKernelGGL((segwgrad_f32_kernel<CI, CO, H, W, WO, PADW, SEGS, MSPLIT, NSPLIT, KSPLIT, PH, PW>), dim3(grid), dim3(512), C::LDS, st,
                       x, dz, dw, n_segments, bn);
}

// exact fp32, same five shapes and the same contract as nisqa_segconv_wgrad_bf16 / nisqa_segconv_wgrad_bn_bf16 (z == NULL: dz_out
// holds dz on entry and nothing is folded; otherwise the BatchNorm backward runs inside and dz_out, dgamma, dbeta are written)
extern "C" int nisqa_segconv_wgrad_f32(const float* x, const float* z, const float* dy, const int32_t* arg, const float* drop,
                                       const float* mean_rstd, const float* gamma, const float* beta, const double* sums2,
                                       float* dz_out, float* dgamma, float* dbeta, float* dw, int32_t n_segments, int32_t h, int32_t w,
                                       int32_t ci, int32_t co, int32_t pad_w, int32_t ho, int32_t wo, void* stream) {
    if (!x || !dz_out || !dw || n_segments <= 0 || !nisqa_segconv_supported(h, w, ci, co, pad_w)) return NISQA_ERR_ARG;
    const bool fold = z != nullptr;
    if (fold && (!dy || !arg || !mean_rstd || !gamma || !beta || !sums2 || !dgamma || !dbeta)) return NISQA_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const segw_bn bn = {z, dy, arg, drop, mean_rstd, gamma, beta, sums2, dz_out, dgamma, dbeta};
    const int key = SC_KEY(h, w, ci, co);
    NQ_LAUNCH_BEGIN();
    if (!fold) {
        if (key == SC_KEY(24, 7, 16, 32)) segwgrad_f32_launch<16, 32, 24, 7, 7, 1, 1, 1, 1, 8>(st, x, dz_out, dw, n_segments);
        else if (key == SC_KEY(12, 5, 32, 64)) segwgrad_f32_launch<32, 64, 12, 5, 5, 1, 1, 2, 2, 2>(st, x, dz_out, dw, n_segments);
        else if (key == SC_KEY(12, 5, 64, 64)) segwgrad_f32_launch<64, 64, 12, 5, 5, 1, 1, 2, 4, 1>(st, x, dz_out, dw, n_segments);
        else if (pad_w == 1) segwgrad_f32_launch<64, 64, 6, 3, 3, 1, 4, 2, 4, 1>(st, x, dz_out, dw, n_segments);
        else segwgrad_f32_launch<64, 64, 6, 3, 1, 0, 8, 2, 4, 1>(st, x, dz_out, dw, n_segments);
        return NQ_LAUNCH_STATUS();
    }
    if (key == SC_KEY(24, 7, 16, 32) && ho == 12 && wo == 5) segwgrad_f32_launch<16, 32, 24, 7, 7, 1, 1, 1, 1, 8, 12, 5>(st, x, nullptr, dw, n_segments, bn);
    else if (key == SC_KEY(12, 5, 32, 64) && ho == 12 && wo == 5) segwgrad_f32_launch<32, 64, 12, 5, 5, 1, 1, 2, 2, 2, 12, 5>(st, x, nullptr, dw, n_segments, bn);
    else if (key == SC_KEY(12, 5, 64, 64) && ho == 6 && wo == 3) segwgrad_f32_launch<64, 64, 12, 5, 5, 1, 1, 2, 4, 1, 6, 3>(st, x, nullptr, dw, n_segments, bn);
    else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 1 && ho == 6 && wo == 3) segwgrad_f32_launch<64, 64, 6, 3, 3, 1, 4, 2, 4, 1, 6, 3>(st, x, nullptr, dw, n_segments, bn);
    else if (key == SC_KEY(6, 3, 64, 64) && pad_w == 0 && ho == 6 && wo == 1) segwgrad_f32_launch<64, 64, 6, 3, 1, 0, 8, 2, 4, 1, 6, 1>(st, x, nullptr, dw, n_segments, bn);
    else return NISQA_ERR_ARG;
    return NQ_LAUNCH_STATUS();
}
