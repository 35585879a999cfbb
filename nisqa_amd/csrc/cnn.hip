// Framewise AdaptCNN for gfx950 -- replaces segment_specs + Framewise.forward + AdaptCNN.forward
// (reference nisqa/NISQA_lib.py:2239-2282, 487-502, 688-710), eval mode, BatchNorm folded.
//
// Design (MI355X-first, see DESIGN.md "CNN"):
//   * ONE WAVE OWNS ONE SEGMENT from the mel spectrogram to the pooled conv4 output.  Every
//     activation lives in a 15 KiB wave-private LDS region, so there is no inter-wave barrier
//     anywhere and up to 8-10 waves share a CU (MFMA of one wave overlaps the VALU/LDS phases
//     of another).  The 15-frame window is read straight out of the frame-major spectrogram:
//     the reference's [1300,1,48,15] padded segment tensor (3.7 MB / clip) is never built.
//   * conv2..conv6 are implicit GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32).  Output channels
//     sit on the D-fragment's columns (= lanes) and pixels on its rows (= registers), and the
//     row->pixel assignment is chosen so that EVERY adaptive-max-pool window falls inside one
//     lane's registers: pooling is a handful of v_max in the epilogue, no cross-lane traffic.
//   * activations are stored pixel-major [pixel][channel] with an XOR swizzle on the 16-byte
//     chunk index, which makes both the ds_read_b128 operand gathers (consecutive pixels, same
//     channel chunk) and the ds_write_b32 epilogue stores bank-conflict free without padding.
//   * weight fragments come pre-packed from the host and are streamed from L2 with one
//     contiguous 1 KiB global_load_dwordx4 per wave, K-step and column tile (f32 MFMA needs
//     only ~2 B/clk/wave of operands, far below L1/L2 bandwidth).
//   * conv5/conv6 have only 18 / 6 output pixels per segment, so a second kernel batches FOUR
//     segments per wave (72 / 24 rows of the 32-row MFMA tile).
#include "common.hpp"
#include "layout.hpp"
#include "conv_f32.hpp"
#include "../../include/nisqa_hip.h"

#define FRONT_ACT_BYTES 15360            /* max(A1 10752 + input 2880, A3 15360) */
#define FRONT_ZERO_OFF FRONT_ACT_BYTES   /* 256 B of zeros: target of out-of-image taps */
#define FRONT_LDS_BYTES (FRONT_ACT_BYTES + 256)
#define FRONT_IN_OFF 10752               /* conv1 input patch [15 frames][48 mels] */

#define BACK_ACT_BYTES 18432             /* 4 segments x 18 px x 64 ch x 4 B */
#define BACK_ZERO_OFF BACK_ACT_BYTES
#define BACK_LDS_BYTES (BACK_ACT_BYTES + 256)
#define NQ_FIRST_ROUND_BLOCKS (256 * 8)   /* single-wave workgroups resident at launch: 256 CUs x 2 waves/SIMD */

// adaptive_max_pool2d window starts/ends along the width (height windows are always [2a, 2a+2))
//   15 -> 7 : [2b, 2b+3)           7 -> 5 : (0,2)(1,3)(2,5)(4,6)(5,7)        5 -> 3 : (0,2)(1,4)(3,5)
__device__ constexpr int win75_lo(int b) { return b == 0 ? 0 : b == 1 ? 1 : b == 2 ? 2 : b == 3 ? 4 : 5; }
__device__ constexpr int win75_hi(int b) { return b == 0 ? 2 : b == 1 ? 3 : b == 2 ? 5 : b == 3 ? 6 : 7; }
__device__ constexpr int win53_lo(int b) { return b == 0 ? 0 : b == 1 ? 1 : 3; }
__device__ constexpr int win53_hi(int b) { return b == 0 ? 2 : b == 1 ? 4 : 5; }

// ---------------------------------------------------------------------------------------------
// conv1 (VALU) + pool1 + conv2 + pool2 + conv3 + conv4 + pool3, one wave per segment.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64, 2) void cnn_front_kernel(
    const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off,
    const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
    const float* __restrict__ clip_floor, int n_clips, int seg_hop,
    const float* __restrict__ cw, float* __restrict__ p3,
    const float* __restrict__ seg_x, int seg_L, int stagger_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int p = blockIdx.x;                       // padded token index
    const int b = find_segment(tok_off, n_clips, p);
    const int k = p - tok_off[b];
    if (k >= n_wins[b]) return;                     // padding token of this clip
    stagger_odd_wave_slot(stagger_blocks, 5);

    // ---- stage the 15-frame window (frame-major: 720 contiguous floats), apply the top_db floor
    {
        float* in_lds = (float*)(smem + FRONT_IN_OFF);
        if (seg_x) {
            // segment-tensor mode: x[b][k][0][48][15] as the reference's model.forward receives it
            const float* src = seg_x + ((size_t)b * seg_L + k) * 720;
            for (int i = lane; i < 720; i += 64) {
                const int m = i / 15;
                in_lds[(i - 15 * m) * 48 + m] = src[i];
            }
        } else {
            const float* src = mel_tm + (size_t)(frame_off[b] + k * seg_hop) * 48;
            const float fl = clip_floor[b];
            for (int i = lane; i < 720; i += 64) in_lds[i] = fmaxf(src[i], fl);
        }
        ((float*)(smem + FRONT_ZERO_OFF))[lane] = 0.f;
    }
    __syncthreads();

    // ---- conv1 1->16 (3x3, pad 1) + BN + ReLU + adaptive max pool 48x15 -> 24x7, on the VALU.
    //      lane <-> pooled pixel; weights are wave-uniform (scalar loads).
    {
        const float* in_lds = (const float*)(smem + FRONT_IN_OFF);
        const float* w1 = cw + CNN_W1;
        const float* t1 = cw + CNN_T1;
        for (int pp = lane; pp < 168; pp += 64) {
            const int a = pp / 7, bc = pp % 7;
            float v[4][5];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const int m = 2 * a - 1 + r, j = 2 * bc - 1 + c;
                    const bool ok = (unsigned)m < 48u && (unsigned)j < 15u;
                    v[r][c] = ok ? in_lds[j * 48 + m] : 0.f;
                }
            f32x4 o4[4];
#pragma unroll
            for (int ch = 0; ch < 16; ++ch) {
                float w[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) w[q] = w1[ch * 9 + q];
                float mx = -3.0e38f;
#pragma unroll
                for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                    for (int xx = 0; xx < 3; ++xx) {
                        float o = 0.f;
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) o = fmaf(w[dy * 3 + dx], v[yy + dy][xx + dx], o);
                        mx = fmaxf(mx, o);
                    }
                o4[ch >> 2][ch & 3] = fmaxf(mx + t1[ch], 0.f);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(f32x4*)(smem + pp * 64 + ((q ^ ((pp >> 2) & 3)) << 4)) = o4[q];
        }
    }
    __syncthreads();

    const int i = lane & 31;                 // A-fragment row of this lane
    const int hfi = (i >> 2) & 1;            // D-fragment lane half that owns row i
    const int qi = (i & 3) + 4 * (i >> 3);   // index of row i among that half's 16 rows
    const int n = lane & 31, hf = lane >> 5; // D-fragment column / lane half

    // ---- conv2 16->32 on 24x7, pool -> 12x5.  Each lane half owns 6 pooled rows = 6 groups of
    //      14 pixels (2 rows x 7 cols); local pixel u = 14*gl + 7*yy + x  ->  tile u>>4, reg u&15.
    {
        f32x16 acc[6][1];
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[t][0] = zero16();
        int py[6], px[6], pbase[6];
        bool pv[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int u = 16 * t + qi;
            pv[t] = u < 84;
            const int gl = u / 14, w = u % 14, yy = w / 7;
            py[t] = 2 * (6 * hfi + gl) + yy;
            px[t] = w - 7 * yy;
            pbase[t] = 0;
        }
        conv3x3_mfma<16, 6, 1, 24, 7, FRONT_ZERO_OFF>(acc, smem, (const f32x4*)(cw + CNN_WF2), py, px, pbase, pv, lane);
        __syncthreads();                      // A1 fully consumed; A2 aliases it
        const float tn = cw[CNN_T2 + n];
#pragma unroll
        for (int gl = 0; gl < 6; ++gl)
#pragma unroll
            for (int bb = 0; bb < 5; ++bb) {
                float mx = -3.0e38f;
#pragma unroll
                for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                    for (int x = win75_lo(bb); x < win75_hi(bb); ++x) {
                        const int u = 14 * gl + 7 * yy + x;
                        mx = fmaxf(mx, acc[u >> 4][0][u & 15]);
                    }
                const int pp = (6 * hf + gl) * 5 + bb;
                *(float*)(smem + pp * 128 + (((n >> 2) ^ ((pp >> 1) & 7)) << 4) + (n & 3) * 4) = fmaxf(mx + tn, 0.f);
            }
    }
    __syncthreads();

    // conv3 / conv4 work on 12x5 = 60 pixels: each lane half owns 3 pooled rows = 3 groups of
    // 10 pixels (2 rows x 5 cols); local pixel u = 10*gl + 5*yy + x -> tile u>>4, reg u&15.
    int py[2], px[2], pbase[2];
    bool pv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int u = 16 * t + qi;
        pv[t] = u < 30;
        const int gl = u / 10, w = u % 10, yy = w / 5;
        py[t] = 2 * (3 * hfi + gl) + yy;
        px[t] = w - 5 * yy;
        pbase[t] = 0;
    }

    // ---- conv3 32->64 on 12x5
    {
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[t][nt] = zero16();
        conv3x3_mfma<32, 2, 2, 12, 5, FRONT_ZERO_OFF>(acc, smem, (const f32x4*)(cw + CNN_WF3), py, px, pbase, pv, lane);
        __syncthreads();                      // A2 fully consumed; A3 aliases it
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = n + 32 * nt;
            const float tn = cw[CNN_T3 + c];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int u = 16 * t + r;
                    if (u < 30) {
                        const int gl = u / 10, w = u % 10, yy = w / 5, x = w - 5 * yy;
                        const int pp = (2 * (3 * hf + gl) + yy) * 5 + x;
                        *(float*)(smem + pp * 256 + (((c >> 2) ^ (pp & 15)) << 4) + (c & 3) * 4) =
                            fmaxf(acc[t][nt][r] + tn, 0.f);
                    }
                }
        }
    }
    __syncthreads();

    // ---- conv4 64->64 on 12x5, pool -> 6x3, straight to HBM as p3[token][18][64]
    {
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[t][nt] = zero16();
        conv3x3_mfma<64, 2, 2, 12, 5, FRONT_ZERO_OFF>(acc, smem, (const f32x4*)(cw + CNN_WF4), py, px, pbase, pv, lane);
        float* dst = p3 + (size_t)p * (18 * 64);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = n + 32 * nt;
            const float tn = cw[CNN_T4 + c];
#pragma unroll
            for (int gl = 0; gl < 3; ++gl)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb) {
                    float mx = -3.0e38f;
#pragma unroll
                    for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                        for (int x = win53_lo(bb); x < win53_hi(bb); ++x) {
                            const int u = 10 * gl + 5 * yy + x;
                            mx = fmaxf(mx, acc[u >> 4][nt][u & 15]);
                        }
                    dst[((3 * hf + gl) * 3 + bb) * 64 + c] = fmaxf(mx + tn, 0.f);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// conv5 + conv6 (kernel 3x3 on a 6x3 image, padding (1,0) -> 6x1), four segments per wave.
// conv6 is evaluated as a padding-1 conv at the centre column x = 1 of the 3-wide image, which is
// the same sum as the reference's (3 x pool_3[1]) kernel with no width padding (NISQA_lib.py:672-676).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64, 2) void cnn_back_kernel(
    const float* __restrict__ p3, const int32_t* __restrict__ tok_off,
    const int32_t* __restrict__ n_wins, int n_clips, const float* __restrict__ cw,
    float* __restrict__ feat, int stagger_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int p0 = blockIdx.x * 4;                  // tok_off is a multiple of 32: no clip straddling
    const int b = find_segment(tok_off, n_clips, p0);
    const int nvalid = min(4, n_wins[b] - (p0 - tok_off[b]));
    if (nvalid <= 0) return;
    stagger_odd_wave_slot(stagger_blocks, 5);

    {
        const f32x4* src = (const f32x4*)p3 + (size_t)p0 * (18 * 16);
#pragma unroll
        for (int it = 0; it < 18; ++it) {
            const int q = lane + 64 * it;
            const int sp = q >> 4, chunk = q & 15;   // (slot, pixel) linear index 0..71
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (sp < nvalid * 18) v = src[q];
            *(f32x4*)(smem + sp * 256 + ((chunk ^ (sp & 15)) << 4)) = v;
        }
        ((float*)(smem + BACK_ZERO_OFF))[lane] = 0.f;
    }
    __syncthreads();

    const int i = lane & 31, n = lane & 31, hf = lane >> 5;

    // ---- conv5 64->64 on 6x3: rows rho = 32*t + i  <->  (slot = rho/18, pixel = rho%18)
    {
        f32x16 acc[3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[t][nt] = zero16();
        int py[3], px[3], pbase[3];
        bool pv[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int rho = 32 * t + i;
            pv[t] = rho < 72;
            const int slot = rho / 18, pix = rho - 18 * slot;
            py[t] = pix / 3;
            px[t] = pix - 3 * py[t];
            pbase[t] = slot * 18;
        }
        conv3x3_mfma<64, 3, 2, 6, 3, BACK_ZERO_OFF>(acc, smem, (const f32x4*)(cw + CNN_WF5), py, px, pbase, pv, lane);
        __syncthreads();                       // conv5 input fully consumed; output aliases it
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = n + 32 * nt;
            const float tn = cw[CNN_T5 + c];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rho = 32 * t + NQ_DROW(r, hf);
                    if (rho < 72)
                        *(float*)(smem + rho * 256 + (((c >> 2) ^ (rho & 15)) << 4) + (c & 3) * 4) =
                            fmaxf(acc[t][nt][r] + tn, 0.f);
                }
        }
    }
    __syncthreads();

    // ---- conv6: rows rho = i <-> (slot = i/6, y = i%6), centre column
    {
        f32x16 acc[1][2];
        acc[0][0] = zero16();
        acc[0][1] = zero16();
        int py[1], px[1], pbase[1];
        bool pv[1];
        pv[0] = i < 24;
        const int slot = i / 6;
        py[0] = i - 6 * slot;
        px[0] = 1;
        pbase[0] = slot * 18;
        conv3x3_mfma<64, 1, 2, 6, 3, BACK_ZERO_OFF>(acc, smem, (const f32x4*)(cw + CNN_WF6), py, px, pbase, pv, lane);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = n + 32 * nt;
            const float tn = cw[CNN_T6 + c];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rho = NQ_DROW(r, hf);
                const int s6 = rho / 6, y = rho - 6 * s6;
                if (rho < 24 && s6 < nvalid)
                    feat[(size_t)(p0 + s6) * 384 + c * 6 + y] = fmaxf(acc[0][nt][r] + tn, 0.f);
            }
        }
    }
}

extern "C" int nisqa_cnn_front(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                               const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                               int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w, float* p3_ws,
                               void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31) || seg_hop <= 0) return 1;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(cnn_front_kernel, dim3(total_tok_padded), dim3(64), FRONT_LDS_BYTES, (hipStream_t)stream,
                       mel_tm, frame_off, tok_off, n_wins, clip_floor, n_clips, seg_hop, cnn_w, p3_ws,
                       (const float*)nullptr, 0, NQ_FIRST_ROUND_BLOCKS);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_cnn_adapt_segments(const float* x, int32_t seg_len_padded, const int32_t* tok_off,
                                        const int32_t* n_wins, int32_t n_clips, int32_t total_tok_padded,
                                        const float* cnn_w, float* p3_ws, float* feat, void* stream) {
    if (!x || seg_len_padded <= 0 || n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31)) return 1;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(cnn_front_kernel, dim3(total_tok_padded), dim3(64), FRONT_LDS_BYTES, (hipStream_t)stream,
                       (const float*)nullptr, (const int32_t*)nullptr, tok_off, n_wins, (const float*)nullptr, n_clips,
                       1, cnn_w, p3_ws, x, seg_len_padded, NQ_FIRST_ROUND_BLOCKS);
    const int rc = NQ_LAUNCH_STATUS();
    if (rc) return rc;
    return nisqa_cnn_back(p3_ws, tok_off, n_wins, n_clips, total_tok_padded, cnn_w, feat, stream);
}

extern "C" int nisqa_cnn_back(const float* p3_ws, const int32_t* tok_off, const int32_t* n_wins, int32_t n_clips,
                              int32_t total_tok_padded, const float* cnn_w, float* feat, void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31)) return 1;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(cnn_back_kernel, dim3(total_tok_padded / 4), dim3(64), BACK_LDS_BYTES, (hipStream_t)stream,
                       p3_ws, tok_off, n_wins, n_clips, cnn_w, feat, NQ_FIRST_ROUND_BLOCKS);
    return NQ_LAUNCH_STATUS();
}

extern "C" int nisqa_cnn_adapt(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                               const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                               int32_t total_tok_padded, int32_t seg_hop, const float* cnn_w,
                               float* p3_ws, float* feat, void* stream) {
    int rc = nisqa_cnn_front(mel_tm, frame_off, tok_off, n_wins, clip_floor, n_clips, total_tok_padded, seg_hop,
                             cnn_w, p3_ws, stream);
    if (rc) return rc;
    return nisqa_cnn_back(p3_ws, tok_off, n_wins, n_clips, total_tok_padded, cnn_w, feat, stream);
}
