// StandardCNN (fixed 2x2 max-pools) + fc_out for the nisqa_tts.tar architecture -- replaces segment_specs +
// Framewise.forward + StandardCNN.forward (reference nisqa/NISQA_lib.py:2239-2282, 487-502, 811-836), eval
// mode, BatchNorm folded, exact fp32 on v_mfma_f32_32x32x2_f32.
//
// Same construction as cnn.hip (one wave owns one segment, wave-private swizzled LDS activations, pooling
// in-lane through the row->pixel map, weight fragments streamed from L2); only the geometry differs:
//   48x15 -conv1-> pool 2x2 pad (0,1) -> 24x8 -conv2-> pool -> 12x4 -conv3,conv4-> pool -> 6x2 -conv5,conv6->
//   64 x 6 x 2 = 768 -fc_out-> 20.
// The 2x2 windows make the maps regular: a pooled row of conv2 is exactly one 16-row half tile (100 % tile
// efficiency), conv3/4 use 24 of 32 rows.  conv5/conv6/fc batch four segments per wave.
#include "common.hpp"
#include "layout.hpp"
#include "conv_f32.hpp"
#include "../../include/nisqa_hip.h"

#define SF_ACT 15360
#define SF_ZERO SF_ACT
#define SF_LDS (SF_ACT + 256)
#define SF_IN 12288                    /* conv1 input patch behind the A1 activations (192 px x 16 ch) */
#define SB_ACT 12288                   /* 4 segments x 12 px x 64 ch x 4 B */
#define SB_ZERO SB_ACT
#define SB_LDS (SB_ACT + 256)

__global__ __launch_bounds__(64, 2) void cnn_std_front_kernel(
    const float* __restrict__ mel_tm, const int32_t* __restrict__ frame_off,
    const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
    const float* __restrict__ clip_floor, int n_clips, int seg_hop,
    const float* __restrict__ cw, float* __restrict__ p3) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int p = blockIdx.x;
    const int b = find_segment(tok_off, n_clips, p);
    const int k = p - tok_off[b];
    if (k >= n_wins[b]) return;

    {
        float* in_lds = (float*)(smem + SF_IN);
        const float* src = mel_tm + (size_t)(frame_off[b] + k * seg_hop) * 48;
        const float fl = clip_floor[b];
        for (int i = lane; i < 720; i += 64) in_lds[i] = fmaxf(src[i], fl);
        ((float*)(smem + SF_ZERO))[lane] = 0.f;
    }
    __syncthreads();

    // ---- conv1 + MaxPool2d(2, stride 2, padding (0,1)): pooled column bq covers conv columns {2bq-1, 2bq}
    {
        const float* in_lds = (const float*)(smem + SF_IN);
        const float* w1 = cw + CNN_W1;
        const float* t1 = cw + CNN_T1;
#pragma unroll 1
        for (int pp = lane; pp < 192; pp += 64) {
            const int a = pp >> 3, bq = pp & 7;
            float v[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int m = 2 * a - 1 + r, j = 2 * bq - 2 + c;
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
                    for (int xx = 0; xx < 2; ++xx) {
                        float o = 0.f;
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) o = fmaf(w[dy * 3 + dx], v[yy + dy][xx + dx], o);
                        if (xx == 0 && bq == 0) o = -3.0e38f;       // conv column -1 is max-pool padding
                        mx = fmaxf(mx, o);
                    }
                o4[ch >> 2][ch & 3] = fmaxf(mx + t1[ch], 0.f);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4*)(smem + pp * 64 + ((q ^ ((pp >> 2) & 3)) << 4)) = o4[q];
        }
    }
    __syncthreads();

    const int i = lane & 31, hfi = (i >> 2) & 1, qi = (i & 3) + 4 * (i >> 3);
    const int n = lane & 31, hf = lane >> 5;

    // ---- conv2 16->32 on 24x8, pool -> 12x4: tile t of a lane half = pooled row 6*half + t (2 rows x 8 cols)
    {
        f32x16 acc[6][1];
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[t][0] = zero16();
        int py[6], px[6], pbase[6];
        bool pv[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            pv[t] = true;
            py[t] = 2 * (6 * hfi + t) + (qi >> 3);
            px[t] = qi & 7;
            pbase[t] = 0;
        }
        conv3x3_mfma<16, 6, 1, 24, 8, SF_ZERO>(acc, smem, (const f32x4*)(cw + CNN_WF2), py, px, pbase, pv, lane);
        __syncthreads();
        const float tn = cw[CNN_T2 + n];
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const float mx = fmaxf(fmaxf(acc[t][0][2 * bb], acc[t][0][2 * bb + 1]),
                                       fmaxf(acc[t][0][8 + 2 * bb], acc[t][0][8 + 2 * bb + 1]));
                const int pp = (6 * hf + t) * 4 + bb;
                *(float*)(smem + pp * 128 + (((n >> 2) ^ ((pp >> 1) & 7)) << 4) + (n & 3) * 4) = fmaxf(mx + tn, 0.f);
            }
    }
    __syncthreads();

    // conv3 / conv4 on 12x4: a lane half owns 3 pooled rows = 3 groups of 8 pixels; u = 8*gl + 4*yy + x
    int py[2], px[2], pbase[2];
    bool pv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int u = 16 * t + qi;
        pv[t] = u < 24;
        py[t] = 2 * (3 * hfi + (u >> 3)) + ((u >> 2) & 1);
        px[t] = u & 3;
        pbase[t] = 0;
    }
    {
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[t][nt] = zero16();
        conv3x3_mfma<32, 2, 2, 12, 4, SF_ZERO>(acc, smem, (const f32x4*)(cw + CNN_WF3), py, px, pbase, pv, lane);
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = n + 32 * nt;
            const float tn = cw[CNN_T3 + c];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int u = 16 * t + r;
                    if (u < 24) {
                        const int pp = (2 * (3 * hf + (u >> 3)) + ((u >> 2) & 1)) * 4 + (u & 3);
                        *(float*)(smem + pp * 256 + (((c >> 2) ^ (pp & 15)) << 4) + (c & 3) * 4) =
                            fmaxf(acc[t][nt][r] + tn, 0.f);
                    }
                }
        }
    }
    __syncthreads();
    {
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[t][nt] = zero16();
        conv3x3_mfma<64, 2, 2, 12, 4, SF_ZERO>(acc, smem, (const f32x4*)(cw + CNN_WF4), py, px, pbase, pv, lane);
        float* dst = p3 + (size_t)p * (12 * 64);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = n + 32 * nt;
            const float tn = cw[CNN_T4 + c];
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
                    dst[((3 * hf + gl) * 2 + bb) * 64 + c] = fmaxf(mx + tn, 0.f);
                }
        }
    }
}

// conv5 + conv6 (both 3x3, padding 1, on 6x2) + fc_out 768 -> 20; four segments per wave
__global__ __launch_bounds__(64, 2) void cnn_std_back_kernel(
    const float* __restrict__ p3, const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
    int n_clips, const float* __restrict__ cw, float* __restrict__ feat20) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int p0 = blockIdx.x * 4;
    const int b = find_segment(tok_off, n_clips, p0);
    const int nvalid = min(4, n_wins[b] - (p0 - tok_off[b]));
    if (nvalid <= 0) return;
    {
        const f32x4* src = (const f32x4*)p3 + (size_t)p0 * (12 * 16);
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const int q = lane + 64 * it;
            const int sp = q >> 4, chunk = q & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (sp < nvalid * 12) v = src[q];
            *(f32x4*)(smem + sp * 256 + ((chunk ^ (sp & 15)) << 4)) = v;
        }
        ((float*)(smem + SB_ZERO))[lane] = 0.f;
    }
    __syncthreads();
    const int i = lane & 31, n = lane & 31, hf = lane >> 5;
    int py[2], px[2], pbase[2];
    bool pv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int rho = 32 * t + i;
        pv[t] = rho < 48;
        const int slot = rho / 12, pix = rho - 12 * slot;
        py[t] = pix >> 1;
        px[t] = pix & 1;
        pbase[t] = slot * 12;
    }
#pragma unroll 1
    for (int layer = 0; layer < 2; ++layer) {
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[t][nt] = zero16();
        conv3x3_mfma<64, 2, 2, 6, 2, SB_ZERO>(acc, smem, (const f32x4*)(cw + (layer ? CNN_WF6 : CNN_WF5)), py, px, pbase, pv, lane);
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = n + 32 * nt;
            const float tn = cw[(layer ? CNN_T6 : CNN_T5) + c];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rho = 32 * t + NQ_DROW(r, hf);
                    if (rho < 48)
                        *(float*)(smem + rho * 256 + (((c >> 2) ^ (rho & 15)) << 4) + (c & 3) * 4) =
                            fmaxf(acc[t][nt][r] + tn, 0.f);
                }
        }
        __syncthreads();
    }
    // ---- fc_out: 16 lanes per segment slot, lane sub-index i16 takes k' = i16 + 16 m (k' = pixel*64 + c)
    {
        const int slot = lane >> 4, i16 = lane & 15;
        float o[20];
#pragma unroll
        for (int j = 0; j < 20; ++j) o[j] = 0.f;
        const float* wfc = cw + CNNS_FC_W;
#pragma unroll 2
        for (int m = 0; m < 48; ++m) {
            const int kq = i16 + 16 * m;
            const int pix = slot * 12 + (kq >> 6), c = kq & 63;
            const float a = *(const float*)(smem + pix * 256 + (((c >> 2) ^ (pix & 15)) << 4) + (c & 3) * 4);
            const f32x4* wr = (const f32x4*)(wfc + kq * 20);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const f32x4 w4 = wr[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 * q + e] = fmaf(a, w4[e], o[4 * q + e]);
            }
        }
#pragma unroll
        for (int j = 0; j < 20; ++j) {
            float v = o[j];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            o[j] = v;
        }
        if (slot < nvalid) {
#pragma unroll
            for (int j = 0; j < 20; ++j)
                if (i16 == (j & 15)) feat20[(size_t)(p0 + slot) * 20 + j] = o[j] + cw[CNNS_FC_B + j];
        }
    }
}

extern "C" int nisqa_cnn_standard(const float* mel_tm, const int32_t* frame_off, const int32_t* tok_off,
                                  const int32_t* n_wins, const float* clip_floor, int32_t n_clips,
                                  int32_t total_tok_padded, int32_t seg_hop, const float* cnn_std_w,
                                  float* p3_ws, float* feat20, void* stream) {
    if (n_clips <= 0 || total_tok_padded <= 0 || (total_tok_padded & 31) || seg_hop <= 0) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(cnn_std_front_kernel, dim3(total_tok_padded), dim3(64), SF_LDS, (hipStream_t)stream, mel_tm,
                       frame_off, tok_off, n_wins, clip_floor, n_clips, seg_hop, cnn_std_w, p3_ws);
    hipLaunchKernelGGL(cnn_std_back_kernel, dim3(total_tok_padded / 4), dim3(64), SB_LDS, (hipStream_t)stream,
                       (const float*)p3_ws, tok_off, n_wins, n_clips, cnn_std_w, feat20);
    return NQ_LAUNCH_STATUS();
}
