// Bidirectional single-layer LSTM (hidden 128, input 20) + last-step pooling for the nisqa_tts.tar
// architecture -- replaces LSTM.forward (pack / nn.LSTM / pad, reference nisqa/NISQA_lib.py:925-943) and
// PoolLastStepBi.forward (NL:1107-1115).
//
// The recurrence is sequential in time (up to 5,986 steps), so parallelism comes from clips x directions x
// gate rows: one 512-thread workgroup per (clip, direction); thread i owns gate row i (PyTorch order i,f,g,o)
// and keeps its W_hh row (128 floats) and W_ih row (20 floats) in REGISTERS for the whole clip; h lives in
// LDS (double-buffered, read as wave-broadcast float4s), x_t arrives by scalar loads.  Two barriers per step.
// Only what the pooling needs leaves the kernel: the forward direction's last state and the backward
// direction's state at position 0 (hfin[clip][2][128]); the full [n,256] sequence is written only when a
// caller asks for it (parity tests).
#include "common.hpp"
#include "layout.hpp"
#include "../../include/nisqa_hip.h"

NQ_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(512, 2) void lstm_dir_kernel(
    const float* __restrict__ feat20, const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
    const float* __restrict__ lw, float* __restrict__ hfin, float* __restrict__ seq) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][128];
    __shared__ float gates[512];
    const int i = threadIdx.x, b = blockIdx.x, dir = blockIdx.y;
    const int n = n_wins[b], c0 = tok_off[b];
    const float* w = lw + (size_t)dir * LSTM_DIR_FLOATS;
    f32x4 whh[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) whh[q] = *(const f32x4*)(w + LSTM_WHH + (size_t)i * 128 + 4 * q);
    float wih[20];
#pragma unroll
    for (int j = 0; j < 20; ++j) wih[j] = w[LSTM_WIH + i * 20 + j];
    const float bias = w[LSTM_B + i];
    float c = 0.f, h = 0.f;
    if (i < 128) hbuf[0][i] = 0.f;
    __syncthreads();
    for (int t = 0; t < n; ++t) {
        const int tok = c0 + (dir == 0 ? t : n - 1 - t);
        const float* x = feat20 + (size_t)tok * 20;             // wave-uniform: scalar loads
        float a0 = bias, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int j = 0; j < 20; ++j) a0 = fmaf(wih[j], x[j], a0);
        const f32x4* hp = (const f32x4*)hbuf[t & 1];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const f32x4 hv = hp[q];
            a0 = fmaf(whh[q][0], hv[0], a0);
            a1 = fmaf(whh[q][1], hv[1], a1);
            a2 = fmaf(whh[q][2], hv[2], a2);
            a3 = fmaf(whh[q][3], hv[3], a3);
        }
        gates[i] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (i < 128) {
            const float ig = sigmoidf_(gates[i]), fg = sigmoidf_(gates[128 + i]);
            const float gg = tanhf(gates[256 + i]), og = sigmoidf_(gates[384 + i]);
            c = fg * c + ig * gg;
            h = og * tanhf(c);
            hbuf[(t + 1) & 1][i] = h;
            if (seq) seq[(size_t)tok * 256 + dir * 128 + i] = h;
        }
        __syncthreads();
    }
    if (i < 128) hfin[((size_t)b * 2 + dir) * 128 + i] = h;
}

__global__ __launch_bounds__(64) void pool_last_kernel(const float* __restrict__ hfin, const float* __restrict__ lw,
                                                       float* __restrict__ out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* w = lw + LSTM_POOL_W;
    const float* hf = hfin + (size_t)b * 256;                     // [fwd last step | bwd at position 0]
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) s = fmaf(w[lane + 64 * q], hf[lane + 64 * q], s);
    s = wave_sum(s);
    if (lane == 0) out[b] = s + w[256];
}

extern "C" int nisqa_lstm_laststep(const float* feat20, const int32_t* tok_off, const int32_t* n_wins,
                                   int32_t n_clips, const float* lstm_w, float* hfin_ws, float* seq_opt,
                                   float* out, void* stream) {
    if (n_clips <= 0) return NISQA_ERR_ARG;
    NQ_LAUNCH_BEGIN();
    hipLaunchKernelGGL(lstm_dir_kernel, dim3(n_clips, 2), dim3(512), 0, (hipStream_t)stream, feat20, tok_off, n_wins,
                       lstm_w, hfin_ws, seq_opt);
    hipLaunchKernelGGL(pool_last_kernel, dim3(n_clips), dim3(64), 0, (hipStream_t)stream, (const float*)hfin_ws, lstm_w,
                       out);
    return NQ_LAUNCH_STATUS();
}
