// Bidirectional single-layer LSTM (hidden 128, input 20) + last-step pooling for the nisqa_tts.tar
// architecture -- replaces LSTM.forward (pack / nn.LSTM / pad, reference nisqa/NISQA_lib.py:925-943) and
// PoolLastStepBi.forward (NL:1107-1115).
//
// The recurrence is sequential in time (2 987 steps for a 30 s clip), so what matters is the LATENCY of one step;
// parallelism comes from clips x directions x gate rows: one 512-thread workgroup per (clip, direction).  A step is a
// 512 x 128 matrix-vector product (64 v_pk_fma_f32 per thread, ~500 cycles of VALU issue on the CU's four SIMDs) and
// everything else is latency on the critical path, so the step is organised around ONE LDS round trip and ONE barrier:
//   * thread (u, q) = (hidden unit, quarter of the K dimension) holds the weights of ALL FOUR gates (i, f, g, o) of
//     unit u for h[32 q .. 32 q + 31] in 128 registers; the four quarters of a unit are the four lanes of a DPP quad, so
//     the gate pre-activations are completed with two quad_perm adds per gate -- the gates of a unit never travel
//     through LDS (round 1: one thread per gate ROW, gates exchanged through LDS + barrier, state re-laid out through a
//     second LDS hop: 0.97 us per step);
//   * every lane of the quad then holds i, f, g, o of its unit and updates (c, h) itself; lane q = 0 publishes h[u]
//     in a double-buffered LDS vector, one workgroup barrier, and every lane reads the 32 h values of its quarter
//     (8 ds_read_b128, four distinct addresses per wave: conflict-free) straight into the FMA operands;
//   * the input projection W_ih x_t is split the same way (5 of the 20 inputs per quarter); x_{t+1} is requested one
//     step ahead with vector loads (vmcnt, not the lgkmcnt the LDS waits use).
// Only what the pooling needs leaves the kernel: the forward direction's last state and the backward
// direction's state at position 0 (hfin[clip][2][128]); the full [n,256] sequence is written only when a
// caller asks for it (parity tests).
#include <stdlib.h>
#include "common.hpp"
#include "layout.hpp"
#include "../../include/nisqa_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

// 1 / (1 + e^-x) and tanh on v_exp_f32 / v_rcp_f32 (absolute error ~1e-7; the states are bounded by 1)
NQ_DEV float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// sum over the four lanes of a DPP quad, result in every lane
NQ_DEV float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}

// lane N of the quad, in every lane of the quad
template <int N>
NQ_DEV float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), N * 0x55, 0xF, 0xF, true));   // quad_perm [N,N,N,N]
}

// per-step phase clock (tools/lstm_clock.py; empty macros unless the unit is built with -DNQ_EXPERIMENTAL): phases 0..4, [7] = steps
NQ_CLK_EXPORT(g_lstm_clk, nisqa_debug_lstm_clock)

__global__ __launch_bounds__(512, 1) void lstm_dir_kernel(
    const float* __restrict__ feat20, const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
    const float* __restrict__ lw, float* __restrict__ hfin, float* __restrict__ seq) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][128];       // h_t, double-buffered over steps
    const int i = threadIdx.x, b = blockIdx.x, dir = blockIdx.y;
    const int lane = i & 63, wave = __builtin_amdgcn_readfirstlane(i >> 6);
    const int u = 16 * wave + (lane >> 2), q = lane & 3;               // hidden unit, K quarter
    const int n = n_wins[b], c0 = tok_off[b];
    const float* w = lw + (size_t)dir * LSTM_DIR_FLOATS;
    // weights of the four gate rows of unit u (PyTorch order i, f, g, o: rows g * 128 + u), columns 32 q .. 32 q + 31
    f32x2 whh[4][16], wih[4][3];
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = g * 128 + u;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const f32x4 v = *(const f32x4*)(w + LSTM_WHH + (size_t)row * 128 + 32 * q + 4 * kk);
            whh[g][2 * kk] = f32x2{v[0], v[1]};
            whh[g][2 * kk + 1] = f32x2{v[2], v[3]};
        }
        const float* wi = w + LSTM_WIH + row * 20 + 5 * q;             // inputs 5 q .. 5 q + 4 (the sixth slot is zero)
        wih[g][0] = f32x2{wi[0], wi[1]};
        wih[g][1] = f32x2{wi[2], wi[3]};
        wih[g][2] = f32x2{wi[4], 0.f};
        bias[g] = q == 0 ? w[LSTM_B + row] : 0.f;                      // added once per quad
    }
    const float gk = q == 2 ? 2.0f : 1.0f, gb = q == 2 ? -1.0f : 0.0f;
    float c = 0.f, h = 0.f;
    if (i < 256) ((float*)hbuf)[i] = 0.f;
    __syncthreads();

    // x_t: this lane's five inputs, requested one step ahead
    auto xload = [&](int t, float (&xv)[5]) {
        const int tok = c0 + (dir == 0 ? t : n - 1 - t);
        const float* x = feat20 + (size_t)tok * 20 + 5 * q;
#pragma unroll
        for (int j = 0; j < 5; ++j) xv[j] = x[j];
    };
    float xv[5];
    if (n > 0) xload(0, xv);
    NQ_SUM_BEGIN();
    for (int t = 0; t < n; ++t) {
        // h_{t-1}: the 32 values of this lane's quarter
        const f32x4* hp = (const f32x4*)(hbuf[t & 1] + 32 * q);
        f32x4 hv[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) hv[kk] = hp[kk];
        f32x2 a[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            a[g] = f32x2{bias[g], 0.f};
            a[g] = __builtin_elementwise_fma(wih[g][0], f32x2{xv[0], xv[1]}, a[g]);
            a[g] = __builtin_elementwise_fma(wih[g][1], f32x2{xv[2], xv[3]}, a[g]);
            a[g] = __builtin_elementwise_fma(wih[g][2], f32x2{xv[4], 0.f}, a[g]);
        }
        NQ_SUM(0);                                                   // h reads issued + input projection
        if (t + 1 < n) xload(t + 1, xv);                               // in flight across the product and the barrier
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                a[g] = __builtin_elementwise_fma(whh[g][2 * kk], f32x2{hv[kk][0], hv[kk][1]}, a[g]);
                a[g] = __builtin_elementwise_fma(whh[g][2 * kk + 1], f32x2{hv[kk][2], hv[kk][3]}, a[g]);
            }
        // lane q of the quad finishes gate q: the pre-activation sum over the quad, then the non-linearity -- sigmoid for
        // i, f, o and tanh(x) = 2 sigmoid(2 x) - 1 for g share one formula with per-lane constants (ONE v_exp / v_rcp per
        // lane instead of four); the four results come back through quad broadcasts
        NQ_SUM(1);                                                   // recurrent product
        const float p0 = quad_sum(a[0][0] + a[0][1]), p1 = quad_sum(a[1][0] + a[1][1]);
        const float p2 = quad_sum(a[2][0] + a[2][1]), p3 = quad_sum(a[3][0] + a[3][1]);
        const float pre = q == 0 ? p0 : q == 1 ? p1 : q == 2 ? p2 : p3;
        const float act = fmaf(gk, __builtin_amdgcn_rcpf(1.0f + __expf(-gk * pre)), gb);      // gk = 1 or 2, gb = 0 or -1
        const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act), gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
        NQ_SUM(2);                                                   // quad sums, gate non-linearity, broadcasts
        c = fmaf(fg, c, ig * gg);
        h = og * tanh_fast(c);
        if (q == 0) {
            hbuf[(t + 1) & 1][u] = h;
            if (seq) seq[(size_t)(c0 + (dir == 0 ? t : n - 1 - t)) * 256 + dir * 128 + u] = h;
        }
        NQ_SUM(3);                                                   // state update + publish
        __syncthreads();
        NQ_SUM(4);                                                   // barrier
    }
    NQ_SUM_COUNT(7, n);
    NQ_SUM_END(g_lstm_clk, blockIdx.y * gridDim.x + blockIdx.x, i == 0);
    if (q == 0) hfin[((size_t)b * 2 + dir) * 128 + u] = h;
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
    hipLaunchKernelGGL(lstm_dir_kernel, dim3(n_clips, 2), dim3(512), 0, (hipStream_t)stream, feat20, tok_off, n_wins, lstm_w,
                       hfin_ws, seq_opt);
    hipLaunchKernelGGL(pool_last_kernel, dim3(n_clips), dim3(64), 0, (hipStream_t)stream, (const float*)hfin_ws, lstm_w,
                       out);
    return NQ_LAUNCH_STATUS();
}
