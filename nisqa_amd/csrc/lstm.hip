// Bidirectional single-layer LSTM (hidden 128, input 20) + last-step pooling for the nisqa_tts.tar
// architecture -- replaces LSTM.forward (pack / nn.LSTM / pad, reference nisqa/NISQA_lib.py:925-943) and
// PoolLastStepBi.forward (NL:1107-1115).
//
// The recurrence is sequential in time (up to 5,986 steps), so what matters is the LATENCY of one step, and
// parallelism comes from clips x directions x gate rows: one 512-thread workgroup per (clip, direction); thread
// i owns gate row i (PyTorch order i,f,g,o) and keeps its W_hh row (128 floats) and W_ih row (20 floats) in
// REGISTERS for the whole clip.  A step is then a 512 x 128 matrix-vector product = 1024 VALU cycles on one
// CU; everything else is arranged so that nothing but those FMAs sits on the critical path:
//   * h never travels through LDS for the product (eight waves reading 512 B each per step is ~2000 LDS
//     cycles, twice the FMA time): every wave keeps the WHOLE h in 8 registers, lane l holding h[16k + (l & 15)]
//     in register k, and feeds it to the FMAs through the DPP row_share broadcast (v_fmac_f32_dpp, no extra
//     instruction);
//   * the gate non-linearity is applied by the row's own thread before the exchange; after the single
//     workgroup barrier of the step every wave redundantly updates (c, h) for all 128 units (two per lane), so
//     h needs no second barrier -- it is re-laid out through a wave-private LDS strip;
//   * x_{t+1} is requested (vector loads, vmcnt) before the product of step t and W_ih x_{t+1} + b is formed right
//     after it, so no memory latency is left on the step's critical path.
// Only what the pooling needs leaves the kernel: the forward direction's last state and the backward
// direction's state at position 0 (hfin[clip][2][128]); the full [n,256] sequence is written only when a
// caller asks for it (parity tests).
#include <stdlib.h>
#include "common.hpp"
#include "layout.hpp"
#include "../../include/nisqa_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

// 1 / (1 + e^-x) and tanh on v_exp_f32 / v_rcp_f32 (absolute error ~1e-7; the states are bounded by 1)
NQ_DEV float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
NQ_DEV float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// value of lane (row base + N) for every lane of the 16-lane row; folds into the consuming v_fmac as a DPP operand
template <int N>
NQ_DEV float row_share(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + N, 0xf, 0xf, true));
}
template <int N>
NQ_DEV void fma16(f32x2 (&a)[2], const float* w, float hk) {          // two FMAs per v_pk_fma_f32
    const f32x2 hv = {row_share<N>(hk), row_share<N + 1>(hk)}, wv = {w[N], w[N + 1]};
    a[(N >> 1) & 1] = __builtin_elementwise_fma(hv, wv, a[(N >> 1) & 1]);
    if constexpr (N < 14) fma16<N + 2>(a, w, hk);
}

template <int KD>
__global__ __launch_bounds__(512, 1) void lstm_dir_kernel(
    const float* __restrict__ feat20, const int32_t* __restrict__ tok_off, const int32_t* __restrict__ n_wins,
    const float* __restrict__ lw, float* __restrict__ hfin, float* __restrict__ seq) {
    __shared__ __attribute__((aligned(16))) float gact[2][512];       // activated gates, double-buffered over steps
    __shared__ __attribute__((aligned(16))) float hstrip[8][128];     // wave-private h re-layout
    const int i = threadIdx.x, b = blockIdx.x, dir = blockIdx.y;
    const int lane = i & 63, wave = __builtin_amdgcn_readfirstlane(i >> 6);
    const int n = n_wins[b], c0 = tok_off[b];
    const float* w = lw + (size_t)dir * LSTM_DIR_FLOATS;
    float whh[128];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const f32x4 v = *(const f32x4*)(w + LSTM_WHH + (size_t)i * 128 + 4 * q);
        whh[4 * q] = v[0]; whh[4 * q + 1] = v[1]; whh[4 * q + 2] = v[2]; whh[4 * q + 3] = v[3];
    }
    f32x2 wih[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) wih[j] = f32x2{w[LSTM_WIH + i * 20 + 2 * j], w[LSTM_WIH + i * 20 + 2 * j + 1]};
    const float bias = w[LSTM_B + i];
    const bool is_g = (wave >> 1) == 2;                                // rows 256..383: the cell candidate (tanh)
    float hk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) hk[k] = 0.f;
    float c2[2] = {0.f, 0.f}, h2[2] = {0.f, 0.f};
    float* hs = hstrip[wave];
    hs[lane] = 0.f;
    hs[lane + 64] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // x_t by VECTOR loads (every lane the same address): they are counted by vmcnt, so a request issued one
    // step ahead stays in flight across the LDS waits of the step (scalar loads share lgkmcnt with LDS and would
    // be waited for at the first of them)
    const int vzero = __builtin_amdgcn_mbcnt_lo(0u, 0u);               // 0, opaque to the compiler: keeps the loads VMEM
    auto xload = [&](int t, f32x4 (&xv)[5]) {
        const int tok = c0 + (dir == 0 ? t : n - 1 - t);
        const f32x4* x = (const f32x4*)(feat20 + (size_t)tok * 20) + vzero;
#pragma unroll
        for (int q = 0; q < 5; ++q) xv[q] = x[q];
    };
    auto xproj = [&](const f32x4 (&xv)[5]) {
        f32x2 a = {bias, 0.f};
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            a = __builtin_elementwise_fma(wih[2 * q], f32x2{xv[q][0], xv[q][1]}, a);
            a = __builtin_elementwise_fma(wih[2 * q + 1], f32x2{xv[q][2], xv[q][3]}, a);
        }
        return a[0] + a[1];
    };
    f32x4 xv[5];
    float xin = 0.f;
    if (n > 0) { xload(0, xv); xin = xproj(xv); }
    for (int t = 0; t < n; ++t) {
        if (t + 1 < n) xload(t + 1, xv);                               // consumed after the matrix-vector product
        f32x2 a[2] = {{xin, 0.f}, {0.f, 0.f}};
        // units 0 .. 16 KD - 1 through the DPP broadcast, the rest as LDS broadcast reads: the VALU (one v_mov_dpp per
        // unit) and the LDS pipe (8 waves x 16 B per 4 units) share the cost of distributing h
#pragma unroll
        for (int k = 0; k < KD; ++k) fma16<0>(a, whh + 16 * k, hk[k]);
        {
            const f32x4* hp = (const f32x4*)(hs + 16 * KD);
#pragma unroll
            for (int q = 0; q < 32 - 4 * KD; ++q) {
                const f32x4 hv = hp[q];
                const float* wq = whh + 16 * KD + 4 * q;
                a[0] = __builtin_elementwise_fma(f32x2{hv[0], hv[1]}, f32x2{wq[0], wq[1]}, a[0]);
                a[1] = __builtin_elementwise_fma(f32x2{hv[2], hv[3]}, f32x2{wq[2], wq[3]}, a[1]);
            }
        }
        const float pre = (a[0][0] + a[0][1]) + (a[1][0] + a[1][1]);
        gact[t & 1][i] = is_g ? tanh_fast(pre) : sigmoid_fast(pre);
        if (t + 1 < n) xin = xproj(xv);
        __syncthreads();
        // every wave: units 2*lane, 2*lane + 1
        const float* g = gact[t & 1];
        const f32x2 ig = *(const f32x2*)(g + 2 * lane), fg = *(const f32x2*)(g + 128 + 2 * lane);
        const f32x2 gg = *(const f32x2*)(g + 256 + 2 * lane), og = *(const f32x2*)(g + 384 + 2 * lane);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            c2[e] = fmaf(fg[e], c2[e], ig[e] * gg[e]);
            h2[e] = og[e] * tanh_fast(c2[e]);
        }
        *(f32x2*)(hs + 2 * lane) = f32x2{h2[0], h2[1]};
        if (seq && wave == 0) {
            const int tok = c0 + (dir == 0 ? t : n - 1 - t);
            *(f32x2*)(seq + (size_t)tok * 256 + dir * 128 + 2 * lane) = f32x2{h2[0], h2[1]};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < KD; ++k) hk[k] = hs[16 * k + (lane & 15)];
        __builtin_amdgcn_wave_barrier();
    }
    if (wave == 0) *(f32x2*)(hfin + ((size_t)b * 2 + dir) * 128 + 2 * lane) = f32x2{h2[0], h2[1]};
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
    static const int kd = getenv("NISQA_LSTM_KD") ? atoi(getenv("NISQA_LSTM_KD")) : 3;
#define NQ_LSTM(K) hipLaunchKernelGGL(lstm_dir_kernel<K>, dim3(n_clips, 2), dim3(512), 0, (hipStream_t)stream, feat20, \
                                      tok_off, n_wins, lstm_w, hfin_ws, seq_opt)
    switch (kd) {
        case 0: NQ_LSTM(0); break;
        case 2: NQ_LSTM(2); break;
        case 4: NQ_LSTM(4); break;
        case 8: NQ_LSTM(8); break;
        default: NQ_LSTM(3); break;
    }
#undef NQ_LSTM
    hipLaunchKernelGGL(pool_last_kernel, dim3(n_clips), dim3(64), 0, (hipStream_t)stream, (const float*)hfin_ws, lstm_w,
                       out);
    return NQ_LAUNCH_STATUS();
}
