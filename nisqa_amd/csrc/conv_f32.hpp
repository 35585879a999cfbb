// fp32 implicit-GEMM 3x3 convolution on v_mfma_f32_32x32x2_f32, shared by the AdaptCNN (cnn.hip) and
// StandardCNN (cnn_std.hip) kernels.
#pragma once
#include "common.hpp"

// 3x3 conv (padding 1) as implicit GEMM over one wave's MT x NT grid of 32x32 MFMA tiles.
//   smem : pixel-major activations, CIN floats per pixel, 16-B chunks XOR-swizzled
//   wf   : B fragments [tap][S][NT][64 lanes] float4
//   py/px/pbase/pvalid : per M-tile, the output pixel this lane's A-row stands for
template <int CIN, int MT, int NT, int H, int W, int ZERO_OFF>
NQ_DEV void conv3x3_mfma(f32x16 (&acc)[MT][NT], const char* smem, const f32x4* __restrict__ wf,
                         const int (&py)[MT], const int (&px)[MT], const int (&pbase)[MT],
                         const bool (&pvalid)[MT], int lane) {
    constexpr int S = CIN / 8;        // K-steps per tap (even: the double-buffer parity of step s is s & 1)
    constexpr int C = CIN / 4;        // 16-byte chunks per pixel
    static_assert(S % 2 == 0, "double buffer parity");
    const int h = lane >> 5;
    const f32x4* wl = wf + lane;
    // Software pipeline over the K-steps (tap-major): the operands of the NEXT step (B fragments from L2,
    // A rows from LDS) are requested BEFORE the MFMAs of the current step are issued, into the other half
    // of a register double buffer, so an L2 round trip (~600 clk) hides under a step of MFMAs (>= 1024 clk).
    f32x4 bf[2][NT], af[2][MT];
    int rowbyte[MT], swz[MT];
    auto tap_addr = [&](int tap) {
        const int dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int y = py[t] + dy, x = px[t] + dx;
            const bool ok = pvalid[t] && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const int pix = pbase[t] + y * W + x;
            rowbyte[t] = ok ? pix * (CIN * 4) : ZERO_OFF;
            swz[t] = ok ? (((pix * C) >> 4) & (C - 1)) : 0;
        }
    };
    tap_addr(0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[0][nt] = wl[nt * 64];
#pragma unroll
    for (int t = 0; t < MT; ++t) af[0][t] = *(const f32x4*)(smem + rowbyte[t] + ((h ^ swz[t]) << 4));
    for (int tap = 0; tap < 9; ++tap) {
        const f32x4* wt = wl + tap * (S * NT * 64);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            constexpr int dummy = 0; (void)dummy;
            const int nb = (s + 1) & 1;
            if (s + 1 < S) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[nb][nt] = wt[((s + 1) * NT + nt) * 64];
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    af[nb][t] = *(const f32x4*)(smem + rowbyte[t] + (((2 * (s + 1) + h) ^ swz[t]) << 4));
            } else if (tap < 8) {
                tap_addr(tap + 1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[nb][nt] = wt[(S * NT + nt) * 64];
#pragma unroll
                for (int t = 0; t < MT; ++t) af[nb][t] = *(const f32x4*)(smem + rowbyte[t] + ((h ^ swz[t]) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);     // keep the prefetch above ahead of this step's MFMAs
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[t][nt] = mfma32(af[s & 1][t][kk], bf[s & 1][nt][kk], acc[t][nt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

