// Measurement probe, not part of the predict path: what does THIS GPU sustain on dense bf16 MFMA work, and at which shader
// clock?  Two waves per SIMD issue back-to-back v_mfma_f32_32x32x16_bf16 on register operands; nothing else competes.
// With all-zero (or constant) operands an MI355X holds ~2.38 GHz and the 2.5 PFLOP/s of the data sheet; with random bf16
// operands the power management drops the clock to ~1.8 GHz (1.86 PFLOP/s): the ceiling for any bf16 MFMA kernel working on
// real data, and the reason timing experiments that replace operands by constants overstate what they remove
// (tools/micro/clk.hip is the standalone version).
#include "common.hpp"
#include "../../include/nisqa_hip.h"

typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void mfma_sustained_kernel(const f32x4* __restrict__ src, float* __restrict__ out,
                                                                unsigned long long* __restrict__ clk, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    f32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 65535]; b[i] = src[(tid * 8 + 4 + i) & 65535]; }
    f32x16 c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) c[q][i] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(probe_bf16x8, a[i]),
                                                               __builtin_bit_cast(probe_bf16x8, b[(i + q) & 3]), c[q], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += c[q][i];
    out[tid] = s;
    if ((threadIdx.x & 63) == 0) { clk[(tid >> 6) * 2] = t1 - t0; clk[(tid >> 6) * 2 + 1] = r1 - r0; }
}

// operands [dev] 65536 x 16 bytes (any bf16 bit patterns), out [dev] blocks * 256 floats, clk [dev] blocks * 4 pairs of
// (shader clock ticks, 100 MHz ticks) per wave.  16 MFMAs (16 * 32768 flop) per wave and iteration.
extern "C" int nisqa_probe_mfma_sustained(const void* operands, float* out, uint64_t* clk, int32_t blocks, int32_t iters,
                                          void* stream) {
    if (!operands || !out || !clk || blocks <= 0 || iters <= 0) return NISQA_ERR_ARG;
    hipLaunchKernelGGL(mfma_sustained_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)operands, out,
                       (unsigned long long*)clk, iters);
    return hipGetLastError() == hipSuccess ? NISQA_OK : NISQA_ERR_LAUNCH;
}
