// Can ONE wave hide the VALU + LDS-store work of an epilogue inside the MFMA gaps of an independent K loop (two segments
// software-pipelined in a wave, one wave per SIMD on the 512-register budget)?  Stream 1 = a conv3-shaped K loop (18 steps of
// 12 x v_mfma_f32_32x32x16_bf16 = 216 MFMAs, A fragments from LDS one step ahead, B fragments from L2 through a 3-deep ring);
// stream 2 = a conv3-shaped epilogue of ANOTHER accumulator set (64 values per lane: bias, ReLU, bf16 hi/lo split, 128
// ds_write_b16).  Modes: 0 K loop only, 1 epilogue only, 2 K loop then epilogue (source order), 3 four epilogue values
// placed by hand behind the first MFMA group of every K step (the next step's A reads are issued BEFORE them).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -Inisqa_amd/csrc -Iinclude -o ab_libs/interleave tools/micro/interleave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "common.hpp"
#include "conv_bf16.hpp"

#define RS 80
#define PLANE 9600

template <int MODE>
__global__ __launch_bounds__(256, 1) void kern(const unsigned short* __restrict__ wb, const float* __restrict__ init, float* __restrict__ out,
                                               long long* __restrict__ clk, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 60000 / 4; i += 256) ((float*)smem)[i] = 0.001f * ((i * 7) & 1023);
    __syncthreads();
    const unsigned lane16 = lane * 16;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, 1 << 20, 0x00020000);
    unsigned base[2];
    base[0] = 256u + ((lane & 31) % 25) * RS + ((lane >> 5) << 4);            // two DIFFERENT row maps: nothing to merge
    base[1] = 256u + (25 + (lane & 31) % 23) * RS + ((lane >> 5) << 4);
    f32x16 accB[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) accB[t][n] = *(const f32x16*)(init + ((t * 2 + n) * 64 + lane) * 16);
    const unsigned wr = 24000u + wave * 8000u + (lane & 31) * 2 + (lane >> 5) * 1152;
    float sink = 0.f;
    const long long t0 = clock64();
    for (int rep = 0; rep < reps; ++rep) {
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[t][n] = zero16();
        auto epi = [&](int idx) {                       // idx 0..63 -> (t, n, r)
            const int r = idx & 15, t = (idx >> 4) & 1, n = idx >> 5;
            lds_store_split(wr + (16 * t + r) * 72 + 64 * n, 3600, fmaxf(accB[t][n][r] + 0.25f, 0.f));
        };
        constexpr int TOTAL = 18;
        const int wrep = __builtin_amdgcn_readfirstlane((rep & 7) * 73728);     // nothing is loop-invariant across repetitions
        const unsigned arep = (rep & 3) * 16;
        f32x4 bh[3][2], bl[3][2], ah[2][2], al[2][2];
        auto load_b = [&](int g, int slot) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                bh[slot][nt] = wfrag_load(rsrc, lane16, wrep + ((g * 2 + nt) * 2 + 0) * 1024);
                bl[slot][nt] = wfrag_load(rsrc, lane16, wrep + ((g * 2 + nt) * 2 + 1) * 1024);
            }
        };
        auto load_a = [&](int g, int slot) {
            const int tap = g / 2, s = g - tap * 2;
            const int tapoff = ((tap / 3) * 5 + tap % 3) * RS;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[slot][t] = lds_ld128(base[t] + arep + tapoff + 32 * s);
                al[slot][t] = lds_ld128(base[t] + arep + PLANE + tapoff + 32 * s);
            }
        };
        if (MODE != 1) {
            load_b(0, 0); load_b(1, 1); load_a(0, 0);
#pragma unroll
            for (int g = 0; g < TOTAL; ++g) {
                if (g + 2 < TOTAL) load_b(g + 2, (g + 2) % 3);
                if (g + 1 < TOTAL) load_a(g + 1, (g + 1) & 1);
                const int sa = g & 1, sb = g % 3;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[t][nt] = mfma_bf(ah[sa][t], bl[sb][nt], acc[t][nt]);
                if (MODE == 3) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (4 * g + q < 64) epi(4 * g + q);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[t][nt] = mfma_bf(al[sa][t], bh[sb][nt], acc[t][nt]);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[t][nt] = mfma_bf(ah[sa][t], bh[sb][nt], acc[t][nt]);
            }
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int idx = 0; idx < 64; ++idx) epi(idx);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) sink += acc[t][n][0] + acc[t][n][7] + acc[t][n][15];
        accB[0][0][0] += sink * 1e-12f;
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = sink + ((float*)smem)[6000 + threadIdx.x] + accB[1][1][3];
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, const unsigned short* wb, const float* init, float* out, long long* clk) {
    const int blocks = 256, reps = 200;
    for (int k = 0; k < 2; ++k) { hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(256), 62000, 0, wb, init, out, clk, reps); hipDeviceSynchronize(); }
    std::vector<long long> c(blocks);
    hipMemcpy(c.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += v;
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)kern<MODE>);
    printf("%-48s %8.1f cycles per repetition (%d VGPRs, %zu B scratch)\n", name, s / blocks / reps, fa.numRegs, (size_t)fa.localSizeBytes);
}

int main() {
    unsigned short* wb; float* out; float* init; long long* clk;
    hipMalloc(&wb, 1 << 20); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 8); hipMalloc(&init, 4 * 64 * 16 * 4);
    std::vector<unsigned short> h(1 << 19);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0x1ff);
    hipMemcpy(wb, h.data(), 1 << 20, hipMemcpyHostToDevice);
    std::vector<float> hi(4 * 64 * 16);
    for (auto& v : hi) v = (rand() % 2000 - 1000) * 0.01f;
    hipMemcpy(init, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    run<0>("K loop only (216 MFMAs)", wb, init, out, clk);
    run<1>("epilogue only (64 values, 128 ds_write_b16)", wb, init, out, clk);
    run<2>("K loop, then epilogue (source order)", wb, init, out, clk);
    run<3>("epilogue spread over the K steps by hand", wb, init, out, clk);
    return 0;
}
