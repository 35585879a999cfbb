// Round 4, the stop-ruled structural question for cnn_front_bf16_kernel: would conv3 + conv4
// K loops with M = 4 tiles -- TWO segments per wave, every weight fragment used by both, one 512-register wave per SIMD --
// beat today's M = 2 tiles (one segment per wave, two waves per SIMD) by >= 15 % in cycles?  K loops ONLY, real operand streams:
//   * B (weight) fragments through a buffer descriptor from a blob of the real size (conv3 18 steps x 2 N tiles x 2 KB = 72 KB,
//     conv4 36 x 2 x 2 KB = 144 KB per pass; every wave streams its own copy out of L2, as in the kernel),
//   * A fragments by ds_read_b128 from zero-bordered-by-mask bf16 hi / lo planes of 12 x 5 pixels (rows padded by 16 bytes),
//     lane-static 9-bit tap masks, the kernel's own conv_k_bf16 (conv_bf16.hpp), random bf16 data in both operands (the
//     shader clock under bf16 MFMA load depends on the data: DESIGN.md 4.5),
//   * a CU holds EIGHT segments at a time in both configurations (8 waves x 1, or 4 waves x 2): the same LDS footprint.
// Modes: 0 = M 2, two waves per SIMD (today); 1 = M 4, one wave per SIMD, A rows read when needed; 2 = M 4, A rows one
// step ahead; 3 = M 4 with a 5-slot fragment ring.  Output per mode: cycles per segment (shader clock, mean over waves), wall
// microseconds per launch, shader clock MHz, and the matrix-pipe duty (648 MFMAs x 32 cycles x 2 segments per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -Inisqa_amd/csrc -Iinclude -o ab_libs/klm4 tools/micro/klm4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.hpp"
#include "conv_bf16.hpp"

#define W_ 5
#define H_ 12
#define RS3 (2 * 32 + 16)
#define RS4 (2 * 64 + 16)
#define PL3 (H_ * W_ * RS3)
#define PL4 (H_ * W_ * RS4)
#define ZADDR 2048u
#define SEG_BASE 2304u
#define SEG_BYTES (2 * PL4)                 /* a segment's region holds the larger (conv4) planes; conv3's alias its start */

template <int MT, bool APF, int RING, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void kern(const unsigned short* __restrict__ wb, const unsigned* __restrict__ rnd,
                                                      float* __restrict__ out, long long* __restrict__ clk, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int SEGS = MT / 2;                                      // segments per wave
    const unsigned total = SEG_BASE + WAVES * SEGS * SEG_BYTES;
    for (unsigned i = threadIdx.x; i < total / 4; i += WAVES * 64) ((unsigned*)smem)[i] = i * 4 < SEG_BASE ? 0u : rnd[i & 65535];
    __syncthreads();
    const unsigned lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, (18 + 36) * 2 * 2048, 0x00020000);
    unsigned base3[MT], base4[MT], m9[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int sg = t / 2, r = (t & 1) * 32 + (lane & 31);
        const bool valid = r < H_ * W_;
        const int y = r / W_, x = r - y * W_;
        m9[t] = tap_mask(valid, y, x, H_, W_);
        const unsigned sb = SEG_BASE + (unsigned)(wave * SEGS + sg) * SEG_BYTES;
        base3[t] = valid ? sb + (unsigned)(((y - 1) * W_ + (x - 1)) * RS3) + 16u * (lane >> 5) : sb + 16u * (lane >> 5);
        base4[t] = valid ? sb + (unsigned)(((y - 1) * W_ + (x - 1)) * RS4) + 16u * (lane >> 5) : sb + 16u * (lane >> 5);
    }
    float sink = 0.f;
    const long long r0 = wall_clock64();
    const long long t0 = clock64();
    for (int rep = 0; rep < reps; ++rep) {
        f32x16 acc[MT][2];
#pragma unroll
        for (int t = 0; t < MT; ++t) { acc[t][0] = zero16(); acc[t][1] = zero16(); }
#pragma unroll
        for (int t = 0; t < MT; ++t) asm volatile("" : "+v"(base3[t]), "+v"(base4[t]), "+v"(m9[t]));
        conv_k_bf16<32, MT, 2, W_, RS3, PL3, ZADDR, APF, RING>(acc, rsrc, 0, lane16, base3, m9);
        conv_k_bf16<64, MT, 2, W_, RS4, PL4, ZADDR, APF, RING>(acc, rsrc, 18 * 2 * 2048, lane16, base4, m9);
#pragma unroll
        for (int t = 0; t < MT; ++t) sink += acc[t][0][rep & 15] + acc[t][1][(rep + 3) & 15];
    }
    const long long t1 = clock64();
    const long long r1 = wall_clock64();
    if (lane == 0) {
        clk[(blockIdx.x * WAVES + wave) * 2] = t1 - t0;                  // shader clocks (s_memtime)
        clk[(blockIdx.x * WAVES + wave) * 2 + 1] = r1 - r0;              // 100 MHz ticks (s_memrealtime)
    }
    out[(blockIdx.x * WAVES + wave) * 64 + lane] = sink;
}

template <int MT, bool APF, int RING, int WAVES>
static void run(const char* name, const unsigned short* wb, const unsigned* rnd, float* out, long long* clk, int reps) {
    constexpr int SEGS = MT / 2;
    const unsigned lds = SEG_BASE + WAVES * SEGS * SEG_BYTES;
    hipFuncSetAttribute((const void*)kern<MT, APF, RING, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int it = 0; it < 3; ++it) {                                    // the last launch counts (clock ramp)
        hipEventRecord(e0);
        hipLaunchKernelGGL((kern<MT, APF, RING, WAVES>), dim3(blocks), dim3(WAVES * 64), lds, 0, wb, rnd, out, clk, reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> h(blocks * WAVES * 2);
    hipMemcpy(h.data(), clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (size_t i = 0; i < h.size(); i += 2) { cyc += (double)h[i]; rt += (double)h[i + 1]; }
    const double n_w = (double)(blocks * WAVES);
    const double cyc_rep = cyc / n_w / reps;                            // a wave's cycles per rep = per 2 segments of its SIMD (either configuration)
    const double mhz = cyc / rt * 100.0;
    const double mfma_cycles = 648.0 * 32.0 * 2.0;                      // per SIMD and rep: 648 MFMAs per segment x 32 cycles (8 passes x 4)
    printf("%-36s LDS %6u B  launch %8.1f us  shader clock %5.0f MHz  cycles per 2 segments of a SIMD %7.0f  (%6.2f us)  MFMA-only %6.0f -> pipe duty %.2f\n",
           name, lds, ms * 1e3, mhz, cyc_rep, cyc_rep / mhz, mfma_cycles, mfma_cycles / cyc_rep);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    unsigned short* wb;
    unsigned* rnd;
    float* out;
    long long* clk;
    const size_t wbytes = (18 + 36) * 2 * 2048;
    hipMalloc(&wb, wbytes);
    hipMalloc(&rnd, 65536 * 4);
    hipMalloc(&out, 256 * 8 * 64 * 4);
    hipMalloc(&clk, 256 * 8 * 8 * 2);
    std::vector<unsigned short> hw(wbytes / 2);
    std::vector<unsigned> hr(65536);
    unsigned s = 12345u;
    auto rb = [&]() { s = s * 1664525u + 1013904223u; const unsigned v = s >> 16; return (unsigned short)((v & 0x807f) | 0x3f00 | ((v >> 3) & 0x0080)); };
    for (auto& v : hw) v = rb();
    for (auto& v : hr) v = (unsigned)rb() | ((unsigned)rb() << 16);
    hipMemcpy(wb, hw.data(), wbytes, hipMemcpyHostToDevice);
    hipMemcpy(rnd, hr.data(), 65536 * 4, hipMemcpyHostToDevice);
    run<2, true, 3, 8>("M=2, 2 waves/SIMD (today)", wb, rnd, out, clk, reps);
    run<4, false, 3, 4>("M=4, 1 wave/SIMD, A on demand", wb, rnd, out, clk, reps);
    run<4, true, 3, 4>("M=4, 1 wave/SIMD, A one step ahead", wb, rnd, out, clk, reps);
    run<4, true, 5, 4>("M=4, 1 wave/SIMD, A ahead, ring 5", wb, rnd, out, clk, reps);
    run<2, true, 3, 4>("M=2, 1 wave/SIMD (reference point)", wb, rnd, out, clk, reps);
    return 0;
}
