// Round 4: would an fp32-GRADE AdaptCNN on the bf16 matrix pipe pay?  Operands carried as THREE bf16 terms (hi + mid + lo = 24
// mantissa bits, the fp32 operand itself) and SIX products per term pair (hh, hm, mh, hl, lh, mm; what is dropped is of the
// size of an fp32 multiply-add's own rounding of the product): twice the MFMAs of the shipped two-term / three-product form, against the
// exact-fp32 kernels' 5.3 x.  Three planes per activation tensor do not leave room for two workgroups per CU, so the question is
// what ONE wave per SIMD reaches in the conv3 + conv4 K loops (1 296 MFMAs per segment) with real operand streams -- same
// set-up as klm4.hip: fragments through a buffer descriptor from a blob of the real size, A rows by ds_read_b128 from padded
// planes with lane-static tap masks, random bf16 data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -Inisqa_amd/csrc -Iinclude -o ab_libs/klx6 tools/micro/klx6.hip
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

// conv_k_bf16 generalised to T terms per operand; products (i, j) with i + j <= T - 1, smallest first
template <int T, int CIN, int MT, int NT, int W, int RS, int PLANE, unsigned ZA, int RING, int FENCE = 0>
NQ_DEV void conv_k_local(f32x16 (&acc)[MT][NT], __amdgpu_buffer_rsrc_t rsrc, int wbyte, unsigned lane16,
                         const unsigned (&base)[MT], const unsigned (&m9)[MT]) {
    constexpr int S16 = CIN / 16, TOTAL = 9 * S16;
    f32x4 b[RING][NT][T], a[2][MT][T];
    unsigned a_ad[MT][T];
    auto load_b = [&](int g, int slot) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int t = 0; t < T; ++t) b[slot][nt][t] = wfrag_load(rsrc, lane16, wbyte + ((g * NT + nt) * T + t) * 1024);
    };
    auto load_a = [&](int g, int slot) {
        const int tap = g / S16, s = g - tap * S16;
        const int tapoff = ((tap / 3) * W + tap % 3) * RS;
        if (s == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool ok = (m9[m] >> tap) & 1u;
#pragma unroll
                for (int t = 0; t < T; ++t) a_ad[m][t] = ok ? base[m] + t * PLANE : ZA - tapoff;
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < T; ++t) a[slot][m][t] = lds_ld128_a(a_ad[m][t] + tapoff + 32 * s);
    };
#pragma unroll
    for (int g = 0; g < RING - 1; ++g) load_b(g, g);
    load_a(0, 0);
#pragma unroll
    for (int g = 0; g < TOTAL; ++g) {
        if (g + RING - 1 < TOTAL) load_b(g + RING - 1, (g + RING - 1) % RING);
        if (FENCE & 1) __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < TOTAL) load_a(g + 1, (g + 1) & 1);
        if (FENCE & 2) __builtin_amdgcn_sched_barrier(0);
        const int sa = g & 1, sb = g % RING;
#pragma unroll
        for (int order = 2 * (T - 1); order >= 0; --order)            // smallest products first
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const int j = order - i;
                if (j < 0 || j >= T || i + j > T - 1) continue;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma_bf(a[sa][m][i], b[sb][nt][j], acc[m][nt]);
            }
    }
}

template <int T, int RING, int WAVES, int FENCE = 0>
__global__ __launch_bounds__(WAVES * 64, 1) void kern(const unsigned short* __restrict__ wb, const unsigned* __restrict__ rnd,
                                                      float* __restrict__ out, long long* __restrict__ clk, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = 2;
    constexpr unsigned SEGB = T * PL4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned total = SEG_BASE + WAVES * SEGB;
    for (unsigned i = threadIdx.x; i < total / 4; i += WAVES * 64) ((unsigned*)smem)[i] = i * 4 < SEG_BASE ? 0u : rnd[i & 65535];
    __syncthreads();
    const unsigned lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, (18 + 36) * 2 * T * 1024, 0x00020000);
    unsigned base3[MT], base4[MT], m9[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int r = t * 32 + (lane & 31);
        const bool valid = r < H_ * W_;
        const int y = r / W_, x = r - y * W_;
        m9[t] = tap_mask(valid, y, x, H_, W_);
        const unsigned sb = SEG_BASE + (unsigned)wave * SEGB;
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
        conv_k_local<T, 32, MT, 2, W_, RS3, PL3, ZADDR, RING, FENCE>(acc, rsrc, 0, lane16, base3, m9);
        conv_k_local<T, 64, MT, 2, W_, RS4, PL4, ZADDR, RING, FENCE>(acc, rsrc, 18 * 2 * T * 1024, lane16, base4, m9);
#pragma unroll
        for (int t = 0; t < MT; ++t) sink += acc[t][0][rep & 15] + acc[t][1][(rep + 3) & 15];
    }
    const long long t1 = clock64();
    const long long r1 = wall_clock64();
    if (lane == 0) {
        clk[(blockIdx.x * WAVES + wave) * 2] = t1 - t0;
        clk[(blockIdx.x * WAVES + wave) * 2 + 1] = r1 - r0;
    }
    out[(blockIdx.x * WAVES + wave) * 64 + lane] = sink;
}

// ---- Round 5: the stop-rule measurement for a wave PAIR per segment (task 1b).  Same operand streams and planes as `kern`
//      above, now with the layer EPILOGUES a real kernel has behind its K loops (bias + ReLU + three-term split + 16-bit plane stores of
//      every accumulator value, conv3-shaped: 60 pixels x 64 channels per segment and layer):
//        kern_epi<PAIR = false>: four waves per workgroup, one wave per segment (NT = 2: all 64 output channels) -- today's structure;
//        kern_epi<PAIR = true>:  eight waves per workgroup on the SAME four segments' planes, two waves per segment, each with NT = 1
//                                (its 32 output channels: half the MFMAs, half the epilogue, all of the A reads): the second wave of a
//                                SIMD runs under the other's epilogues.
//      Reported per SEGMENT: if the pair does not save >= 12 % of the microseconds, the restructure is not worth its rewrite.
template <int NT>
NQ_DEV void epilogue_like_conv3(const f32x16 (&acc)[2][NT], unsigned wr, int plane, float tn) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int u = 16 * t + r;
                if (u < 30) lds_store_terms2<3>(wr + u * RS4 + 64 * nt, wr + (u + 1) * RS4 + 64 * nt, plane, fmaxf(acc[t][nt][r] + tn, 0.f), fmaxf(acc[t][nt][r + 1] + tn, 0.f));
            }
}
template <bool PAIR>
__global__ __launch_bounds__(PAIR ? 512 : 256, 1) void kern_epi(const unsigned short* __restrict__ wb, const unsigned* __restrict__ rnd,
                                                                float* __restrict__ out, long long* __restrict__ clk, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = 3, MT = 2, NT = PAIR ? 1 : 2, WAVES = PAIR ? 8 : 4;
    constexpr unsigned SEGB = T * PL4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = PAIR ? wave >> 1 : wave, half = PAIR ? wave & 1 : 0;
    const unsigned total = SEG_BASE + 4 * SEGB;
    for (unsigned i = threadIdx.x; i < total / 4; i += WAVES * 64) ((unsigned*)smem)[i] = i * 4 < SEG_BASE ? 0u : rnd[i & 65535];
    __syncthreads();
    const unsigned lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, (18 + 36) * 2 * T * 1024, 0x00020000);
    unsigned base3[MT], base4[MT], m9[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int r = t * 32 + (lane & 31);
        const bool valid = r < H_ * W_;
        const int y = r / W_, x = r - y * W_;
        m9[t] = tap_mask(valid, y, x, H_, W_);
        const unsigned sb = SEG_BASE + (unsigned)seg * SEGB;
        base3[t] = valid ? sb + (unsigned)(((y - 1) * W_ + (x - 1)) * RS3) + 16u * (lane >> 5) : sb + 16u * (lane >> 5);
        base4[t] = valid ? sb + (unsigned)(((y - 1) * W_ + (x - 1)) * RS4) + 16u * (lane >> 5) : sb + 16u * (lane >> 5);
    }
    const unsigned wr = SEG_BASE + (unsigned)seg * SEGB + (30 * (lane >> 5)) * RS4 + (lane & 31) * 2 + 64 * half;
    const float tn = __uint_as_float(rnd[lane]) * 1e-3f;
    float sink = 0.f;
    const long long r0 = wall_clock64();
    const long long t0 = clock64();
    for (int rep = 0; rep < reps; ++rep) {
        f32x16 acc[MT][NT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = zero16();
#pragma unroll
        for (int t = 0; t < MT; ++t) asm volatile("" : "+v"(base3[t]), "+v"(base4[t]), "+v"(m9[t]));
        // conv3-shaped K loop (18 steps) + its epilogue, then conv4-shaped (36 steps) + its epilogue; a pair wave streams its half of the
        // fragments ([step][NT = 2][T] in the blob: every second (step, tile) block)
        conv_k_local<T, 32, MT, NT, W_, RS3, PL3, ZADDR, 3, 2>(acc, rsrc, half * T * 1024, lane16, base3, m9);
        epilogue_like_conv3<NT>(acc, wr, PL4, tn);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = zero16();
        conv_k_local<T, 64, MT, NT, W_, RS4, PL4, ZADDR, 3, 2>(acc, rsrc, 18 * 2 * T * 1024 + half * T * 1024, lane16, base4, m9);
        epilogue_like_conv3<NT>(acc, wr, PL4, tn);
#pragma unroll
        for (int t = 0; t < MT; ++t) sink += acc[t][0][rep & 15];
        if (PAIR) __builtin_amdgcn_s_barrier();              // the pair meets once per layer pair (a real kernel: before the next layer reads)
    }
    const long long t1 = clock64();
    const long long r1 = wall_clock64();
    if (lane == 0) {
        clk[(blockIdx.x * WAVES + wave) * 2] = t1 - t0;
        clk[(blockIdx.x * WAVES + wave) * 2 + 1] = r1 - r0;
    }
    out[(blockIdx.x * WAVES + wave) * 64 + lane] = sink;
}
template <bool PAIR>
static void run_epi(const char* name, const unsigned short* wb, const unsigned* rnd, float* out, long long* clk, int reps) {
    constexpr int WAVES = PAIR ? 8 : 4;
    const unsigned lds = SEG_BASE + 4 * 3 * PL4;
    hipFuncSetAttribute((const void*)kern_epi<PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kern_epi<PAIR>), dim3(blocks), dim3(WAVES * 64), lds, 0, wb, rnd, out, clk, reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<long long> h(blocks * WAVES * 2);
    hipMemcpy(h.data(), clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (size_t i = 0; i < h.size(); i += 2) { cyc += (double)h[i]; rt += (double)h[i + 1]; }
    const double n_w = (double)(blocks * WAVES);
    const double mhz = cyc / rt * 100.0;
    const double us_seg = ms * 1e3 / reps;                    // every workgroup does 4 segments per repetition on its CU: launch time / reps = time per round of 4 segments per CU
    printf("%-58s launch %8.1f us  clock %5.0f MHz  %7.2f us per round of four segments per CU (a wave's cycles per round %7.0f)\n", name, ms * 1e3, mhz, us_seg,
           cyc / n_w / reps);
}

template <int T, int RING, int WAVES, int FENCE = 0>
static void run(const char* name, const unsigned short* wb, const unsigned* rnd, float* out, long long* clk, int reps) {
    const unsigned lds = SEG_BASE + WAVES * T * PL4;
    hipFuncSetAttribute((const void*)kern<T, RING, WAVES, FENCE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kern<T, RING, WAVES, FENCE>), dim3(blocks), dim3(WAVES * 64), lds, 0, wb, rnd, out, clk, reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<long long> h(blocks * WAVES * 2);
    hipMemcpy(h.data(), clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (size_t i = 0; i < h.size(); i += 2) { cyc += (double)h[i]; rt += (double)h[i + 1]; }
    const double n_w = (double)(blocks * WAVES);
    const double cyc_seg = cyc / n_w / reps;                             // a wave's cycles per segment
    const double mhz = cyc / rt * 100.0;
    const int prods = T == 2 ? 3 : 6;
    const double mfma_cycles = 216.0 * prods * 32.0;                     // conv3 + conv4: 216 (A, B) tile pairs per segment
    const double per_simd = cyc_seg / (WAVES / 4.0);                     // SIMD cycles per segment
    printf("%-44s LDS %6u B  launch %8.1f us  clock %5.0f MHz  SIMD cycles per segment %7.0f (%6.2f us)  MFMA-only %6.0f -> duty %.2f\n",
           name, lds, ms * 1e3, mhz, per_simd, per_simd / mhz, mfma_cycles, mfma_cycles / per_simd);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    unsigned short* wb;
    unsigned* rnd;
    float* out;
    long long* clk;
    const size_t wbytes = (18 + 36) * 2 * 3 * 1024;
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
    run<2, 3, 8>("2 terms / 3 products, 2 waves per SIMD (today)", wb, rnd, out, clk, reps);
    run<2, 3, 4>("2 terms / 3 products, 1 wave per SIMD", wb, rnd, out, clk, reps);
    run<3, 3, 4>("3 terms / 6 products, 1 wave per SIMD, ring 3", wb, rnd, out, clk, reps);
    run<3, 2, 4>("3 terms / 6 products, 1 wave per SIMD, ring 2", wb, rnd, out, clk, reps);
    run<3, 4, 4>("3 terms / 6 products, 1 wave per SIMD, ring 4", wb, rnd, out, clk, reps);
    run<3, 3, 4, 2>("3 terms, ring 3, fence behind the step's requests", wb, rnd, out, clk, reps);
    run<3, 3, 4, 3>("3 terms, ring 3, fences between B / A requests / MFMAs", wb, rnd, out, clk, reps);
    run<3, 4, 4, 2>("3 terms, ring 4, fence behind the step's requests", wb, rnd, out, clk, reps);
    run_epi<false>("K loops + epilogues, ONE wave per segment (4 waves per CU)", wb, rnd, out, clk, reps);
    run_epi<true>("K loops + epilogues, a wave PAIR per segment (8 waves per CU)", wb, rnd, out, clk, reps);
    run_epi<false>("K loops + epilogues, ONE wave per segment (again)", wb, rnd, out, clk, reps);
    run_epi<true>("K loops + epilogues, a wave PAIR per segment (again)", wb, rnd, out, clk, reps);
    return 0;
}
