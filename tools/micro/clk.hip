// Does the shader clock hold under sustained bf16 MFMA load, and does it depend on the DATA?  One kernel, 2 waves per SIMD
// of back-to-back v_mfma_f32_32x32x16_bf16 on register operands; operands are (0) zeros, (1) small constants, (2) random
// bf16 bit patterns.  Reports s_memtime (shader clock) against s_memrealtime (100 MHz) per wave -> effective MHz, and the
// wall time of the launch.
//   hipcc --offload-arch=gfx950 -O3 -o ab_libs/clk tools/micro/clk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void kern(const f32x4* __restrict__ src, float* out, unsigned long long* clk, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    f32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 65535]; b[i] = src[(tid * 8 + 4 + i) & 65535]; }
    f32x16 c[4];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) c[q][i] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, a[i]), __builtin_bit_cast(bfx8, b[(i + q) & 3]), c[q], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += c[q][i];
    out[tid] = s;
    if ((threadIdx.x & 63) == 0) { clk[(tid >> 6) * 2] = t1 - t0; clk[(tid >> 6) * 2 + 1] = r1 - r0; }
}

int main() {
    const int blocks = 512 * 8, iters = 20000;      // 512 resident workgroups; x8 waves of them
    f32x4* src; float* out; unsigned long long* clk;
    hipMalloc(&src, 65536 * 16); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 4 * 16);
    std::vector<unsigned> h(65536 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"zeros", "constant 1.0", "random bf16"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
        for (auto& v : h) {
            if (mode == 0) v = 0;
            else if (mode == 1) v = 0x3f803f80u;
            else { unsigned r = (unsigned)rand() ^ ((unsigned)rand() << 16); v = (r & 0x807f807fu) | 0x3f003f00u | ((r >> 3) & 0x00800080u); }
        }
        hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, src, out, clk, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(blocks * 4 * 2);
        hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
        double st = 0, rt = 0; for (int w = 0; w < blocks * 4; ++w) { st += c[2 * w]; rt += c[2 * w + 1]; }
        const double mhz = st / rt * 100.0;
        const double flops = (double)blocks * 4 * iters * 16 * 32768.0;
        printf("%-14s wall %8.3f ms  %7.1f TFLOP/s  cycles/MFMA/wave %6.2f  shader clock %7.1f MHz (memtime/memrealtime)\n", names[mode], ms, flops / ms * 1e-9,
               st / (blocks * 4) / (iters * 16.0), mhz);
    }
    return 0;
}
