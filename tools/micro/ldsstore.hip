// What does an LDS store cost a wave?  N stores per repetition from 64 lanes, four addressing patterns, timed with clock64,
// one wave per SIMD and nothing else running:
//   b16 / stride 2 bytes (two lanes share a dword: the bf16 planes of cnn_front_bf16_kernel, channel = lane)
//   b16 / stride 4 bytes (one lane per dword)      b32 / stride 4 bytes      b64 / stride 8 bytes
//   hipcc --offload-arch=gfx950 -O3 -o ab_libs/ldsstore tools/micro/ldsstore.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define AS3 __attribute__((address_space(3)))
template <int MODE>
__global__ __launch_bounds__(256, 1) void kern(float* out, long long* clk, int reps) {
    extern __shared__ char smem[];
    smem[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = 4096u + wave * 12288u;
    unsigned a;
    if (MODE == 0) a = base + (lane & 31) * 2 + (lane >> 5) * 1152;
    else if (MODE == 1) a = base + (lane & 31) * 4 + (lane >> 5) * 1152;
    else if (MODE == 2) a = base + lane * 4;
    else a = base + lane * 8;
    unsigned v = lane * 77u;
    const long long t0 = clock64();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            v = v * 3u + 1u;
            if (MODE <= 1) *(AS3 unsigned short*)(a + 72 * i) = (unsigned short)v;
            else if (MODE == 2) *(AS3 unsigned*)(a + 144 * i) = v;
            else { typedef unsigned u2 __attribute__((ext_vector_type(2))); *(AS3 u2*)(a + 144 * (i & 31)) = u2{v, v + 1}; }
        }
    }
    const long long t1 = clock64();
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = (float)smem[4096 + threadIdx.x] + v;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int MODE> static void run(const char* name, float* out, long long* clk) {
    const int blocks = 256, reps = 500;
    for (int k = 0; k < 2; ++k) { hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(256), 60000, 0, out, clk, reps); hipDeviceSynchronize(); }
    long long c[256]; hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += v;
    printf("%-44s %6.1f cycles per store instruction (4 waves per CU storing)\n", name, s / blocks / reps / 64);
}
int main() {
    float* out; long long* clk; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 8);
    run<0>("ds_write_b16, lanes 2 bytes apart", out, clk);
    run<1>("ds_write_b16, lanes 4 bytes apart", out, clk);
    run<2>("ds_write_b32, lanes 4 bytes apart", out, clk);
    run<3>("ds_write_b64, lanes 8 bytes apart", out, clk);
    return 0;
}
