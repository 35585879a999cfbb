// Issue rate of packed fp32 VALU instructions on gfx950: N independent accumulators updated in a loop by v_fma_f32,
// v_pk_fma_f32, v_pk_mul_f32 + v_pk_add_f32, with one and with two waves per SIMD (cycles per instruction and wave).
//   hipcc --offload-arch=gfx950 -O3 -o ab_libs/pkrate tools/micro/pkrate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(512) void kern(float* out, long long* clk, int iters) {
    const int lane = threadIdx.x;
    f2 a[8], k1 = {0.999f, 1.001f}, k2 = {0.001f, -0.001f};
    for (int i = 0; i < 8; ++i) a[i] = f2{0.1f * lane + i, 0.2f * lane - i};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(k1.x), "v"(k2.x)); }
                if (OP == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k1), "v"(k2)); }
                if (OP == 2) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k1)); }
                if (OP == 3) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k2)); }
            }
    }
    const long long t1 = clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
template <int OP> static void run(const char* name, float* out, long long* clk) {
    const int iters = 2000;
    for (int waves : {4, 8}) {                    // 4 waves per CU = one per SIMD, 8 = two per SIMD
        hipLaunchKernelGGL(kern<OP>, dim3(256), dim3(64 * waves), 0, 0, out, clk, iters);
        hipDeviceSynchronize();
        long long c[2048]; hipMemcpy(c, clk, 256 * waves * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256 * waves; ++i) s += c[i];
        printf("%-14s %d wave(s) per SIMD: %5.2f cycles per instruction and wave\n", name, waves / 4, s / (256 * waves) / (iters * 32.0));
    }
}
int main() {
    float* out; long long* clk; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 2048 * 8);
    run<0>("v_fma_f32", out, clk); run<1>("v_pk_fma_f32", out, clk); run<2>("v_pk_mul_f32", out, clk); run<3>("v_pk_add_f32", out, clk);
    return 0;
}
