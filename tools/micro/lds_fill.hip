// Probe kernel for tools/exp_split.py: workgroups that fill their dynamic LDS with a NaN pattern for a while and exit
// (does a co-resident / subsequent kernel depend on LDS contents it did not write?).  Built into ab_libs/ldsfill.so.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void lds_fill_kernel(unsigned pattern, int bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < bytes / 4; i += 256) lds[i] = pattern + (it & 1);
        __syncthreads();
        for (int i = threadIdx.x; i < bytes / 4; i += 256) acc += lds[i];
        __syncthreads();
    }
    if (acc == 12345u) sink[0] = acc;
}
// the same with the access widths of the conv kernels: 16-bit stores, 128-bit loads, addresses spread over the allocation
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void lds_fill16_kernel(unsigned pattern, int bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    unsigned short* h = (unsigned short*)lds;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < bytes / 2; i += 256) h[i] = (unsigned short)(pattern + it);
        __syncthreads();
        for (int i = threadIdx.x; i < bytes / 16; i += 256) { const f32x4 v = ((const f32x4*)lds)[i]; acc += v[0] + v[3]; }
        __syncthreads();
    }
    if (acc == 12345.f) sink[0] = 1;
}
extern "C" int lds_fill16(unsigned pattern, int bytes, int iters, int blocks, unsigned* sink, void* stream) {
    hipLaunchKernelGGL(lds_fill16_kernel, dim3(blocks), dim3(256), bytes, (hipStream_t)stream, pattern, bytes, iters, sink);
    return (int)hipGetLastError();
}
extern "C" int lds_fill(unsigned pattern, int bytes, int iters, int blocks, unsigned* sink, void* stream) {
    hipLaunchKernelGGL(lds_fill_kernel, dim3(blocks), dim3(256), bytes, (hipStream_t)stream, pattern, bytes, iters, sink);
    return (int)hipGetLastError();
}
