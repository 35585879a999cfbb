// Semantics of ds_read_b64_tr_b16 on gfx950, read off the hardware: LDS holds lds[i] = i (16-bit), every lane passes its own
// byte address (lane * 8 here: lane l points at elements 4l .. 4l+3) and prints the four elements it received.
//   hipcc --offload-arch=gfx950 -O3 -o ab_libs/trread tools/micro/trread.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int stride_elems) {
    extern __shared__ short lds[];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + stride_elems * l));
    *(s4*)(out + 4 * l) = v;
}
int main() {
    short* d; hipMalloc(&d, 64 * 8);
    short h[256];
    for (int stride : {4, 16, 40}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, stride);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("lane address = element %d * lane:\n", stride);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}
