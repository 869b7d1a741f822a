#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out) {
    __shared__ unsigned short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + 4 * threadIdx.x));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 512);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    return 0;
}
