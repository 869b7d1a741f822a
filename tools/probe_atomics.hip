// Probe: cost of per-tile f64 atomic adds into S slot rows of [C][2] accumulators (the candidate replacement of
// the BatchNorm partial-row + finalise launch). Every block does a few us of dummy ALU work, then adds 2*C values.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_atomics.hip -o build/exp/probe_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_atom(double* acc, int C, int S, int work, float* sink, int mode) {
    float v = threadIdx.x;
    for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.678f) sink[0] = v;
    double* row = acc + (size_t)(blockIdx.x % S) * 2 * C;
    if (mode == 1) {
        for (int c = threadIdx.x; c < 2 * C; c += blockDim.x)
            __hip_atomic_fetch_add(row + c, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (mode == 2) {
        float* rowf = (float*)acc + (size_t)(blockIdx.x % S) * 2 * C;
        for (int c = threadIdx.x; c < 2 * C; c += blockDim.x)
            __hip_atomic_fetch_add(rowf + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (mode == 3) {   // plain store of a partial row (today's scheme)
        float* rowf = (float*)acc + (size_t)blockIdx.x * 2 * C;
        for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) rowf[c] = v;
    }
}
int main() {
    const int blocks_list[] = {800, 3200, 12800};
    const int C_list[] = {48, 192};
    double* acc; float* sink;
    hipMalloc(&acc, 64 << 20); hipMalloc(&sink, 64); hipMemset(acc, 0, 64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nb : blocks_list) for (int C : C_list) {
        for (int mode = 0; mode <= 3; ++mode) for (int S : {1, 8, 64}) {
            if ((mode == 0 || mode == 3) && S != 1) continue;
            for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_atom, dim3(nb), dim3(256), 0, 0, acc, C, S, 2000, sink, mode);
            hipEventRecord(e0);
            for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k_atom, dim3(nb), dim3(256), 0, 0, acc, C, S, 2000, sink, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("blocks=%5d C=%3d mode=%d(%s) S=%2d : %.1f us/launch\n", nb, C, mode,
                   mode == 0 ? "none" : mode == 1 ? "f64 atomics" : mode == 2 ? "f32 atomics" : "row stores", S, ms * 1000 / 20);
        }
    }
    return 0;
}
