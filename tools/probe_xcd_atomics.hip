// Probe: 256 workgroups x 512 threads each add a C x C f32 tile (the weight-gradient tile of bwd_pw_kernel) into
//   mode 1: ONE buffer with agent-scope atomics (what atomicAdd does),
//   mode 2: the buffer of the workgroup's OWN XCD (s_getreg HW_REG_XCC_ID) with workgroup-scope atomics: no sc1 bit, the
//           atomic executes in that XCD's L2, which every CU of the XCD shares -- 8 buffers, summed afterwards,
//   mode 3: as 2 with agent scope (isolates the effect of the scope from that of 8x less contention),
//   mode 0: nothing (the dummy work alone).
// Checks the sums (every element must equal the number of workgroups) and the XCD ids seen, and times the launch.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_xcd_atomics.hip -o build/exp/probe_xcd_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
__global__ __launch_bounds__(512) void k(float* buf, unsigned* seen, int n, int mode, int work, float* sink) {
    float v = threadIdx.x;
    for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.678f) sink[0] = v;
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) atomicAdd(seen + x, 1u);
    float* dst = buf + (mode >= 2 ? (size_t)x * n : 0);
    const int rot = (blockIdx.x * 997) % n;            // rotated start: the workgroups do not walk the addresses in lock step
    if (mode == 1 || mode == 3) {
        for (int c = threadIdx.x; c < n; c += blockDim.x) { int i = c + rot; i -= i >= n ? n : 0;
            __hip_atomic_fetch_add(dst + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    } else if (mode == 2) {
        for (int c = threadIdx.x; c < n; c += blockDim.x) { int i = c + rot; i -= i >= n ? n : 0;
            __hip_atomic_fetch_add(dst + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
}
int main() {
    float *buf, *sink; unsigned* seen;
    const int NMAX = 192 * 192;
    hipMalloc(&buf, 8 * NMAX * 4); hipMalloc(&sink, 64); hipMalloc(&seen, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> h(8 * NMAX);
    for (int C : {96, 192}) for (int mode = 0; mode <= 3; ++mode) {
        const int n = C * C;
        float best = 1e9;
        bool ok = true;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(buf, 0, 8 * NMAX * 4); hipMemset(seen, 0, 64);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, buf, seen, n, mode, 2000, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            hipMemcpy(h.data(), buf, 8 * NMAX * 4, hipMemcpyDeviceToHost);
            if (mode) for (int i = 0; i < n; ++i) {
                float s = 0; for (int x = 0; x < (mode >= 2 ? 8 : 1); ++x) s += h[(size_t)x * n + i];
                if (s != 256.0f) { ok = false; if (rep == 0 && i < 3) printf("   element %d: %f\n", i, s); }
            }
        }
        unsigned hs[16]; hipMemcpy(hs, seen, 64, hipMemcpyDeviceToHost);
        printf("C=%3d mode=%d: %.1f us  sums %s   workgroups per XCD id:", C, mode, best * 1000, mode ? (ok ? "OK" : "WRONG") : "-");
        for (int x = 0; x < 10; ++x) printf(" %u", hs[x]);
        printf("\n");
    }
    return 0;
}
