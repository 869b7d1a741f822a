// Probe: LDS-DMA (buffer_load_dwordx4 ... lds) semantics on gfx950.
//   (1) destination = M0 base + lane*16 (wave-uniform base), (2) out-of-range lanes write zeros,
//   (3) counted vmcnt + barrier ordering. Prints PASS/FAIL lines. Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}

__global__ __launch_bounds__(256) void probe(const unsigned* src, unsigned nbytes, unsigned* out, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // poison
    for (int i = tid; i < 8192 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00020000);
    const unsigned base = (unsigned)(uintptr_t)(smem) ;   // LDS byte address of the dynamic region
    // each wave fills 1 KiB at wid*1024 (+4096 for a second instruction); source permuted: lane l reads chunk (l ^ 5)
    unsigned voff = (unsigned)((wid * 64 + (lane ^ 5)) * 16);
    if (mode == 1 && (lane & 3) == 3) voff = 0x80000000u;          // OOB lanes
    if (mode == 2 && (lane & 3) == 3) voff = nbytes + 64;          // just past num_records
    dma16(rs, voff, __builtin_amdgcn_readfirstlane(base + wid * 1024));
    dma16(rs, voff, __builtin_amdgcn_readfirstlane(base + 4096 + wid * 1024));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = tid; i < 8192 / 4; i += 256) out[i] = reinterpret_cast<unsigned*>(smem)[i];
}

int main() {
    const int N = 4096;   // bytes
    std::vector<unsigned> h(N / 4);
    for (int i = 0; i < N / 4; ++i) h[i] = 0x1000000u + i;
    unsigned *d, *o;
    hipMalloc(&d, N); hipMalloc(&o, 8192);
    hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    std::vector<unsigned> r(2048);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(o, 0, 8192);
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 8192, 0, d, (unsigned)N, o, mode);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("mode %d: launch error %s\n", mode, hipGetErrorString(e)); return 1; }
        hipMemcpy(r.data(), o, 8192, hipMemcpyDeviceToHost);
        int bad = 0, zeros = 0, poison = 0;
        for (int rep = 0; rep < 2; ++rep)
            for (int w = 0; w < 4; ++w)
                for (int l = 0; l < 64; ++l)
                    for (int k = 0; k < 4; ++k) {
                        const unsigned got = r[(rep * 4096 + w * 1024 + l * 16) / 4 + k];
                        const bool oob = mode != 0 && (l & 3) == 3;
                        const unsigned want = 0x1000000u + ((w * 64 + (l ^ 5)) * 16) / 4 + k;
                        if (oob) { if (got == 0) ++zeros; else if (got == 0xdeadbeefu) ++poison; else ++bad; }
                        else if (got != want) { if (bad < 4) printf("  mode %d w%d l%d k%d got %08x want %08x\n", mode, w, l, k, got, want); ++bad; }
                    }
        printf("mode %d: %s  bad=%d oob_zero=%d oob_untouched=%d\n", mode, bad == 0 ? "PASS" : "FAIL", bad, zeros, poison);
    }
    return 0;
}
