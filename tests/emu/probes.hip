// Probes of the wave-level primitives the CPU lane-level executor MODELS BY HAND (tests/emu/include/hip/hip_runtime.h): each kernel
// runs one primitive on a known pattern and writes what every lane received. TEST INFRASTRUCTURE -- one source, two builds:
//   * tests/emu/build.py: for x86-64 against the executor (build/emu/libprobes_emu.so) -> tests/golden/emu_probes.npz, and the CPU
//     test that the executor still produces it;
//   * __graft_entry__.build(): hipcc --offload-arch=gfx950 -> tests/emu/libprobes_gfx950.so (in-tree: travels to the GPU box) ->
//     tests/test_gpu_emu_probes.py: the HARDWARE must produce the same table. That test is what turns "the executor models MFMA /
//     ds_read_b64_tr_b16 / DPP / buffer-resource / LDS-DMA semantics as the builder understands them" (circular)
//     into a statement a GPU run decides. The two MFMA probes are also checked against a plain numpy matrix product.
// Nothing under yolov5m_amd/ includes or links this file.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 0: v_mfma_f32_16x16x32_bf16. in = A[16][32] then B[16][32] as bf16 bit patterns (row-major, k contiguous); lane l supplies
// A[l & 15][8 (l >> 4) + j] and B[l & 15][8 (l >> 4) + j]; out[l * 4 + r] = accumulator register r (as f32 bits)
__global__ void probe_mfma_bf16(const uint16_t* __restrict__ in, uint32_t* __restrict__ out) {
    const int l = threadIdx.x;
    uint4 a = *reinterpret_cast<const uint4*>(in + (l & 15) * 32 + 8 * (l >> 4));
    uint4 b = *reinterpret_cast<const uint4*>(in + 512 + (l & 15) * 32 + 8 * (l >> 4));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = __float_as_uint(acc[r]);
}
// 1: v_mfma_f32_16x16x4_f32. in = A[16][4] then B[16][4] f32 bits; lane l supplies A[l & 15][l >> 4], B[l & 15][l >> 4]
__global__ void probe_mfma_f32(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    const int l = threadIdx.x;
    const float a = __uint_as_float(in[(l & 15) * 4 + (l >> 4)]);
    const float b = __uint_as_float(in[64 + (l & 15) * 4 + (l >> 4)]);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = __float_as_uint(acc[r]);
}
// 2: ds_read_b64_tr_b16. LDS word i holds i; lane l reads at its own 8-byte piece l; out[l * 4 + j] = received element j
__global__ void probe_tr16(const uint32_t* __restrict__, uint32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* lds = reinterpret_cast<unsigned short*>(smem);
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(lds + 4 * l));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
// 3: DPP, the controls the kernels use (+ row_shr:1 / row_shl:1), bound_ctrl on and off. out[c * 64 + l] for control slot c
#define PROBE_DPP(slot, ctrl, bc) out[(slot) * 64 + l] = (uint32_t)__builtin_amdgcn_update_dpp(0x5a5a0000 + l, 1000 + l, ctrl, 0xF, 0xF, bc);
__global__ void probe_dpp(const uint32_t* __restrict__, uint32_t* __restrict__ out) {
    const int l = threadIdx.x;
    PROBE_DPP(0, 0xB1, true) PROBE_DPP(1, 0x4E, true) PROBE_DPP(2, 0x141, true) PROBE_DPP(3, 0x140, true)
    PROBE_DPP(4, 0x111, true) PROBE_DPP(5, 0x101, true) PROBE_DPP(6, 0x111, false) PROBE_DPP(7, 0x101, false)
}
// 4: cross-lane: readlane, shuffles, ballot / any / all
__global__ void probe_lanes(const uint32_t* __restrict__, uint32_t* __restrict__ out) {
    const int l = threadIdx.x;
    const int v = 100 + 3 * l;
    out[0 * 64 + l] = (uint32_t)__builtin_amdgcn_readlane(v, 13);
    out[1 * 64 + l] = (uint32_t)__shfl(v, (l * 7 + 3) & 63, 64);
    out[2 * 64 + l] = (uint32_t)__shfl_xor(v, 5, 64);
    out[3 * 64 + l] = (uint32_t)__shfl_up(v, 3, 64);
    out[4 * 64 + l] = (uint32_t)__shfl_down(v, 3, 64);
    out[5 * 64 + l] = (uint32_t)__shfl(v, l + 1, 16);                     // width 16: the source wraps inside the 16-lane segment
    const unsigned long long m = __ballot(l % 3 == 0);
    out[6 * 64 + l] = (uint32_t)m;
    out[7 * 64 + l] = (uint32_t)(m >> 32);
    out[8 * 64 + l] = (uint32_t)(__any(l == 63) * 2 + __all(l < 64) + 4 * __all(l < 63));
    out[9 * 64 + l] = (uint32_t)__builtin_amdgcn_sbfe(0x00f0a5c3 >> (l & 7), l & 15, (l & 7) + 1);
}
// 5: buffer-resource loads: in = 64 words; offsets inside, straddling and behind num_records (= 256 bytes), and the "out of range" bit
__global__ void probe_buffer(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    const int l = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(in), 0, 256u, 0x00020000);
    unsigned voff = (unsigned)(l * 16);                                   // lanes 0-15 inside, 16+ behind num_records
    if (l == 20) voff = 248u;                                             // straddles the end: 8 bytes inside, 8 behind
    if (l == 21) voff = 0x80000000u;
    if (l == 22) voff = 0x80100000u;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, l == 23 ? 16u : 0u, 0);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
// 6: LDS-DMA (buffer_load_dwordx4 ... lds): destination = m0 base + lane * 16, out-of-range lanes write zeros; completion behind
// s_waitcnt vmcnt(0) + barrier. in = 256 words. The LDS is poisoned first; out = the 1 KiB the wave's DMA targets
__device__ __forceinline__ void probe_dma16(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__global__ void probe_lds_dma(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int l = threadIdx.x;
    for (int i = l; i < 512; i += 64) reinterpret_cast<uint32_t*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(in), 0, 1024u, 0x00020000);
    unsigned voff = (unsigned)((l ^ 5) * 16);                             // per-lane SOURCE, lane-linear destination
    if ((l & 7) == 7) voff = 0x80000000u;                                 // out of range: zeros land
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    probe_dma16(rs, voff, 0u, base + 1024u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 512; i += 64) out[i] = reinterpret_cast<uint32_t*>(smem)[i];
}

// which: 0..6 (above). Returns 0, or -1 for an unknown probe. in / out are device (GPU build) or host (executor build) pointers.
extern "C" int emu_probe(int which, const void* in, void* out, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const uint32_t* i32 = reinterpret_cast<const uint32_t*>(in);
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    switch (which) {
    case 0: hipLaunchKernelGGL(probe_mfma_bf16, dim3(1), dim3(64), 0, st, reinterpret_cast<const uint16_t*>(in), o); break;
    case 1: hipLaunchKernelGGL(probe_mfma_f32, dim3(1), dim3(64), 0, st, i32, o); break;
    case 2: hipLaunchKernelGGL(probe_tr16, dim3(1), dim3(64), 2048, st, i32, o); break;
    case 3: hipLaunchKernelGGL(probe_dpp, dim3(1), dim3(64), 0, st, i32, o); break;
    case 4: hipLaunchKernelGGL(probe_lanes, dim3(1), dim3(64), 0, st, i32, o); break;
    case 5: hipLaunchKernelGGL(probe_buffer, dim3(1), dim3(64), 0, st, i32, o); break;
    case 6: hipLaunchKernelGGL(probe_lds_dma, dim3(1), dim3(64), 2048, st, i32, o); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
