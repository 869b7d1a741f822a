// Internal helpers shared by the HIP translation units of liby5m.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/y5m.h"

extern "C" void y5m_set_error(const char* fmt, ...);

#define Y5M_CHECK_LAUNCH(name)                                                       \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            y5m_set_error("%s: %s", name, hipGetErrorString(e__));                   \
            return Y5M_ELAUNCH;                                                      \
        }                                                                            \
    } while (0)

#define Y5M_REQUIRE(cond, msg)                                                       \
    do {                                                                             \
        if (!(cond)) {                                                               \
            y5m_set_error("%s:%d: %s (%s)", __FILE__, __LINE__, msg, #cond);         \
            return Y5M_EINVAL;                                                       \
        }                                                                            \
    } while (0)

// "which kernel would this call launch?" (y5m_conv_kernel_name / y5m_wgrad_kernel_name): the dispatch runs as usual, the
// launcher writes the instantiation it chose here instead of launching it
extern thread_local int y5m_name_only;
extern thread_local char y5m_name_buf[192];
#define Y5M_NAME_ONLY(RET, ...)                                                      \
    do {                                                                             \
        if (y5m_name_only) {                                                         \
            snprintf(y5m_name_buf, sizeof(y5m_name_buf), __VA_ARGS__);               \
            return RET;                                                              \
        }                                                                            \
    } while (0)

// Workgroups a PERSISTENT launch (one workgroup per CU: conv_halo_kernel, conv_gemm8_kernel) may use: the device's CU count,
// capped by Y5M_PERSIST_CUS. The cap exists for data-parallel runs: a collective's kernels hold some CUs for as long as
// a bucket is on the wire, and a persistent workgroup whose CU is taken only starts when that kernel ends -- with a static
// tile assignment the whole launch then waits for it. Read once per process.
static inline int y5m_persistent_cus() {
    static int cus = 0;
    if (cus <= 0) {
        int dev = 0, n = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
        const char* e = getenv("Y5M_PERSIST_CUS");
        const int cap = e ? atoi(e) : 0;
        cus = (cap > 0 && cap < n) ? cap : n;
    }
    return cus;
}

// Round-4 kernel forms. Five hot-path kernels were rewritten in round 4 from a static audit of their generated code
// (tools/isa_audit.py) while no GPU was reachable; their round-3 forms passed the GPU suite on hardware. Both forms are compiled
// (a template parameter per kernel); Y5M_R4_KERNELS is the bit mask of the round-4 forms to use. The default is 0 -- only code
// that has run on a gfx950 is on the default path -- until tools/ab_r4_kernels.sh has decided each bit on hardware.
enum {
    Y5M_R4_WGRAD_ROWS = 1,      // wgrad_rows_kernel: run state on the scalar unit
    Y5M_R4_BN_ACT = 2,          // bn_act_kernel: residual rows loaded raw
    Y5M_R4_BWD_STEM = 4,        // bwd_stem_kernel: straight-line streaming loop
    Y5M_R4_BWD_PW = 8,          // bwd_pw_kernel: unconditional re-requests
    Y5M_R4_BN_BWD_REDUCE = 16,  // bn_bwd_reduce_kernel: raw loads behind one scheduling barrier
};
static inline int y5m_r4_forms() {
    static int mask = -1;
    if (mask < 0) { const char* e = getenv("Y5M_R4_KERNELS"); mask = e ? (atoi(e) & 31) : 0; }
    return mask;
}

int y5m_fill32(void* p, uint32_t v, size_t n_words, hipStream_t st);      // y5m_core.hip: 32-bit fill as a kernel launch

static inline hipStream_t y5m_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t y5m_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- bf16 <-> f32 (round to nearest even), raw 16-bit storage -------------------------------
typedef uint16_t bf16_t;
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// gfx950 has a hardware packed convert (v_cvt_pk_bf16_f32, round-to-nearest-even): one instruction per
// TWO values instead of ~6 integer ops per value -- the epilogues of every bf16 kernel convert a lot.
typedef __bf16 y5m_bf16x2 __attribute__((ext_vector_type(2)));
typedef float y5m_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
    const y5m_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, y5m_bf16x2));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
