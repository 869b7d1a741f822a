// Stand-in for <hip/hip_runtime.h> when the kernel sources are compiled for the CPU lane-level executor (emu_rt.h).
// TEST INFRASTRUCTURE: only tests/emu/build.py puts this directory on an include path.
// It provides exactly what yolov5m_amd/csrc uses: the execution-space keywords, dim3 / thread indices, vector types, the
// handful of HIP runtime calls, device math / bit helpers, atomics, wave64 cross-lane operations and the gfx950 builtins
// (semantics as documented in /opt/skills/guides and verified on hardware by the tools/probe_*.hip programs of rounds 1-3).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <tuple>

#include "emu_rt.h"

#define Y5M_EMU 1

// ---- keywords -------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define HIP_SYMBOL(x) x

typedef emu::Dim3 dim3;
#define threadIdx (emu::g.cur->tid)
#define blockIdx (emu::g.bid)
#define blockDim (emu::g.bdim)
#define gridDim (emu::g.gdim)

// ---- vector types (members are lvalues as in HIP; implicit conversion from / to the native clang vectors) ---------
template <typename T> struct alignas(2 * sizeof(T)) emu_vec2 {
    typedef T native __attribute__((ext_vector_type(2)));
    T x, y;
    emu_vec2() = default;
    constexpr emu_vec2(T a, T b) : x(a), y(b) {}
    emu_vec2(native v) : x(v[0]), y(v[1]) {}
    operator native() const { return native{x, y}; }
};
template <typename T> struct alignas(4 * sizeof(T)) emu_vec4 {
    typedef T native __attribute__((ext_vector_type(4)));
    T x, y, z, w;
    emu_vec4() = default;
    constexpr emu_vec4(T a, T b, T c, T d) : x(a), y(b), z(c), w(d) {}
    emu_vec4(native v) : x(v[0]), y(v[1]), z(v[2]), w(v[3]) {}
    operator native() const { return native{x, y, z, w}; }
};
typedef emu_vec2<float> float2;
typedef emu_vec4<float> float4;
typedef emu_vec2<int> int2;
typedef emu_vec4<int> int4;
typedef emu_vec2<unsigned> uint2;
typedef emu_vec4<unsigned> uint4;
typedef emu_vec4<unsigned short> ushort4;
typedef emu_vec2<double> double2;
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

// ---- runtime calls ---------------------------------------------------------------------------------------------------
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t {
    int multiProcessorCount;
    size_t totalGlobalMem;
    char name[64];
    char gcnArchName[64];
};
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    const char* e = getenv("Y5M_EMU_CUS");           // small by default: persistent kernels launch one workgroup per "CU"
    p->multiProcessorCount = e ? atoi(e) : 16;
    p->totalGlobalMem = (size_t)64 << 30;
    strcpy(p->name, "emu-gfx950");
    strcpy(p->gcnArchName, "gfx950-emu");
    return hipSuccess;
}
template <typename F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
template <typename T> static inline hipError_t hipMemcpyFromSymbol(void* dst, const T& sym, size_t n) {
    memcpy(dst, &sym, n);
    return hipSuccess;
}
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) \
    emu::launch((kern), dim3(grid), dim3(block), (size_t)(lds), (void*)(stream), ##__VA_ARGS__)

// ---- scalar helpers --------------------------------------------------------------------------------------------------
static inline float __uint_as_float(unsigned u) { return __builtin_bit_cast(float, u); }
static inline float __int_as_float(int u) { return __builtin_bit_cast(float, u); }
static inline unsigned __float_as_uint(float f) { return __builtin_bit_cast(unsigned, f); }
static inline int __float_as_int(float f) { return __builtin_bit_cast(int, f); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline int __mul24(int a, int b) { return ((a << 8) >> 8) * ((b << 8) >> 8); }
// __expf on the GPU is v_exp_f32(x * log2(e)) with the product rounded to f32 (argument error ~ |x| 6e-8 relative in the result);
// -DY5M_EMU_FAST_EXP models exactly that, the default is libm's expf (tools: attribution of the bf16 first-step loss shift)
#ifdef Y5M_EMU_FAST_EXP
#define __expf(x) exp2f((float)(x) * 1.44269504088896341f)
#else
#define __expf(x) expf(x)
#endif
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
template <typename T> static inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T max(T a, T b) { return a > b ? a : b; }
static inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }

// ---- atomics (workgroups of a launch run on several OS threads) ---------------------------------------------------------
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_SYSTEM 3
template <typename T> static inline T emu_atomic_add(T* p, T v) {
    if constexpr (__is_integral(T)) {
        return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
    } else {
        using U = std::conditional_t<sizeof(T) == 4, uint32_t, uint64_t>;
        U* q = reinterpret_cast<U*>(p);
        U old = __atomic_load_n(q, __ATOMIC_RELAXED), neu;
        do {
            const T s = __builtin_bit_cast(T, old) + v;
            neu = __builtin_bit_cast(U, s);
        } while (!__atomic_compare_exchange_n(q, &old, neu, true, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED));
        return __builtin_bit_cast(T, old);
    }
}
#define __hip_atomic_fetch_add(p, v, order, scope) emu_atomic_add((p), (std::remove_reference_t<decltype(*(p))>)(v))
template <typename T> static inline T emu_atomic_load(const T* p) {
    T v;
    __atomic_load(const_cast<T*>(p), &v, __ATOMIC_SEQ_CST);
    return v;
}
template <typename T, typename V> static inline void emu_atomic_store(T* p, V v) {
    T t = (T)v;
    __atomic_store(p, &t, __ATOMIC_SEQ_CST);
}
#define __hip_atomic_load(p, order, scope) emu_atomic_load((p))
#define __hip_atomic_store(p, v, order, scope) emu_atomic_store((p), (v))
static inline float atomicAdd(float* p, float v) { return emu_atomic_add(p, v); }
static inline double atomicAdd(double* p, double v) { return emu_atomic_add(p, v); }
static inline int atomicAdd(int* p, int v) { return emu_atomic_add(p, v); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return emu_atomic_add(p, v); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return emu_atomic_add(p, v); }
static inline float unsafeAtomicAdd(float* p, float v) { return emu_atomic_add(p, v); }
static inline double unsafeAtomicAdd(double* p, double v) { return emu_atomic_add(p, v); }
static inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicMin(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicAnd(unsigned* p, unsigned v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicExch(int* p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicCAS(int* p, int cmp, int v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return cmp; }
static inline unsigned atomicCAS(unsigned* p, unsigned cmp, unsigned v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return cmp; }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- workgroup / wave synchronisation ------------------------------------------------------------------------------------
static inline void __syncthreads() { emu::barrier(); }
static inline void __builtin_amdgcn_s_barrier() { emu::barrier(); }
static inline void __builtin_amdgcn_wave_barrier() { emu::wave_sync(); }
#define __builtin_amdgcn_fence(order, scope) emu::wave_sync()
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __builtin_amdgcn_s_nop(n) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
// stand-ins of `s_waitcnt vmcnt(n)` / `s_waitcnt lgkmcnt(n)` (tests/emu/translate.py keeps the operand): the LDS-DMA loads of this
// wave beyond the n newest land (emu_rt.h: dma_wait), then a wave-level rendezvous (lanes run one after the other)
static inline void emu_waitcnt_vm(int n) { emu::dma_wait(n); emu::wave_sync(); }
static inline void emu_waitcnt_lgkm(int) { emu::wave_sync(); }
static inline unsigned long long emu_memtime() { return 0; }
#define __builtin_readcyclecounter() 0ull

// ---- wave64 cross-lane operations ------------------------------------------------------------------------------------------
template <typename T> static inline T emu_shfl_idx(T v, int src) {
    auto c = emu::coll_begin();
    c.mine<T>() = v;
    emu::coll_sync(2);
    src &= 63;
    return c.live(src) ? c.of<T>(src) : v;
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    const int lane = emu::g.cur->lane;
    return emu_shfl_idx(v, (lane & ~(width - 1)) | (src & (width - 1)));
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    return emu_shfl_idx(v, emu::g.cur->lane ^ mask);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int lane = emu::g.cur->lane, base = lane & ~(width - 1), src = lane - (int)d;
    return emu_shfl_idx(v, src < base ? lane : src);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int lane = emu::g.cur->lane, base = lane & ~(width - 1), src = lane + (int)d;
    return emu_shfl_idx(v, src >= base + width ? lane : src);
}
static inline unsigned long long __ballot(int pred) {
    auto c = emu::coll_begin();
    c.mine<int>() = pred ? 1 : 0;
    emu::coll_sync(3);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (c.live(l) && c.of<int>(l)) m |= 1ull << l;
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) {
    auto c = emu::coll_begin();
    c.mine<int>() = pred ? 1 : 0;
    emu::coll_sync(4);
    for (int l = 0; l < 64; ++l)
        if (c.live(l) && !c.of<int>(l)) return 0;
    return 1;
}
// readfirstlane: its uses make a wave-uniform value scalar -> the calling lane's own value (no rendezvous, so it may sit in
// divergent code exactly as on the GPU). The GPU would broadcast the FIRST ACTIVE lane's value: a kernel that passes a value
// that is not uniform computes something else there than here, so every call is logged and the scheduler compares the lanes
// (emu_rt.h: note_uniform / emu_rt.cpp: check_uniform; aborts with the source line).
template <typename T> static inline T emu_readfirstlane(T v, const char* file, int line) {
    emu::note_uniform(file, line, &v, sizeof(T));
    return v;
}
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane((v), __FILE__, __LINE__)
// readlane: the lane index is a scalar operand on the GPU -- it must be the same on every lane (checked)
static inline int __builtin_amdgcn_readlane(int v, int l) {
    struct Op { int v, l; };
    auto c = emu::coll_begin();
    c.mine<Op>() = Op{v, l};
    emu::coll_sync(9);
    for (int i = 0; i < 64; ++i)
        if (c.live(i) && c.of<Op>(i).l != l) {
            fprintf(stderr, "emu: readlane with a lane index that is not wave-uniform (lane %d: %d, lane %d: %d)\n", c.lane, l, i, c.of<Op>(i).l);
            abort();
        }
    l &= 63;
    return c.live(l) ? c.of<Op>(l).v : v;
}
static inline int __builtin_amdgcn_sbfe(int v, int off, int width) {
    off &= 31; width &= 31;            // (s_bfe_i32: sign-extended bit field)
    if (width == 0) return 0;
    return (int)((unsigned)v << (32 - off - width)) >> (32 - width);
}
// v_rcp_f32 is accurate to 1 ulp, its rounding is not specified: the default is the correctly rounded quotient;
// -DY5M_EMU_RCP_ULP=n moves every result by n ulps (a worst-case systematic bias, for sensitivity experiments)
#ifndef Y5M_EMU_RCP_ULP
#define Y5M_EMU_RCP_ULP 0
#endif
static inline float __builtin_amdgcn_rcpf(float x) {
    float r = 1.0f / x;
    if (Y5M_EMU_RCP_ULP != 0 && r == r && r != 0.0f && fabsf(r) < 3.0e38f)
        r = __builtin_bit_cast(float, __builtin_bit_cast(int, r) + (r > 0 ? Y5M_EMU_RCP_ULP : -(Y5M_EMU_RCP_ULP)));
    return r;
}
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline const void* __builtin_amdgcn_kernarg_segment_ptr() { return emu::g.kernarg; }

// DPP: the four controls the kernels use, row_mask = bank_mask = 0xF, bound_ctrl: quad_perm [1,0,3,2] (0xB1),
// quad_perm [2,3,0,1] (0x4E), row_half_mirror (0x141), row_mirror (0x140)
static inline int __builtin_amdgcn_update_dpp(int old, int v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int lane = emu::g.cur->lane;
    int src;
    if (ctrl >= 0 && ctrl <= 0xFF) src = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    else if (ctrl == 0x141) src = (lane & ~7) | (7 - (lane & 7));
    else if (ctrl == 0x140) src = (lane & ~15) | (15 - (lane & 15));
    else if (ctrl >= 0x111 && ctrl <= 0x11F) {                   // row_shr:n
        const int n = ctrl & 15, s = (lane & 15) - n;
        src = s < 0 ? -1 : (lane & ~15) | s;
    } else if (ctrl >= 0x101 && ctrl <= 0x10F) {                 // row_shl:n
        const int n = ctrl & 15, s = (lane & 15) + n;
        src = s > 15 ? -1 : (lane & ~15) | s;
    } else { fprintf(stderr, "emu: DPP control 0x%x not modelled\n", ctrl); abort(); }
    auto c = emu::coll_begin();
    c.mine<int>() = v;
    emu::coll_sync(5);
    if (src < 0 || !c.live(src)) return bound_ctrl ? 0 : old;
    return c.of<int>(src);
}

// ---- buffer resources: (base, stride, num_records, flags); raw loads return 0 beyond num_records ---------------------
struct emu_rsrc { const unsigned char* base; unsigned num; };
typedef emu_rsrc __amdgpu_buffer_rsrc_t;
typedef unsigned emu_u32x4 __attribute__((ext_vector_type(4)));
template <typename P> static inline emu_rsrc __builtin_amdgcn_make_buffer_rsrc(P* base, short stride, int num, int flags) {
    return emu_rsrc{reinterpret_cast<const unsigned char*>(base), (unsigned)num};
}
static inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(emu_rsrc r, unsigned voff, unsigned soff, int aux) {
    const unsigned long long off = (unsigned long long)voff + soff;
    emu_u32x4 v = {0, 0, 0, 0};
    if (off + 16 <= r.num) memcpy(&v, r.base + off, 16);
    return v;
}
// LDS-DMA (`buffer_load_dwordx4 ... offen lds` with m0 = LDS base): lane l's 16 bytes land at LDS[m0 + 16 l]
static inline void emu_buffer_load_lds16(emu_rsrc r, unsigned voff, unsigned soff, unsigned lds_addr) {
    const emu_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    unsigned char* lds = reinterpret_cast<unsigned char*>(((uintptr_t)emu::g.dyn_lds & ~(uintptr_t)0xffffffffu) | lds_addr);
    emu::dma_issue(lds + 16 * emu::g.cur->lane, &v);          // lands at a covering s_waitcnt vmcnt(n), not before
}

// ---- matrix cores --------------------------------------------------------------------------------------------------------
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef short emu_s16x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x32_bf16: A[i][k] in lane (i = l & 15), k = 8 (l >> 4) + j; B[k][n] in lane (n = l & 15), same k;
// D[i][n] in lane (n = l & 15), rows i = 4 (l >> 4) + r
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 acc, int, int, int) {
    struct Op { float a[8], b[8]; };                  // widened once at the deposit: the 4 x 32 products per lane are plain f32
    static_assert(sizeof(Op) <= emu::SLOT_BYTES, "slot");
    auto c = emu::coll_begin();
    Op& me = c.mine<Op>();
    for (int j = 0; j < 8; ++j) { me.a[j] = (float)a[j]; me.b[j] = (float)b[j]; }
    emu::coll_sync(6);
    const int n = c.lane & 15, g4 = c.lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g4 + r;
        float s = 0.0f;
        for (int q = 0; q < 4; ++q) {
            const float* pa = c.of<Op>(i + 16 * q).a;       // A row i, k = 8 q + j
            const float* pb = c.of<Op>(n + 16 * q).b;       // B column n, k = 8 q + j
            for (int j = 0; j < 8; ++j) s += pa[j] * pb[j];
        }
        acc[r] += s;
    }
    return acc;
}
// v_mfma_f32_16x16x4_f32: A[i][k] in lane (i = l & 15, k = l >> 4), B[k][n] in lane (n = l & 15, k = l >> 4)
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 acc, int, int, int) {
    struct Op { float a, b; };
    auto c = emu::coll_begin();
    c.mine<Op>() = Op{a, b};
    emu::coll_sync(7);
    const int n = c.lane & 15, g4 = c.lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g4 + r;
        float s = 0.0f;
        for (int k = 0; k < 4; ++k) s += c.of<Op>(i + 16 * k).a * c.of<Op>(n + 16 * k).b;
        acc[r] += s;
    }
    return acc;
}
// ds_read_b64_tr_b16: within a 16-lane group lane i supplies the address of its own 8-byte piece (4 x 16 bit) of a 4 x 16
// block -- piece i = row (i >> 2), columns 4 (i & 3) .. +3 -- and receives column i, rows 0..3 (tools/probe_tr16.hip)
template <typename P> static inline emu_s16x4 __builtin_amdgcn_ds_read_tr16_b64_v4i16(P p) {
    auto c = emu::coll_begin();
    emu_s16x4 mine;
    memcpy(&mine, (const void*)(uintptr_t)p, 8);
    c.mine<emu_s16x4>() = mine;
    emu::coll_sync(8);
    const int base = c.lane & ~15, i = c.lane & 15;
    emu_s16x4 out;
    for (int j = 0; j < 4; ++j) out[j] = c.of<emu_s16x4>(base + 4 * j + (i >> 2))[i & 3];
    return out;
}

namespace emu {
static inline unsigned char* dyn_lds() { return g.dyn_lds; }
}
