// Implicit-GEMM convolution for gfx950 (CDNA4), shared by forward conv, data-gradient and (through a
// different kernel in y5m_conv_wgrad.hip) weight-gradient.
//
// Layouts
//   activations : pixel-major / channel-minor (NHWC), element type T (bf16 or f32), (ptr, ld)
//   weights     : packed [Np][Kp], k = tap*Cin + c contiguous ("K-contiguous"), zero padded
// GEMM          : D[n][m] = sum_k W[n][k] * X[m][k]   with m = output pixel, n = output channel
//   MFMA "A" operand = weight rows (n), "B" operand = pixel rows (m): each lane then owns 4
//   CONSECUTIVE CHANNELS of one pixel in its accumulator (C/D layout row=(lane>>4)*4+reg,
//   col=lane&15), so epilogue loads/stores are 8/16-byte vectors along the channel axis.
// Tiles         : BM=128 pixels x BN (48|96) channels x 128 bytes of K per step (64 bf16 | 32 f32);
//   256 threads = 4 waves; LDS rows are 128 B = 8 chunks of 16 B, chunk index XOR-swizzled with
//   (row>>1)&7 so both the 16-byte staging writes and the ds_read_b128 fragment reads are
//   bank-conflict free (MI355X_MICROARCH LDS table: b128 lane groups).
//   Global->register prefetch of tile k+1 is issued before the MFMAs of tile k (one barrier / tile).
#pragma once
#include "y5m_common.h"
#include "y5m_bnfuse.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#define CV_THREADS 256
#define CV_BM 128

enum { EPI_RAW_STATS = 0, EPI_AFFINE_ACT = 1, EPI_HEAD = 2, EPI_DGRAD = 3 };

typedef y5m_conv_args ConvParams;   // the public ABI struct (include/y5m.h) is the kernel argument

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int CH = 4;     // elements per 16-byte chunk
    static constexpr int BK = 32;    // elements per 128-byte K step
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int CH = 8;
    static constexpr int BK = 64;
};

__device__ __forceinline__ int lds_off(int row, int q) { return row * 128 + ((q ^ ((row >> 1) & 7)) << 4); }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

// 4 consecutive channels <-> registers
template <typename T> __device__ __forceinline__ void load4(const T* p, float v[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float v[4]) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float v[4]) {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
    v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float v[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float v[4]) {
    uint2 q;
    q.x = f32x2_to_bf16x2(v[0], v[1]);
    q.y = f32x2_to_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = q;
}

// m -> (m / d, m % d) with a float reciprocal estimate + ONE branch-free integer correction (~8 VALU
// instead of the ~35 of a 32-bit integer division). The estimate m*rcp is off by < (m/d) * 2^-22 + 1 ulp,
// i.e. by at most 1 whenever the quotient is < 2^22 (here: < 52 M pixels / 20), so one +-1 fix is exact.
__device__ __forceinline__ void fast_divmod(int m, int d, float rcp, int& q, int& r) {
    q = (int)((float)m * rcp);
    r = m - q * d;
    const int lo = r < 0 ? 1 : 0, hi = r >= d ? 1 : 0;
    q += hi - lo;
    r += (lo - hi) * d;
}

#ifdef Y5M_ACCURATE_EXP
#define Y5M_EXPF expf
#else
#define Y5M_EXPF __expf
#endif
// 1 / x by the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the IEEE division (v_rcp + 2 Newton steps + fix-up, ~10 VALU):
// sigmoid's denominator is in [1, inf), no denormal / overflow corner. Inside the step the BatchNorm / SiLU passes and the fused
// backward kernels share their CUs with the forked weight gradient: -0.13 ms per step for the fused kernels alone.
// (-DY5M_IEEE_RCP: A/B build with the division.)
#ifdef Y5M_IEEE_RCP
#define Y5M_RCPF(x) (1.0f / (x))
#else
#define Y5M_RCPF(x) __builtin_amdgcn_rcpf(x)
#endif
__device__ __forceinline__ float sigmoid_fast(float x) { return Y5M_RCPF(1.0f + Y5M_EXPF(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_fast(x); }
__device__ __forceinline__ float silu_grad(float t) {
    const float s = Y5M_RCPF(1.0f + Y5M_EXPF(-t));
    return s * (1.0f + t * (1.0f - s));
}
