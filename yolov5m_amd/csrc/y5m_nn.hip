// Non-GEMM kernels of the YOLOv5m train/infer step for gfx950: weight packing, input space-to-depth,
// training-mode BatchNorm (statistics finalise, normalise+SiLU, backward), SPPF pooling, nearest
// upsample, gradient plumbing and the fused clip+Adam optimizer. All HBM-bound: 16/32-byte vector
// accesses along the channel axis of (ptr, ld) NHWC activations.
#include "y5m_conv.h"
#include <stdlib.h>

#define EW_T 256
template <typename T> struct Vec8 { float v[8]; };
template <typename T> __device__ __forceinline__ void load8(const T* p, float v[8]) { load4<T>(p, v); load4<T>(p + 4, v + 4); }
template <typename T> __device__ __forceinline__ void store8(T* p, const float v[8]) { store4<T>(p, v); store4<T>(p + 4, v + 4); }
// streaming variant for tensors that are not read again soon (nontemporal: do not displace what the next kernel needs)
typedef unsigned nn_u32x4 __attribute__((ext_vector_type(4)));
typedef float nn_f32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ void load8_nt(const T* p, float v[8]);
template <> __device__ __forceinline__ void load8_nt<bf16_t>(const bf16_t* p, float v[8]) {
    const nn_u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const nn_u32x4*>(p));
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(q[i] << 16); v[2 * i + 1] = __uint_as_float(q[i] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void load8_nt<float>(const float* p, float v[8]) {
    const nn_f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const nn_f32x4*>(p));
    const nn_f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const nn_f32x4*>(p) + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
// (measured in the full step: nontemporal loads of y in bn_act and of dz, y in bn_bwd_apply -- their last reads before
//  the tensors go cold -- 31.95 -> 31.65 ms)
#ifdef Y5M_EW_NO_NT
#define LOAD8_STREAM load8
#else
#define LOAD8_STREAM load8_nt
#endif

// 8 elements as they lie in memory, unpacked later: a load whose VALUE is first touched in another basic block does not force a
// `s_waitcnt vmcnt(0)` right behind it (tools/isa_audit.py "dep": bn_act_kernel's residual loads sat inside `if (res)` blocks
// together with their bf16 -> f32 unpacking, so every row waited for ALL loads in flight before the next row was requested)
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { nn_u32x4 q; };
template <> struct Raw8<float> { nn_f32x4 a, b; };
__device__ __forceinline__ void raw_load8(const bf16_t* p, Raw8<bf16_t>& r) { r.q = *reinterpret_cast<const nn_u32x4*>(p); }
__device__ __forceinline__ void raw_load8(const float* p, Raw8<float>& r) {
    r.a = *reinterpret_cast<const nn_f32x4*>(p);
    r.b = *(reinterpret_cast<const nn_f32x4*>(p) + 1);
}
__device__ __forceinline__ void raw_unpack8(const Raw8<bf16_t>& r, float v[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r.q[i] << 16); v[2 * i + 1] = __uint_as_float(r.q[i] & 0xffff0000u); }
}
__device__ __forceinline__ void raw_unpack8(const Raw8<float>& r, float v[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = r.a[i]; v[4 + i] = r.b[i]; }
}

#define DISPATCH_T(dtype, ...)                                         \
    if ((dtype) == Y5M_BF16) { using T = bf16_t; __VA_ARGS__ }        \
    else if ((dtype) == Y5M_F32) { using T = float; __VA_ARGS__ }     \
    else { y5m_set_error("bad dtype"); return Y5M_EINVAL; }

static inline unsigned ew_blocks(int64_t n) { return (unsigned)((n + EW_T - 1) / EW_T); }

// =================================================================================================
// weight packing: master f32 [Cout][Cin][KH][KW] (reference state_dict layout) -> K-contiguous rows
// =================================================================================================
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ src, int Cout, int Cin, int KH, int KW, int mode,
                                    int kh0, int khs, int th, int kw0, int kws, int tw, T* __restrict__ dst,
                                    int rows_p, int Kp, int cstride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows_p * Kp) return;
    const int r = (int)(i / Kp), k = (int)(i - (int64_t)r * Kp);
    float v = 0.0f;
    if (mode == 2) {                       // stem: 6x6 s2 on 3ch == 3x3 s1 on space-to-depth 12(+4)ch
        const int tap = k >> 4, cc = k & 15;
        if (r < Cout && tap < 9 && cc < 12) {
            const int a = tap / 3, b = tap - 3 * a;
            const int dyx = cc / 3, c = cc - 3 * dyx;
            const int dy = dyx >> 1, dx = dyx & 1;
            v = src[((r * Cin + c) * KH + (2 * a + dy)) * KW + (2 * b + dx)];
        }
    } else {
        const int Cc = mode == 0 ? Cin : Cout;      // channels per tap of the packed operand
        const int Cs = cstride > 0 ? cstride : Cc;  // per-tap stride (>= Cc; the gap is zero padding)
        const int R = mode == 0 ? Cout : Cin;
        const int tap = k / Cs, cc = k - tap * Cs;
        if (r < R && tap < th * tw && cc < Cc) {
            const int ta = tap / tw, tb = tap - ta * tw;
            const int kh = kh0 + ta * khs, kw = kw0 + tb * kws;
            const int co = mode == 0 ? r : cc, ci = mode == 0 ? cc : r;
            v = src[((co * Cin + ci) * KH + kh) * KW + kw];
        }
    }
    dst[i] = from_f32<T>(v);
}

extern "C" int y5m_pack_weights(const float* src, int Cout, int Cin, int KH, int KW, int mode, int kh0, int khs,
                                int th, int kw0, int kws, int tw, void* dst, int rows_p, int Kp, int cstride,
                                int dtype, void* stream) {
    const int64_t n = (int64_t)rows_p * Kp;
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weights_kernel<T>, dim3(ew_blocks(n)), dim3(EW_T), 0, y5m_stream(stream),
                                         src, Cout, Cin, KH, KW, mode, kh0, khs, th, kw0, kws, tw, (T*)dst, rows_p, Kp, cstride);)
    Y5M_CHECK_LAUNCH("pack_weights_kernel");
    return Y5M_OK;
}

// All weight packs of a step in ONE launch: a device table of jobs with element-count prefixes; each
// thread produces 8 CONSECUTIVE k of one packed row (one job search, one row/tap decomposition and one
// 16/32-byte store per 8 gathered values; Kp and the job starts are multiples of 8).
template <typename T>
__device__ __forceinline__ float pack_value(const y5m_pack_job& J, int r, int k) {
    if (J.mode == 2) {
        const int tap = k >> 4, cc = k & 15;
        if (r < J.Cout && tap < 9 && cc < 12) {
            const int a = tap / 3, b = tap - 3 * a;
            const int dyx = cc / 3, c = cc - 3 * dyx;
            return J.src[((r * J.Cin + c) * J.KH + (2 * a + (dyx >> 1))) * J.KW + (2 * b + (dyx & 1))];
        }
        return 0.0f;
    }
    const int Cc = J.mode == 0 ? J.Cin : J.Cout;
    const int Cs = J.cstride > 0 ? J.cstride : Cc;
    const int R = J.mode == 0 ? J.Cout : J.Cin;
    const int tap = k / Cs, cc = k - tap * Cs;
    if (r < R && tap < J.th * J.tw && cc < Cc) {
        const int ta = tap / J.tw, tb = tap - ta * J.tw;
        const int kh = J.kh0 + ta * J.khs, kw = J.kw0 + tb * J.kws;
        const int co = J.mode == 0 ? r : cc, ci = J.mode == 0 ? cc : r;
        return J.src[((co * J.Cin + ci) * J.KH + kh) * J.KW + kw];
    }
    return 0.0f;
}
template <typename T>
__global__ void pack_batched_kernel(const y5m_pack_job* __restrict__ jobs, int njobs, int64_t total8) {
    const int64_t i8 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i8 >= total8) return;
    const int64_t i = i8 * 8;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].start <= i) lo = mid; else hi = mid - 1;
    }
    const y5m_pack_job J = jobs[lo];
    const int64_t e = i - J.start;
    const int r = (int)(e / J.Kp), k0 = (int)(e - (int64_t)r * J.Kp);
    float v[8];
    const int Cc = J.mode == 0 ? J.Cin : J.Cout;
    const int Cs = J.cstride > 0 ? J.cstride : Cc;
    const int tap = J.mode == 2 ? 0 : k0 / Cs, cc0 = k0 - tap * Cs;
    if (J.mode != 2 && cc0 + 8 <= Cs) {
        // the 8 values share one tap: one decomposition, a constant source stride
        const int R = J.mode == 0 ? J.Cout : J.Cin;
        const bool ok = r < R && tap < J.th * J.tw;
        const int ta = tap / J.tw, tb = tap - ta * J.tw;
        const int kh = J.kh0 + ta * J.khs, kw = J.kw0 + tb * J.kws;
        const int64_t tapoff = (int64_t)kh * J.KW + kw;
        const int64_t khw = (int64_t)J.KH * J.KW;
        // mode 0: src[((r*Cin + cc)*KH + kh)*KW + kw] ; mode 1: src[((cc*Cin + r)*KH + kh)*KW + kw]
        const int64_t base = J.mode == 0 ? (int64_t)r * J.Cin * khw + tapoff : (int64_t)r * khw + tapoff;
        const int64_t step = J.mode == 0 ? khw : (int64_t)J.Cin * khw;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (ok && cc0 + j < Cc) ? J.src[base + (cc0 + j) * step] : 0.0f;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pack_value<T>(J, r, k0 + j);
    }
    store8<T>(reinterpret_cast<T*>(J.dst) + (J.ldd > 0 ? (int64_t)r * J.ldd + k0 : e), v);
}
extern "C" int y5m_pack_weights_batched(const y5m_pack_job* d_jobs, int njobs, int64_t total, int dtype, void* stream) {
    if (njobs <= 0 || total <= 0) return Y5M_OK;
    Y5M_REQUIRE(total % 8 == 0, "packed sizes must be multiples of 8 elements");
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_batched_kernel<T>, dim3(ew_blocks(total / 8)), dim3(EW_T), 0, y5m_stream(stream),
                                         d_jobs, njobs, total / 8);)
    Y5M_CHECK_LAUNCH("pack_batched_kernel");
    return Y5M_OK;
}

// packed weight gradient f32 [N][taps][Cc] -> reference layout [Cout][Cin][KH][KW] (mode as above)
__global__ void unpack_wgrad_kernel(const float* __restrict__ gp, int nslices, int64_t slice_stride, int Cout, int Cin,
                                    int KH, int KW, int mode, int ldg, float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)Cout * Cin * KH * KW;
    if (i >= n) return;
    const int kw = (int)(i % KW);
    int64_t t = i / KW;
    const int kh = (int)(t % KH);
    t /= KH;
    const int ci = (int)(t % Cin), co = (int)(t / Cin);
    int k;
    if (mode == 2) {
        const int a = kh >> 1, dy = kh & 1, b = kw >> 1, dx = kw & 1;
        k = (a * 3 + b) * 16 + (dy * 2 + dx) * 3 + ci;
    } else {
        k = (kh * KW + kw) * Cin + ci;
    }
    const float* p = gp + (size_t)co * ldg + k;
    float v = p[0];
    for (int sidx = 1; sidx < nslices; ++sidx) v += p[(size_t)sidx * slice_stride];     // fixed order: deterministic
    dst[i] = v;
}

extern "C" int y5m_unpack_wgrad(const float* gp, int Cout, int Cin, int KH, int KW, int mode, int ldg, float* dst,
                                void* stream) {
    const int64_t n = (int64_t)Cout * Cin * KH * KW;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(ew_blocks(n)), dim3(EW_T), 0, y5m_stream(stream), gp, 1, (int64_t)0, Cout,
                       Cin, KH, KW, mode, ldg, dst);
    Y5M_CHECK_LAUNCH("unpack_wgrad_kernel");
    return Y5M_OK;
}

// =================================================================================================
// input: NCHW f32 image (B,3,H,W) -> space-to-depth NHWC (B,H/2,W/2,16): ch = (dy*2+dx)*3 + c
// =================================================================================================
template <typename T>
__global__ void s2d_input_kernel(const float* __restrict__ img, int B, int H, int W, T* __restrict__ out) {
    const int W2 = W >> 1, H2 = H >> 1;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per output pixel
    if (i >= (int64_t)B * H2 * W2) return;
    const int x = (int)(i % W2);
    int64_t t = i / W2;
    const int y = (int)(t % H2);
    const int b = (int)(t / H2);
    float v[16];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const float2 q = *reinterpret_cast<const float2*>(img + (((size_t)b * 3 + c) * H + (2 * y + dy)) * W + 2 * x);
            v[(dy * 2 + 0) * 3 + c] = q.x;
            v[(dy * 2 + 1) * 3 + c] = q.y;
        }
    v[12] = v[13] = v[14] = v[15] = 0.0f;
    store8<T>(out + i * 16, v);
    store8<T>(out + i * 16 + 8, v + 8);
}

extern "C" int y5m_s2d_input(const float* img, int B, int H, int W, void* out, int dtype, void* stream) {
    Y5M_REQUIRE((H % 2) == 0 && (W % 2) == 0, "H,W must be even");
    const int64_t n = (int64_t)B * (H / 2) * (W / 2);
    DISPATCH_T(dtype, hipLaunchKernelGGL(s2d_input_kernel<T>, dim3(ew_blocks(n)), dim3(EW_T), 0, y5m_stream(stream), img,
                                         B, H, W, (T*)out);)
    Y5M_CHECK_LAUNCH("s2d_input_kernel");
    return Y5M_OK;
}

// =================================================================================================
// input stage on the device: uint8 (B,3,Hs,Ws) -> float /255 -> bilinear resize to (H,W)
//   == F.interpolate(img.float()/255, size=(H,W), mode="bilinear", align_corners=False), the reference's
//   `images.float()/255` + multi_scale (utils/training_utils.py:11-28, :98-100) -- on the GPU the host only
//   ships the uint8 batch (4x fewer bytes over PCIe) and no fp32 image ever exists on the host.
// One thread per output pixel (3 channels); writes are coalesced along x in each channel plane.
// =================================================================================================
__global__ void preprocess_u8_kernel(const unsigned char* __restrict__ img, int B, int Hs, int Ws, float* __restrict__ out,
                                     int H, int W, float sh, float sw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * H * W) return;
    const int x = (int)(i % W);
    int64_t t = i / W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    // ATen area_pixel_compute_source_index(align_corners=false): max(scale*(dst+0.5)-0.5, 0)
    const float fy = fmaxf(sh * ((float)y + 0.5f) - 0.5f, 0.0f), fx = fmaxf(sw * ((float)x + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned char* p = img + ((size_t)b * 3 + c) * Hs * Ws;
        // the reference divides by 255 BEFORE interpolating: same order here (v/255 is not exact in fp32)
        const float v00 = (float)p[(size_t)y0 * Ws + x0] / 255.0f, v01 = (float)p[(size_t)y0 * Ws + x1] / 255.0f;
        const float v10 = (float)p[(size_t)y1 * Ws + x0] / 255.0f, v11 = (float)p[(size_t)y1 * Ws + x1] / 255.0f;
        out[(((size_t)b * 3 + c) * H + y) * W + x] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    }
}

extern "C" int y5m_preprocess_u8(const unsigned char* img, int B, int Hs, int Ws, float* out, int H, int W, void* stream) {
    Y5M_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0, "shape");
    const int64_t n = (int64_t)B * H * W;
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(ew_blocks(n)), dim3(EW_T), 0, y5m_stream(stream), img, B, Hs, Ws, out, H, W,
                       (float)Hs / (float)H, (float)Ws / (float)W);
    Y5M_CHECK_LAUNCH("preprocess_u8_kernel");
    return Y5M_OK;
}

// =================================================================================================
// BatchNorm (training): finalise statistics from the conv epilogue partials
// =================================================================================================
// stats [tiles_m][2][Np] (sum, sumsq) -> mean, biased var; scale = g*invstd, shift = b - mean*scale;
// running stats: momentum, UNBIASED variance (reference model.py:17, nn.BatchNorm2d semantics)
//
// ONE launch: every block reduces a row range of the partials (64 channels x 4 row lanes, coalesced
// 256-byte rows) into stage[S][2][C]; the block that takes the LAST ticket of its channel group then sums
// the S stage rows in a fixed order (f64) and finalises -- deterministic whatever the arrival order.
// ctr[] (one counter per 64-channel group) must be zero on entry and is left zero.
struct BnFinArgs {
    double count;
    const float* gamma; const float* beta; float* rmean; float* rvar; float momentum, eps;
    float* scale; float* shift; float* mean_o; float* invstd_o; int update_running;
};
struct BnBwdFinArgs {
    const float* scale; const float* mean; const float* invstd; float invM;
    float* cB; float* cD; float* dgamma; float* dbeta; int accumulate; int centered;
};

// [R][2][ld] partial sums -> per-channel totals -> finalise (MODE 0: forward statistics, MODE 1: backward
// coefficients). 1024 threads = 64 channels x 16 row lanes; gridDim.x = S row splits (<= 64).
template <int MODE>
__global__ __launch_bounds__(1024) void bn_reduce_finalize_kernel(const float* __restrict__ in, int R, int ld, int C,
                                                                 float* __restrict__ stage, unsigned* __restrict__ ctr,
                                                                 BnFinArgs F, BnBwdFinArgs G) {
    __shared__ float sm[2][16][64];
    __shared__ double sd[2][16][64];
    __shared__ int s_last;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const int S = gridDim.x;
    const int chunk = (R + S - 1) / S;
    const int r0 = blockIdx.x * chunk, r1 = min(R, r0 + chunk);
    float a = 0.f, b = 0.f;
    if (c < C) {
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += 16) {
            a += in[((size_t)r * 2 + 0) * ld + c];
            b += in[((size_t)r * 2 + 1) * ld + c];
        }
    }
    sm[0][rl][cl] = a; sm[1][rl][cl] = b;
    __syncthreads();
    // Cross-workgroup hand-off WITHOUT a device-scope fence: __threadfence() is a write-back of the whole
    // (per-XCD, non-coherent) L2 on gfx950 -- tens of microseconds right after a conv that dirtied it. The
    // few stage floats are instead written / read with agent-scope relaxed atomics (sc1: written through to,
    // and read from, the memory side), ordered against the ticket by the stores' own completion
    // (workgroup-scope release = s_waitcnt vmcnt(0), then the barrier, then the agent-scope ticket).
    if (rl < 2 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += sm[rl][l][cl];
        __hip_atomic_store(stage + ((size_t)blockIdx.x * 2 + rl) * C + c, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the block that takes the last ticket of this channel group finalises ---------------------
    // Publish form R1 of the CDNA4 guide (section 6, Guideline 16): write-through (sc1) payload stores -> EVERY storing
    // wave drains them (explicit s_waitcnt vmcnt(0): inline asm, so the compiler cannot drop it the way it may drop the
    // wait of a fence it considers redundant) -> workgroup barrier -> ONE lane takes the agent-scope ticket; the last
    // arriver reads the stage with sc1 loads, which are served from the memory side and cannot hit a stale L1 / foreign-L2
    // line. No agent-scope fence is needed for data that never sits dirty in an L2 (a fence would write back the whole
    // per-XCD L2: ~30 us right after a conv).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = (__hip_atomic_fetch_add(ctr + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)S - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    double da = 0.0, db = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int r = rl; r < S; r += 16) {
            da += (double)__hip_atomic_load(stage + ((size_t)r * 2 + 0) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            db += (double)__hip_atomic_load(stage + ((size_t)r * 2 + 1) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    sd[0][rl][cl] = da; sd[1][rl][cl] = db;
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(ctr + blockIdx.y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (rl != 0 || c >= C) return;
#pragma unroll
    for (int l = 1; l < 16; ++l) { da += sd[0][l][cl]; db += sd[1][l][cl]; }
    if constexpr (MODE == 0) {
        const double mean = da / F.count;
        double var = db / F.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)F.eps));
        const float sc = F.gamma[c] * invstd;
        F.scale[c] = sc;
        F.shift[c] = F.beta[c] - (float)mean * sc;
        F.mean_o[c] = (float)mean;
        F.invstd_o[c] = invstd;
        if (F.update_running) {
            const double unb = F.count > 1.0 ? var * F.count / (F.count - 1.0) : var;
            F.rmean[c] = (1.0f - F.momentum) * F.rmean[c] + F.momentum * (float)mean;
            F.rvar[c] = (1.0f - F.momentum) * F.rvar[c] + F.momentum * (float)unb;
        }
    } else {
        const float mu = G.mean[c], is = G.invstd[c], sc = G.scale[c];
        // centred (standalone reduce: db = sum dt*(y - mean), apply uses y - mean) or not (partials emitted by a conv
        // epilogue, which has no mean at hand). The centred form is the one ATen's batch_norm_backward evaluates; the
        // uncentred one cancels |mean|/std digits in sum dt*y - mean*sum dt and again in cB*y + cD, per layer: it left
        // the stem's gradients 2-3e-3 from a float64 evaluation where the reference's own f32 path is at 1e-4.
        const float dbeta = (float)da;
        const float dgamma = G.centered ? is * (float)db : is * (float)(db - (double)mu * da);
        const float cB = -sc * dgamma * is * G.invM;
        G.cB[c] = cB;
        G.cD[c] = G.centered ? -sc * dbeta * G.invM : -sc * dbeta * G.invM - cB * mu;
        if (G.dgamma && G.dbeta) {
            if (G.accumulate) { G.dbeta[c] += dbeta; G.dgamma[c] += dgamma; }
            else { G.dbeta[c] = dbeta; G.dgamma[c] = dgamma; }
        }
    }
}

#define BN_SPLITS 64       // max row splits (= stage rows) of a fused reduce + finalise launch
static inline int bn_splits(int64_t R) {
    int64_t s = (R + 15) / 16;                 // >= one row per row lane
    return (int)(s < 1 ? 1 : (s > BN_SPLITS ? BN_SPLITS : s));
}
static inline size_t bn_stage_bytes(int C) { return y5m_align((size_t)BN_SPLITS * 2 * (size_t)C * sizeof(float)); }
// The ticket counters live at the START of a workspace, at a FIXED offset and size: one workspace is shared
// by layers of different widths, and a counter at a width-dependent offset would sit inside another layer's
// stage rows (garbage instead of the zero the protocol needs).
#define BN_CTR_BYTES 1024      /* 256 counters = 16384 channels */

extern "C" size_t y5m_bn_finalize_workspace_bytes(int Np) { return BN_CTR_BYTES + bn_stage_bytes(Np); }

// ws: zero-filled by the caller before its FIRST use (ticket counters; every call leaves them zero); one
// call at a time per workspace.
extern "C" int y5m_bn_finalize(const float* stats, int tiles_m, int Np, int C, int64_t count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* scale, float* shift, float* mean_out, float* invstd_out, int update_running,
                               void* ws, size_t ws_bytes, void* stream) {
    if (ws_bytes < y5m_bn_finalize_workspace_bytes(Np)) { y5m_set_error("bn_finalize ws too small"); return Y5M_EWS; }
    Y5M_REQUIRE(C <= Np, "C <= Np");
    hipStream_t st = y5m_stream(stream);
    Y5M_REQUIRE(C <= 16384, "C too large for the ticket counters");
    unsigned* ctr = reinterpret_cast<unsigned*>(ws);
    float* stage = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + BN_CTR_BYTES);
    BnFinArgs F{(double)count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean_out, invstd_out, update_running};
    BnBwdFinArgs G{};
    hipLaunchKernelGGL(bn_reduce_finalize_kernel<0>, dim3((unsigned)bn_splits(tiles_m), (unsigned)((C + 63) / 64)), dim3(1024), 0,
                       st, stats, tiles_m, Np, C, stage, ctr, F, G);
    Y5M_CHECK_LAUNCH("bn_reduce_finalize_kernel");
    return Y5M_OK;
}

// eval-mode fold: scale = g/sqrt(rv+eps), shift = b - rm*scale
__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                               int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rvar[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}
__global__ void bn_fold_batched_kernel(const y5m_fold_job* __restrict__ jobs, int njobs, int total, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].start <= i) lo = mid; else hi = mid - 1;
    }
    const y5m_fold_job J = jobs[lo];
    const int c = i - J.start;
    const float sc = J.gamma[c] / sqrtf(J.running_var[c] + eps);
    J.scale[c] = sc;
    J.shift[c] = J.beta[c] - J.running_mean[c] * sc;
}
extern "C" int y5m_bn_fold_batched(const y5m_fold_job* d_jobs, int njobs, int total_channels, float eps, void* stream) {
    if (njobs <= 0 || total_channels <= 0) return Y5M_OK;
    hipLaunchKernelGGL(bn_fold_batched_kernel, dim3((unsigned)((total_channels + 255) / 256)), dim3(256), 0, y5m_stream(stream),
                       d_jobs, njobs, total_channels, eps);
    Y5M_CHECK_LAUNCH("bn_fold_batched_kernel");
    return Y5M_OK;
}
extern "C" int y5m_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                           float eps, int C, float* scale, float* shift, void* stream) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, y5m_stream(stream), gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    Y5M_CHECK_LAUNCH("bn_fold_kernel");
    return Y5M_OK;
}

// ---- row-loop geometry of the BN elementwise kernels --------------------------------------------------
// A block covers CG 16-byte channel chunks x RP = 256/CG pixels per pass and walks the pixels with a grid
// stride: a thread keeps ITS chunk's per-channel coefficients in registers for the whole kernel (they were
// re-loaded per pixel before: 6 parameter vectors per 2 data vectors) and has U independent rows in flight.
struct EwGeom { int CG, groups, RP; unsigned gx; };
static int g_ew_gx = -1;
static inline EwGeom ew_geom(int64_t M, int C8, int max_gx, int threads = 256) {
    if (g_ew_gx < 0) { const char* e = getenv("Y5M_EW_GX"); g_ew_gx = e ? atoi(e) : 0; }
    if (g_ew_gx > 0 && max_gx > 512) max_gx = g_ew_gx;
    EwGeom g;
    g.CG = 1;
    for (int d = C8 < 32 ? C8 : 32; d >= 1; --d) if (C8 % d == 0) { g.CG = d; break; }
    g.groups = C8 / g.CG;
    g.RP = threads / g.CG;
    int64_t gx = (M + g.RP - 1) / g.RP;
    const int64_t cap = max_gx / g.groups > 0 ? max_gx / g.groups : 1;
    g.gx = (unsigned)(gx < 1 ? 1 : (gx > cap ? cap : gx));
    return g;
}

// FUSED forms of the BatchNorm consumers (y5m_bnfuse.h): the per-channel sums come from the accumulator rows the
// producer launch added into, and every workgroup derives the coefficients of ITS channel group in a prologue (one L2
// round trip + a little f64 arithmetic, next to the other resident workgroups' streaming); the workgroups with
// blockIdx.x == 0 also write the per-channel arrays other launches read.
#define BNF_EW_GX 1024       /* grid cap of the fused consumers (see y5m_bn_act_fused); re-swept inside the step after the SiLU reciprocal change: 640 / 768 / 1024 / 1280 / 1536 / 2048 = +0.2 / 0 / 0 / +0.1 / +0.12 / +0.07 ms (Y5M_EW_GX) */
struct BnFusedFwd {
    const double* acc; int ldacc;     // [BNF_SLOTS][2][ldacc], already offset to this layer's first channel
    double count;
    const float* gamma; const float* beta; float* rmean; float* rvar; float momentum, eps;
    float* scale; float* shift; float* mean_o; float* invstd_o; int update_running;
};

// z = act(y*scale + shift) (+ res)
// R4 (Y5M_R4_KERNELS bit 1, y5m_common.h): residual rows kept raw and unpacked where they are added (round 4, not yet measured
// on hardware); false = the round-3 form (residual rows unpacked behind their loads), hardware-verified.
template <typename T, bool FUSED, bool R4>
__global__ __launch_bounds__(256) void bn_act_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const T* __restrict__ res, int ldres,
                                                    T* __restrict__ out, int ldout, int64_t M, int CG, int RP, int act,
                                                    const BnFusedFwd F) {
    const int cl = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int c = (blockIdx.y * CG + cl) * 8;
    const int64_t stride = (int64_t)gridDim.x * RP;
    int64_t m = (int64_t)blockIdx.x * RP + rl;
    const bool active = rl < RP;
    float sc[8], sh[8];
    float v[4][8];
    Raw8<T> r[4];                                           // R4: residual rows, unpacked where they are added (finish4)
    float r3[R4 ? 1 : 4][8];                                // round-3 form: residual rows as floats
    auto load_res = [&](int u) __attribute__((always_inline)) {
        if constexpr (R4) raw_load8(res + (m + u * stride) * ldres + c, r[u]);
        else load8<T>(res + (m + u * stride) * ldres + c, r3[u]);
    };
    // FUSED: the first four rows are requested BEFORE the coefficient prologue, so that its round trip to the accumulator
    // rows and its f64 arithmetic run under their HBM latency (without this every workgroup starts with ~3 us of nothing in
    // flight: +4 us per launch with the ~4 rounds of workgroups a CU runs)
    const bool first4 = FUSED && active && m + 3 * stride < M;
    if (first4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            LOAD8_STREAM<T>(y + (m + u * stride) * ldy + c, v[u]);
            if (res) load_res(u);
        }
    }
    if constexpr (FUSED) {
        __shared__ float s_co[2][256];                      // CG * 8 <= 256 channels per workgroup
        const int nch = CG * 8, cbase = blockIdx.y * nch;
        for (int i = threadIdx.x; i < nch; i += 256) {
            const int ch = cbase + i;
            double da, db;
            bnf_sum(F.acc, F.ldacc, ch, da, db);
            // same arithmetic as bn_reduce_finalize_kernel<0>
            const double mean = da / F.count;
            double var = db / F.count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)F.eps));
            const float s1 = F.gamma[ch] * invstd;
            const float s0 = F.beta[ch] - (float)mean * s1;
            s_co[0][i] = s1;
            s_co[1][i] = s0;
            if (blockIdx.x == 0) {
                F.scale[ch] = s1;
                F.shift[ch] = s0;
                F.mean_o[ch] = (float)mean;
                F.invstd_o[ch] = invstd;
                if (F.update_running) {
                    const double unb = F.count > 1.0 ? var * F.count / (F.count - 1.0) : var;
                    F.rmean[ch] = (1.0f - F.momentum) * F.rmean[ch] + F.momentum * (float)mean;
                    F.rvar[ch] = (1.0f - F.momentum) * F.rvar[ch] + F.momentum * (float)unb;
                }
            }
        }
        __syncthreads();
        if (!active) return;
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = s_co[0][cl * 8 + k]; sh[k] = s_co[1][cl * 8 + k]; }
    } else {
        if (!active) return;
        load8<float>(scale + c, sc);
        load8<float>(shift + c, sh);
    }
    auto finish4 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float rr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if constexpr (R4) { if (res) raw_unpack8(r[u], rr); }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[u][k] = v[u][k] * sc[k] + sh[k];
                if (act == Y5M_ACT_SILU) v[u][k] = silu_f(v[u][k]);
                if constexpr (R4) { if (res) v[u][k] += rr[k]; }
                else { if (res) v[u][k] += r3[u][k]; }
            }
            store8<T>(out + (m + u * stride) * ldout + c, v[u]);
        }
    };
    if (first4) { finish4(); m += 4 * stride; }
    for (; m + 3 * stride < M; m += 4 * stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            LOAD8_STREAM<T>(y + (m + u * stride) * ldy + c, v[u]);
            if (res) load_res(u);
        }
        finish4();
    }
    for (; m < M; m += stride) {
        float w[8], q[8];
        LOAD8_STREAM<T>(y + m * ldy + c, w);
        if (res) load8<T>(res + m * ldres + c, q);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            w[k] = w[k] * sc[k] + sh[k];
            if (act == Y5M_ACT_SILU) w[k] = silu_f(w[k]);
            if (res) w[k] += q[k];
        }
        store8<T>(out + m * ldout + c, w);
    }
}

extern "C" int y5m_bn_act(const void* y, int ldy, const float* scale, const float* shift, const void* res, int ldres,
                          void* out, int ldout, int64_t M, int C, int act, int dtype, void* stream) {
    Y5M_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    Y5M_REQUIRE(!res || (ldres % 4 == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0 && (dtype != Y5M_BF16 || ldres % 8 == 0)),
                "res: 16-byte aligned rows (the residual is read in 16-byte pieces)");
    const EwGeom g = ew_geom(M, C / 8, 4096);
    DISPATCH_T(dtype, {
        auto kern = (y5m_r4_forms() & Y5M_R4_BN_ACT) ? bn_act_kernel<T, false, true> : bn_act_kernel<T, false, false>;
        hipLaunchKernelGGL(kern, dim3(g.gx, (unsigned)g.groups), dim3(256), 0, y5m_stream(stream), (const T*)y, ldy, scale, shift,
                           (const T*)res, ldres, (T*)out, ldout, M, g.CG, g.RP, act, BnFusedFwd{});
    })
    Y5M_CHECK_LAUNCH("bn_act_kernel");
    return Y5M_OK;
}

// Y5M_BN_FUSE (default 1): accumulator rows + fused consumers (y5m_bnfuse.h); 0 = partial rows + bn_reduce_finalize_kernel
// everywhere (A/B runs); 2 = forward only, 3 = backward only
static int bn_fuse_mode(void) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("Y5M_BN_FUSE"); v = e ? atoi(e) : 1; }
    return v;
}
extern "C" int y5m_bn_fuse_enabled(void) { return bn_fuse_mode(); }
extern "C" int y5m_bn_acc_slots(void) { return BNF_SLOTS; }

// training-mode BatchNorm + activation straight from the accumulator rows a y5m_conv(bn_acc) launch filled:
// finalise (reference model.py:17: eps, momentum, biased batch variance for the normalisation, unbiased for the running
// estimate) + normalise + act (+ res) in ONE launch. acc points at this layer's first channel inside rows of ldacc
// doubles. Writes scale / shift / mean / invstd [C] (read by the backward pass) and updates the running statistics.
extern "C" int y5m_bn_act_fused(const void* y, int ldy, const double* acc, int ldacc, int64_t count, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                int update_running, float* scale, float* shift, float* mean_out, float* invstd_out,
                                const void* res, int ldres, void* out, int ldout, int64_t M, int C, int act, int dtype,
                                void* stream) {
    Y5M_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    Y5M_REQUIRE(acc && gamma && beta && scale && shift && mean_out && invstd_out, "null pointer");
    Y5M_REQUIRE(!update_running || (running_mean && running_var), "running statistics missing");
    Y5M_REQUIRE(!res || (ldres % 4 == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0 && (dtype != Y5M_BF16 || ldres % 8 == 0)),
                "res: 16-byte aligned rows (the residual is read in 16-byte pieces)");
    // (every workgroup pays the coefficient prologue, so fewer, longer workgroups than the plain form: 4096 / 2048 / 1536 /
    //  1024 = 27.44 / 27.24 / 27.14 / 27.3 ms per step, B=64 @ 640^2; the plain form wants 4096)
    const EwGeom g = ew_geom(M, C / 8, BNF_EW_GX);
    const BnFusedFwd F{acc, ldacc, (double)count, gamma, beta, running_mean, running_var, momentum, eps,
                       scale, shift, mean_out, invstd_out, update_running};
    DISPATCH_T(dtype, {
        auto kern = (y5m_r4_forms() & Y5M_R4_BN_ACT) ? bn_act_kernel<T, true, true> : bn_act_kernel<T, true, false>;
        hipLaunchKernelGGL(kern, dim3(g.gx, (unsigned)g.groups), dim3(256), 0, y5m_stream(stream), (const T*)y, ldy,
                           (const float*)nullptr, (const float*)nullptr, (const T*)res, ldres, (T*)out, ldout, M, g.CG, g.RP, act, F);
    })
    Y5M_CHECK_LAUNCH("bn_act_kernel");
    return Y5M_OK;
}

// =================================================================================================
// BatchNorm + SiLU backward
//   t = y*scale + shift ; z = silu(t) ; dt = dz * silu'(t) ; xhat = (y - mean)*invstd
//   dbeta = sum dt ; dgamma = sum dt*xhat = invstd*(sum dt*y - mean*sum dt)
//   dy = gamma*invstd*(dt - dbeta/M - xhat*dgamma/M) = scale*dt + cB*y + cD
//        cB = -scale*dgamma*invstd/M ; cD = -scale*dbeta/M - cB*mean            (scale = gamma*invstd)
// evaluated in the CENTRED form dy = scale*dt + cB*(y - mean) - scale*dbeta/M with sum dt*(y - mean) from the reduce pass
// (see bn_reduce_finalize_kernel<1>).
// Three launches: (1) reduce (sum dt, sum dt*y) -> <= 512 partial rows, (2) bn_reduce_finalize_kernel<1>:
// dgamma/dbeta and the apply coefficients cB/cD, (3) apply.
// =================================================================================================

#ifndef BNR_THREADS
#define BNR_THREADS 256
#endif
// accum != NULL: the block partials are ADDED into the accumulator rows accum[BNF_SLOTS][2][C] (y5m_bnfuse.h) instead of
// being stored as partial rows
// FORM (Y5M_R4_KERNELS bit 4, y5m_common.h): 0 = the round-3 form, hardware-verified (the activation a run-time argument, rows
// unpacked behind their loads); 1 / 2 = the round-4 form without / with SiLU (compile-time activation, the eight loads of four
// rows raw behind one scheduling barrier; not yet measured on hardware).
template <typename T, int FORM>
__global__ __launch_bounds__(BNR_THREADS) void bn_bwd_reduce_kernel(const T* __restrict__ dz, int lddz, const T* __restrict__ y,
                                                           int ldy, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ mean,
                                                           int64_t M, int C, int CG, int RP, int act, float* __restrict__ part,
                                                           double* __restrict__ accum) {
    const bool SILU = FORM == 0 ? act == Y5M_ACT_SILU : FORM == 2;
    __shared__ float sm[2][BNR_THREADS][9];    // [which][thread][k] (+1: the 8-float rows land on distinct banks)
    const int cl = threadIdx.x % CG, rl = threadIdx.x / CG;
    const bool active = rl < RP;
    const int c = (blockIdx.y * CG + cl) * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
    if (active) {
        float sc[8], sh[8], mu[8];
        load8<float>(scale + c, sc);
        load8<float>(shift + c, sh);
        load8<float>(mean + c, mu);
        const int64_t stride = (int64_t)gridDim.x * RP;
        int64_t m = (int64_t)blockIdx.x * RP + rl;
        if constexpr (FORM == 0) {
            for (; m + 3 * stride < M; m += 4 * stride) {
                float g[4][8], yv[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    load8<T>(dz + (m + u * stride) * lddz + c, g[u]);
                    load8<T>(y + (m + u * stride) * ldy + c, yv[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float dt = act == Y5M_ACT_SILU ? g[u][k] * silu_grad(yv[u][k] * sc[k] + sh[k]) : g[u][k];
                        s1[k] += dt;
                        s2[k] += dt * (yv[u][k] - mu[k]);
                    }
            }
        } else
        for (; m + 3 * stride < M; m += 4 * stride) {
            // the eight 16-byte loads of four rows are issued together and kept RAW (32 registers instead of 64 unpacked floats:
            // four waves per SIMD); a row is unpacked where it is consumed. SILU is a template parameter: the per-element test of
            // `act` was 32 uniform branches + their mask arithmetic per iteration (636 -> ~400 instructions, tools/isa_audit.py);
            // without the scheduling barrier the compiler sinks every load in front of its first use (load, vmcnt(0), compute, ...)
            Raw8<T> rg[4], ry[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                raw_load8(dz + (m + u * stride) * lddz + c, rg[u]);
                raw_load8(y + (m + u * stride) * ldy + c, ry[u]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float g[8], yv[8];
                raw_unpack8(rg[u], g);
                raw_unpack8(ry[u], yv);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float dt = SILU ? g[k] * silu_grad(yv[k] * sc[k] + sh[k]) : g[k];
                    s1[k] += dt;
                    s2[k] += dt * (yv[k] - mu[k]);
                }
            }
        }
        for (; m < M; m += stride) {
            float g[8], yv[8];
            load8<T>(dz + m * lddz + c, g);
            load8<T>(y + m * ldy + c, yv);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float dt = SILU ? g[k] * silu_grad(yv[k] * sc[k] + sh[k]) : g[k];
                s1[k] += dt;
                s2[k] += dt * (yv[k] - mu[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { sm[0][threadIdx.x][k] = s1[k]; sm[1][threadIdx.x][k] = s2[k]; }
    __syncthreads();
    // block partial: NCH = CG*8 channels x 2 sums, summed over the RP row lanes in a fixed order
    const int NCH = CG * 8;
    for (int t = threadIdx.x; t < 2 * NCH; t += BNR_THREADS) {
        const int which = t / NCH, ch = t - which * NCH;
        float acc = 0.f;
        for (int r = 0; r < RP; ++r) acc += sm[which][r * CG + (ch >> 3)][ch & 7];
        if (accum) bnf_add(accum, C, blockIdx.x, which, blockIdx.y * NCH + ch, acc);
        else part[((size_t)blockIdx.x * 2 + which) * C + blockIdx.y * NCH + ch] = acc;
    }
}

// dy = scale*dt + cB*(y - mean) + cD   (mean == NULL: the uncentred coefficients of the fused-reduction path, y as is)
struct BnFusedBwd {
    const double* acc;                // [BNF_SLOTS][2][C]: (sum dt, sum dt*(y - mean)) from bn_bwd_reduce_kernel
    const float* invstd; float invM;
    float* dgamma; float* dbeta; int accumulate;
};
template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dz, int lddz, const T* __restrict__ y, int ldy,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ cB, const float* __restrict__ cD,
                                                          const float* __restrict__ mean,
                                                          T* __restrict__ dy, int lddy, int64_t M, int C, int CG, int RP, int act,
                                                          const BnFusedBwd G) {
    const int cl = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int c = (blockIdx.y * CG + cl) * 8;
    const int64_t stride = (int64_t)gridDim.x * RP;
    int64_t m = (int64_t)blockIdx.x * RP + rl;
    const bool active = rl < RP;
    float sc[8], sh[8], kb[8], kd[8], mu[8];
    float g[4][8], yv[4][8];
    // FUSED: first four rows requested ahead of the coefficient prologue (see bn_act_kernel)
    const bool first4 = FUSED && active && m + 3 * stride < M;
    if (first4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            LOAD8_STREAM<T>(dz + (m + u * stride) * lddz + c, g[u]);
            LOAD8_STREAM<T>(y + (m + u * stride) * ldy + c, yv[u]);
        }
    }
    if constexpr (FUSED) {
        // the apply coefficients from the reduce launch's accumulator rows (arithmetic of bn_reduce_finalize_kernel<1>,
        // centred form); blockIdx.x == 0 also writes the parameter gradients
        __shared__ float s_co[2][256];
        const int nch = CG * 8, cbase = blockIdx.y * nch;
        for (int i = threadIdx.x; i < nch; i += 256) {
            const int ch = cbase + i;
            double da, db;
            bnf_sum(G.acc, C, ch, da, db);
            const float is = G.invstd[ch], s1 = scale[ch];
            const float dbeta = (float)da;
            const float dgamma = is * (float)db;
            s_co[0][i] = -s1 * dgamma * is * G.invM;
            s_co[1][i] = -s1 * dbeta * G.invM;
            if (blockIdx.x == 0 && G.dgamma && G.dbeta) {
                if (G.accumulate) { G.dbeta[ch] += dbeta; G.dgamma[ch] += dgamma; }
                else { G.dbeta[ch] = dbeta; G.dgamma[ch] = dgamma; }
            }
        }
        __syncthreads();
        if (!active) return;
#pragma unroll
        for (int k = 0; k < 8; ++k) { kb[k] = s_co[0][cl * 8 + k]; kd[k] = s_co[1][cl * 8 + k]; }
    } else {
        if (!active) return;
        load8<float>(cB + c, kb); load8<float>(cD + c, kd);
    }
    load8<float>(scale + c, sc); load8<float>(shift + c, sh);
    if (mean) load8<float>(mean + c, mu);
    else {
#pragma unroll
        for (int k = 0; k < 8; ++k) mu[k] = 0.f;
    }
    // the activation as a BLEND (a1 = 1: SiLU, 0: none) instead of a test of `act` per element: the loop body becomes straight-line
    // packed-f32 code (42.8 -> 36.6 us per launch inside the step). The same change made bn_bwd_reduce_kernel SLOWER (the scheduler then
    // sinks its loads in front of their uses, and pinning them costs registers: 44 -> 52-68 us), so that kernel keeps the test.
    const float a1 = act == Y5M_ACT_SILU ? 1.0f : 0.0f, a0 = 1.0f - a1;
    auto finish4 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float dt = g[u][k] * fmaf(a1, silu_grad(yv[u][k] * sc[k] + sh[k]), a0);
                o[k] = sc[k] * dt + (kb[k] * (yv[u][k] - mu[k]) + kd[k]);
            }
            store8<T>(dy + (m + u * stride) * lddy + c, o);
        }
    };
    if (first4) { finish4(); m += 4 * stride; }
    for (; m + 3 * stride < M; m += 4 * stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            LOAD8_STREAM<T>(dz + (m + u * stride) * lddz + c, g[u]);
            LOAD8_STREAM<T>(y + (m + u * stride) * ldy + c, yv[u]);
        }
        if constexpr (!FUSED) __builtin_amdgcn_sched_barrier(0);   // (this variant's straight-line body otherwise gets its loads sunk in front of their uses)
        finish4();
    }
    for (; m < M; m += stride) {
        float g1[8], y1[8], o[8];
        LOAD8_STREAM<T>(dz + m * lddz + c, g1);
        LOAD8_STREAM<T>(y + m * ldy + c, y1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float dt = g1[k] * fmaf(a1, silu_grad(y1[k] * sc[k] + sh[k]), a0);
            o[k] = sc[k] * dt + (kb[k] * (y1[k] - mu[k]) + kd[k]);
        }
        store8<T>(dy + m * lddy + c, o);
    }
}

#define BNR_MAX_GX 512     // partial rows of the backward reduce (gx * groups <= 512 blocks: 2 per CU)
extern "C" size_t y5m_bn_bwd_workspace_bytes(int64_t M, int C) {
    return BN_CTR_BYTES + y5m_align((size_t)BNR_MAX_GX * 2 * (size_t)C * 4) + y5m_align((size_t)2 * C * 4) + bn_stage_bytes(C);
}

// Full BN+SiLU backward of one CBL: param grads (dgamma, dbeta: accumulate flag) and dy.
// ws: zero-filled by the caller before its FIRST use (ticket counters; every call leaves them zero); one
// call at a time per workspace.
extern "C" int y5m_bn_bwd(const void* dz, int lddz, const void* y, int ldy, const float* scale, const float* shift,
                          const float* mean, const float* invstd, int64_t M, int C, int act, float* dgamma,
                          float* dbeta, int accumulate_param_grads, void* dy, int lddy, void* ws, size_t ws_bytes,
                          int dtype, void* stream) {
    Y5M_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    if (ws_bytes < y5m_bn_bwd_workspace_bytes(M, C)) { y5m_set_error("bn_bwd ws too small"); return Y5M_EWS; }
    Y5M_REQUIRE(C <= 16384, "C too large for the ticket counters");
    char* w = reinterpret_cast<char*>(ws);
    unsigned* ctr = reinterpret_cast<unsigned*>(w);             // fixed place: see BN_CTR_BYTES
    w += BN_CTR_BYTES;
    float* part = reinterpret_cast<float*>(w);
    w += y5m_align((size_t)BNR_MAX_GX * 2 * C * 4);
    float* cB = reinterpret_cast<float*>(w), *cD = cB + C;
    w += y5m_align((size_t)2 * C * 4);
    float* stage = reinterpret_cast<float*>(w);
    hipStream_t st = y5m_stream(stream);
    const EwGeom gr = ew_geom(M, C / 8, BNR_MAX_GX, BNR_THREADS);   // (swept 256..2048 in the full step: 512 is best)
    DISPATCH_T(dtype, {
        auto kern = !(y5m_r4_forms() & Y5M_R4_BN_BWD_REDUCE) ? bn_bwd_reduce_kernel<T, 0>
                    : act == Y5M_ACT_SILU ? bn_bwd_reduce_kernel<T, 2> : bn_bwd_reduce_kernel<T, 1>;
        hipLaunchKernelGGL(kern, dim3(gr.gx, (unsigned)gr.groups), dim3(BNR_THREADS), 0, st, (const T*)dz, lddz, (const T*)y, ldy,
                           scale, shift, mean, M, C, gr.CG, gr.RP, act, part, (double*)nullptr);
    })
    Y5M_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    BnFinArgs F{};
    BnBwdFinArgs G{scale, mean, invstd, 1.0f / (float)M, cB, cD, dgamma, dbeta, accumulate_param_grads, 1};
    hipLaunchKernelGGL(bn_reduce_finalize_kernel<1>, dim3((unsigned)bn_splits(gr.gx), (unsigned)((C + 63) / 64)), dim3(1024), 0,
                       st, part, (int)gr.gx, C, C, stage, ctr, F, G);
    Y5M_CHECK_LAUNCH("bn_reduce_finalize_kernel");
    const EwGeom ga = ew_geom(M, C / 8, 4096);
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false>), dim3(ga.gx, (unsigned)ga.groups), dim3(256), 0, st,
                                         (const T*)dz, lddz, (const T*)y, ldy, scale, shift, cB, cD, mean, (T*)dy, lddy, M, C, ga.CG,
                                         ga.RP, act, BnFusedBwd{});)
    Y5M_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return Y5M_OK;
}

// The same backward in TWO launches: the reduce pass adds its block partials into acc [y5m_bn_acc_slots()][2][C] f64
// (zeroed by the caller; left dirty), the apply pass derives its coefficients and dgamma / dbeta from them (y5m_bnfuse.h).
extern "C" int y5m_bn_bwd_fused_phase(const void* dz, int lddz, const void* y, int ldy, const float* scale, const float* shift,
                                      const float* mean, const float* invstd, int64_t M, int C, int act, float* dgamma,
                                      float* dbeta, int accumulate_param_grads, void* dy, int lddy, double* acc, int dtype,
                                      void* stream, int phase) {
    Y5M_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    Y5M_REQUIRE(acc && scale && shift && mean && invstd, "null pointer");
    Y5M_REQUIRE(phase >= 1 && phase <= 3, "phase: 1 = reduce, 2 = apply, 3 = both");
    hipStream_t st = y5m_stream(stream);
    if (phase & 1) {
        static int rgx = -1;                   // Y5M_BNR_GX: workgroups of the reduce pass (no partial rows to pay for here)
        if (rgx < 0) { const char* e = getenv("Y5M_BNR_GX"); rgx = e ? atoi(e) : BNR_MAX_GX; }
        const EwGeom grr = ew_geom(M, C / 8, rgx, BNR_THREADS);
        DISPATCH_T(dtype, {
            auto kern = !(y5m_r4_forms() & Y5M_R4_BN_BWD_REDUCE) ? bn_bwd_reduce_kernel<T, 0>
                        : act == Y5M_ACT_SILU ? bn_bwd_reduce_kernel<T, 2> : bn_bwd_reduce_kernel<T, 1>;
            hipLaunchKernelGGL(kern, dim3(grr.gx, (unsigned)grr.groups), dim3(BNR_THREADS), 0, st, (const T*)dz, lddz, (const T*)y,
                               ldy, scale, shift, mean, M, C, grr.CG, grr.RP, act, (float*)nullptr, acc);
        })
        Y5M_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    }
    if (phase & 2) {
        const BnFusedBwd G{acc, invstd, 1.0f / (float)M, dgamma, dbeta, accumulate_param_grads};
        const EwGeom ga = ew_geom(M, C / 8, BNF_EW_GX);
        DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), dim3(ga.gx, (unsigned)ga.groups), dim3(256), 0, st,
                                             (const T*)dz, lddz, (const T*)y, ldy, scale, shift, (const float*)nullptr,
                                             (const float*)nullptr, mean, (T*)dy, lddy, M, C, ga.CG, ga.RP, act, G);)
        Y5M_CHECK_LAUNCH("bn_bwd_apply_kernel");
    }
    return Y5M_OK;
}

extern "C" int y5m_bn_bwd_fused(const void* dz, int lddz, const void* y, int ldy, const float* scale, const float* shift,
                                const float* mean, const float* invstd, int64_t M, int C, int act, float* dgamma,
                                float* dbeta, int accumulate_param_grads, void* dy, int lddy, double* acc, int dtype,
                                void* stream) {
    return y5m_bn_bwd_fused_phase(dz, lddz, y, ldy, scale, shift, mean, invstd, M, C, act, dgamma, dbeta,
                                  accumulate_param_grads, dy, lddy, acc, dtype, stream, 3);
}


// =================================================================================================
// gradient plumbing: dst (+)= src over (ptr, ld) views
// =================================================================================================
template <typename T>
__global__ void add_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd, int64_t M, int C8,
                           int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * C8) return;
    const int64_t m = i / C8;
    const int c = (int)(i - m * C8) * 8;
    float v[8];
    load8<T>(src + m * lds_ + c, v);
    if (accumulate) {
        float o[8];
        load8<T>(dst + m * ldd + c, o);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += o[k];
    }
    store8<T>(dst + m * ldd + c, v);
}
extern "C" int y5m_add(const void* src, int ldsrc, void* dst, int lddst, int64_t M, int C, int accumulate, int dtype,
                       void* stream) {
    Y5M_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    const int64_t n = M * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL(add_kernel<T>, dim3(ew_blocks(n)), dim3(EW_T), 0, y5m_stream(stream), (const T*)src,
                                         ldsrc, (T*)dst, lddst, M, C / 8, accumulate);)
    Y5M_CHECK_LAUNCH("add_kernel");
    return Y5M_OK;
}

// =================================================================================================
// nearest 2x upsample (reference model.py:225) and its backward (2x2 sum)
// =================================================================================================
template <typename T>
__global__ void upsample2x_kernel(const T* __restrict__ in, int ldin, int B, int H, int W, int C8, T* __restrict__ out,
                                  int ldout) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)B * (2 * H) * (2 * W) * C8;
    if (i >= n) return;
    const int c = (int)(i % C8) * 8;
    int64_t t = i / C8;
    const int x = (int)(t % (2 * W)); t /= (2 * W);
    const int y = (int)(t % (2 * H));
    const int b = (int)(t / (2 * H));
    const uint4* s = reinterpret_cast<const uint4*>(in + (((size_t)b * H + (y >> 1)) * W + (x >> 1)) * ldin + c);
    uint4* d = reinterpret_cast<uint4*>(out + (((size_t)b * 2 * H + y) * 2 * W + x) * ldout + c);
    d[0] = s[0];
    if (sizeof(T) == 4) d[1] = s[1];
}
template <typename T>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ g, int ldg, int B, int H, int W, int C8, T* __restrict__ gin,
                                      int ldgin, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)B * H * W * C8;
    if (i >= n) return;
    const int c = (int)(i % C8) * 8;
    int64_t t = i / C8;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            float v[8];
            load8<T>(g + (((size_t)b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * ldg + c, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k];
        }
    T* d = gin + (((size_t)b * H + y) * W + x) * ldgin + c;
    if (accumulate) {
        float o[8];
        load8<T>(d, o);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += o[k];
    }
    store8<T>(d, acc);
}
extern "C" int y5m_upsample2x(const void* in, int ldin, int B, int H, int W, int C, void* out, int ldout, int dtype,
                              void* stream) {
    Y5M_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    const int64_t n = (int64_t)B * 4 * H * W * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL(upsample2x_kernel<T>, dim3(ew_blocks(n)), dim3(EW_T), 0, y5m_stream(stream),
                                         (const T*)in, ldin, B, H, W, C / 8, (T*)out, ldout);)
    Y5M_CHECK_LAUNCH("upsample2x_kernel");
    return Y5M_OK;
}
extern "C" int y5m_upsample2x_bwd(const void* gout, int ldg, int B, int H, int W, int C, void* gin, int ldgin,
                                  int accumulate, int dtype, void* stream) {
    Y5M_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    const int64_t n = (int64_t)B * H * W * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL(upsample2x_bwd_kernel<T>, dim3(ew_blocks(n)), dim3(EW_T), 0, y5m_stream(stream),
                                         (const T*)gout, ldg, B, H, W, C / 8, (T*)gin, ldgin, accumulate);)
    Y5M_CHECK_LAUNCH("upsample2x_bwd_kernel");
    return Y5M_OK;
}

// =================================================================================================
// SPPF pooling (reference model.py:103-112): three cascaded MaxPool2d(5,1,2) == windows 5, 9, 13 of x
// (padding is -inf, so max-of-max composes exactly). One launch writes the 3 concat slices.
// =================================================================================================
// Separable: a horizontal pass builds the row maxima for windows 5 / 9 / 13, a vertical pass finishes
// them; 8 channels (16 B) per thread and 13 + 27 vector loads per output instead of 169 scalar ones.
template <typename T>
__global__ void sppf_rowmax_kernel(const T* __restrict__ x, int ld, int B, int H, int W, int C8, T* __restrict__ h5,
                                   T* __restrict__ h9, T* __restrict__ h13) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)B * H * W * C8;
    if (i >= n) return;
    const int c = (int)(i % C8) * 8;
    int64_t t = i / C8;
    const int xx = (int)(t % W);
    const int64_t row = t / W;               // b*H + y
    float m5[8], m9[8], m13[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m5[k] = m9[k] = m13[k] = -INFINITY;
    for (int dx = -6; dx <= 6; ++dx) {
        const int x2 = xx + dx;
        if (x2 < 0 || x2 >= W) continue;
        float v[8];
        load8<T>(x + (row * W + x2) * ld + c, v);
        const int r = dx < 0 ? -dx : dx;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            m13[k] = fmaxf(m13[k], v[k]);
            if (r <= 4) m9[k] = fmaxf(m9[k], v[k]);
            if (r <= 2) m5[k] = fmaxf(m5[k], v[k]);
        }
    }
    const size_t o = (size_t)(row * W + xx) * (C8 * 8) + c;
    store8<T>(h5 + o, m5); store8<T>(h9 + o, m9); store8<T>(h13 + o, m13);
}
template <typename T>
__global__ void sppf_colmax_kernel(const T* __restrict__ h5, const T* __restrict__ h9, const T* __restrict__ h13, int B,
                                   int H, int W, int C8, T* __restrict__ o1, T* __restrict__ o2, T* __restrict__ o3, int ld) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)B * H * W * C8;
    if (i >= n) return;
    const int c = (int)(i % C8) * 8;
    int64_t t = i / C8;
    const int xx = (int)(t % W); t /= W;
    const int yy = (int)(t % H);
    const int b = (int)(t / H);
    const int Cc = C8 * 8;
    float m5[8], m9[8], m13[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m5[k] = m9[k] = m13[k] = -INFINITY;
    for (int dy = -6; dy <= 6; ++dy) {
        const int y2 = yy + dy;
        if (y2 < 0 || y2 >= H) continue;
        const size_t o = (((size_t)b * H + y2) * W + xx) * Cc + c;
        const int r = dy < 0 ? -dy : dy;
        float v[8];
        load8<T>(h13 + o, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) m13[k] = fmaxf(m13[k], v[k]);
        if (r <= 4) {
            load8<T>(h9 + o, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) m9[k] = fmaxf(m9[k], v[k]);
        }
        if (r <= 2) {
            load8<T>(h5 + o, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) m5[k] = fmaxf(m5[k], v[k]);
        }
    }
    const size_t o = (((size_t)b * H + yy) * W + xx) * ld + c;
    store8<T>(o1 + o, m5); store8<T>(o2 + o, m9); store8<T>(o3 + o, m13);
}
static int pool_tile_cs(int H, int W, int C, int dtype, int bytes_per_item_bf16, int bytes_per_item_f32);
extern "C" int y5m_sppf_pool_tiled(int H, int W, int C, int dtype);
template <typename T> __global__ void sppf_pool_tile_kernel(const T* __restrict__ x, int ld, int H, int W, int C8, int CS,
                                                            T* __restrict__ o1, T* __restrict__ o2, T* __restrict__ o3);
// once per kernel instantiation: raise its dynamic-LDS cap to the largest tile pool_tile_cs admits (150 KB); false (checked on
// every call, set never again) = the runtime refused, the caller runs the separable kernels instead
template <auto KERN> static bool pool_tile_ready() {
    static int ok = -1;
    if (ok < 0) ok = hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, 150 << 10) == hipSuccess ? 1 : 0;
    return ok == 1;
}
extern "C" size_t y5m_sppf_pool_workspace_bytes(int B, int H, int W, int C) { return (size_t)3 * B * H * W * C * 4 + 256; }
extern "C" int y5m_sppf_pool(const void* x, int ld, int B, int H, int W, int C, void* o1, void* o2, void* o3, void* ws,
                             size_t ws_bytes, int dtype, void* stream) {
    Y5M_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    if (ws_bytes < y5m_sppf_pool_workspace_bytes(B, H, W, C)) { y5m_set_error("sppf_pool ws too small"); return Y5M_EWS; }
    const int64_t n = (int64_t)B * H * W * (C / 8);
    const size_t plane = (size_t)B * H * W * C;
    hipStream_t st = y5m_stream(stream);
    if (y5m_sppf_pool_tiled(H, W, C, dtype)) {         // Y5M_POOL_TILE: one image x CS pieces per workgroup, everything in LDS
        const int cs = pool_tile_cs(H, W, C, dtype, 2 * 16, 2 * 32);
        const int slabs = (C / 8 + cs - 1) / cs;
        const size_t lds = (size_t)H * W * cs * (dtype == Y5M_BF16 ? 2 * 16 : 2 * 32);
        bool launched = false;
        DISPATCH_T(dtype, {
            if (pool_tile_ready<sppf_pool_tile_kernel<T>>()) {
                hipLaunchKernelGGL(sppf_pool_tile_kernel<T>, dim3((unsigned)(B * slabs)), dim3(256), lds, st, (const T*)x, ld, H, W, C / 8, cs,
                                   (T*)o1, (T*)o2, (T*)o3);
                launched = true;
            }
        })
        if (launched) {
            Y5M_CHECK_LAUNCH("sppf_pool_tile_kernel");
            return Y5M_OK;
        }
    }
    DISPATCH_T(dtype, T* h5 = (T*)ws; T* h9 = h5 + plane; T* h13 = h9 + plane;
               hipLaunchKernelGGL(sppf_rowmax_kernel<T>, dim3(ew_blocks(n)), dim3(EW_T), 0, st, (const T*)x, ld, B, H, W, C / 8, h5, h9, h13);
               hipLaunchKernelGGL(sppf_colmax_kernel<T>, dim3(ew_blocks(n)), dim3(EW_T), 0, st, (const T*)h5, (const T*)h9, (const T*)h13, B, H, W, C / 8, (T*)o1, (T*)o2, (T*)o3, ld);)
    Y5M_CHECK_LAUNCH("sppf_pool kernels");
    return Y5M_OK;
}

// backward of ONE MaxPool2d(5,1,2): gin[i] (+)= sum over outputs o whose window argmax is i of g[o].
// argmax = first maximum in (row, col) scan order, as ATen's max_pool2d_with_indices.
// Pass 1 stores, per output, the window-relative position (0..24) of its argmax; pass 2 gathers.
template <typename T>
__global__ void maxpool5_argmax_kernel(const T* __restrict__ z, int ldz, int B, int H, int W, int C8,
                                       unsigned char* __restrict__ code) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)B * H * W * C8;
    if (i >= n) return;
    const int c = (int)(i % C8) * 8;
    int64_t t = i / C8;
    const int ox = (int)(t % W); t /= W;
    const int oy = (int)(t % H);
    const int b = (int)(t / H);
    float best[8];
    int bc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bc[k] = -1; }
    for (int dy = 0; dy < 5; ++dy) {
        const int wy = oy + dy - 2;
        if (wy < 0 || wy >= H) continue;
        for (int dx = 0; dx < 5; ++dx) {
            const int wx = ox + dx - 2;
            if (wx < 0 || wx >= W) continue;
            float v[8];
            load8<T>(z + (((size_t)b * H + wy) * W + wx) * ldz + c, v);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (v[k] > best[k] || bc[k] < 0) { best[k] = v[k]; bc[k] = dy * 5 + dx; }
        }
    }
    uint2 pk;
    pk.x = (unsigned)bc[0] | ((unsigned)bc[1] << 8) | ((unsigned)bc[2] << 16) | ((unsigned)bc[3] << 24);
    pk.y = (unsigned)bc[4] | ((unsigned)bc[5] << 8) | ((unsigned)bc[6] << 16) | ((unsigned)bc[7] << 24);
    *reinterpret_cast<uint2*>(code + ((((size_t)b * H + oy) * W + ox) * (C8 * 8) + c)) = pk;
}
template <typename T>
__global__ void maxpool5_gather_kernel(const unsigned char* __restrict__ code, const T* __restrict__ g, int ldg, int B,
                                       int H, int W, int C8, T* __restrict__ gin, int ldgin, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)B * H * W * C8;
    if (i >= n) return;
    const int c = (int)(i % C8) * 8;
    int64_t t = i / C8;
    const int xx = (int)(t % W); t /= W;
    const int yy = (int)(t % H);
    const int b = (int)(t / H);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int dy = -2; dy <= 2; ++dy) {
        const int oy = yy + dy;
        if (oy < 0 || oy >= H) continue;
        for (int dx = -2; dx <= 2; ++dx) {
            const int ox = xx + dx;
            if (ox < 0 || ox >= W) continue;
            const size_t o = ((size_t)b * H + oy) * W + ox;
            const uint2 pk = *reinterpret_cast<const uint2*>(code + o * (C8 * 8) + c);
            const unsigned want = (unsigned)((2 - dy) * 5 + (2 - dx));   // (yy,xx) inside the window of (oy,ox)
            float gv[8];
            load8<T>(g + o * ldg + c, gv);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned cd = ((k < 4 ? pk.x : pk.y) >> (8 * (k & 3))) & 0xffu;
                if (cd == want) acc[k] += gv[k];
            }
        }
    }
    T* d = gin + (((size_t)b * H + yy) * W + xx) * ldgin + c;
    if (accumulate) {
        float o[8];
        load8<T>(d, o);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += o[k];
    }
    store8<T>(d, acc);
}
extern "C" size_t y5m_maxpool5_bwd_workspace_bytes(int B, int H, int W, int C) { return (size_t)B * H * W * C + 256; }
extern "C" int y5m_maxpool5_bwd(const void* z, int ldz, const void* g, int ldg, int B, int H, int W, int C, void* gin,
                                int ldgin, int accumulate, void* ws, size_t ws_bytes, int dtype, void* stream) {
    Y5M_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    const int64_t n = (int64_t)B * H * W * (C / 8);
    if (ws_bytes < y5m_maxpool5_bwd_workspace_bytes(B, H, W, C)) { y5m_set_error("maxpool5_bwd ws too small"); return Y5M_EWS; }
    unsigned char* code = reinterpret_cast<unsigned char*>(ws);
    hipStream_t st = y5m_stream(stream);
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool5_argmax_kernel<T>, dim3(ew_blocks(n)), dim3(EW_T), 0, st, (const T*)z, ldz, B,
                                         H, W, C / 8, code);)
    Y5M_CHECK_LAUNCH("maxpool5_argmax_kernel");
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool5_gather_kernel<T>, dim3(ew_blocks(n)), dim3(EW_T), 0, st, code, (const T*)g,
                                         ldg, B, H, W, C / 8, (T*)gin, ldgin, accumulate);)
    Y5M_CHECK_LAUNCH("maxpool5_gather_kernel");
    return Y5M_OK;
}

// =================================================================================================
// LDS-tiled forms of the SPPF pooling (round 5; Y5M_POOL_TILE, default 0: written without a GPU, A/B staged).
// Why: profiles/r05_step_bytes.txt -- the four pooling kernels move 1.55 GB per step for 0.32 GB of tensors (separable 13-tap
// passes through workspaces, an argmax code plane, 25-tap gathers: 446 us per step in round 3's kernel statistics against ~80 us
// of HBM time). A 20x20 (40x40 at 1280^2) image is small: one workgroup takes ONE IMAGE x a slab of CS 8-channel pieces into
// LDS and does everything there --
//   forward : x once in, the three cascaded 5x5 maxima (row pass / column pass, ping-pong between two LDS images) out: 4 tensor
//             passes instead of 2 launches + 3 workspace planes;
//   backward: the WHOLE cascade g2 += bwd(p2; g3), g1 += bwd(p1; g2), g0 += bwd(x; g1) in one launch: per level the pool input
//             and the incoming gradient sit in LDS, the argmax codes are formed there, the gather reads them there; the updated
//             gradient is stored (rounded to T exactly as the per-level launches store it) and kept in LDS as the next level's
//             incoming gradient: 10 tensor passes instead of 6 launches, 12 passes and a code plane.
// Same arithmetic in the same order as the kernels above (first maximum in row-major window order; gather sums in (dy, dx)
// order): bit-identical results, tests/test_gpu_model.py::test_sppf_pool_tiled_forms_equal_the_separable_ones.
// =================================================================================================
__device__ __forceinline__ void raw_pack8(const float v[8], Raw8<bf16_t>& r) {      // exact for values that ARE bf16; else RNE
#pragma unroll
    for (int i = 0; i < 4; ++i) r.q[i] = f32x2_to_bf16x2(v[2 * i], v[2 * i + 1]);
}
__device__ __forceinline__ void raw_pack8(const float v[8], Raw8<float>& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.a[i] = v[i]; r.b[i] = v[4 + i]; }
}

template <typename T>
__global__ __launch_bounds__(256) void sppf_pool_tile_kernel(const T* __restrict__ x, int ld, int H, int W, int C8, int CS,
                                                            T* __restrict__ o1, T* __restrict__ o2, T* __restrict__ o3) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int slabs = (C8 + CS - 1) / CS;
    const int b = blockIdx.x / slabs, slab = blockIdx.x - b * slabs;
    const int c80 = slab * CS, ncs = min(CS, C8 - c80);
    const int n = H * W * ncs;                         // items: i = pixel * ncs + piece
    Raw8<T>* A = reinterpret_cast<Raw8<T>*>(smem);
    Raw8<T>* Bf = A + H * W * CS;
    const size_t img = (size_t)b * H * W;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int pix = i / ncs, cs = i - pix * ncs;
        raw_load8(x + (img + pix) * ld + (c80 + cs) * 8, A[i]);
    }
    __syncthreads();
#pragma unroll 1
    for (int lvl = 0; lvl < 3; ++lvl) {
        T* const o = lvl == 0 ? o1 : (lvl == 1 ? o2 : o3);
        for (int i = threadIdx.x; i < n; i += 256) {   // row pass: A -> Bf
            const int pix = i / ncs, cs = i - pix * ncs;
            const int yy = pix / W, xx = pix - yy * W;
            float m[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
            for (int dx = -2; dx <= 2; ++dx) {
                const int x2 = xx + dx;
                if (x2 < 0 || x2 >= W) continue;
                float v[8];
                raw_unpack8(A[(yy * W + x2) * ncs + cs], v);
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], v[k]);
            }
            raw_pack8(m, Bf[i]);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {   // column pass: Bf -> A (= the pool's output: stored, and the next level's input)
            const int pix = i / ncs, cs = i - pix * ncs;
            const int yy = pix / W, xx = pix - yy * W;
            float m[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
            for (int dy = -2; dy <= 2; ++dy) {
                const int y2 = yy + dy;
                if (y2 < 0 || y2 >= H) continue;
                float v[8];
                raw_unpack8(Bf[(y2 * W + xx) * ncs + cs], v);
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], v[k]);
            }
            raw_pack8(m, A[i]);
            store8<T>(o + (img + pix) * ld + (c80 + cs) * 8, m);
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sppf_pool_bwd_tile_kernel(const T* __restrict__ z0, const T* __restrict__ z1,
                                                                const T* __restrict__ z2, int ldz, T* __restrict__ g0,
                                                                T* __restrict__ g1, T* __restrict__ g2, const T* __restrict__ g3,
                                                                int ldg, int H, int W, int C8, int CS) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int slabs = (C8 + CS - 1) / CS;
    const int b = blockIdx.x / slabs, slab = blockIdx.x - b * slabs;
    const int c80 = slab * CS, ncs = min(CS, C8 - c80);
    const int n = H * W * ncs, cap = H * W * CS;
    Raw8<T>* Z = reinterpret_cast<Raw8<T>*>(smem);
    Raw8<T>* Ga = Z + cap;                             // incoming gradient of the level being differentiated
    Raw8<T>* Gb = Ga + cap;                            // ... of the next one (written by this level's gather)
    uint2* code = reinterpret_cast<uint2*>(Gb + cap);
    const size_t img = (size_t)b * H * W;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int pix = i / ncs, cs = i - pix * ncs;
        raw_load8(g3 + (img + pix) * ldg + (c80 + cs) * 8, Ga[i]);
    }
#pragma unroll 1
    for (int lvl = 2; lvl >= 0; --lvl) {
        const T* const z = lvl == 2 ? z2 : (lvl == 1 ? z1 : z0);
        T* const gin = lvl == 2 ? g2 : (lvl == 1 ? g1 : g0);
        for (int i = threadIdx.x; i < n; i += 256) {
            const int pix = i / ncs, cs = i - pix * ncs;
            raw_load8(z + (img + pix) * ldz + (c80 + cs) * 8, Z[i]);
        }
        __syncthreads();                               // Z and Ga complete
        for (int i = threadIdx.x; i < n; i += 256) {   // argmax code of every OUTPUT pixel (maxpool5_argmax_kernel's scan)
            const int pix = i / ncs, cs = i - pix * ncs;
            const int oy = pix / W, ox = pix - oy * W;
            float best[8];
            int bc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bc[k] = -1; }
            for (int dy = 0; dy < 5; ++dy) {
                const int wy = oy + dy - 2;
                if (wy < 0 || wy >= H) continue;
                for (int dx = 0; dx < 5; ++dx) {
                    const int wx = ox + dx - 2;
                    if (wx < 0 || wx >= W) continue;
                    float v[8];
                    raw_unpack8(Z[(wy * W + wx) * ncs + cs], v);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (v[k] > best[k] || bc[k] < 0) { best[k] = v[k]; bc[k] = dy * 5 + dx; }
                }
            }
            uint2 pk;
            pk.x = (unsigned)bc[0] | ((unsigned)bc[1] << 8) | ((unsigned)bc[2] << 16) | ((unsigned)bc[3] << 24);
            pk.y = (unsigned)bc[4] | ((unsigned)bc[5] << 8) | ((unsigned)bc[6] << 16) | ((unsigned)bc[7] << 24);
            code[i] = pk;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {   // gather (maxpool5_gather_kernel's order), accumulate onto gin
            const int pix = i / ncs, cs = i - pix * ncs;
            const int yy = pix / W, xx = pix - yy * W;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int dy = -2; dy <= 2; ++dy) {
                const int oy = yy + dy;
                if (oy < 0 || oy >= H) continue;
                for (int dx = -2; dx <= 2; ++dx) {
                    const int ox = xx + dx;
                    if (ox < 0 || ox >= W) continue;
                    const int o = (oy * W + ox) * ncs + cs;
                    const uint2 pk = code[o];
                    const unsigned want = (unsigned)((2 - dy) * 5 + (2 - dx));
                    float gv[8];
                    raw_unpack8(Ga[o], gv);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const unsigned cd = ((k < 4 ? pk.x : pk.y) >> (8 * (k & 3))) & 0xffu;
                        if (cd == want) acc[k] += gv[k];
                    }
                }
            }
            T* d = gin + (img + pix) * ldg + (c80 + cs) * 8;
            float old[8];
            load8<T>(d, old);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += old[k];
            store8<T>(d, acc);
            raw_pack8(acc, Gb[i]);                     // the value as STORED (rounded to T): the next level's incoming gradient
        }
        __syncthreads();                               // every reader of Ga / Z / code is done before they are overwritten
        Raw8<T>* t = Ga; Ga = Gb; Gb = t;
    }
}

// Y5M_POOL_TILE (default 0): the tiled forms above where an image x >= 1 piece fits 64 KB of LDS; CS = pieces per workgroup
static int pool_tile_cs(int H, int W, int C, int dtype, int bytes_per_item_bf16, int bytes_per_item_f32) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("Y5M_POOL_TILE"); on = e ? atoi(e) : 0; }
    if (!on || C % 8 != 0 || H <= 0 || W <= 0) return 0;
    const size_t per = (size_t)H * W * (dtype == Y5M_BF16 ? bytes_per_item_bf16 : bytes_per_item_f32);
    int cs = (int)((64u << 10) / per);
    if (cs < 1) cs = per <= (150u << 10) ? 1 : 0;
    if (cs > C / 8) cs = C / 8;
    if (cs > 8) cs = 8;                                // (enough workgroups: B x C / 64 at least)
    return cs;
}
// 1 when y5m_sppf_pool / y5m_sppf_pool_bwd run their LDS-tiled forms for this shape (Engine.algorithmic_bytes asks)
extern "C" int y5m_sppf_pool_tiled(int H, int W, int C, int dtype) {
    return pool_tile_cs(H, W, C, dtype, 2 * 16, 2 * 32) > 0 && pool_tile_cs(H, W, C, dtype, 3 * 16 + 8, 3 * 32 + 8) > 0;
}

// the whole backward cascade of the SPPF pools (reference model.py:108-110): g2 += bwd(z2; g3), g1 += bwd(z1; g2), g0 += bwd(z0; g1)
// with z0 = x, z1 = pool(x), z2 = pool(pool(x)). One launch where the tiled form applies, else three y5m_maxpool5_bwd calls.
extern "C" int y5m_sppf_pool_bwd(const void* z0, const void* z1, const void* z2, int ldz, void* g0, void* g1, void* g2,
                                 const void* g3, int ldg, int B, int H, int W, int C, void* ws, size_t ws_bytes, int dtype,
                                 void* stream) {
    Y5M_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    Y5M_REQUIRE(dtype == Y5M_F32 || dtype == Y5M_BF16, "dtype");
    const int cs = y5m_sppf_pool_tiled(H, W, C, dtype) ? pool_tile_cs(H, W, C, dtype, 3 * 16 + 8, 3 * 32 + 8) : 0;
    if (cs > 0) {
        const int slabs = (C / 8 + cs - 1) / cs;
        const size_t lds = (size_t)H * W * cs * (dtype == Y5M_BF16 ? 3 * 16 + 8 : 3 * 32 + 8);
        hipStream_t st = y5m_stream(stream);
        bool launched = false;
        DISPATCH_T(dtype, {
            if (pool_tile_ready<sppf_pool_bwd_tile_kernel<T>>()) {
                hipLaunchKernelGGL(sppf_pool_bwd_tile_kernel<T>, dim3((unsigned)(B * slabs)), dim3(256), lds, st, (const T*)z0, (const T*)z1,
                                   (const T*)z2, ldz, (T*)g0, (T*)g1, (T*)g2, (const T*)g3, ldg, H, W, C / 8, cs);
                launched = true;
            }
        })
        if (launched) {
            Y5M_CHECK_LAUNCH("sppf_pool_bwd_tile_kernel");
            return Y5M_OK;
        }
    }
    const void* zs[3] = {z0, z1, z2};
    void* gs[4] = {g0, g1, g2, const_cast<void*>(g3)};
    for (int lvl = 2; lvl >= 0; --lvl) {
        const int rc = y5m_maxpool5_bwd(zs[lvl], ldz, gs[lvl + 1], ldg, B, H, W, C, gs[lvl], ldg, 1, ws, ws_bytes, dtype, stream);
        if (rc != Y5M_OK) return rc;
    }
    return Y5M_OK;
}

// =================================================================================================
// head gradient: d(loss)/d(logits) f32 (B,naxs,ny,nx,nch) -> dY [M=B*ny*nx][ldp] in compute dtype
// (channel n = a*nch + c, columns >= naxs*nch zero) + bias gradient (column sums)
// =================================================================================================
// One block walks whole pixel rows: thread = output column (channel n = a*nch + c). The three
// 85-float runs of a pixel are read coalesced, the packed row is written coalesced, and each thread
// keeps the column sum (= bias gradient) in a register: one atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void head_grad_pack_kernel(const float* __restrict__ dl, int B, int naxs, int64_t hw,
                                                            int nch, T* __restrict__ dyp, int ldp,
                                                            float* __restrict__ dbias) {
    const int col = threadIdx.x;
    const int64_t M = (int64_t)B * hw;
    const int N = naxs * nch;
    const int a = col / nch, c = col - a * nch;
    float acc = 0.f;
    if (col < ldp) {
        // 4 pixel rows per iteration: the loop is one 4-byte load per thread and row, i.e. latency-bound with a
        // single load in flight (1.5 TB/s); four independent loads per thread quadruple the bytes in flight
        const int64_t G = gridDim.x;
        int64_t m = blockIdx.x;
        for (; m + 3 * G < M; m += 4 * G) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (col < N) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t mm = m + u * G;
                    const int64_t b = mm / hw, pix = mm - b * hw;
                    v[u] = dl[((b * naxs + a) * hw + pix) * nch + c];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dyp[(m + u * G) * ldp + col] = from_f32<T>(v[u]);
                acc += v[u];
            }
        }
        for (; m < M; m += G) {
            float v = 0.f;
            if (col < N) {
                const int64_t b = m / hw, pix = m - b * hw;
                v = dl[((b * naxs + a) * hw + pix) * nch + c];
            }
            dyp[m * ldp + col] = from_f32<T>(v);
            acc += v;
        }
        if (dbias && col < N) atomicAdd(&dbias[col], acc);
    }
}
extern "C" int y5m_head_grad_pack(const float* dlogits, int B, int naxs, int ny, int nx, int nch, void* dyp, int ldp,
                                  float* dbias, int dtype, void* stream) {
    Y5M_REQUIRE(ldp <= 256 && ldp >= naxs * nch, "head dims: naxs*nch <= ldp <= 256");
    hipStream_t st = y5m_stream(stream);
    if (dbias && y5m_fill32(dbias, 0u, (size_t)naxs * nch, st) != Y5M_OK) return Y5M_ELAUNCH;
    const int64_t M = (int64_t)B * ny * nx;
    const unsigned grid = (unsigned)(M < 4096 ? M : 4096);
    DISPATCH_T(dtype, hipLaunchKernelGGL(head_grad_pack_kernel<T>, dim3(grid), dim3(256), 0, st, dlogits, B, naxs,
                                         (int64_t)ny * nx, nch, (T*)dyp, ldp, dbias);)
    Y5M_CHECK_LAUNCH("head_grad_pack_kernel");
    return Y5M_OK;
}

// Sparse-aware variant for a gradient produced by y5m_compute_loss[_sparse]: zero outside channel 4 of every cell and
// the rows of the cells a target row hit.
//   (1) head_grad_pack_obj_kernel: every packed row = zeros + the three objectness gradients (from the loss
//       workspace's compact plane). One wave owns 64 consecutive pixel rows: lane l fetches row l's three values
//       (coalesced), then the wave writes the rows one by one, ONE 8-byte (bf16) store per lane and row.
//   (2) head_grad_rows_kernel: one wave per target row j that owns its cell (owner[cell] == j: every hit cell has exactly
//       one such row): the 5+nc accumulated values of that cell's dense row overwrite the packed row's box / class
//       columns (channel 4 is already there) and go to the bias gradient.
// Reads ~1/85 of the tensor the dense kernel walks (0.48 -> ~0.1 ms per step at B=64).
template <typename T>
__global__ __launch_bounds__(256) void head_grad_pack_obj_kernel(const float* __restrict__ gobj, int B, int naxs, int64_t hw, int nch,
                                                                T* __restrict__ dyp, int ldp, float* __restrict__ dbias) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t M = (int64_t)B * hw;
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + wid) * 64;
    if (m0 >= M) return;
    const int64_t m = m0 + lane;
    const bool rv = m < M;
    float go[3] = {0.f, 0.f, 0.f};
    if (rv) {
        const int64_t b = m / hw, pix = m - b * hw;
#pragma unroll
        for (int a = 0; a < 3; ++a) go[a] = gobj[(b * naxs + a) * hw + pix];
    }
    // the three objectness columns inside the 4-column pieces: lane ca >> 2, slot ca & 3
    const int c0 = 4, c1 = nch + 4, c2 = 2 * nch + 4;
    const bool l0 = lane == (c0 >> 2), l1 = lane == (c1 >> 2), l2 = lane == (c2 >> 2);
    const bool writer = 4 * lane < ldp;
    const int nrow = (int)((M - m0) < 64 ? (M - m0) : 64);
    for (int r = 0; r < nrow; ++r) {
        const float g0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(go[0]), r));
        const float g1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(go[1]), r));
        const float g2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(go[2]), r));
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (l0 && k == (c0 & 3)) v[k] = g0;
            if (l1 && k == (c1 & 3)) v[k] = g1;
            if (l2 && k == (c2 & 3)) v[k] = g2;
        }
        if (writer) store4<T>(dyp + (m0 + r) * ldp + 4 * lane, v);
    }
    if (dbias) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float sacc = wave_sum(go[a]);
            if (lane == 0) atomicAdd(&dbias[a * nch + 4], sacc);
        }
    }
}
// The same rows written TWO at a time with 16-byte pieces (Y5M_HEAD_PACK16, default 0: round 5, written without a GPU). The kernel
// above writes one 512-byte row per store instruction in 8-byte pieces and takes 0.26 ms per step for 0.28 GB (4.6x its bandwidth
// floor, profiles/r05_step_bytes.txt): 8-byte accesses run at 0.54-0.70 of the 16-byte rate (MI355X_MICROARCH.md) and the loop issues
// 64 dependent readlane triples. Here lanes 0-31 own row r, lanes 32-63 row r + 1, 8 columns (16 bytes of bf16) per lane; the row's
// three objectness gradients come through one ds_bpermute each. Same values, same bias sums.
template <typename T>
__global__ __launch_bounds__(256) void head_grad_pack_obj16_kernel(const float* __restrict__ gobj, int B, int naxs, int64_t hw, int nch,
                                                                  T* __restrict__ dyp, int ldp, float* __restrict__ dbias) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t M = (int64_t)B * hw;
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + wid) * 64;
    if (m0 >= M) return;
    const int64_t m = m0 + lane;
    float go[3] = {0.f, 0.f, 0.f};
    if (m < M) {
        const int64_t b = m / hw, pix = m - b * hw;
#pragma unroll
        for (int a = 0; a < 3; ++a) go[a] = gobj[(b * naxs + a) * hw + pix];
    }
    const int half = lane >> 5, cl = lane & 31;           // which of the two rows, which 8-column piece
    const int c0 = 4, c1 = nch + 4, c2 = 2 * nch + 4;
    const bool l0 = cl == (c0 >> 3), l1 = cl == (c1 >> 3), l2 = cl == (c2 >> 3);
    const bool writer = 8 * cl < ldp;
    const int nrow = (int)((M - m0) < 64 ? (M - m0) : 64);
    for (int r = 0; r < nrow; r += 2) {
        const int rr = r + half;                           // this lane's row (rr < 64 always; rr >= nrow: nothing stored)
        const float g0 = __shfl(go[0], rr, 64), g1 = __shfl(go[1], rr, 64), g2 = __shfl(go[2], rr, 64);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (l0 && k == (c0 & 7)) v[k] = g0;
            if (l1 && k == (c1 & 7)) v[k] = g1;
            if (l2 && k == (c2 & 7)) v[k] = g2;
        }
        if (writer && rr < nrow) store8<T>(dyp + (m0 + rr) * ldp + 8 * cl, v);
    }
    if (dbias) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float sacc = wave_sum(go[a]);
            if (lane == 0) atomicAdd(&dbias[a * nch + 4], sacc);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void head_grad_rows_kernel(const float* __restrict__ dl, const int32_t* __restrict__ owner,
                                                            const int32_t* __restrict__ bagg, const int32_t* __restrict__ count,
                                                            int naxs, int ny, int nx, int nch, T* __restrict__ dyp, int ldp,
                                                            float* __restrict__ dbias) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= *count) return;
    const int4 q = reinterpret_cast<const int4*>(bagg)[j];              // (b, a, gj, gi)
    const int64_t hw = (int64_t)ny * nx, pix = (int64_t)q.z * nx + q.w;
    const int64_t cell = ((int64_t)q.x * naxs + q.y) * hw + pix;
    if (owner[cell] != j) return;                                       // another row of the same cell does it
    const float* row = dl + cell * nch;
    T* drow = dyp + ((int64_t)q.x * hw + pix) * ldp + q.y * nch;
    const float v0 = lane < nch ? row[lane] : 0.f;
    const float v1 = lane + 64 < nch ? row[lane + 64] : 0.f;
    if (lane < nch && lane != 4) {
        drow[lane] = from_f32<T>(v0);
        if (dbias) atomicAdd(&dbias[q.y * nch + lane], v0);
    }
    if (lane + 64 < nch) {
        drow[lane + 64] = from_f32<T>(v1);
        if (dbias) atomicAdd(&dbias[q.y * nch + lane + 64], v1);
    }
}
extern "C" int y5m_head_grad_pack_sparse(const float* dlogits, const int32_t* owner, const float* gobj, const int32_t* bagg,
                                         const int32_t* count, int cap, int B, int naxs, int ny, int nx, int nch, void* dyp,
                                         int ldp, float* dbias, int dtype, void* stream) {
    Y5M_REQUIRE(naxs == 3 && nch >= 5 && nch <= 128, "sparse head gradient: 3 anchors per scale, 5+nc <= 128");
    Y5M_REQUIRE(ldp <= 256 && ldp >= naxs * nch && ldp % 4 == 0, "head dims: naxs*nch <= ldp <= 256, ldp % 4 == 0");
    Y5M_REQUIRE(owner && gobj && bagg && count && cap >= 0, "owner / objectness-gradient / target-row tables missing");
    hipStream_t st = y5m_stream(stream);
    if (dbias && y5m_fill32(dbias, 0u, (size_t)naxs * nch, st) != Y5M_OK) return Y5M_ELAUNCH;
    const int64_t M = (int64_t)B * ny * nx;
    const unsigned grid = (unsigned)((M + 255) / 256);
    static int pack16 = -1;                            // Y5M_HEAD_PACK16 (default 0): two rows per store instruction, 16-byte pieces
    if (pack16 < 0) { const char* e = getenv("Y5M_HEAD_PACK16"); pack16 = e ? atoi(e) : 0; }
    if (pack16 && ldp % 8 == 0 && (reinterpret_cast<uintptr_t>(dyp) & 15) == 0) {
        DISPATCH_T(dtype, hipLaunchKernelGGL(head_grad_pack_obj16_kernel<T>, dim3(grid), dim3(256), 0, st, gobj, B, naxs,
                                             (int64_t)ny * nx, nch, (T*)dyp, ldp, dbias);)
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL(head_grad_pack_obj_kernel<T>, dim3(grid), dim3(256), 0, st, gobj, B, naxs,
                                             (int64_t)ny * nx, nch, (T*)dyp, ldp, dbias);)
    }
    Y5M_CHECK_LAUNCH("head_grad_pack_obj_kernel");
    if (cap > 0) {
        DISPATCH_T(dtype, hipLaunchKernelGGL(head_grad_rows_kernel<T>, dim3((unsigned)((cap + 3) / 4)), dim3(256), 0, st, dlogits,
                                             owner, bagg, count, naxs, ny, nx, nch, (T*)dyp, ldp, dbias);)
        Y5M_CHECK_LAUNCH("head_grad_rows_kernel");
    }
    return Y5M_OK;
}

// =================================================================================================
// optimizer: global-norm clip (max_norm) + Adam with L2 weight decay (reference train.py:61,
// utils/training_utils.py:116-122), one fused pass over the flat parameter buffer
// =================================================================================================
#define SQ_BLOCKS 1024
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
    __shared__ float sm[4];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 q = reinterpret_cast<const float4*>(g)[i];
        acc += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    }
    if (blockIdx.x == 0) for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += g[i] * g[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out) {
    __shared__ double sm[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += (double)part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)sqrt(sm[0] + sm[1] + sm[2] + sm[3]);   // total L2 norm
}
// omb1 / omb2 = 1 - beta, rounded ONCE from the double-precision difference as torch.optim.Adam does (python floats:
// 1 - 0.999 = 0.001 -> f32); 1.0f - 0.999f would be 1.3e-5 off, a systematic scale error of the second moment
__device__ __forceinline__ void adam_one(float& pw, float g, float& mi, float& vi, float clip, float lr_bc1, float b1, float b2,
                                         float omb1, float omb2, float eps, float wd, float bc2_sqrt) {
    float gi = g * clip;
    gi = gi + wd * pw;                            // Adam(weight_decay=): L2 added to the gradient
    mi = b1 * mi + omb1 * gi;
    vi = b2 * vi + omb2 * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pw = pw - lr_bc1 * (mi / denom);
}
// 4 parameters per thread (16-byte accesses on all 7 streams); the tail is handled by the last threads one by one
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, const float* __restrict__ gnorm, float max_norm, double lr, double b1d, double b2d,
                            float eps, float wd, const int32_t* __restrict__ d_step) {
    // the step counter lives on the device so a captured hipGraph replays with the right bias correction; the corrections
    // are evaluated in double by one thread per block, exactly the scalars torch.optim.Adam computes on the host
    // (step_size = lr / (1 - beta1^t), sqrt(1 - beta2^t)) and then rounds to f32
    __shared__ float s_bc[2];
    if (threadIdx.x == 0) {
        const double t = (double)d_step[0];
        s_bc[0] = (float)(lr / (1.0 - pow(b1d, t)));
        s_bc[1] = (float)sqrt(1.0 - pow(b2d, t));
    }
    __syncthreads();
    const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n4 = n >> 2;
    if (i4 > n4) return;
    const float lr_bc1 = s_bc[0], bc2_sqrt = s_bc[1];
    const float b1 = (float)b1d, b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
    float clip = 1.0f;
    if (max_norm > 0.f) {                         // torch.nn.utils.clip_grad_norm_: coef = max/(norm+1e-6), clamped to 1
        clip = max_norm / (gnorm[0] + 1e-6f);
        clip = clip > 1.0f ? 1.0f : clip;
    }
    if (i4 < n4) {
        float4 pw = reinterpret_cast<float4*>(p)[i4], mi = reinterpret_cast<float4*>(m)[i4], vi = reinterpret_cast<float4*>(v)[i4];
        const float4 gq = reinterpret_cast<const float4*>(g)[i4];
        adam_one(pw.x, gq.x, mi.x, vi.x, clip, lr_bc1, b1, b2, omb1, omb2, eps, wd, bc2_sqrt);
        adam_one(pw.y, gq.y, mi.y, vi.y, clip, lr_bc1, b1, b2, omb1, omb2, eps, wd, bc2_sqrt);
        adam_one(pw.z, gq.z, mi.z, vi.z, clip, lr_bc1, b1, b2, omb1, omb2, eps, wd, bc2_sqrt);
        adam_one(pw.w, gq.w, mi.w, vi.w, clip, lr_bc1, b1, b2, omb1, omb2, eps, wd, bc2_sqrt);
        reinterpret_cast<float4*>(p)[i4] = pw; reinterpret_cast<float4*>(m)[i4] = mi; reinterpret_cast<float4*>(v)[i4] = vi;
    } else {
        for (int64_t i = n4 << 2; i < n; ++i) {
            float pw = p[i], mi = m[i], vi = v[i];
            adam_one(pw, g[i], mi, vi, clip, lr_bc1, b1, b2, omb1, omb2, eps, wd, bc2_sqrt);
            p[i] = pw; m[i] = mi; v[i] = vi;
        }
    }
}
extern "C" size_t y5m_adam_workspace_bytes(void) { return (SQ_BLOCKS + 64) * sizeof(float); }
extern "C" int y5m_grad_norm(const float* g, int64_t n, float* norm_out, void* ws, size_t ws_bytes, void* stream) {
    if (ws_bytes < y5m_adam_workspace_bytes()) { y5m_set_error("adam ws too small"); return Y5M_EWS; }
    float* part = reinterpret_cast<float*>(ws);
    hipStream_t st = y5m_stream(stream);
    hipLaunchKernelGGL(sumsq_kernel, dim3(SQ_BLOCKS), dim3(256), 0, st, g, n, part);
    Y5M_CHECK_LAUNCH("sumsq_kernel");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, part, SQ_BLOCKS, norm_out);
    Y5M_CHECK_LAUNCH("sumsq_final_kernel");
    return Y5M_OK;
}
extern "C" int y5m_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* gnorm, float max_norm,
                             double lr, double beta1, double beta2, double eps, double weight_decay, const int32_t* d_step,
                             void* stream) {
    Y5M_REQUIRE(d_step != nullptr, "d_step (device int32, >= 1) is required");
    Y5M_REQUIRE(((uintptr_t)p & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)m & 15) == 0 && ((uintptr_t)v & 15) == 0,
                "adam buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks((n >> 2) + 1)), dim3(EW_T), 0, y5m_stream(stream), p, g, m, v, n, gnorm, max_norm, lr,
                       beta1, beta2, (float)eps, (float)weight_decay, d_step);
    Y5M_CHECK_LAUNCH("adam_kernel");
    return Y5M_OK;
}
