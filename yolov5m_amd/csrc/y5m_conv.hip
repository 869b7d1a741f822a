// Implicit-GEMM convolution kernel (forward conv, fused epilogues, and data-gradient) for gfx950.
// See y5m_conv.h for the tiling / layout description.
#include "y5m_conv.h"

#include <stddef.h>
#include <stdlib.h>
#include <string.h>

// DB = true : two LDS buffers, one barrier per K step (long K: MFMA-bound layers)
// DB = false: one LDS buffer, two barriers per K step, half the LDS -> more resident workgroups per CU
//             (short K, e.g. 1x1 convs with C <= 384: latency/HBM-bound, TLP hides the load latency)
// Channel permutation inside a wave's NF*16-channel tile: MFMA output row rho of fragment a holds local channel
// cv_pch(a, rho). Fragments are paired so that lane (pixel, fq) holds the 8 CONSECUTIVE channels
// p*32 + fq*8 + [0,8) of pair p = a >> 1: one 16-byte store per lane, 64 contiguous bytes per pixel and store
// instruction (natural order = 4 channels per lane, 32-byte runs: the L2 then sees twice the write requests; on
// the pointwise kernel the same change cut the kernel time by 10 %). An odd last fragment keeps the natural order.
// The permutation is applied where the weight rows are fetched (LDS row r holds channel n0 + perm(r)), so the
// LDS layout and the conflict-free fragment reads are untouched.
template <int NF>
__device__ __forceinline__ constexpr int cv_pch(int a, int rho) {
    return (2 * (a >> 1) + 1 < NF) ? (a >> 1) * 32 + (rho >> 2) * 8 + (a & 1) * 4 + (rho & 3) : (a >> 1) * 32 + rho;
}

// logical tile id of hardware workgroup `bid` (workgroup b runs on XCD b % 8): workgroups with consecutive logical
// ids run on ONE XCD, so tiles that share operands (same pixel tile, different channel tile / parity class) meet in
// that XCD's L2
__device__ __forceinline__ int xcd_logical_id(int bid, int nblk) {
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}

template <typename T, int WM, int WN, int MF, int NF, bool DB>
__device__ __forceinline__ void conv_igemm_body(const ConvParams& P, const int lid) {
    constexpr int THREADS = WM * WN * 64, RSTEP = THREADS / 8;   // RSTEP: tile rows staged per pass (8 chunks per row)
    constexpr int CH = ElemTraits<T>::CH, BK = ElemTraits<T>::BK;
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
    static_assert(BM % CV_BM == 0 && (WM * WN == 4 || WM * WN == 8), "tile config");
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int NA = BM * 8 / THREADS;                          // 16-byte chunks per thread (A)
    constexpr int NB = (BN * 8 + THREADS - 1) / THREADS;          // (B)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid % WM, wn = wid / WM;

    // XCD-aware tile order (xcd_logical_id): blocks that share a pixel tile (same tile_m, different tile_n) have
    // consecutive logical ids
    const int tile_n = lid % P.tiles_n, tile_m = lid / P.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const T* __restrict__ X = reinterpret_cast<const T*>(P.in);
    const T* __restrict__ W = reinterpret_cast<const T*>(P.w);

    // ---- per-thread staging state ------------------------------------------------------------
    // Global loads go through BUFFER resources (raw, stride 0): address = SGPR base + a 32-bit VGPR byte
    // offset (+ an SGPR offset for the weights' K position), and an offset >= num_records returns zeros in
    // hardware. Image-border taps, K padding and the pixel tail cost ONE select of the offset; there is no
    // 64-bit address arithmetic in the K loop and no data-dependent fix-up after the loads, so all NA+NB
    // loads of a K step stay in flight under the MFMAs of the previous one. (views are < 2 GiB: y5m_conv
    // splits larger batches into slabs)
    constexpr unsigned OOB = 0x80000000u;
    constexpr int ESZ = (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(X), 0, (unsigned)((size_t)P.B * P.Hin * P.Win * P.ldin * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(W), 0, (unsigned)((size_t)P.Np * P.Kp * ESZ), 0x00020000);
    const int q = tid & 7, r0 = tid >> 3;
    int iy0[NA], ix0[NA];
    unsigned pb[NA];                                  // byte offset of the pixel's row at tap offset (0, 0)
    // pointwise fast path (1x1, stride 1, no padding, same grid): input pixel == output pixel index,
    // no (b, y, x) decomposition (two integer divisions per row) needed
    const float rcpW = 1.0f / (float)P.Wg, rcpH = 1.0f / (float)P.Hg;
    const bool lin_in = P.th == 1 && P.tw == 1 && P.sy == 1 && P.sx == 1 && P.dh0 == 0 && P.dw0 == 0 &&
                        P.Hin == P.Hg && P.Win == P.Wg;
    const unsigned rowb = (unsigned)(P.ldin * ESZ);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + r0 + RSTEP * i;
        int pv = m, yv = 0, xv = 0;
        if (!lin_in) {
            int gx, t, gy, b;
            fast_divmod(m, P.Wg, rcpW, t, gx);        // rows past M decompose to garbage and are masked by iy0
            fast_divmod(t, P.Hg, rcpH, b, gy);
            yv = gy * P.sy;
            xv = gx * P.sx;
            pv = (b * P.Hin + yv) * P.Win + xv;
        }
        pb[i] = m < P.M ? (unsigned)pv * rowb : 0u;
        iy0[i] = m < P.M ? yv : -(1 << 28);
        ix0[i] = xv;
    }
    // K position of this thread's chunk: kk = kt*BK + q*CH -> (tap row ta, tap col tb, channel c)
    int c, ta, tb;
    {
        const int kk = q * CH;
        const int tap = kk / P.Cin;
        c = kk - tap * P.Cin;
        ta = tap / P.tw;
        tb = tap - ta * P.tw;
    }
    // weight rows r0 + RSTEP*i of this channel tile: loop-invariant offsets, the K position is an SGPR offset
    unsigned wb_off[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int r = r0 + RSTEP * i;
        const int rw = r / (NF * 16), rl = r - rw * (NF * 16);
        const int rp = r < BN ? rw * (NF * 16) + cv_pch<NF>(rl >> 4, rl & 15) : 0;      // LDS row r <- channel n0 + rp
        wb_off[i] = (unsigned)(((size_t)(n0 + rp) * P.Kp + q * CH) * ESZ);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[NA], rb[NB];
    auto load_tile = [&](int kt) __attribute__((always_inline)) {
        // K padding (ta >= th) is folded into the row offset: the bounds test then fails for every pixel,
        // so there is ONE per-lane condition and no separate (uniform) code path for the padding tile
        const int dh = P.dh0 + ta * P.dhs, dw = P.dw0 + tb * P.dws;
        const int dhb = ta < P.th ? dh : (1 << 24);
        // (dh*Win + dw)*ldin + c  in bytes; |dh*Win + dw| < 2^23 (checked at launch)
        const unsigned koff = (unsigned)((__mul24(dh * P.Win + dw, P.ldin) + c) * ESZ);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int iy = iy0[i] + dhb, ix = ix0[i] + dw;
            const bool v = (unsigned)iy < (unsigned)P.Hin && (unsigned)ix < (unsigned)P.Win;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, v ? pb[i] + koff : OOB, 0, 0);
        }
        const int ksoff = kt * (BK * ESZ);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, wb_off[i], ksoff, 0);
    };
    auto advance_k = [&]() __attribute__((always_inline)) {
        c += BK;
        while (c >= P.Cin) {
            c -= P.Cin;
            if (++tb == P.tw) { tb = 0; ++ta; }
        }
    };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
        unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
        unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(As + lds_off(r0 + RSTEP * i, q)) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (r0 + RSTEP * i < BN) *reinterpret_cast<u32x4*>(Bs + lds_off(r0 + RSTEP * i, q)) = rb[i];
    };

    f32x4 acc[NF][MF];
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int b = 0; b < MF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 xa[MF], wb[NF];
#pragma unroll
            for (int b = 0; b < MF; ++b)
                xa[b] = *reinterpret_cast<const uint4*>(As + lds_off(wm * MF * 16 + b * 16 + frow, 4 * ks + fq));
#pragma unroll
            for (int a = 0; a < NF; ++a)
                wb[a] = *reinterpret_cast<const uint4*>(Bs + lds_off(wn * NF * 16 + a * 16 + frow, 4 * ks + fq));
#pragma unroll
            for (int a = 0; a < NF; ++a)
#pragma unroll
                for (int b = 0; b < MF; ++b) {
                    if constexpr (sizeof(T) == 2) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8_t, wb[a]), __builtin_bit_cast(bf16x8_t, xa[b]), acc[a][b], 0, 0, 0);
                    } else {
                        // 4 floats of a chunk feed 4 MFMAs; A and B use the same k permutation
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wb[a].x), __uint_as_float(xa[b].x), acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wb[a].y), __uint_as_float(xa[b].y), acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wb[a].z), __uint_as_float(xa[b].z), acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wb[a].w), __uint_as_float(xa[b].w), acc[a][b], 0, 0, 0);
                    }
                }
        }
    };

    // ---- main loop: register prefetch of tile kt+1 under the MFMAs of tile kt ------------------
    const int nkt = P.Kp / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    if constexpr (DB) {
        int cur = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            const bool more = kt + 1 < nkt;
            if (more) { advance_k(); load_tile(kt + 1); }
            compute(cur);
            if (more) store_tile(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    } else {
        for (int kt = 0; kt < nkt; ++kt) {
            const bool more = kt + 1 < nkt;
            if (more) { advance_k(); load_tile(kt + 1); }
            compute(0);
            __syncthreads();
            if (more) { store_tile(0); __syncthreads(); }
        }
    }

    // ---- epilogue --------------------------------------------------------------------------------
    // lane owns channels n = nb + cv_pch(a, (lane>>4)*4) + {0..3} of pixel m = mb + b*16 + (lane&15)
    const int nb = n0 + wn * NF * 16;
    const int mb = m0 + wm * MF * 16 + frow;

    if (P.epi == EPI_RAW_STATS && (P.stats || P.bn_acc)) {
        float* red = reinterpret_cast<float*>(smem);          // [2][WM][BN]; tiles are dead after the last barrier
#pragma unroll
        for (int a = 0; a < NF; ++a) {
            float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < MF; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float v = acc[a][b][r]; s[r] += v; ss[r] += v * v; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s[r] += __shfl_xor(s[r], o, 64); ss[r] += __shfl_xor(ss[r], o, 64); }
            }
            if (frow == 0) {
                const int nl = wn * NF * 16 + cv_pch<NF>(a, fq * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[(0 * WM + wm) * BN + nl + r] = s[r];
                    red[(1 * WM + wm) * BN + nl + r] = ss[r];
                }
            }
        }
        __syncthreads();
        for (int tt = tid; tt < 2 * BN; tt += THREADS) {
            const int which = tt / BN, nl = tt - which * BN;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += red[(which * WM + w) * BN + nl];
            if (P.bn_acc) bnf_add(P.bn_acc, P.Np, tile_m, which, n0 + nl, t);      // accumulator rows (y5m_bnfuse.h)
            else P.stats[((size_t)tile_m * 2 + which) * P.Np + n0 + nl] = t;
        }
    }

    // folded BatchNorm of the lane's channels (EPI_AFFINE_ACT): loaded once per tile, not once per fragment and pixel row
    float4 scv[NF], shv[NF];
    const bool opnd = (P.epi == EPI_AFFINE_ACT && P.res != nullptr) || (P.epi == EPI_DGRAD && P.accumulate);
    if (P.epi == EPI_AFFINE_ACT) {
#pragma unroll
        for (int a = 0; a < NF; ++a) {
            const int n = nb + cv_pch<NF>(a, fq * 4);
            const bool ok = n < P.N;
            scv[a] = ok ? *reinterpret_cast<const float4*>(P.scale + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            shv[a] = ok ? *reinterpret_cast<const float4*>(P.shift + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // dense output (no strided scatter): output pixel index == m, skip the decomposition
    const bool lin_out = P.epi != EPI_HEAD && P.osy == 1 && P.osx == 1 && P.ooy == 0 && P.oox == 0 &&
                         P.Hout == P.Hg && P.Wout == P.Wg;
#pragma unroll
    for (int b = 0; b < MF; ++b) {
        const int m = mb + b * 16;
        if (m >= P.M) continue;
        int gx = 0, gy = 0, bi = 0;
        size_t opix = (size_t)m;
        if (!lin_out) {
            int t;
            fast_divmod(m, P.Wg, rcpW, t, gx);
            fast_divmod(t, P.Hg, rcpH, bi, gy);
            opix = ((size_t)bi * P.Hout + (gy * P.osy + P.ooy)) * P.Wout + (gx * P.osx + P.oox);
        }
        float fv[NF][4];                 // finished values of the fragments that go to the dense / scattered T output
        unsigned okm = 0u;
        // the pixel's residual (EPI_AFFINE_ACT) / read-modify-write operand (EPI_DGRAD accumulate: out = conv + (res ? res : out),
        // `res` lets the first accumulation read ANOTHER tensor of the output's shape -- the bottleneck's residual gradient -- instead
        // of a copy made beforehand): ALL fragments requested before the first is used, not one dependent round trip per fragment
        float ov[NF][4];
        if (opnd) {
            const T* src = P.res ? reinterpret_cast<const T*>(P.res) + opix * P.ldres : reinterpret_cast<const T*>(P.out) + opix * P.ldout;
#pragma unroll
            for (int a = 0; a < NF; ++a) {
                const int n = nb + cv_pch<NF>(a, fq * 4);
                if (n < P.N) load4<T>(src + n, ov[a]);
            }
        }
#pragma unroll
        for (int a = 0; a < NF; ++a) {
            const int n = nb + cv_pch<NF>(a, fq * 4);
            if (n >= P.N) continue;
            float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
            if (P.epi == EPI_HEAD) {
                float* o = reinterpret_cast<float*>(P.out);
                const int an0 = n / P.nch, cn0 = n - an0 * P.nch;
                if (n + 3 < P.N && cn0 + 3 < P.nch) {
                    // the lane's 4 channels belong to one anchor: 4 consecutive floats of the permuted output
                    // (only dword aligned, 5+nc is odd) -> one 16-byte store instead of four scalar ones
                    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                    const f32x4u bq = *reinterpret_cast<const f32x4u*>(P.scale + n);      // (the bias is a slice of the flat parameter buffer)
                    f32x4u w = {v[0] + bq[0], v[1] + bq[1], v[2] + bq[2], v[3] + bq[3]};
                    *reinterpret_cast<f32x4u*>(o + ((((size_t)bi * P.naxs + an0) * P.Hg + gy) * P.Wg + gx) * P.nch + cn0) = w;
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nn = n + r;
                    if (nn < P.N) {
                        const int an = nn / P.nch, cn = nn - an * P.nch;
                        o[((((size_t)bi * P.naxs + an) * P.Hg + gy) * P.Wg + gx) * P.nch + cn] = v[r] + P.scale[nn];
                    }
                }
                continue;
            }
            if (P.epi == EPI_AFFINE_ACT) {
                v[0] = v[0] * scv[a].x + shv[a].x; v[1] = v[1] * scv[a].y + shv[a].y;
                v[2] = v[2] * scv[a].z + shv[a].z; v[3] = v[3] * scv[a].w + shv[a].w;
                if (P.act == Y5M_ACT_SILU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
                }
            }
            if (opnd) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += ov[a][r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) fv[a][r] = v[r];
            okm |= 1u << a;
        }
        // stores: a complete fragment pair is 8 consecutive channels of the lane -> one 16-byte (bf16) piece
        if (P.epi != EPI_HEAD) {
            T* const ob = reinterpret_cast<T*>(P.out) + opix * P.ldout + nb;
#pragma unroll
            for (int a = 0; a < NF; a += 2) {
                const bool ok0 = (okm >> a) & 1u, ok1 = a + 1 < NF && ((okm >> (a + 1)) & 1u);
                if (ok0 && ok1) {
                    if constexpr (sizeof(T) == 2) {
                        typedef unsigned u32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
                        u32x4a8 q4;
                        q4[0] = f32x2_to_bf16x2(fv[a][0], fv[a][1]);
                        q4[1] = f32x2_to_bf16x2(fv[a][2], fv[a][3]);
                        q4[2] = f32x2_to_bf16x2(fv[a + 1][0], fv[a + 1][1]);
                        q4[3] = f32x2_to_bf16x2(fv[a + 1][2], fv[a + 1][3]);
                        *reinterpret_cast<u32x4a8*>(ob + cv_pch<NF>(a, fq * 4)) = q4;
                    } else {
                        store4<T>(ob + cv_pch<NF>(a, fq * 4), fv[a]);
                        store4<T>(ob + cv_pch<NF>(a + 1, fq * 4), fv[a + 1]);
                    }
                } else {
                    if (ok0) store4<T>(ob + cv_pch<NF>(a, fq * 4), fv[a]);
                    if (ok1) store4<T>(ob + cv_pch<NF>(a + 1, fq * 4), fv[a + 1]);
                }
            }
        }
    }
}

template <typename T, int WM, int WN, int MF, int NF, bool DB>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm_kernel(const ConvParams P) {
    conv_igemm_body<T, WM, WN, MF, NF, DB>(P, xcd_logical_id(blockIdx.x, gridDim.x));
}

// Up to 4 problems of the same tiling in ONE launch (the parity classes of a stride-2 data gradient: same dY, same
// pixel grid, 1 / 2 / 2 / 4 taps). Logical id -> (tile, class) with the class fastest: the classes of a tile run on
// one XCD at the same time, so dY comes from HBM once instead of once per class (separate launches of 80-315 MB
// tensors do not meet in a 4 MB L2).
struct ConvMulti { ConvParams p[4]; int n; };        // p MUST stay the first member (see conv_igemm_multi_kernel)
static_assert(offsetof(ConvMulti, p) == 0, "kernarg layout");
template <typename T, int WM, int WN, int MF, int NF>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm_multi_kernel(const ConvMulti MP) {
    const int lid = xcd_logical_id(blockIdx.x, gridDim.x);
    const int cls = lid % MP.n;
    // the class's parameter block straight from the kernarg segment (MP is its first and only explicit argument): a
    // dynamically indexed by-value struct would be copied to scratch, this is a uniform scalar load
    typedef const __attribute__((address_space(4))) unsigned* kernarg_words_t;
    static_assert(sizeof(ConvParams) % 4 == 0, "dword copy");
    constexpr int NW = (int)(sizeof(ConvParams) / 4);
    const kernarg_words_t src = (kernarg_words_t)__builtin_amdgcn_kernarg_segment_ptr() + cls * NW;
    ConvParams P;
    unsigned* dst = reinterpret_cast<unsigned*>(&P);
#pragma unroll
    for (int i = 0; i < NW; ++i) dst[i] = src[i];
    conv_igemm_body<T, WM, WN, MF, NF, false>(P, lid / MP.n);
}

template <typename T, int WM, int WN, int MF, int NF>
static int launch_conv_multi(ConvMulti& MP, hipStream_t st) {
    constexpr int BN = WN * NF * 16, BM = WM * MF * 16;
    int tiles = 0;
    for (int i = 0; i < MP.n; ++i) {
        MP.p[i].tiles_m = (MP.p[i].M + BM - 1) / BM;
        MP.p[i].tiles_n = (MP.p[i].N + BN - 1) / BN;
        tiles = MP.p[i].tiles_m * MP.p[i].tiles_n;
    }
    const size_t lds = (size_t)(BM + BN) * 128;
    auto kern = conv_igemm_multi_kernel<T, WM, WN, MF, NF>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    Y5M_NAME_ONLY(Y5M_OK, "conv_igemm_multi_kernel<%s,%d,%d,%d,%d>", sizeof(T) == 2 ? "bf16" : "f32", WM, WN, MF, NF);
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * MP.n)), dim3(WM * WN * 64), lds, st, MP);
    Y5M_CHECK_LAUNCH("conv_igemm_multi_kernel");
    return Y5M_OK;
}

template <typename T, int WM, int WN, int MF, int NF, bool DB>
static int launch_conv_db(ConvParams& P, hipStream_t st) {
    constexpr int BN = WN * NF * 16, BM = WM * MF * 16;
    P.tiles_m = (P.M + BM - 1) / BM;
    P.tiles_n = (P.N + BN - 1) / BN;
    const size_t lds = (DB ? 2 : 1) * (size_t)(BM + BN) * 128;
    auto kern = conv_igemm_kernel<T, WM, WN, MF, NF, DB>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    Y5M_NAME_ONLY(Y5M_OK, "conv_igemm_kernel<%s,%d,%d,%d,%d,%d>", sizeof(T) == 2 ? "bf16" : "f32", WM, WN, MF, NF, (int)DB);
    hipLaunchKernelGGL(kern, dim3((unsigned)(P.tiles_m * P.tiles_n)), dim3(WM * WN * 64), lds, st, P);
    Y5M_CHECK_LAUNCH("conv_igemm_kernel");
    return Y5M_OK;
}

// Channel tile: 48 for the 48-channel layers, 192 (wave tile 64x96: half the LDS bytes per MFMA of the
// 96 tile) when N is a multiple of 192, else 96. Y5M_CONV_BN192=0 disables the wide tile (A/B runs).
static int g_bn192 = -1;
static int g_w8 = -1;      // Y5M_CONV_W8: 8-wave workgroups (0 off, 1 the 192-channel tile, 2 also the 96-channel tile)
static int g_sbuf_kt = -1;   // K steps up to which the single-buffer variant is used (Y5M_CONV_SBUF_KT, default: always)
template <typename T, int WM, int WN, int MF, int NF>
static int launch_conv(ConvParams& P, hipStream_t st) {
    // measured with the 8-wave tile: the single-buffer variant (half the LDS, more resident workgroups) wins for
    // EVERY K, not just short ones (32.8 -> 32.3 ms/step); the double-buffered variant stays as an A/B knob
    if (g_sbuf_kt < 0) { const char* e = getenv("Y5M_CONV_SBUF_KT"); g_sbuf_kt = e ? atoi(e) : 1 << 20; }
    const int BK = sizeof(T) == 2 ? 64 : 32;
    if (P.Kp / BK <= g_sbuf_kt) return launch_conv_db<T, WM, WN, MF, NF, false>(P, st);
    return launch_conv_db<T, WM, WN, MF, NF, true>(P, st);
}

static int name_of(int rc, char* buf, int n) {
    y5m_name_only = 0;
    if (rc != Y5M_OK) return rc;
    snprintf(buf, (size_t)n, "%s", y5m_name_buf);
    return Y5M_OK;
}
extern "C" int y5m_conv_kernel_name(const y5m_conv_args* args, int dtype, char* buf, int n) {
    y5m_name_only = 1;
    y5m_name_buf[0] = 0;
    return name_of(y5m_conv(args, dtype, nullptr), buf, n);
}
extern "C" int y5m_conv_multi_kernel_name(const y5m_conv_args* args, int cnt, int dtype, char* buf, int n) {
    y5m_name_only = 1;
    y5m_name_buf[0] = 0;
    return name_of(y5m_conv_multi(args, cnt, dtype, nullptr), buf, n);
}

extern "C" int y5m_conv_tile_n(int N) {
    if (g_bn192 < 0) { const char* e = getenv("Y5M_CONV_BN192"); g_bn192 = (e && e[0] == '0') ? 0 : 1; }
    if (N <= 48) return 48;
    if (g_bn192 && N % 192 == 0) return 192;
    return 96;
}

int y5m_conv_pw_try(const ConvParams& P, int dtype, hipStream_t st);      // y5m_conv_pw.hip
int y5m_conv_halo_try(const ConvParams& P, int dtype, hipStream_t st);    // y5m_conv_halo.hip
int y5m_conv_gemm8_try(const ConvParams& P, int dtype, hipStream_t st);   // y5m_conv_gemm.hip

static int conv_dispatch(ConvParams& P, int dtype, hipStream_t st) {
    {
        // 3x3 stride-1 layers with >= 64 input channels: persistent halo-patch kernel (y5m_conv_halo.hip)
        const int r = y5m_conv_halo_try(P, dtype, st);
        if (r != 0) return r < 0 ? r : Y5M_OK;
    }
    {
        // 1x1 layers with >= 384 input channels and N % 192 == 0: persistent two-phase GEMM kernel (y5m_conv_gemm.hip)
        const int r = y5m_conv_gemm8_try(P, dtype, st);
        if (r != 0) return r < 0 ? r : Y5M_OK;
    }
    {
        // short-K pointwise layers stream through the barrier-free kernel (y5m_conv_pw.hip)
        const int r = y5m_conv_pw_try(P, dtype, st);
        if (r != 0) return r < 0 ? r : Y5M_OK;
    }
    int BN = y5m_conv_tile_n(P.N);
    if (dtype == Y5M_BF16) {
        if (BN == 48) return launch_conv<bf16_t, 4, 1, 2, 3>(P, st);
        if (BN == 192) {
            // Y5M_CONV_W8=1: the same 128x192 tile with 8 waves (64x48 wave tiles): twice the waves per SIMD to
            // hide barrier / LDS / load waits, 40 % more LDS reads per MFMA (A/B knob)
            if (g_w8 < 0) { const char* e = getenv("Y5M_CONV_W8"); g_w8 = e ? atoi(e) : 1; }
            if (g_w8) return launch_conv<bf16_t, 2, 4, 4, 3>(P, st);
            return launch_conv<bf16_t, 2, 2, 4, 6>(P, st);
        }
        if (g_w8 < 0) { const char* e = getenv("Y5M_CONV_W8"); g_w8 = e ? atoi(e) : 1; }
        if (g_w8 >= 2) return launch_conv<bf16_t, 4, 2, 2, 3>(P, st);      // 128x96 tile, 8 waves of 32x48
        return launch_conv<bf16_t, 2, 2, 4, 3>(P, st);
    } else {
        if (BN == 48) return launch_conv<float, 4, 1, 2, 3>(P, st);
        if (BN == 192) return launch_conv<float, 2, 2, 4, 6>(P, st);
        return launch_conv<float, 2, 2, 4, 3>(P, st);
    }
}

extern "C" int y5m_conv(const y5m_conv_args* args, int dtype, void* stream) {
    ConvParams P;
    static_assert(sizeof(P) == sizeof(*args), "abi struct");
    memcpy(&P, args, sizeof(P));
    const int CH = dtype == Y5M_BF16 ? 8 : 4, BK = dtype == Y5M_BF16 ? 64 : 32;
    Y5M_REQUIRE(dtype == Y5M_F32 || dtype == Y5M_BF16, "dtype");
    Y5M_REQUIRE(P.zeros != nullptr, "args.zeros (16 zero bytes in device memory) is required");
    Y5M_REQUIRE(P.Cin % CH == 0 && P.ldin % CH == 0, "Cin/ldin must be multiples of the 16-byte chunk");
    Y5M_REQUIRE(P.Kp % BK == 0 && P.Kp >= P.K && P.K == P.th * P.tw * P.Cin, "K padding");
    Y5M_REQUIRE(P.M == P.B * P.Hg * P.Wg && P.M > 0, "M");
    Y5M_REQUIRE(P.epi == EPI_HEAD || (P.N % 4 == 0 && P.ldout % 4 == 0), "N/ldout must be multiples of 4");
    const int BN = y5m_conv_tile_n(P.N);
    Y5M_REQUIRE(P.epi != EPI_RAW_STATS || !(P.stats || P.bn_acc) || P.Np >= (P.N + BN - 1) / BN * BN, "stats stride Np too small");
    Y5M_REQUIRE(P.Np >= (P.N + BN - 1) / BN * BN, "Np (rows of the packed weights) must cover the channel tiles");
    Y5M_REQUIRE(!P.bn_acc || P.epi == EPI_RAW_STATS, "bn_acc: EPI_RAW_STATS launches only");
    hipStream_t st = y5m_stream(stream);
    // The kernels address the input view with 32-bit byte offsets (buffer resources; the top bit marks
    // "out of range"), so one launch sees at most 2 GiB of input: larger batches go in slabs of whole images.
    const size_t esz = dtype == Y5M_BF16 ? 2 : 4;
    const size_t img_in = (size_t)P.Hin * P.Win * P.ldin * esz;
    Y5M_REQUIRE(img_in < (1ull << 31) && (size_t)P.Hin * P.Win < (1ull << 22), "one input image must be < 2 GiB and < 2^22 pixels");
    // (Y5M_CONV_SLAB_BYTES, read once: a smaller limit, so that tests reach the slab path without a 2 GiB tensor)
    static size_t slab_bytes = 0;
    if (!slab_bytes) {
        const char* e = getenv("Y5M_CONV_SLAB_BYTES");
        const unsigned long long v = e ? strtoull(e, nullptr, 10) : 0ull;
        slab_bytes = (v >= 4096 && v < (1ull << 31)) ? (size_t)v : (size_t)((1ull << 31) - 1);
    }
    int per_slab = (int)(slab_bytes / img_in);
    if (per_slab < 1) per_slab = 1;                   // (only with the test limit: one image is < 2 GiB, checked above)
    if (P.B <= per_slab) return conv_dispatch(P, dtype, st);
    // (accumulator rows -- bn_acc -- simply keep adding across the slabs; partial rows are indexed per launch)
    Y5M_REQUIRE(P.epi != EPI_RAW_STATS || !P.stats, "training-mode conv with partial rows (stats): input view must be < 2 GiB");
    for (int b0 = 0; b0 < P.B; b0 += per_slab) {
        ConvParams S = P;
        S.B = P.B - b0 < per_slab ? P.B - b0 : per_slab;
        S.M = S.B * P.Hg * P.Wg;
        S.in = (const char*)P.in + (size_t)b0 * img_in;
        if (P.epi == EPI_HEAD) S.out = (char*)P.out + (size_t)b0 * P.naxs * P.Hg * P.Wg * P.nch * sizeof(float);
        else S.out = (char*)P.out + (size_t)b0 * P.Hout * P.Wout * P.ldout * esz;
        if (P.res) S.res = (const char*)P.res + (size_t)b0 * P.Hout * P.Wout * P.ldres * esz;
        const int r = conv_dispatch(S, dtype, st);
        if (r != Y5M_OK) return r;
    }
    return Y5M_OK;
}

// n <= 4 data-gradient problems (EPI_DGRAD) over the same pixel grid and output width -- the parity classes of a
// stride-2 convolution's data gradient -- as ONE launch of the tiled kernel (conv_igemm_multi_kernel). Anything that
// does not fit that pattern falls back to n separate y5m_conv calls, so callers may always use this entry point.
extern "C" int y5m_conv_multi(const y5m_conv_args* args, int n, int dtype, void* stream) {
    Y5M_REQUIRE(args && n >= 1, "args");
    static int on = -1;                              // Y5M_CONV_MULTI=0: always separate launches (A/B runs)
    if (on < 0) { const char* e = getenv("Y5M_CONV_MULTI"); on = (e && e[0] == '0') ? 0 : 1; }
    bool ok = on && n >= 2 && n <= 4 && dtype == Y5M_BF16;
    for (int i = 0; ok && i < n; ++i) {
        const y5m_conv_args& a = args[i];
        ok = a.epi == EPI_DGRAD && a.M == args[0].M && a.N == args[0].N && a.B == args[0].B &&
             a.Hg == args[0].Hg && a.Wg == args[0].Wg && a.in == args[0].in && a.ldin == args[0].ldin &&
             a.Kp % 64 == 0 && a.Kp >= a.K && a.K == a.th * a.tw * a.Cin && a.Cin % 8 == 0 && a.ldin % 8 == 0 &&
             a.N % 4 == 0 && a.ldout % 4 == 0 && a.zeros != nullptr && a.M == a.B * a.Hg * a.Wg && a.M > 0 &&
             (size_t)a.B * a.Hin * a.Win * a.ldin * 2 < (1ull << 31) && (size_t)a.Hin * a.Win < (1ull << 22);
        if (ok) {
            const int BNt = y5m_conv_tile_n(a.N);
            ok = a.Np >= (a.N + BNt - 1) / BNt * BNt;
        }
    }
    if (!ok) {
        for (int i = 0; i < n; ++i) {
            const int r = y5m_conv(&args[i], dtype, stream);
            if (r != Y5M_OK) return r;
        }
        return Y5M_OK;
    }
    ConvMulti MP;
    memset(&MP, 0, sizeof(MP));
    for (int i = 0; i < n; ++i) memcpy(&MP.p[i], &args[i], sizeof(ConvParams));
    MP.n = n;
    hipStream_t st = y5m_stream(stream);
    const int BN = y5m_conv_tile_n(args[0].N);
    if (BN == 48) return launch_conv_multi<bf16_t, 4, 1, 2, 3>(MP, st);
    if (BN == 192) return launch_conv_multi<bf16_t, 2, 4, 4, 3>(MP, st);
    return launch_conv_multi<bf16_t, 2, 2, 4, 3>(MP, st);
}
