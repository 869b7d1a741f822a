// Weight-gradient kernel for gfx950:  dW[n][tap][c] += sum_m dY[m][n] * X[pix(m,tap)][c]
//
// GEMM with K = pixels. Both operands are pixel-major in HBM (NHWC), i.e. K is the STRIDED axis, so
// the tiles are staged [pixel][channel] in LDS exactly as they are read (coalesced 16-byte chunks along
// channels) and the MFMA fragments are produced by LDS reads that transpose on the fly:
//   bf16: ds_read_b64_tr_b16 on [32 pixel][16 channel] sub-tiles (1 KiB each, the conflict-free
//         layout of the CDNA4 guide) -> 4 pixels of one channel per lane per read, 2 reads per operand;
//   f32 : plain ds_read_b32, lane (channel = lane&15, pixel = lane>>4) is exactly the 16x16x4 operand.
// Block tile 96(n) x 96(c), 2x2 waves of 48x48, one tap and one pixel range (split-K) per block; the
// f32 partial tiles are combined with atomicAdd into the packed f32 gradient (coalesced along c).
#include "y5m_conv.h"
#include <stdlib.h>

#define WG_THREADS 256
#define WG_TN 96
#define WG_TC 96

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef y5m_wgrad_args WgradParams;

template <typename T> struct WgTraits;
template <> struct WgTraits<bf16_t> { static constexpr int KCH = 64; static constexpr int OPB = 64 * 96 * 2; };
template <> struct WgTraits<float> { static constexpr int KCH = 32; static constexpr int LDW = 112; static constexpr int OPB = 32 * 112 * 4; };

// 4 k-values (pixels 4g+j of a 16-row block) of channel (lane&15) from a [32][16] bf16 sub-tile.
// Each lane supplies the address of ITS OWN 8-byte piece of the 4x16 block its 16-lane group covers:
// row (i>>2), columns 4*(i&3)..+3; the hardware returns column i, rows 0..3 (transposed).
__device__ __forceinline__ s16x4_t tr_read(const unsigned char* sub, int lane, int rowblk) {
    const int i = lane & 15, g = lane >> 4;
    const unsigned char* p = sub + ((rowblk * 16 + 4 * g + (i >> 2)) * 32 + (i & 3) * 8);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p));
}

template <typename T, bool DB>
__global__ __launch_bounds__(WG_THREADS) void wgrad_kernel(const WgradParams P) {
    constexpr int KCH = WgTraits<T>::KCH;
    constexpr int OPB = WgTraits<T>::OPB;            // bytes of one operand tile
    constexpr int CH = ElemTraits<T>::CH;
    constexpr int CPR = 96 / CH;                     // 16-byte chunks per pixel row (12 | 24)
    constexpr int NCHUNK = KCH * CPR;                // chunks per operand tile (768)
    constexpr int NLD = NCHUNK / WG_THREADS;         // 3
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid & 1, wc = wid >> 1;
    // Block order: all (tap, c-tile, n-tile) blocks of ONE pixel range are consecutive logical ids and
    // (XCD-aware remap: hardware block b runs on XCD b%8) land on the same XCD, so the 9 taps re-read
    // the same dY / shifted-X pixels from that XCD's L2 instead of from HBM.
    const int nblk = gridDim.x, hb = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = hb & 7;
    int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (hb >> 3);
    const int tap = bid % (P.th * P.tw); bid /= (P.th * P.tw);
    const int ct = bid % P.tiles_c; bid /= P.tiles_c;
    const int nt = bid % P.tiles_n;
    const int ksp = bid / P.tiles_n;
    const int n0 = nt * WG_TN, c0 = ct * WG_TC;
    const int ta = tap / P.tw, tb = tap - ta * P.tw;
    const int dh = P.dh0 + ta * P.dhs, dw = P.dw0 + tb * P.dws;

    const T* __restrict__ DY = reinterpret_cast<const T*>(P.dy);
    const T* __restrict__ X = reinterpret_cast<const T*>(P.x);

    // pixel range of this split (multiples of KCH)
    const int chunks_total = (P.M + KCH - 1) / KCH;
    const int per = (chunks_total + P.ksplit - 1) / P.ksplit;
    const int ch_lo = ksp * per, ch_hi = min(chunks_total, ch_lo + per);

    // staging assignment: id -> (4 consecutive chunks = 64 B) x pixel; cc = (id>>2)/KCH*4 + (id&3)
    int pl[NLD], ccl[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int id = tid + WG_THREADS * i;
        pl[i] = (id >> 2) % KCH;
        ccl[i] = ((id >> 2) / KCH) * 4 + (id & 3);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ry[NLD], rx[NLD];
    // unconditional loads: padding / out-of-range chunks come from the device zero page
    const char* Yb = reinterpret_cast<const char*>(DY);
    const char* Xb = reinterpret_cast<const char*>(X);
    const ptrdiff_t zy = reinterpret_cast<const char*>(P.zeros) - Yb, zx = reinterpret_cast<const char*>(P.zeros) - Xb;
    auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int m = chunk * KCH + pl[i];
            const int ch = ccl[i] * CH;
            const bool mv = m < P.M;
            const bool yv = mv && (n0 + ch < P.N);
            const int gx = m % P.Wg;
            const int t = m / P.Wg;
            const int gy = t % P.Hg;
            const int b = t / P.Hg;
            const int iy = gy * P.sy + dh, ix = gx * P.sx + dw;
            const bool xv = mv && (c0 + ch < P.C) && (unsigned)iy < (unsigned)P.Hin && (unsigned)ix < (unsigned)P.Win;
            const ptrdiff_t yo = yv ? (ptrdiff_t)(((size_t)m * P.lddy + n0 + ch) * sizeof(T)) : zy;
            const ptrdiff_t xo = xv ? (ptrdiff_t)((((size_t)(b * P.Hin + iy) * P.Win + ix) * P.ldx + c0 + ch) * sizeof(T)) : zx;
            ry[i] = *reinterpret_cast<const u32x4*>(Yb + yo);
            rx[i] = *reinterpret_cast<const u32x4*>(Xb + xo);
        }
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
        unsigned char* Ys = smem + buf * 2 * OPB;
        unsigned char* Xs = Ys + OPB;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int off;
            if constexpr (sizeof(T) == 2) {
                const int ch = ccl[i] * 8;
                off = ((pl[i] >> 5) * 6 + (ch >> 4)) * 1024 + (pl[i] & 31) * 32 + ((ch >> 3) & 1) * 16;
            } else {
                off = (pl[i] * WgTraits<float>::LDW + ccl[i] * 4) * 4;
            }
            *reinterpret_cast<u32x4*>(Ys + off) = ry[i];
            *reinterpret_cast<u32x4*>(Xs + off) = rx[i];
        }
    };

    f32x4 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* Ys = smem + buf * 2 * OPB;
        const unsigned char* Xs = Ys + OPB;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < KCH / 32; ++ks) {
                uint4 ya[3], xb[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const unsigned char* sub = Ys + (ks * 6 + wn * 3 + a) * 1024;
                    const s16x4_t lo = tr_read(sub, lane, 0), hi = tr_read(sub, lane, 1);
                    ya[a] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
                }
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const unsigned char* sub = Xs + (ks * 6 + wc * 3 + b) * 1024;
                    const s16x4_t lo = tr_read(sub, lane, 0), hi = tr_read(sub, lane, 1);
                    xb[b] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
                }
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ya[a]),
                                                                            __builtin_bit_cast(bf16x8_t, xb[b]), acc[a][b], 0, 0, 0);
            }
        } else {
            constexpr int LDW = WgTraits<float>::LDW;
            const float* Yf = reinterpret_cast<const float*>(Ys);
            const float* Xf = reinterpret_cast<const float*>(Xs);
            const int i = lane & 15, g = lane >> 4;
#pragma unroll
            for (int kk = 0; kk < KCH / 4; ++kk) {
                float ya[3], xb[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) ya[a] = Yf[(kk * 4 + g) * LDW + wn * 48 + a * 16 + i];
#pragma unroll
                for (int b = 0; b < 3; ++b) xb[b] = Xf[(kk * 4 + g) * LDW + wc * 48 + b * 16 + i];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[a], xb[b], acc[a][b], 0, 0, 0);
            }
        }
    };

    if (ch_lo < ch_hi) {
        load_chunk(ch_lo);
        store_chunk(0);
        __syncthreads();
        if constexpr (DB) {
            int cur = 0;
            for (int chk = ch_lo; chk < ch_hi; ++chk) {
                const bool more = chk + 1 < ch_hi;
                if (more) load_chunk(chk + 1);
                compute(cur);
                if (more) store_chunk(cur ^ 1);
                __syncthreads();
                cur ^= 1;
            }
        } else {
            for (int chk = ch_lo; chk < ch_hi; ++chk) {
                const bool more = chk + 1 < ch_hi;
                if (more) load_chunk(chk + 1);
                compute(0);
                __syncthreads();
                if (more) { store_chunk(0); __syncthreads(); }
            }
        }
        // D[n][c]: lane owns n = (lane>>4)*4 + r, c = lane&15 -> atomics coalesced along c
        const int i = lane & 15, g = lane >> 4;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int c = c0 + wc * 48 + b * 16 + i;
                if (c >= P.C) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * 48 + a * 16 + g * 4 + r;
                    if (n < P.N) atomicAdd(P.dwgt + (size_t)n * P.lddw + tap * P.C + c, acc[a][b][r]);
                }
            }
    }
}

extern "C" int y5m_wgrad(const y5m_wgrad_args* args, int dtype, void* stream) {
    WgradParams P = *args;
    const int CH = dtype == Y5M_BF16 ? 8 : 4;
    Y5M_REQUIRE(dtype == Y5M_F32 || dtype == Y5M_BF16, "dtype");
    Y5M_REQUIRE(P.zeros != nullptr, "args.zeros (16 zero bytes in device memory) is required");
    Y5M_REQUIRE(P.C % CH == 0 && P.N % CH == 0 && P.ldx % CH == 0 && P.lddy % CH == 0, "channel counts must be multiples of 16 bytes");
    Y5M_REQUIRE(P.M == P.B * P.Hg * P.Wg && P.M > 0, "M");
    P.tiles_n = (P.N + WG_TN - 1) / WG_TN;
    P.tiles_c = (P.C + WG_TC - 1) / WG_TC;
    const int taps = P.th * P.tw;
    const int KCH = dtype == Y5M_BF16 ? 64 : 32;
    const int chunks = (P.M + KCH - 1) / KCH;
    if (P.ksplit <= 0) {
        // fill the chip: ~4 blocks per CU, but keep >= 8 chunks per block so the prologue amortises
        const int base = P.tiles_n * P.tiles_c * taps;
        int ks = (1024 + base - 1) / base;
        const int maxks = (chunks + 7) / 8;
        ks = ks < 1 ? 1 : ks;
        ks = ks > maxks ? maxks : ks;
        P.ksplit = ks < 1 ? 1 : ks;
    }
    static int sbuf = -1;      // Y5M_WGRAD_SBUF=1: single LDS buffer (half the LDS, more resident workgroups)
    if (sbuf < 0) { const char* e = getenv("Y5M_WGRAD_SBUF"); sbuf = (e && e[0] == '1') ? 1 : 0; }
    const size_t opb = (size_t)(dtype == Y5M_BF16 ? WgTraits<bf16_t>::OPB : WgTraits<float>::OPB);
    const size_t lds = (sbuf ? 2 : 4) * opb;
    const unsigned grid = (unsigned)(P.tiles_n * P.tiles_c * taps * P.ksplit);
    hipStream_t st = y5m_stream(stream);
#define WG_LAUNCH(TT, DBV)                                                                                   \
    {                                                                                                        \
        static bool attr = false;                                                                            \
        if (!attr) { (void)hipFuncSetAttribute((const void*)wgrad_kernel<TT, DBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; } \
        hipLaunchKernelGGL((wgrad_kernel<TT, DBV>), dim3(grid), dim3(WG_THREADS), lds, st, P);               \
    }
    if (dtype == Y5M_BF16) { if (sbuf) WG_LAUNCH(bf16_t, false) else WG_LAUNCH(bf16_t, true) }
    else { if (sbuf) WG_LAUNCH(float, false) else WG_LAUNCH(float, true) }
#undef WG_LAUNCH
    Y5M_CHECK_LAUNCH("wgrad_kernel");
    return Y5M_OK;
}
