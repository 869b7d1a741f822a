// Weight-gradient kernel for gfx950:  dW[n][tap][c] += sum_m dY[m][n] * X[pix(m,tap)][c]
//
// GEMM with K = pixels. Both operands are pixel-major in HBM (NHWC), i.e. K is the STRIDED axis, so
// the tiles are staged [pixel][channel] in LDS exactly as they are read (coalesced 16-byte chunks along
// channels) and the MFMA fragments are produced by LDS reads that transpose on the fly:
//   bf16: ds_read_b64_tr_b16 on [32 pixel][16 channel] sub-tiles (1 KiB each, the conflict-free
//         layout of the CDNA4 guide) -> 4 pixels of one channel per lane per read, 2 reads per operand;
//   f32 : plain ds_read_b32, lane (channel = lane&15, pixel = lane>>4) is exactly the 16x16x4 operand.
// Block tile (WN*48)(n) x (WC*16*CFR)(c); the 4 waves are arranged WN x WC x WK: for the small-channel
// layers (N or C <= 48, stem C = 16) the spare waves split the pixels of each chunk (WK) instead of
// multiplying zero padding. One tap and one pixel range (split-K) per block; f32 partial tiles are
// combined with atomicAdd into the packed f32 gradient (coalesced along c).
#include "y5m_conv.h"
#include <stdlib.h>

#define WG_THREADS 256

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef y5m_wgrad_args WgradParams;

// 4 k-values (pixels 4g+j of a 16-row block) of channel (lane&15) from a [32][16] bf16 sub-tile.
// Each lane supplies the address of ITS OWN 8-byte piece of the 4x16 block its 16-lane group covers:
// row (i>>2), columns 4*(i&3)..+3; the hardware returns column i, rows 0..3 (transposed).
// (semantics verified on hardware: tools/probe_tr16.hip)
// lane_off = (4*(lane>>4) + ((lane&15)>>2))*32 + (lane&3)*8 is loop invariant and computed once per thread.
#define WG_SUB 1088     // sub-tile stride: 1024 + 64 so the 16-byte staging writes of neighbouring sub-tiles
                        // and pixels fall on distinct LDS banks (a plain 1024 stride is a 2-way write conflict)
__device__ __forceinline__ s16x4_t tr_read(const unsigned char* sub, int lane_off, int rowblk) {
    const unsigned char* p = sub + lane_off + rowblk * 512;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p));
}

template <typename T, int WN, int WC, int WK, int CFR>
struct WgCfg {
    static constexpr int TN = WN * 48;                 // channels of dY per block
    static constexpr int TC = WC * 16 * CFR;           // channels of X per block
    static constexpr bool BF = sizeof(T) == 2;
    static constexpr int KCH = BF ? 32 * (WK > 2 ? WK : 2) : 32;   // pixels per LDS chunk
    static constexpr int CH = BF ? 8 : 4;              // elements per 16-byte chunk
    static constexpr int LDY = BF ? TN : TN + 16;      // f32 row strides (+16: rows k, k+1 on disjoint banks)
    static constexpr int LDX = BF ? TC : TC + 16;
    static constexpr int YB = BF ? (KCH / 32) * (TN / 16) * WG_SUB : KCH * LDY * 4;   // bytes of the dY tile
    static constexpr int XB = BF ? (KCH / 32) * (TC / 16) * WG_SUB : KCH * LDX * 4;
    static constexpr int TPP = WG_THREADS / KCH;       // threads per pixel row of a chunk
    static constexpr int YCPR = TN / CH, XCPR = TC / CH;   // 16-byte chunks per pixel row
    static constexpr int NLDY = (YCPR + TPP - 1) / TPP;
    static constexpr int NLDX = (XCPR + TPP - 1) / TPP;
};

template <bool BF>
__device__ __forceinline__ int lds_chunk_off(int pl, int cc, int tile_ch, int ldrow) {
    if constexpr (BF) {
        const int ch = cc * 8;
        return ((pl >> 5) * (tile_ch / 16) + (ch >> 4)) * WG_SUB + (pl & 31) * 32 + ((ch >> 3) & 1) * 16;
    } else {
        return (pl * ldrow + cc * 4) * 4;
    }
}

template <typename T, int WN, int WC, int WK, int CFR>
__global__ __launch_bounds__(WG_THREADS) void wgrad_kernel(const WgradParams P) {
    using C = WgCfg<T, WN, WC, WK, CFR>;
    constexpr int KCH = C::KCH, CH = C::CH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid % WN, wc = (wid / WN) % WC, wk = wid / (WN * WC);
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
    const int n0 = nt * C::TN, c0 = ct * C::TC;
    const int ta = tap / P.tw, tb = tap - ta * P.tw;
    const int dh = P.dh0 + ta * P.dhs, dw = P.dw0 + tb * P.dws;

    const T* __restrict__ DY = reinterpret_cast<const T*>(P.dy);
    const T* __restrict__ X = reinterpret_cast<const T*>(P.x);

    const int chunks_total = (P.M + KCH - 1) / KCH;
    const int per = (chunks_total + P.ksplit - 1) / P.ksplit;
    const int ch_lo = ksp * per, ch_hi = min(chunks_total, ch_lo + per);

    u32x4 ry[C::NLDY], rx[C::NLDX];
    const char* Yb = reinterpret_cast<const char*>(DY);
    const char* Xb = reinterpret_cast<const char*>(X);
    const ptrdiff_t zy = reinterpret_cast<const char*>(P.zeros) - Yb, zx = reinterpret_cast<const char*>(P.zeros) - Xb;
    const float rcpW = 1.0f / (float)P.Wg, rcpH = 1.0f / (float)P.Hg;
    // pointwise layers: X pixel == dY pixel, no (b, y, x) decomposition at all
    const bool lin = P.th == 1 && P.tw == 1 && P.sy == 1 && P.sx == 1 && P.dh0 == 0 && P.dw0 == 0 && P.Hin == P.Hg &&
                     P.Win == P.Wg;
    // Staging: thread = (pixel pl of the chunk, lane tp among the TPP threads of that pixel); it loads the
    // 16-byte channel chunks tp, tp+TPP, ... of dY and of X for THAT pixel: one pixel decomposition per
    // thread per chunk, TPP consecutive threads read a contiguous 16*TPP-byte run.
    // Loads are unconditional: padding / out-of-range chunks come from the zero page (see y5m_conv.hip).
    const int pl = tid / C::TPP, tp = tid % C::TPP;
    auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
        const int m = chunk * KCH + pl;
        const bool mv = m < P.M;
        bool xin = mv;
        size_t pix = (size_t)m;
        if (!lin) {
            int gx, t, gy, b;
            fast_divmod(m, P.Wg, rcpW, t, gx);
            fast_divmod(t, P.Hg, rcpH, b, gy);
            const int iy = gy * P.sy + dh, ix = gx * P.sx + dw;
            xin = mv && (unsigned)iy < (unsigned)P.Hin && (unsigned)ix < (unsigned)P.Win;
            pix = (size_t)(b * P.Hin + iy) * P.Win + ix;
        }
        const ptrdiff_t ybase = (ptrdiff_t)(((size_t)m * P.lddy + n0) * sizeof(T));
        const ptrdiff_t xbase = (ptrdiff_t)((pix * P.ldx + c0) * sizeof(T));
#pragma unroll
        for (int i = 0; i < C::NLDY; ++i) {
            const int cc = tp + C::TPP * i;
            const bool yv = mv && cc < C::YCPR && (n0 + cc * CH < P.N);
            ry[i] = *reinterpret_cast<const u32x4*>(Yb + (yv ? ybase + cc * 16 : zy));
        }
#pragma unroll
        for (int i = 0; i < C::NLDX; ++i) {
            const int cc = tp + C::TPP * i;
            const bool xv = xin && cc < C::XCPR && (c0 + cc * CH < P.C);
            rx[i] = *reinterpret_cast<const u32x4*>(Xb + (xv ? xbase + cc * 16 : zx));
        }
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
        unsigned char* Ys = smem + buf * (C::YB + C::XB);
        unsigned char* Xs = Ys + C::YB;
#pragma unroll
        for (int i = 0; i < C::NLDY; ++i) {
            const int cc = tp + C::TPP * i;
            if (cc < C::YCPR) *reinterpret_cast<u32x4*>(Ys + lds_chunk_off<C::BF>(pl, cc, C::TN, C::LDY)) = ry[i];
        }
#pragma unroll
        for (int i = 0; i < C::NLDX; ++i) {
            const int cc = tp + C::TPP * i;
            if (cc < C::XCPR) *reinterpret_cast<u32x4*>(Xs + lds_chunk_off<C::BF>(pl, cc, C::TC, C::LDX)) = rx[i];
        }
    };

    f32x4 acc[3][CFR];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < CFR; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* Ys = smem + buf * (C::YB + C::XB);
        const unsigned char* Xs = Ys + C::YB;
        if constexpr (C::BF) {
            constexpr int KS = KCH / 32 / WK;           // k-steps (32 pixels) of this wave per chunk
            const int lane_off = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ks = wk * KS + s;
                uint4 ya[3], xb[CFR];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const unsigned char* sub = Ys + (ks * (C::TN / 16) + wn * 3 + a) * WG_SUB;
                    const s16x4_t lo = tr_read(sub, lane_off, 0), hi = tr_read(sub, lane_off, 1);
                    ya[a] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
                }
#pragma unroll
                for (int b = 0; b < CFR; ++b) {
                    const unsigned char* sub = Xs + (ks * (C::TC / 16) + wc * CFR + b) * WG_SUB;
                    const s16x4_t lo = tr_read(sub, lane_off, 0), hi = tr_read(sub, lane_off, 1);
                    xb[b] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
                }
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < CFR; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ya[a]),
                                                                            __builtin_bit_cast(bf16x8_t, xb[b]), acc[a][b], 0, 0, 0);
            }
        } else {
            const float* Yf = reinterpret_cast<const float*>(Ys);
            const float* Xf = reinterpret_cast<const float*>(Xs);
            const int i = lane & 15, g = lane >> 4;
            constexpr int KK = KCH / 4 / WK;
#pragma unroll
            for (int s = 0; s < KK; ++s) {
                const int kk = wk * KK + s;
                float ya[3], xb[CFR];
#pragma unroll
                for (int a = 0; a < 3; ++a) ya[a] = Yf[(kk * 4 + g) * C::LDY + wn * 48 + a * 16 + i];
#pragma unroll
                for (int b = 0; b < CFR; ++b) xb[b] = Xf[(kk * 4 + g) * C::LDX + (wc * CFR + b) * 16 + i];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < CFR; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[a], xb[b], acc[a][b], 0, 0, 0);
            }
        }
    };

    if (ch_lo < ch_hi) {
        load_chunk(ch_lo);
        store_chunk(0);
        __syncthreads();
        int cur = 0;
        for (int chk = ch_lo; chk < ch_hi; ++chk) {
            const bool more = chk + 1 < ch_hi;
            if (more) load_chunk(chk + 1);
            compute(cur);
            if (more) store_chunk(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
        // D[n][c]: lane owns n = (lane>>4)*4 + r, c = lane&15 -> atomics coalesced along c
        const int i = lane & 15, g = lane >> 4;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < CFR; ++b) {
                const int c = c0 + (wc * CFR + b) * 16 + i;
                if (c >= P.C) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * 48 + a * 16 + g * 4 + r;
                    if (n < P.N) atomicAdd(P.dwgt + (size_t)n * P.lddw + tap * P.C + c, acc[a][b][r]);
                }
            }
    }
}

template <typename T, int WN, int WC, int WK, int CFR>
static int launch_wgrad(WgradParams& P, hipStream_t st) {
    using C = WgCfg<T, WN, WC, WK, CFR>;
    P.tiles_n = (P.N + C::TN - 1) / C::TN;
    P.tiles_c = (P.C + C::TC - 1) / C::TC;
    const int taps = P.th * P.tw;
    const int chunks = (P.M + C::KCH - 1) / C::KCH;
    if (P.ksplit <= 0) {
        // fill the chip (~4 blocks per CU) but keep >= 8 chunks per block so the prologue amortises
        static int target = -1, minch = -1;     // Y5M_WGRAD_BLOCKS / Y5M_WGRAD_MINCH: tuning knobs
        if (target < 0) { const char* e = getenv("Y5M_WGRAD_BLOCKS"); target = e ? atoi(e) : 0; }
        if (minch < 0) { const char* e = getenv("Y5M_WGRAD_MINCH"); minch = e ? atoi(e) : 8; }
        const int base = P.tiles_n * P.tiles_c * taps;
        // measured (MI355X, B=64): every split adds one f32 atomic per output element, so pointwise layers
        // (few, large output tiles) want ~1 block per CU, 3x3 layers ~4 per CU, the 48x16 stem tile more
        const int tgt = target > 0 ? target : (taps == 1 ? 320 : (C::TC <= 16 ? 2048 : 1024));
        int ks = (tgt + base - 1) / base;
        const int maxks = (chunks + minch - 1) / minch;
        ks = ks > maxks ? maxks : ks;
        P.ksplit = ks < 1 ? 1 : ks;
    }
    const size_t lds = 2 * (size_t)(C::YB + C::XB);
    const unsigned grid = (unsigned)(P.tiles_n * P.tiles_c * taps * P.ksplit);
    auto kern = wgrad_kernel<T, WN, WC, WK, CFR>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WG_THREADS), lds, st, P);
    Y5M_CHECK_LAUNCH("wgrad_kernel");
    return Y5M_OK;
}

template <typename T>
static int dispatch_wgrad(WgradParams& P, hipStream_t st) {
    const bool n48 = P.N <= 48, c48 = P.C <= 48, c16 = P.C <= 16;
    if (n48 && c16) return launch_wgrad<T, 1, 1, 4, 1>(P, st);     // stem: 48 x 16
    if (n48 && c48) return launch_wgrad<T, 1, 1, 4, 3>(P, st);     // 48 x 48
    if (n48) return launch_wgrad<T, 1, 2, 2, 3>(P, st);            // 48 x 96
    if (c48) return launch_wgrad<T, 2, 1, 2, 3>(P, st);            // 96 x 48
    return launch_wgrad<T, 2, 2, 1, 3>(P, st);                     // 96 x 96
}

extern "C" int y5m_wgrad(const y5m_wgrad_args* args, int dtype, void* stream) {
    WgradParams P = *args;
    const int CH = dtype == Y5M_BF16 ? 8 : 4;
    Y5M_REQUIRE(dtype == Y5M_F32 || dtype == Y5M_BF16, "dtype");
    Y5M_REQUIRE(P.zeros != nullptr, "args.zeros (16 zero bytes in device memory) is required");
    Y5M_REQUIRE(P.C % CH == 0 && P.N % CH == 0 && P.ldx % CH == 0 && P.lddy % CH == 0, "channel counts must be multiples of 16 bytes");
    Y5M_REQUIRE(P.M == P.B * P.Hg * P.Wg && P.M > 0, "M");
    hipStream_t st = y5m_stream(stream);
    if (dtype == Y5M_BF16) return dispatch_wgrad<bf16_t>(P, st);
    return dispatch_wgrad<float>(P, st);
}
