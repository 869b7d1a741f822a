// Weight-gradient kernels for gfx950:  dW[n][tap][c] += sum_m dY[m][n] * X[pix(m,tap)][c]
//
// GEMM with K = pixels. Both operands are pixel-major in HBM (NHWC), i.e. K is the STRIDED axis, so
// the tiles are staged [pixel][channel] in LDS exactly as they are read (coalesced 16-byte chunks along
// channels) and the MFMA fragments are produced by LDS reads that transpose on the fly:
//   bf16: ds_read_b64_tr_b16 on [32 pixel][16 channel] sub-tiles (1 KiB each, the conflict-free
//         layout of the CDNA4 guide) -> 4 pixels of one channel per lane per read, 2 reads per operand;
//   f32 : plain ds_read_b32, lane (channel = lane&15, pixel = lane>>4) is exactly the 16x16x4 operand.
// wgrad_kernel: block tile (WN*16*NFR)(n) x (WC*16*CFR)(c), wave tile (16*NFR) x (16*CFR); the 4 waves are arranged
// WN x WC x WK: for the small-channel layers (N or C <= 48, stem C = 16) the spare waves split the pixels of each chunk
// (WK) instead of multiplying zero padding. One pixel range (split-K) per block; f32 partial tiles are combined with
// atomicAdd into the packed f32 gradient (coalesced along c).
// Taps: normally ONE tap per block (the 9 tap blocks of a pixel range run on one XCD and share dY through
// its L2). For the stem (C = 16) a block covers all 9 taps at once: the X tile is [pixel][9 x 16 "virtual channels"]
// (an im2col slice built by the loader, every 16-byte chunk with its own tap offset and bounds test), so dY is read once
// and a wave gets 3 x 9 MFMAs per 32-pixel step instead of 3 x 1.
// wgrad_dma_kernel: the same GEMM as a producer / consumer workgroup -- two producer waves issue LDS-DMA loads straight into
// the sub-tiles, four consumer waves only read fragments and issue MFMAs (192 x 192 block tile, one workgroup per CU).
// The forms that were tried and lost (row-of-taps, fragment pipelining, 128-pixel chunks, 8-wave tiles, register-staged
// producers, non-atomic slices) are recorded in NOTES.md.
#include "y5m_conv.h"
#include <stdlib.h>

#define WG_THREADS 256
#ifndef Y5M_EXP
#define Y5M_EXP 0
#endif

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

template <typename T, int WN, int WC, int WK, int CFR, int TPB, int NFR, int KX = 1>
struct WgCfg {
    static constexpr int TN = WN * 16 * NFR;                // channels of dY per block
    static constexpr int TC = WC * 16 * CFR;           // (virtual) channels of X per block = TPB taps x CBLK
    static constexpr int CBLK = TC / TPB;              // real channels per tap in the tile
    static_assert(TC % TPB == 0 && CBLK % 16 == 0, "a 16-channel fragment must not straddle two taps");
    static constexpr bool BF = sizeof(T) == 2;
    static constexpr int KCH = (BF ? 32 * (WK > 2 ? WK : 2) : 32) * KX;   // pixels per LDS chunk
    static constexpr int CH = BF ? 8 : 4;              // elements per 16-byte chunk
    static constexpr int LDY = BF ? TN : TN + 16;      // f32 row strides (+16: rows k, k+1 on disjoint banks)
    static constexpr int LDX = BF ? TC : TC + 16;
    static constexpr int YB = BF ? (KCH / 32) * (TN / 16) * WG_SUB : KCH * LDY * 4;   // bytes of the dY tile
    static constexpr int XB = BF ? (KCH / 32) * (TC / 16) * WG_SUB : KCH * LDX * 4;
    static constexpr int THREADS = WN * WC * WK * 64;  // 4 waves, or 8 for the wide 8-wave tile
    static constexpr int TPP = THREADS / KCH;          // threads per pixel row of a chunk
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

// SB: ONE LDS buffer (two barriers per chunk, half the LDS -> more resident blocks) instead of two
// KX: chunk = KX x 64 pixels (KX x 2 K-steps per wave and chunk). FP: fragment pipelining -- the transposing reads of
// K-step s+1 are issued one per MFMA of K-step s (sched_group_barrier), into a second fragment register set
template <typename T, int WN, int WC, int WK, int CFR, int TPB, int NFR, bool SB, int KX = 1, bool FP = false>
__global__ __launch_bounds__(WN * WC * WK * 64) void wgrad_kernel(const WgradParams P) {
    using C = WgCfg<T, WN, WC, WK, CFR, TPB, NFR, KX>;
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
    const int tgroups = (P.th * P.tw) / TPB;
    const int tap0 = (bid % tgroups) * TPB; bid /= tgroups;
    const int ct = bid % P.tiles_c; bid /= P.tiles_c;
    const int nt = bid % P.tiles_n;
    const int ksp = bid / P.tiles_n;
    const int n0 = nt * C::TN, c0 = ct * C::CBLK;

    const T* __restrict__ DY = reinterpret_cast<const T*>(P.dy);
    const T* __restrict__ X = reinterpret_cast<const T*>(P.x);

    const int chunks_total = (P.M + KCH - 1) / KCH;
    const int per = (chunks_total + P.ksplit - 1) / P.ksplit;
    const int ch_lo = ksp * per, ch_hi = min(chunks_total, ch_lo + per);

    u32x4 ry[C::NLDY], rx[C::NLDX];
    // Global loads go through BUFFER resources (raw, stride 0): the address is SGPR base + one 32-bit VGPR
    // byte offset, and an offset >= num_records returns zeros in hardware. Out-of-image taps, channel
    // padding and the pixel tail (rows >= M of dY lie behind num_records by construction) therefore cost a
    // select of the offset at most -- no 64-bit address arithmetic, no zero-page redirection, no masking of
    // the data. (This loop was VALU-bound on exactly that arithmetic: ~120 VALU incl. 27 quarter-rate
    // integer multiplies per 18 MFMAs; with everything but the loop skeleton removed it still took 45 % of
    // the kernel's time.)
    constexpr unsigned OOB = 0x80000000u;                                   // tensors are < 2 GiB (checked at launch)
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(DY), 0, (unsigned)((size_t)P.M * P.lddy * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(X), 0, (unsigned)((size_t)P.B * P.Hin * P.Win * P.ldx * sizeof(T)), 0x00020000);
    // pointwise layers: X pixel == dY pixel, no (b, y, x) decomposition at all
    const bool lin = P.th == 1 && P.tw == 1 && P.sy == 1 && P.sx == 1 && P.dh0 == 0 && P.dw0 == 0 && P.Hin == P.Hg &&
                     P.Win == P.Wg;
    // Staging: thread = (pixel pl of the chunk, lane tp among the TPP threads of that pixel); it loads the
    // 16-byte channel chunks tp, tp+TPP, ... of dY and of X for THAT pixel, TPP consecutive threads read a
    // contiguous 16*TPP-byte run. Per-thread state advances by uniform steps from chunk to chunk.
    const int pl = tid / C::TPP, tp = tid % C::TPP;
    // loop-invariant per-load byte offsets inside the pixel row (or OOB for channel padding)
    unsigned yadd[C::NLDY], xadd[C::NLDX];
    int xdh[C::NLDX], xdw[C::NLDX];
#pragma unroll
    for (int i = 0; i < C::NLDY; ++i) {
        const int cc = tp + C::TPP * i;
        yadd[i] = (cc < C::YCPR && n0 + cc * CH < P.N) ? (unsigned)((n0 + cc * CH) * sizeof(T)) : OOB;
    }
#pragma unroll
    for (int i = 0; i < C::NLDX; ++i) {
        const int cc = tp + C::TPP * i;                    // virtual chunk -> (tap tap0 + tl, real channel chunk)
        const int tl = cc / (C::CBLK / CH);
        const int tap = tap0 + (tl < TPB ? tl : 0);
        const int ta = tap / P.tw, tb = tap - ta * P.tw;
        xdh[i] = P.dh0 + ta * P.dhs;
        xdw[i] = P.dw0 + tb * P.dws;
        const int chn = c0 + (cc - tl * (C::CBLK / CH)) * CH;
        xadd[i] = (cc < C::XCPR && tl < TPB && chn < P.C) ? (unsigned)(chn * sizeof(T)) : OOB;
    }
    // pixel m = chunk*KCH + pl as (image gb, row gy*sy, column gx*sx); advanced incrementally per chunk
    int gx = 0, gy = 0, gb = 0;
    unsigned yrow, xrow;                                   // byte offsets of dY row m / X row m (pointwise layers)
    {
        const int m = ch_lo * KCH + pl;
        yrow = (unsigned)m * (unsigned)(P.lddy * sizeof(T));
        xrow = (unsigned)m * (unsigned)(P.ldx * sizeof(T));
        const float rcpW = 1.0f / (float)P.Wg, rcpH = 1.0f / (float)P.Hg;
        int t;
        fast_divmod(m, P.Wg, rcpW, t, gx);
        fast_divmod(t, P.Hg, rcpH, gb, gy);
    }
    const int stepx = KCH % P.Wg, stepy = (KCH / P.Wg) % P.Hg, stepb = KCH / (P.Wg * P.Hg);
    const unsigned ystep = (unsigned)(KCH * P.lddy * sizeof(T));
    const unsigned ldxb = (unsigned)(P.ldx * sizeof(T));
    auto load_chunk = [&]() __attribute__((always_inline)) {       // loads the chunk the state points at, then advances
#pragma unroll
        for (int i = 0; i < C::NLDY; ++i) ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, yrow + yadd[i], 0, 0);
        if (lin) {
            // X row == dY row index: same row offset scaled by the X row pitch
#pragma unroll
            for (int i = 0; i < C::NLDX; ++i)
                rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xrow + xadd[i], 0, 0);
        } else {
            const int y0 = gy * P.sy, x0 = gx * P.sx, r0 = gb * P.Hin;
#pragma unroll
            for (int i = 0; i < C::NLDX; ++i) {
                const int iy = y0 + xdh[i], ix = x0 + xdw[i];
                const bool in = (unsigned)iy < (unsigned)P.Hin && (unsigned)ix < (unsigned)P.Win;
                // < 2^24 pixels (checked at launch): 24-bit multiplies are full rate
                const unsigned pix = __umul24((unsigned)(r0 + iy), (unsigned)P.Win) + (unsigned)ix;
                const unsigned off = __umul24(pix, ldxb) + xadd[i];
                rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, in ? off : OOB, 0, 0);
            }
        }
        // advance to the next chunk: m += KCH
        yrow += ystep;
        xrow += KCH * ldxb;
        gx += stepx;
        const int c1 = gx >= P.Wg ? 1 : 0;
        gx -= c1 ? P.Wg : 0;
        gy += stepy + c1;
        const int c2 = gy >= P.Hg ? 1 : 0;
        gy -= c2 ? P.Hg : 0;
        gb += stepb + c2;
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

    f32x4 acc[NFR][CFR];
#pragma unroll
    for (int a = 0; a < NFR; ++a)
#pragma unroll
        for (int b = 0; b < CFR; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* Ys = smem + buf * (C::YB + C::XB);
        const unsigned char* Xs = Ys + C::YB;
        if constexpr (C::BF && FP) {
            constexpr int KS = KCH / 32 / WK;
            const int lane_off = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
            uint4 ya[2][NFR], xb[2][CFR];
            auto rd = [&](int s, uint4* yr, uint4* xr) __attribute__((always_inline)) {
                const int ks = wk * KS + s;
                // X fragments first: every one of them is an operand of the K-step's FIRST MFMAs (a outer, b inner), the
                // dY fragment read last only of its last CFR ones
#pragma unroll
                for (int b = 0; b < CFR; ++b) {
                    const unsigned char* sub = Xs + (ks * (C::TC / 16) + wc * CFR + b) * WG_SUB;
                    const s16x4_t lo = tr_read(sub, lane_off, 0), hi = tr_read(sub, lane_off, 1);
                    xr[b] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
                }
#pragma unroll
                for (int a = 0; a < NFR; ++a) {
                    const unsigned char* sub = Ys + (ks * (C::TN / 16) + wn * NFR + a) * WG_SUB;
                    const s16x4_t lo = tr_read(sub, lane_off, 0), hi = tr_read(sub, lane_off, 1);
                    yr[a] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
                }
            };
            rd(0, ya[0], xb[0]);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) rd(s + 1, ya[(s + 1) & 1], xb[(s + 1) & 1]);
#pragma unroll
                for (int a = 0; a < NFR; ++a)
#pragma unroll
                    for (int b = 0; b < CFR; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ya[s & 1][a]),
                                                                            __builtin_bit_cast(bf16x8_t, xb[s & 1][b]), acc[a][b], 0, 0, 0);
            }
            // the schedule of the whole chunk, in program order: the 2 * (NFR + CFR) reads of K-step 0; then per K-step two of
            // its MFMAs, two reads (one fragment) of the next K-step, ...; the last K-step's MFMAs back to back
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (NFR + CFR), 0);
#pragma unroll
            for (int s = 0; s + 1 < KS; ++s)
#pragma unroll
                for (int i = 0; i < NFR + CFR; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            __builtin_amdgcn_sched_group_barrier(0x008, NFR * CFR, 0);
        } else if constexpr (C::BF) {
            constexpr int KS = KCH / 32 / WK;           // k-steps (32 pixels) of this wave per chunk
            const int lane_off = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ks = wk * KS + s;
                uint4 ya[NFR], xb[CFR];
#pragma unroll
                for (int a = 0; a < NFR; ++a) {
                    const unsigned char* sub = Ys + (ks * (C::TN / 16) + wn * NFR + a) * WG_SUB;
#if Y5M_EXP & 1
                    ya[a] = *reinterpret_cast<const uint4*>(sub + lane * 16);      // timing ablation: WRONG data
#else
                    const s16x4_t lo = tr_read(sub, lane_off, 0), hi = tr_read(sub, lane_off, 1);
                    ya[a] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
#endif
                }
#pragma unroll
                for (int b = 0; b < CFR; ++b) {
                    const unsigned char* sub = Xs + (ks * (C::TC / 16) + wc * CFR + b) * WG_SUB;
#if Y5M_EXP & 1
                    xb[b] = *reinterpret_cast<const uint4*>(sub + lane * 16);
#else
                    const s16x4_t lo = tr_read(sub, lane_off, 0), hi = tr_read(sub, lane_off, 1);
                    xb[b] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
#endif
                }
#if !(Y5M_EXP & 8)
#pragma unroll
                for (int a = 0; a < NFR; ++a)
#pragma unroll
                    for (int b = 0; b < CFR; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ya[a]),
                                                                            __builtin_bit_cast(bf16x8_t, xb[b]), acc[a][b], 0, 0, 0);
#else
#pragma unroll
                for (int a = 0; a < NFR; ++a)
#pragma unroll
                    for (int b = 0; b < CFR; ++b) acc[a][b][0] += __builtin_bit_cast(float, ya[a].x ^ xb[b].y);
#endif
            }
        } else {
            const float* Yf = reinterpret_cast<const float*>(Ys);
            const float* Xf = reinterpret_cast<const float*>(Xs);
            const int i = lane & 15, g = lane >> 4;
            constexpr int KK = KCH / 4 / WK;
#pragma unroll
            for (int s = 0; s < KK; ++s) {
                const int kk = wk * KK + s;
                float ya[NFR], xb[CFR];
#pragma unroll
                for (int a = 0; a < NFR; ++a) ya[a] = Yf[(kk * 4 + g) * C::LDY + wn * (16 * NFR) + a * 16 + i];
#pragma unroll
                for (int b = 0; b < CFR; ++b) xb[b] = Xf[(kk * 4 + g) * C::LDX + (wc * CFR + b) * 16 + i];
#pragma unroll
                for (int a = 0; a < NFR; ++a)
#pragma unroll
                    for (int b = 0; b < CFR; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[a], xb[b], acc[a][b], 0, 0, 0);
            }
        }
    };

    const bool slices = P.slices_cap > 0;          // non-atomic mode: every pixel range writes its own slice (even an empty one)
    if (ch_lo < ch_hi) {
        load_chunk();
        store_chunk(0);
        __syncthreads();
        int cur = 0;
        for (int chk = ch_lo; chk < ch_hi; ++chk) {
            const bool more = chk + 1 < ch_hi;
            if (more && !(Y5M_EXP & 4)) load_chunk();
            compute(cur);
            if constexpr (SB) {
                __syncthreads();                     // every wave is done reading the only buffer
                if (more && !(Y5M_EXP & 2)) store_chunk(0);
                __syncthreads();
            } else {
                if (more && !(Y5M_EXP & 2)) store_chunk(cur ^ 1);
                __syncthreads();
                cur ^= 1;
            }
        }
    }
    // K-waves (WK > 1: the waves of a block that split the pixels of every chunk) first add their partial tiles in
    // LDS, so the block issues ONE set of atomics instead of WK (pointwise layers are bound by those atomics)
    bool writer = true;
    if constexpr (WK > 1) {
        if (!slices) {
            constexpr int TILE_F = NFR * CFR * 4 * 64;             // floats of one wave's accumulators
            float* red = reinterpret_cast<float*>(smem);          // [(WK-1)][WN*WC][TILE_F]
            __syncthreads();                                       // every wave is done with the staged tiles
            if (wk > 0) {
                float* dstp = red + ((size_t)(wk - 1) * (WN * WC) + (wid % (WN * WC))) * TILE_F + lane;
#pragma unroll
                for (int a = 0; a < NFR; ++a)
#pragma unroll
                    for (int b = 0; b < CFR; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) dstp[((a * CFR + b) * 4 + r) * 64] = acc[a][b][r];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int k = 1; k < WK; ++k) {
                    const float* srcp = red + ((size_t)(k - 1) * (WN * WC) + (wid % (WN * WC))) * TILE_F + lane;
#pragma unroll
                    for (int a = 0; a < NFR; ++a)
#pragma unroll
                        for (int b = 0; b < CFR; ++b)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[a][b][r] += srcp[((a * CFR + b) * 4 + r) * 64];
                }
            }
            writer = wk == 0;
        }
    }
    if ((ch_lo < ch_hi || slices) && writer) {
        // D[n][c]: lane owns n = (lane>>4)*4 + r, c = lane&15 -> atomics coalesced along c
        const int i = lane & 15, g = lane >> 4;
#pragma unroll
        for (int a = 0; a < NFR; ++a)
#pragma unroll
            for (int b = 0; b < CFR; ++b) {
                const int v0 = (wc * CFR + b) * 16;                 // virtual channel of the fragment
                const int tap = tap0 + v0 / C::CBLK;
                const int c = c0 + v0 % C::CBLK + i;
                if (c >= P.C) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * (16 * NFR) + a * 16 + g * 4 + r;
                    if (n < P.N) {
                        float* d = P.dwgt + (size_t)n * P.lddw + tap * P.C + c;
                        if (slices) d[(size_t)(ksp * WK + wk) * P.N * P.lddw] = acc[a][b][r];   // one slice per (range, K-wave)
                        else atomicAdd(d, acc[a][b][r]);
                    }
                }
            }
    }
}

// ---- producer / consumer form (Y5M_WGRAD_PC=1; bf16, one tap per block, no K-waves) -----------------------------------
// What the timing ablations of the kernel above say (tools/exp_wgrad.sh, 192 -> 192 @ 40x40, B=64, one block per CU:
// 115 us): with the LDS stores removed 90 us, with the global loads removed as well 90, with the transposing reads replaced
// by half as many ds_read_b128 115 (they are free), with the MFMAs removed 108 -- the wave's own staging (vmcnt wait, 9
// ds_write_b128 at 13 cycles each, the second barrier) is the largest removable item, the matrix pipe is idle most of the
// time. Here the staging moves to four PRODUCER waves (4-7: global -> VGPR -> LDS, one chunk ahead in registers, one more
// in LDS) and waves 0-3 only read fragments and issue MFMAs; a SIMD holds one wave of each kind, so the producer's waits
// and stores run next to the consumer's MFMAs instead of in front of them. Two LDS buffers, ONE barrier per chunk for all
// eight waves (raw s_barrier behind lgkmcnt(0): a producer must not wait for the loads it has just issued).
#define PC_BARRIER() \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier(); \
    asm volatile("" ::: "memory");
// scheduling pattern of one half-iteration of the consumer loop: MFMA i, then its share of the R fragment reads
template <int I, int MF, int R>
__device__ __forceinline__ void pc_interleave() {
    if constexpr (I < MF) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int n = (I + 1) * R / MF - I * R / MF;
        if constexpr (n > 0) __builtin_amdgcn_sched_group_barrier(0x100, n, 0);
        pc_interleave<I + 1, MF, R>();
    }
}
#if Y5M_EXP & 256
#define PC_LOOP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#define PC_LOOP_BARRIER() PC_BARRIER()
#endif
template <int WN, int WC, int CFR, int NFR>
__global__ __launch_bounds__(512) void wgrad_pc_kernel(const WgradParams P) {
    using T = bf16_t;
    using C = WgCfg<T, WN, WC, 1, CFR, 1, NFR, 1>;
    static_assert(C::THREADS == 256, "four consumer waves");
    constexpr int KCH = C::KCH, CH = C::CH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wid >= 4;
    const int nblk = gridDim.x, hb = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = hb & 7;
    int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (hb >> 3);
    const int tgroups = P.th * P.tw;
    const int tap0 = bid % tgroups; bid /= tgroups;
    const int ct = bid % P.tiles_c; bid /= P.tiles_c;
    const int nt = bid % P.tiles_n;
    const int ksp = bid / P.tiles_n;
    const int n0 = nt * C::TN, c0 = ct * C::CBLK;
    const int chunks_total = (P.M + KCH - 1) / KCH;
    const int per = (chunks_total + P.ksplit - 1) / P.ksplit;
    const int ch_lo = ksp * per, ch_hi = min(chunks_total, ch_lo + per);
    if (ch_lo >= ch_hi) return;

    if (producer) {
        __builtin_amdgcn_s_setprio(3);          // the few producer instructions win the issue arbitration against the MFMA stream (+10 %)
        const T* __restrict__ DY = reinterpret_cast<const T*>(P.dy);
        const T* __restrict__ X = reinterpret_cast<const T*>(P.x);
        constexpr unsigned OOB = 0x80000000u;
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(DY), 0, (unsigned)((size_t)P.M * P.lddy * sizeof(T)), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(X), 0, (unsigned)((size_t)P.B * P.Hin * P.Win * P.ldx * sizeof(T)), 0x00020000);
        const bool lin = P.th == 1 && P.tw == 1 && P.sy == 1 && P.sx == 1 && P.dh0 == 0 && P.dw0 == 0 && P.Hin == P.Hg &&
                         P.Win == P.Wg;
        const int st = tid & 255;
        const int pl = st / C::TPP, tp = st % C::TPP;
        // two register sets: the chunk stored in iteration k was loaded in iteration k-2 (a full iteration of latency
        // tolerance on top of the one the LDS double buffer gives)
        u32x4 ry0[C::NLDY], rx0[C::NLDX], ry1[C::NLDY], rx1[C::NLDX];
        unsigned yadd[C::NLDY], xadd[C::NLDX];
#pragma unroll
        for (int i = 0; i < C::NLDY; ++i) {
            const int cc = tp + C::TPP * i;
            yadd[i] = (cc < C::YCPR && n0 + cc * CH < P.N) ? (unsigned)((n0 + cc * CH) * sizeof(T)) : OOB;
        }
#pragma unroll
        for (int i = 0; i < C::NLDX; ++i) {
            const int cc = tp + C::TPP * i;
            const int chn = c0 + cc * CH;
            xadd[i] = (cc < C::XCPR && chn < P.C) ? (unsigned)(chn * sizeof(T)) : OOB;
        }
        const int ta = tap0 / P.tw, tb = tap0 - ta * P.tw;
        const int xdh = P.dh0 + ta * P.dhs, xdw = P.dw0 + tb * P.dws;
        int gx = 0, gy = 0, gb = 0;
        unsigned yrow, xrow;
        {
            const int m = ch_lo * KCH + pl;
            yrow = (unsigned)m * (unsigned)(P.lddy * sizeof(T));
            xrow = (unsigned)m * (unsigned)(P.ldx * sizeof(T));
            const float rcpW = 1.0f / (float)P.Wg, rcpH = 1.0f / (float)P.Hg;
            int t;
            fast_divmod(m, P.Wg, rcpW, t, gx);
            fast_divmod(t, P.Hg, rcpH, gb, gy);
        }
        const int stepx = KCH % P.Wg, stepy = (KCH / P.Wg) % P.Hg, stepb = KCH / (P.Wg * P.Hg);
        const unsigned ystep = (unsigned)(KCH * P.lddy * sizeof(T));
        const unsigned ldxb = (unsigned)(P.ldx * sizeof(T));
        auto load_chunk = [&](u32x4 (&ry)[C::NLDY], u32x4 (&rx)[C::NLDX]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < C::NLDY; ++i) ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, yrow + yadd[i], 0, 0);
            if (lin) {
#pragma unroll
                for (int i = 0; i < C::NLDX; ++i) rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xrow + xadd[i], 0, 0);
            } else {
                // one tap per block: the tap-shifted pixel and its bounds test are shared by the thread's X loads
                const int iy = gy * P.sy + xdh, ix = gx * P.sx + xdw;
                const bool in = (unsigned)iy < (unsigned)P.Hin && (unsigned)ix < (unsigned)P.Win;
                const unsigned pix = __umul24((unsigned)(gb * P.Hin + iy), (unsigned)P.Win) + (unsigned)ix;
                const unsigned rowo = in ? __umul24(pix, ldxb) : OOB;
#pragma unroll
                for (int i = 0; i < C::NLDX; ++i)
                    rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (rowo + xadd[i]) | ((rowo | xadd[i]) & OOB), 0, 0);   // either part out of range -> top bit set
            }
            yrow += ystep;
            xrow += KCH * ldxb;
            gx += stepx;
            const int c1 = gx >= P.Wg ? 1 : 0;
            gx -= c1 ? P.Wg : 0;
            gy += stepy + c1;
            const int c2 = gy >= P.Hg ? 1 : 0;
            gy -= c2 ? P.Hg : 0;
            gb += stepb + c2;
        };
        auto store_chunk = [&](int buf, const u32x4 (&ry)[C::NLDY], const u32x4 (&rx)[C::NLDX]) __attribute__((always_inline)) {
            unsigned char* Ys = smem + buf * (C::YB + C::XB);
            unsigned char* Xs = Ys + C::YB;
#pragma unroll
            for (int i = 0; i < C::NLDY; ++i) {
                const int cc = tp + C::TPP * i;
                if (cc < C::YCPR) *reinterpret_cast<u32x4*>(Ys + lds_chunk_off<true>(pl, cc, C::TN, C::LDY)) = ry[i];
            }
#pragma unroll
            for (int i = 0; i < C::NLDX; ++i) {
                const int cc = tp + C::TPP * i;
                if (cc < C::XCPR) *reinterpret_cast<u32x4*>(Xs + lds_chunk_off<true>(pl, cc, C::TC, C::LDX)) = rx[i];
            }
        };
        // chunk j of the range (j = 0, 1, ...) travels through register set j & 1 into LDS buffer j & 1
        const int nch = ch_hi - ch_lo;
        load_chunk(ry0, rx0);
        if (1 < nch) load_chunk(ry1, rx1);
        store_chunk(0, ry0, rx0);
        if (2 < nch) load_chunk(ry0, rx0);
        PC_BARRIER()
        for (int j = 0; j < nch; j += 2) {
            if (j + 1 < nch) {                        // consumers are on chunk j (buffer 0)
                if (!(Y5M_EXP & 16)) store_chunk(1, ry1, rx1);
                if (j + 3 < nch && !(Y5M_EXP & 32)) load_chunk(ry1, rx1);
            }
            PC_LOOP_BARRIER()
            if (j + 1 >= nch) break;
            if (j + 2 < nch) {                        // consumers are on chunk j + 1 (buffer 1)
                if (!(Y5M_EXP & 16)) store_chunk(0, ry0, rx0);
                if (j + 4 < nch && !(Y5M_EXP & 32)) load_chunk(ry0, rx0);
            }
            PC_LOOP_BARRIER()
        }
        return;
    }

    // ---- consumers ----
    const int wn = wid % WN, wc = wid / WN;
    f32x4 acc[NFR][CFR];
#pragma unroll
    for (int a = 0; a < NFR; ++a)
#pragma unroll
        for (int b = 0; b < CFR; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane_off = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
    // Software pipeline over the K-steps (32 pixels each, two per chunk): the transposing reads of K-step t+1 are issued
    // between the MFMAs of K-step t (second fragment register set), and the chunk barrier sits between the LAST READS of a
    // chunk and its last MFMAs -- "buffer free" only needs the reads -- so the first reads of the next chunk fly under
    // those MFMAs too. The loop body has no branch: behind the last chunk the other buffer is read once more and ignored.
    static_assert(KCH == 64, "two K-steps per chunk");
    if constexpr (NFR == 6 && CFR == 6) {
        // 96 x 96 wave tile (192 x 192 block: a third less L2 -> LDS traffic and a third fewer fragment reads per MFMA than
        // 192 x 96): 144 accumulator registers, so only the X fragments get a second register set; a dY fragment is
        // re-read for the next K-step one row of MFMAs after its last use (the last row's fragment has two copies).
        uint4 ya[6], ya5b, xbA[6], xbB[6];
        auto frag = [&](const unsigned char* sub, uint4& r) __attribute__((always_inline)) {
            const s16x4_t lo = tr_read(sub, lane_off, 0), hi = tr_read(sub, lane_off, 1);
            r = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
        };
        auto rdx = [&](int buf, int ks, int b, uint4& r) __attribute__((always_inline)) {
            frag(smem + buf * (C::YB + C::XB) + C::YB + (ks * (C::TC / 16) + wc * CFR + b) * WG_SUB, r);
        };
        auto rdy = [&](int buf, int ks, int a, uint4& r) __attribute__((always_inline)) {
            frag(smem + buf * (C::YB + C::XB) + (ks * (C::TN / 16) + wn * NFR + a) * WG_SUB, r);
        };
        // MFMAs of the current K-step (X set A / B and the last dY fragment's copy by parity), reads of K-step (bufN, ksN)
#define PC_HALF(XC, XN, Y5C, Y5N, bufN, ksN) \
_Pragma("unroll") \
        for (int a = 0; a < 6; ++a) { \
_Pragma("unroll") \
            for (int b = 0; b < 6; ++b) \
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a == 5 ? Y5C : ya[a]), \
                                                                    __builtin_bit_cast(bf16x8_t, XC[b]), acc[a][b], 0, 0, 0); \
            rdx(bufN, ksN, a, XN[a]); \
            if (a == 0) rdy(bufN, ksN, 5, Y5N); else rdy(bufN, ksN, a - 1, ya[a - 1]); \
        } \
_Pragma("unroll") \
        for (int a = 0; a < 6; ++a) { \
_Pragma("unroll") \
            for (int i = 0; i < 4; ++i) { \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
            } \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); \
        }
        PC_BARRIER()
#pragma unroll
        for (int b = 0; b < 6; ++b) rdx(0, 0, b, xbA[b]);
#pragma unroll
        for (int a = 0; a < 6; ++a) rdy(0, 0, a, ya[a]);
        int cur = 0;
        for (int chk = ch_lo; chk < ch_hi; ++chk) {
            PC_HALF(xbA, xbB, ya[5], ya5b, cur, 1)
            PC_LOOP_BARRIER()                          // every read of this chunk has returned: its buffer is free
            PC_HALF(xbB, xbA, ya5b, ya[5], cur ^ 1, 0)
            cur ^= 1;
        }
#undef PC_HALF
    } else {
    uint4 ya0[NFR], xb0[CFR], ya1[NFR], xb1[CFR];
    auto rd = [&](int buf, int ks, uint4 (&yr)[NFR], uint4 (&xr)[CFR]) __attribute__((always_inline)) {
        const unsigned char* Ys = smem + buf * (C::YB + C::XB);
        const unsigned char* Xs = Ys + C::YB;
#pragma unroll
        for (int b = 0; b < CFR; ++b) {
            const unsigned char* sub = Xs + (ks * (C::TC / 16) + wc * CFR + b) * WG_SUB;
            const s16x4_t lo = tr_read(sub, lane_off, 0), hi = tr_read(sub, lane_off, 1);
            xr[b] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
        }
#pragma unroll
        for (int a = 0; a < NFR; ++a) {
            const unsigned char* sub = Ys + (ks * (C::TN / 16) + wn * NFR + a) * WG_SUB;
            const s16x4_t lo = tr_read(sub, lane_off, 0), hi = tr_read(sub, lane_off, 1);
            yr[a] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
        }
    };
    auto mm = [&](const uint4 (&yr)[NFR], const uint4 (&xr)[CFR]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < NFR; ++a)
#pragma unroll
            for (int b = 0; b < CFR; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, yr[a]),
                                                                    __builtin_bit_cast(bf16x8_t, xr[b]), acc[a][b], 0, 0, 0);
    };
    // issue order of one half-iteration: MFMA, then its share of the 2 * (NFR + CFR) reads of the next K-step
    auto interleave = [&]() __attribute__((always_inline)) { pc_interleave<0, NFR * CFR, 2 * (NFR + CFR)>(); };
    PC_BARRIER()
    rd(0, 0, ya0, xb0);
    int cur = 0;
    for (int chk = ch_lo; chk < ch_hi; ++chk) {
        if (!(Y5M_EXP & 128)) rd(cur, 1, ya1, xb1);
        if (!(Y5M_EXP & 64)) mm(ya0, xb0);
        if (!(Y5M_EXP & (64 | 128))) interleave();
        PC_LOOP_BARRIER()                              // every read of this chunk has returned: its buffer is free
        if (!(Y5M_EXP & 128)) rd(cur ^ 1, 0, ya0, xb0);
        if (!(Y5M_EXP & 64)) mm(ya1, xb1);
        if (!(Y5M_EXP & (64 | 128))) interleave();
        cur ^= 1;
    }
    }
    {
        const bool slices = P.slices_cap > 0;
        const int i = lane & 15, g = lane >> 4;
#pragma unroll
        for (int a = 0; a < NFR; ++a)
#pragma unroll
            for (int b = 0; b < CFR; ++b) {
                const int c = c0 + (wc * CFR + b) * 16 + i;
                if (c >= P.C) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * (16 * NFR) + a * 16 + g * 4 + r;
                    if (n < P.N) {
                        float* d = P.dwgt + (size_t)n * P.lddw + tap0 * P.C + c;
                        if (slices) d[(size_t)ksp * P.N * P.lddw] = acc[a][b][r];
                        else if (!(Y5M_EXP & 512) || acc[a][b][r] == 12345.678f) atomicAdd(d, acc[a][b][r]);
                    }
                }
            }
    }
}

// ---- producer / consumer form with LDS-DMA staging (Y5M_WGRAD_PC bit 4; bf16, one tap per block, full channel tiles) ----
// Same consumers as wgrad_pc_kernel; the NPW producer waves issue `buffer_load_dwordx4 ... lds` straight into the
// [32 pixel][16 channel] sub-tiles (one DMA instruction = one 1 KiB sub-tile: lane l supplies the address of pixel l >> 1,
// channel half l & 1, and the hardware writes LDS[m0 + 16 l]; out-of-image taps and the pixel tail are offsets >= num_records
// = zeros). No staging registers, no ds_write; THREE LDS buffers: in iteration j the producers issue chunk j + 2 and wait
// (counted vmcnt) for chunk j + 1, so a DMA has a whole iteration to land. Producer wave (g, h): pixel group g (32 of the
// chunk's 64 pixels), the h-th share of the dY and X sub-tiles.
__device__ __forceinline__ void wg_dma16(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}
template <int WN, int WC, int CFR, int NFR, int NPW>
__global__ __launch_bounds__(256 + 64 * NPW) void wgrad_dma_kernel(const WgradParams P) {
    using T = bf16_t;
    using C = WgCfg<T, WN, WC, 1, CFR, 1, NFR, 1>;
    static_assert(C::THREADS == 256, "four consumer waves");
    constexpr int KCH = C::KCH;
    static_assert(KCH == 64, "two 32-pixel groups per chunk");
    constexpr int BUF = C::YB + C::XB, NB = 3;
    constexpr int YS = C::TN / 16, XS = C::TC / 16;       // sub-tiles per 32-pixel group
    constexpr int NG = NPW == 1 ? 2 : 1;                  // pixel groups per producer wave (one wave: both)
    constexpr int SH = NPW == 1 ? 1 : NPW / 2;            // producer waves per pixel group
    static_assert(NPW == 1 || NPW == 2 || NPW == 4, "producer waves");
    static_assert(YS % SH == 0 && XS % SH == 0, "sub-tiles split evenly");
    constexpr int NDY = YS / SH, NDX = XS / SH, NDW = NG * (NDY + NDX);      // DMA instructions per producer wave and chunk
    static_assert(NDW <= 63, "counted vmcnt");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wid >= 4;
    const int nblk = gridDim.x, hb = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = hb & 7;
    int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (hb >> 3);
    const int tgroups = P.th * P.tw;
    const int tap0 = bid % tgroups; bid /= tgroups;
    const int ct = bid % P.tiles_c; bid /= P.tiles_c;
    const int nt = bid % P.tiles_n;
    const int ksp = bid / P.tiles_n;
    const int n0 = nt * C::TN, c0 = ct * C::CBLK;
    const int chunks_total = (P.M + KCH - 1) / KCH;
    const int per = (chunks_total + P.ksplit - 1) / P.ksplit;
    const int ch_lo = ksp * per, ch_hi = min(chunks_total, ch_lo + per);
    if (ch_lo >= ch_hi) return;
    const int nch = ch_hi - ch_lo;

    if (producer) {
        __builtin_amdgcn_s_setprio(3);
        constexpr unsigned OOB = 0x80000000u;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(P.dy), 0, (unsigned)((size_t)P.M * P.lddy * sizeof(T)), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(P.x), 0, (unsigned)((size_t)P.B * P.Hin * P.Win * P.ldx * sizeof(T)), 0x00020000);
        const int pw = wid - 4, g0 = NPW == 1 ? 0 : pw % 2, h = NPW == 1 ? 0 : pw / 2;
        const int px = lane >> 1, half = lane & 1;
        const int ta = tap0 / P.tw, tb = tap0 - ta * P.tw;
        const int xdh = P.dh0 + ta * P.dhs, xdw = P.dw0 + tb * P.dws;
        const unsigned ldxb = (unsigned)(P.ldx * sizeof(T)), ldyb = (unsigned)(P.lddy * sizeof(T));
        // this lane's pixel of the chunk the state points at, per pixel group of this wave: m = chunk * 64 + g * 32 + px
        int gx[NG], gy[NG], gb[NG];
        unsigned yoff[NG];                                    // dY: row m, this wave's first sub-tile, this lane's half
        unsigned ydst[NG], xdst[NG];                          // LDS address of the wave's first dY / X sub-tile inside a buffer
        const float rcpW = 1.0f / (float)P.Wg, rcpH = 1.0f / (float)P.Hg;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int g = g0 + q;
            const int m = ch_lo * KCH + g * 32 + px;
            yoff[q] = (unsigned)m * ldyb + (unsigned)((n0 + h * NDY * 16 + half * 8) * sizeof(T));
            int t;
            fast_divmod(m, P.Wg, rcpW, t, gx[q]);
            fast_divmod(t, P.Hg, rcpH, gb[q], gy[q]);
            ydst[q] = lds0 + (unsigned)((g * YS + h * NDY) * WG_SUB);
            xdst[q] = lds0 + (unsigned)(C::YB + (g * XS + h * NDX) * WG_SUB);
        }
        const unsigned xch = (unsigned)((c0 + h * NDX * 16 + half * 8) * sizeof(T));
        const int stepx = KCH % P.Wg, stepy = (KCH / P.Wg) % P.Hg, stepb = KCH / (P.Wg * P.Hg);
        auto issue = [&](unsigned boff) __attribute__((always_inline)) {     // the chunk the state points at -> buffer at boff; advance
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const int iy = gy[q] * P.sy + xdh, ix = gx[q] * P.sx + xdw;
                const bool in = (unsigned)iy < (unsigned)P.Hin && (unsigned)ix < (unsigned)P.Win;
                const unsigned pix = __umul24((unsigned)(gb[q] * P.Hin + iy), (unsigned)P.Win) + (unsigned)ix;
                const unsigned xoff = in ? __umul24(pix, ldxb) + xch : OOB;
#pragma unroll
                for (int i = 0; i < NDY; ++i) wg_dma16(rs_y, yoff[q] + (unsigned)(i * 32), ydst[q] + boff + (unsigned)(i * WG_SUB));
#pragma unroll
                for (int i = 0; i < NDX; ++i) wg_dma16(rs_x, xoff + (unsigned)(i * 32), xdst[q] + boff + (unsigned)(i * WG_SUB));
                yoff[q] += (unsigned)KCH * ldyb;
                gx[q] += stepx;
                const int c1 = gx[q] >= P.Wg ? 1 : 0;
                gx[q] -= c1 ? P.Wg : 0;
                gy[q] += stepy + c1;
                const int c2 = gy[q] >= P.Hg ? 1 : 0;
                gy[q] -= c2 ? P.Hg : 0;
                gb[q] += stepb + c2;
            }
        };
        // (chunks behind the block's range are issued as well -- the loop has no branch; behind the tensor they read as zeros)
        issue(0u);
        issue((unsigned)BUF);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDW) : "memory");
        PC_BARRIER()
        unsigned nb = 2u * BUF;                                // buffer of chunk j + 2
        for (int j = 0; j < nch; ++j) {
            issue(nb);
            nb = nb == 2u * BUF ? 0u : nb + (unsigned)BUF;
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDW) : "memory");     // chunk j + 1 has landed
            PC_BARRIER()
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing may still be writing this workgroup's LDS when it ends
        return;
    }

    // ---- consumers (wgrad_pc_kernel's loops over three buffers) ----
    const int wn = wid % WN, wc = wid / WN;
    f32x4 acc[NFR][CFR];
#pragma unroll
    for (int a = 0; a < NFR; ++a)
#pragma unroll
        for (int b = 0; b < CFR; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane_off = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
    auto frag = [&](const unsigned char* sub, uint4& r) __attribute__((always_inline)) {
        const s16x4_t lo = tr_read(sub, lane_off, 0), hi = tr_read(sub, lane_off, 1);
        r = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
    };
    auto rdx = [&](unsigned boff, int ks, int b, uint4& r) __attribute__((always_inline)) {
        frag(smem + boff + C::YB + (ks * XS + wc * CFR + b) * WG_SUB, r);
    };
    auto rdy = [&](unsigned boff, int ks, int a, uint4& r) __attribute__((always_inline)) {
        frag(smem + boff + (ks * YS + wn * NFR + a) * WG_SUB, r);
    };
    unsigned cur = 0u;
    if constexpr (NFR == 6 && CFR == 6) {
        uint4 ya[6], ya5b, xbA[6], xbB[6];
#define DM_HALF(XC, XN, Y5C, Y5N, boffN, ksN) \
_Pragma("unroll") \
        for (int a = 0; a < 6; ++a) { \
_Pragma("unroll") \
            for (int b = 0; b < 6; ++b) \
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a == 5 ? Y5C : ya[a]), \
                                                                    __builtin_bit_cast(bf16x8_t, XC[b]), acc[a][b], 0, 0, 0); \
            rdx(boffN, ksN, a, XN[a]); \
            if (a == 0) rdy(boffN, ksN, 5, Y5N); else rdy(boffN, ksN, a - 1, ya[a - 1]); \
        } \
_Pragma("unroll") \
        for (int a = 0; a < 6; ++a) { \
_Pragma("unroll") \
            for (int i = 0; i < 4; ++i) { \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
            } \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); \
        }
        PC_BARRIER()
#pragma unroll
        for (int b = 0; b < 6; ++b) rdx(0u, 0, b, xbA[b]);
#pragma unroll
        for (int a = 0; a < 6; ++a) rdy(0u, 0, a, ya[a]);
        for (int j = 0; j < nch; ++j) {
            const unsigned nxt = cur == 2u * BUF ? 0u : cur + (unsigned)BUF;
            DM_HALF(xbA, xbB, ya[5], ya5b, cur, 1)
            PC_BARRIER()
            DM_HALF(xbB, xbA, ya5b, ya[5], nxt, 0)
            cur = nxt;
        }
#undef DM_HALF
    } else {
        uint4 ya0[NFR], xb0[CFR], ya1[NFR], xb1[CFR];
        auto rd = [&](unsigned boff, int ks, uint4 (&yr)[NFR], uint4 (&xr)[CFR]) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < CFR; ++b) rdx(boff, ks, b, xr[b]);
#pragma unroll
            for (int a = 0; a < NFR; ++a) rdy(boff, ks, a, yr[a]);
        };
        auto mm = [&](const uint4 (&yr)[NFR], const uint4 (&xr)[CFR]) __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < NFR; ++a)
#pragma unroll
                for (int b = 0; b < CFR; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, yr[a]),
                                                                        __builtin_bit_cast(bf16x8_t, xr[b]), acc[a][b], 0, 0, 0);
        };
        PC_BARRIER()
        rd(0u, 0, ya0, xb0);
        for (int j = 0; j < nch; ++j) {
            const unsigned nxt = cur == 2u * BUF ? 0u : cur + (unsigned)BUF;
            rd(cur, 1, ya1, xb1);
            mm(ya0, xb0);
            pc_interleave<0, NFR * CFR, 2 * (NFR + CFR)>();
            PC_BARRIER()
            rd(nxt, 0, ya0, xb0);
            mm(ya1, xb1);
            pc_interleave<0, NFR * CFR, 2 * (NFR + CFR)>();
            cur = nxt;
        }
    }
    {
        const int i = lane & 15, gq = lane >> 4;
#pragma unroll
        for (int a = 0; a < NFR; ++a)
#pragma unroll
            for (int b = 0; b < CFR; ++b) {
                const int c = c0 + (wc * CFR + b) * 16 + i;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * (16 * NFR) + a * 16 + gq * 4 + r;
                    atomicAdd(P.dwgt + (size_t)n * P.lddw + tap0 * P.C + c, acc[a][b][r]);
                }
            }
    }
}

static thread_local bool g_plan_only = false;   // y5m_wgrad_slices: run the dispatch + split-K sizing, launch nothing
static thread_local int g_plan_slices = 0;

template <typename T, int WN, int WC, int WK, int CFR, int TPB = 1, int NFR = 3, int KX = 1, bool FP = false, int PC = 0>
static int launch_wgrad(WgradParams& P, hipStream_t st) {
    using C = WgCfg<T, WN, WC, WK, CFR, TPB, NFR, KX>;
    P.tiles_n = (P.N + C::TN - 1) / C::TN;
    P.tiles_c = (P.C + C::CBLK - 1) / C::CBLK;
    const int taps = (P.th * P.tw) / TPB;                 // tap groups (blocks along the tap axis)
    const int chunks = (P.M + C::KCH - 1) / C::KCH;
    if (P.ksplit <= 0) {
        // fill the chip (~4 blocks per CU) but keep >= 8 chunks per block so the prologue amortises
        static int target = -1, minch = -1;     // Y5M_WGRAD_BLOCKS / Y5M_WGRAD_MINCH: tuning knobs
        if (target < 0) { const char* e = getenv("Y5M_WGRAD_BLOCKS"); target = e ? atoi(e) : 0; }
        if (minch < 0) { const char* e = getenv("Y5M_WGRAD_MINCH"); minch = e ? atoi(e) : 8; }
        const int base = P.tiles_n * P.tiles_c * taps;
        // measured (MI355X, B=64): every split adds one f32 atomic per output element, so pointwise layers
        // (few, large output tiles) want ~1 block per CU. 3x3 layers: fill the chip with ONE resident round --
        // blocks <= resident capacity (a second, nearly empty round cost 20 %: 137 -> 111 us on the 192x192
        // layers when the count dropped from 522 to <= 512); capacity = blocks per CU (LDS / VGPR limited:
        // 2 for the 96x48 wave tile, 3 otherwise) x 256 CUs. The 48x16 stem tile (one tap per block) wants more.
        static int sbk = -1;
        if (sbk < 0) { const char* e = getenv("Y5M_WGRAD_SB"); sbk = e ? atoi(e) : 1; }
        const int per_cu = (int)((160 * 1024) / (2 * (size_t)(C::YB + C::XB)));
        const int resident = 256 * ((NFR == 6 || C::THREADS > 256 || PC) ? 2 : (per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu)));
        int ks;
        if (target > 0) ks = (target + base - 1) / base;
        else if (TPB > 1) ks = (256 * (per_cu < 1 ? 1 : per_cu) + base - 1) / base;
        else if (taps == 1 && P.slices_cap > 0) {
            static int pws = -1;                            // Y5M_WGRAD_PW_SLICE_BLOCKS: non-atomic mode is not atomics-bound
            if (pws < 0) { const char* e = getenv("Y5M_WGRAD_PW_SLICE_BLOCKS"); pws = e ? atoi(e) : 384; }
            ks = (pws + base - 1) / base;
        }
        else if (taps == 1) {
            static int pwb = -1;                            // Y5M_WGRAD_PW_BLOCKS: blocks of a pointwise weight gradient
            if (pwb < 0) { const char* e = getenv("Y5M_WGRAD_PW_BLOCKS"); pwb = e ? atoi(e) : 160; }   // (swept 128..768 inside the full step: atomics-bound, fewer is better)
            ks = (pwb + base - 1) / base;
        }
        else if (C::TC <= 16) ks = (2048 + base - 1) / base;
        else {
            // Y5M_WGRAD_RES_PCT: share of the one-round block budget. Alone on the GPU a full round is best; in the step
            // the kernel runs NEXT TO the following layer's BatchNorm backward (engine.py), and half a round -- one
            // block per CU for the 96x48 wave tile -- is: 25 / 38 / 50 / 62 / 75 / 100 / 150 % = +2.3 / +0.2 / 0 / +0.15 /
            // +0.1 / +0.35 / +0.5 ms per step
            static int res_pct = -1;
            if (res_pct < 0) { const char* e = getenv("Y5M_WGRAD_RES_PCT"); res_pct = e ? atoi(e) : 50; }
            ks = (sbk && NFR != 6 && C::THREADS == 256 && !PC ? 1024 : resident) * res_pct / 100 / base;
            if (PC && NFR == 6 && CFR == 6) ks = 256 / base;     // 104 KB of LDS: one block per CU IS the resident round
            if (PC >= 2 && NFR == 6) {                            // three LDS buffers (117 / 156 KB): one block per CU
                static int dmab = -1;                             // Y5M_WGRAD_DMA_BLOCKS: CUs the launch may own (the rest stay with the main stream)
                if (dmab < 0) { const char* e = getenv("Y5M_WGRAD_DMA_BLOCKS"); dmab = e ? atoi(e) : 256; }
                ks = dmab / base;
            }   // floor: never more blocks than fit at once (SB: 4 per CU)
        }
        const int maxks = (chunks + minch - 1) / minch;
        ks = ks > maxks ? maxks : ks;
        P.ksplit = ks < 1 ? 1 : ks;
    }
    if (P.slices_cap > 0) {
        if (P.ksplit * WK > P.slices_cap) P.ksplit = P.slices_cap / WK;
        if (P.ksplit < 1) { y5m_set_error("y5m_wgrad: slices_cap smaller than the K-waves of the tile"); return Y5M_EINVAL; }
    }
    g_plan_slices = P.ksplit * WK;
    if (g_plan_only) return Y5M_OK;
    static int sb = -1;                                   // single LDS buffer (default; Y5M_WGRAD_SB=0: double buffer): in the full step -0.15 ms
    if (sb < 0) { const char* e = getenv("Y5M_WGRAD_SB"); sb = e ? atoi(e) : 1; }
    const bool use_sb = sb && TPB == 1 && !(NFR == 6 && CFR == 6 && WK == 2);   // the 8-wave 192 x 192 tile owns its CU: double buffer, one barrier per chunk
    const size_t red_bytes = WK > 1 ? (size_t)(WK - 1) * WN * WC * NFR * CFR * 4 * 64 * sizeof(float) : 0;
    const size_t tile_bytes = (use_sb ? 1 : 2) * (size_t)(C::YB + C::XB);
    const size_t lds = tile_bytes > red_bytes ? tile_bytes : red_bytes;
    const unsigned grid = (unsigned)(P.tiles_n * P.tiles_c * taps * P.ksplit);
    auto kern = use_sb ? wgrad_kernel<T, WN, WC, WK, CFR, TPB, NFR, true, KX, FP> : wgrad_kernel<T, WN, WC, WK, CFR, TPB, NFR, false, KX, FP>;
    static bool attr = false;
    if (!attr) {
        const int cap = (int)(2 * (size_t)(C::YB + C::XB) > red_bytes ? 2 * (size_t)(C::YB + C::XB) : red_bytes);
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<T, WN, WC, WK, CFR, TPB, NFR, true, KX, FP>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<T, WN, WC, WK, CFR, TPB, NFR, false, KX, FP>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        attr = true;
    }
    if constexpr (PC >= 2 && sizeof(T) == 2 && TPB == 1 && WK == 1 && KX == 1) {
        constexpr int NPW = PC == 2 ? 4 : PC == 3 ? 2 : 1;
        auto dk = wgrad_dma_kernel<WN, WC, CFR, NFR, NPW>;
        static bool dattr = false;
        if (!dattr) {
            (void)hipFuncSetAttribute((const void*)dk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(3 * (size_t)(C::YB + C::XB)));
            dattr = true;
        }
        Y5M_NAME_ONLY(Y5M_OK, "wgrad_dma_kernel<%d,%d,%d,%d,%d>", WN, WC, CFR, NFR, NPW);
        hipLaunchKernelGGL(dk, dim3(grid), dim3(256 + 64 * NPW), 3 * (size_t)(C::YB + C::XB), st, P);
        Y5M_CHECK_LAUNCH("wgrad_dma_kernel");
        return Y5M_OK;
    }
    if constexpr (PC == 1 && sizeof(T) == 2 && TPB == 1 && WK == 1 && KX == 1) {
        auto pk = wgrad_pc_kernel<WN, WC, CFR, NFR>;
        static bool pattr = false;
        if (!pattr) {
            (void)hipFuncSetAttribute((const void*)pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * (size_t)(C::YB + C::XB)));
            pattr = true;
        }
        Y5M_NAME_ONLY(Y5M_OK, "wgrad_pc_kernel<%d,%d,%d,%d>", WN, WC, CFR, NFR);
        hipLaunchKernelGGL(pk, dim3(grid), dim3(512), 2 * (size_t)(C::YB + C::XB), st, P);
        Y5M_CHECK_LAUNCH("wgrad_pc_kernel");
        return Y5M_OK;
    }
    Y5M_NAME_ONLY(Y5M_OK, "wgrad_kernel<%s,%d,%d,%d,%d,%d,%d,%d,%d,%d>", sizeof(T) == 2 ? "bf16" : "f32", WN, WC, WK, CFR, TPB, NFR, (int)use_sb, KX, (int)FP);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), lds, st, P);
    Y5M_CHECK_LAUNCH("wgrad_kernel");
    return Y5M_OK;
}

template <typename T>
static int dispatch_wgrad(WgradParams& P, hipStream_t st) {
    const bool n48 = P.N <= 48, c48 = P.C <= 48, c16 = P.C <= 16;
    static int multitap = -1;                                      // Y5M_WGRAD_MULTITAP=0: one tap per block everywhere
    if (multitap < 0) { const char* e = getenv("Y5M_WGRAD_MULTITAP"); multitap = (e && e[0] == '0') ? 0 : 1; }
    const int taps = P.th * P.tw;
    if (multitap && taps % 9 == 0 && n48 && c16) return launch_wgrad<T, 1, 1, 4, 9, 9>(P, st);      // stem: 48 x (9 taps x 16)
    // (3 taps x 48 channels per block was measured too: the 108 accumulator registers leave one block per
    //  CU and the 48-channel 3x3 layers get SLOWER, 254 -> 447 us; only the stem's 16-channel input pays)
    if (n48 && c16) return launch_wgrad<T, 1, 1, 4, 1>(P, st);     // stem: 48 x 16
    if (n48 && c48) return launch_wgrad<T, 1, 1, 4, 3>(P, st);     // 48 x 48
    if (n48) return launch_wgrad<T, 1, 2, 2, 3>(P, st);            // 48 x 96
    if (c48) return launch_wgrad<T, 2, 1, 2, 3>(P, st);            // 96 x 48
    static int pw8 = -1;                                           // Y5M_WGRAD_PW8: 8-wave 96 x 96 tile (2 K-waves) for pointwise layers
    if (pw8 < 0) { const char* e = getenv("Y5M_WGRAD_PW8"); pw8 = e ? atoi(e) : 0; }
    if (pw8 && taps == 1) return launch_wgrad<T, 2, 2, 2, 3>(P, st);
    // wide layers: 192 x 96 block (wave 96 x 48: 9 transposing reads per 18 MFMAs instead of 12 per 9, and
    // 1.33x fewer staged bytes per MFMA); Y5M_WGRAD_BIG=2: 192 x 192 (wave 96 x 96, one block per CU), 0: off
    static int big = -1;
    if (big < 0) { const char* e = getenv("Y5M_WGRAD_BIG"); big = e ? atoi(e) : 1; }
    if constexpr (sizeof(T) == 2) {
        if (big == 2 && P.N % 192 == 0 && P.C % 192 == 0 && getenv("Y5M_WGRAD_FP") && atoi(getenv("Y5M_WGRAD_FP")) == 1)
            return launch_wgrad<T, 2, 2, 1, 6, 1, 6, 1, true>(P, st);
    }
    if (big == 2 && P.N % 192 == 0 && P.C % 192 == 0) return launch_wgrad<T, 2, 2, 1, 6, 1, 6>(P, st);
    if (big == 3 && P.N % 192 == 0) return launch_wgrad<T, 4, 2, 1, 3, 1, 3>(P, st);        // 192 x 96, 8 waves of 48 x 48
    // Y5M_WGRAD_BIG=4: 192 x 192 block, 8 waves = 2 K-waves of 96 x 96 (227 VGPRs, double-buffered LDS, one block per CU).
    // Measured (round 2, B=64): isolated 585 vs 570 TFLOP/s on 192 -> 192 @ 40x40, 599 vs 517 on 384 -> 384 @ 20x20,
    // 444 vs 355 on the 384 -> 768 stride-2 layer, 461 vs 497 on 192 -> 384 stride 2 -- but 28.95 vs 28.08 ms/step inside
    // the train step, where the weight gradient shares the chip with the BatchNorm backward: off.
    if (big == 4 && P.N % 192 == 0 && P.C % 192 == 0 && taps > 1) return launch_wgrad<T, 2, 2, 2, 6, 1, 6>(P, st);
    // Y5M_WGRAD_BIG=4: 192 x 192 block, 8 waves = 2 K-waves of 96 x 96 (227 VGPRs, double-buffered LDS, one block per CU).
    // Measured (round 2, B=64): isolated 585 vs 570 TFLOP/s on 192 -> 192 @ 40x40, 599 vs 517 on 384 -> 384 @ 20x20,
    // 444 vs 355 on the 384 -> 768 stride-2 layer, 461 vs 497 on 192 -> 384 stride 2 -- but 28.95 vs 28.08 ms/step inside
    // the train step, where the weight gradient shares the chip with the BatchNorm backward: off.
    if (big == 4 && P.N % 192 == 0 && P.C % 192 == 0 && taps > 1) return launch_wgrad<T, 2, 2, 2, 6, 1, 6>(P, st);
    // experiment: the SAME 192 x 96 block tile with 2 K-waves (8 waves of 96 x 48: two waves per SIMD on one block per CU)
    if (big == 5 && P.N % 192 == 0 && taps > 1) return launch_wgrad<T, 2, 2, 2, 3, 1, 6>(P, st);
    if constexpr (sizeof(T) == 2) {
        static int fp = -1;                 // Y5M_WGRAD_FP: 0 = compiler schedule, 1 = fragment pipelining, 2 = + 128-pixel chunks
        if (fp < 0) { const char* e = getenv("Y5M_WGRAD_FP"); fp = e ? atoi(e) : 0; }
        if (big && P.N % 192 == 0 && fp == 1) return launch_wgrad<T, 2, 2, 1, 3, 1, 6, 1, true>(P, st);
        if (big && P.N % 192 == 0 && fp == 2) return launch_wgrad<T, 2, 2, 1, 3, 1, 6, 2, true>(P, st);
        if (big && P.N % 192 == 0 && fp == 3) return launch_wgrad<T, 2, 2, 1, 3, 1, 6, 2, false>(P, st);
    }
    if constexpr (sizeof(T) == 2) {
        // Y5M_WGRAD_PC: producer / consumer form (bit 0: the 192 x 96 tile, bit 1: the 96 x 96 tile; bit 2: also pointwise layers;
        // bit 3: a 192 x 192 tile where both channel counts allow it; bits 4 / 5: wgrad_dma_kernel with 4 / 2 producer waves, bit 6 (with 4 or 5): one)
        static int pc = -1;
        if (pc < 0) { const char* e = getenv("Y5M_WGRAD_PC"); pc = e ? atoi(e) : 0; }
        static int pcsel = -1;              // Y5M_WGRAD_PC_SEL: 0 = every eligible layer, 1 = stride-2 layers only, 2 = C >= 384 only, 3 = either
        if (pcsel < 0) { const char* e = getenv("Y5M_WGRAD_PC_SEL"); pcsel = e ? atoi(e) : 0; }
        const bool sel = pcsel == 0 || ((pcsel & 1) && P.sy == 2) || ((pcsel & 2) && P.C >= 384);
        if (sel && (pc & 48) && P.slices_cap <= 0 && (taps > 1 || (pc & 4)) && P.C % 16 == 0 && P.N % 16 == 0) {
            // bits 4 / 5: LDS-DMA producers (4 / 2 producer waves); full channel tiles only
            const bool two = (pc & 32) != 0;
            if ((pc & 64) && (pc & 8) && P.N % 192 == 0 && P.C % 192 == 0) return launch_wgrad<T, 2, 2, 1, 6, 1, 6, 1, false, 4>(P, st);   // bit 6: ONE producer wave
            if ((pc & 64) && (pc & 1) && big && P.N % 192 == 0 && P.C % 96 == 0) return launch_wgrad<T, 2, 2, 1, 3, 1, 6, 1, false, 4>(P, st);
            if ((pc & 8) && P.N % 192 == 0 && P.C % 192 == 0)
                return two ? launch_wgrad<T, 2, 2, 1, 6, 1, 6, 1, false, 3>(P, st) : launch_wgrad<T, 2, 2, 1, 6, 1, 6, 1, false, 2>(P, st);
            if ((pc & 1) && big && P.N % 192 == 0 && P.C % 96 == 0)
                return two ? launch_wgrad<T, 2, 2, 1, 3, 1, 6, 1, false, 3>(P, st) : launch_wgrad<T, 2, 2, 1, 3, 1, 6, 1, false, 2>(P, st);
            if ((pc & 2) && P.N % 96 == 0 && P.C % 96 == 0 && !(big && P.N % 192 == 0))
                return two ? launch_wgrad<T, 2, 2, 1, 3, 1, 3, 1, false, 3>(P, st) : launch_wgrad<T, 2, 2, 1, 3, 1, 3, 1, false, 2>(P, st);
        }
        if (sel && pc && P.slices_cap <= 0 && (taps > 1 || (pc & 4))) {
            if ((pc & 8) && P.N % 192 == 0 && P.C % 192 == 0) return launch_wgrad<T, 2, 2, 1, 6, 1, 6, 1, false, 1>(P, st);
            if ((pc & 1) && big && P.N % 192 == 0) return launch_wgrad<T, 2, 2, 1, 3, 1, 6, 1, false, 1>(P, st);
            if ((pc & 2) && !(big && P.N % 192 == 0)) return launch_wgrad<T, 2, 2, 1, 3, 1, 3, 1, false, 1>(P, st);
        }
    }
    if (big && P.N % 192 == 0) return launch_wgrad<T, 2, 2, 1, 3, 1, 6>(P, st);
    return launch_wgrad<T, 2, 2, 1, 3>(P, st);                     // 96 x 96
}

extern "C" int y5m_wgrad_kernel_name(const y5m_wgrad_args* args, int dtype, char* buf, int n) {
    y5m_name_only = 1;
    y5m_name_buf[0] = 0;
    const int rc = y5m_wgrad(args, dtype, nullptr);
    y5m_name_only = 0;
    if (rc != Y5M_OK) return rc;
    snprintf(buf, (size_t)n, "%s", y5m_name_buf);
    return Y5M_OK;
}

extern "C" int y5m_wgrad_slices(const y5m_wgrad_args* args, int dtype) {
    g_plan_only = true;
    g_plan_slices = 0;
    const int rc = y5m_wgrad(args, dtype, nullptr);
    g_plan_only = false;
    return rc == Y5M_OK ? g_plan_slices : rc;
}

extern "C" int y5m_wgrad(const y5m_wgrad_args* args, int dtype, void* stream) {
    WgradParams P = *args;
    const int CH = dtype == Y5M_BF16 ? 8 : 4;
    Y5M_REQUIRE(dtype == Y5M_F32 || dtype == Y5M_BF16, "dtype");
    Y5M_REQUIRE(P.zeros != nullptr, "args.zeros (16 zero bytes in device memory) is required");
    Y5M_REQUIRE(P.C % CH == 0 && P.N % CH == 0 && P.ldx % CH == 0 && P.lddy % CH == 0, "channel counts must be multiples of 16 bytes");
    Y5M_REQUIRE(P.M == P.B * P.Hg * P.Wg && P.M > 0, "M");
    {
        // buffer-resource addressing: 32-bit byte offsets with the top bit reserved for "out of range"
        const size_t esz = dtype == Y5M_BF16 ? 2 : 4;
        Y5M_REQUIRE((size_t)P.M * P.lddy * esz < (1ull << 31), "dY view must be < 2 GiB");
        Y5M_REQUIRE((size_t)P.B * P.Hin * P.Win * P.ldx * esz < (1ull << 31), "X view must be < 2 GiB");
        Y5M_REQUIRE((size_t)P.B * P.Hin * P.Win < (1ull << 24), "X must have < 2^24 pixels");
    }
    hipStream_t st = y5m_stream(stream);
    if (dtype == Y5M_BF16) return dispatch_wgrad<bf16_t>(P, st);
    return dispatch_wgrad<float>(P, st);
}
