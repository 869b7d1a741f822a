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
// The forms that were tried and lost inside the train step (row-of-taps, fragment pipelining, 128-pixel chunks, 8-wave tiles,
// producer / consumer workgroups with register or LDS-DMA staging -- 0.82-0.95 PFLOP/s alone on the chip, slower in the step
// forked AND inline --, non-atomic slices) are recorded in NOTES.md.
#include "y5m_conv.h"
#include <stdlib.h>
#include <string.h>

#define WG_THREADS 256

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef y5m_wgrad_args WgradParams;
// launch geometry of the last y5m_wgrad call of this thread (also filled by a name-only call): y5m_wgrad_geometry
static thread_local int32_t g_wg_geom[8];

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

template <typename T, int WN, int WC, int WK, int CFR, int TPB, int NFR>
struct WgCfg {
    static constexpr int TN = WN * 16 * NFR;                // channels of dY per block
    static constexpr int TC = WC * 16 * CFR;           // (virtual) channels of X per block = TPB taps x CBLK
    static constexpr int CBLK = TC / TPB;              // real channels per tap in the tile
    static_assert(TC % TPB == 0 && CBLK % 16 == 0, "a 16-channel fragment must not straddle two taps");
    static constexpr bool BF = sizeof(T) == 2;
    static constexpr int KCH = BF ? 32 * (WK > 2 ? WK : 2) : 32;   // pixels per LDS chunk
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
template <typename T, int WN, int WC, int WK, int CFR, int TPB, int NFR, bool SB>
__global__ __launch_bounds__(WN * WC * WK * 64) void wgrad_kernel(const WgradParams P) {
    using C = WgCfg<T, WN, WC, WK, CFR, TPB, NFR>;
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
        if constexpr (C::BF) {
            constexpr int KS = KCH / 32 / WK;           // k-steps (32 pixels) of this wave per chunk
            const int lane_off = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ks = wk * KS + s;
                uint4 ya[NFR], xb[CFR];
#pragma unroll
                for (int a = 0; a < NFR; ++a) {
                    const unsigned char* sub = Ys + (ks * (C::TN / 16) + wn * NFR + a) * WG_SUB;
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
                for (int a = 0; a < NFR; ++a)
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

    if (ch_lo < ch_hi) {
        load_chunk();
        store_chunk(0);
        __syncthreads();
        int cur = 0;
        for (int chk = ch_lo; chk < ch_hi; ++chk) {
            const bool more = chk + 1 < ch_hi;
            if (more) load_chunk();
            compute(cur);
            if constexpr (SB) {
                __syncthreads();                     // every wave is done reading the only buffer
                if (more) store_chunk(0);
                __syncthreads();
            } else {
                if (more) store_chunk(cur ^ 1);
                __syncthreads();
                cur ^= 1;
            }
        }
    }
    // K-waves (WK > 1: the waves of a block that split the pixels of every chunk) first add their partial tiles in
    // LDS, so the block issues ONE set of atomics instead of WK (pointwise layers are bound by those atomics)
    bool writer = true;
    if constexpr (WK > 1) {
        {
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
    if (ch_lo < ch_hi && writer) {
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
                        atomicAdd(d, acc[a][b][r]);
                    }
                }
            }
    }
}

template <typename T, int WN, int WC, int WK, int CFR, int TPB = 1, int NFR = 3>
static int launch_wgrad(WgradParams& P, hipStream_t st) {
    using C = WgCfg<T, WN, WC, WK, CFR, TPB, NFR>;
    P.tiles_n = (P.N + C::TN - 1) / C::TN;
    P.tiles_c = (P.C + C::CBLK - 1) / C::CBLK;
    const int taps = (P.th * P.tw) / TPB;                 // tap groups (blocks along the tap axis)
    const int chunks = (P.M + C::KCH - 1) / C::KCH;
    if (P.ksplit <= 0) {
        // fill the chip but keep >= 8 chunks per block so the prologue amortises
        static int target = -1, minch = -1;     // Y5M_WGRAD_BLOCKS / Y5M_WGRAD_MINCH: tuning knobs
        if (target < 0) { const char* e = getenv("Y5M_WGRAD_BLOCKS"); target = e ? atoi(e) : 0; }
        if (minch < 0) { const char* e = getenv("Y5M_WGRAD_MINCH"); minch = e ? atoi(e) : 8; }
        const int base = P.tiles_n * P.tiles_c * taps;
        // measured (MI355X, B=64): every split adds one f32 atomic per output element, so pointwise layers
        // (few, large output tiles) want ~1 block per CU. 3x3 layers: fill the chip with ONE resident round --
        // blocks <= resident capacity (a second, nearly empty round cost 20 %: 137 -> 111 us on the 192x192
        // layers when the count dropped from 522 to <= 512); capacity = blocks per CU (LDS / VGPR limited:
        // 2 for the 96x48 wave tile, 3 otherwise) x 256 CUs. The 48x16 stem tile (one tap per block) wants more.
        const int per_cu = (int)((160 * 1024) / (2 * (size_t)(C::YB + C::XB)));
        const int resident = 256 * (NFR == 6 ? 2 : (per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu)));
        int ks;
        if (target > 0) ks = (target + base - 1) / base;
        else if (TPB > 1) ks = (256 * (per_cu < 1 ? 1 : per_cu) + base - 1) / base;
        else if (taps == 1) {
            static int pwb = -1;                            // Y5M_WGRAD_PW_BLOCKS: blocks of a pointwise weight gradient
            if (pwb < 0) { const char* e = getenv("Y5M_WGRAD_PW_BLOCKS"); pwb = e ? atoi(e) : 224; }   // (swept inside the full step; re-swept after the fused pointwise backward took the big 1x1 layers: 128 / 160 / 224 / 256 / 320 / 448 = +0.2 / +0.1 / 0 / +0.05 / +0.05 / +0.2 ms)
            ks = (pwb + base - 1) / base;
        }
        else if (C::TC <= 16) ks = (2048 + base - 1) / base;
        else {
            // Y5M_WGRAD_RES_PCT: share of the one-round block budget. Alone on the GPU a full round is best; next to the
            // following layer's BatchNorm backward on the main stream (engine.py) half a round -- one block per CU for the
            // 96x48 wave tile -- is: 25 / 38 / 50 / 62 / 75 / 100 / 150 % = +2.3 / +0.2 / 0 / +0.15 / +0.1 / +0.35 / +0.5 ms per step
            static int res_pct = -1;
            if (res_pct < 0) { const char* e = getenv("Y5M_WGRAD_RES_PCT"); res_pct = e ? atoi(e) : 50; }
            // Y5M_WGRAD_ROUND: 0 = budget / tiles rounded down, 1 = to the nearest, 2 (default) = down, except that ONE block per tile
            // becomes two when the budget covers 1.5 tiles' worth: 384 -> 768 stride 2 has 144 tiles, and 256 / 144 rounded down left it
            // on 144 of the 256 CUs (362 us alone, 276 with two; MIOpen's best solver 238). Inside the step: 2 = -0.05 ms against 0 (three
            // alternating rounds), 1 = +0.2 ms (the 72-tile layers go from 216 to 288 blocks: fewer blocks is better next to the main stream).
            static int rnd = -1;
            if (rnd < 0) { const char* e = getenv("Y5M_WGRAD_ROUND"); rnd = e ? atoi(e) : 2; }
            const int budget = (NFR != 6 ? 1024 : resident) * res_pct / 100;
            ks = (budget + (rnd == 1 ? base / 2 : 0)) / base;
            if (rnd == 2 && ks == 1 && 2 * budget >= 3 * base) ks = 2;
        }
        const int maxks = (chunks + minch - 1) / minch;
        ks = ks > maxks ? maxks : ks;
        P.ksplit = ks < 1 ? 1 : ks;
    }
    const unsigned grid = (unsigned)(P.tiles_n * P.tiles_c * taps * P.ksplit);
    {
        // one LDS buffer (two barriers per chunk, half the LDS: more resident blocks; -0.15 ms in the full step against two
        // buffers); the multi-tap stem tile keeps two
        constexpr bool SB = TPB == 1;
        const size_t red_bytes = WK > 1 ? (size_t)(WK - 1) * WN * WC * NFR * CFR * 4 * 64 * sizeof(float) : 0;
        const size_t tile_bytes = (SB ? 1 : 2) * (size_t)(C::YB + C::XB);
        const size_t lds = tile_bytes > red_bytes ? tile_bytes : red_bytes;
        auto kern = wgrad_kernel<T, WN, WC, WK, CFR, TPB, NFR, SB>;
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        const int32_t geom[8] = {P.tiles_n, P.tiles_c, taps, P.ksplit, (int32_t)grid, C::TN, C::CBLK, C::KCH};
        memcpy(g_wg_geom, geom, sizeof(geom));
        Y5M_NAME_ONLY(Y5M_OK, "wgrad_kernel<%s,%d,%d,%d,%d,%d,%d,%d>", sizeof(T) == 2 ? "bf16" : "f32", WN, WC, WK, CFR, TPB, NFR, (int)SB);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), lds, st, P);
        Y5M_CHECK_LAUNCH("wgrad_kernel");
        return Y5M_OK;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad_rows_kernel: the 3x3 weight gradients of the layers with 48 INPUT channels (48 -> 48 stride 1 and 48 -> 96 stride 2 at
// 160x160; bf16). With one tap per block (wgrad_kernel) these launches re-stage dY and X nine times through L2 and are bound by
// that traffic (48 -> 96: 4.2 GB of L2 reads for 0.94 GB of operands, 405 us alone on the chip). Here a block owns ONE KERNEL ROW ty:
//   * pixels are taken in RUNS of 32 consecutive output pixels of one output row; the X pixels the run's three horizontal taps
//     touch are 32*S + 2 CONSECUTIVE input pixels of ONE input row -- staged once, as they lie in memory (no per-tap gather), and
//     tap tx reads them from LDS at pixel offset tx with pixel stride S (every lane of a transposing read supplies its own row
//     address, so offset and stride are free);
//   * the run's dY tile is staged once for the three taps: dY is read 3x instead of 9x, X 1.5x (stride 2) / 3x (stride 1);
//   * a wave holds 3 taps x 48 x 48 outputs (27 accumulator fragments); the 4 waves are WN (48 dY channels each) x WK (runs of a
//     chunk), WK partial tiles are added in LDS before ONE set of atomics per block.
template <int WN, int WK, int S>
struct WrCfg {
    static constexpr int N = WN * 48;                   // dY channels (all of them in one block)
    static constexpr int NPX = 32 * S + 2;              // X pixels of a run
    static constexpr int XG = NPX * 32 + 64;            // bytes of one 16-channel group of a run's X pixels (+64: groups on other banks)
    static constexpr int YB = WK * (N / 16) * WG_SUB;   // dY tile: [run][16-channel group][32 pixel][16 channel]
    static constexpr int XB = WK * 3 * XG;              // X tile:  [run][16-channel group][NPX pixel][16 channel]
    static constexpr int YP = WK * 32 * (N / 8);        // 16-byte pieces per chunk
    static constexpr int XP = WK * NPX * 6;
    static constexpr int NLDY = (YP + 255) / 256, NLDX = (XP + 255) / 256;
    static constexpr int TILE_F = 27 * 4 * 64;          // floats of one wave's accumulators
    static constexpr int RED = (WK / 2) * WN * TILE_F * 4;
    static constexpr int LDS = (YB + XB) > RED ? (YB + XB) : RED;
};

// R4: the round-4 form of the address arithmetic (scalar run state; made from the static ISA audit and not yet measured on
// hardware: selected by Y5M_R4_KERNELS bit 0, y5m_common.h); false = the round-3 form (per-piece divisions), hardware-verified.
template <int WN, int WK, int S, bool R4>
__global__ __launch_bounds__(256) void wgrad_rows_kernel(const WgradParams P, const int spr, const int nruns) {
    using C = WrCfg<WN, WK, S>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid % WN, wk = wid / WN;
    // the three kernel-row blocks of one pixel range are consecutive logical ids on one XCD (they share dY through its L2)
    const int nblk = gridDim.x, hb = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = hb & 7;
    const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (hb >> 3);
    const int ty = bid % 3, ksp = bid / 3;
    const int chunks_total = (nruns + WK - 1) / WK;
    const int per = (chunks_total + P.ksplit - 1) / P.ksplit;
    const int ch_lo = ksp * per, ch_hi = min(chunks_total, ch_lo + per);

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(P.dy), 0, (unsigned)((size_t)P.M * P.lddy * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(P.x), 0, (unsigned)((size_t)P.B * P.Hin * P.Win * P.ldx * 2), 0x00020000);
    // piece -> (run of the chunk, pixel of the run, 16-byte channel piece): loop invariant
    int yks[C::NLDY], ypx[C::NLDY], ycc[C::NLDY], xks[C::NLDX], xpx[C::NLDX], xcc[C::NLDX];
#pragma unroll
    for (int i = 0; i < C::NLDY; ++i) {
        const int q = tid + 256 * i;
        yks[i] = q < C::YP ? q / (32 * (C::N / 8)) : -1;
        const int rem = q % (32 * (C::N / 8));
        ypx[i] = rem / (C::N / 8);
        ycc[i] = rem % (C::N / 8);
    }
#pragma unroll
    for (int i = 0; i < C::NLDX; ++i) {
        const int q = tid + 256 * i;
        xks[i] = q < C::XP ? q / (C::NPX * 6) : -1;
        const int rem = q % (C::NPX * 6);
        xpx[i] = rem / 6;
        xcc[i] = rem % 6;
    }
    const float rcpS = 1.0f / (float)spr, rcpH = 1.0f / (float)P.Hg;
    const unsigned ldyb = (unsigned)(P.lddy * 2), ldxb = (unsigned)(P.ldx * 2);
    const int dh = P.dh0 + ty * P.dhs;
    u32x4 ry[C::NLDY], rx[C::NLDX];
    // Addresses (round 4: tools/isa_audit.py counted 213-321 VALU + 111-136 SALU per chunk against 27 MFMAs -- two divisions, the
    // bounds tests and the multiplies PER 16-BYTE PIECE AND LANE). A chunk is WK consecutive runs, and everything that depends on
    // the run -- (image, output row, run of the row), the input row of this block's kernel row, their validity -- is the same
    // for all lanes: it lives in scalar registers, advances by WK runs per chunk with adds and a carry, and a piece only SELECTS
    // its run's base offset / limit (yks / xks are loop invariant) and adds its own loop-invariant lane constant.
    unsigned ycst[C::NLDY], xcst[C::NLDX];
#pragma unroll
    for (int i = 0; i < C::NLDY; ++i) ycst[i] = (unsigned)ypx[i] * ldyb + (unsigned)(ycc[i] * 16);
#pragma unroll
    for (int i = 0; i < C::NLDX; ++i) xcst[i] = (unsigned)xpx[i] * ldxb + (unsigned)(xcc[i] * 16);
    // run 0 of the NEXT chunk to load: its index su0 and, advanced run by run with adds (a run that starts a new output row
    // recomputes them: one uniform branch, taken once per row): the run's number in its row (sj), its output row and image (soy,
    // sb), the byte offset of its first dY pixel (sybase), of the X pixel its tap 0 reads (sxbase; garbage while the input row is
    // outside the image: sxix then makes every pixel fail its bounds test), that pixel's column (sxix, or a huge value) and the
    // pixels left in the output row (sylim)
    int su0, sj, soy, sb, sylim, sxix;
    unsigned sybase, sxbase;
    auto row_start = [&]() __attribute__((always_inline)) {           // the scalars of run 0 of output row (sb, soy)
        const int iy = soy * P.sy + dh;
        sybase = (unsigned)((sb * P.Hg + soy) * P.Wg) * ldyb;
        sxbase = ((unsigned)((sb * P.Hin + iy) * P.Win) + (unsigned)P.dw0) * ldxb;
        sxix = (unsigned)iy < (unsigned)P.Hin ? P.dw0 : 0x40000000;
        sylim = P.Wg;
        sj = 0;
    };
    const unsigned ystep = 32u * ldyb, xstep = (unsigned)(32 * P.sx) * ldxb;
    su0 = sj = soy = sb = sylim = sxix = 0; sybase = sxbase = 0u;
    if constexpr (R4) {
        su0 = __builtin_amdgcn_readfirstlane(ch_lo * WK);
        int row, j, b, oy;
        fast_divmod(su0, spr, rcpS, row, j);
        fast_divmod(row, P.Hg, rcpH, b, oy);
        j = __builtin_amdgcn_readfirstlane(j);
        soy = __builtin_amdgcn_readfirstlane(oy); sb = __builtin_amdgcn_readfirstlane(b);
        row_start();
        sj = j; sybase += (unsigned)j * ystep; sxbase += (unsigned)j * xstep; sylim -= 32 * j;
        if (sxix != 0x40000000) sxix += 32 * P.sx * j;
    }
    auto load_next_r4 = [&]() __attribute__((always_inline)) {
        unsigned ybase[WK], xbase[WK];
        int ylim[WK], xix0[WK];
#pragma unroll
        for (int k = 0; k < WK; ++k) {
            const bool valid = su0 + k < nruns;
            ybase[k] = sybase;
            ylim[k] = valid ? sylim : 0;                                  // a piece is inside the row while its pixel < ylim
            xix0[k] = valid ? sxix : 0x40000000;                           // (no input row / no such run: every pixel out of range)
            xbase[k] = sxbase;
            // the next run (uniform: scalar adds and one scalar branch)
            if (++sj == spr) {
                if (++soy == P.Hg) { soy = 0; ++sb; }
                row_start();
            } else {
                sybase += ystep; sxbase += xstep; sylim -= 32;
                sxix = sxix != 0x40000000 ? sxix + 32 * P.sx : sxix;
            }
        }
        su0 += WK;
        // piece i of a thread is piece tid + 256 i of the chunk: its run is one of the (at most two) runs that 256-piece window
        // touches -- known at compile time, so a piece selects between two scalars, whatever WK is
        constexpr int PERY = 32 * (C::N / 8), PERX = C::NPX * 6;
#pragma unroll
        for (int i = 0; i < C::NLDY; ++i) {
            const int ka = (256 * i) / PERY < WK ? (256 * i) / PERY : WK - 1, kb = (256 * i + 255) / PERY < WK ? (256 * i + 255) / PERY : WK - 1;
            unsigned base = ybase[ka];
            int lim = ylim[ka];
#pragma unroll
            for (int k = ka + 1; k <= kb; ++k) { base = yks[i] == k ? ybase[k] : base; lim = yks[i] == k ? ylim[k] : lim; }
            const bool ok = yks[i] >= 0 && ypx[i] < lim;
            ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, ok ? base + ycst[i] : OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < C::NLDX; ++i) {
            const int ka = (256 * i) / PERX < WK ? (256 * i) / PERX : WK - 1, kb = (256 * i + 255) / PERX < WK ? (256 * i + 255) / PERX : WK - 1;
            unsigned base = xbase[ka];
            int ix0 = xix0[ka];
#pragma unroll
            for (int k = ka + 1; k <= kb; ++k) { base = xks[i] == k ? xbase[k] : base; ix0 = xks[i] == k ? xix0[k] : ix0; }
            const bool ok = xks[i] >= 0 && (unsigned)(ix0 + xpx[i]) < (unsigned)P.Win;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? base + xcst[i] : OOB, 0, 0);
        }
    };
    // round-3 form: (run, row, image) of every 16-byte piece recomputed per lane with two divisions
    auto load_chunk_r3 = [&](int chk) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::NLDY; ++i) {
            const int u = chk * WK + yks[i];
            int row, j;
            fast_divmod(u, spr, rcpS, row, j);                        // row = b * Hg + oy
            const int ox = 32 * j + ypx[i];
            const bool ok = yks[i] >= 0 && u < nruns && ox < P.Wg;
            const unsigned m = (unsigned)(row * P.Wg + ox);
            ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, ok ? m * ldyb + (unsigned)(ycc[i] * 16) : OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < C::NLDX; ++i) {
            const int u = chk * WK + xks[i];
            int row, j, b, oy;
            fast_divmod(u, spr, rcpS, row, j);
            fast_divmod(row, P.Hg, rcpH, b, oy);
            const int iy = oy * P.sy + dh, ix = 32 * j * P.sx + P.dw0 + xpx[i];
            const bool ok = xks[i] >= 0 && u < nruns && (unsigned)iy < (unsigned)P.Hin && (unsigned)ix < (unsigned)P.Win;
            const unsigned pix = __umul24((unsigned)(b * P.Hin + iy), (unsigned)P.Win) + (unsigned)ix;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? __umul24(pix, ldxb) + (unsigned)(xcc[i] * 16) : OOB, 0, 0);
        }
    };
    auto load_chunk = [&](int chk) __attribute__((always_inline)) {   // chunk chk (R4: chunks are taken in order, chk is implied)
        if constexpr (R4) load_next_r4(); else load_chunk_r3(chk);
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
        unsigned char* Ys = smem;
        unsigned char* Xs = smem + C::YB;
#pragma unroll
        for (int i = 0; i < C::NLDY; ++i)
            if (yks[i] >= 0)
                *reinterpret_cast<u32x4*>(Ys + (yks[i] * (C::N / 16) + (ycc[i] >> 1)) * WG_SUB + ypx[i] * 32 + (ycc[i] & 1) * 16) = ry[i];
#pragma unroll
        for (int i = 0; i < C::NLDX; ++i)
            if (xks[i] >= 0)
                *reinterpret_cast<u32x4*>(Xs + (xks[i] * 3 + (xcc[i] >> 1)) * C::XG + xpx[i] * 32 + (xcc[i] & 1) * 16) = rx[i];
    };

    f32x4 acc[3][3][3];                                   // [tap tx][dY fragment][X fragment]
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[t][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int lrow = 4 * (lane >> 4) + ((lane & 15) >> 2), lb = (lane & 3) * 8;
    const int y_off = lrow * 32 + lb, x_off = S * lrow * 32 + lb;
    auto tr8 = [&](const unsigned char* p, int second) __attribute__((always_inline)) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p + second));
        return make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
    };
    auto compute = [&]() __attribute__((always_inline)) {
        const unsigned char* Ys = smem + (wk * (C::N / 16) + wn * 3) * WG_SUB + y_off;
        const unsigned char* Xs = smem + C::YB + wk * 3 * C::XG + x_off;
        uint4 ya[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) ya[a] = tr8(Ys + a * WG_SUB, 512);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            uint4 xb[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) xb[b] = tr8(Xs + b * C::XG + t * 32, S * 512);
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ya[a]),
                                                                           __builtin_bit_cast(bf16x8_t, xb[b]), acc[t][a][b], 0, 0, 0);
        }
    };

    if (ch_lo < ch_hi) {
        load_chunk(ch_lo);
        store_chunk();
        __syncthreads();
        for (int chk = ch_lo; chk < ch_hi; ++chk) {
            const bool more = chk + 1 < ch_hi;
            if (more) load_chunk(chk + 1);
            compute();
            __syncthreads();
            if (more) store_chunk();
            __syncthreads();
        }
    }
    // the WK waves of a channel group add their tiles pairwise in LDS (wk >= st writes, wk - st adds), then wk == 0 issues the atomics
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int st = WK / 2; st >= 1; st >>= 1) {
        __syncthreads();
        if (wk >= st && wk < 2 * st) {
            float* d = red + ((size_t)(wk - st) * WN + wn) * C::TILE_F + lane;
#pragma unroll
            for (int f = 0; f < 27; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) d[(f * 4 + r) * 64] = acc[f / 9][(f / 3) % 3][f % 3][r];
        }
        __syncthreads();
        if (wk < st) {
            const float* sp = red + ((size_t)wk * WN + wn) * C::TILE_F + lane;
#pragma unroll
            for (int f = 0; f < 27; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[f / 9][(f / 3) % 3][f % 3][r] += sp[(f * 4 + r) * 64];
        }
    }
    if (ch_lo < ch_hi && wk == 0) {
        const int i = lane & 15, g = lane >> 4;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = wn * 48 + a * 16 + g * 4 + r;
                        atomicAdd(P.dwgt + (size_t)n * P.lddw + (ty * 3 + t) * P.C + b * 16 + i, acc[t][a][b][r]);
                    }
    }
}

template <int WN, int WK, int S, bool R4>
static int launch_wgrad_rows_form(WgradParams& P, hipStream_t st) {
    using C = WrCfg<WN, WK, S>;
    const int spr = (P.Wg + 31) / 32;                    // runs per output row
    const int nruns = P.B * P.Hg * spr;
    const int chunks = (nruns + WK - 1) / WK;
    if (P.ksplit <= 0) {
        static int target = -1;                           // Y5M_WGRAD_ROWS_BLOCKS: blocks of a launch (3 kernel rows x pixel ranges)
        if (target < 0) { const char* e = getenv("Y5M_WGRAD_ROWS_BLOCKS"); target = e ? atoi(e) : 512; }
        int ks = target / 3;
        const int maxks = (chunks + 7) / 8;               // >= 8 chunks per block
        ks = ks > maxks ? maxks : ks;
        P.ksplit = ks < 1 ? 1 : ks;
    }
    P.tiles_n = P.tiles_c = 1;
    auto kern = wgrad_rows_kernel<WN, WK, S, R4>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        attr = true;
    }
    const int32_t geom[8] = {1, 1, 3, P.ksplit, 3 * P.ksplit, C::N, 48, 32 * WK};       // (one kernel ROW of three taps per block)
    memcpy(g_wg_geom, geom, sizeof(geom));
    Y5M_NAME_ONLY(Y5M_OK, "wgrad_rows_kernel<%d,%d,%d,%d>", WN, WK, S, (int)R4);
    hipLaunchKernelGGL(kern, dim3((unsigned)(3 * P.ksplit)), dim3(256), C::LDS, st, P, spr, nruns);
    Y5M_CHECK_LAUNCH("wgrad_rows_kernel");
    return Y5M_OK;
}

template <int WN, int WK, int S>
static int launch_wgrad_rows(WgradParams& P, hipStream_t st) {
    return (y5m_r4_forms() & Y5M_R4_WGRAD_ROWS) ? launch_wgrad_rows_form<WN, WK, S, true>(P, st)
                                                : launch_wgrad_rows_form<WN, WK, S, false>(P, st);
}

// 1 when the launch qualifies for wgrad_rows_kernel
static bool wgrad_rows_eligible(const WgradParams& P) {
    static int on = -1;                                   // Y5M_WGRAD_ROWS=0: one tap per block (wgrad_kernel) as before (A/B runs)
    if (on < 0) { const char* e = getenv("Y5M_WGRAD_ROWS"); on = (e && e[0] == '0') ? 0 : 1; }
    return on && P.th == 3 && P.tw == 3 && P.dhs == 1 && P.dws == 1 && P.C == 48 && (P.N == 48 || P.N == 96) && P.sy == P.sx &&
           (P.sy == 1 || P.sy == 2) && P.ldx % 8 == 0 && P.lddy % 8 == 0 && P.Hg > 0 && P.Wg > 0 &&
           (long long)P.B * P.Hg * ((P.Wg + 31) / 32) < (1ll << 24);
}

template <typename T>
static int dispatch_wgrad(WgradParams& P, hipStream_t st) {
    const bool n48 = P.N <= 48, c48 = P.C <= 48, c16 = P.C <= 16;
    const int taps = P.th * P.tw;
    if (taps % 9 == 0 && n48 && c16) return launch_wgrad<T, 1, 1, 4, 9, 9>(P, st);      // stem: 48 x (9 taps x 16)
    if constexpr (sizeof(T) == 2) {
        if (wgrad_rows_eligible(P)) {                      // 48 input channels, 3x3: one kernel row per block (wgrad_rows_kernel)
            if (P.N == 48) return P.sy == 1 ? launch_wgrad_rows<1, 4, 1>(P, st) : launch_wgrad_rows<1, 4, 2>(P, st);
            return P.sy == 1 ? launch_wgrad_rows<2, 2, 1>(P, st) : launch_wgrad_rows<2, 2, 2>(P, st);
        }
    }
    // (3 taps x 48 channels per block was measured too: the 108 accumulator registers leave one block per
    //  CU and the 48-channel 3x3 layers get SLOWER, 254 -> 447 us; only the stem's 16-channel input pays)
    if (n48 && c16) return launch_wgrad<T, 1, 1, 4, 1>(P, st);     // 48 x 16
    if (n48 && c48) return launch_wgrad<T, 1, 1, 4, 3>(P, st);     // 48 x 48
    if (n48) return launch_wgrad<T, 1, 2, 2, 3>(P, st);            // 48 x 96
    if (c48) return launch_wgrad<T, 2, 1, 2, 3>(P, st);            // 96 x 48
    // wide layers: 192 x 96 block (wave 96 x 48: 9 transposing reads per 18 MFMAs instead of 12 per 9, and
    // 1.33x fewer staged bytes per MFMA)
    if (P.N % 192 == 0) return launch_wgrad<T, 2, 2, 1, 3, 1, 6>(P, st);
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

// The launch geometry y5m_wgrad would use for these arguments, nothing is launched: out = {n tiles, c tiles, tap groups (blocks along
// the tap axis), pixel-range splits, workgroups, dY channels per block, X channels per block and tap, pixels per LDS chunk}.
// Logical block order = ((range * n_tiles + n) * c_tiles + c) * tap_groups + tap, dealt to the XCDs in 8 contiguous pieces.
extern "C" int y5m_wgrad_geometry(const y5m_wgrad_args* args, int dtype, int32_t out[8]) {
    Y5M_REQUIRE(out != nullptr, "out");
    y5m_name_only = 1;
    const int rc = y5m_wgrad(args, dtype, nullptr);
    y5m_name_only = 0;
    if (rc != Y5M_OK) return rc;
    memcpy(out, g_wg_geom, sizeof(g_wg_geom));
    return Y5M_OK;
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
