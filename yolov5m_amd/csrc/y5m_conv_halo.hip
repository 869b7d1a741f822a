// 3x3 / stride-1 / pad-1 convolution (forward and data gradient) for gfx950, bf16: "halo patch" implicit GEMM.
//
// Why a third conv kernel. The tiled kernel (y5m_conv.hip) re-stages the same input pixels once per tap (9x) through
// VGPRs into LDS (ds_write_b128: 79 B/clk/CU) and synchronises twice per 64-deep K step: its LDS pipe is busier than
// its matrix pipe (966 vs 768 cycles per K step of a 128x192 tile), which is the ~30 % of peak it measures. Here
//   * a tile is 256 CONSECUTIVE output pixels in (b, y, x) raster order; the input pixels all 9 taps of the tile touch
//     are the raster run [m0 - W - 1, m0 + 256 + W + 1): one "patch" of 258 + 2W rows x 64 channels (128 B rows).
//     It is staged ONCE per 64-channel slab and every tap reads it at a row offset dy*W + dx; taps that fall outside
//     the image (or rows behind the tensor) are redirected per lane to a 128-byte zero row;
//   * the weights of one (slab, tap) unit (BN rows x 128 B) stream through a 3-stage LDS ring;
//   * BOTH operands arrive by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction; semantics probed on
//     hardware by tools/probe_lds_dma.hip: destination = M0 + lane*16, out-of-range lanes write zeros). The LDS image
//     is lane-linear, so the bank swizzle is applied to the per-lane SOURCE address and again to the fragment read
//     address. The swizzle is chunk ^= row & 6 (not the tiled kernel's (row>>1)&7): a tap shifts the 16 rows of a
//     fragment by dy*W+dx, and (row>>1)&7 is 2-way conflicted for 3 of 4 row alignments, row&6 for none (bank model
//     of the ds_read_b128 lane groups, MI355X_MICROARCH LDS table, searched exhaustively over linear swizzles);
//   * ONE s_barrier and ONE counted s_waitcnt vmcnt(N) per unit (48 MFMAs per wave at BN = 192): the weights of unit
//     g+2 and a piece of the NEXT slab's (or next tile's) patch are issued at the top of unit g and are only waited
//     for at the end of unit g+1 -- nothing in the loop drains the VMEM queue;
//   * workgroups are persistent (one per CU, 8 waves = 4 pixel groups x 2 channel groups, wave tile 64 x {48,96}):
//     the next tile's first patch and weights are in flight while the current tile finishes, and the epilogue of a
//     tile is issued at the top of the next tile's first unit so its stores drain under that unit's MFMAs.
// Traffic per tile (192 -> 192 channels, W = 40): 130 KB of patch + 663 KB of (L2-resident) weights for 170 MFLOP;
// the tiled kernel staged 1.15 MB of activations + 0.66 MB of weights for the same work, all through ds_write.
#include "y5m_conv.h"

#include <stdlib.h>
#include <string.h>

#define HL_THREADS 512
#define HL_TP 256                 // pixels per tile
#define HL_NS 3                   // weight ring stages

struct HaloArgs {
    int PR8;                      // patch rows, rounded up to the 8-row DMA piece
    int npieces;                  // PR8 / 8
    int NPU;                      // units of a slab that carry a patch piece per wave (<= 8)
    int S;                        // 64-channel slabs (the last one may hold 32 channels)
    int tiles_n, total;           // channel tiles per pixel tile, work items
    int Mtot;                     // B*H*W
    int stat_rows;                // rows of the statistics buffer (4 per pixel tile)
    int dbg;                      // Y5M_HALO_DBG ablation bits (timing experiments only: results are wrong when set)
    unsigned long long* dbg_out;  // bit 256: per-region cycle counters of (block 0, waves 0 and 7)
};

template <int NF>
__device__ __forceinline__ constexpr int hl_pch(int a, int rho) {     // same channel permutation as y5m_conv.hip (cv_pch)
    return (2 * (a >> 1) + 1 < NF) ? (a >> 1) * 32 + (rho >> 2) * 8 + (a & 1) * 4 + (rho & 3) : (a >> 1) * 32 + rho;
}

__device__ __forceinline__ int hl_logical_id(int bid, int nblk) {
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}

// one LDS-DMA instruction: 64 lanes x 16 B from (rsrc, voff + soff) to LDS [lds_addr + lane*16]
__device__ __forceinline__ void hl_dma16(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

__device__ __forceinline__ unsigned long long hl_clock() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define HL_T(i) do { if (G.dbg & 256) { const unsigned long long now_ = hl_clock(); tacc[i] += now_ - tlast; tlast = now_; } } while (0)

template <int NF, int EPI>
__global__ __launch_bounds__(HL_THREADS) void conv_halo_kernel(const ConvParams P, const HaloArgs G) {
    constexpr int BN = 2 * NF * 16;                       // channels per tile (2 channel groups of waves)
    constexpr int WB = BN * 128;                          // bytes of one weight stage
    constexpr int NWP = (BN / 8 + 7) / 8;                 // weight DMA pieces per wave and unit (3 | 2, the second partial)
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid & 3, wn = wid >> 2;
    const int frow = lane & 15, fq = lane >> 4;
    const int W = P.Win, H = P.Hin;
    const int PB = G.PR8 * 128;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    // LDS map: [zero row 128][patch 0][patch 1][weight stage 0..2]
    const unsigned Z_OFF = 0, P_OFF = 128, W_OFF = 128 + 2 * PB;

    if (tid < 8) *reinterpret_cast<uint4*>(smem + Z_OFF + tid * 16) = make_uint4(0u, 0u, 0u, 0u);

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(P.in), 0, (unsigned)((size_t)G.Mtot * P.ldin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(P.w), 0, (unsigned)((size_t)P.Np * P.Kp * 2), 0x00020000);

    // ---- loop-invariant per-lane DMA source offsets ----------------------------------------------------------
    const int dq = lane & 7, dr = lane >> 3;               // a DMA piece = 8 rows x 8 chunks
    unsigned wvoff[NWP];                                   // weights: LDS row r <- channel n0 + perm(r), chunk swizzled
#pragma unroll
    for (int j = 0; j < NWP; ++j) {
        const int r = (wid + 8 * j) * 8 + dr;
        const int rw = r / (NF * 16), rl = r - rw * (NF * 16);
        const int rp = rw * (NF * 16) + hl_pch<NF>(rl >> 4, rl & 15);
        wvoff[j] = (unsigned)((rp * P.Kp + ((dq ^ (r & 6)) << 3)) * 2);
    }
    const unsigned ldb = (unsigned)(P.ldin * 2);

    // ---- fragment read addresses -----------------------------------------------------------------------------
    // weights (MFMA A operand): row nl = wn*NF*16 + a*16 + frow; (nl >> 1) & 7 does not depend on a
    const unsigned wl = (unsigned)((wn * NF * 16 + frow) * 128 + ((fq ^ (frow & 6)) << 4));
    // pixels (MFMA B operand): patch row of tile pixel ml at tap offset 0 is ml + W + 1
    const unsigned prow = (unsigned)((wm * 64 + frow + W + 1) * 128);

    // ---- work items --------------------------------------------------------------------------------------------
    const int nblk = gridDim.x;
    const int lid0 = hl_logical_id(blockIdx.x, nblk);
    // current unit
    int it = lid0, s = 0, t = 0;
    // weight prefetch cursor (two units ahead)
    int pit = lid0, ps = 0, pt = 0;
    unsigned wst0 = W_OFF, wst1 = W_OFF + WB, wst2 = W_OFF + 2 * WB;      // stage of unit g, g+1, g+2
    int pcur = 0;                                                        // patch buffer of the current slab

    auto issue_weights = [&](int xit, int xs, int xt, unsigned stage) __attribute__((always_inline)) {
        const int n0 = (xit % G.tiles_n) * BN;
        const unsigned soff = (unsigned)((n0 * P.Kp + xt * P.Cin + xs * 64) * 2);
#pragma unroll
        for (int j = 0; j < NWP; ++j) {
            if (wid + 8 * j < BN / 8) hl_dma16(rs_w, wvoff[j], soff, lds0 + stage + (unsigned)((wid + 8 * j) * 1024));
        }
    };
    auto issue_patch_piece = [&](int xit, int xs, int pc, int buf) __attribute__((always_inline)) {
        if (pc < G.npieces) {
            const int m0 = (xit / G.tiles_n) * HL_TP;
            const int r = pc * 8 + dr;
            const int pix = m0 - (W + 1) + r;
            const int ch = dq ^ (r & 6);                         // logical 16-byte chunk (8 channels) of this LDS position
            const unsigned off = __umul24((unsigned)pix, ldb) + (unsigned)(xs * 128 + (ch << 4));
            // rows outside the tensor and the channels behind Cin (upper half of a 32-channel last slab) read as zeros
            const bool ok = (unsigned)pix < (unsigned)G.Mtot && xs * 64 + ch * 8 < P.Cin;
            hl_dma16(rs_x, ok ? off : OOB, 0u, lds0 + P_OFF + (unsigned)(buf * PB + pc * 1024));
        }
    };
    auto advance = [&](int& xit, int& xs, int& xt) __attribute__((always_inline)) {
        if (++xt == 9) {
            xt = 0;
            if (++xs == G.S) { xs = 0; xit += nblk; }
        }
    };

    // ---- prologue: first patch, weights of units 0 and 1 --------------------------------------------------------
    for (int i = 0; i < G.NPU; ++i) issue_patch_piece(it, 0, i * 8 + wid, 0);
    issue_weights(pit, ps, pt, wst0);
    advance(pit, ps, pt);
    if (pit < G.total) issue_weights(pit, ps, pt, wst1);
    advance(pit, ps, pt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc[NF][4];
    unsigned vmask[4] = {0u, 0u, 0u, 0u};                  // 9 tap-valid bits per pixel fragment of this lane
    int em0 = 0, en0 = 0, etile = 0;                       // tile whose accumulators are waiting for their epilogue
    bool pending = false;

    auto epilogue = [&]() __attribute__((always_inline)) {
        const int nb = en0 + wn * NF * 16;
        if constexpr (EPI == EPI_RAW_STATS) {
            if (P.stats) {
                // rows past Mtot and out-of-image taps contributed exact zeros: no masking needed
#pragma unroll
                for (int a = 0; a < NF; ++a) {
                    float sv[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float v = acc[a][b][r]; sv[r] += v; ss[r] += v * v; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) { sv[r] += __shfl_xor(sv[r], o, 64); ss[r] += __shfl_xor(ss[r], o, 64); }
                    }
                    if (frow == 0) {
                        const int n = nb + hl_pch<NF>(a, fq * 4);
                        float* row = P.stats + ((size_t)(etile * 4 + wm) * 2) * P.Np + n;
                        *reinterpret_cast<float4*>(row) = make_float4(sv[0], sv[1], sv[2], sv[3]);
                        *reinterpret_cast<float4*>(row + P.Np) = make_float4(ss[0], ss[1], ss[2], ss[3]);
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int m = em0 + wm * 64 + b * 16 + frow;
            if (m >= G.Mtot) continue;
            bf16_t* const ob = reinterpret_cast<bf16_t*>(P.out) + (size_t)m * P.ldout + nb;
            float fv[NF][4];
#pragma unroll
            for (int a = 0; a < NF; ++a) {
                const int nl = hl_pch<NF>(a, fq * 4), n = nb + nl;
                float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
                if constexpr (EPI == EPI_AFFINE_ACT) {
                    const float4 sc = *reinterpret_cast<const float4*>(P.scale + n);
                    const float4 sh = *reinterpret_cast<const float4*>(P.shift + n);
                    v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y;
                    v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                    if (P.act == Y5M_ACT_SILU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
                    }
                    if (P.res) {
                        float rv[4];
                        load4<bf16_t>(reinterpret_cast<const bf16_t*>(P.res) + (size_t)m * P.ldres + n, rv);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rv[r];
                    }
                } else if constexpr (EPI == EPI_DGRAD) {
                    if (P.accumulate) {
                        float ov[4];
                        load4<bf16_t>(P.res ? reinterpret_cast<const bf16_t*>(P.res) + (size_t)m * P.ldres + n : ob + nl, ov);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += ov[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) fv[a][r] = v[r];
            }
#pragma unroll
            for (int a = 0; a < NF; a += 2) {
                if (a + 1 < NF) {
                    typedef unsigned u32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
                    u32x4a8 q4;
                    q4[0] = f32x2_to_bf16x2(fv[a][0], fv[a][1]);
                    q4[1] = f32x2_to_bf16x2(fv[a][2], fv[a][3]);
                    q4[2] = f32x2_to_bf16x2(fv[a + 1][0], fv[a + 1][1]);
                    q4[3] = f32x2_to_bf16x2(fv[a + 1][2], fv[a + 1][3]);
                    *reinterpret_cast<u32x4a8*>(ob + hl_pch<NF>(a, fq * 4)) = q4;
                } else {
                    store4<bf16_t>(ob + hl_pch<NF>(a, fq * 4), fv[a]);
                }
            }
        }
    };

    // ---- fragment addressing ------------------------------------------------------------------------------------
    auto setup_masks = [&](int xit) __attribute__((always_inline)) {
        const int m0 = (xit / G.tiles_n) * HL_TP;
        const float rcpW = 1.0f / (float)W, rcpH = 1.0f / (float)H;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int m = m0 + wm * 64 + b * 16 + frow;
            int tq, x, bi, y;
            fast_divmod(m, W, rcpW, tq, x);
            fast_divmod(tq, H, rcpH, bi, y);
            unsigned mk = 0u;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int dy = P.dh0 + (k / 3) * P.dhs, dx = P.dw0 + (k % 3) * P.dws;
                const bool ok = (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
                mk |= ok ? (1u << k) : 0u;
            }
            vmask[b] = m < G.Mtot ? mk : 0u;
        }
    };
    unsigned a0[4];                                        // LDS byte address of this lane's k-step-0 chunk per pixel fragment
    auto tap_addr = [&](int buf, int xt) __attribute__((always_inline)) {
        const int ta = xt / 3, tb = xt - ta * 3;
        const int d = (P.dh0 + ta * P.dhs) * W + (P.dw0 + tb * P.dws);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned rowb = prow + (unsigned)(b * 2048) + (unsigned)(d * 128);     // row * 128 inside the patch
            const unsigned sw = (rowb >> 3) & 0x60u;                                     // (row & 6) << 4
            const unsigned adr = P_OFF + (unsigned)(buf * PB) + rowb + (((unsigned)fq << 4) ^ sw);
            a0[b] = ((vmask[b] >> xt) & 1u) ? adr : (Z_OFF + ((unsigned)fq << 4));
        }
    };
    auto mfma_step = [&](const uint4 (&xa)[4], const uint4 (&wb)[NF]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, wb[a]), __builtin_bit_cast(bf16x8_t, xa[b]), acc[a][b], 0, 0, 0);
    };

    // pixel fragments of the first unit (the patch is complete and does not change during a slab, so the NEXT unit's
    // pixel fragments are always fetched under the current unit's MFMAs; only the weight fragments of k-step 0 are read
    // after the unit's barrier)
    uint4 xa0[4], xa1[4], wb0[NF], wb1[NF];
    setup_masks(it);
    tap_addr(0, 0);
#pragma unroll
    for (int b = 0; b < 4; ++b) xa0[b] = *reinterpret_cast<const uint4*>(smem + a0[b]);

    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = (G.dbg & 256) ? hl_clock() : 0ull;
    // ---- unit loop ---------------------------------------------------------------------------------------------
    while (it < G.total) {
        const int nks = (P.Cin - s * 64) >= 64 ? 2 : 1;
        const unsigned wbase = wst0 + wl;
        // (a) this unit's remaining fragments
        if (!(G.dbg & 8)) {
#pragma unroll
        for (int a = 0; a < NF; ++a) wb0[a] = *reinterpret_cast<const uint4*>(smem + (wbase + (unsigned)(a * 2048)));
        if (nks == 2) {
#pragma unroll
            for (int b = 0; b < 4; ++b) xa1[b] = *reinterpret_cast<const uint4*>(smem + (a0[b] ^ 64u));
#pragma unroll
            for (int a = 0; a < NF; ++a) wb1[a] = *reinterpret_cast<const uint4*>(smem + ((wbase ^ 64u) + (unsigned)(a * 2048)));
        }
        }
        // (b) prefetch: one patch piece of the next (tile, slab) pair, weights of unit g+2
        int nissued = 0;                                   // VMEM operations this wave issues in this unit (wave-uniform)
        {
            int nit = it, ns = s + 1;
            if (ns == G.S) { ns = 0; nit += nblk; }
            if (t < G.NPU && nit < G.total && t * 8 + wid < G.npieces && !(G.dbg & 2)) {
                issue_patch_piece(nit, ns, t * 8 + wid, pcur ^ 1);
                nissued = 1;
            }
        }
        if (pit < G.total && !(G.dbg & 1)) {
            issue_weights(pit, ps, pt, wst2);
            nissued += (NWP == 3) ? 3 : (wid < 4 ? 2 : 1);
        }
        advance(pit, ps, pt);
        HL_T(0);
        // (c) first unit of a tile: retire the previous tile
        if (s == 0 && t == 0) {
            if (pending) epilogue();
#pragma unroll
            for (int a = 0; a < NF; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            etile = it / G.tiles_n;
            em0 = etile * HL_TP;
            en0 = (it % G.tiles_n) * BN;
            pending = true;
        }
        HL_T(1);
        // (d) k-step 0
        if (!(G.dbg & 4)) mfma_step(xa0, wb0);
        else {
#pragma unroll
            for (int b = 0; b < 4; ++b) asm volatile("" :: "v"(xa0[b].x), "v"(xa0[b].y), "v"(xa0[b].z), "v"(xa0[b].w));
#pragma unroll
            for (int a = 0; a < NF; ++a) asm volatile("" :: "v"(wb0[a].x), "v"(wb0[a].y), "v"(wb0[a].z), "v"(wb0[a].w));
        }
        HL_T(2);
        // (e) pixel fragments of the NEXT unit (its patch buffer is complete: see the wait in (g))
        {
            int uit = it, us = s, ut = t + 1, ubuf = pcur;
            if (ut == 9) {
                ut = 0;
                ubuf ^= 1;
                if (++us == G.S) { us = 0; uit += nblk; }
            }
            if (uit < G.total) {
                if (us == 0 && ut == 0) setup_masks(uit);
                tap_addr(ubuf, ut);
                if (!(G.dbg & 8)) {
#pragma unroll
                for (int b = 0; b < 4; ++b) xa0[b] = *reinterpret_cast<const uint4*>(smem + a0[b]);
                }
            }
        }
        HL_T(3);
        // (f) k-step 1
        if (nks == 2) {
            if (!(G.dbg & 4)) mfma_step(xa1, wb1);
            else {
#pragma unroll
                for (int b = 0; b < 4; ++b) asm volatile("" :: "v"(xa1[b].x), "v"(xa1[b].y), "v"(xa1[b].z), "v"(xa1[b].w));
#pragma unroll
                for (int a = 0; a < NF; ++a) asm volatile("" :: "v"(wb1[a].x), "v"(wb1[a].y), "v"(wb1[a].z), "v"(wb1[a].w));
            }
        }
        HL_T(4);
        // (g) everything issued before this unit has landed; the unit's own prefetches stay in flight
        //     (vmcnt retires loads in order: "at most nissued outstanding" = every older DMA has landed; the epilogue's
        //     own loads / stores, issued after them, only make the wait stricter). lgkmcnt(0): this wave's fragment
        //     reads of the stage that the next unit's DMA overwrites have returned before anyone passes the barrier.
        switch (nissued) {
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
        }
        HL_T(5);
        if (!(G.dbg & 16)) __builtin_amdgcn_s_barrier();
        HL_T(6);
        // (h) next unit
        { const unsigned x = wst0; wst0 = wst1; wst1 = wst2; wst2 = x; }
        if (++t == 9) {
            t = 0;
            pcur ^= 1;
            if (++s == G.S) { s = 0; it += nblk; }
        }
    }
    if (pending) epilogue();
    if ((G.dbg & 256) && blockIdx.x == 0 && (wid == 0 || wid == 7) && lane == 0) {
        HL_T(7);
        for (int i = 0; i < 8; ++i) G.dbg_out[(wid ? 8 : 0) + i] = tacc[i];
    }
}

// ------------------------------------------------------------------------------------------------------------------
static int g_halo = -1;            // Y5M_CONV_HALO=0: keep the 3x3 stride-1 layers on the tiled kernel (A/B runs)
static int g_halo_cus = 0;

static bool halo_geom(const ConvParams& P, int dtype, HaloArgs& G, int& BN) {
    if (g_halo < 0) { const char* e = getenv("Y5M_CONV_HALO"); g_halo = (e && e[0] == '0') ? 0 : 1; }
    if (!g_halo || dtype != Y5M_BF16) return false;
    if (P.th != 3 || P.tw != 3 || P.sy != 1 || P.sx != 1) return false;
    if (!((P.dh0 == -1 && P.dhs == 1) || (P.dh0 == 1 && P.dhs == -1))) return false;
    if (!((P.dw0 == -1 && P.dws == 1) || (P.dw0 == 1 && P.dws == -1))) return false;
    if (P.Hin != P.Hg || P.Win != P.Wg || P.Hout != P.Hg || P.Wout != P.Wg) return false;
    if (P.osy != 1 || P.osx != 1 || P.ooy != 0 || P.oox != 0) return false;
    if (P.epi != EPI_RAW_STATS && P.epi != EPI_AFFINE_ACT && P.epi != EPI_DGRAD) return false;
    if (P.bn_part) return false;
    if (P.Cin < 64 || P.Cin % 32 != 0 || P.ldin % 8 != 0) return false;
    if (P.N % 96 != 0) return false;
    BN = P.N % 192 == 0 ? 192 : 96;
    if (P.Np < (P.N + BN - 1) / BN * BN) return false;
    if (P.ldout % 8 != 0 || (reinterpret_cast<uintptr_t>(P.out) & 15) != 0) return false;      // 16-byte output pieces
    if (P.res && P.ldres % 4 != 0) return false;
    const int S = (P.Cin + 63) / 64;
    if (P.Kp < 8 * P.Cin + S * 64) return false;            // the last unit's 128-byte weight rows stay inside the packed rows
    const long long Mtot = (long long)P.B * P.Hin * P.Win;
    if (Mtot >= (1ll << 24) || (long long)P.ldin * 2 >= (1ll << 24)) return false;           // 24-bit multiply in the patch address
    if (Mtot * P.ldin * 2 >= (1ll << 31)) return false;
    const int PR = HL_TP + 2 * P.Win + 2;
    G.PR8 = (PR + 7) / 8 * 8;
    G.npieces = G.PR8 / 8;
    G.NPU = (G.npieces + 7) / 8;
    if (G.NPU > 8) return false;
    const size_t lds = 128 + 2 * (size_t)G.PR8 * 128 + HL_NS * (size_t)BN * 128;
    if (lds > 160 * 1024) return false;
    G.S = S;
    G.tiles_n = (P.N + BN - 1) / BN;
    const int tiles_m = (int)((Mtot + HL_TP - 1) / HL_TP);
    G.total = tiles_m * G.tiles_n;
    G.Mtot = (int)Mtot;
    G.stat_rows = tiles_m * 4;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("Y5M_HALO_DBG"); dbg = e ? atoi(e) : 0; } G.dbg = dbg; G.dbg_out = nullptr; }
    return true;
}

// rows of the statistics buffer a RAW_STATS launch writes: 4 per 256-pixel tile here (one per wave row), else one per
// 128-pixel tile (tiled and pointwise kernels)
extern "C" int y5m_conv_stats_rows(const y5m_conv_args* args, int dtype) {
    ConvParams P;
    memcpy(&P, args, sizeof(P));
    HaloArgs G;
    int BN;
    if (halo_geom(P, dtype, G, BN)) return G.stat_rows;
    return (P.M + CV_BM - 1) / CV_BM;
}

// 1 when y5m_conv runs this launch on the halo-patch kernel
extern "C" int y5m_conv_is_halo(const y5m_conv_args* args, int dtype) {
    ConvParams P;
    memcpy(&P, args, sizeof(P));
    HaloArgs G;
    int BN;
    return halo_geom(P, dtype, G, BN) ? 1 : 0;
}

template <int NF, int EPI>
static int launch_halo(const ConvParams& P, const HaloArgs& G, hipStream_t st) {
    constexpr int BN = 2 * NF * 16;
    const size_t lds = 128 + 2 * (size_t)G.PR8 * 128 + HL_NS * (size_t)BN * 128;
    auto kern = conv_halo_kernel<NF, EPI>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    if (g_halo_cus <= 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_halo_cus = prop.multiProcessorCount;
        if (g_halo_cus <= 0) g_halo_cus = 256;
    }
    const int grid = G.total < g_halo_cus ? G.total : g_halo_cus;
    if (G.dbg & 256) {
        static unsigned long long* d = nullptr;
        if (!d) (void)hipMalloc(&d, 16 * sizeof(unsigned long long));
        HaloArgs G2 = G;
        G2.dbg_out = d;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(HL_THREADS), lds, st, P, G2);
        (void)hipDeviceSynchronize();
        unsigned long long h[16];
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        static int printed = 0;
        if (printed++ % 50 == 10) {
            const char* nm[8] = {"reads+dma", "epi/zero", "mfma k0", "pre-read", "mfma k1", "wait", "barrier", "tail"};
            for (int w = 0; w < 2; ++w) {
                fprintf(stderr, "halo dbg wave %d:", w ? 7 : 0);
                for (int i = 0; i < 8; ++i) fprintf(stderr, " %s=%llu", nm[i], h[w * 8 + i]);
                fprintf(stderr, "\n");
            }
        }
        return Y5M_OK;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(HL_THREADS), lds, st, P, G);
    Y5M_CHECK_LAUNCH("conv_halo_kernel");
    return Y5M_OK;
}

// 0: not taken (caller falls through to the other kernels), 1: launched, < 0: error
int y5m_conv_halo_try(const ConvParams& P, int dtype, hipStream_t st) {
    HaloArgs G;
    int BN;
    if (!halo_geom(P, dtype, G, BN)) return 0;
    int r;
    if (BN == 192) {
        r = P.epi == EPI_RAW_STATS ? launch_halo<6, EPI_RAW_STATS>(P, G, st)
          : P.epi == EPI_AFFINE_ACT ? launch_halo<6, EPI_AFFINE_ACT>(P, G, st) : launch_halo<6, EPI_DGRAD>(P, G, st);
    } else {
        r = P.epi == EPI_RAW_STATS ? launch_halo<3, EPI_RAW_STATS>(P, G, st)
          : P.epi == EPI_AFFINE_ACT ? launch_halo<3, EPI_AFFINE_ACT>(P, G, st) : launch_halo<3, EPI_DGRAD>(P, G, st);
    }
    return r == Y5M_OK ? 1 : r;
}
