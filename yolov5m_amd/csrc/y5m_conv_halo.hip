// 3x3 / stride-1 / pad-1 convolution (forward and data gradient) for gfx950, bf16: "halo patch" implicit GEMM.
//
// Why a third conv kernel. The tiled kernel (y5m_conv.hip) re-stages the same input pixels once per tap (9x) through
// VGPRs into LDS (ds_write_b128: 79 B/clk/CU) and synchronises twice per 64-deep K step: its LDS pipe is busier than
// its matrix pipe (966 vs 768 cycles per K step of a 128x192 tile), which is the ~30 % of peak it measures. Here
//   * a tile is 256 CONSECUTIVE output pixels in (b, y, x) raster order; the input pixels all 9 taps of the tile touch
//     are the raster run [m0 - W - 1, m0 + 256 + W + 1): one "patch" of 258 + 2W rows x 64 channels (128 B rows).
//     It is staged ONCE per 64-channel slab and every tap reads it at a row offset dy*W + dx; taps that fall outside
//     the image (or rows behind the tensor) are redirected per lane to a 128-byte zero row;
//   * the weights of one (slab, tap) unit (BN rows x 128 B, L2-resident) stream through a 3-stage LDS ring;
//   * operands travel global -> VGPR (buffer loads with hardware zero fill of out-of-range rows, issued between the
//     MFMAs of one unit) -> LDS (ds_write_b128 one unit later): two units of latency tolerance, waits counted by the
//     compiler. The LDS image of a 1 KiB piece is lane-linear (piece base + lane*16) and the bank swizzle sits on the
//     per-lane SOURCE address and again on the fragment read address -- the layout an LDS-DMA produces (probed on
//     hardware, tools/probe_lds_dma.hip), which the prologue still uses. The swizzle is chunk ^= row & 6 (not the tiled
//     kernel's (row>>1)&7): a tap shifts the 16 rows of a fragment by dy*W+dx, and (row>>1)&7 is 2-way conflicted for 3
//     of 4 row alignments, row&6 for none (bank model of the ds_read_b128 lane groups, MI355X_MICROARCH LDS table,
//     searched over all linear swizzles; PMC: SQ_LDS_BANK_CONFLICT = 5 % of SQ_LDS_IDX_ACTIVE);
//   * workgroups are persistent (one per CU, 8 waves = 4 pixel groups x 2 channel groups, wave tile 64 x {48,96}):
//     the next tile's first patch and weights are in flight while the current tile finishes;
//   * waves 0-3 and 4-7 (the two waves of each SIMD) run one PHASE apart: while one group issues the 48 MFMAs of a
//     unit, the other fetches its fragments, stores the staged operands and computes addresses (see HL_UNITS).
// Traffic per tile (192 -> 192 channels, W = 40): 130 KB of patch + 663 KB of weights for 170 MFLOP; the tiled kernel
// staged 1.15 MB of activations + 0.66 MB of weights for the same work.
// Measured (B = 80 x 40x40 x 192 -> 192, two full rounds of tiles): 1.0-1.1 PFLOP/s against 0.67-0.77 for the tiled
// kernel. What was tried on the way, with numbers, is in DESIGN.md section 4a (LDS-DMA for everything: each
// buffer_load ... lds stalled its wave 100-300 cycles inside the MFMA stream; four 24-MFMA phases per unit instead of
// two 48-MFMA ones; s_setprio for the fetching group; all within +-3 % of this version).
#include "y5m_conv.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define HL_THREADS 512
#define HL_TP 256                 // pixels per tile
#define HL_NS 3                   // weight ring stages (template parameter NS of the kernel: 3, or 2 for wide images -- see halo_geom)

struct HaloArgs {
    int PR8;                      // patch rows, rounded up to the 8-row DMA piece
    int npieces;                  // PR8 / 8
    int NPU;                      // units of a slab that carry a patch piece per wave (<= 8)
    int S;                        // 64-channel slabs (the last one may hold 32 channels)
    int tiles_n, total;           // channel tiles per pixel tile, work items
    int Mtot;                     // B*H*W
    int stat_rows;                // rows of the statistics buffer (4 per pixel tile)
    int ns;                       // weight ring stages of the instantiation that fits the LDS: 3, or 2 (Y5M_CONV_HALO_NS2)
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

// sum over the 16 lanes of a DPP row (= the 16 pixels of an accumulator fragment), result in every lane: four VALU
// instructions with DPP operands (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror) instead of four
// ds_bpermute round trips through the LDS pipe
__device__ __forceinline__ float hl_row_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// v_bfi_b32: (mask & a) | (~mask & b)
__device__ __forceinline__ unsigned hl_bfi(unsigned mask, unsigned a, unsigned b) { return (mask & a) | (~mask & b); }

// The unit loop is ISSUE-bound if written naively: a wave issues about one instruction per 4 cycles, so the ~700
// scalar / vector instructions per unit of the first version (index arithmetic with divisions, per-fragment address
// selects, dynamic waits) cost 3x the 48 MFMAs they surround (measured with s_memtime: 4350 cycles per unit, 1100 of
// them matrix work). This version keeps a unit at ~40 VALU + ~25 SALU + 24 LDS + 4 VMEM instructions + 48 MFMAs:
//   * the 9 taps are unrolled statically (HL_UNITS): weight ring stage = tap % 3 (9 % 3 == 0), tap offsets and LDS
//     store offsets are immediates;
//   * everything that depends on the (tile, slab) pair is computed once per pair (9 units) in SGPRs;
//   * the four pixel fragments of a lane share one swizzle (fragment stride 2048 B does not touch row bits 1-2), so a
//     tap costs 4 VALU for the base address + 3 per fragment (offset, v_bfe_i32 of the tap-valid bit, v_bfi_b32 select
//     of the zero row) + 1 per fragment for the second k-step (address ^ 64);
//   * a patch piece's source offset is one v_add of a lane constant and a per-unit scalar (rows outside the tensor wrap
//     to / land behind num_records and come back as zeros).
// The 9 tap units of one (tile, slab) pair are expanded inside conv_halo_kernel with NKS = 2 or 1 (k-steps per unit: 1
// for a 32-channel last slab). A unit is TWO PHASES, each closed by an s_barrier:
//   R: (first unit of a tile: epilogue of the previous tile, zero the accumulators;) fetch the 4 pixel + NF weight
//      fragments of every k-step (ds_read_b128); store the operands the PREVIOUS unit loaded (ds_write_b128: one patch piece
//      of the next pair, NWP weight pieces of unit g+1 -- pieces that do not exist go to a 1 KiB dummy region so that the
//      instruction stream has no branch); compute the NEXT unit's fragment addresses; s_waitcnt lgkmcnt(0): the fragments
//      are there, and this wave's reads / writes of the ring stage and patch buffer are complete before anyone passes the
//      barrier (the next writer of a stage is two phases away);
//   M: the NKS * NF * 4 MFMAs, back to back, with this unit's buffer loads (patch piece of the next pair, weights of unit
//      g+2) between their quarters and NOTHING else: 20 address VALU instructions interleaved with 24 MFMAs made a phase
//      690 instead of 380 cycles (s_memtime).
// Waves 4-7 run ONE PHASE BEHIND waves 0-3 (an extra barrier before the loop, see the kernel): wave w and w + 4 share a
// SIMD, so one of them is in its M phase while the other is in its R phase.
#ifdef HL_TIMING
__device__ __forceinline__ unsigned long long hl_clock() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define HL_PHASE_END() \
                __builtin_amdgcn_sched_barrier(0); \
                { const unsigned long long n_ = hl_clock(); tacc[2 * (tph & 1)] += n_ - tlast; tlast = n_; } \
                __builtin_amdgcn_s_barrier(); \
                { const unsigned long long n_ = hl_clock(); tacc[2 * (tph & 1) + 1] += n_ - tlast; tlast = n_; ++tph; } \
                __builtin_amdgcn_sched_barrier(0);
#else
#define HL_PHASE_END() \
                __builtin_amdgcn_sched_barrier(0); \
                __builtin_amdgcn_s_barrier(); \
                __builtin_amdgcn_sched_barrier(0);
#endif
#define HL_NEXT_ADDR() \
                if (t < 8) { \
                    tap_addr(pbo, t + 1); \
                } else { \
                    if (ns == 0 && nvalid) setup_masks(nm0); \
                    tap_addr(npbo, 0); \
                }
#define HL_LD(dst, adr) dst = *reinterpret_cast<const uint4*>(smem + (adr));
#define HL_PRIO(x) __builtin_amdgcn_s_setprio(x);
#define HL_MFMAS2(WB_, XA_, A0, A1) \
_Pragma("unroll") \
                for (int a = A0; a < A1; ++a) \
_Pragma("unroll") \
                    for (int b = 0; b < 4; ++b) \
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16( \
                            __builtin_bit_cast(bf16x8_t, WB_[a]), __builtin_bit_cast(bf16x8_t, XA_[b]), acc[a][b], 0, 0, 0);
#define HL_UNITS(NKS) \
_Pragma("unroll") \
            for (int t = 0; t < 9; ++t) { \
                HL_PRIO(2) \
                if (t == 0 && s == 0) { \
                    if (pending) epilogue(); \
_Pragma("unroll") \
                    for (int a = 0; a < NF; ++a) \
_Pragma("unroll") \
                        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f}; \
                    etile = tile_m; \
                    em0 = m0; \
                    en0 = n0; \
                    pending = true; \
                } \
                const unsigned wbase = wl + (NS == 3 ? (unsigned)((t % 3) * WB) : wst); \
                const unsigned wsrc = t + 2 < 9 ? wso : nwso; \
                const int wtap = t + 2 < 9 ? t + 2 : t + 2 - 9; \
                const int tp = (t + 8) % 9;                       /* the unit whose loads are stored now */ \
_Pragma("unroll") \
                for (int b = 0; b < 4; ++b) { HL_LD(xa[b], a0[b]) } \
_Pragma("unroll") \
                for (int a = 0; a < NF; ++a) { HL_LD(wb[a], wbase + (unsigned)(a * 2048)) } \
                if (NKS == 2) { \
_Pragma("unroll") \
                    for (int b = 0; b < 4; ++b) { HL_LD(xa1[b], a0[b] ^ 64u) } \
_Pragma("unroll") \
                    for (int a = 0; a < NF; ++a) { HL_LD(wb1[a], (wbase ^ 64u) + (unsigned)(a * 2048)) } \
                } \
                store_patch_piece(preg, tp, tp == 8 ? cpst_v : npst_v); \
                store_weights(wreg, (tp + 2) % 9, 0, NWP, wst ^ (unsigned)WB); \
                HL_NEXT_ADDR() \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                HL_PRIO(0) \
                HL_PHASE_END() \
                HL_MFMAS2(wb, xa, 0, NF / 2) \
                preg = load_patch_piece(npl_v, nsp, t); \
                load_weights(wreg, wsrc, wtap, 0, 1); \
                HL_MFMAS2(wb, xa, NF / 2, NF) \
                if (NKS == 2) { \
                    HL_MFMAS2(wb1, xa1, 0, NF / 2) \
                    load_weights(wreg, wsrc, wtap, 1, NWP); \
                    HL_MFMAS2(wb1, xa1, NF / 2, NF) \
                } else { \
                    load_weights(wreg, wsrc, wtap, 1, NWP); \
                } \
                if (NS == 2) wst ^= (unsigned)WB; \
                HL_PHASE_END() \
            }

// NS: stages of the weight ring. 3 = the measured form (stage = tap % 3, a compile-time constant of the unrolled taps). 2 (round 5,
// Y5M_CONV_HALO_NS2, unmeasured): the stage is a scalar that toggles every unit -- the long-K GEMM kernel's ring (y5m_conv_gemm.hip:
// "two stages are enough"): unit t + 1's weights are stored in the R phase of unit t into the stage unit t - 1 was read from, which
// both wave groups have left (group A read it two phases, group B one phase earlier, and every phase ends in a barrier). It frees
// 24 KB of LDS, which is what lets images 45..88 pixels wide -- the 80x80 stage of a 1280x1280 model -- use this kernel at all
// (their two patch buffers + three stages exceed 160 KB, so they ran on the tiled kernel).
template <int NF, int EPI, int NS>
__global__ __launch_bounds__(HL_THREADS) void conv_halo_kernel(const ConvParams P, const HaloArgs G) {
    constexpr int BN = 2 * NF * 16;                       // channels per tile (2 channel groups of waves)
    constexpr int WB = BN * 128;                          // bytes of one weight stage
    constexpr int NWP = (BN / 8 + 7) / 8;                 // weight DMA pieces per wave and unit (3 | 2)
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid & 3, wn = wid >> 2;
    const int frow = lane & 15, fq = lane >> 4;
    const int W = P.Win, H = P.Hin;
    const int PB = G.PR8 * 128;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    // LDS map: [patch 0][patch 1][weight stage 0..2][zero row 128][dummy 1024]
    const unsigned P_OFF = 0, W_OFF = 2 * PB, Z_OFF = W_OFF + NS * WB, D_OFF = Z_OFF + 128;
    unsigned wst = 0u;                                     // NS == 2: byte offset (0 | WB) of the ring stage the current unit reads

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
        // (a piece behind the tile's BN rows only exists for BN = 96, waves 4-7, j = 1: it goes to the dummy region)
        wvoff[j] = wid + 8 * j < BN / 8 ? (unsigned)((rp * P.Kp + ((dq ^ (r & 6)) << 3)) * 2) : OOB;
    }
    const unsigned ldb = (unsigned)(P.ldin * 2);
    const int pch = dq ^ (dr & 6);                         // logical chunk this lane fetches of a patch row (piece rows start at a multiple of 8)
    const unsigned pl_off = (unsigned)(pch << 4);
    const bool pl_last_ok = (G.S - 1) * 64 + pch * 8 < P.Cin;      // upper half of a 32-channel last slab reads as zeros
    // patch piece t*8 + wid of a pair, row (t*64 + wid*8 + dr): byte offset = lane part + a per-unit SCALAR part
    // ((first pixel + t*64) * ldb + slab*128, negative before the tensor). One v_add per piece; rows outside the tensor
    // wrap to / land at >= num_records and read as zeros (halo_geom: tensor < 1 GiB, (W + 2) * ldb < 1 MiB), and so does
    // the lane part HL_VOOB of the lanes whose channels lie behind Cin in a 32-channel last slab.
    constexpr unsigned HL_VOOB = 0x80100000u;
    const unsigned pl_v = (unsigned)((wid * 8 + dr) * (int)ldb) + pl_off;
    const unsigned pl_v_last = pl_last_ok ? pl_v : HL_VOOB;
    const unsigned wdst = lds0 + (unsigned)(wid * 1024);   // + stage + j*8192: this wave's weight pieces

    // ---- fragment read addresses -----------------------------------------------------------------------------
    // weights (MFMA A operand): row nl = wn*NF*16 + a*16 + frow; nl & 6 == frow & 6
    const unsigned wl = W_OFF + (unsigned)((wn * NF * 16 + frow) * 128 + ((fq ^ (frow & 6)) << 4));
    // pixels (MFMA B operand): patch row of tile pixel ml at tap offset 0 is ml + W + 1
    const unsigned prow = P_OFF + (unsigned)((wm * 64 + frow + W + 1) * 128);
    const unsigned fq16 = (unsigned)fq << 4;
    const unsigned zadr = Z_OFF + fq16;

    // tap offsets in patch rows * 128 (SGPRs)
    int tapd[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) tapd[k] = ((P.dh0 + (k / 3) * P.dhs) * W + (P.dw0 + (k % 3) * P.dws)) * 128;
    const unsigned C2 = (unsigned)(P.Cin * 2);

    // ---- work items --------------------------------------------------------------------------------------------
    const int nblk = gridDim.x;
    int it = hl_logical_id(blockIdx.x, nblk);

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
                    for (int r = 0; r < 4; ++r) { sv[r] = hl_row_sum(sv[r]); ss[r] = hl_row_sum(ss[r]); }
                    if (frow == 0) {
                        const int n = nb + hl_pch<NF>(a, fq * 4);
                        // (with accumulator rows -- P.bn_acc -- the rows are this workgroup's private staging: see the
                        //  end of the kernel. Atomics issued from here, 8 per lane and fragment with 4 active lanes, cost
                        //  the forward launches 74 -> 127 us: VMEM issue slots inside the MFMA stream)
                        float* row = P.stats + ((size_t)(etile * 4 + wm) * 2) * P.Np + n;
                        *reinterpret_cast<float4*>(row) = make_float4(sv[0], sv[1], sv[2], sv[3]);
                        *reinterpret_cast<float4*>(row + P.Np) = make_float4(ss[0], ss[1], ss[2], ss[3]);
                    }
                }
            }
        }
        float4 scv[EPI == EPI_AFFINE_ACT ? NF : 1], shv[EPI == EPI_AFFINE_ACT ? NF : 1];     // folded BatchNorm of the lane's channels: once per tile
        if constexpr (EPI == EPI_AFFINE_ACT) {
#pragma unroll
            for (int a = 0; a < NF; ++a) {
                scv[a] = *reinterpret_cast<const float4*>(P.scale + nb + hl_pch<NF>(a, fq * 4));
                shv[a] = *reinterpret_cast<const float4*>(P.shift + nb + hl_pch<NF>(a, fq * 4));
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int m = em0 + wm * 64 + b * 16 + frow;
            if (m >= G.Mtot) continue;
            bf16_t* const ob = reinterpret_cast<bf16_t*>(P.out) + (size_t)m * P.ldout + nb;
            float fv[NF][4];
            // the row's read-modify-write / residual operands: ALL fragments requested before the first is used (one load, one wait,
            // one fragment at a time made the epilogue a chain of NF dependent memory round trips per row)
            float ov[NF][4];
            constexpr bool OPND = EPI == EPI_AFFINE_ACT || EPI == EPI_DGRAD;
            const bool opnd = EPI == EPI_AFFINE_ACT ? P.res != nullptr : (EPI == EPI_DGRAD && P.accumulate);
            if constexpr (OPND) {
                if (opnd) {
                    const bf16_t* src = P.res ? reinterpret_cast<const bf16_t*>(P.res) + (size_t)m * P.ldres + nb : ob;
#pragma unroll
                    for (int a = 0; a < NF; ++a) load4<bf16_t>(src + hl_pch<NF>(a, fq * 4), ov[a]);
                }
            }
#pragma unroll
            for (int a = 0; a < NF; ++a) {
                float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
                if constexpr (EPI == EPI_AFFINE_ACT) {
                    v[0] = v[0] * scv[a].x + shv[a].x; v[1] = v[1] * scv[a].y + shv[a].y;
                    v[2] = v[2] * scv[a].z + shv[a].z; v[3] = v[3] * scv[a].w + shv[a].w;
                    if (P.act == Y5M_ACT_SILU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
                    }
                }
                if constexpr (OPND) {
                    if (opnd) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += ov[a][r];
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

    auto setup_masks = [&](int m0) __attribute__((always_inline)) {
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
    // addresses of this lane's k-step-0 chunks of the 4 pixel fragments at tap k of the patch buffer at byte offset pbo
    unsigned a0[4];
    auto tap_addr = [&](unsigned pbo, int k) __attribute__((always_inline)) {
        const unsigned rowb = prow + (pbo + (unsigned)tapd[k]);
        const unsigned adr = rowb + (fq16 ^ ((rowb >> 3) & 0x60u));          // chunk ^= row & 6; same for all 4 fragments
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned ok = (unsigned)__builtin_amdgcn_sbfe((int)vmask[b], k, 1);     // 0 or 0xffffffff
            a0[b] = hl_bfi(ok, adr + (unsigned)(b * 2048), zadr);
        }
    };
    // ---- operand transport: global -> VGPR (buffer loads, issued between the MFMAs of an M phase) -> LDS (ds_write_b128 in
    // the R phases of the NEXT unit). The LDS image is the lane-linear one an LDS-DMA would produce (piece base + lane*16,
    // swizzle on the source address). Why not LDS-DMA, which this kernel used first: every buffer_load ... lds stalled its
    // wave for 100-300 cycles inside the MFMA stream (ablation at B=80, 192 channels: 81 us with DMA, 62 us without any
    // transport, reads + address arithmetic included); a plain buffer load costs a few issue slots and the compiler counts
    // the waits (vmcnt(2) at every store: two phases of latency tolerance).
    const unsigned lane16 = (unsigned)lane << 4;
    const unsigned wst_v = W_OFF + (unsigned)(wid * 1024) + lane16;        // + stage*WB + j*8192
    const unsigned pst_v = P_OFF + (unsigned)(wid * 1024) + lane16;        // + buffer + t*8192
    // weights of (channel-tile offset wso = (n0*Kp + slab*64)*2, tap k): pieces j0 <= j < j1 of this wave
    auto load_weights = [&](u32x4 (&wr)[NWP], unsigned wso, int k, int j0, int j1) __attribute__((always_inline)) {
        const unsigned soff = wso + (unsigned)k * C2;
#pragma unroll
        for (int j = 0; j < NWP; ++j)
            if (j >= j0 && j < j1) wr[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wvoff[j], soff, 0);
    };
    // ... into ring stage k % 3 (a piece behind the tile's BN rows -- BN = 96, waves 4-7, j = 1 -- goes to the dummy region)
    auto store_weights = [&](const u32x4 (&wr)[NWP], int k, int j0, int j1, unsigned stage2) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NWP; ++j)
            if (j >= j0 && j < j1) {
                const unsigned stg = NS == 3 ? (unsigned)((k % 3) * WB) : stage2;        // (NS == 2: the stage the NEXT unit reads)
                const unsigned dst = wid + 8 * j < BN / 8 ? wst_v + stg + (unsigned)(j * 8192) : D_OFF + lane16;
                *reinterpret_cast<u32x4*>(smem + dst) = wr[j];
            }
    };
    // piece pc of the patch whose first row is pixel pix0 (may be negative), channel byte offset cbo
    auto patch_piece_voff = [&](int pix0, unsigned cbo, bool chan_ok, int pc) __attribute__((always_inline)) {
        const int pix = pix0 + pc * 8 + dr;
        const unsigned off = __umul24((unsigned)pix, ldb) + (cbo + pl_off);
        const bool ok = (unsigned)pix < (unsigned)G.Mtot && chan_ok && pc < G.npieces;
        return ok ? off : OOB;
    };
    // (unit loop) piece t*8 + wid: lane part lv (pl_v or pl_v_last) + scalar part sp of the next pair + t*64 rows
    auto load_patch_piece = [&](unsigned lv, unsigned sp, int t) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(rs_x, lv + (sp + (unsigned)(t * 64) * ldb), 0, 0);
    };
    // ... loaded in unit tu, into the patch buffer at byte offset pbo (pieces behind the patch: dummy region)
    auto store_patch_piece = [&](const u32x4& pr, int tu, unsigned pv) __attribute__((always_inline)) {
        const int pc = tu * 8 + wid;
        const unsigned dst = pc < G.npieces ? pv + (unsigned)(tu * 8192) : D_OFF + lane16;
        *reinterpret_cast<u32x4*>(smem + dst) = pr;
    };
    // (prologue only) LDS-DMA versions
    auto issue_weights = [&](unsigned wso, int k) __attribute__((always_inline)) {
        const unsigned soff = wso + (unsigned)k * C2;
#pragma unroll
        for (int j = 0; j < NWP; ++j)
            hl_dma16(rs_w, wvoff[j], soff, wid + 8 * j < BN / 8 ? wdst + W_OFF + (unsigned)((k % 3) * WB + j * 8192) : lds0 + D_OFF);
    };
    auto issue_patch_piece = [&](unsigned voff, int pc, unsigned pbo) __attribute__((always_inline)) {
        hl_dma16(rs_x, voff, 0u, pc < G.npieces ? lds0 + P_OFF + pbo + (unsigned)(pc * 1024) : lds0 + D_OFF);
    };

    // ---- prologue: first patch, weights of units 0 and 1 --------------------------------------------------------
    {
        const int m0 = (it / G.tiles_n) * HL_TP;
        const unsigned wso = (unsigned)(((it % G.tiles_n) * BN * P.Kp) * 2);
        for (int i = 0; i < G.NPU; ++i)
            issue_patch_piece(patch_piece_voff(m0 - (W + 1), 0u, G.S > 1 || pl_last_ok, i * 8 + wid), i * 8 + wid, 0u);
        issue_weights(wso, 0);
        setup_masks(m0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    uint4 xa[4], wb[NF], xa1[4], wb1[NF];
    u32x4 wreg[NWP], preg;                                 // operands in flight: loaded in unit g, stored in unit g+1
    {
        const unsigned wso0 = (unsigned)(((it % G.tiles_n) * BN * P.Kp) * 2);
        load_weights(wreg, wso0, 1, 0, NWP);               // what "unit -1" would have loaded: weights of unit 1, no patch piece
        preg = load_patch_piece(pl_v, 0x80000000u, 0);     // (all lanes out of range)
    }
#ifdef HL_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = hl_clock();
    int tph = 0;
#endif
    tap_addr(0u, 0);
    // waves 4-7 run one phase behind waves 0-3 (HL_UNITS); every wave executes the same number of barriers in total
    if (wid >= 4) __builtin_amdgcn_s_barrier();

    // ---- (tile, slab) pair loop, 9 statically unrolled tap units each -------------------------------------------
    int s = 0;
    unsigned pbo = 0u;                                     // byte offset of the current slab's patch buffer
    while (it < G.total) {
        // per-pair scalars: this pair and the next one (same tile next slab, or the next tile's first slab)
        const int tile_m = it / G.tiles_n;
        const int m0 = tile_m * HL_TP, n0 = (it - tile_m * G.tiles_n) * BN;
        const unsigned wso = (unsigned)((n0 * P.Kp + s * 64) * 2);
        const int nks = (P.Cin - s * 64) >= 64 ? 2 : 1;
        int ns = s + 1, nit = it;
        if (ns == G.S) { ns = 0; nit += nblk; }
        const bool nvalid = nit < G.total;
        const int ntile_m = nit / G.tiles_n;
        const int nm0 = ntile_m * HL_TP, nn0 = (nit - ntile_m * G.tiles_n) * BN;
        const unsigned nwso = nvalid ? (unsigned)((nn0 * P.Kp + ns * 64) * 2) : wso;       // (no next pair: any valid address)
        // (no next pair: every lane out of range)
        const unsigned nsp = nvalid ? (unsigned)((nm0 - (W + 1)) * (int)ldb + ns * 128) : 0x80000000u;
        const unsigned npl_v = ns < G.S - 1 ? pl_v : pl_v_last;
        const unsigned npbo = pbo ^ (unsigned)PB;          // patch buffers at 0 and PB
        const unsigned npst_v = pst_v + npbo, cpst_v = pst_v + pbo;
        // the 9 tap units of this pair; NKS (k-steps per unit: 2, or 1 for a 32-channel last slab) is a compile-time
        // constant of the unrolled sequence, so that no fragment read sits in a conditional block (the compiler's
        // counted lgkmcnt before the first MFMA otherwise has to assume the shorter path and waits for the k-step-1 reads)
        // the 9 tap units of this pair (HL_UNITS, defined above the kernel); NKS (k-steps per unit: 2, or 1 for a 32-channel
        // last slab) is a literal in each expansion, so that no fragment read sits in a conditional block
        if constexpr (NF == 3) {
            if (nks == 2) { HL_UNITS(2) } else { HL_UNITS(1) }
        } else {
            HL_UNITS(2)                                    // (the 192-channel tile requires Cin % 64 == 0: halo_geom)
        }
        pbo = npbo;
        s = ns;
        it = nit;
    }
    if (wid < 4) __builtin_amdgcn_s_barrier();
    if (pending) epilogue();
    if constexpr (EPI == EPI_RAW_STATS) {
        if (P.bn_acc && P.stats) {
            // accumulator rows (y5m_bnfuse.h): the workgroup sums the 4 partial rows of each of ITS tiles (its own stores:
            // drained, then read back past the L1) and adds the tiles' totals -- 2 * BN atomics per run of tiles that share
            // a channel tile, issued once, behind the last MFMA
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid < 2 * BN) {
                const int which = tid / BN, cl = tid - which * BN;
                float tot = 0.f;
                int cur_n0 = -1, cur_tile = 0;
                for (int t = hl_logical_id(blockIdx.x, nblk); t < G.total; t += nblk) {
                    const int tile_m = t / G.tiles_n, n0 = (t - tile_m * G.tiles_n) * BN;
                    if (n0 != cur_n0) {
                        if (cur_n0 >= 0) bnf_add(P.bn_acc, P.Np, cur_tile, which, cur_n0 + cl, tot);
                        tot = 0.f; cur_n0 = n0; cur_tile = tile_m;
                    }
                    const float* row = P.stats + ((size_t)(tile_m * 4) * 2 + which) * P.Np + n0 + cl;
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        tot += __hip_atomic_load(row + (size_t)w * 2 * P.Np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (cur_n0 >= 0) bnf_add(P.bn_acc, P.Np, cur_tile, which, cur_n0 + cl, tot);
            }
        }
    }
#ifdef HL_TIMING
    if (blockIdx.x == 0 && (wid == 0 || wid == 4) && lane == 0)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(const_cast<void*>(P.zeros))[32 + (wid ? 8 : 0) + i] = tacc[i];
#endif
}

// ------------------------------------------------------------------------------------------------------------------
int y5m_conv_gemm8_stat_rows(const ConvParams& P, int dtype);      // y5m_conv_gemm.hip
static int g_halo = -1;            // Y5M_CONV_HALO: 0 = tiled kernel everywhere (A/B runs), 1 (default) = the 192-channel tile
                                   // (a 96-channel tile, NF = 3, was built and measured in round 2: 24 MFMAs per wave and phase
                                   // do not cover the other group's R phase -- 490-580 TFLOP/s against 520-670 for the tiled
                                   // kernel on 96 -> 96 @ 80x80; removed in round 3, NOTES.md)
static int g_halo_cus = 0;

static bool halo_geom(const ConvParams& P, int dtype, HaloArgs& G, int& BN) {
    if (g_halo < 0) { const char* e = getenv("Y5M_CONV_HALO"); g_halo = e ? atoi(e) : 1; }
    if (!g_halo || dtype != Y5M_BF16) return false;
    if (P.th != 3 || P.tw != 3 || P.sy != 1 || P.sx != 1) return false;
    if (!((P.dh0 == -1 && P.dhs == 1) || (P.dh0 == 1 && P.dhs == -1))) return false;
    if (!((P.dw0 == -1 && P.dws == 1) || (P.dw0 == 1 && P.dws == -1))) return false;
    if (P.Hin != P.Hg || P.Win != P.Wg || P.Hout != P.Hg || P.Wout != P.Wg) return false;
    if (P.osy != 1 || P.osx != 1 || P.ooy != 0 || P.oox != 0) return false;
    if (P.epi != EPI_RAW_STATS && P.epi != EPI_AFFINE_ACT && P.epi != EPI_DGRAD) return false;
    if (P.bn_acc && !P.stats) return false;                 // accumulator rows: this kernel stages its tiles' sums in stats rows
    if (P.Cin < 64 || P.Cin % 32 != 0 || P.ldin % 8 != 0) return false;
    if (P.N % 192 != 0 || P.Cin % 64 != 0) return false;
    BN = 192;
    if (P.Np < (P.N + BN - 1) / BN * BN) return false;
    if (P.ldout % 8 != 0 || (reinterpret_cast<uintptr_t>(P.out) & 15) != 0) return false;      // 16-byte output pieces
    if (P.res && P.ldres % 4 != 0) return false;
    const int S = (P.Cin + 63) / 64;
    if (P.Kp < 8 * P.Cin + S * 64) return false;            // the last unit's 128-byte weight rows stay inside the packed rows
    const long long Mtot = (long long)P.B * P.Hin * P.Win;
    if (Mtot >= (1ll << 24) || (long long)P.ldin * 2 >= (1ll << 24)) return false;           // 24-bit multiply in the patch address
    if (Mtot * P.ldin * 2 >= (1ll << 30) || (long long)(P.Win + 2) * P.ldin * 2 >= (1ll << 20)) return false;   // see pl_v in the kernel
    const int PR = HL_TP + 2 * P.Win + 2;
    G.PR8 = (PR + 7) / 8 * 8;
    G.npieces = G.PR8 / 8;
    G.NPU = (G.npieces + 7) / 8;
    if (G.NPU > 8) return false;           // one piece per wave and unit, units 0..7 (a piece issued in unit 8 would not be waited for before the slab switch)
    G.ns = HL_NS;
    if (2 * (size_t)G.PR8 * 128 + HL_NS * (size_t)BN * 128 + 128 + 1024 > 160 * 1024) {
        // two patch buffers + three weight stages do not fit (images wider than 44 pixels): a two-stage ring does up to 88
        static int ns2 = -1;                // Y5M_CONV_HALO_NS2 (default 0: written without a GPU; A/B staged in tools/r5_gpu_job.sh)
        if (ns2 < 0) { const char* e = getenv("Y5M_CONV_HALO_NS2"); ns2 = e ? atoi(e) : 0; }
        if (!ns2 || 2 * (size_t)G.PR8 * 128 + 2 * (size_t)BN * 128 + 128 + 1024 > 160 * 1024) return false;
        G.ns = 2;
    }
    G.S = S;
    G.tiles_n = (P.N + BN - 1) / BN;
    const int tiles_m = (int)((Mtot + HL_TP - 1) / HL_TP);
    G.total = tiles_m * G.tiles_n;
    G.Mtot = (int)Mtot;
    G.stat_rows = tiles_m * 4;
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
    const int gr = y5m_conv_gemm8_stat_rows(P, dtype);
    if (gr > 0) return gr;
    return (P.M + CV_BM - 1) / CV_BM;
}

// 1 when the kernel y5m_conv would run for this RAW_STATS launch stages its tiles' statistics in partial rows even with
// accumulator rows (bn_acc): the persistent halo-patch kernel, whose LDS is full -- the caller then passes stats as well
extern "C" int y5m_conv_stages_stats(const y5m_conv_args* args, int dtype) {
    ConvParams P;
    memcpy(&P, args, sizeof(P));
    HaloArgs G;
    int BN;
    return halo_geom(P, dtype, G, BN) ? 1 : 0;         // (the long-K GEMM kernel sums its tiles in an LDS table instead)
}

// 1 when y5m_conv runs this launch on the halo-patch kernel
extern "C" int y5m_conv_is_halo(const y5m_conv_args* args, int dtype) {
    ConvParams P;
    memcpy(&P, args, sizeof(P));
    HaloArgs G;
    int BN;
    return halo_geom(P, dtype, G, BN) ? 1 : 0;
}

template <int NF, int EPI, int NS>
static int launch_halo(const ConvParams& P, const HaloArgs& G, hipStream_t st) {
    constexpr int BN = 2 * NF * 16;
    const size_t lds = 2 * (size_t)G.PR8 * 128 + NS * (size_t)BN * 128 + 128 + 1024;
    auto kern = conv_halo_kernel<NF, EPI, NS>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    g_halo_cus = y5m_persistent_cus();
    const int grid = G.total < g_halo_cus ? G.total : g_halo_cus;
    if (NS == 3) { Y5M_NAME_ONLY(Y5M_OK, "conv_halo_kernel<%d,%d>", NF, EPI); }
    else { Y5M_NAME_ONLY(Y5M_OK, "conv_halo_kernel<%d,%d,ns%d>", NF, EPI, NS); }
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
    if (G.ns == 2)
        r = P.epi == EPI_RAW_STATS ? launch_halo<6, EPI_RAW_STATS, 2>(P, G, st)
          : P.epi == EPI_AFFINE_ACT ? launch_halo<6, EPI_AFFINE_ACT, 2>(P, G, st) : launch_halo<6, EPI_DGRAD, 2>(P, G, st);
    else
        r = P.epi == EPI_RAW_STATS ? launch_halo<6, EPI_RAW_STATS, 3>(P, G, st)
          : P.epi == EPI_AFFINE_ACT ? launch_halo<6, EPI_AFFINE_ACT, 3>(P, G, st) : launch_halo<6, EPI_DGRAD, 3>(P, G, st);
    return r == Y5M_OK ? 1 : r;
}
