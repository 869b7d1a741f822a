// 1x1 / stride-1 convolution (forward and data gradient) with a LONG K for gfx950, bf16: the persistent two-phase
// structure of the halo-patch kernel (y5m_conv_halo.hip) as a plain GEMM  D[n][m] = sum_k W[n][k] X[m][k].
//
// Why a fourth conv kernel. The 1x1 layers with >= 384 input channels (C3 outputs of the 40x40 / 20x20 stages, the SPPF
// and neck joins) are MFMA-shaped, but the tiled kernel runs them at 16-25 % of the bf16 peak: its 64 x 48 wave tiles move
// 4.7 LDS fragment rows per MFMA and it synchronises twice per 64-deep K step. Here
//   * a tile is 256 consecutive pixels x 192 channels, a workgroup is persistent (one per CU), 8 waves = 4 pixel groups x
//     2 channel groups, wave tile 64 x 96 (3.3 fragment rows per MFMA), exactly the halo kernel's accumulator layout, so
//     the epilogues (raw + statistics, folded BN + SiLU + residual, data gradient store / accumulate) are the same code;
//   * a unit is one 64-channel slab of K: 256 pixel rows + 192 weight rows of 128 B. Both operands travel global -> VGPR
//     (4 + 3 buffer loads per wave, issued between the quarters of the 48 MFMAs of unit g, for unit g+2) -> LDS
//     (ds_write_b128 in the fetch phase of unit g+1) into 2-stage rings; rows behind the tensor read as zeros through the
//     buffer resource's range check, units behind the last tile load any valid address (no branch in the stream);
//   * waves 0-3 and 4-7 (the two waves of each SIMD) run one phase apart: one group's 48 MFMAs next to the other group's
//     20 fragment reads + 7 operand stores. Two stages are enough: the stage a group writes in its fetch phase of unit g
//     (unit g+1's) was last read by the other group one phase earlier, and every phase ends in a barrier.
// The LDS image, the swizzle (chunk ^= row & 6) and the channel permutation of the weight rows are the halo kernel's.
#include "y5m_conv.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define GM_THREADS 512
#define GM_TP 256                 // pixels per tile
#define GM_A_STAGE (GM_TP * 128)  // bytes of one pixel stage (256 rows x 64 channels)
#define GM_W_STAGE (192 * 128)    // bytes of one weight stage

struct GemmArgs {
    int NU;                       // 64-channel units per tile
    int tiles_n, total;           // channel tiles per pixel tile, work items
    int Mtot;
    int stat_rows;                // rows of the statistics buffer (4 per pixel tile)
};

template <int NF>
__device__ __forceinline__ constexpr int gm_pch(int a, int rho) {     // same channel permutation as y5m_conv.hip (cv_pch)
    return (2 * (a >> 1) + 1 < NF) ? (a >> 1) * 32 + (rho >> 2) * 8 + (a & 1) * 4 + (rho & 3) : (a >> 1) * 32 + rho;
}
__device__ __forceinline__ int gm_logical_id(int bid, int nblk) {
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}
__device__ __forceinline__ float gm_row_sum(float v) {              // sum over the 16 lanes of a DPP row, in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

#define GM_PHASE_END() \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier(); \
    __builtin_amdgcn_sched_barrier(0);
#define GM_LD(dst, adr) dst = *reinterpret_cast<const uint4*>(smem + (adr));
#define GM_MFMAS(WB_, XA_, A0, A1) \
_Pragma("unroll") \
    for (int a = A0; a < A1; ++a) \
_Pragma("unroll") \
        for (int b = 0; b < 4; ++b) \
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16( \
                __builtin_bit_cast(bf16x8_t, WB_[a]), __builtin_bit_cast(bf16x8_t, XA_[b]), acc[a][b], 0, 0, 0);

template <int EPI>
__global__ __launch_bounds__(GM_THREADS) void conv_gemm8_kernel(const ConvParams P, const GemmArgs G) {
    constexpr int NF = 6, BN = 192;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid & 3, wn = wid >> 2;
    const int frow = lane & 15, fq = lane >> 4;
    // LDS map: [pixel stage 0][pixel stage 1][weight stage 0][weight stage 1]
    constexpr unsigned A_OFF = 0, W_OFF = 2 * GM_A_STAGE, T_OFF = W_OFF + 2 * GM_W_STAGE;      // T: statistics table [2][N] f32

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(P.in), 0, (unsigned)((size_t)G.Mtot * P.ldin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(P.w), 0, (unsigned)((size_t)P.Np * P.Kp * 2), 0x00020000);

    // ---- loop-invariant per-lane source offsets: a piece = 8 rows x 8 chunks of 16 B, lane (dr, dq) fetches logical chunk
    // dq ^ (dr & 6) of row dr and stores it lane-linear (piece base + lane * 16)
    const int dq = lane & 7, dr = lane >> 3;
    const unsigned ldb = (unsigned)(P.ldin * 2);
    const unsigned pch16 = (unsigned)((dq ^ (dr & 6)) << 4);
    unsigned avoff[4], wvoff[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) avoff[j] = (unsigned)((wid + 8 * j) * 8 + dr) * ldb + pch16;        // + (m0 * ldb + u * 128)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int r = (wid + 8 * j) * 8 + dr;                       // LDS row r <- channel n0 + perm(r)
        const int rw = r / (NF * 16), rl = r - rw * (NF * 16);
        const int rp = rw * (NF * 16) + gm_pch<NF>(rl >> 4, rl & 15);
        wvoff[j] = (unsigned)(rp * P.Kp * 2) + pch16;                // + (n0 * Kp + u * 64) * 2
    }
    const unsigned lane16 = (unsigned)lane << 4;
    const unsigned ast_v = A_OFF + (unsigned)(wid * 1024) + lane16;    // + stage + j * 8192
    const unsigned wst_v = W_OFF + (unsigned)(wid * 1024) + lane16;

    // ---- fragment read addresses (k-step 0; k-step 1 = address ^ 64)
    const unsigned swz = (unsigned)((fq ^ (frow & 6)) << 4);
    const unsigned xl = A_OFF + (unsigned)((wm * 64 + frow) * 128) + swz;          // + stage + b * 2048
    const unsigned wl = W_OFF + (unsigned)((wn * NF * 16 + frow) * 128) + swz;     // + stage + a * 2048

    // ---- work items in SGPRs: (tile row tm, channel tile tn, unit u) of the unit being computed, and the same two units
    // AHEAD (what the M phase loads); tile ids advance by the grid size, kept as (quotient, remainder) so that no division
    // sits in the loop
    const int nblk = gridDim.x;
    const int qn = nblk / G.tiles_n, rn = nblk - qn * G.tiles_n;
    int it = gm_logical_id(blockIdx.x, nblk), u = 0;
    int tm = it / G.tiles_n, tn = it - tm * G.tiles_n;
    int it2 = it, u2 = 0, tm2 = tm, tn2 = tn;
    auto ahead_offsets = [&](unsigned& asoff, unsigned& wsoff) __attribute__((always_inline)) {
        // (behind the last tile: any valid address -- what is loaded there is stored to LDS and never multiplied)
        const bool valid = it2 < G.total;
        asoff = valid ? (unsigned)(tm2 * GM_TP) * ldb + (unsigned)(u2 * 128) : 0u;
        wsoff = valid ? (unsigned)((tn2 * BN * P.Kp + u2 * 64) * 2) : 0u;
    };
    auto ahead_step = [&]() __attribute__((always_inline)) {
        if (++u2 == G.NU) {
            u2 = 0; it2 += nblk; tm2 += qn; tn2 += rn;
            if (tn2 >= G.tiles_n) { tn2 -= G.tiles_n; ++tm2; }
        }
    };
    u32x4 areg[4], wreg[3];
    // pixel rows: the WHOLE offset goes through the VGPR operand -- the buffer range check (rows behind the tensor read as
    // zeros: the statistics rely on it) looks at the vector offset only
    unsigned av[4];
    auto a_addr = [&](unsigned asoff) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = avoff[j] + asoff;
    };
    auto load_a = [&](int j0, int j1) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j >= j0 && j < j1) areg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, av[j], 0, 0);
    };
    auto load_w = [&](unsigned wsoff, int j0, int j1) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j >= j0 && j < j1) wreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wvoff[j], wsoff, 0);
    };
    auto store_ops = [&](unsigned sa, unsigned sw) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(smem + ast_v + sa + (unsigned)(j * 8192)) = areg[j];
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4*>(smem + wst_v + sw + (unsigned)(j * 8192)) = wreg[j];
    };

    f32x4 acc[NF][4];
    int em0 = 0, en0 = 0, etile = 0;                       // tile whose accumulators are waiting for their epilogue
    bool pending = false;

    auto epilogue = [&]() __attribute__((always_inline)) {
        const int nb = en0 + wn * NF * 16;
        if constexpr (EPI == EPI_RAW_STATS) {
            if (P.stats || P.bn_acc) {
                // rows past Mtot contributed exact zeros: no masking needed
#pragma unroll
                for (int a = 0; a < NF; ++a) {
                    float sv[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float v = acc[a][b][r]; sv[r] += v; ss[r] += v * v; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sv[r] = gm_row_sum(sv[r]); ss[r] = gm_row_sum(ss[r]); }
                    if (frow == 0) {
                        const int n = nb + gm_pch<NF>(a, fq * 4);
                        if (P.bn_acc) {
                            // accumulator rows (y5m_bnfuse.h): the workgroup's tiles are summed in an LDS table [2][N] (LDS
                            // atomics from the 4 pixel-group waves) and added to the f64 rows once, at the end of the kernel
                            float* tab = reinterpret_cast<float*>(smem + T_OFF);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                __hip_atomic_fetch_add(tab + n + r, sv[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(tab + P.N + n + r, ss[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        } else {
                            float* row = P.stats + ((size_t)(etile * 4 + wm) * 2) * P.Np + n;
                            *reinterpret_cast<float4*>(row) = make_float4(sv[0], sv[1], sv[2], sv[3]);
                            *reinterpret_cast<float4*>(row + P.Np) = make_float4(ss[0], ss[1], ss[2], ss[3]);
                        }
                    }
                }
            }
        }
        float4 scv[EPI == EPI_AFFINE_ACT ? NF : 1], shv[EPI == EPI_AFFINE_ACT ? NF : 1];     // folded BatchNorm of the lane's channels: once per tile
        if constexpr (EPI == EPI_AFFINE_ACT) {
#pragma unroll
            for (int a = 0; a < NF; ++a) {
                scv[a] = *reinterpret_cast<const float4*>(P.scale + nb + gm_pch<NF>(a, fq * 4));
                shv[a] = *reinterpret_cast<const float4*>(P.shift + nb + gm_pch<NF>(a, fq * 4));
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
                    for (int a = 0; a < NF; ++a) load4<bf16_t>(src + gm_pch<NF>(a, fq * 4), ov[a]);
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
                typedef unsigned u32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
                u32x4a8 q4;
                q4[0] = f32x2_to_bf16x2(fv[a][0], fv[a][1]);
                q4[1] = f32x2_to_bf16x2(fv[a][2], fv[a][3]);
                q4[2] = f32x2_to_bf16x2(fv[a + 1][0], fv[a + 1][1]);
                q4[3] = f32x2_to_bf16x2(fv[a + 1][2], fv[a + 1][3]);
                *reinterpret_cast<u32x4a8*>(ob + gm_pch<NF>(a, fq * 4)) = q4;
            }
        }
    };

    if constexpr (EPI == EPI_RAW_STATS) {
        if (P.bn_acc)
            for (int i = tid; i < 2 * P.N; i += GM_THREADS) reinterpret_cast<float*>(smem + T_OFF)[i] = 0.f;
    }
    // ---- prologue: unit 0 into stage 0 (synchronously), unit 1 into the registers ------------------------------------
    {
        unsigned asoff, wsoff;
        ahead_offsets(asoff, wsoff);
        a_addr(asoff);
        load_a(0, 4);
        load_w(wsoff, 0, 3);
        store_ops(0u, 0u);
        ahead_step();
        ahead_offsets(asoff, wsoff);
        a_addr(asoff);
        load_a(0, 4);
        load_w(wsoff, 0, 3);
        ahead_step();
    }
    __syncthreads();

    uint4 xa[4], wb[NF], xa1[4], wb1[NF];
    unsigned sa = 0u, sw = 0u;                              // stage (byte offset) of the unit being computed
    // waves 4-7 run one phase behind waves 0-3; every wave executes the same number of barriers in total
    if (wid >= 4) __builtin_amdgcn_s_barrier();

    while (it < G.total) {
        // ---- R phase: (first unit of a tile: epilogue of the previous one;) fragments of this unit, operands of the next
        __builtin_amdgcn_s_setprio(2);
        if (u == 0) {
            if (pending) epilogue();
#pragma unroll
            for (int a = 0; a < NF; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            etile = tm;
            em0 = tm * GM_TP;
            en0 = tn * BN;
            pending = true;
        }
        const unsigned xbase = xl + sa, wbase = wl + sw;
#pragma unroll
        for (int b = 0; b < 4; ++b) { GM_LD(xa[b], xbase + (unsigned)(b * 2048)) }
#pragma unroll
        for (int a = 0; a < NF; ++a) { GM_LD(wb[a], wbase + (unsigned)(a * 2048)) }
#pragma unroll
        for (int b = 0; b < 4; ++b) { GM_LD(xa1[b], (xbase ^ 64u) + (unsigned)(b * 2048)) }
#pragma unroll
        for (int a = 0; a < NF; ++a) { GM_LD(wb1[a], (wbase ^ 64u) + (unsigned)(a * 2048)) }
        store_ops(sa ^ (unsigned)GM_A_STAGE, sw ^ (unsigned)GM_W_STAGE);
        unsigned asoff, wsoff;
        ahead_offsets(asoff, wsoff);
        a_addr(asoff);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(0);
        GM_PHASE_END()
        // ---- M phase: 48 MFMAs, the 7 buffer loads of unit g+2 between their quarters and nothing else
        GM_MFMAS(wb, xa, 0, NF / 2)
        load_a(0, 2);
        GM_MFMAS(wb, xa, NF / 2, NF)
        load_a(2, 4);
        GM_MFMAS(wb1, xa1, 0, NF / 2)
        load_w(wsoff, 0, 2);
        GM_MFMAS(wb1, xa1, NF / 2, NF)
        load_w(wsoff, 2, 3);
        GM_PHASE_END()
        ahead_step();
        sa ^= (unsigned)GM_A_STAGE;
        sw ^= (unsigned)GM_W_STAGE;
        if (++u == G.NU) {
            u = 0; it += nblk; tm += qn; tn += rn;
            if (tn >= G.tiles_n) { tn -= G.tiles_n; ++tm; }
        }
    }
    if (wid < 4) __builtin_amdgcn_s_barrier();
    if (pending) epilogue();
    if constexpr (EPI == EPI_RAW_STATS) {
        if (P.bn_acc) {
            __syncthreads();                                  // every wave's LDS atomics are done (lgkmcnt is drained by the barrier)
            const float* tab = reinterpret_cast<const float*>(smem + T_OFF);
            for (int i = tid; i < 2 * P.N; i += GM_THREADS) {
                const float v = tab[i];
                const int which = i >= P.N ? 1 : 0;
                if (v != 0.f) bnf_add(P.bn_acc, P.Np, blockIdx.x, which, i - which * P.N, v);     // (channel tiles this workgroup never saw hold zeros)
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Y5M_CONV_GEMM8: 0 = off (tiled kernel everywhere), 1 = every eligible launch, 2 (default) = forward epilogues only, 3 = data
// gradients only. Measured (round 2, B=64, one launch alone on the chip, tiled -> this kernel): data gradient 384 -> 384 @
// 40x40 534 -> 659 TFLOP/s, 768 -> 768 @ 20x20 589 -> 736, 768 -> 384 @ 40x40 633 -> 746, 1536 -> 768 @ 20x20 632 -> 824;
// forward with statistics 400 -> 478, 504 -> 630, 362 -> 453 (384 -> 192 @ 80x80), 532 -> 655. Inside the train step the
// forward launches keep about half of that (407 -> 445, 524 -> 621, 659 -> 806 TFLOP/s per layer, eager timing) and the
// step is unchanged within noise (27.58 ms either way); the data-gradient launches LOSE 0.2 ms per step: a persistent
// workgroup holds its CU (226 VGPRs x 8 waves) for the whole launch, and the weight-gradient blocks of the forked stream
// (100+ us each) and these workgroups then wait for each other per CU instead of interleaving block by block as the tiled
// kernel's short-lived workgroups do -- the same reason the halo kernel's data gradient takes 108 us in the step against
// 73 us for its forward. Launching the data gradients with one work item per workgroup (short-lived workgroups, a knob
// since removed) recovered a third of the loss (27.71 vs 27.78 vs 27.57 ms). Hence forward only.
static int g_gemm8 = -1;
static int g_gemm8_cus = 0;
static int g_gemm8_min_tiles = -1; // Y5M_CONV_GEMM8_MIN: fewer work items than this stay on the tiled kernel

static bool gemm8_geom(const ConvParams& P, int dtype, GemmArgs& G) {
    if (g_gemm8 < 0) { const char* e = getenv("Y5M_CONV_GEMM8"); g_gemm8 = e ? atoi(e) : 2; }
    if (g_gemm8_min_tiles < 0) { const char* e = getenv("Y5M_CONV_GEMM8_MIN"); g_gemm8_min_tiles = e ? atoi(e) : 192; }
    if (!g_gemm8 || dtype != Y5M_BF16) return false;
    if ((g_gemm8 == 2 && P.epi == EPI_DGRAD) || (g_gemm8 == 3 && P.epi != EPI_DGRAD)) return false;     // 2: forward only, 3: data gradients only
    if (P.th != 1 || P.tw != 1 || P.sy != 1 || P.sx != 1 || P.dh0 != 0 || P.dw0 != 0) return false;
    if (P.Hin != P.Hg || P.Win != P.Wg || P.Hout != P.Hg || P.Wout != P.Wg) return false;
    if (P.osy != 1 || P.osx != 1 || P.ooy != 0 || P.oox != 0) return false;
    if (P.epi != EPI_RAW_STATS && P.epi != EPI_AFFINE_ACT && P.epi != EPI_DGRAD) return false;
    if (P.Cin < 384 || P.Cin % 64 != 0 || P.K != P.Cin || P.Kp < P.Cin || P.ldin % 8 != 0) return false;
    if (P.N % 192 != 0 || P.Np < P.N || P.N > 4096) return false;                  // (N <= 4096: the LDS statistics table)
    if (P.ldout % 8 != 0 || (reinterpret_cast<uintptr_t>(P.out) & 15) != 0) return false;      // 16-byte output pieces
    if (P.res && P.ldres % 4 != 0) return false;
    const long long Mtot = (long long)P.B * P.Hin * P.Win;
    if (Mtot * P.ldin * 2 >= (1ll << 31) || (long long)P.Np * P.Kp * 2 >= (1ll << 31)) return false;   // 32-bit offsets, top bit = out of range
    G.NU = P.Cin / 64;
    G.tiles_n = P.N / 192;
    const int tiles_m = (int)((Mtot + GM_TP - 1) / GM_TP);
    G.total = tiles_m * G.tiles_n;
    if (G.total < g_gemm8_min_tiles) return false;          // too few work items for one-workgroup-per-CU persistence
    G.Mtot = (int)Mtot;
    G.stat_rows = tiles_m * 4;
    return true;
}

// rows of the statistics buffer a RAW_STATS launch on this kernel writes (4 per 256-pixel tile), or 0 when the launch is
// not taken by it
int y5m_conv_gemm8_stat_rows(const ConvParams& P, int dtype) {
    GemmArgs G;
    return gemm8_geom(P, dtype, G) ? G.stat_rows : 0;
}

template <int EPI>
static int launch_gemm8(const ConvParams& P, const GemmArgs& G, hipStream_t st) {
    const size_t lds = 2 * (size_t)GM_A_STAGE + 2 * (size_t)GM_W_STAGE + (EPI == EPI_RAW_STATS && P.bn_acc ? 2 * (size_t)P.N * 4 : 0);
    auto kern = conv_gemm8_kernel<EPI>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    g_gemm8_cus = y5m_persistent_cus();
    const int grid = G.total < g_gemm8_cus ? G.total : g_gemm8_cus;
    Y5M_NAME_ONLY(Y5M_OK, "conv_gemm8_kernel<%d>", EPI);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(GM_THREADS), lds, st, P, G);
    Y5M_CHECK_LAUNCH("conv_gemm8_kernel");
    return Y5M_OK;
}

// 0: not taken (caller falls through to the other kernels), 1: launched, < 0: error
int y5m_conv_gemm8_try(const ConvParams& P, int dtype, hipStream_t st) {
    GemmArgs G;
    if (!gemm8_geom(P, dtype, G)) return 0;
    const int r = P.epi == EPI_RAW_STATS ? launch_gemm8<EPI_RAW_STATS>(P, G, st)
                : P.epi == EPI_AFFINE_ACT ? launch_gemm8<EPI_AFFINE_ACT>(P, G, st) : launch_gemm8<EPI_DGRAD>(P, G, st);
    return r == Y5M_OK ? 1 : r;
}
