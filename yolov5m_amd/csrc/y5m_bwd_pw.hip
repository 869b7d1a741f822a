// Fused backward of a pointwise (1x1, stride 1) CBL with C input = C output channels, C in {48, 96, 192}, bf16:
//
//     dy = BatchNorm+SiLU backward of (dz, y)      -- never written to HBM
//     dx (+)= dy . W                                -- data gradient (reference autograd of nn.Conv2d wrt its input)
//     dW  += dy^T . x                               -- weight gradient, f32 atomics straight into the flat gradient
//
// in ONE persistent launch per layer (reference model.py:12-28 CBL backward; 25 of the 79 CBLs and both halves of the
// merged C3 pairs of the 160x160 / 80x80 stages qualify). Before: bn_bwd_apply (reads dz, y, writes dy: 6 B per element),
// the pointwise data gradient (reads dy, writes dx) and, on the forked stream, the weight gradient (reads dy and x): 14 B per
// element and three launches, one of them competing with the main stream for CUs. Now dz, y and x are read once and dx is
// written once: 8 B per element, and the layer's weight gradient costs no launch at all (the step without the 1x1 weight
// gradients measured 1.6 ms faster, tools/ab_step.sh). The BatchNorm reduction (sum dt, sum dt (y - mean)) still needs its own
// pass over (dz, y) ahead of this kernel: dy depends on the channel sums of the whole tensor (y5m_bn_bwd_fused_phase, phase 1).
//
// The kernel is HBM-bound by construction (96 flop per byte at C = 192, 19 % of the MFMA rate at the HBM rate), so its
// structure is a streaming one: a tile of TP = 64 / 128 / 256 pixels (always 1536 16-byte pieces of dz, y and x each, three per
// thread) is prefetched into registers one tile ahead; phase A turns (dz, y) into dy on the VALU and writes dy and x into
// two LDS tiles in the weight-gradient kernel's [32 pixel][16 channel] sub-tile layout; after ONE barrier phase B computes the
// data gradient (A operand = the layer's weights, staged once per workgroup in MFMA fragment order; B operand = dy rows read
// back from the tile with conflict-free ds_read_b128) and phase C the weight gradient (both operands by transposing
// ds_read_b64_tr_b16 reads of the two tiles) into accumulators that live for the whole launch; a second barrier frees the
// tiles. The coefficients of dy (BatchNorm scale / shift / mean and the two backward coefficients from the reduce pass's
// f64 accumulator rows, y5m_bnfuse.h) are derived in the prologue; workgroup 0 writes dgamma / dbeta.
#include "y5m_conv.h"
#include <string.h>
#include <type_traits>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef y5m_bwd_pw_args BwdPwParams;

#define BP_THREADS 512
#define BP_SUB 1088        // sub-tile stride: 1024 + 64 (neighbouring sub-tiles' 16-byte staging writes on distinct banks)

template <int C>
struct BpCfg {
    static constexpr int NC = C / 16;                  // 16-channel fragments
    static constexpr int KS = (C + 31) / 32;           // 32-channel K steps of the data gradient (K = output channels)
    static constexpr int NSUB = KS * 2;                // sub-tile columns of the dy tile (C = 48: one zero column pads K to 64)
    static constexpr int TP = 1536 * 8 / C;            // pixels per tile: 64 / 128 / 256 (1536 16-byte pieces per operand)
    static constexpr int NBLK = TP / 32;               // 32-pixel blocks = K steps of the weight gradient
    static constexpr int CPR = C / 8;                  // 16-byte pieces per pixel row
    static constexpr int NCF = C >= 96 ? 6 : 3;        // input-channel fragments of one data-gradient job (96- / 48-channel chunk)
    static constexpr int NCHK = NC / NCF;
    static constexpr int NPG = TP / 16;                // 16-pixel groups per tile
    static constexpr int JOBS = NPG * NCHK / 8;        // data-gradient jobs per wave: 1, 1, 2
    // weight gradient: 8 waves = WKS (K steps taken in turn) x GA (output-channel groups) x GB (input-channel groups)
    static constexpr int WKS = C == 192 ? 1 : C == 96 ? 2 : 8;
    static constexpr int GA = C == 192 ? 4 : C == 96 ? 2 : 1;
    static constexpr int GB = C == 192 ? 2 : C == 96 ? 2 : 1;
    static constexpr int NA = NC / GA, NB = NC / GB;   // fragments per wave: 3 x 6, 3 x 3, 3 x 3
    static constexpr int WBYTES = NC * KS * 1024;      // the weights in A-fragment order
    static constexpr int YB = NBLK * NSUB * BP_SUB, XB = NBLK * NC * BP_SUB;
    static constexpr int COEF = 5 * C * 4;             // scale, shift, cB, mean, cD
    static constexpr int LDS = WBYTES + YB + XB + COEF;
    static_assert(WKS * GA * GB == 8 && NC % GA == 0 && NC % GB == 0 && NBLK % WKS == 0, "wave decomposition");
    static_assert(NPG * NCHK % 8 == 0 && TP * CPR == 3 * BP_THREADS, "tile decomposition");
};

// local channel of accumulator register 0 of fragment a for the lanes fq = lane >> 4 (conv_pw_kernel's permutation: the
// four lanes of a pixel hold 4 x 8 consecutive channels of a fragment PAIR, so one store instruction writes 64 contiguous
// bytes per pixel); the weight rows are staged with the same function
template <int NCF>
__device__ __forceinline__ constexpr int bp_lch(int a, int fq) {
    const int p = a >> 1;
    return (2 * p + 1 < NCF) ? p * 32 + fq * 8 + (a & 1) * 4 : p * 32 + fq * 4;
}

__device__ __forceinline__ s16x4_t bp_tr_read(const unsigned char* sub, int lane_off, int rowblk) {
    const unsigned char* p = sub + lane_off + rowblk * 512;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p));
}

#ifdef BP_TIMING      /* experiment build (tools/bwd_pw_bench.py with Y5M_LIB): s_memtime stamps of workgroups 0 and 255 */
__device__ unsigned long long g_bp_t[2][8];
extern "C" int y5m_debug_bp_timing(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bp_t), sizeof(g_bp_t)) == hipSuccess ? 0 : -1;
}
#define BP_STAMP(i) do { if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) g_bp_t[blockIdx.x != 0][i] = __builtin_readcyclecounter(); } while (0)
#else
#define BP_STAMP(i)
#endif

// R4 (Y5M_R4_KERNELS bit 3, y5m_common.h): the next tile's pieces re-requested unconditionally and rows behind M masked with a bit
// mask (round 4, not yet measured on hardware); false = the round-3 form (branch around the re-request, select), hardware-verified.
template <int C, bool OLD, bool R4>
__global__ __launch_bounds__(BP_THREADS, 2) void bwd_pw_kernel(const BwdPwParams P, const int ntiles) {
    using G = BpCfg<C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wsm = smem;
    unsigned char* const ytile = smem + G::WBYTES;
    unsigned char* const xtile = ytile + G::YB;
    float* const coef = reinterpret_cast<float*>(xtile + G::XB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    BP_STAMP(0);
    // ---- prologue 1: the coefficients of dy = scale dt + cB (y - mean) + cD, dt = dz silu'(scale y + shift) ------------
    {
        const float invM = 1.0f / (float)P.M;
        for (int c = tid; c < C; c += BP_THREADS) {
            const int sg = (P.nseg > 1 && c >= P.seg[1].c0) ? 1 : 0;
            const int cl = c - P.seg[sg].c0;
            double da, db;
            bnf_sum(P.seg[sg].acc, P.seg[sg].cn, cl, da, db);
            const float is = P.seg[sg].invstd[cl], s1 = P.seg[sg].scale[cl];
            const float dbeta = (float)da;
            const float dgamma = is * (float)db;
            coef[0 * C + c] = s1;
            coef[1 * C + c] = P.seg[sg].shift[cl];
            coef[2 * C + c] = -s1 * dgamma * is * invM;
            coef[3 * C + c] = P.seg[sg].mean[cl];
            coef[4 * C + c] = -s1 * dbeta * invM;
            if (blockIdx.x == 0) {
                if (P.seg[sg].dbeta) P.seg[sg].dbeta[cl] = dbeta;
                if (P.seg[sg].dgamma) P.seg[sg].dgamma[cl] = dgamma;
            }
        }
    }
    // ---- prologue 2: data-gradient weights -> LDS in A-fragment order; zero column of the dy tile ----------------------
    if (P.dx) {
        const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(P.wd);
        constexpr int NF = (G::NC * G::KS + 7) / 8;          // fragments per wave: all loads first, then the LDS stores
        u32x4 wv[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = wid + 8 * i;
            const int ag = f / G::KS, s = f - ag * G::KS;
            const int chunk = ag / G::NCF, a = ag - chunk * G::NCF;
            const int ci = chunk * (G::NCF * 16) + bp_lch<G::NCF>(a, fr >> 2) + (fr & 3);
            const int k0 = s * 32 + fq * 8;
            wv[i] = (u32x4){0u, 0u, 0u, 0u};
            if (f < G::NC * G::KS && k0 < P.N) wv[i] = *reinterpret_cast<const u32x4*>(W + (size_t)ci * P.Kp + k0);
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = wid + 8 * i;
            if (f < G::NC * G::KS) *reinterpret_cast<u32x4*>(wsm + ((size_t)f * 64 + lane) * 16) = wv[i];
        }
    }
    if constexpr (G::NSUB * 16 > C) {
        for (int i = tid; i < G::NBLK * 64; i += BP_THREADS) {                 // 64 x 16 B per padding sub-tile
            const int blk = i >> 6, q = i & 63;
            *reinterpret_cast<u32x4*>(ytile + (blk * G::NSUB + G::NSUB - 1) * BP_SUB + q * 16) = (u32x4){0u, 0u, 0u, 0u};
        }
    }

    // ---- streaming state ---------------------------------------------------------------------------------------------
    // this thread's three 16-byte pieces of a tile: pixel inside the tile, piece of the row, LDS offsets; dz comes from the
    // piece's OWN segment tensor (the two halves of a merged pair keep their output gradients in different buffers)
    int pp[3], pc[3];
    unsigned lofs_y[3], lofs_x[3];
    const bf16_t* dzp[3];
    int dzld[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = tid + BP_THREADS * i;
        pp[i] = j / G::CPR;
        pc[i] = j - pp[i] * G::CPR;
        lofs_y[i] = (unsigned)(((pp[i] >> 5) * G::NSUB + (pc[i] >> 1)) * BP_SUB + (pp[i] & 31) * 32 + (pc[i] & 1) * 16);
        lofs_x[i] = (unsigned)(((pp[i] >> 5) * G::NC + (pc[i] >> 1)) * BP_SUB + (pp[i] & 31) * 32 + (pc[i] & 1) * 16);
        const int c = pc[i] * 8;
        const int sg = (P.nseg > 1 && c >= P.seg[1].c0) ? 1 : 0;
        dzp[i] = reinterpret_cast<const bf16_t*>(P.seg[sg].dz) + (c - P.seg[sg].c0);
        dzld[i] = P.seg[sg].lddz;
    }
    const bf16_t* __restrict__ Yp = reinterpret_cast<const bf16_t*>(P.y);
    const bf16_t* __restrict__ Xp = reinterpret_cast<const bf16_t*>(P.x);
    u32x4 rdz[3], ry[3], rx[3];
    auto issue_piece = [&](int tile, int i) __attribute__((always_inline)) {
        long long m = (long long)tile * G::TP + pp[i];
        m = m < P.M ? m : P.M - 1;                      // rows behind M: any valid row (phase A zeroes their dy, nothing of them is stored)
        rdz[i] = *reinterpret_cast<const u32x4*>(dzp[i] + m * dzld[i]);
        ry[i] = *reinterpret_cast<const u32x4*>(Yp + m * P.ldy + pc[i] * 8);
        rx[i] = *reinterpret_cast<const u32x4*>(Xp + m * P.ldx + pc[i] * 8);
    };

    // weight-gradient accumulators (whole launch)
    const int wk = wid % G::WKS, ga = (wid / G::WKS) % G::GA, gb = wid / (G::WKS * G::GA);
    f32x4 accw[G::NA][G::NB];
#pragma unroll
    for (int a = 0; a < G::NA; ++a)
#pragma unroll
        for (int b = 0; b < G::NB; ++b) accw[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane_off = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;

    int tile = blockIdx.x;
    if (tile < ntiles) {
#pragma unroll
        for (int i = 0; i < 3; ++i) issue_piece(tile, i);
    }
    __syncthreads();                                    // coefficients, weights and the zero column are in place
    BP_STAMP(1);
    for (; tile < ntiles; tile += gridDim.x) {
        // ---- phase A: dy from (dz, y), dy and x into the LDS tiles ---------------------------------------------------
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bool in = (long long)tile * G::TP + pp[i] < P.M;
            const unsigned inm = in ? 0xffffffffu : 0u;                                         // all ones for a row in front of M
            const int c0 = pc[i] * 8;
            float sc[8], sh[8], kb[8], mu[8], kd[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(coef + 0 * C + c0 + 4 * h);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(coef + 1 * C + c0 + 4 * h);
                const f32x4 a2 = *reinterpret_cast<const f32x4*>(coef + 2 * C + c0 + 4 * h);
                const f32x4 a3 = *reinterpret_cast<const f32x4*>(coef + 3 * C + c0 + 4 * h);
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(coef + 4 * C + c0 + 4 * h);
#pragma unroll
                for (int k = 0; k < 4; ++k) { sc[4 * h + k] = a0[k]; sh[4 * h + k] = a1[k]; kb[4 * h + k] = a2[k]; mu[4 * h + k] = a3[k]; kd[4 * h + k] = a4[k]; }
            }
            float dy[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned gz = rdz[i][q], gy = ry[i][q];
                const float z0 = __uint_as_float(gz << 16), z1 = __uint_as_float(gz & 0xffff0000u);
                const float y0 = __uint_as_float(gy << 16), y1 = __uint_as_float(gy & 0xffff0000u);
                const float t0 = z0 * silu_grad(y0 * sc[2 * q] + sh[2 * q]);
                const float t1 = z1 * silu_grad(y1 * sc[2 * q + 1] + sh[2 * q + 1]);
                dy[2 * q] = sc[2 * q] * t0 + kb[2 * q] * (y0 - mu[2 * q]) + kd[2 * q];
                dy[2 * q + 1] = sc[2 * q + 1] * t1 + kb[2 * q + 1] * (y1 - mu[2 * q + 1]) + kd[2 * q + 1];
            }
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                                       // rows behind M contribute nothing
                if constexpr (R4) o[q] = f32x2_to_bf16x2(dy[2 * q], dy[2 * q + 1]) & inm;
                else o[q] = in ? f32x2_to_bf16x2(dy[2 * q], dy[2 * q + 1]) : 0u;
            }
            *reinterpret_cast<u32x4*>(ytile + lofs_y[i]) = o;
            *reinterpret_cast<u32x4*>(xtile + lofs_x[i]) = rx[i];
            // this piece's registers are free: the next tile's piece is requested right away and flies under the rest of
            // phase A and phases C and B. R4: UNCONDITIONALLY (behind the last tile the clamped row M - 1 is read once more and never
            // used): with a branch around the request the compiler cannot count the loads across it and waits with vmcnt(0) for the
            // last group of a tile -- i.e. also for the two groups it has just re-requested (round 4, tools/isa_audit.py)
            if constexpr (R4) issue_piece(tile + (int)gridDim.x, i);
            else { if (tile + (int)gridDim.x < ntiles) issue_piece(tile + gridDim.x, i); }
        }
        __syncthreads();

        // ---- phase C: weight gradient, K = the tile's pixels -------------------------------------------------------------
#pragma unroll
        for (int kk = 0; kk < G::NBLK / G::WKS; ++kk) {
            const int ks = wk + kk * G::WKS;
            uint4 ya[G::NA], xb[G::NB];
#pragma unroll
            for (int a = 0; a < G::NA; ++a) {
                const unsigned char* sub = ytile + (ks * G::NSUB + ga * G::NA + a) * BP_SUB;
                const s16x4_t lo = bp_tr_read(sub, lane_off, 0), hi = bp_tr_read(sub, lane_off, 1);
                ya[a] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
            }
#pragma unroll
            for (int b = 0; b < G::NB; ++b) {
                const unsigned char* sub = xtile + (ks * G::NC + gb * G::NB + b) * BP_SUB;
                const s16x4_t lo = bp_tr_read(sub, lane_off, 0), hi = bp_tr_read(sub, lane_off, 1);
                xb[b] = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
            }
#pragma unroll
            for (int a = 0; a < G::NA; ++a)
#pragma unroll
                for (int b = 0; b < G::NB; ++b)
                    accw[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ya[a]),
                                                                         __builtin_bit_cast(bf16x8_t, xb[b]), accw[a][b], 0, 0, 0);
        }

        // ---- phase B: data gradient of the tile ------------------------------------------------------------------------------
        if (P.dx) {
#pragma unroll
            for (int jb = 0; jb < G::JOBS; ++jb) {
                const int job = wid + 8 * jb;
                const int pg = job % G::NPG, chunk = job / G::NPG;
                const long long m = (long long)tile * G::TP + pg * 16 + fr;
                const bool in = m < P.M;
                bf16_t* o = reinterpret_cast<bf16_t*>(P.dx) + (size_t)m * P.lddx + chunk * (G::NCF * 16);
                u32x2 old[OLD ? G::NCF : 1];
                if constexpr (OLD) {
                    const bf16_t* src = P.res ? reinterpret_cast<const bf16_t*>(P.res) + (size_t)m * P.ldres + chunk * (G::NCF * 16) : o;
#pragma unroll
                    for (int a = 0; a < G::NCF; ++a)
                        old[a] = in ? *reinterpret_cast<const u32x2*>(src + bp_lch<G::NCF>(a, fq)) : (u32x2){0u, 0u};
                }
                f32x4 acc[G::NCF];
#pragma unroll
                for (int a = 0; a < G::NCF; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const int px = pg * 16 + fr;
#pragma unroll
                for (int s = 0; s < G::KS; ++s) {
                    const int cc = s * 4 + fq;
                    const u32x4 bfrag = *reinterpret_cast<const u32x4*>(
                        ytile + ((px >> 5) * G::NSUB + (cc >> 1)) * BP_SUB + (px & 31) * 32 + (cc & 1) * 16);
#pragma unroll
                    for (int a = 0; a < G::NCF; ++a) {
                        const u32x4 w = *reinterpret_cast<const u32x4*>(wsm + ((size_t)((chunk * G::NCF + a) * G::KS + s) * 64 + lane) * 16);
                        acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w),
                                                                         __builtin_bit_cast(bf16x8_t, bfrag), acc[a], 0, 0, 0);
                    }
                }
                u32x2 qa[G::NCF];
#pragma unroll
                for (int a = 0; a < G::NCF; ++a) {
                    float v[4] = {acc[a][0], acc[a][1], acc[a][2], acc[a][3]};
                    if constexpr (OLD) {
                        v[0] += __uint_as_float(old[a].x << 16); v[1] += __uint_as_float(old[a].x & 0xffff0000u);
                        v[2] += __uint_as_float(old[a].y << 16); v[3] += __uint_as_float(old[a].y & 0xffff0000u);
                    }
                    qa[a].x = f32x2_to_bf16x2(v[0], v[1]);
                    qa[a].y = f32x2_to_bf16x2(v[2], v[3]);
                }
                if (in) {
#pragma unroll
                    for (int a = 0; a < G::NCF; a += 2) {
                        if (a + 1 < G::NCF) {
                            const u32x4 q4 = {qa[a].x, qa[a].y, qa[a + 1].x, qa[a + 1].y};
                            *reinterpret_cast<u32x4*>(o + bp_lch<G::NCF>(a, fq)) = q4;
                        } else {
                            *reinterpret_cast<u32x2*>(o + bp_lch<G::NCF>(a, fq)) = qa[a];
                        }
                    }
                }
            }
        }
        __syncthreads();                                 // every wave is done reading the tiles
    }

    BP_STAMP(2);
    // ---- weight gradient: f32 atomics into [N][lddw] (rows of the segment's own tensor). Every workgroup adds the same C x C
    // tile, so the order is ROTATED by workgroup (fragment row a first: blockIdx % NA; input-channel fragments forwards or
    // backwards): the workgroups that arrive here together do not walk the same addresses in lock step
    // K-waves first (WKS > 1: waves that split the tile's pixels hold partial tiles of the SAME dW block): summed in LDS --
    // the tiles and the weights are dead now -- so that a workgroup issues ONE set of atomics; with 8 K-waves (C = 48) the
    // 2048 waves of the launch otherwise queue on the same 2304 addresses (measured 35-50 us of a 180 us launch)
    if constexpr (G::WKS > 1) {
        constexpr int TILE_F = G::NA * G::NB * 256;                // floats of one wave's accumulators
        static_assert((G::WKS - 1) * G::GA * G::GB * TILE_F * 4 <= G::WBYTES + G::YB + G::XB, "reduction buffer fits the dead tiles");
        float* red = reinterpret_cast<float*>(smem);              // [(WKS-1)][GA*GB][TILE_F]
        const int grp = ga + G::GA * gb;
        if (wk > 0) {
            float* dstp = red + ((size_t)(wk - 1) * (G::GA * G::GB) + grp) * TILE_F + lane;
#pragma unroll
            for (int a = 0; a < G::NA; ++a)
#pragma unroll
                for (int b = 0; b < G::NB; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dstp[((a * G::NB + b) * 4 + r) * 64] = accw[a][b][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int k = 1; k < G::WKS; ++k) {
                const float* srcp = red + ((size_t)(k - 1) * (G::GA * G::GB) + grp) * TILE_F + lane;
#pragma unroll
                for (int a = 0; a < G::NA; ++a)
#pragma unroll
                    for (int b = 0; b < G::NB; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) accw[a][b][r] += srcp[((a * G::NB + b) * 4 + r) * 64];
            }
        }
    }
    if (wk == 0) {
        const int i = lane & 15, g = lane >> 4;
        auto add_rows = [&](auto A0, auto REV) __attribute__((always_inline)) {
#pragma unroll
            for (int aa = 0; aa < G::NA; ++aa) {
                constexpr int a0 = decltype(A0)::value;
                const int a = (aa + a0) % G::NA;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = (ga * G::NA + a) * 16 + g * 4 + r;
                    const int sg = (P.nseg > 1 && n >= P.seg[1].c0) ? 1 : 0;
                    float* row = P.seg[sg].dw + (size_t)(n - P.seg[sg].c0) * P.lddw;
#pragma unroll
                    for (int bb = 0; bb < G::NB; ++bb) {
                        const int b = decltype(REV)::value ? G::NB - 1 - bb : bb;
                        atomicAdd(row + (gb * G::NB + b) * 16 + i, accw[a][b][r]);
                    }
                }
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using F = std::integral_constant<bool, false>; using T = std::integral_constant<bool, true>;
        static_assert(G::NA == 3, "three fragment rows per wave in every geometry");
        switch (blockIdx.x % 6) {
        case 0: add_rows(I0{}, F{}); break;
        case 1: add_rows(I1{}, T{}); break;
        case 2: add_rows(I2{}, F{}); break;
        case 3: add_rows(I0{}, T{}); break;
        case 4: add_rows(I1{}, F{}); break;
        default: add_rows(I2{}, T{}); break;
        }
    }
#ifdef BP_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BP_STAMP(3);
#endif
}

static bool bp_eligible(const BwdPwParams& P, int dtype) {
    if (dtype != Y5M_BF16) return false;
    if (P.N != P.C || !(P.C == 48 || P.C == 96 || P.C == 192)) return false;
    if (P.M <= 0 || P.nseg < 1 || P.nseg > 2 || P.act != Y5M_ACT_SILU) return false;
    if (P.ldy % 8 != 0 || P.ldx % 8 != 0 || !P.y || !P.x) return false;
    if (P.ldy < P.N || P.ldx < P.C) return false;                                // a pixel row holds at least the channels read
    if ((reinterpret_cast<uintptr_t>(P.y) | reinterpret_cast<uintptr_t>(P.x)) & 15) return false;
    if (P.dx) {
        if (!P.wd || P.Kp < (P.N + 7) / 8 * 8 || P.Kp % 8 != 0) return false;     // the 16-byte pieces with k0 < N stay inside a row
        if (P.lddx % 8 != 0 || P.lddx < P.C || (reinterpret_cast<uintptr_t>(P.dx) & 15) != 0) return false;
        if (P.res && (P.ldres < P.C || P.ldres % 4 != 0 || (reinterpret_cast<uintptr_t>(P.res) & 7) != 0)) return false;
    }
    int covered = 0;
    for (int s = 0; s < P.nseg; ++s) {
        const auto& S = P.seg[s];
        if (S.c0 != covered || S.cn <= 0 || S.cn % 16 != 0 || !S.acc || !S.scale || !S.shift || !S.mean || !S.invstd || !S.dw) return false;
        if (!S.dz || S.lddz % 8 != 0 || S.lddz < S.cn || (reinterpret_cast<uintptr_t>(S.dz) & 15) != 0) return false;
        covered += S.cn;
    }
    return covered == P.N && P.lddw >= P.C;
}

extern "C" int y5m_bwd_pw_eligible(const y5m_bwd_pw_args* args, int dtype) { return bp_eligible(*args, dtype) ? 1 : 0; }

template <int C>
static int launch_bp(const BwdPwParams& P, hipStream_t st) {
    using G = BpCfg<C>;
    const int ntiles = (int)((P.M + G::TP - 1) / G::TP);
    int grid = y5m_persistent_cus();          // one 8-wave workgroup per CU (the 48- / 96-channel forms would need <= 128 VGPRs for two: they spill)
    if (grid > ntiles) grid = ntiles;
    const bool old = P.dx && (P.accumulate || P.res);
    const bool r4 = (y5m_r4_forms() & Y5M_R4_BWD_PW) != 0;
    auto k0 = r4 ? bwd_pw_kernel<C, false, true> : bwd_pw_kernel<C, false, false>;
    auto k1 = r4 ? bwd_pw_kernel<C, true, true> : bwd_pw_kernel<C, true, false>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        attr = true;
    }
    Y5M_NAME_ONLY(Y5M_OK, "bwd_pw_kernel<%d,%d,%d>", C, (int)old, (int)r4);
    if (old) hipLaunchKernelGGL(k1, dim3((unsigned)grid), dim3(BP_THREADS), G::LDS, st, P, ntiles);
    else hipLaunchKernelGGL(k0, dim3((unsigned)grid), dim3(BP_THREADS), G::LDS, st, P, ntiles);
    Y5M_CHECK_LAUNCH("bwd_pw_kernel");
    return Y5M_OK;
}

extern "C" int y5m_bwd_pw(const y5m_bwd_pw_args* args, int dtype, void* stream) {
    const BwdPwParams& P = *args;
    if (!bp_eligible(P, dtype)) { y5m_set_error("y5m_bwd_pw: arguments not eligible (see y5m_bwd_pw_eligible)"); return Y5M_EINVAL; }
    hipStream_t st = y5m_stream(stream);
    if (P.C == 192) return launch_bp<192>(P, st);
    if (P.C == 96) return launch_bp<96>(P, st);
    return launch_bp<48>(P, st);
}
