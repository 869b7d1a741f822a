// Fused backward of the STEM CBL (reference model.py:181 CBL(3, first_out, 6, 2, 2); executed as a 3x3 / stride 1 / pad 1 conv over
// the 16-channel 2x2 space-to-depth image, so its weight gradient is dW[48][9 taps x 16]), bf16:
//
//     dy = BatchNorm+SiLU backward of (dz, y)      -- never written to HBM (the stem has no data gradient: dy had ONE reader)
//     dW += dy^T . x(tap)                           -- f32 atomics into the packed gradient, 9 taps
//
// The stem is the LAST unit of the backward pass: its bn_bwd_apply (reads dz, y, writes dy: 1.9 GB at B=64 @ 640^2, 374 us) and
// its weight gradient (reads dy, x: 0.84 GB, 298 us, ALONE on the chip behind the last main-stream kernel) were the exposed tail
// of the step. Fused, dz / y / x are read once (1.47 GB). Structure = bwd_pw_kernel's streaming (one 8-wave workgroup per CU,
// every 16-byte piece re-requested for the next chunk as soon as it has been consumed, dy formed on the VALU into an LDS tile)
// with wgrad_rows_kernel's geometry: pixels in RUNS of 32 output pixels of one row, a run's X pixels are three contiguous
// 34-pixel row segments (1088 B each) staged as they lie in memory, tap (ty, tx) = segment ty read at pixel offset tx by the
// transposing LDS read. One run per wave and chunk: 9 taps x 48 x 16 outputs = 27 accumulator fragments per wave for the whole
// launch, added across the 8 waves in LDS before ONE set of atomics per workgroup.
// The BatchNorm reduction needs its own pass over (dz, y) ahead of this kernel (y5m_bn_bwd_fused_phase, phase 1).
#include "y5m_conv.h"
#include <string.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef y5m_bwd_stem_args BwdStemParams;

#define BS_THREADS 512
#define BS_SUB 1088                        // [32 pixel][16 channel] sub-tile + 64 (see y5m_conv_wgrad.hip)
#define BS_RUNS 8                          // runs per chunk = waves
#define BS_NPX 34                          // X pixels of a run (32 + the two horizontal halo pixels)
#define BS_XG (BS_NPX * 32 + 64)           // one row segment of a run in LDS
#define BS_N 48
#define BS_YB (BS_RUNS * 3 * BS_SUB)
#define BS_XB (BS_RUNS * 3 * BS_XG)
#define BS_COEF (5 * BS_N * 4)
#define BS_NF 27                           // accumulator fragments per wave: 9 taps x 3 dY fragments x 1 X fragment
#define BS_TILE_F (BS_NF * 4 * 64)
#define BS_RED (4 * BS_TILE_F * 4)         // first round of the wave reduction: 4 tiles
#define BS_DUMMY 64                         // where the pieces behind the X tile (threads 1632..2047 of the fourth round) store
#define BS_LDS ((BS_YB + BS_XB + BS_COEF + BS_DUMMY) > BS_RED ? (BS_YB + BS_XB + BS_COEF + BS_DUMMY) : BS_RED)
#define BS_YP (BS_RUNS * 32 * 6)           // 16-byte pieces of dz (and of y) per chunk: 1536 = 3 per thread
#define BS_XP (BS_RUNS * 3 * BS_NPX * 2)   // of x: 1632 -> 4 per thread, the last partly empty

// R4 (Y5M_R4_KERNELS bit 2, y5m_common.h): the streaming loop as straight-line code (round 4, not yet measured on hardware);
// false = the round-3 form (branches around the re-requests and the X stores), hardware-verified.
template <bool R4>
__global__ __launch_bounds__(BS_THREADS, 1) void bwd_stem_kernel(const BwdStemParams P, const int spr, const int nruns, const int nchunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ytile = smem;
    unsigned char* const xtile = smem + BS_YB;
    float* const coef = reinterpret_cast<float*>(smem + BS_YB + BS_XB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = P.B * P.H * P.W;

    // ---- prologue: the coefficients of dy = scale dt + cB (y - mean) + cD, dt = dz silu'(scale y + shift) (as bwd_pw_kernel) ----
    {
        const float invM = 1.0f / (float)M;
        for (int c = tid; c < BS_N; c += BS_THREADS) {
            double da, db;
            bnf_sum(P.acc, BS_N, c, da, db);
            const float is = P.invstd[c], s1 = P.scale[c];
            const float dbeta = (float)da;
            const float dgamma = is * (float)db;
            coef[0 * BS_N + c] = s1;
            coef[1 * BS_N + c] = P.shift[c];
            coef[2 * BS_N + c] = -s1 * dgamma * is * invM;
            coef[3 * BS_N + c] = P.mean[c];
            coef[4 * BS_N + c] = -s1 * dbeta * invM;
            if (blockIdx.x == 0) {
                if (P.dbeta) P.dbeta[c] = dbeta;
                if (P.dgamma) P.dgamma[c] = dgamma;
            }
        }
    }

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(P.dz), 0, (unsigned)((size_t)M * P.lddz * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(P.y), 0, (unsigned)((size_t)M * P.ldy * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(P.x), 0, (unsigned)((size_t)M * P.ldx * 2), 0x00020000);
    // piece -> (run, pixel, 16-byte piece): loop invariant
    int zrun[3], zpx[3], zcc[3], zofs[3];
    int xrun[4], xty[4], xpx[4], xhalf[4], xofs[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = tid + BS_THREADS * i;
        zrun[i] = q / 192;
        const int rem = q % 192;
        zpx[i] = rem / 6;
        zcc[i] = rem % 6;
        zofs[i] = (zrun[i] * 3 + (zcc[i] >> 1)) * BS_SUB + zpx[i] * 32 + (zcc[i] & 1) * 16;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + BS_THREADS * i;
        xrun[i] = q < BS_XP ? q / (3 * BS_NPX * 2) : -1;
        const int rem = q % (3 * BS_NPX * 2);
        xty[i] = rem / (BS_NPX * 2);
        const int rem2 = rem % (BS_NPX * 2);
        xpx[i] = rem2 >> 1;
        xhalf[i] = rem2 & 1;
        // (a piece behind the tile stores into the dummy region behind the coefficients: the store is unconditional)
        if constexpr (R4) xofs[i] = q < BS_XP ? (xrun[i] * 3 + xty[i]) * BS_XG + xpx[i] * 32 + xhalf[i] * 16 : BS_XB + BS_COEF + (lane & 3) * 16;
        else xofs[i] = ((q < BS_XP ? xrun[i] : 0) * 3 + xty[i]) * BS_XG + xpx[i] * 32 + xhalf[i] * 16;
    }
    const float rcpS = 1.0f / (float)spr, rcpH = 1.0f / (float)P.H;
    const unsigned ldzb = (unsigned)(P.lddz * 2), ldyb = (unsigned)(P.ldy * 2), ldxb = (unsigned)(P.ldx * 2);
    u32x4 rz[3], ry[3], rx[4];
    unsigned okm = 0u;                                    // bit i: piece i of (dz, y) in flight is a real pixel
    // The streaming loop below is STRAIGHT-LINE code (round 4): "is there a next chunk", "is this piece a real pixel" and "is
    // this piece inside the tile" are folded into load offsets (an out-of-range offset reads zeros and moves no data), a bit mask
    // and a dummy store address. With branches around the loads the compiler could not count them across the blocks and waited
    // with `s_waitcnt vmcnt(0)` at every use -- 21 full waits per chunk (tools/isa_audit.py), each of them also waiting for the
    // pieces that had JUST been re-requested for the next chunk: a memory round trip per piece instead of one per chunk.
    auto issue_zy = [&](int chunk, bool more, int i) __attribute__((always_inline)) {
        const int u = chunk * BS_RUNS + zrun[i];
        int row, j;
        fast_divmod(u, spr, rcpS, row, j);                // row = b * H + oy
        const int ox = 32 * j + zpx[i];
        const bool ok = more && u < nruns && ox < P.W;
        const unsigned m = (unsigned)(row * P.W + ox);
        rz[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, ok ? m * ldzb + (unsigned)(zcc[i] * 16) : OOB, 0, 0);
        ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, ok ? m * ldyb + (unsigned)(zcc[i] * 16) : OOB, 0, 0);
        okm = (okm & ~(1u << i)) | ((ok ? 1u : 0u) << i);
    };
    auto issue_x = [&](int chunk, bool more, int i) __attribute__((always_inline)) {
        const int u = chunk * BS_RUNS + xrun[i];
        int row, j, b, oy;
        fast_divmod(u, spr, rcpS, row, j);
        fast_divmod(row, P.H, rcpH, b, oy);
        const int iy = oy + xty[i] - 1, ix = 32 * j + xpx[i] - 1;
        const bool ok = more && xrun[i] >= 0 && u < nruns && (unsigned)iy < (unsigned)P.H && (unsigned)ix < (unsigned)P.W;
        const unsigned pix = __umul24((unsigned)(b * P.H + iy), (unsigned)P.W) + (unsigned)ix;
        rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? __umul24(pix, ldxb) + (unsigned)(xhalf[i] * 16) : OOB, 0, 0);
    };

    f32x4 acc[9][3];                                      // [tap][dY fragment]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[t][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int lrow = 4 * (lane >> 4) + ((lane & 15) >> 2), lb = (lane & 3) * 8;
    const int l_off = lrow * 32 + lb;
    auto tr8 = [&](const unsigned char* p) __attribute__((always_inline)) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p + 512));
        return make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
    };

    int chunk = blockIdx.x;
    if constexpr (R4) {
        const bool first = chunk < nchunks;
#pragma unroll
        for (int i = 0; i < 3; ++i) issue_zy(chunk, first, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_x(chunk, first, i);
    } else if (chunk < nchunks) {
#pragma unroll
        for (int i = 0; i < 3; ++i) issue_zy(chunk, true, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_x(chunk, true, i);
    }
    __syncthreads();                                      // coefficients are in place
    for (; chunk < nchunks; chunk += gridDim.x) {
        const int next = chunk + (int)gridDim.x;
        const bool more = next < nchunks;
        // ---- phase A: dy from (dz, y) into the dY tile, x into the X tile; every piece is re-requested for the next chunk ------
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c0 = zcc[i] * 8;
            float sc[8], sh[8], kb[8], mu[8], kd[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(coef + 0 * BS_N + c0 + 4 * h);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(coef + 1 * BS_N + c0 + 4 * h);
                const f32x4 a2 = *reinterpret_cast<const f32x4*>(coef + 2 * BS_N + c0 + 4 * h);
                const f32x4 a3 = *reinterpret_cast<const f32x4*>(coef + 3 * BS_N + c0 + 4 * h);
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(coef + 4 * BS_N + c0 + 4 * h);
#pragma unroll
                for (int k = 0; k < 4; ++k) { sc[4 * h + k] = a0[k]; sh[4 * h + k] = a1[k]; kb[4 * h + k] = a2[k]; mu[4 * h + k] = a3[k]; kd[4 * h + k] = a4[k]; }
            }
            const bool in = (okm >> i) & 1u;
            const unsigned inm = 0u - ((okm >> i) & 1u);   // all ones for a real pixel
            float dy[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned gz = rz[i][q], gy = ry[i][q];
                const float z0 = __uint_as_float(gz << 16), z1 = __uint_as_float(gz & 0xffff0000u);
                const float y0 = __uint_as_float(gy << 16), y1 = __uint_as_float(gy & 0xffff0000u);
                const float t0 = z0 * silu_grad(y0 * sc[2 * q] + sh[2 * q]);
                const float t1 = z1 * silu_grad(y1 * sc[2 * q + 1] + sh[2 * q + 1]);
                dy[2 * q] = sc[2 * q] * t0 + kb[2 * q] * (y0 - mu[2 * q]) + kd[2 * q];
                dy[2 * q + 1] = sc[2 * q + 1] * t1 + kb[2 * q + 1] * (y1 - mu[2 * q + 1]) + kd[2 * q + 1];
            }
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                                         // masked pixels contribute nothing
                if constexpr (R4) o[q] = f32x2_to_bf16x2(dy[2 * q], dy[2 * q + 1]) & inm;
                else o[q] = in ? f32x2_to_bf16x2(dy[2 * q], dy[2 * q + 1]) : 0u;
            }
            *reinterpret_cast<u32x4*>(ytile + zofs[i]) = o;
            if constexpr (R4) issue_zy(next, more, i);
            else { if (next < nchunks) issue_zy(next, true, i); }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (R4) {
                *reinterpret_cast<u32x4*>(xtile + xofs[i]) = rx[i];
                issue_x(next, more, i);
            } else {
                if (xrun[i] >= 0) *reinterpret_cast<u32x4*>(xtile + xofs[i]) = rx[i];
                if (next < nchunks) issue_x(next, true, i);
            }
        }
        __syncthreads();
        // ---- phase C: this wave's run: 9 taps x 3 dY fragments, K = the run's 32 pixels ------------------------------------------
        {
            const unsigned char* Ys = ytile + wid * 3 * BS_SUB + l_off;
            const unsigned char* Xs = xtile + wid * 3 * BS_XG + l_off;
            uint4 ya[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) ya[a] = tr8(Ys + a * BS_SUB);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const uint4 xb = tr8(Xs + (t / 3) * BS_XG + (t % 3) * 32);
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    acc[t][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ya[a]),
                                                                        __builtin_bit_cast(bf16x8_t, xb), acc[t][a], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- the 8 waves add their tiles pairwise in LDS (wave >= st writes, wave - st adds), wave 0 issues the atomics --------------
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int st = 4; st >= 1; st >>= 1) {
        __syncthreads();
        if (wid >= st && wid < 2 * st) {
            float* d = red + (size_t)(wid - st) * BS_TILE_F + lane;
#pragma unroll
            for (int f = 0; f < BS_NF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) d[(f * 4 + r) * 64] = acc[f / 3][f % 3][r];
        }
        __syncthreads();
        if (wid < st) {
            const float* sp = red + (size_t)wid * BS_TILE_F + lane;
#pragma unroll
            for (int f = 0; f < BS_NF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[f / 3][f % 3][r] += sp[(f * 4 + r) * 64];
        }
    }
    if (wid == 0 && (int)blockIdx.x < nchunks) {
        const int i = lane & 15, g = lane >> 4;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    atomicAdd(P.dwgt + (size_t)(a * 16 + g * 4 + r) * P.lddw + t * P.C + i, acc[t][a][r]);
    }
}

extern "C" int y5m_bwd_stem_eligible(const y5m_bwd_stem_args* a) {
    if (!a || !a->dz || !a->y || !a->x || !a->dwgt || !a->acc || !a->scale || !a->shift || !a->mean || !a->invstd) return 0;
    if (a->N != BS_N || a->C != 16 || a->act != Y5M_ACT_SILU) return 0;
    if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->lddz % 8 != 0 || a->ldy % 8 != 0 || a->ldx % 8 != 0 || a->ldx < 16) return 0;
    if (a->lddz < BS_N || a->ldy < BS_N || a->lddw < 9 * 16) return 0;
    const long long M = (long long)a->B * a->H * a->W;
    if (M * a->lddz * 2 >= (1ll << 31) || M * a->ldy * 2 >= (1ll << 31) || M * a->ldx * 2 >= (1ll << 31) || M >= (1ll << 24)) return 0;
    if (((uintptr_t)a->dz | (uintptr_t)a->y | (uintptr_t)a->x) & 15) return 0;
    return 1;
}

extern "C" int y5m_bwd_stem(const y5m_bwd_stem_args* args, void* stream) {
    Y5M_REQUIRE(args != nullptr, "args");
    Y5M_REQUIRE(y5m_bwd_stem_eligible(args), "y5m_bwd_stem: bf16, 48 output channels, 16 input channels per tap, 16-byte aligned views < 2 GiB, "
                                            "< 2^24 pixels, acc / scale / shift / mean / invstd given (see y5m_bwd_stem_eligible)");
    BwdStemParams P = *args;
    const int spr = (P.W + 31) / 32;
    const int nruns = P.B * P.H * spr;
    const int nchunks = (nruns + BS_RUNS - 1) / BS_RUNS;
    int grid = y5m_persistent_cus();
    if (grid > nchunks) grid = nchunks;
    const bool r4 = (y5m_r4_forms() & Y5M_R4_BWD_STEM) != 0;
    auto kern = r4 ? bwd_stem_kernel<true> : bwd_stem_kernel<false>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)bwd_stem_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, BS_LDS);
        (void)hipFuncSetAttribute((const void*)bwd_stem_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, BS_LDS);
        attr = true;
    }
    Y5M_NAME_ONLY(Y5M_OK, "bwd_stem_kernel<%d>", (int)r4);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BS_THREADS), BS_LDS, y5m_stream(stream), P, spr, nruns, nchunks);
    Y5M_CHECK_LAUNCH("bwd_stem_kernel");
    return Y5M_OK;
}
