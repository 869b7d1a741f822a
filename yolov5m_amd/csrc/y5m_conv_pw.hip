// Pointwise (1x1, stride 1, dense in / dense out) convolution for short K (<= 192 input channels), bf16.
//
// Why a second kernel: the tiled implicit-GEMM kernel (y5m_conv.hip) spends 2-6 K steps per 128-pixel
// tile on these layers, so its per-tile fixed work (staging through LDS, 2 barriers per K step, tile
// prologue / epilogue) dominates: with ALL global traffic removed it only gets ~15 % faster, i.e. it is
// neither HBM- nor MFMA-bound there. A pointwise conv is a streaming problem (read x once, write y once,
// ~K flops per byte), so here:
//   * the WEIGHTS of the workgroup's channel chunk (<= 96 channels x K) are staged into LDS once, already
//     in MFMA A-fragment order (lane-linear 16 B per lane: every ds_read_b128 is conflict free);
//   * every WAVE streams its own 16-pixel groups: x fragments go global -> VGPR directly in the MFMA B
//     layout (lane = pixel l&15, 8 channels at (l>>4)*8 of a 32-channel K step), no LDS round trip and
//     NO barrier in the streaming loop; the next group's loads are issued before the current group's MFMAs;
//   * output channels are PERMUTED across the A rows so that a lane's 4*NCF accumulators are 4*NCF
//     CONSECUTIVE channels of its pixel: D row rho = (l>>4)*4 + reg of fragment a holds channel
//     n0 + (rho>>2)*(4*NCF) + a*4 + (rho&3)  ->  8*NCF contiguous bytes per lane, 32*NCF per pixel;
//   * BatchNorm partial sums stay in registers for the whole stream and are reduced across the 16 pixel
//     lanes and the 4 waves once at the end (one stats row per workgroup; unused rows are zeroed).
#include "y5m_conv.h"
#include <string.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define PW_THREADS 256

// NCF: 16-channel fragments per wave (3 -> 48-channel chunk, 6 -> 96); KS: 32-channel K steps.
// OLD: the epilogue also reads a tensor of the output's shape (EPI_DGRAD: accumulate onto the existing
// gradient; EPI_AFFINE_ACT: residual add).
// The streaming loop has NO conditional memory instruction (M % 16 == 0 is a launch precondition and the
// prefetch index is clamped instead of guarded): with every load / store on the straight path the compiler
// can count them, so the wait for the prefetched x fragments is a vmcnt(#stores issued since) and the
// output stores of group g drain under the MFMAs of group g+1 instead of being waited for.
// Local channel (within the workgroup's chunk) of accumulator register 0 of fragment `a` for the lanes with
// fq = lane >> 4 (the same function places the weight rows: A row rho of fragment a holds channel
// pw_lch(a, rho >> 2) + (rho & 3)).
//   PERM (default): fragments are paired; the lanes fq = 0..3 of a pixel hold channels p*32 + fq*8 + [0,8) of
//     pair p = a >> 1, so ONE store instruction writes 64 contiguous bytes per pixel (16 B per lane, the four
//     lanes of a pixel adjacent). With the lane-contiguous layout below the L2 received 21 B per write request
//     (PMC: 7.4 M write requests for 157 MB, 72 % of all its requests) and the stores cost 40 % of the kernel.
//   (!PERM: a lane holds 4*NCF consecutive channels -- 48 B runs, 48 B apart; the layout PERM replaced.)
template <int NCF, bool PERM>
__device__ __forceinline__ constexpr int pw_lch(int a, int fq) {
    if (!PERM) return fq * (4 * NCF) + a * 4;
    const int p = a >> 1;
    return (2 * p + 1 < NCF) ? p * 32 + fq * 8 + (a & 1) * 4 : p * 32 + fq * 4;
}

// TAP: the 3x3 / stride-1 / pad-1 conv over a 16-channel input (the stem on the space-to-depth image: K = 9 taps x 16
// channels = 144 -> KS = 5 steps, the last half step padding): the same barrier-free stream, but every 16-byte piece
// of an x fragment comes from its own tap (k = s*32 + fq*8 -> tap k >> 4 = 2s + (fq >> 1), channel half (fq & 1) * 8)
// with its own image-border test (out-of-image taps read the zero page). The tiled kernel spends 3 K steps per
// 128-pixel tile here and runs at 40 % of the HBM rate this layer is bound by (839 MB in 0.43 ms).
template <int NCF, int KS, int EPI, bool OLD, int TAPC = 0>
__global__ __launch_bounds__(PW_THREADS, 2) void conv_pw_kernel(const ConvParams P, const int nchunks, const int nstreams,
                                                               const int ngroups, const int stat_rows) {
    constexpr int NC = NCF * 16;
    constexpr bool PERM = true;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NCF][KS][64 lanes] x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    // workgroup -> (stream block, channel chunk): the nchunks workgroups that re-read the same pixels run
    // on the SAME XCD (hardware workgroup b is on XCD b % 8; gridDim.x is a multiple of 8 * nchunks)
    const int b = blockIdx.x;
    const int chunk = (b >> 3) % nchunks;
    const int sblock = ((b >> 3) / nchunks) * 8 + (b & 7);
    const int stream = sblock * 4 + wid;
    const int n0 = chunk * NC;

    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(P.in);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(P.w);

    // ---- weights -> LDS in A-fragment order (once) -------------------------------------------------
    for (int f = wid; f < NCF * KS; f += PW_THREADS / 64) {
        const int a = f / KS, s = f - a * KS;
        const int ch = n0 + pw_lch<NCF, PERM>(a, fr >> 2) + (fr & 3);
        const u32x4 v = *reinterpret_cast<const u32x4*>(W + (size_t)ch * P.Kp + s * 32 + fq * 8);
        *reinterpret_cast<u32x4*>(smem + ((size_t)f * 64 + lane) * 16) = v;
    }
    __syncthreads();

    const int cbase = n0 + fq * (4 * NCF);          // (!PERM) the lane's 4*NCF consecutive output channels
    float sc[EPI == EPI_AFFINE_ACT ? 4 * NCF : 1], sh[EPI == EPI_AFFINE_ACT ? 4 * NCF : 1];
    if constexpr (EPI == EPI_AFFINE_ACT) {
#pragma unroll
        for (int j = 0; j < 4 * NCF; ++j) {
            const int c = n0 + pw_lch<NCF, PERM>(j >> 2, fq) + (j & 3);
            sc[j] = P.scale[c]; sh[j] = P.shift[c];
        }
    }
    constexpr bool SUMS = EPI == EPI_RAW_STATS;
    float ssum[SUMS ? 4 * NCF : 1], ssq[SUMS ? 4 * NCF : 1];
    if constexpr (SUMS) {
#pragma unroll
        for (int j = 0; j < 4 * NCF; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
    }

    // ---- x fragment loads -----------------------------------------------------------------------------
    const ptrdiff_t zoff = reinterpret_cast<const bf16_t*>(P.zeros) - X;
    const bool klast_ok = (KS - 1) * 32 + fq * 8 < P.Cin;        // K % 32 == 16: upper half of the last step is padding
    // TAPC > 0 (channels per tap): per K step, this lane's 16-byte piece k0 = st*32 + fq*8 belongs to tap k0 / TAPC at
    // channel k0 % TAPC (TAPC % 8 == 0: a piece never straddles two taps): element offset from the group's base pixel
    // and the tap's (dy, dx) for the border test; K padding behind the last tap is marked by dy = 1 << 20
    constexpr bool TAP = TAPC > 0;
    int tdy[TAP ? KS : 1], tdx[TAP ? KS : 1], tof[TAP ? KS : 1];
    if constexpr (TAP) {
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            const int k0 = st * 32 + fq * 8;
            const int tap = k0 / TAPC, ch = k0 - tap * TAPC;
            const int ta = tap / P.tw, tb = tap - ta * P.tw;
            const bool real = tap < P.th * P.tw;
            tdy[st] = real ? P.dh0 + ta * P.dhs : (1 << 20);
            tdx[st] = P.dw0 + tb * P.dws;
            tof[st] = real ? (tdy[st] * P.Win + tdx[st]) * P.ldin + ch : 0;
        }
    }
    const float rcpW = 1.0f / (float)P.Wg, rcpH = 1.0f / (float)P.Hg;
    auto load_x = [&](u32x4 (&xr)[KS], int g) __attribute__((always_inline)) {
        const size_t p = (size_t)g * 16 + fr;
        if constexpr (TAP) {
            int t, x, b, y;
            fast_divmod((int)p, P.Wg, rcpW, t, x);
            fast_divmod(t, P.Hg, rcpH, b, y);
            y *= P.sy; x *= P.sx;                                  // input position of tap offset (0, 0)
            const ptrdiff_t base = ((ptrdiff_t)(b * P.Hin + y) * P.Win + x) * P.ldin;
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                const int yy = y + tdy[st], xx = x + tdx[st];
                const bool ok = (unsigned)yy < (unsigned)P.Hin && (unsigned)xx < (unsigned)P.Win;
                const ptrdiff_t off = ok ? base + tof[st] : zoff;
                xr[st] = *reinterpret_cast<const u32x4*>(X + off);
            }
        } else {
        const ptrdiff_t off = (ptrdiff_t)(p * P.ldin + fq * 8);
#pragma unroll
        for (int s = 0; s < KS - 1; ++s) xr[s] = *reinterpret_cast<const u32x4*>(X + off + s * 32);   // (nontemporal: measured slower)
        ptrdiff_t last = off + (KS - 1) * 32;
        asm("" : "+v"(last));
        xr[KS - 1] = *reinterpret_cast<const u32x4*>(X + (klast_ok ? last : zoff));
        }
    };

    // one 16-pixel group: x fragments in xr, prefetch of the stream's next group into xp
    auto process = [&](const u32x4 (&xr)[KS], u32x4 (&xp)[KS], int g) __attribute__((always_inline)) {
        const size_t p = (size_t)g * 16 + fr;
        bf16_t* o = reinterpret_cast<bf16_t*>(P.out) + p * P.ldout + n0;     // + pw_lch(a, fq): the lane's channels
        // read-modify-write / residual operand of THIS group first, then the next group's x fragments:
        // the wait for `old` then leaves the prefetch in flight
        u32x2 old[OLD ? NCF : 1];
        if constexpr (OLD) {
            const bf16_t* src = (EPI == EPI_DGRAD && !P.res) ? o : reinterpret_cast<const bf16_t*>(P.res) + p * P.ldres + n0;
#pragma unroll
            for (int a = 0; a < NCF; ++a) old[a] = *reinterpret_cast<const u32x2*>(src + pw_lch<NCF, PERM>(a, fq));
        }
        {
            const int gn = g + nstreams;
            load_x(xp, gn < ngroups ? gn : g);      // clamped, not guarded (the last group is re-read once)
        }
        __builtin_amdgcn_sched_barrier(0);          // the prefetch is ISSUED before the MFMAs, not sunk below them

        f32x4 acc[NCF];
#pragma unroll
        for (int a = 0; a < NCF; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int a = 0; a < NCF; ++a) {
                const u32x4 w = *reinterpret_cast<const u32x4*>(smem + ((size_t)(a * KS + s) * 64 + lane) * 16);
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w),
                                                                 __builtin_bit_cast(bf16x8_t, xr[s]), acc[a], 0, 0, 0);
            }

        // ---- epilogue: lane = pixel p, channels n0 + pw_lch(a, fq) + r --------------------------------
        u32x2 qa[NCF];
#pragma unroll
        for (int a = 0; a < NCF; ++a) {
            float v[4] = {acc[a][0], acc[a][1], acc[a][2], acc[a][3]};
            if constexpr (EPI == EPI_RAW_STATS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { ssum[a * 4 + r] += v[r]; ssq[a * 4 + r] += v[r] * v[r]; }
            }
            if constexpr (EPI == EPI_AFFINE_ACT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = v[r] * sc[a * 4 + r] + sh[a * 4 + r];
                    if (P.act == Y5M_ACT_SILU) v[r] = silu_f(v[r]);
                }
            }
            if constexpr (OLD) {
                v[0] += __uint_as_float(old[a].x << 16); v[1] += __uint_as_float(old[a].x & 0xffff0000u);
                v[2] += __uint_as_float(old[a].y << 16); v[3] += __uint_as_float(old[a].y & 0xffff0000u);
            }
            u32x2 q;
            q.x = f32x2_to_bf16x2(v[0], v[1]);
            q.y = f32x2_to_bf16x2(v[2], v[3]);
            qa[a] = q;
        }
        // stores: one 16-byte piece per fragment pair (PERM: the 4 lanes of a pixel write 64 contiguous bytes)
#pragma unroll
        for (int a = 0; a < NCF; a += 2) {
            if (a + 1 < NCF) {
                if constexpr (PERM) {
                    const u32x4 q4 = {qa[a].x, qa[a].y, qa[a + 1].x, qa[a + 1].y};
                    *reinterpret_cast<u32x4*>(o + pw_lch<NCF, PERM>(a, fq)) = q4;
                } else {
                    *reinterpret_cast<u32x2*>(o + pw_lch<NCF, PERM>(a, fq)) = qa[a];
                    *reinterpret_cast<u32x2*>(o + pw_lch<NCF, PERM>(a + 1, fq)) = qa[a + 1];
                }
            } else {
                *reinterpret_cast<u32x2*>(o + pw_lch<NCF, PERM>(a, fq)) = qa[a];
            }
        }
    };

    // ping-pong between two fragment sets (no register copies on the loop back edge)
    u32x4 xa[KS], xb[KS];
    int g = stream;
    if (g < ngroups) {
        load_x(xa, g);
        for (;;) {
            process(xa, xb, g);
            g += nstreams;
            if (g >= ngroups) break;
            process(xb, xa, g);
            g += nstreams;
            if (g >= ngroups) break;
        }
    }

    if constexpr (SUMS) {
        float* const rows_out = P.stats;
        const bool fuse = P.bn_acc != nullptr;           // accumulator rows instead of partial rows (y5m_bnfuse.h)
        if (rows_out || fuse) {
            // lanes -> wave (16 pixel lanes share a channel set), waves -> workgroup (through LDS, behind the
            // weights), one stats row per WORKGROUP: row `sblock`; rows sblock + k*gridDim-per-chunk are zero
            // padding so that the consumer (y5m_bn_finalize over stat_rows rows) needs no knowledge of the launch
#pragma unroll
            for (int j = 0; j < 4 * NCF; ++j) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { ssum[j] += __shfl_xor(ssum[j], o, 64); ssq[j] += __shfl_xor(ssq[j], o, 64); }
            }
            float* red = reinterpret_cast<float*>(smem + (size_t)NCF * KS * 64 * 16);      // [4 waves][2][NC]
            if (fr == 0) {
#pragma unroll
                for (int j = 0; j < 4 * NCF; ++j) {
                    const int c = pw_lch<NCF, PERM>(j >> 2, fq) + (j & 3);
                    red[(wid * 2 + 0) * NC + c] = ssum[j];
                    red[(wid * 2 + 1) * NC + c] = ssq[j];
                }
            }
            __syncthreads();
            const int nsb = nstreams >> 2;
            for (int t = tid; t < 2 * NC; t += PW_THREADS) {
                const int which = t / NC, c = t - which * NC;
                const float v = red[(0 * 2 + which) * NC + c] + red[(1 * 2 + which) * NC + c] +
                                red[(2 * 2 + which) * NC + c] + red[(3 * 2 + which) * NC + c];
                if (fuse) { bnf_add(P.bn_acc, P.Np, sblock, which, n0 + c, v); continue; }
                for (int row = sblock, first = 1; row < stat_rows; row += nsb, first = 0)
                    rows_out[((size_t)row * 2 + which) * P.Np + n0 + c] = first ? v : 0.f;
            }
        }
    }
}

static int g_pw = -1;          // Y5M_CONV_PW=0 routes every layer through the tiled kernel (A/B runs)
static int g_pw_occ = -1;      // Y5M_CONV_PW_OCC: workgroups per CU of the persistent grid

template <int NCF, int KS, int EPI, bool OLD, int TAPC = 0>
static int launch_pw(const ConvParams& P, hipStream_t st) {
    constexpr int NC = NCF * 16;
    if (g_pw_occ < 0) { const char* e = getenv("Y5M_CONV_PW_OCC"); g_pw_occ = e ? atoi(e) : 2; }      // (re-swept with the weight gradient forked after the data gradient: 2 beats 4 by ~0.15 ms/step)
    static int occ_fwd = -1;       // forward launches run alone on the GPU, backward ones next to the weight gradient
    if (occ_fwd < 0) { const char* e = getenv("Y5M_CONV_PW_OCC_FWD"); occ_fwd = e ? atoi(e) : g_pw_occ; }
    const int occ = EPI == EPI_DGRAD ? g_pw_occ : occ_fwd;
    const int nchunks = P.N / NC;
    const int ngroups = (P.M + 15) / 16;
    const int stat_rows = (P.M + CV_BM - 1) / CV_BM;                 // rows the caller sized the stats buffer for
    // stream blocks (4 streams each): every CU gets g_pw_occ workgroups, in multiples of 8 per chunk (XCD mapping)
    int sblocks = 256 * occ / nchunks / 8 * 8;
    if (sblocks < 8) sblocks = 8;
    const int need = (ngroups + 3) / 4;
    if (sblocks > (need + 7) / 8 * 8) sblocks = (need + 7) / 8 * 8;
    if (EPI == EPI_RAW_STATS && P.stats && !P.bn_acc && sblocks > stat_rows) sblocks = stat_rows / 8 * 8;
    if (sblocks < 8) return 0;                                       // tiny problem: leave it to the tiled kernel
    const int nstreams = sblocks * 4;
    const size_t lds = (size_t)NCF * KS * 64 * 16 + (EPI == EPI_RAW_STATS ? 4 * 2 * NC * sizeof(float) : 0);
    auto kern = conv_pw_kernel<NCF, KS, EPI, OLD, TAPC>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    Y5M_NAME_ONLY(1, "conv_pw_kernel<%d,%d,%d,%d,%d>", NCF, KS, EPI, (int)OLD, TAPC);
    hipLaunchKernelGGL(kern, dim3((unsigned)(sblocks * nchunks)), dim3(PW_THREADS), lds, st, P, nchunks, nstreams, ngroups, stat_rows);
    Y5M_CHECK_LAUNCH("conv_pw_kernel");
    return 1;
}

template <int NCF, int KS>
static int launch_pw_epi(const ConvParams& P, hipStream_t st) {
    if (P.epi == EPI_RAW_STATS) return launch_pw<NCF, KS, EPI_RAW_STATS, false>(P, st);
    if (P.epi == EPI_AFFINE_ACT)
        return P.res ? launch_pw<NCF, KS, EPI_AFFINE_ACT, true>(P, st) : launch_pw<NCF, KS, EPI_AFFINE_ACT, false>(P, st);
    return P.accumulate ? launch_pw<NCF, KS, EPI_DGRAD, true>(P, st) : launch_pw<NCF, KS, EPI_DGRAD, false>(P, st);
}

// Returns 1 when the launch was taken by the pointwise kernel, 0 when the layer does not qualify (the
// caller then uses the tiled kernel), < 0 on error.
static bool pw_eligible(const ConvParams& P, int dtype);

// 1 when y5m_conv would run this launch on the pointwise streaming kernel, 0 for the tiled implicit-GEMM kernel.
extern "C" int y5m_conv_is_pointwise(const y5m_conv_args* args, int dtype) {
    ConvParams P;
    memcpy(&P, args, sizeof(P));
    return pw_eligible(P, dtype) ? 1 : 0;
}

// Tapped streaming mode. (a) the stem: 3x3 / stride 1 / pad 1 over the 16-channel space-to-depth image, 48 output
// channels; (b) the 48 -> 48 3x3 layers of the 160x160 stage (K = 432 -> 14 K steps): forward / inference epilogue and
// the stride-1 data gradient (Y5M_CONV_PW_TAP48 bit mask; bit 2 = the stride-2 48 -> 96 forward, off: one 84 KB weight
// image per CU leaves too few waves). Dense output only.
static int tap_kind(const ConvParams& P, int dtype) {
    static int on = -1, on48 = -1;                   // Y5M_CONV_PW_STEM=0: leave the stem to the tiled kernel (A/B runs)
    if (on < 0) { const char* e = getenv("Y5M_CONV_PW_STEM"); on = (e && e[0] == '0') ? 0 : 1; }
    if (on48 < 0) { const char* e = getenv("Y5M_CONV_PW_TAP48"); on48 = e ? atoi(e) : 3; }   // measured: bit 0 -0.13, bit 1 -0.08, bit 2 +-0 ms/step
    if (dtype != Y5M_BF16 || P.th != 3 || P.tw != 3 || P.K != 9 * P.Cin || P.ldin % 8 != 0) return 0;
    const bool dense_out = P.osy == 1 && P.osx == 1 && P.ooy == 0 && P.oox == 0 && P.Hout == P.Hg && P.Wout == P.Wg;
    if (!dense_out || P.M % 16 != 0 || P.epi == EPI_HEAD) return 0;
    if (P.ldout % 8 != 0 || (reinterpret_cast<uintptr_t>(P.out) & 15) != 0) return 0;
    if (P.res && P.ldres % 4 != 0) return 0;
    if (P.epi == EPI_RAW_STATS && (!(P.stats || P.bn_acc) || (P.M + CV_BM - 1) / CV_BM < 8)) return 0;
    if ((size_t)P.B * P.Hin * P.Win * P.ldin * 2 >= (1ull << 31)) return 0;          // 32-bit pixel arithmetic in the loader
    if (on && P.Cin == 16 && P.N == 48 && P.Kp >= 160 && P.sy == 1 && P.sx == 1 && !P.res && !P.accumulate &&
        (P.epi == EPI_RAW_STATS || P.epi == EPI_AFFINE_ACT))
        return 16;
    if (on48 && P.Cin == 48 && (P.N == 48 || P.N == 96) && P.Kp >= 448 && P.sy == P.sx && (P.sy == 1 || P.sy == 2)) {
        // bit 0: 48 -> 48 forward / inference, bit 1: 48 -> 48 data gradient, bit 2: the stride-2 48 -> 96 forward
        const int bit = P.N == 96 ? 4 : (P.epi == EPI_DGRAD ? 2 : 1);
        if (on48 & bit) return 48;
    }
    return 0;
}

template <int NCF, int KS, int TAPC>
static int launch_tap_epi(const ConvParams& P, hipStream_t st) {
    if (P.epi == EPI_RAW_STATS) return launch_pw<NCF, KS, EPI_RAW_STATS, false, TAPC>(P, st);
    if (P.epi == EPI_AFFINE_ACT)
        return P.res ? launch_pw<NCF, KS, EPI_AFFINE_ACT, true, TAPC>(P, st) : launch_pw<NCF, KS, EPI_AFFINE_ACT, false, TAPC>(P, st);
    return (P.accumulate || P.res) ? launch_pw<NCF, KS, EPI_DGRAD, true, TAPC>(P, st) : launch_pw<NCF, KS, EPI_DGRAD, false, TAPC>(P, st);
}

int y5m_conv_pw_try(const ConvParams& P, int dtype, hipStream_t st) {
    switch (tap_kind(P, dtype)) {
    case 16: return launch_tap_epi<3, 5, 16>(P, st);
    case 48: return P.N == 48 ? launch_tap_epi<3, 14, 48>(P, st) : launch_tap_epi<6, 14, 48>(P, st);
    default: break;
    }
    if (!pw_eligible(P, dtype)) return 0;
    const int KS = (P.Cin + 31) / 32;
    // channel chunk per workgroup: 96 when it divides N, else 48
    if (P.N % 96 == 0) {
        if (KS == 2) return launch_pw_epi<6, 2>(P, st);
        if (KS == 3) return launch_pw_epi<6, 3>(P, st);
        if (KS == 6) return launch_pw_epi<6, 6>(P, st);
    } else {
        if (KS == 2) return launch_pw_epi<3, 2>(P, st);
        if (KS == 3) return launch_pw_epi<3, 3>(P, st);
        if (KS == 6) return launch_pw_epi<3, 6>(P, st);
    }
    return 0;
}

static bool pw_eligible(const ConvParams& P, int dtype) {
    if (g_pw < 0) { const char* e = getenv("Y5M_CONV_PW"); g_pw = (e && e[0] == '0') ? 0 : 1; }
    if (!g_pw || dtype != Y5M_BF16) return 0;
    const bool pointwise = P.th == 1 && P.tw == 1 && P.sy == 1 && P.sx == 1 && P.dh0 == 0 && P.dw0 == 0 &&
                           P.Hin == P.Hg && P.Win == P.Wg;
    const bool dense_out = P.epi != EPI_HEAD && P.osy == 1 && P.osx == 1 && P.ooy == 0 && P.oox == 0 &&
                           P.Hout == P.Hg && P.Wout == P.Wg;
    if (!pointwise || !dense_out) return 0;
    if (P.Cin % 16 != 0 || P.Cin > 192 || P.Cin <= 32 || P.N % 48 != 0 || P.M % 16 != 0) return 0;
    if (P.res && P.ldres % 4 != 0) return 0;
    if (P.ldout % 8 != 0 || (reinterpret_cast<uintptr_t>(P.out) & 15) != 0) return 0;      // 16-byte output pieces
    const int KS = (P.Cin + 31) / 32;
    if (KS * 32 > P.Kp) return 0;
    if (KS != 2 && KS != 3 && KS != 6) return 0;
    // (tiny problems stay on the tiled kernel: launch_pw needs >= 8 stream blocks)
    const int ngroups = (P.M + 15) / 16, stat_rows = (P.M + CV_BM - 1) / CV_BM;
    if ((ngroups + 3) / 4 < 1) return 0;
    if (P.epi == EPI_RAW_STATS && P.stats && stat_rows < 8) return 0;
    return 1;
}
