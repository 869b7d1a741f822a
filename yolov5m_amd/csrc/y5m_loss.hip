// ComputeLoss for gfx950: build_targets (bit-exact) + fused loss forward/backward.
// Restates reference ultralytics_loss.py:60-311. Compiled with -ffp-contract=off.
#include "y5m_box.h"

#define BT_T 1024

struct BtOut {
    int32_t* count[3];
    int32_t* bagg[3];
    float* tbox[3];
    float* anch[3];
    int32_t* tcls[3];
};

__device__ __forceinline__ float tmax(float a, float b) {   // torch.max: NaN propagates
    return (isnan(a) || isnan(b)) ? __int_as_float(0x7fc00000) : fmaxf(a, b);
}
__device__ __forceinline__ float py_mod1(float a) {          // torch `% 1` (remainder, sign of divisor)
    float r = fmodf(a, 1.0f);
    if (r != 0.0f && r < 0.0f) r += 1.0f;
    return r;
}

// ordered block compaction: returns the exclusive rank of this thread among flagged threads and the
// block total; contains two barriers.
__device__ __forceinline__ int block_rank(bool flag, int* wave_tot, int& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    if (lane == 0) wave_tot[wid] = __popcll(bal);
    __syncthreads();
    int wbase = 0;
    total = 0;
#pragma unroll
    for (int q = 0; q < BT_T / 64; ++q) {
        const int c = wave_tot[q];
        if (q < wid) wbase += c;
        total += c;
    }
    __syncthreads();
    return wbase + __popcll(bal & ((1ull << lane) - 1ull));
}

// One workgroup per scale. Row order: offset-major [0,0],[.5,0],[0,.5],[-.5,0],[0,-.5];
// inside an offset block: anchor-major, then original target order.
__global__ __launch_bounds__(BT_T) void build_targets_kernel(
    const float* __restrict__ targets, int nt_static, const int32_t* __restrict__ d_nt, int nt_max,
    const float* __restrict__ anchors, int naxs, int ny0, int ny1, int ny2, int nx0, int nx1, int nx2,
    float anchor_t, BtOut out, int32_t* __restrict__ frow_ws) {
    __shared__ int wave_tot[BT_T / 64];
    const int s = blockIdx.x;
    const int ny = s == 0 ? ny0 : (s == 1 ? ny1 : ny2);
    const int nx = s == 0 ? nx0 : (s == 1 ? nx1 : nx2);
    int nt = d_nt ? *d_nt : nt_static;
    if (nt > nt_max) nt = nt_max;
    const float* anc = anchors + s * naxs * 2;
    int32_t* frow = frow_ws + (size_t)s * naxs * nt_max;
    const float fnx = (float)nx, fny = (float)ny;
    const float g = 0.5f;
    const int tid = threadIdx.x;

    // stage 1: anchor-ratio filter (:186-213), ordered compaction -> frow
    int nf = 0;
    const int nrow = naxs * nt;
    for (int r0 = 0; r0 < nrow; r0 += BT_T) {
        const int r = r0 + tid;
        bool pass = false;
        if (r < nrow) {
            const int a = r / nt, i = r - a * nt;
            const float gw = targets[6 * i + 4] * fnx, gh = targets[6 * i + 5] * fny;   // t = targets*gain
            const float rw = gw / anc[2 * a], rh = gh / anc[2 * a + 1];                 // :190
            const float m = tmax(tmax(rw, 1.0f / rw), tmax(rh, 1.0f / rh));            // :199
            pass = m < anchor_t;
        }
        int total;
        const int rank = block_rank(pass, wave_tot, total);
        if (pass) frow[nf + rank] = r;
        nf += total;
    }
    __syncthreads();   // frow visible block-wide (global writes drained by the barrier)

    // stage 2: neighbour offsets (:222-255) and outputs
    int n = 0;
    for (int o = 0; o < 5; ++o) {
        for (int q0 = 0; q0 < nf; q0 += BT_T) {
            const int q = q0 + tid;
            bool take = false;
            int a = 0, i = 0;
            float gx = 0.f, gy = 0.f;
            if (q < nf) {
                const int r = frow[q];
                a = r / nt; i = r - a * nt;
                gx = targets[6 * i + 2] * fnx;
                gy = targets[6 * i + 3] * fny;
                const float gxi = fnx - gx, gyi = fny - gy;                              // :219
                switch (o) {
                    case 0: take = true; break;
                    case 1: take = (py_mod1(gx) < g) && (gx > 1.0f); break;              // j  :225
                    case 2: take = (py_mod1(gy) < g) && (gy > 1.0f); break;              // k
                    case 3: take = (py_mod1(gxi) < g) && (gxi > 1.0f); break;            // l  :235
                    default: take = (py_mod1(gyi) < g) && (gyi > 1.0f); break;           // m
                }
            }
            int total;
            const int rank = block_rank(take, wave_tot, total);
            if (take) {
                const int pos = n + rank;
                const float ox = (o == 1 ? 1.0f : (o == 3 ? -1.0f : 0.0f)) * g;
                const float oy = (o == 2 ? 1.0f : (o == 4 ? -1.0f : 0.0f)) * g;
                int gi = (int)(gx - ox);                                                 // .long() :278
                int gj = (int)(gy - oy);
                gj = gj < 0 ? 0 : (gj > ny - 1 ? ny - 1 : gj);                           // clamp_ :285
                gi = gi < 0 ? 0 : (gi > nx - 1 ? nx - 1 : gi);
                reinterpret_cast<int4*>(out.bagg[s])[pos] =
                    make_int4((int)targets[6 * i + 0], a, gj, gi);
                reinterpret_cast<float4*>(out.tbox[s])[pos] =
                    make_float4(gx - (float)gi, gy - (float)gj, targets[6 * i + 4] * fnx, targets[6 * i + 5] * fny);
                reinterpret_cast<float2*>(out.anch[s])[pos] = make_float2(anc[2 * a], anc[2 * a + 1]);
                out.tcls[s][pos] = (int)targets[6 * i + 1];
            }
            n += total;
        }
    }
    if (tid == 0) *out.count[s] = n;
}

extern "C" size_t y5m_build_targets_workspace_bytes(int nt_max, int naxs) {
    return y5m_align((size_t)3 * (size_t)naxs * (size_t)(nt_max > 0 ? nt_max : 1) * sizeof(int32_t)) + 256;
}

extern "C" int y5m_build_targets(const float* targets, int nt, const int32_t* d_nt, int nt_max,
                                 const float* anchors, int naxs, const int* ny, const int* nx, float anchor_t,
                                 y5m_targets out[3], void* ws, size_t ws_bytes, void* stream) {
    Y5M_REQUIRE(naxs >= 1 && nt_max >= 0 && nt >= 0, "bad dims");
    Y5M_REQUIRE(nt <= nt_max || d_nt, "nt exceeds nt_max");
    if (ws_bytes < y5m_build_targets_workspace_bytes(nt_max, naxs)) { y5m_set_error("build_targets ws too small"); return Y5M_EWS; }
    BtOut o;
    for (int s = 0; s < 3; ++s) {
        o.count[s] = out[s].count; o.bagg[s] = out[s].bagg; o.tbox[s] = out[s].tbox;
        o.anch[s] = out[s].anch; o.tcls[s] = out[s].tcls;
    }
    hipLaunchKernelGGL(build_targets_kernel, dim3(3), dim3(BT_T), 0, y5m_stream(stream), targets, nt, d_nt,
                       nt_max, anchors, naxs, ny[0], ny[1], ny[2], nx[0], nx[1], nx[2], anchor_t, o,
                       reinterpret_cast<int32_t*>(ws));
    Y5M_CHECK_LAUNCH("build_targets_kernel");
    return Y5M_OK;
}

// =================================================================================================
// loss forward + backward (reference ultralytics_loss.py:60-120)
// =================================================================================================
struct LossScale {
    const float* p;        // logits (B,naxs,ny,nx,nch)
    float* grad;           // same shape or NULL
    const int32_t* count;
    const int32_t* bagg;
    const float* tbox;
    const float* anch;
    const int32_t* tcls;
    int32_t* owner;        // [cells] last row index that targets the cell, -1 = none
    float* rowiou;         // [cap] clamp(giou,0)
    float* rowlbox;        // [cap] 1 - giou
    float* rowcls;         // [cap] sum_c bce
    float* objpart;        // [nblk] per-block partial sums of the objectness BCE
    float* gobj;           // [cells] d loss / d objectness logit (compact copy of channel 4 of grad)
    int ny, nx, cap, nblk;
    int64_t cells;
    float balance;
    const float* dense;    // YOLO_LOSS mode: dense targets (B,naxs,ny,nx,6); objectness target = channel 4
};
struct LossArgs {
    LossScale s[3];
    int B, naxs, nc, nch;
    float lam_box, lam_obj, lam_cls;
    int dense_mode;        // 1: reference loss.py semantics (ignore cells keep target -1, empty mean = NaN)
    int sparse_grad;       // 1: grad rows are written ONLY for owned cells (the rest of grad is left untouched): the
                           //    consumer reads gobj + owner + those rows (y5m_head_grad_pack_sparse)
};

__device__ __forceinline__ float bce_logits(float x, float t) {   // BCEWithLogits, pos_weight = 1
    return fmaxf(x, 0.0f) - x * t + log1pf(expf(-fabsf(x)));
}

// one wave per target row. BWD=false: forward partials + tobj ownership; BWD=true: gradients.
template <bool BWD>
__global__ __launch_bounds__(256) void loss_rows_kernel(LossArgs A) {
    const LossScale& S = A.s[blockIdx.y];
    if (!S.p) return;                                  // absent scale (single-scale YOLO_LOSS.compute_loss)
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = *S.count;
    if (j >= n) return;
    const int4 q = reinterpret_cast<const int4*>(S.bagg)[j];
    const int64_t cell = (((int64_t)q.x * A.naxs + q.y) * S.ny + q.z) * S.nx + q.w;
    const float* row = S.p + cell * A.nch;
    const float v0 = lane < A.nch ? row[lane] : 0.0f;
    const float v1 = lane + 64 < A.nch ? row[lane + 64] : 0.0f;
    const float l0 = __shfl(v0, 0, 64), l1 = __shfl(v0, 1, 64), l2 = __shfl(v0, 2, 64), l3 = __shfl(v0, 3, 64);
    const float2 an = reinterpret_cast<const float2*>(S.anch)[j];
    const float4 tb = reinterpret_cast<const float4*>(S.tbox)[j];
    const float s0 = sigmoidf_(l0), s1 = sigmoidf_(l1), s2 = sigmoidf_(l2), s3 = sigmoidf_(l3);
    float pb[4], tbv[4] = {tb.x, tb.y, tb.z, tb.w};
    pb[0] = s0 * 2.0f - 0.5f;                         // :81
    pb[1] = s1 * 2.0f - 0.5f;
    const float e2 = s2 * 2.0f, e3 = s3 * 2.0f;
    pb[2] = (e2 * e2) * an.x;                         // :82
    pb[3] = (e3 * e3) * an.y;
    const BoxFwd r = box_iou_fwd(pb, tbv, true, 1e-7f);
    const int tc = S.tcls[j];
    if (!BWD) {
        float c = 0.0f;
        if (lane >= 5 && lane < A.nch) c += bce_logits(v0, (lane - 5) == tc ? 1.0f : 0.0f);
        if (lane + 64 < A.nch) c += bce_logits(v1, (lane + 59) == tc ? 1.0f : 0.0f);
        c = wave_sum(c);
        if (lane == 0) {
            S.rowlbox[j] = 1.0f - r.out;              // :85
            S.rowiou[j] = r.out < 0.0f ? 0.0f : r.out;   // :88 detach().clamp(0)
            S.rowcls[j] = c;
            atomicMax(&S.owner[cell], j);             // :89 last row wins (CPU index_put order)
        }
    } else {
        float* grow = S.grad + cell * A.nch;
        const float bs = (float)A.B;
        const float gg = -(A.lam_box * bs) / (float)n;          // d loss / d giou_j
        const float gc = (A.lam_cls * bs) / ((float)n * (float)A.nc);
        if (lane == 0) {
            float ga[4], gb[4];
            box_iou_bwd(r, true, gg, ga, gb);
            atomicAdd(&grow[0], ga[0] * 2.0f * s0 * (1.0f - s0));
            atomicAdd(&grow[1], ga[1] * 2.0f * s1 * (1.0f - s1));
            atomicAdd(&grow[2], ga[2] * an.x * 8.0f * s2 * s2 * (1.0f - s2));
            atomicAdd(&grow[3], ga[3] * an.y * 8.0f * s3 * s3 * (1.0f - s3));
        }
        if (A.nc > 1) {
            if (lane >= 5 && lane < A.nch)
                atomicAdd(&grow[lane], (sigmoidf_(v0) - ((lane - 5) == tc ? 1.0f : 0.0f)) * gc);
            if (lane + 64 < A.nch)
                atomicAdd(&grow[lane + 64], (sigmoidf_(v1) - ((lane + 59) == tc ? 1.0f : 0.0f)) * gc);
        }
    }
}

// objectness BCE over every cell of a scale + (optional) full overwrite of the gradient tensor:
// zeros everywhere, d/d(obj logit) in channel 4. One wave owns 64 consecutive cells so the gradient
// rows are written as one contiguous, coalesced run.
template <bool GRAD>
__global__ __launch_bounds__(256) void loss_obj_kernel(LossArgs A) {
    __shared__ float red[4];
    __shared__ float gsm[4][64];
    const LossScale& S = A.s[blockIdx.y];
    if (!S.p || (int)blockIdx.x >= S.nblk) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t c0 = ((int64_t)blockIdx.x * 4 + wid) * 64;
    const int64_t cell = c0 + lane;
    float bce = 0.0f, gobj = 0.0f;
    if (cell < S.cells) {
        const float x = S.p[cell * A.nch + 4];
        const int ow = S.owner[cell];
        float t = ow >= 0 ? S.rowiou[ow] : 0.0f;
        if (S.dense) {                                   // loss.py:217-220: obj cells *= giou, -1 (ignore) stays -1
            const float d = S.dense[cell * 6 + 4];
            t = d == 1.0f ? t : d;
        }
        bce = bce_logits(x, t);                                          // :101
        gobj = (sigmoidf_(x) - t) * (S.balance * A.lam_obj * (float)A.B / (float)S.cells);
    }
    if (GRAD) {
        if (cell < S.cells) S.gobj[cell] = gobj;
    }
    if (GRAD && A.sparse_grad) {
        // rows of owned cells only: zeros + the objectness gradient, the target rows then accumulate onto them
        const int ow = cell < S.cells ? S.owner[cell] : -1;
        unsigned long long owned = __ballot(ow >= 0);
        while (owned) {
            const int cl = __ffsll((long long)owned) - 1;
            owned &= owned - 1ull;
            const float go = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gobj), cl));
            float* dst = S.grad + (c0 + cl) * A.nch;
            if (lane < A.nch) dst[lane] = lane == 4 ? go : 0.0f;
            if (lane + 64 < A.nch) dst[lane + 64] = 0.0f;
        }
    } else if (GRAD) {
        gsm[wid][lane] = gobj;                       // same-wave LDS traffic is in order: no workgroup barrier
        __builtin_amdgcn_wave_barrier();             // (no instruction: pins the order for the compiler -- and is the point where
                                                     //  the CPU lane-level executor of tests/emu lets the other lanes catch up)
        if (c0 < S.cells) {
            const int ncell = (int)((S.cells - c0) < 64 ? (S.cells - c0) : 64);
            const int nelem = ncell * A.nch;
            float* dst = S.grad + c0 * A.nch;
            const volatile float* gl = gsm[wid];
            if ((((uintptr_t)dst) & 15) == 0) {
                const int nvec = nelem >> 2;
                for (int v = lane; v < nvec; v += 64) {
                    const int e = 4 * v;
                    const int cl = e / A.nch, ch = e - cl * A.nch;
                    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int k = 4 - ch;             // position of channel 4 inside this vector
                    if (k >= 0 && k < 4) {
                        const float go = gl[cl];
                        if (k == 0) w.x = go; else if (k == 1) w.y = go; else if (k == 2) w.z = go; else w.w = go;
                    }
                    reinterpret_cast<float4*>(dst)[v] = w;
                }
                for (int e = (nvec << 2) + lane; e < nelem; e += 64) {
                    const int cl = e / A.nch, ch = e - cl * A.nch;
                    dst[e] = ch == 4 ? gl[cl] : 0.0f;
                }
            } else {
                for (int e = lane; e < nelem; e += 64) {
                    const int cl = e / A.nch, ch = e - cl * A.nch;
                    dst[e] = ch == 4 ? gl[cl] : 0.0f;
                }
            }
        }
    }
    bce = wave_sum(bce);
    if (lane == 0) red[wid] = bce;
    __syncthreads();
    if (threadIdx.x == 0) S.objpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ float block_sum_f(const float* __restrict__ v, int n, float* sm) {
    float acc = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += v[i];
    acc = wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    float t = 0.0f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i];
    return t;
}

// deterministic final reduction (fixed tree), single workgroup
__global__ __launch_bounds__(1024) void loss_finalize_kernel(LossArgs A, float* __restrict__ loss_out) {
    __shared__ float sm[16];
    float lbox = 0.0f, lobj = 0.0f, lcls = 0.0f;
    for (int s = 0; s < 3; ++s) {
        const LossScale& S = A.s[s];
        if (!S.p) continue;
        const int n = *S.count;
        if (n > 0) {
            lbox += block_sum_f(S.rowlbox, n, sm) / (float)n;                    // :85 mean
            if (A.nc > 1) lcls += block_sum_f(S.rowcls, n, sm) / ((float)n * (float)A.nc);   // :95
        } else if (A.dense_mode) {                       // loss.py:212,231: mean over an empty selection is NaN
            lbox += __int_as_float(0x7fc00000);
            lcls += __int_as_float(0x7fc00000);
        }
        lobj += (block_sum_f(S.objpart, S.nblk, sm) / (float)S.cells) * S.balance;            // :101-102
    }
    if (threadIdx.x == 0) {
        lbox *= A.lam_box; lobj *= A.lam_obj; lcls *= A.lam_cls;                 // :104-106
        loss_out[0] = (lbox + lobj + lcls) * (float)A.B;                         // :120
        loss_out[1] = lbox; loss_out[2] = lobj; loss_out[3] = lcls;
    }
}

static size_t loss_ws_layout(int B, int naxs, const int* ny, const int* nx, int nt_max, size_t off[3][6], int nblk[3]) {
    size_t cur = 0;
    const size_t cap = (size_t)5 * naxs * (nt_max > 0 ? nt_max : 1);
    for (int s = 0; s < 3; ++s) {
        const size_t cells = (size_t)B * naxs * ny[s] * nx[s];
        nblk[s] = (int)((cells + 255) / 256);
        off[s][0] = cur; cur += y5m_align(cells * 4);
        off[s][1] = cur; cur += y5m_align(cap * 4);
        off[s][2] = cur; cur += y5m_align(cap * 4);
        off[s][3] = cur; cur += y5m_align(cap * 4);
        off[s][4] = cur; cur += y5m_align((size_t)(nblk[s] > 0 ? nblk[s] : 1) * 4);
        off[s][5] = cur; cur += y5m_align(cells * 4);
    }
    return cur + 256;
}

extern "C" size_t y5m_compute_loss_workspace_bytes(int B, int naxs, const int* ny, const int* nx, int nt_max) {
    size_t off[3][6]; int nblk[3];
    return loss_ws_layout(B, naxs, ny, nx, nt_max, off, nblk);
}

extern "C" int y5m_compute_loss_owner_ptrs(void* ws, int B, int naxs, const int* ny, const int* nx, int nt_max,
                                           int32_t* owner_out[3], float* gobj_out[3]) {
    Y5M_REQUIRE(ws && owner_out && gobj_out, "null");
    size_t off[3][6]; int nblk[3];
    loss_ws_layout(B, naxs, ny, nx, nt_max, off, nblk);
    for (int s = 0; s < 3; ++s) {
        owner_out[s] = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + off[s][0]);
        gobj_out[s] = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + off[s][5]);
    }
    return Y5M_OK;
}

static int run_loss(const float* const p[3], float* const grad[3], const float* const dense[3], int B, int naxs,
                    const int* ny, const int* nx, int nc, const y5m_targets tg[3], int nt_max, const float balance[3],
                    float lambda_box, float lambda_obj, float lambda_cls, float* loss_out, void* ws, size_t ws_bytes,
                    hipStream_t st, int sparse_grad = 0) {
    Y5M_REQUIRE(B >= 1 && naxs >= 1 && nc >= 1, "bad dims");
    Y5M_REQUIRE(5 + nc <= 128, "5+nc must be <= 128 (one wave holds a logit row in two registers)");
    size_t off[3][6]; int nblk[3];
    const size_t need = loss_ws_layout(B, naxs, ny, nx, nt_max, off, nblk);
    if (ws_bytes < need) { y5m_set_error("compute_loss ws too small"); return Y5M_EWS; }
    char* w = reinterpret_cast<char*>(ws);
    LossArgs A;
    A.B = B; A.naxs = naxs; A.nc = nc; A.nch = 5 + nc;
    A.lam_box = lambda_box; A.lam_obj = lambda_obj; A.lam_cls = lambda_cls;
    A.dense_mode = dense ? 1 : 0;
    A.sparse_grad = sparse_grad;
    bool want_grad = grad != nullptr;
    for (int s = 0; s < 3; ++s) if (p[s] && !(grad && grad[s])) want_grad = false;
    int max_blk = 0;
    const int cap = 5 * naxs * (nt_max > 0 ? nt_max : 1);
    for (int s = 0; s < 3; ++s) {
        LossScale& S = A.s[s];
        S.p = p[s]; S.grad = want_grad ? grad[s] : nullptr;
        S.count = tg[s].count; S.bagg = tg[s].bagg; S.tbox = tg[s].tbox; S.anch = tg[s].anch; S.tcls = tg[s].tcls;
        S.owner = reinterpret_cast<int32_t*>(w + off[s][0]);
        S.rowiou = reinterpret_cast<float*>(w + off[s][1]);
        S.rowlbox = reinterpret_cast<float*>(w + off[s][2]);
        S.rowcls = reinterpret_cast<float*>(w + off[s][3]);
        S.objpart = reinterpret_cast<float*>(w + off[s][4]);
        S.gobj = reinterpret_cast<float*>(w + off[s][5]);
        S.ny = ny[s]; S.nx = nx[s]; S.cap = cap; S.nblk = nblk[s];
        S.cells = (int64_t)B * naxs * ny[s] * nx[s];
        S.balance = balance[s];
        S.dense = dense ? dense[s] : nullptr;
        if (!S.p) { S.nblk = 0; S.cells = 0; continue; }
        max_blk = nblk[s] > max_blk ? nblk[s] : max_blk;
        if (y5m_fill32(S.owner, 0xFFFFFFFFu, (size_t)S.cells, st) != Y5M_OK) return Y5M_ELAUNCH;
    }
    const dim3 rgrid((unsigned)((cap + 3) / 4), 3);
    if (nt_max > 0) {
        hipLaunchKernelGGL(loss_rows_kernel<false>, rgrid, dim3(256), 0, st, A);
        Y5M_CHECK_LAUNCH("loss_rows_kernel<fwd>");
    }
    if (want_grad) hipLaunchKernelGGL(loss_obj_kernel<true>, dim3((unsigned)max_blk, 3), dim3(256), 0, st, A);
    else hipLaunchKernelGGL(loss_obj_kernel<false>, dim3((unsigned)max_blk, 3), dim3(256), 0, st, A);
    Y5M_CHECK_LAUNCH("loss_obj_kernel");
    if (want_grad && nt_max > 0) {
        hipLaunchKernelGGL(loss_rows_kernel<true>, rgrid, dim3(256), 0, st, A);
        Y5M_CHECK_LAUNCH("loss_rows_kernel<bwd>");
    }
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, st, A, loss_out);
    Y5M_CHECK_LAUNCH("loss_finalize_kernel");
    return Y5M_OK;
}

extern "C" int y5m_compute_loss(const float* const p[3], float* const grad[3], int B, int naxs, const int* ny,
                                const int* nx, int nc, const y5m_targets tg[3], int nt_max,
                                const float balance[3], float lambda_box, float lambda_obj, float lambda_cls,
                                float* loss_out, void* ws, size_t ws_bytes, void* stream) {
    return run_loss(p, grad, nullptr, B, naxs, ny, nx, nc, tg, nt_max, balance, lambda_box, lambda_obj, lambda_cls,
                    loss_out, ws, ws_bytes, y5m_stream(stream));
}

// Same loss; of grad only the rows of the cells a target row hit are written (zeros + objectness + box / class terms),
// the objectness gradient of every cell goes to the workspace's compact plane (y5m_compute_loss_owner_ptrs).
extern "C" int y5m_compute_loss_sparse(const float* const p[3], float* const grad[3], int B, int naxs, const int* ny,
                                       const int* nx, int nc, const y5m_targets tg[3], int nt_max,
                                       const float balance[3], float lambda_box, float lambda_obj, float lambda_cls,
                                       float* loss_out, void* ws, size_t ws_bytes, void* stream) {
    Y5M_REQUIRE(grad != nullptr, "y5m_compute_loss_sparse is the gradient-producing variant");
    return run_loss(p, grad, nullptr, B, naxs, ny, nx, nc, tg, nt_max, balance, lambda_box, lambda_obj, lambda_cls,
                    loss_out, ws, ws_bytes, y5m_stream(stream), 1);
}

// =================================================================================================
// YOLO_LOSS.compute_loss (reference loss.py:195-246) on dense targets (B,naxs,ny,nx,6)
// =================================================================================================
// positives (targets[...,4] == 1) -> row table in row-major cell order (= boolean-mask order of :198-204)
__global__ __launch_bounds__(BT_T) void dense_to_rows_kernel(const float* d0, const float* d1, const float* d2,
                                                            const float* __restrict__ anchors, int B, int naxs,
                                                            int ny0, int ny1, int ny2, int nx0, int nx1, int nx2,
                                                            int cap, BtOut out) {
    __shared__ int wave_tot[BT_T / 64];
    const int s = blockIdx.x;
    const float* d = s == 0 ? d0 : (s == 1 ? d1 : d2);
    const int ny = s == 0 ? ny0 : (s == 1 ? ny1 : ny2);
    const int nx = s == 0 ? nx0 : (s == 1 ? nx1 : nx2);
    const int64_t cells = (int64_t)B * naxs * ny * nx;
    if (!d) { if (threadIdx.x == 0) *out.count[s] = 0; return; }
    int n = 0;
    for (int64_t c0 = 0; c0 < cells; c0 += BT_T) {
        const int64_t c = c0 + threadIdx.x;
        const bool pos = c < cells && d[c * 6 + 4] == 1.0f;
        int total;
        const int rank = block_rank(pos, wave_tot, total);
        if (pos && n + rank < cap) {
            const int row = n + rank;
            const int gi = (int)(c % nx);
            int64_t t = c / nx;
            const int gj = (int)(t % ny);
            t /= ny;
            const int a = (int)(t % naxs);
            const int b = (int)(t / naxs);
            reinterpret_cast<int4*>(out.bagg[s])[row] = make_int4(b, a, gj, gi);
            reinterpret_cast<float4*>(out.tbox[s])[row] = make_float4(d[c * 6 + 0], d[c * 6 + 1], d[c * 6 + 2], d[c * 6 + 3]);
            reinterpret_cast<float2*>(out.anch[s])[row] = make_float2(anchors[(s * naxs + a) * 2], anchors[(s * naxs + a) * 2 + 1]);
            out.tcls[s][row] = (int)d[c * 6 + 5];
        }
        n += total;
    }
    if (threadIdx.x == 0) *out.count[s] = n < cap ? n : cap;
}

static size_t dense_tables_bytes(int rows) {
    const size_t r = (size_t)(rows > 0 ? rows : 1);
    return 3 * (y5m_align(4) + y5m_align(r * 16) + y5m_align(r * 16) + y5m_align(r * 8) + y5m_align(r * 4));
}
static int dense_nt(int rows_max, int naxs) { return (rows_max + 5 * naxs - 1) / (5 * naxs) + 1; }

extern "C" size_t y5m_compute_loss_dense_workspace_bytes(int B, int naxs, const int* ny, const int* nx, int rows_max) {
    size_t off[3][6]; int nblk[3];
    return dense_tables_bytes(5 * naxs * dense_nt(rows_max, naxs)) + loss_ws_layout(B, naxs, ny, nx, dense_nt(rows_max, naxs), off, nblk) + 256;
}

// the row tables of the dense loss at the front of its workspace (count / bagg / tbox / anch / tcls per scale); returns the rest
static char* dense_tables(void* ws, int cap, y5m_targets tg[3]) {
    char* w = reinterpret_cast<char*>(ws);
    for (int s = 0; s < 3; ++s) {
        tg[s].count = reinterpret_cast<int32_t*>(w); w += y5m_align(4);
        tg[s].bagg = reinterpret_cast<int32_t*>(w); w += y5m_align((size_t)cap * 16);
        tg[s].tbox = reinterpret_cast<float*>(w); w += y5m_align((size_t)cap * 16);
        tg[s].anch = reinterpret_cast<float*>(w); w += y5m_align((size_t)cap * 8);
        tg[s].tcls = reinterpret_cast<int32_t*>(w); w += y5m_align((size_t)cap * 4);
    }
    return w;
}

static int run_loss_dense(const float* const p[3], float* const grad[3], const float* const dense[3], int B,
                          int naxs, const int* ny, const int* nx, int nc, const float* anchors, int rows_max,
                          const float balance[3], float lambda_box, float lambda_obj, float lambda_cls,
                          float* loss_out, void* ws, size_t ws_bytes, void* stream, int sparse_grad) {
    if (ws_bytes < y5m_compute_loss_dense_workspace_bytes(B, naxs, ny, nx, rows_max)) { y5m_set_error("loss_dense ws too small"); return Y5M_EWS; }
    hipStream_t st = y5m_stream(stream);
    const int nt = dense_nt(rows_max, naxs);
    const int cap = 5 * naxs * nt;
    y5m_targets tg[3];
    BtOut o;
    char* w = dense_tables(ws, cap, tg);
    for (int s = 0; s < 3; ++s) {
        o.count[s] = tg[s].count; o.bagg[s] = tg[s].bagg; o.tbox[s] = tg[s].tbox; o.anch[s] = tg[s].anch; o.tcls[s] = tg[s].tcls;
    }
    hipLaunchKernelGGL(dense_to_rows_kernel, dim3(3), dim3(BT_T), 0, st, dense[0], dense[1], dense[2], anchors, B, naxs,
                       ny[0], ny[1], ny[2], nx[0], nx[1], nx[2], cap, o);
    Y5M_CHECK_LAUNCH("dense_to_rows_kernel");
    const size_t used = (size_t)(w - reinterpret_cast<char*>(ws));
    return run_loss(p, grad, dense, B, naxs, ny, nx, nc, tg, nt, balance, lambda_box, lambda_obj, lambda_cls, loss_out, w,
                    ws_bytes - used, st, sparse_grad);
}

extern "C" int y5m_compute_loss_dense(const float* const p[3], float* const grad[3], const float* const dense[3], int B,
                                      int naxs, const int* ny, const int* nx, int nc, const float* anchors, int rows_max,
                                      const float balance[3], float lambda_box, float lambda_obj, float lambda_cls,
                                      float* loss_out, void* ws, size_t ws_bytes, void* stream) {
    return run_loss_dense(p, grad, dense, B, naxs, ny, nx, nc, anchors, rows_max, balance, lambda_box, lambda_obj, lambda_cls,
                          loss_out, ws, ws_bytes, stream, 0);
}

// The dense-target loss with the SPARSE gradient of y5m_compute_loss_sparse: of grad only the rows of the positive cells are
// written, the objectness gradient of every cell (ignore cells included: their BCE target stays -1) goes to the compact plane.
extern "C" int y5m_compute_loss_dense_sparse(const float* const p[3], float* const grad[3], const float* const dense[3], int B,
                                             int naxs, const int* ny, const int* nx, int nc, const float* anchors, int rows_max,
                                             const float balance[3], float lambda_box, float lambda_obj, float lambda_cls,
                                             float* loss_out, void* ws, size_t ws_bytes, void* stream) {
    Y5M_REQUIRE(grad != nullptr, "y5m_compute_loss_dense_sparse is the gradient-producing variant");
    return run_loss_dense(p, grad, dense, B, naxs, ny, nx, nc, anchors, rows_max, balance, lambda_box, lambda_obj, lambda_cls,
                          loss_out, ws, ws_bytes, stream, 1);
}

// what y5m_head_grad_pack_sparse needs of a y5m_compute_loss_dense_sparse workspace: per scale the owner / objectness-gradient
// planes and the positive-row table (bagg, count); *cap_out = the tables' capacity
extern "C" int y5m_compute_loss_dense_owner_ptrs(void* ws, int B, int naxs, const int* ny, const int* nx, int rows_max,
                                                 int32_t* owner_out[3], float* gobj_out[3], int32_t* bagg_out[3],
                                                 int32_t* count_out[3], int* cap_out) {
    Y5M_REQUIRE(ws && owner_out && gobj_out && bagg_out && count_out && cap_out, "null");
    const int nt = dense_nt(rows_max, naxs);
    const int cap = 5 * naxs * nt;
    y5m_targets tg[3];
    char* w = dense_tables(ws, cap, tg);
    size_t off[3][6]; int nblk[3];
    loss_ws_layout(B, naxs, ny, nx, nt, off, nblk);
    for (int s = 0; s < 3; ++s) {
        owner_out[s] = reinterpret_cast<int32_t*>(w + off[s][0]);
        gobj_out[s] = reinterpret_cast<float*>(w + off[s][5]);
        bagg_out[s] = tg[s].bagg;
        count_out[s] = tg[s].count;
    }
    *cap_out = cap;
    return Y5M_OK;
}

// =================================================================================================
// YOLO_LOSS.build_targets (reference loss.py:101-192) + iou_width_height (utils/bboxes_utils.py:6-29)
// =================================================================================================
// One workgroup per image. The assignment is sequential by construction in the reference -- boxes claim
// (scale, anchor, cell) slots first come first served, and EVERY box divides the loss object's anchors by 640 in
// place before it is matched (bboxes_utils.py:18) -- so lane 0 walks the image's boxes in order; the
// other lanes only clear the image's slices of the dense targets. Image b starts from the anchor state after
// off[b] boxes: the decay is replayed (at most ~17 divisions until fp32 reaches exact zeros, where it stops).
// Arithmetic as torch evaluates the reference's expressions: anchors fp32, true division by 640; the box's w, h enter
// the min() as fp32 (a 0-dim float64 tensor does not promote a dimensioned float32 one), w*h is a float64 product
// rounded to fp32 when added; cell indices and offsets are float64 (numpy scalars) and rounded once on the store;
// argsort(descending) of 9 values = libstdc++ insertion sort = stable, NaN first. Compiled with -ffp-contract=off.
struct YbtArgs {
    float* dense[3];
    int ny[3], nx[3], stride[3];
};

__device__ __forceinline__ bool ybt_before(float x, float y) {      // ATen KeyValueCompDesc
    return (isnan(x) && !isnan(y)) || x > y;
}

__global__ __launch_bounds__(256) void yolo_build_targets_kernel(const double* __restrict__ boxes, const int* __restrict__ off,
                                                                 int B, YbtArgs A, const float* __restrict__ anc_in,
                                                                 float* __restrict__ anc_out, float ignore_thr) {
    const int b = blockIdx.x;
    for (int s = 0; s < 3; ++s) {
        const int64_t n = (int64_t)3 * A.ny[s] * A.nx[s] * 6;
        float* d = A.dense[s] + (int64_t)b * n;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) d[i] = 0.f;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float a[18];
    for (int i = 0; i < 18; ++i) a[i] = anc_in[i];
    const int k0 = off[b], k1 = off[b + 1];
    for (int k = 0; k < k0; ++k) {
        bool any = false;
        for (int i = 0; i < 18; ++i) { a[i] = __fdiv_rn(a[i], 640.0f); any |= a[i] != 0.f; }
        if (!any) break;                                    // exact zeros (or NaN-free fixed point): further steps change nothing
    }
    for (int k = k0; k < k1; ++k) {
        for (int i = 0; i < 18; ++i) a[i] = __fdiv_rn(a[i], 640.0f);               // bboxes_utils.py:18
        const double cls = boxes[k * 5 + 0], x = boxes[k * 5 + 1], y = boxes[k * 5 + 2], w = boxes[k * 5 + 3], h = boxes[k * 5 + 4];
        const float w32 = (float)w, h32 = (float)h, wh32 = (float)(w * h);
        float iou[9];
        int order[9];
        for (int j = 0; j < 9; ++j) {
            const float st = (float)A.stride[j / 3];
            const float aw = a[2 * j] * st, ah = a[2 * j + 1] * st;                // :20-23
            const float mw = (w32 != w32 || aw != aw) ? NAN : (w32 < aw ? w32 : aw);
            const float mh = (h32 != h32 || ah != ah) ? NAN : (h32 < ah ? h32 : ah);
            const float inter = mw * mh;                                           // :25-27
            const float uni = (wh32 + aw * ah) - inter;                            // :28-30
            iou[j] = __fdiv_rn(inter, uni);
            // insertion into the sorted prefix (stable: an equal key never moves in front of an earlier one)
            int pos = j;
            while (pos > 0 && ybt_before(iou[j], iou[order[pos - 1]])) { order[pos] = order[pos - 1]; --pos; }
            order[pos] = j;
        }
        bool has[3] = {false, false, false};
        for (int q = 0; q < 9; ++q) {
            const int idx = order[q], s = idx / 3, an = idx - s * 3;
            const int ny = A.ny[s], nx = A.nx[s];
            const int i = (int)((double)ny * y), j = (int)((double)nx * x);        // :152
            if (i < 0 || i >= ny || j < 0 || j >= nx) continue;                    // (the reference raises IndexError here)
            float* cell = A.dense[s] + ((((int64_t)b * 3 + an) * ny + i) * nx + j) * 6;
            const bool taken = cell[4] != 0.f;
            if (!taken && !has[s]) {
                cell[4] = 1.f;
                cell[0] = (float)((double)nx * x - (double)j);
                cell[1] = (float)((double)ny * y - (double)i);
                cell[2] = (float)(w * (double)nx);
                cell[3] = (float)(h * (double)ny);
                cell[5] = (float)(int)cls;
                has[s] = true;
            } else if (!taken && iou[idx] > ignore_thr) {
                cell[4] = -1.f;                                                    // :190
            }
        }
    }
    if (b == B - 1)
        for (int i = 0; i < 18; ++i) anc_out[i] = a[i];
}

extern "C" int y5m_yolo_build_targets(const double* boxes, const int32_t* img_off, int B, const int* ny, const int* nx,
                                      const int* stride, const float* anchors_in, float* anchors_out, float ignore_iou_thresh,
                                      float* const dense[3], void* stream) {
    Y5M_REQUIRE(B > 0 && img_off && anchors_in && anchors_out && anchors_in != anchors_out, "arguments");
    YbtArgs A;
    for (int s = 0; s < 3; ++s) {
        Y5M_REQUIRE(dense[s] && ny[s] > 0 && nx[s] > 0, "dense targets");
        A.dense[s] = dense[s]; A.ny[s] = ny[s]; A.nx[s] = nx[s]; A.stride[s] = stride[s];
    }
    hipLaunchKernelGGL(yolo_build_targets_kernel, dim3((unsigned)B), dim3(256), 0, y5m_stream(stream), boxes, img_off, B, A,
                       anchors_in, anchors_out, ignore_iou_thresh);
    Y5M_CHECK_LAUNCH("yolo_build_targets_kernel");
    return Y5M_OK;
}
