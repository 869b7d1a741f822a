// Detect path for gfx950: box decode, per-image NMS (one workgroup per image, LDS-resident
// top-K selection + bitonic sort + wave-ballot greedy scan) and the IoU/GIoU entry points.
//
// Compiled with -ffp-contract=off: the NMS arithmetic must reproduce the CPU reference bit for bit
// (SURVEY B.5/B.7): IEEE fp32 division, no FMA fusion, double compare against the IoU threshold.
#include "y5m_box.h"

// =================================================================================================
// decode  (reference utils/plot_utils.py:10-54)
// =================================================================================================
#define DEC_THREADS 64
#define DEC_WAVES (DEC_THREADS / 64)

// One wave decodes 64 consecutive cells. The 64*(5+nc) logits of those cells are one contiguous
// run in HBM: they are read with coalesced 16-byte loads into an LDS tile with an odd row stride,
// then lane i walks the channels of cell i conflict-free.
//
// Specialisation for a compile-time channel count (85 = 5 + 80 classes), full tiles only. A tile is
// exactly NV = 64*NCH/4 float4; NCH is odd, so the conflict-free row stride equals NCH and the LDS image
// of a tile IS the global byte stream (16-byte linear LDS writes, no (cell, channel) arithmetic).
// The grid is persistent: a one-wave block walks tiles blockIdx.x, +gridDim.x, ... and issues ALL loads of
// its next tile before it starts the per-cell phase of the current one, so the memory pipe stays busy
// during the LDS/ALU phase (measured: the load stream alone runs at 5.4 TB/s, the un-pipelined kernel at 3.0).
template <int NCH>
__device__ __forceinline__ void decode_cell_c(const float* __restrict__ row, int64_t cell, int naxs, int ny, int nx,
                                              float aw0, float ah0, float aw1, float ah1, float aw2, float ah2,
                                              float stride, float* __restrict__ out, int64_t N_total,
                                              int64_t row_offset) {
    const int gx = (int)(cell % nx);
    int64_t t = cell / nx;
    const int gy = (int)(t % ny);
    t /= ny;
    const int a = (int)(t % naxs);
    const int64_t b = t / naxs;
    const float sx = sigmoidf_(row[0]), sy = sigmoidf_(row[1]);
    const float sw = sigmoidf_(row[2]), sh = sigmoidf_(row[3]);
    const float obj = sigmoidf_(row[4]);
    const float aw = a == 0 ? aw0 : (a == 1 ? aw1 : aw2);
    const float ah = a == 0 ? ah0 : (a == 1 ? ah1 : ah2);
    const float x = (2.0f * sx + (float)gx - 0.5f) * stride;
    const float y = (2.0f * sy + (float)gy - 0.5f) * stride;
    const float tw = 2.0f * sw, th = 2.0f * sh;
    const float w = (tw * tw) * (aw * stride);
    const float h = (th * th) * (ah * stride);
    // argmax over SIGMOID values, first max wins (plot_utils.py:27). Branch-free walk over the LOGITS: `best` is
    // the first index of the largest logit, `prev` the largest logit before it. sigmoid is monotone, so an
    // earlier class can only tie in sigmoid space if sigmoid(prev) == sigmoid(best): only then replay the
    // reference's walk over sigmoid values.
    int best = 0;
    float best_logit = row[5], prev = -INFINITY;
#pragma unroll
    for (int c = 1; c < NCH - 5; ++c) {
        const float l = row[5 + c];
        const bool up = l > best_logit;
        prev = up ? best_logit : prev;
        best = up ? c : best;
        best_logit = up ? l : best_logit;
    }
    if (sigmoidf_(prev) == sigmoidf_(best_logit)) {
        best = 0;
        float bs = sigmoidf_(row[5]);
        for (int c = 1; c < NCH - 5; ++c) {
            const float sc = sigmoidf_(row[5 + c]);
            if (sc > bs) { bs = sc; best = c; }
        }
    }
    float* o = out + (b * N_total + row_offset + ((int64_t)a * ny + gy) * nx + gx) * 6;
    reinterpret_cast<float2*>(o)[0] = make_float2((float)best, obj);
    reinterpret_cast<float2*>(o)[1] = make_float2(x, y);
    reinterpret_cast<float2*>(o)[2] = make_float2(w, h);
}

template <int NCH>
__global__ __launch_bounds__(64) void decode_pred_kernel_c(
    const float* __restrict__ logits, int64_t tiles, int naxs, int ny, int nx,
    float aw0, float ah0, float aw1, float ah1, float aw2, float ah2, float stride,
    float* __restrict__ out, int64_t N_total, int64_t row_offset) {
    static_assert((NCH & 1) == 1, "specialisation assumes an odd channel count");
    constexpr int NV = 64 * NCH / 4;            // float4 per tile
    constexpr int IT = (NV + 63) / 64;
    __shared__ __attribute__((aligned(16))) float tile[64 * NCH];
    const int lane = threadIdx.x;
    float4 q[IT];
    int64_t t = blockIdx.x;
    {
        const float4* src = reinterpret_cast<const float4*>(logits + t * 64 * NCH);
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int v = lane + 64 * it;
            q[it] = v < NV ? src[v] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (; t < tiles; t += gridDim.x) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int v = lane + 64 * it;
            if (v < NV) reinterpret_cast<float4*>(tile)[v] = q[it];
        }
        __syncthreads();
        const int64_t tn = t + gridDim.x;
        if (tn < tiles) {
            const float4* src = reinterpret_cast<const float4*>(logits + tn * 64 * NCH);
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int v = lane + 64 * it;
                q[it] = v < NV ? src[v] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        decode_cell_c<NCH>(tile + lane * NCH, t * 64 + lane, naxs, ny, nx, aw0, ah0, aw1, ah1, aw2, ah2, stride, out,
                           N_total, row_offset);
        __syncthreads();
    }
}

__global__ __launch_bounds__(DEC_THREADS) void decode_pred_kernel(
    const float* __restrict__ logits, int64_t cell_base, int64_t cells, int naxs, int ny, int nx, int nch,
    float aw0, float ah0, float aw1, float ah1, float aw2, float ah2, float stride,
    float* __restrict__ out, int64_t N_total, int64_t row_offset) {
    extern __shared__ __attribute__((aligned(16))) float dec_lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int lstride = nch | 1;
    float* tile = dec_lds + (size_t)wid * 64 * lstride;
    const int64_t c0 = cell_base + ((int64_t)blockIdx.x * DEC_WAVES + wid) * 64;   // cells [cell_base, cells)
    if (c0 >= cells) return;
    const int ncell = (int)((cells - c0) < 64 ? (cells - c0) : 64);
    const int nelem = ncell * nch;
    const float* src = logits + c0 * nch;
    // coalesced fill: element e -> (cell = e / nch, ch = e % nch), tracked incrementally
    if (((uintptr_t)src & 15) == 0) {
        const int nvec = nelem >> 2;
        // element e = 4*v -> (cell, ch); v advances by 64 per iteration = 256 elements = d256 cells + r256
        // channels: tracked incrementally, no division in the loop
        int cell0 = (4 * lane) / nch, ch0 = 4 * lane - cell0 * nch;
        const int d256 = 256 / nch, r256 = 256 - d256 * nch;
        for (int v = lane; v < nvec; v += 64) {
            const float4 q = *reinterpret_cast<const float4*>(src + 4 * v);
            int cell = cell0, ch = ch0;
            const float vals[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tile[cell * lstride + ch] = vals[k];
                if (++ch == nch) { ch = 0; ++cell; }
            }
            cell0 += d256; ch0 += r256;
            if (ch0 >= nch) { ch0 -= nch; ++cell0; }
        }
        for (int e = (nvec << 2) + lane; e < nelem; e += 64) {
            int cell = e / nch, ch = e - cell * nch;
            tile[cell * lstride + ch] = src[e];
        }
    } else {
        for (int e = lane; e < nelem; e += 64) {
            int cell = e / nch, ch = e - cell * nch;
            tile[cell * lstride + ch] = src[e];
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane >= ncell) return;
    const float* row = tile + lane * lstride;
    const int64_t cell = c0 + lane;
    const int gx = (int)(cell % nx);
    int64_t t = cell / nx;
    const int gy = (int)(t % ny);
    t /= ny;
    const int a = (int)(t % naxs);
    const int64_t b = t / naxs;
    const float sx = sigmoidf_(row[0]), sy = sigmoidf_(row[1]);
    const float sw = sigmoidf_(row[2]), sh = sigmoidf_(row[3]);
    const float obj = sigmoidf_(row[4]);
    const float aw = a == 0 ? aw0 : (a == 1 ? aw1 : aw2);
    const float ah = a == 0 ? ah0 : (a == 1 ? ah1 : ah2);
    // :25  xy = (2*s + grid - 0.5) * stride ;  :26  wh = ((2*s)**2) * (anchors*stride)
    const float x = (2.0f * sx + (float)gx - 0.5f) * stride;
    const float y = (2.0f * sy + (float)gy - 0.5f) * stride;
    const float tw = 2.0f * sw, th = 2.0f * sh;
    const float w = (tw * tw) * (aw * stride);
    const float h = (th * th) * (ah * stride);
    // :27 argmax over SIGMOID values (first max wins). sigmoid is monotone, so a class whose logit
    // does not exceed the best logit so far cannot have a strictly larger sigmoid: skip its exp.
    int best = 0;
    float best_logit = row[5], best_sig = sigmoidf_(row[5]);
    for (int c = 1; c < nch - 5; ++c) {
        const float l = row[5 + c];
        if (l > best_logit) {
            const float s = sigmoidf_(l);
            if (s > best_sig) { best_sig = s; best = c; best_logit = l; }
        }
    }
    float* o = out + (b * N_total + row_offset + ((int64_t)a * ny + gy) * nx + gx) * 6;
    reinterpret_cast<float2*>(o)[0] = make_float2((float)best, obj);
    reinterpret_cast<float2*>(o)[1] = make_float2(x, y);
    reinterpret_cast<float2*>(o)[2] = make_float2(w, h);
}

extern "C" int y5m_decode_scale(const float* logits, int B, int naxs, int ny, int nx, int nc,
                                const float* anchors_scale_host, float stride, float* out,
                                int64_t N_total, int64_t row_offset, void* stream) {
    Y5M_REQUIRE(naxs == 3, "decode supports 3 anchors per scale");
    Y5M_REQUIRE(B >= 0 && ny > 0 && nx > 0 && nc >= 1, "bad dims");
    const int64_t cells = (int64_t)B * naxs * ny * nx;
    if (cells == 0) return Y5M_OK;
    const int nch = 5 + nc;
    const size_t lds = (size_t)DEC_WAVES * 64 * (nch | 1) * sizeof(float);
    Y5M_REQUIRE(lds <= 160 * 1024, "nc too large for the LDS tile");
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)decode_pred_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const float* a = anchors_scale_host;
    int64_t cell_base = 0;
    if (nch == 85 && (((uintptr_t)logits) & 15) == 0 && cells >= 64) {
        // full tiles on the pipelined kernel (persistent grid: 7 one-wave blocks fit a CU's LDS); a ragged last
        // tile falls through to the generic kernel
        const int64_t tiles = cells / 64;
        static int grid_cap = 0;
        if (!grid_cap) {
            const char* e = getenv("Y5M_DECODE_GRID");
            grid_cap = e ? atoi(e) : 7 * 256;
            if (grid_cap < 1) grid_cap = 1;
        }
        hipLaunchKernelGGL(decode_pred_kernel_c<85>, dim3((unsigned)(tiles < grid_cap ? tiles : grid_cap)), dim3(64), 0,
                           y5m_stream(stream), logits, tiles, naxs, ny, nx, a[0], a[1], a[2], a[3], a[4], a[5], stride, out,
                           N_total, row_offset);
        Y5M_CHECK_LAUNCH("decode_pred_kernel_c<85>");
        cell_base = tiles * 64;
        if (cell_base == cells) return Y5M_OK;
    }
    const int64_t blocks = (cells - cell_base + DEC_THREADS - 1) / DEC_THREADS;
    hipLaunchKernelGGL(decode_pred_kernel, dim3((unsigned)blocks), dim3(DEC_THREADS), lds, y5m_stream(stream),
                       logits, cell_base, cells, naxs, ny, nx, nch, a[0], a[1], a[2], a[3], a[4], a[5], stride, out,
                       N_total, row_offset);
    Y5M_CHECK_LAUNCH("decode_pred_kernel");
    return Y5M_OK;
}

__global__ void decode_tgt_kernel(const float* __restrict__ tg, int64_t cells, int naxs, int ny, int nx,
                                  float stride, float* __restrict__ out, int64_t N_total, int64_t row_offset) {
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= cells) return;
    const float* r = tg + cell * 6;
    const int gx = (int)(cell % nx);
    int64_t t = cell / nx;
    const int gy = (int)(t % ny);
    t /= ny;
    const int a = (int)(t % naxs);
    const int64_t b = t / naxs;
    float* o = out + (b * N_total + row_offset + ((int64_t)a * ny + gy) * nx + gx) * 6;
    o[0] = r[5];
    o[1] = r[4];
    o[2] = (r[0] + (float)gx) * stride;   // plot_utils.py:32
    o[3] = (r[1] + (float)gy) * stride;
    o[4] = r[2] * stride;                 // :33
    o[5] = r[3] * stride;
}

extern "C" int y5m_decode_targets_scale(const float* tgt, int B, int naxs, int ny, int nx, float stride,
                                        float* out, int64_t N_total, int64_t row_offset, void* stream) {
    const int64_t cells = (int64_t)B * naxs * ny * nx;
    if (cells == 0) return Y5M_OK;
    hipLaunchKernelGGL(decode_tgt_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, y5m_stream(stream),
                       tgt, cells, naxs, ny, nx, stride, out, N_total, row_offset);
    Y5M_CHECK_LAUNCH("decode_tgt_kernel");
    return Y5M_OK;
}

// =================================================================================================
// check_class_accuracy (reference utils/validation_utils.py:44-83): per scale, over the cells the dense target marks as
// objects (t[..., 4] == 1): how many there are, how many have argmax(logits[5:]) == t[..., 5], how many have
// sigmoid(logits[..., 0]) > conf (the reference reads the objectness from channel 0, :66 -- reproduced). One wave scans 64
// cells' flags at a time and resolves the marked ones cooperatively (the 5 + nc logits of a row across the lanes, first
// maximum wins like torch.argmax); three 64-bit counters per launch, accumulated with atomics (the caller zeroes them).
// =================================================================================================
__global__ __launch_bounds__(256) void class_obj_accuracy_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                                 int64_t cells, int nch, float conf, unsigned long long* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    unsigned long long n_obj = 0, n_cls = 0, n_ok = 0;
    for (int64_t c0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64; c0 < cells; c0 += nwaves * 64) {
        const int64_t cell = c0 + lane;
        const bool isobj = cell < cells && tgt[cell * 6 + 4] == 1.0f;                       // :60
        unsigned long long m = __ballot(isobj);
        while (m) {
            const int cl = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            const float* row = out + (c0 + cl) * nch;
            // argmax over channels 5 .. nch-1: every lane takes channels lane+5, lane+69, ...; (value, index) reduced with
            // "greater value, or equal value and smaller index" = the first maximum
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int ch = 5 + lane; ch < nch; ch += 64) {
                const float v = row[ch];
                if (v > bv || (v == bv && ch < bi) || bi == 0x7fffffff) { bv = v; bi = ch; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
            }
            if (lane == 0) {
                const float t5 = tgt[(c0 + cl) * 6 + 5];
                const float s0 = 1.0f / (1.0f + expf(-row[0]));                              // :66 (channel 0)
                n_obj += 1;
                n_cls += ((float)(bi - 5) == t5) ? 1 : 0;                                     // :62
                n_ok += (s0 > conf) ? 1 : 0;                                                  // :67 (obj_preds == 1 on object cells)
            }
        }
    }
    if (lane == 0 && n_obj) {
        atomicAdd(&counts[0], n_obj);
        atomicAdd(&counts[1], n_cls);
        atomicAdd(&counts[2], n_ok);
    }
}

extern "C" int y5m_class_obj_accuracy(const float* out, const float* tgt, int64_t cells, int nch, float conf_threshold,
                                      int64_t* counts, void* stream) {
    Y5M_REQUIRE(out && tgt && counts && nch > 5 && cells >= 0, "y5m_class_obj_accuracy: bad arguments");
    if (cells == 0) return Y5M_OK;
    const int64_t blocks = (cells + 255) / 256;
    hipLaunchKernelGGL(class_obj_accuracy_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, y5m_stream(stream),
                       out, tgt, cells, nch, conf_threshold, reinterpret_cast<unsigned long long*>(counts));
    Y5M_CHECK_LAUNCH("class_obj_accuracy_kernel");
    return Y5M_OK;
}

// =================================================================================================
// NMS  (reference utils/bboxes_utils.py:175-209 + torchvision 0.12 nms_kernel.cpp semantics)
// =================================================================================================
#define NMS_T 1024          // threads per image-workgroup (16 waves)
#define NMS_CAP 4096        // keys sorted per round
#define NMS_MAXK 1024       // upper bound for max_det
#define NMS_HBITS 11
#define NMS_HBINS (1 << NMS_HBITS)

struct NmsLds {
    unsigned long long keys[NMS_CAP];
    float kbox[NMS_MAXK * 4];
    float karea[NMS_MAXK];
    int kidx[NMS_MAXK];           // source row of every kept box (rows are written out at the end)
    unsigned long long colmask[64];   // colmask[i]: lanes j > i of the current 64-candidate tile that candidate i suppresses
    unsigned long long deadm[NMS_T / 64];   // per wave: tile lanes suppressed by boxes kept earlier in this group
    float sbox[NMS_T * 4];
    float sarea[NMS_T];
    int sidx[NMS_T];
    unsigned hist[NMS_HBINS];
    int wave_tot[NMS_T / 64];
    int red[NMS_T / 64];
    int s_m;                      // gathered count
    int s_nk;                     // kept so far
    int s_ns;                     // survivors of the current group
    unsigned s_digit;
    unsigned s_need;
    int s_done;
};

struct NmsBox { float x1, y1, x2, y2, area; };

// MODE 0: reference non_max_suppression (utils/bboxes_utils.py:175-209 + torchvision nms)
// MODE 1 / 2: non_max_suppression_aladdin (utils/bboxes_utils.py:129-173), box_format corners / midpoint:
//   candidates truncated to max_detections BEFORE suppression; a kept box removes later boxes of the SAME class
//   with intersection_over_union(kept, box) >= iou_threshold (A8: fp32, eps = 1e-7 in the union); rows are
//   returned unchanged. NmsBox.area carries the class id in these modes.
// reference utils/bboxes_utils.py:190-195 for one candidate row (exact op order)
template <int MODE>
__device__ __forceinline__ void nms_load(const float* __restrict__ row, float& cls, float& score,
                                         float& x1, float& y1, float& x2, float& y2, NmsBox& ob) {
    cls = row[0]; score = row[1];
    if constexpr (MODE == 0) {
        const float x = row[2], y = row[3], w = row[4], h = row[5];
        x1 = x - (w / 2.0f);          // :190
        y1 = y - (h / 2.0f);          // :191
        y2 = h + y1;                  // :192
        x2 = w + x1;                  // :193
        ob.x1 = x1 + cls; ob.y1 = y1 + cls; ob.x2 = x2 + cls; ob.y2 = y2 + cls;   // :195
        ob.area = (ob.x2 - ob.x1) * (ob.y2 - ob.y1);
    } else {
        x1 = row[2]; y1 = row[3]; x2 = row[4]; y2 = row[5];                       // returned as given
        if constexpr (MODE == 2) {    // bboxes_utils.py:52-60 (midpoint -> corners, the operation order of A8)
            ob.x1 = row[2] - row[4] / 2.0f; ob.y1 = row[3] - row[5] / 2.0f;
            ob.x2 = row[2] + row[4] / 2.0f; ob.y2 = row[3] + row[5] / 2.0f;
        } else {
            ob.x1 = x1; ob.y1 = y1; ob.x2 = x2; ob.y2 = y2;
        }
        ob.area = cls;
    }
}

// MODE 0: torchvision nms_kernel.cpp inner test; i = already kept (higher score), j = candidate
// MODE 1/2: bboxes_utils.py:160-168: drop j unless class differs or iou(i, j) < float32(threshold)
template <int MODE>
__device__ __forceinline__ bool nms_suppresses(float ix1, float iy1, float ix2, float iy2, float iarea,
                                               const NmsBox& j, double thr) {
    const float xx1 = ix1 > j.x1 ? ix1 : j.x1;
    const float yy1 = iy1 > j.y1 ? iy1 : j.y1;
    const float xx2 = ix2 < j.x2 ? ix2 : j.x2;
    const float yy2 = iy2 < j.y2 ? iy2 : j.y2;
    float w = xx2 - xx1; w = w > 0.0f ? w : 0.0f;
    float h = yy2 - yy1; h = h > 0.0f ? h : 0.0f;
    const float inter = w * h;
    if constexpr (MODE == 0) {
        const float ovr = inter / (iarea + j.area - inter);
        return (double)ovr > thr;
    } else {
        if (iarea != j.area) return false;                               // different class: kept (:163)
        const float w1 = ix2 - ix1, h1 = iy2 - iy1, w2 = j.x2 - j.x1, h2 = j.y2 - j.y1;
        const float uni = w1 * h1 + w2 * h2 - inter + 1e-7f;             // :78
        const float iou = inter / uni;
        return !(iou < (float)thr);                                      // kept only if iou < threshold (:164-168)
    }
}

// order-preserving transform: larger score -> smaller key (ascending sort = best first)
__device__ __forceinline__ unsigned desc_key(float s) {
    const unsigned u = __float_as_uint(s);
    const unsigned asc = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
    return ~asc;
}

__device__ __forceinline__ int block_sum(int v, int* red) {
    v = wave_sum_i(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int i = 0; i < NMS_T / 64; ++i) t += red[i];
    __syncthreads();
    return t;
}

template <int MODE>
__global__ __launch_bounds__(NMS_T) void nms_kernel(
    const float* __restrict__ boxes, int64_t N, float conf_thr, double conf_thr_d, double iou_thr, int max_det, int ibits,
    float* __restrict__ out_rows, int32_t* __restrict__ out_idx, int32_t* __restrict__ out_count,
    unsigned* __restrict__ ws_keys) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nms_raw[];
    NmsLds& L = *reinterpret_cast<NmsLds*>(nms_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t img = blockIdx.x;
    const float* bx = boxes + img * N * 6;
    unsigned* skey = ws_keys + img * N;
    float* orow = out_rows + img * (int64_t)max_det * 6;
    int32_t* oidx = out_idx + img * (int64_t)max_det;
    const int kbits = 32 + ibits;
    const unsigned long long imask = (1ull << ibits) - 1ull;

    // ---- phase A: candidate filter (:186 strict >) + score keys ------------------------------
    int my = 0;
    for (int64_t i = tid; i < N; i += NMS_T) {
        const float s = bx[i * 6 + 1];
        // MODE 0: tensor > python float compares in float32 (:186); aladdin: python floats, i.e. doubles (:149)
        const bool c = MODE == 0 ? s > conf_thr : (double)s > conf_thr_d;
        skey[i] = c ? desc_key(s) : 0xFFFFFFFFu;
        my += c ? 1 : 0;
    }
    if (tid == 0) { L.s_nk = 0; L.s_done = 0; }
    int remaining = block_sum(my, L.red);     // contains the barriers that publish skey to the block
    unsigned long long lo_excl = 0ull;        // keys <= lo_excl are already consumed
    int nk_seen = 0;

    while (remaining > 0 && nk_seen < max_det) {
        // ---- select T = the need-th smallest unconsumed key (radix select, 11-bit digits) -----
        const int need0 = remaining < NMS_CAP ? remaining : NMS_CAP;
        unsigned long long T;
        if (remaining <= NMS_CAP) {
            T = ~0ull;
        } else {
            unsigned long long prefix = 0ull;
            int pbits = 0;
            unsigned need = (unsigned)need0;
            bool early = false;
            while (pbits < kbits) {
                const int w = (kbits - pbits) < NMS_HBITS ? (kbits - pbits) : NMS_HBITS;
                const int rest = kbits - pbits - w;
                for (int i = tid; i < NMS_HBINS; i += NMS_T) L.hist[i] = 0u;
                __syncthreads();
                for (int64_t i = tid; i < N; i += NMS_T) {
                    const unsigned k32 = skey[i];
                    if (k32 == 0xFFFFFFFFu) continue;
                    const unsigned long long k = ((unsigned long long)k32 << ibits) | (unsigned long long)i;
                    if (k <= lo_excl) continue;
                    if (pbits > 0 && (k >> (kbits - pbits)) != prefix) continue;
                    atomicAdd(&L.hist[(unsigned)((k >> rest) & ((1ull << w) - 1ull))], 1u);
                }
                __syncthreads();
                if (wid == 0) {
                    // lane l owns bins [l*32, l*32+32)
                    unsigned s = 0;
                    for (int q = 0; q < NMS_HBINS / 64; ++q) s += L.hist[lane * (NMS_HBINS / 64) + q];
                    unsigned incl = s;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned v = __shfl_up(incl, o, 64);
                        if (lane >= o) incl += v;
                    }
                    const unsigned excl = incl - s;
                    if (excl < need && need <= incl) {
                        unsigned cum = excl;
                        for (int q = 0; q < NMS_HBINS / 64; ++q) {
                            const unsigned hcount = L.hist[lane * (NMS_HBINS / 64) + q];
                            if (cum + hcount >= need) {
                                L.s_digit = (unsigned)(lane * (NMS_HBINS / 64) + q);
                                L.s_need = need - cum;
                                L.s_done = (cum + hcount == need) ? 1 : 0;
                                break;
                            }
                            cum += hcount;
                        }
                    }
                }
                __syncthreads();
                prefix = (prefix << w) | (unsigned long long)L.s_digit;
                pbits += w;
                need = L.s_need;
                const int done = L.s_done;
                __syncthreads();
                if (done) { early = true; break; }
            }
            // early: the whole bucket `prefix` is taken -> every completion of the low bits
            T = early && pbits < kbits ? ((prefix << (kbits - pbits)) | ((1ull << (kbits - pbits)) - 1ull)) : prefix;
        }
        // ---- gather the chunk (lo_excl, T] into LDS and sort it ---------------------------------
        if (tid == 0) L.s_m = 0;
        __syncthreads();
        for (int64_t i = tid; i < N; i += NMS_T) {
            const unsigned k32 = skey[i];
            if (k32 == 0xFFFFFFFFu) continue;
            const unsigned long long k = ((unsigned long long)k32 << ibits) | (unsigned long long)i;
            if (k > lo_excl && k <= T) {
                const int pos = atomicAdd(&L.s_m, 1);
                if (pos < NMS_CAP) L.keys[pos] = k;
            }
        }
        __syncthreads();
        int m = L.s_m < NMS_CAP ? L.s_m : NMS_CAP;
        int n2 = 64;
        while (n2 < m) n2 <<= 1;
        for (int i = m + tid; i < n2; i += NMS_T) L.keys[i] = ~0ull;
        __syncthreads();
        for (int k = 2; k <= n2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (n2 >> 1); t += NMS_T) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int p = i | j;
                    const bool up = (i & k) == 0;
                    const unsigned long long a = L.keys[i], b = L.keys[p];
                    if ((a > b) == up) { L.keys[i] = b; L.keys[p] = a; }
                }
                __syncthreads();
            }
        }
        // aladdin (:151-152): only the max_detections best candidates enter the suppression at all
        if (MODE != 0 && m > max_det) m = max_det;
        // ---- greedy scan of the sorted chunk ---------------------------------------------------
        for (int base = 0; base < m; base += NMS_T) {
            const int nk0 = L.s_nk;
            const int ci = base + tid;
            const bool valid = ci < m;
            NmsBox cb = {0.f, 0.f, 0.f, 0.f, 0.f};
            int src = 0;
            if (valid) {
                src = (int)(L.keys[ci] & imask);
                float cls, sc, x1, y1, x2, y2;
                nms_load<MODE>(bx + (int64_t)src * 6, cls, sc, x1, y1, x2, y2, cb);
            }
            bool alive = valid;
            for (int k = 0; k < nk0 && alive; ++k) {
                if (nms_suppresses<MODE>(L.kbox[4 * k], L.kbox[4 * k + 1], L.kbox[4 * k + 2], L.kbox[4 * k + 3],
                                   L.karea[k], cb, iou_thr))
                    alive = false;
            }
            const unsigned long long bal = __ballot(alive);
            if (lane == 0) L.wave_tot[wid] = __popcll(bal);
            __syncthreads();
            int wbase = 0, ns = 0;
#pragma unroll
            for (int q = 0; q < NMS_T / 64; ++q) {
                const int c = L.wave_tot[q];
                if (q < wid) wbase += c;
                ns += c;
            }
            if (alive) {
                const int pos = wbase + __popcll(bal & ((1ull << lane) - 1ull));
                L.sbox[4 * pos] = cb.x1; L.sbox[4 * pos + 1] = cb.y1;
                L.sbox[4 * pos + 2] = cb.x2; L.sbox[4 * pos + 3] = cb.y2;
                L.sarea[pos] = cb.area;
                L.sidx[pos] = src;
            }
            __syncthreads();
            // ---- resolve the survivors in tiles of 64 (sorted order = lane order), all 16 waves per tile:
            //   (1) wave w tests the tile against the boxes kept earlier in THIS group, k = nk0+w, nk0+w+16, ...
            //   (2) wave w tests tile candidates i = 4w..4w+3 against every later lane j  -> colmask[i]
            //   (3) wave 0 walks the tile on the scalar unit: first alive lane is kept, alive &= ~colmask[kept];
            //       nothing on that chain touches memory (64-bit masks in SGPRs, v_readlane of the column masks)
            int nk = nk0;
            for (int sb = 0; sb < ns && nk < max_det; sb += 64) {
                const int j = sb + lane;
                const bool jv = j < ns;
                NmsBox mb = {0.f, 0.f, 0.f, 0.f, 0.f};
                int msrc = 0;
                if (jv) {
                    mb.x1 = L.sbox[4 * j]; mb.y1 = L.sbox[4 * j + 1];
                    mb.x2 = L.sbox[4 * j + 2]; mb.y2 = L.sbox[4 * j + 3];
                    mb.area = L.sarea[j];
                    msrc = L.sidx[j];
                }
                bool dead = false;
                for (int k = nk0 + wid; k < nk; k += NMS_T / 64)
                    dead |= nms_suppresses<MODE>(L.kbox[4 * k], L.kbox[4 * k + 1], L.kbox[4 * k + 2], L.kbox[4 * k + 3],
                                                 L.karea[k], mb, iou_thr);
                const unsigned long long dm = __ballot(dead && jv);
                if (lane == 0) L.deadm[wid] = dm;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 4 * wid + q;
                    const int ci = sb + i < ns ? sb + i : ns - 1;          // clamped: masked below
                    const bool sup = nms_suppresses<MODE>(L.sbox[4 * ci], L.sbox[4 * ci + 1], L.sbox[4 * ci + 2],
                                                          L.sbox[4 * ci + 3], L.sarea[ci], mb, iou_thr);
                    const unsigned long long cm = __ballot(sup && jv && lane > i && sb + i < ns);
                    if (lane == 0) L.colmask[i] = cm;
                }
                __syncthreads();
                if (wid == 0) {
                    unsigned long long alive = __ballot(jv);
#pragma unroll
                    for (int q = 0; q < NMS_T / 64; ++q) alive &= ~L.deadm[q];
                    // (every lane holds the same value: make that explicit so the walk below stays on the scalar unit)
                    alive = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(alive >> 32)) << 32) |
                            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(alive & 0xFFFFFFFFull));
                    const unsigned long long mycm = L.colmask[lane];
                    const int cm_lo = (int)(unsigned)(mycm & 0xFFFFFFFFull), cm_hi = (int)(unsigned)(mycm >> 32);
                    unsigned long long keptm = 0ull;
                    int cnt = 0;
                    const int room = max_det - nk;
                    while (alive != 0ull && cnt < room) {
                        const int i = __builtin_amdgcn_readfirstlane(__ffsll((long long)alive) - 1);
                        const unsigned lo = (unsigned)__builtin_amdgcn_readlane(cm_lo, i);
                        const unsigned hi = (unsigned)__builtin_amdgcn_readlane(cm_hi, i);
                        keptm |= 1ull << i;
                        ++cnt;
                        alive &= ~(((unsigned long long)hi << 32) | (unsigned long long)lo);
                        alive &= ~(1ull << i);
                    }
                    if ((keptm >> lane) & 1ull) {
                        const int pos = nk + __popcll(keptm & ((1ull << lane) - 1ull));
                        L.kbox[4 * pos] = mb.x1; L.kbox[4 * pos + 1] = mb.y1;
                        L.kbox[4 * pos + 2] = mb.x2; L.kbox[4 * pos + 3] = mb.y2;
                        L.karea[pos] = mb.area;
                        L.kidx[pos] = msrc;
                    }
                    if (lane == 0) L.s_nk = nk + cnt;
                }
                __syncthreads();
                nk = L.s_nk;
            }
            __syncthreads();
            if (L.s_nk >= max_det) break;
        }
        nk_seen = L.s_nk;
        lo_excl = T;
        remaining -= m;
        __syncthreads();
        if (MODE != 0) break;            // aladdin: one pass over the truncated list
    }
    // ---- output: row k = the reference's row for source box kidx[k] (all threads, off the serial chain) ----
    __syncthreads();
    const int nkf = L.s_nk;
    for (int k = tid; k < nkf; k += NMS_T) {
        const int src = L.kidx[k];
        float cls, sc, x1, y1, x2, y2;
        NmsBox tmp;
        nms_load<MODE>(bx + (int64_t)src * 6, cls, sc, x1, y1, x2, y2, tmp);
        float* o = orow + (int64_t)k * 6;
        o[0] = cls; o[1] = sc; o[2] = x1; o[3] = y1; o[4] = x2; o[5] = y2;
        oidx[k] = src;
    }
    if (tid == 0) out_count[img] = nkf;
}

extern "C" size_t y5m_nms_workspace_bytes(int B, int64_t N) {
    return y5m_align((size_t)(B > 0 ? B : 0) * (size_t)(N > 0 ? N : 0) * sizeof(unsigned)) + 256;
}

extern "C" int y5m_nms(const float* boxes, int B, int64_t N, float conf_threshold, double iou_threshold,
                       int max_det, float* out_rows, int32_t* out_idx, int32_t* out_count, void* ws,
                       size_t ws_bytes, void* stream) {
    Y5M_REQUIRE(B >= 0 && N >= 0, "bad dims");
    Y5M_REQUIRE(max_det >= 1 && max_det <= NMS_MAXK, "max_det must be in [1,1024]");
    Y5M_REQUIRE(N < (1ll << 31), "N too large");
    if (B == 0) return Y5M_OK;
    if (ws_bytes < y5m_nms_workspace_bytes(B, N)) { y5m_set_error("nms workspace too small"); return Y5M_EWS; }
    int ibits = 1;
    while ((1ll << ibits) < N) ++ibits;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)nms_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(NmsLds));
        attr_set = true;
    }
    hipLaunchKernelGGL(nms_kernel<0>, dim3((unsigned)B), dim3(NMS_T), sizeof(NmsLds), y5m_stream(stream),
                       boxes, N, conf_threshold, (double)conf_threshold, iou_threshold, max_det, ibits, out_rows, out_idx,
                       out_count, reinterpret_cast<unsigned*>(ws));
    Y5M_CHECK_LAUNCH("nms_kernel");
    return Y5M_OK;
}

// non_max_suppression_aladdin (reference utils/bboxes_utils.py:129-173) for B independent box lists of N rows
// [class, score, 4 coordinates]; midpoint = box_format == "midpoint". Same workspace as y5m_nms.
extern "C" int y5m_nms_aladdin(const float* boxes, int B, int64_t N, double threshold, float iou_threshold, int midpoint,
                               int max_det, float* out_rows, int32_t* out_idx, int32_t* out_count, void* ws,
                               size_t ws_bytes, void* stream) {
    Y5M_REQUIRE(B >= 0 && N >= 0, "bad dims");
    Y5M_REQUIRE(max_det >= 1 && max_det <= NMS_MAXK, "max_det must be in [1,1024]");
    Y5M_REQUIRE(N < (1ll << 31), "N too large");
    if (B == 0) return Y5M_OK;
    if (ws_bytes < y5m_nms_workspace_bytes(B, N)) { y5m_set_error("nms workspace too small"); return Y5M_EWS; }
    int ibits = 1;
    while ((1ll << ibits) < N) ++ibits;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)nms_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(NmsLds));
        hipFuncSetAttribute((const void*)nms_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(NmsLds));
        attr_set = true;
    }
    if (midpoint)
        hipLaunchKernelGGL(nms_kernel<2>, dim3((unsigned)B), dim3(NMS_T), sizeof(NmsLds), y5m_stream(stream), boxes, N,
                           0.0f, threshold, (double)iou_threshold, max_det, ibits, out_rows, out_idx, out_count,
                           reinterpret_cast<unsigned*>(ws));
    else
        hipLaunchKernelGGL(nms_kernel<1>, dim3((unsigned)B), dim3(NMS_T), sizeof(NmsLds), y5m_stream(stream), boxes, N,
                           0.0f, threshold, (double)iou_threshold, max_det, ibits, out_rows, out_idx, out_count,
                           reinterpret_cast<unsigned*>(ws));
    Y5M_CHECK_LAUNCH("nms_kernel(aladdin)");
    return Y5M_OK;
}

// =================================================================================================
// IoU / GIoU entry points (reference utils/bboxes_utils.py:33-87)
// =================================================================================================
__global__ void iou_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, int giou,
                           float eps, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 qa = reinterpret_cast<const float4*>(a)[i], qb = reinterpret_cast<const float4*>(b)[i];
    const float A[4] = {qa.x, qa.y, qa.z, qa.w}, Bv[4] = {qb.x, qb.y, qb.z, qb.w};
    out[i] = box_iou_fwd(A, Bv, giou != 0, eps).out;
}

__global__ void iou_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                               const float* __restrict__ gout, int64_t n, int giou, float eps,
                               float* __restrict__ ga, float* __restrict__ gb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 qa = reinterpret_cast<const float4*>(a)[i], qb = reinterpret_cast<const float4*>(b)[i];
    const float A[4] = {qa.x, qa.y, qa.z, qa.w}, Bv[4] = {qb.x, qb.y, qb.z, qb.w};
    const BoxFwd r = box_iou_fwd(A, Bv, giou != 0, eps);
    float da[4], db[4];
    box_iou_bwd(r, giou != 0, gout[i], da, db);
    if (ga) reinterpret_cast<float4*>(ga)[i] = make_float4(da[0], da[1], da[2], da[3]);
    if (gb) reinterpret_cast<float4*>(gb)[i] = make_float4(db[0], db[1], db[2], db[3]);
}

extern "C" int y5m_iou(const float* a, const float* b, int64_t n, int giou, float eps, float* out, void* stream) {
    if (n <= 0) return Y5M_OK;
    hipLaunchKernelGGL(iou_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, y5m_stream(stream), a, b, n,
                       giou, eps, out);
    Y5M_CHECK_LAUNCH("iou_kernel");
    return Y5M_OK;
}

extern "C" int y5m_iou_bwd(const float* a, const float* b, const float* gout, int64_t n, int giou, float eps,
                           float* ga, float* gb, void* stream) {
    if (n <= 0) return Y5M_OK;
    hipLaunchKernelGGL(iou_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, y5m_stream(stream), a, b,
                       gout, n, giou, eps, ga, gb);
    Y5M_CHECK_LAUNCH("iou_bwd_kernel");
    return Y5M_OK;
}
