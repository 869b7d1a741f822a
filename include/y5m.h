/*
 * y5m.h -- C ABI of liby5m.so: the MI355X-native (gfx950) YOLOv5m hot path.
 *
 * The reference (AlessandroMondin/YOLOV5m) is pure Python and has no FFI of its own; the drop-in
 * boundary is the set of Python call signatures its train.py / detect.py use. Each
 * entry point below cites the reference interface it replaces (file:line under /root/reference)
 * and is what a ctypes binding inside the reference would call.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless named h_*;
 *   - the caller owns every buffer, kernels never allocate; scratch is passed as (ws, ws_bytes)
 *     and sized by the *_workspace_bytes twin;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - return 0 on success, negative Y5M_E* on error (no exceptions cross the ABI);
 *   - dtype: 0 = f32 (parity mode, f32 MFMA), 1 = bf16 (throughput mode, bf16 MFMA, f32 accumulate).
 *   - activations are pixel-major / channel-minor ("NHWC") inside the library: an activation is
 *     (ptr, ld) = base pointer + channel stride per pixel, so a tensor can live inside a wider
 *     concat buffer without copies.
 */
#ifndef Y5M_H
#define Y5M_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y5M_OK 0
#define Y5M_EINVAL (-1)   /* bad argument / unsupported shape */
#define Y5M_ELAUNCH (-2)  /* hip launch error (see y5m_last_error) */
#define Y5M_EWS (-3)      /* workspace too small */

#define Y5M_F32 0
#define Y5M_BF16 1

/* activation flags for epilogues */
#define Y5M_ACT_NONE 0
#define Y5M_ACT_SILU 1

const char* y5m_version(void);
const char* y5m_last_error(void);
/* runtime probe used by the Python side to fail loudly when no gfx950 device is usable */
int y5m_device_ok(void);
/* workgroups a persistent (one-per-CU) launch uses: the device's CU count, capped by the environment variable
 * Y5M_PERSIST_CUS (read once per process; yolov5m_amd/parallel.py sets 240 for data-parallel runs so that the collective's
 * kernels find free CUs). New entry point (the reference has no multi-GPU path). */
int y5m_persistent_cu_count(void);
/* the Y5M_R4_KERNELS bit mask in use (parsed once per process): which of the default-off round-4 kernel forms the launchers pick */
int y5m_r4_kernel_forms(void);

/* ---------------------------------------------------------------------------------------------
 * Detect path
 * ------------------------------------------------------------------------------------------- */

/* Replaces utils/plot_utils.py:10-40 cells_to_bboxes(is_pred=True) + make_grids :42-54 for ONE
 * scale. logits (B,naxs,ny,nx,5+nc) f32 -> out rows [class, obj, x, y, w, h] at
 * out[(b*N_total + row_offset + (a*ny+gy)*nx+gx)*6]. h_anchors_scale: HOST array of naxs*2 floats
 * (stride-divided, = model.head.anchors[i]). */
int y5m_decode_scale(const float* logits, int B, int naxs, int ny, int nx, int nc,
                     const float* h_anchors_scale, float stride, float* out, int64_t N_total,
                     int64_t row_offset, void* stream);

/* Replaces cells_to_bboxes(is_pred=False) (utils/plot_utils.py:29-34): dense targets
 * (B,naxs,ny,nx,6) -> rows [t5, t4, (t0+gx)*s, (t1+gy)*s, t2*s, t3*s]. */
int y5m_decode_targets_scale(const float* tgt, int B, int naxs, int ny, int nx, float stride,
                             float* out, int64_t N_total, int64_t row_offset, void* stream);

/* Replaces the per-scale counting of YOLO_EVAL.check_class_accuracy (utils/validation_utils.py:58-68): over the cells the dense
 * target tgt (cells x 6) marks as objects (tgt[..., 4] == 1), counts[0] += their number, counts[1] += those with
 * argmax(out[..., 5:]) == tgt[..., 5], counts[2] += those with sigmoid(out[..., 0]) > conf_threshold (channel 0, as the
 * reference reads it, :66). out: cells x nch logits. counts: three int64 on the device, zeroed by the caller. */
int y5m_class_obj_accuracy(const float* out, const float* tgt, int64_t cells, int nch, float conf_threshold,
                           int64_t* counts, void* stream);

/* Replaces utils/bboxes_utils.py:175-209 non_max_suppression (per-image loop :185-203 incl. the
 * torchvision.ops.nms call :195) for a whole batch in ONE launch (one workgroup per image).
 * boxes (B,N,6) rows [class, score, x, y, w, h] f32 (not modified).
 * out_rows (B,max_det,6) rows [class, score, x1, y1, x2, y2]; out_idx (B,max_det) source row in
 * [0,N); out_count (B). max_det <= 1024. Index sets are bit-exact vs the CPU reference path. */
size_t y5m_nms_workspace_bytes(int B, int64_t N);
int y5m_nms(const float* boxes, int B, int64_t N, float conf_threshold, double iou_threshold,
            int max_det, float* out_rows, int32_t* out_idx, int32_t* out_count, void* ws,
            size_t ws_bytes, void* stream);
/* non_max_suppression_aladdin (utils/bboxes_utils.py:129-173) for B independent lists of N rows
 * [class, score, c0, c1, c2, c3] (corners x1,y1,x2,y2 or, midpoint != 0, x,y,w,h): rows with score > threshold
 * (double compare), stable descending score order, TRUNCATED to max_det, then a kept row removes later rows of
 * the same class whose A8 IoU (fp32, eps 1e-7) is >= float32(iou_threshold). out_rows: the kept rows unchanged,
 * out_idx: their indices in the input list. Workspace as y5m_nms. */
int y5m_nms_aladdin(const float* boxes, int B, int64_t N, double threshold, float iou_threshold, int midpoint,
                    int max_det, float* out_rows, int32_t* out_idx, int32_t* out_count, void* ws, size_t ws_bytes,
                    void* stream);

/* Replaces utils/bboxes_utils.py:33-87 intersection_over_union(box_format="midpoint").
 * a,b (n,4) f32 -> out (n) ; giou != 0 selects GIoU. */
int y5m_iou(const float* a, const float* b, int64_t n, int giou, float eps, float* out, void* stream);
/* d(out)/d(a), d(out)/d(b) given gout (n): ga, gb (n,4) (either may be NULL). */
int y5m_iou_bwd(const float* a, const float* b, const float* gout, int64_t n, int giou, float eps,
                float* ga, float* gb, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ComputeLoss (ultralytics_loss.py)
 * ------------------------------------------------------------------------------------------- */

/* Target table of ONE scale, capacity cap = 5*naxs*nt_max rows (struct-of-arrays, device). */
typedef struct {
    int32_t* count;   /* [1]  number of valid rows n_i                                        */
    int32_t* bagg;    /* [cap*4] (image b, anchor a, gj, gi)                                   */
    float* tbox;      /* [cap*4] (gx-gi, gy-gj, gw, gh)                                        */
    float* anch;      /* [cap*2]                                                               */
    int32_t* tcls;    /* [cap]                                                                 */
} y5m_targets;

/* Replaces ComputeLoss.build_targets (ultralytics_loss.py:122-311) for all 3 scales in ONE launch.
 * targets (nt,6) [img, cls, x, y, w, h]; nt is read from *d_nt if d_nt != NULL (graph-friendly),
 * else from nt. anchors (3,naxs,2). ny[3], nx[3] host arrays. Row order = the reference's
 * (offset-major, anchor-major, target order), integers and fp32 bits exact.
 * ws: y5m_build_targets_workspace_bytes(nt_max). */
size_t y5m_build_targets_workspace_bytes(int nt_max, int naxs);
int y5m_build_targets(const float* targets, int nt, const int32_t* d_nt, int nt_max,
                      const float* anchors, int naxs, const int* ny, const int* nx, float anchor_t,
                      y5m_targets out[3], void* ws, size_t ws_bytes, void* stream);

/* Replaces ComputeLoss.__call__ (ultralytics_loss.py:60-120) forward AND its autograd backward.
 * p[i] (B,naxs,ny_i,nx_i,5+nc) f32 logits; tg = y5m_build_targets output.
 * loss_out[4] = {total*bs, lbox*lambda, lobj*lambda, lcls*lambda}.
 * grad[i] (same shape as p[i], may be NULL for forward only) receives d(loss_out[0])/d(p[i]),
 * fully overwritten. ws: y5m_compute_loss_workspace_bytes(). */
size_t y5m_compute_loss_workspace_bytes(int B, int naxs, const int* ny, const int* nx, int nt_max);
/* Gradient-producing variant for a consumer that knows the gradient's structure (the native train step): of grad only
 * the rows of the cells a target row hit are written (zeros + objectness + box / class terms); every cell's objectness
 * gradient goes to a compact plane of the workspace. Same loss_out as y5m_compute_loss. */
int y5m_compute_loss_sparse(const float* const p[3], float* const grad[3], int B, int naxs, const int* ny, const int* nx,
                            int nc, const y5m_targets tg[3], int nt_max, const float balance[3], float lambda_box,
                            float lambda_obj, float lambda_cls, float* loss_out, void* ws, size_t ws_bytes, void* stream);
/* device pointers, inside a y5m_compute_loss workspace of these dimensions, of the three per-cell owner tables
 * ([B*naxs*ny*nx] int32: index of the LAST target row that hit the cell, -1 = none) and objectness-gradient planes
 * ([B*naxs*ny*nx] f32, written by the gradient-producing calls); valid after a y5m_compute_loss[_sparse] call */
int y5m_compute_loss_owner_ptrs(void* ws, int B, int naxs, const int* ny, const int* nx, int nt_max, int32_t* owner_out[3],
                                float* gobj_out[3]);
int y5m_compute_loss(const float* const p[3], float* const grad[3], int B, int naxs, const int* ny,
                     const int* nx, int nc, const y5m_targets tg[3], int nt_max,
                     const float balance[3], float lambda_box, float lambda_obj, float lambda_cls,
                     float* loss_out, void* ws, size_t ws_bytes, void* stream);

/* Replaces YOLO_LOSS.compute_loss for all 3 scales (loss.py:195-246, summed as in :93-97) and its
 * autograd backward. dense[i] (B,naxs,ny_i,nx_i,6) = [x,y,w,h,obj,cls] targets as built by
 * YOLO_LOSS.build_targets (loss.py:101-192): obj==1 positives, obj==-1 "ignore" cells whose BCE target
 * stays -1 (reference behaviour, loss.py:220). rows_max >= number of positives per scale.
 * loss_out[4] as y5m_compute_loss; NaN when a scale has no positives (loss.py:212). */
/* Replaces YOLO_LOSS.build_targets (loss.py:101-192, the per-image Python loop over boxes x 9 anchors the reference
 * runs on the host every step) together with iou_width_height (utils/bboxes_utils.py:6-29) for a whole batch:
 * boxes (nt,5) float64 rows [cls, x, y, w, h] of all images back to back (the reference's collate_fn arrays),
 * img_off (B+1) int32 row ranges per image; ny/nx/stride host arrays of the 3 scales; dense[i] (B,3,ny_i,nx_i,6) f32,
 * fully overwritten: [x_cell, y_cell, w_cell, h_cell, obj (1 | -1 ignore | 0), class]. anchors_in (3,3,2) f32 is the
 * loss object's anchor state BEFORE this call, anchors_out (a different buffer) receives the state after it: the
 * reference divides its anchors by 640 in place once per box (bboxes_utils.py:18) and parity is defined on that
 * behaviour, bit for bit (first-come slot claims, stable descending anchor order, float64 cell arithmetic). */
int y5m_yolo_build_targets(const double* boxes, const int32_t* img_off, int B, const int* ny, const int* nx,
                           const int* stride, const float* anchors_in, float* anchors_out, float ignore_iou_thresh,
                           float* const dense[3], void* stream);
size_t y5m_compute_loss_dense_workspace_bytes(int B, int naxs, const int* ny, const int* nx, int rows_max);
int y5m_compute_loss_dense(const float* const p[3], float* const grad[3], const float* const dense[3], int B,
                           int naxs, const int* ny, const int* nx, int nc, const float* anchors, int rows_max,
                           const float balance[3], float lambda_box, float lambda_obj, float lambda_cls,
                           float* loss_out, void* ws, size_t ws_bytes, void* stream);
/* y5m_compute_loss_dense with the sparse gradient of y5m_compute_loss_sparse (the fused train step on the reference's DEFAULT
 * loss, train.py:102-106 -> loss.py:64-99): of grad only the rows of the positive cells are written, the objectness gradient
 * of every cell goes to the workspace's compact plane. y5m_compute_loss_dense_owner_ptrs hands y5m_head_grad_pack_sparse
 * the planes and the positive-row table of such a workspace (same B / naxs / ny / nx / rows_max). */
int y5m_compute_loss_dense_sparse(const float* const p[3], float* const grad[3], const float* const dense[3], int B,
                                  int naxs, const int* ny, const int* nx, int nc, const float* anchors, int rows_max,
                                  const float balance[3], float lambda_box, float lambda_obj, float lambda_cls,
                                  float* loss_out, void* ws, size_t ws_bytes, void* stream);
int y5m_compute_loss_dense_owner_ptrs(void* ws, int B, int naxs, const int* ny, const int* nx, int rows_max,
                                      int32_t* owner_out[3], float* gobj_out[3], int32_t* bagg_out[3],
                                      int32_t* count_out[3], int* cap_out);

/* ---------------------------------------------------------------------------------------------
 * Model (model.py): Conv-BN-SiLU building blocks. Activations are (ptr, ld) NHWC, dtype per call.
 * ------------------------------------------------------------------------------------------- */

#define Y5M_EPI_RAW_STATS 0   /* store raw conv output (+ per-tile channel sum / sum-of-squares)  */
#define Y5M_EPI_AFFINE_ACT 1  /* y*scale[n]+shift[n] -> act -> (+res): eval-mode folded BN         */
#define Y5M_EPI_HEAD 2        /* + bias, f32 store permuted to (B,naxs,ny,nx,nch) (model.py:165-175) */
#define Y5M_EPI_DGRAD 3       /* plain store or accumulate (data gradient)                           */

/* One implicit-GEMM convolution launch: out[m][n] = sum_{tap,c} in[pix(m,tap)][c] * w[n][tap*Cin+c].
 * Replaces nn.Conv2d of CBL (model.py:12-28) / HEADS (model.py:162) and, with transposed packed
 * weights and mirrored tap offsets, its autograd data-gradient. */
/* BatchNorm statistics without partial rows (EPI_RAW_STATS with args->bn_acc != NULL): the conv launch ADDS its per-tile
 * channel sums (sum, sum of squares) as f64 atomics into the accumulator rows bn_acc[slot][2][Np], slot = tile index
 * modulo y5m_bn_acc_slots(); stats may then be NULL and there is no y5m_bn_finalize launch: y5m_bn_act_fused derives
 * scale / shift from the accumulators itself. The caller zeroes the rows before the launch. */
int y5m_bn_acc_slots(void);
/* 1 unless the environment says Y5M_BN_FUSE=0 (A/B runs): whether callers should use the accumulator form
 * (y5m_conv_args.bn_acc + y5m_bn_act_fused, y5m_bn_bwd_fused) instead of partial rows + y5m_bn_finalize / y5m_bn_bwd */
int y5m_bn_fuse_enabled(void);

typedef struct {
    const void* in;      /* activations (ptr, ldin)                                              */
    const void* w;       /* packed weights [Np][Kp], K-contiguous, zero padded                   */
    void* out;
    int32_t B, Hin, Win, ldin;
    int32_t Hg, Wg;      /* output grid enumerated by m = (b*Hg + gy)*Wg + gx                    */
    int32_t sy, sx;      /* input y = gy*sy + dh(tap), x = gx*sx + dw(tap)                       */
    int32_t th, tw;      /* tap grid; tap = ta*tw + tb                                           */
    int32_t dh0, dhs, dw0, dws;   /* dh = dh0 + ta*dhs ; dw = dw0 + tb*dws                       */
    int32_t Cin;         /* channels per tap (multiple of 16 bytes)                              */
    int32_t K, Kp;       /* K = th*tw*Cin ; Kp = weight row stride (multiple of the K step)      */
    int32_t N;           /* output channels                                                      */
    int32_t M;           /* B*Hg*Wg                                                              */
    int32_t Hout, Wout, ldout, osy, osx, ooy, oox;  /* out pixel (b, gy*osy+ooy, gx*osx+oox)      */
    int32_t epi, act, accumulate;
    const float* scale;  /* [N] scale (AFFINE_ACT) or bias (HEAD)                                */
    const float* shift;  /* [N]                                                                  */
    const void* res;     /* optional residual (same pixel placement as out)                      */
    int32_t ldres;
    float* stats;        /* RAW_STATS: [tiles_m][2][Np] partials, or NULL                        */
    int32_t Np;          /* stats row stride >= roundup(N, y5m_conv_tile_n(N))                   */
    int32_t naxs, nch;   /* HEAD                                                                 */
    int32_t tiles_m, tiles_n;   /* filled by the library                                         */
    const void* zeros;   /* >= 16 zero bytes in device memory: padding / out-of-range chunks of a    */
                         /* tile are LOADED from here, which keeps the tile loads branch-free        */
    /* EPI_RAW_STATS only, optional: accumulator rows instead of partial rows (see y5m_bn_acc_slots) */
    double* bn_acc;               /* [slots][2][Np] f64, device memory, zeroed by the caller       */
} y5m_conv_args;

int y5m_conv_tile_n(int N);   /* channel tile (48 or 96) the library uses for N output channels */
/* EPI_RAW_STATS: number of partial rows [rows][2][Np] y5m_conv writes into args->stats for this launch (the
 * caller sizes the buffer with it and passes it to y5m_bn_finalize as tiles_m): one per 128-pixel tile on the
 * tiled / pointwise kernels, four per 256-pixel tile on the 3x3 halo-patch kernel. */
int y5m_conv_stats_rows(const y5m_conv_args* args, int dtype);
int y5m_conv(const y5m_conv_args* args, int dtype, void* stream);
/* n data-gradient problems over the same pixel grid, input tensor and output width (the 4 parity classes of a stride-2
 * convolution's data gradient) as ONE launch: their tiles are interleaved so that the shared input is fetched once.
 * Any other set of problems is run as n y5m_conv calls. */
int y5m_conv_multi(const y5m_conv_args* args, int n, int dtype, void* stream);
/* 1 when y5m_conv would run this launch on the pointwise streaming kernel (HBM-bound), 0 for the tiled one */
/* the kernel instantiation y5m_conv / y5m_conv_multi / y5m_wgrad would launch for these arguments, as text
 * ("conv_halo_kernel<6,0>"), nothing is launched: profilers and bench.py attribute times and rooflines with it */
int y5m_conv_kernel_name(const y5m_conv_args* args, int dtype, char* buf, int n);
int y5m_conv_multi_kernel_name(const y5m_conv_args* args, int cnt, int dtype, char* buf, int n);
int y5m_conv_is_pointwise(const y5m_conv_args* args, int dtype);
/* 1 when y5m_conv would run this launch on the persistent 3x3 halo-patch kernel (bf16, stride 1, >= 64 input channels,
 * output channels a multiple of 96, image width such that two input patches fit the LDS) */
int y5m_conv_is_halo(const y5m_conv_args* args, int dtype);
/* 1 when the kernel this RAW_STATS launch maps to stages its tiles' statistics in partial rows even when accumulator rows
 * (bn_acc) are given -- the persistent 3x3 halo-patch kernel: pass stats (y5m_conv_stats_rows rows) too */
int y5m_conv_stages_stats(const y5m_conv_args* args, int dtype);

/* Weight gradient of the same convolution (autograd of nn.Conv2d wrt weight):
 * dwgt[n][tap*C + c] += sum_m dy[m][n] * x[pix(m,tap)][c], f32 atomics into a ZEROED packed buffer. */
typedef struct {
    const void* dy;      /* [M][lddy] gradient wrt the conv output                               */
    const void* x;       /* conv input activations (ptr, ldx)                                    */
    float* dwgt;         /* packed f32 gradient [N][lddw]                                        */
    int32_t B, Hin, Win, ldx;
    int32_t Hg, Wg, sy, sx, th, tw, dh0, dhs, dw0, dws;   /* same meaning as y5m_conv_args       */
    int32_t C;           /* input channels per tap                                               */
    int32_t N;           /* output channels                                                      */
    int32_t M, lddy, lddw;
    int32_t ksplit;      /* pixel-range splits (<=0: library picks)                              */
    int32_t tiles_n, tiles_c;   /* filled by the library                                         */
    const void* zeros;   /* >= 16 zero bytes in device memory (see y5m_conv_args)                 */
    int32_t reserved0;   /* must be 0 */
    int32_t pad_;
} y5m_wgrad_args;
int y5m_wgrad(const y5m_wgrad_args* args, int dtype, void* stream);
int y5m_wgrad_kernel_name(const y5m_wgrad_args* args, int dtype, char* buf, int n);   /* see y5m_conv_kernel_name */
/* the launch geometry y5m_wgrad would use, nothing is launched: out = {n tiles, c tiles, tap groups, pixel-range splits, workgroups,
 * dY channels per block, X channels per block and tap, pixels per LDS chunk} (tools/wgrad_traffic.py: what the tiling re-reads) */
int y5m_wgrad_geometry(const y5m_wgrad_args* args, int dtype, int32_t out[8]);

/* Fused backward of a pointwise (1x1, stride 1) CBL with N == C in {48, 96, 192} channels, bf16 (csrc/y5m_bwd_pw.hip;
 * reference model.py:12-28 backward): dy = BatchNorm+SiLU backward of (dz, y) is formed in registers and never stored;
 * dx (+)= dy . W; dW += dy^T . x (f32 atomics); dgamma / dbeta are written. The BatchNorm reduction must have been
 * accumulated into seg[].acc by y5m_bn_bwd_fused_phase(..., phase = 1) ahead of this call. Up to two BatchNorm segments
 * side by side on the output channels (the merged C3 pair: two layers, one launch). */
typedef struct {
    int32_t c0, cn;            /* first output channel and channel count of this segment (c0 of segment 0 is 0)          */
    int32_t lddz, pad_;
    const void* dz;            /* [M][lddz] gradient wrt THIS segment's CBL output (channel 0 of the segment)               */
    const double* acc;         /* [y5m_bn_acc_slots()][2][cn] f64 rows of the reduce pass                                 */
    const float* scale; const float* shift; const float* mean; const float* invstd;   /* [cn], forward statistics       */
    float* dgamma; float* dbeta;                                                       /* [cn] outputs (may be NULL)     */
    float* dw;                 /* [cn][lddw] f32 weight gradient of THIS segment's layer, accumulated atomically           */
} y5m_bwd_pw_seg;
typedef struct {
    const void* y; const void* x;   /* [M][ldy] raw conv output (all segments side by side), [M][ldx] conv input */
    const void* wd;            /* data-gradient weight rows [C][Kp] (y5m_pack_weights mode 1), NULL with dx == NULL       */
    void* dx;                  /* [M][lddx] gradient wrt the conv input, or NULL                                           */
    const void* res;           /* accumulation source [M][ldres] (dx = res + dy.W) or NULL (accumulate: dx += dy.W)        */
    int64_t M;
    int32_t ldy, ldx, Kp, lddx, ldres, lddw;
    int32_t N, C;              /* output / input channels (N == C)                                                         */
    int32_t accumulate, act, nseg;
    y5m_bwd_pw_seg seg[2];
} y5m_bwd_pw_args;
int y5m_bwd_pw(const y5m_bwd_pw_args* args, int dtype, void* stream);
int y5m_bwd_pw_eligible(const y5m_bwd_pw_args* args, int dtype);    /* 1 when y5m_bwd_pw accepts these arguments */

/* Fused backward of the STEM CBL (csrc/y5m_bwd_stem.hip; reference model.py:181 CBL(3, first_out, 6, 2, 2), executed as a
 * 3x3 / stride 1 / pad 1 conv over the 16-channel space-to-depth image; bf16): dy = BatchNorm+SiLU backward of (dz, y) is formed
 * in registers and never stored (the stem has no data gradient); dwgt[n][tap * C + c] += sum_m dy[m][n] x[pix(m, tap)][c], f32
 * atomics into the ZEROED packed gradient (the layout y5m_wgrad writes, y5m_unpack_wgrad mode 2 reads); dgamma / dbeta are written.
 * The BatchNorm reduction must have been accumulated into acc by y5m_bn_bwd_fused_phase(..., phase = 1) ahead of this call. */
typedef struct {
    const void* dz;            /* [M][lddz] gradient wrt the CBL output, M = B * H * W                                     */
    const void* y;             /* [M][ldy]  raw conv output                                                                */
    const void* x;             /* [M][ldx]  conv input: the space-to-depth image (same pixel grid: stride 1, pad 1)        */
    float* dwgt;               /* [N][lddw] packed f32 weight gradient, accumulated atomically                             */
    int32_t B, H, W;
    int32_t lddz, ldy, ldx, lddw;
    int32_t N, C;              /* 48 output channels, 16 input channels per tap                                            */
    int32_t act, pad_, pad2_;
    const double* acc;         /* [y5m_bn_acc_slots()][2][N] f64 rows of the reduce pass                                   */
    const float* scale; const float* shift; const float* mean; const float* invstd;   /* [N], forward statistics       */
    float* dgamma; float* dbeta;                                                       /* [N] outputs (may be NULL)      */
} y5m_bwd_stem_args;
int y5m_bwd_stem(const y5m_bwd_stem_args* args, void* stream);
int y5m_bwd_stem_eligible(const y5m_bwd_stem_args* args);           /* 1 when y5m_bwd_stem accepts these arguments */

/* Weight layout conversion. master f32 [Cout][Cin][KH][KW] (the reference state_dict layout,
 * model.py:15) -> packed K-contiguous rows in compute dtype.
 *   mode 0: forward rows  dst[co][(ta*tw+tb)*Cin + ci]   (kh = kh0+ta*khs, kw = kw0+tb*kws)
 *   mode 1: dgrad rows    dst[ci][(ta*tw+tb)*Cout + co]
 *   mode 2: stem 6x6/s2 on 3 channels as 3x3/s1 on the 2x2 space-to-depth input (12 -> 16 channels) */
int y5m_pack_weights(const float* src, int Cout, int Cin, int KH, int KW, int mode, int kh0, int khs,
                     int th, int kw0, int kws, int tw, void* dst, int rows_p, int Kp, int cstride,
                     int dtype, void* stream);   /* cstride: per-tap channel stride (0 = dense) */
/* the same for a whole step in ONE launch: a DEVICE table of jobs (start = running element offset,
 * job j covers [start_j, start_j + rows_p*Kp)); total = sum of rows_p*Kp */
typedef struct {
    const float* src;
    void* dst;
    int32_t Cout, Cin, KH, KW, mode, kh0, khs, th, kw0, kws, tw, rows_p, Kp, cstride;
    int32_t ldd;         /* dst row pitch in elements (0 = Kp): a job may fill a column window of wider rows */
    int32_t pad_;
    int64_t start;
} y5m_pack_job;
int y5m_pack_weights_batched(const y5m_pack_job* d_jobs, int njobs, int64_t total, int dtype, void* stream);
/* packed f32 gradient [Cout][ldg] (mode 0 or 2 ordering) -> [Cout][Cin][KH][KW] */
int y5m_unpack_wgrad(const float* gp, int Cout, int Cin, int KH, int KW, int mode, int ldg, float* dst,
                     void* stream);
/* Input stage on the device (utils/training_utils.py:98 `images.float()/255` and :11-28 multi_scale):
 * uint8 (B,3,Hs,Ws) -> f32 (B,3,H,W) = F.interpolate(img/255, (H,W), "bilinear", align_corners=False);
 * Hs==H && Ws==W is the plain /255 conversion. */
int y5m_preprocess_u8(const unsigned char* img, int B, int Hs, int Ws, float* out, int H, int W, void* stream);
/* images (B,3,H,W) f32 NCHW (model.py:210 input) -> (B,H/2,W/2,16) NHWC, ch=(dy*2+dx)*3+c */
int y5m_s2d_input(const float* img, int B, int H, int W, void* out, int dtype, void* stream);

/* BatchNorm2d(eps=1e-3, momentum=0.03) of CBL (model.py:17), training mode: finalise the batch
 * statistics from the conv epilogue partials, update running stats (unbiased var), emit the fused
 * scale/shift and the saved mean / invstd for backward. */
size_t y5m_bn_finalize_workspace_bytes(int Np);
int y5m_bn_finalize(const float* stats, int tiles_m, int Np, int C, int64_t count, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                    float* scale, float* shift, float* mean_out, float* invstd_out, int update_running,
                    void* ws, size_t ws_bytes, void* stream);
/* the same for every layer of a model in ONE launch: a DEVICE table of jobs, start = running channel offset */
typedef struct {
    const float* gamma; const float* beta; const float* running_mean; const float* running_var;
    float* scale; float* shift;
    int32_t C, start;
} y5m_fold_job;
int y5m_bn_fold_batched(const y5m_fold_job* d_jobs, int njobs, int total_channels, float eps, void* stream);
/* eval mode: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale */
int y5m_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                float eps, int C, float* scale, float* shift, void* stream);
/* out = act(y*scale+shift) (+ res): BN normalise + nn.SiLU (model.py:20) + Bottleneck add (:50) */
int y5m_bn_act(const void* y, int ldy, const float* scale, const float* shift, const void* res, int ldres,
               void* out, int ldout, int64_t M, int C, int act, int dtype, void* stream);
/* Training-mode BatchNorm2d(eps, momentum) (reference model.py:17) + activation + residual straight from the accumulator
 * rows a y5m_conv launch with bn_acc filled (see y5m_bn_acc_slots): batch statistics, normalise, act in ONE launch, no
 * y5m_bn_finalize. acc points at this layer's first channel inside rows of ldacc doubles ([slots][2][ldacc]); count =
 * samples per channel. Writes scale / shift / mean / invstd [C] (the backward pass reads them) and, with update_running,
 * the running statistics (unbiased variance, as nn.BatchNorm2d). */
int y5m_bn_act_fused(const void* y, int ldy, const double* acc, int ldacc, int64_t count, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                     int update_running, float* scale, float* shift, float* mean_out, float* invstd_out,
                     const void* res, int ldres, void* out, int ldout, int64_t M, int C, int act, int dtype, void* stream);
/* autograd of the above: dgamma, dbeta (param grads) and dy given dz */
size_t y5m_bn_bwd_workspace_bytes(int64_t M, int C);
/* ... in two launches instead of three: acc [y5m_bn_acc_slots()][2][C] f64, zeroed by the caller, left dirty */
int y5m_bn_bwd_fused(const void* dz, int lddz, const void* y, int ldy, const float* scale, const float* shift,
                     const float* mean, const float* invstd, int64_t M, int C, int act, float* dgamma, float* dbeta,
                     int accumulate_param_grads, void* dy, int lddy, double* acc, int dtype, void* stream);
/* ... one launch at a time (phase 1 = the reduce pass, 2 = the apply pass, 3 = both). Phase 1 alone is what runs ahead of
 * y5m_bwd_pw, which forms dy itself: dgamma / dbeta / dy may then be NULL */
int y5m_bn_bwd_fused_phase(const void* dz, int lddz, const void* y, int ldy, const float* scale, const float* shift,
                           const float* mean, const float* invstd, int64_t M, int C, int act, float* dgamma, float* dbeta,
                           int accumulate_param_grads, void* dy, int lddy, double* acc, int dtype, void* stream, int phase);
int y5m_bn_bwd(const void* dz, int lddz, const void* y, int ldy, const float* scale, const float* shift,
               const float* mean, const float* invstd, int64_t M, int C, int act, float* dgamma,
               float* dbeta, int accumulate_param_grads, void* dy, int lddy, void* ws, size_t ws_bytes,
               int dtype, void* stream);
/* dst (+)= src on (ptr, ld) views: residual / concat gradient plumbing */
int y5m_add(const void* src, int ldsrc, void* dst, int lddst, int64_t M, int C, int accumulate, int dtype,
            void* stream);
/* nearest x2 (model.py:225) and its backward */
int y5m_upsample2x(const void* in, int ldin, int B, int H, int W, int C, void* out, int ldout, int dtype,
                   void* stream);
int y5m_upsample2x_bwd(const void* gout, int ldg, int B, int H, int W, int C, void* gin, int ldgin,
                       int accumulate, int dtype, void* stream);
/* SPPF (model.py:103-112): the three cascaded MaxPool2d(5,1,2) in one launch; backward per level */
size_t y5m_sppf_pool_workspace_bytes(int B, int H, int W, int C);
int y5m_sppf_pool(const void* x, int ld, int B, int H, int W, int C, void* o1, void* o2, void* o3, void* ws,
                  size_t ws_bytes, int dtype, void* stream);
size_t y5m_maxpool5_bwd_workspace_bytes(int B, int H, int W, int C);
int y5m_maxpool5_bwd(const void* z, int ldz, const void* g, int ldg, int B, int H, int W, int C, void* gin,
                     int ldgin, int accumulate, void* ws, size_t ws_bytes, int dtype, void* stream);
/* the whole backward cascade of the SPPF pools (model.py:108-110; autograd of the three MaxPool2d(5,1,2)): with z0 = x,
 * z1 = pool(z0), z2 = pool(z1) and g0..g3 the gradients of the four concat slices: g2 += bwd(z2; g3), g1 += bwd(z1; g2),
 * g0 += bwd(z0; g1). One launch where the LDS-tiled form applies (y5m_sppf_pool_tiled), else three y5m_maxpool5_bwd calls;
 * ws as for y5m_maxpool5_bwd. */
int y5m_sppf_pool_bwd(const void* z0, const void* z1, const void* z2, int ldz, void* g0, void* g1, void* g2,
                      const void* g3, int ldg, int B, int H, int W, int C, void* ws, size_t ws_bytes, int dtype,
                      void* stream);
/* 1 when y5m_sppf_pool / y5m_sppf_pool_bwd run their LDS-tiled forms for this shape (Y5M_POOL_TILE and an image x one
 * 8-channel piece fits the LDS budget) */
int y5m_sppf_pool_tiled(int H, int W, int C, int dtype);
/* d(loss)/d(logits) (B,naxs,ny,nx,nch) f32 -> head conv output gradient [B*ny*nx][ldp] + bias grad */
int y5m_head_grad_pack(const float* dlogits, int B, int naxs, int ny, int nx, int nch, void* dyp, int ldp,
                       float* dbias, int dtype, void* stream);
/* The same result for a gradient written by y5m_compute_loss / y5m_compute_loss_sparse (ultralytics_loss.py:60-120):
 * that tensor is zero outside channel 4 (objectness) of every cell and the rows of the cells a target row hit. `owner`
 * and `gobj` are the loss workspace's per-cell tables of that scale (y5m_compute_loss_owner_ptrs), `bagg` / `count` the
 * scale's target rows (y5m_targets, cap = their capacity): only the objectness plane and the hit rows of dlogits are
 * read instead of all 5+nc floats of every cell. naxs == 3, ldp % 4 == 0. */
int y5m_head_grad_pack_sparse(const float* dlogits, const int32_t* owner, const float* gobj, const int32_t* bagg,
                              const int32_t* count, int cap, int B, int naxs, int ny, int nx, int nch, void* dyp, int ldp,
                              float* dbias, int dtype, void* stream);
/* optimizer (train.py:61 Adam(lr, weight_decay) + training_utils.py:118 clip_grad_norm_(10)) */
size_t y5m_adam_workspace_bytes(void);
int y5m_grad_norm(const float* g, int64_t n, float* norm_out, void* ws, size_t ws_bytes, void* stream);
int y5m_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* gnorm, float max_norm,
                  double lr, double beta1, double beta2, double eps, double weight_decay, const int32_t* d_step,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* Y5M_H */
