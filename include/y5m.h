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

/* Replaces utils/bboxes_utils.py:175-209 non_max_suppression (per-image loop :185-203 incl. the
 * torchvision.ops.nms call :195) for a whole batch in ONE launch (one workgroup per image).
 * boxes (B,N,6) rows [class, score, x, y, w, h] f32 (not modified).
 * out_rows (B,max_det,6) rows [class, score, x1, y1, x2, y2]; out_idx (B,max_det) source row in
 * [0,N); out_count (B). max_det <= 1024. Index sets are bit-exact vs the CPU reference path. */
size_t y5m_nms_workspace_bytes(int B, int64_t N);
int y5m_nms(const float* boxes, int B, int64_t N, float conf_threshold, double iou_threshold,
            int max_det, float* out_rows, int32_t* out_idx, int32_t* out_count, void* ws,
            size_t ws_bytes, void* stream);

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
int y5m_compute_loss(const float* const p[3], float* const grad[3], int B, int naxs, const int* ny,
                     const int* nx, int nc, const y5m_targets tg[3], int nt_max,
                     const float balance[3], float lambda_box, float lambda_obj, float lambda_cls,
                     float* loss_out, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* Y5M_H */
