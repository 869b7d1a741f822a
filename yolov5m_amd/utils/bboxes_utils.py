"""IoU and NMS -- same signatures as the reference's utils/bboxes_utils.py, HIP underneath.

Mirrored: iou_width_height (:6-29), intersection_over_union (:33-87), non_max_suppression (:175-209).
The box-format helpers (coco_to_yolo*, rescale_bboxes) are data preparation and out of scope.
"""
import torch

from .. import _lib


def iou_width_height(gt_box, anchors, strided_anchors=True, stride=[8, 16, 32]):
    """reference utils/bboxes_utils.py:6-29, INCLUDING its in-place `anchors /= 640` on the caller's
    tensor (:18; SURVEY C.1 -- behaviour kept, not fixed). 9 anchors x 2 floats of host-side target
    assignment: tensor arithmetic on whatever device the caller's anchors live (the reference calls it
    with CPU tensors from YOLO_LOSS.build_targets)."""
    anchors /= 640
    if strided_anchors:
        anchors = anchors.reshape(9, 2) * torch.tensor(stride, device=anchors.device).repeat(6, 1).T.reshape(9, 2)
    intersection = torch.min(gt_box[..., 0], anchors[..., 0]) * torch.min(gt_box[..., 1], anchors[..., 1])
    union = gt_box[..., 0] * gt_box[..., 1] + anchors[..., 0] * anchors[..., 1] - intersection
    return intersection / union


def _corners_to_mid(b):
    # the kernel is written for midpoint boxes; corner boxes are fed as (cx, cy, w, h) so that
    # c -/+ w/2 reproduces x1/x2 up to fp32 rounding.
    x1, y1, x2, y2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], -1)


class _IoUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, giou, eps):
        L = _lib.lib()
        n = a.shape[0]
        out = torch.empty((n,), dtype=torch.float32, device=a.device)
        _lib.check(L.y5m_iou(_lib.ptr(a), _lib.ptr(b), n, int(giou), float(eps), _lib.ptr(out),
                             _lib.stream_ptr()), "y5m_iou")
        ctx.save_for_backward(a, b)
        ctx.giou, ctx.eps = giou, eps
        return out

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        L = _lib.lib()
        n = a.shape[0]
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        g = gout.contiguous().float()
        _lib.check(L.y5m_iou_bwd(_lib.ptr(a), _lib.ptr(b), _lib.ptr(g), n, int(ctx.giou), float(ctx.eps),
                                 _lib.ptr(ga), _lib.ptr(gb), _lib.stream_ptr()), "y5m_iou_bwd")
        return ga, gb, None, None


def intersection_over_union(boxes_preds, boxes_labels, box_format="midpoint", GIoU=False, eps=1e-7):
    """reference utils/bboxes_utils.py:33-87. (...,4),(...,4) -> (...,1); differentiable."""
    _lib.require_cuda(boxes_preds, boxes_labels)
    shape = torch.broadcast_shapes(boxes_preds.shape, boxes_labels.shape)
    a = boxes_preds.float().expand(shape)
    b = boxes_labels.float().expand(shape)
    if box_format != "midpoint":
        a, b = _corners_to_mid(a), _corners_to_mid(b)
    a = a.reshape(-1, 4).contiguous()
    b = b.reshape(-1, 4).contiguous()
    out = _IoUFn.apply(a, b, bool(GIoU), float(eps))
    return out.reshape(*shape[:-1], 1)


def nms_batched(batch_bboxes, iou_threshold, threshold, max_detections=300):
    """Device-resident result of the batched NMS kernel: (rows (B,max_det,6), idx (B,max_det) int32,
    count (B,) int32). No host synchronisation."""
    L = _lib.lib()
    _lib.require_cuda(batch_bboxes)
    bb = batch_bboxes
    if bb.dtype != torch.float32 or not bb.is_contiguous():
        bb = bb.contiguous().float()
    B, N = bb.shape[0], bb.shape[1]
    dev = bb.device
    rows = torch.zeros((B, max_detections, 6), dtype=torch.float32, device=dev)
    idx = torch.zeros((B, max_detections), dtype=torch.int32, device=dev)
    cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    wsb = L.y5m_nms_workspace_bytes(B, N)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    rc = L.y5m_nms(_lib.ptr(bb), B, N, float(threshold), float(iou_threshold), int(max_detections),
                   _lib.ptr(rows), _lib.ptr(idx), _lib.ptr(cnt), _lib.ptr(ws), wsb, _lib.stream_ptr())
    _lib.check(rc, "y5m_nms")
    return rows, idx, cnt


def non_max_suppression_aladdin(bboxes, iou_threshold, threshold, box_format="corners", max_detections=300):
    """reference utils/bboxes_utils.py:129-173 (its own slow pure-Python baseline), same signature: a list of
    [class_pred, prob_score, x1, y1, x2, y2] (or x, y, w, h with box_format="midpoint") -> the kept boxes (the
    SAME list objects, keep order). Filter `score > threshold`, stable descending sort, truncation to
    max_detections BEFORE suppression, same-class suppression at IoU >= iou_threshold -- one native launch
    (y5m_nms_aladdin). Scores / coordinates are compared as float32 (the reference builds float32 tensors for the
    IoU; its score filter and sort run on the Python floats, identical whenever those came from a float32 tensor)."""
    assert type(bboxes) == list
    if not bboxes:
        return []
    L = _lib.lib()
    t = torch.tensor(bboxes, dtype=torch.float32, device=_lib.device()).reshape(1, -1, 6).contiguous()
    N = t.shape[1]
    rows = torch.empty((1, max_detections, 6), dtype=torch.float32, device=t.device)
    idx = torch.empty((1, max_detections), dtype=torch.int32, device=t.device)
    cnt = torch.zeros((1,), dtype=torch.int32, device=t.device)
    wsb = L.y5m_nms_workspace_bytes(1, N)
    ws = torch.empty(wsb, dtype=torch.uint8, device=t.device)
    _lib.check(L.y5m_nms_aladdin(_lib.ptr(t), 1, N, float(threshold), float(iou_threshold),
                                 1 if box_format == "midpoint" else 0, int(max_detections), _lib.ptr(rows), _lib.ptr(idx),
                                 _lib.ptr(cnt), _lib.ptr(ws), wsb, _lib.stream_ptr()), "y5m_nms_aladdin")
    k = int(cnt.item())
    return [bboxes[i] for i in idx[0, :k].tolist()]


def non_max_suppression(batch_bboxes, iou_threshold, threshold, max_detections=300, tolist=True):
    """reference utils/bboxes_utils.py:175-209. batch_bboxes (B,N,6) rows [class, score, x, y, w, h].
    tolist=True -> list (per image) of lists of [class, score, x1, y1, x2, y2];
    tolist=False -> ONE concatenated tensor for the whole batch (image boundaries lost, as in the
    reference :209)."""
    if not torch.is_tensor(batch_bboxes):
        batch_bboxes = torch.as_tensor(batch_bboxes, dtype=torch.float32, device=_lib.device())
    rows, _, cnt = nms_batched(batch_bboxes, iou_threshold, threshold, max_detections)
    counts = cnt.tolist()        # the only host sync of the detect path
    per_img = [rows[b, :c] for b, c in enumerate(counts)]
    if tolist:
        return [r.tolist() for r in per_img]
    return torch.cat(per_img, dim=0) if per_img else rows.reshape(0, 6)
