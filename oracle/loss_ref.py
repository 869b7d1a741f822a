"""Torch/numpy CPU restatement of the reference losses, GIoU and decode -- TEST INFRASTRUCTURE ONLY.

Each function cites the reference lines it follows. Index/target assignment is delegated to the C
restatement (oracle/csrc/y5m_oracle.c) so integers and fp32 bits are produced by plain IEEE C.
Pinned against the imported reference by tests/golden/make_golden.py (G1-G4).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import cnative

ANCHOR_T = 4.0                      # ultralytics_loss.py:35
BALANCE = (4.0, 1.0, 0.4)           # ultralytics_loss.py:37 / loss.py:37


def giou(b1, b2, GIoU=True, eps=1e-7):
    """reference utils/bboxes_utils.py:33-87 (midpoint format). (...,4),(...,4) -> (...,1)."""
    b1x1 = b1[..., 0:1] - b1[..., 2:3] / 2
    b1y1 = b1[..., 1:2] - b1[..., 3:4] / 2
    b1x2 = b1[..., 0:1] + b1[..., 2:3] / 2
    b1y2 = b1[..., 1:2] + b1[..., 3:4] / 2
    b2x1 = b2[..., 0:1] - b2[..., 2:3] / 2
    b2y1 = b2[..., 1:2] - b2[..., 3:4] / 2
    b2x2 = b2[..., 0:1] + b2[..., 2:3] / 2
    b2y2 = b2[..., 1:2] + b2[..., 3:4] / 2
    w1, h1, w2, h2 = b1x2 - b1x1, b1y2 - b1y1, b2x2 - b2x1, b2y2 - b2y1
    inter = (torch.min(b1x2, b2x2) - torch.max(b1x1, b2x1)).clamp(0) * \
            (torch.min(b1y2, b2y2) - torch.max(b1y1, b2y1)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if GIoU:
        cw = torch.max(b1x2, b2x2) - torch.min(b1x1, b2x1)
        ch = torch.max(b1y2, b2y2) - torch.min(b1y1, b2y1)
        c_area = cw * ch + eps
        return iou - (c_area - union) / c_area
    return iou


def lambdas(nc, nl=3, image_size=640):
    """reference ultralytics_loss.py:31-33 / loss.py:32-34."""
    return 0.05 * (3 / nl), 1 * ((image_size / 640) ** 2 * 3 / nl), 0.5 * (nc / 80 * 3 / nl)


def build_targets_ultra(shapes, targets, anchors):
    """reference ultralytics_loss.py:122-311. shapes: list of (ny,nx); targets (nt,6) fp32;
    anchors (nl,na,2) stride-divided. Returns per-scale dicts (see cnative)."""
    t = np.asarray(targets, dtype=np.float32).reshape(-1, 6)
    a = np.asarray(anchors, dtype=np.float32)
    return [cnative.build_targets_ultra_scale(t, a[i], ny, nx, ANCHOR_T)
            for i, (ny, nx) in enumerate(shapes)]


def compute_loss_ultra(p, targets, anchors, nc=80):
    """reference ultralytics_loss.py:60-120. p: list of 3 (B,3,ny,nx,5+nc) torch fp32 (autograd ok);
    targets (nt,6); anchors (3,3,2). Returns (loss(1,), (lbox,lobj,lcls) unscaled-by-bs parts)."""
    lam_box, lam_obj, lam_cls = lambdas(nc, len(p))
    shapes = [(pi.shape[2], pi.shape[3]) for pi in p]
    tgt = build_targets_ultra(shapes, targets.detach().cpu().numpy() if torch.is_tensor(targets)
                              else targets, np.asarray(anchors))
    lcls = torch.zeros(1); lbox = torch.zeros(1); lobj = torch.zeros(1)
    for i, pi in enumerate(p):
        d = tgt[i]
        b, a = torch.from_numpy(d["b"]), torch.from_numpy(d["a"])
        gj, gi = torch.from_numpy(d["gj"]), torch.from_numpy(d["gi"])
        tobj = torch.zeros(pi.shape[:4], dtype=pi.dtype)
        n = b.shape[0]
        if n:
            ps = pi[b, a, gj, gi]
            pxy, pwh, _, pcls = ps.split((2, 2, 1, nc), 1)
            pxy = pxy.sigmoid() * 2 - 0.5
            pwh = (pwh.sigmoid() * 2) ** 2 * torch.from_numpy(d["anch"])
            pbox = torch.cat((pxy, pwh), 1)
            iou = giou(pbox, torch.from_numpy(d["tbox"]), GIoU=True).squeeze(-1)
            lbox = lbox + (1.0 - iou).mean()
            iou_d = iou.detach().clamp(0).type(tobj.dtype)
            # non-accumulating index_put (:89). For duplicate cells the reference's outcome is the LAST
            # row whenever ATen runs index_put_ serially (n below its parallel grain, or 1 thread:
            # measured); with several threads and n >~ 1000 ATen races and the winner is arbitrary.
            # The oracle pins the deterministic serial semantics: last row wins.
            cell = ((d["b"] * pi.shape[1] + d["a"]) * pi.shape[2] + d["gj"]) * pi.shape[3] + d["gi"]
            _, first_rev = np.unique(cell[::-1], return_index=True)
            last = torch.from_numpy((n - 1 - first_rev).astype(np.int64))
            tobj.view(-1)[torch.from_numpy(cell[last.numpy()].astype(np.int64))] = iou_d[last]
            if nc > 1:
                t = torch.zeros_like(pcls)
                t[range(n), torch.from_numpy(d["tcls"])] = 1
                lcls = lcls + F.binary_cross_entropy_with_logits(pcls, t)
        obji = F.binary_cross_entropy_with_logits(pi[..., 4], tobj)
        lobj = lobj + obji * BALANCE[i]
    lbox = lbox * lam_box
    lobj = lobj * lam_obj
    lcls = lcls * lam_cls
    bs = p[-1].shape[0]
    return (lbox + lobj + lcls) * bs, (lbox, lobj, lcls)


# ----------------------------------------------------------------------------------------------
# YOLO_LOSS (reference loss.py) -- dense targets
# ----------------------------------------------------------------------------------------------
class YoloLossRef:
    """reference loss.py:20-246 restated, including the in-place anchor decay of
    utils/bboxes_utils.py:18 (`anchors /= 640` on the caller's tensor on EVERY call).
    State = self.anchors (CPU fp32 (3,3,2)), exactly like YOLO_LOSS.anchors (loss.py:40)."""

    def __init__(self, anchors, nc=80, strides=(8, 16, 32)):
        self.anchors_d = torch.as_tensor(anchors, dtype=torch.float32).clone()
        self.anchors = self.anchors_d.clone()
        self.nc = nc
        self.S = list(strides)
        self.ignore_iou_thresh = 0.5
        self.lam_box, self.lam_obj, self.lam_cls = lambdas(nc, 3)

    def iou_width_height(self, gt_wh):
        """reference utils/bboxes_utils.py:6-29 (strided_anchors=True). gt_wh: float64 tensor (2,)."""
        self.anchors /= 640                                   # :18 in place, stateful
        anc = self.anchors.reshape(9, 2) * torch.tensor(self.S).repeat(6, 1).T.reshape(9, 2)
        inter = torch.min(gt_wh[..., 0], anc[..., 0]) * torch.min(gt_wh[..., 1], anc[..., 1])
        union = gt_wh[..., 0] * gt_wh[..., 1] + anc[..., 0] * anc[..., 1] - inter
        return inter / union

    def build_targets(self, shapes, bboxes):
        """reference loss.py:101-192 for ONE image. shapes: [(ny,nx)]*3; bboxes ndarray (n,5)
        [cls,x,y,w,h] float64."""
        targets = [torch.zeros((3, ny, nx, 6)) for (ny, nx) in shapes]
        classes = bboxes[:, 0].tolist() if len(bboxes) else []
        boxes = bboxes[:, 1:] if len(bboxes) else []
        for idx, box in enumerate(boxes):
            iou_anchors = self.iou_width_height(torch.from_numpy(box[2:4]))
            anchor_indices = iou_anchors.argsort(descending=True, dim=0)
            x, y, width, height = box
            has_anchor = [False] * 3
            for anchor_idx in anchor_indices:
                scale_idx = int(anchor_idx) // 3
                anchor_on_scale = int(anchor_idx) % 3
                scale_y, scale_x = shapes[scale_idx]
                i, j = int(scale_y * y), int(scale_x * x)
                anchor_taken = targets[scale_idx][anchor_on_scale, i, j, 4]
                if not anchor_taken and not has_anchor[scale_idx]:
                    targets[scale_idx][anchor_on_scale, i, j, 4] = 1
                    x_cell, y_cell = scale_x * x - j, scale_y * y - i
                    width_cell, height_cell = width * scale_x, height * scale_y
                    targets[scale_idx][anchor_on_scale, i, j, 0:4] = torch.tensor(
                        [x_cell, y_cell, width_cell, height_cell])
                    targets[scale_idx][anchor_on_scale, i, j, 5] = int(classes[idx])
                    has_anchor[scale_idx] = True
                elif not anchor_taken and iou_anchors[anchor_idx] > self.ignore_iou_thresh:
                    targets[scale_idx][anchor_on_scale, i, j, 4] = -1
        return targets

    def compute_loss(self, preds, targets, anchors, balance):
        """reference loss.py:195-246 (save_logs=False). targets is MUTATED in place (:218)."""
        bs = preds.shape[0]
        anchors = anchors.reshape(1, 3, 1, 1, 2)
        obj = targets[..., 4] == 1
        pxy = (preds[..., 0:2].sigmoid() * 2) - 0.5
        pwh = ((preds[..., 2:4].sigmoid() * 2) ** 2) * anchors
        pbox = torch.cat((pxy[obj], pwh[obj]), dim=-1)
        tbox = targets[..., 0:4][obj]
        iou = giou(pbox, tbox, GIoU=True).squeeze(-1)
        lbox = (1.0 - iou).mean()
        iou = iou.detach().clamp(0)
        targets[..., 4][obj] *= iou
        lobj = F.binary_cross_entropy_with_logits(preds[..., 4], targets[..., 4]) * balance
        tcls = torch.zeros_like(preds[..., 5:][obj])
        tcls[torch.arange(tcls.size(0)), targets[..., 5][obj].long()] = 1.0
        lcls = F.binary_cross_entropy_with_logits(preds[..., 5:][obj], tcls)
        return (self.lam_box * lbox + self.lam_obj * lobj + self.lam_cls * lcls) * bs

    def __call__(self, preds, targets_np):
        """reference loss.py:64-99."""
        shapes = [(p.shape[2], p.shape[3]) for p in preds]
        tg = [self.build_targets(shapes, b) for b in targets_np]
        ts = [torch.stack([t[i] for t in tg], 0) for i in range(3)]
        return sum(self.compute_loss(preds[i], ts[i], self.anchors_d[i], BALANCE[i]) for i in range(3))


# ----------------------------------------------------------------------------------------------
# decode (reference utils/plot_utils.py:10-54)
# ----------------------------------------------------------------------------------------------
def make_grids(anchors, naxs, stride, nx=20, ny=20, i=0):
    """reference utils/plot_utils.py:42-54."""
    xg = torch.arange(nx).repeat(ny).reshape(ny, nx)
    yg = torch.arange(ny).unsqueeze(0).T.repeat(1, nx).reshape(ny, nx)
    xy = torch.stack([xg, yg], dim=-1).expand(1, naxs, ny, nx, 2)
    ag = (anchors[i] * stride).reshape((1, naxs, 1, 1, 2)).expand(1, naxs, ny, nx, 2)
    return xy, ag


def cells_to_bboxes(predictions, anchors, strides, is_pred=False):
    """reference utils/plot_utils.py:10-40 (to_list=False). Returns (B,N,6) [cls,obj,x,y,w,h]."""
    out = []
    for i, pr in enumerate(predictions):
        bs, naxs, ny, nx, _ = pr.shape
        stride = strides[i]
        grid, ag = make_grids(anchors, naxs, ny=ny, nx=nx, stride=stride, i=i)
        if is_pred:
            lp = pr.sigmoid()
            obj = lp[..., 4:5]
            xy = (2 * lp[..., 0:2] + grid - 0.5) * stride
            wh = ((2 * lp[..., 2:4]) ** 2) * ag
            best = torch.argmax(lp[..., 5:], dim=-1).unsqueeze(-1)
        else:
            obj = pr[..., 4:5]
            xy = (pr[..., 0:2] + grid) * stride
            wh = pr[..., 2:4] * stride
            best = pr[..., 5:6]
        out.append(torch.cat((best, obj, xy, wh), dim=-1).reshape(bs, -1, 6))
    return torch.cat(out, dim=1)


def non_max_suppression(batch_bboxes, iou_threshold, threshold, max_detections=300):
    """reference utils/bboxes_utils.py:175-209 (tolist=True semantics, as arrays).
    Returns list over images of (rows (k,6) ndarray, src_idx (k,) ndarray)."""
    bb = batch_bboxes.detach().cpu().numpy() if torch.is_tensor(batch_bboxes) else np.asarray(batch_bboxes)
    return [cnative.non_max_suppression_image(bb[i], iou_threshold, threshold, max_detections)
            for i in range(bb.shape[0])]
