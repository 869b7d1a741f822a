"""YOLO_LOSS -- same signature as the reference's loss.py:20-246.

Split exactly as in the reference:
  * build_targets (loss.py:101-192) is HOST code there (numpy float64 boxes, Python loops, CPU
    tensors, then `.to(device)` at :70-74) and stays host code here, including the reference's stateful
    defect: utils/bboxes_utils.py:18 divides `self.anchors` by 640 IN PLACE on every call,
    so only the first box ever sees the real anchors. Parity is defined on that behaviour.
  * compute_loss (loss.py:195-246) for the 3 scales and its autograd backward run on the MI355X in the
    dense-target variant of the native loss kernels (y5m_compute_loss_dense).
"""
import csv
import os

import numpy as np
import torch

from . import _lib, config
from .utils.bboxes_utils import iou_width_height


class _DenseLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, dense, rows_max, p0, p1, p2):
        L = _lib.lib()
        p = (p0, p1, p2)
        B, naxs = p0.shape[0], p0.shape[1]
        ny = _lib.int_array([t.shape[2] for t in p])
        nx = _lib.int_array([t.shape[3] for t in p])
        need_grad = any(ctx.needs_input_grad[3:6])
        grads = [torch.empty_like(t) for t in p] if need_grad else [None, None, None]
        wsb = L.y5m_compute_loss_dense_workspace_bytes(B, naxs, ny, nx, rows_max)
        ws = torch.empty(wsb, dtype=torch.uint8, device=p0.device)
        out = torch.zeros(4, dtype=torch.float32, device=p0.device)
        _lib.check(L.y5m_compute_loss_dense(_lib.ptr_array(p), _lib.ptr_array(grads), _lib.ptr_array(dense), B, naxs,
                                            ny, nx, owner.nc, _lib.ptr(owner.anchors_d), rows_max,
                                            _lib.float_array(owner.balance), float(owner.lambda_box),
                                            float(owner.lambda_obj), float(owner.lambda_class), _lib.ptr(out),
                                            _lib.ptr(ws), wsb, _lib.stream_ptr()), "y5m_compute_loss_dense")
        ctx.grads = grads
        owner.last_components = out[1:4]
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        g = ctx.grads
        if g[0] is None:
            return None, None, None, None, None, None
        return None, None, None, g[0] * gout, g[1] * gout, g[2] * gout


class YOLO_LOSS:
    """reference loss.py:20-99"""

    def __init__(self, model, rect_training, save_logs=False, filename=None, resume=False):
        self.rect_training = rect_training
        self.lambda_class = 0.5 * (model.head.nc / 80 * 3 / model.head.nl)            # :32
        self.lambda_obj = 1 * ((config.IMAGE_SIZE / 640) ** 2 * 3 / model.head.nl)    # :33
        self.lambda_box = 0.05 * (3 / model.head.nl)                                  # :34
        self.balance = [4.0, 1.0, 0.4]                                                # :36
        self.nc = model.head.nc
        self.anchors_d = model.head.anchors.clone().detach().contiguous()             # :39
        self.anchors = model.head.anchors.clone().detach().to("cpu")                  # :40 (decays in place!)
        self.na = self.anchors.reshape(9, 2).shape[0]
        self.num_anchors_per_scale = self.na // 3
        self.S = model.head.stride
        self.ignore_iou_thresh = 0.5
        self.save_logs = save_logs
        self.filename = filename
        self.last_components = None
        if not self.anchors_d.is_cuda:
            raise _lib.Y5MError("YOLO_LOSS: model must live on the GPU (no CPU fallback)")
        if self.save_logs and not resume:                                             # :51-62
            folder = os.path.join("train_eval_metrics", filename)
            os.makedirs(folder, exist_ok=True)
            with open(os.path.join(folder, "loss.csv"), "w") as f:
                csv.writer(f).writerow(["epoch", "batch_idx", "box_loss", "object_loss", "class_loss"])

    def __call__(self, preds, targets, pred_size, batch_idx=None, epoch=None):
        """reference loss.py:64-99. preds: 3 logits tensors; targets: tuple of per-image ndarrays (n_i,5)
        [cls, x, y, w, h] (reference collate_fn, dataset.py:199-202)."""
        self.batch_idx, self.epoch = batch_idx, epoch
        tg = [self.build_targets(preds, bboxes, pred_size) for bboxes in targets]                    # :68
        dev = self.anchors_d.device
        dense = [torch.stack([t[i] for t in tg], dim=0).to(dev, non_blocking=True).contiguous() for i in range(3)]
        rows_max = max(1, sum(len(b) for b in targets))
        p = [t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous() for t in preds]
        _lib.require_cuda(*p)
        loss = _DenseLossFn.apply(self, dense, rows_max, p[0], p[1], p[2])
        if self.save_logs and batch_idx is not None and batch_idx % 100 == 0:                         # :82-90
            lb, lo, lc = self.last_components.tolist()
            with open(os.path.join("train_eval_metrics", self.filename, "loss.csv"), "a") as f:
                csv.writer(f).writerow([epoch, batch_idx, lb, lo, lc])
        return loss

    def build_targets(self, input_tensor, bboxes, pred_size):
        """reference loss.py:101-192 for ONE image -> list of 3 CPU tensors (3, ny, nx, 6)."""
        shapes = [(input_tensor[i].shape[2], input_tensor[i].shape[3]) for i in range(len(self.S))]
        targets = [torch.zeros((self.num_anchors_per_scale, ny, nx, 6)) for (ny, nx) in shapes]
        classes = bboxes[:, 0].tolist() if len(bboxes) else []
        boxes = bboxes[:, 1:] if len(bboxes) else []
        for idx, box in enumerate(boxes):
            iou_anchors = iou_width_height(torch.from_numpy(np.asarray(box[2:4])), self.anchors)     # :120 (in-place decay)
            anchor_indices = iou_anchors.argsort(descending=True, dim=0)                             # :122
            x, y, width, height = box
            has_anchor = [False] * 3
            for anchor_idx in anchor_indices:
                scale_idx = int(torch.div(anchor_idx, self.num_anchors_per_scale, rounding_mode="floor"))
                anchor_on_scale = int(anchor_idx % self.num_anchors_per_scale)
                scale_y, scale_x = shapes[scale_idx]
                i, j = int(scale_y * y), int(scale_x * x)                                            # :152
                anchor_taken = targets[scale_idx][anchor_on_scale, i, j, 4]
                if not anchor_taken and not has_anchor[scale_idx]:
                    targets[scale_idx][anchor_on_scale, i, j, 4] = 1
                    x_cell, y_cell = scale_x * x - j, scale_y * y - i
                    width_cell, height_cell = width * scale_x, height * scale_y
                    targets[scale_idx][anchor_on_scale, i, j, 0:4] = torch.tensor([x_cell, y_cell, width_cell, height_cell])
                    targets[scale_idx][anchor_on_scale, i, j, 5] = int(classes[idx])
                    has_anchor[scale_idx] = True
                elif not anchor_taken and iou_anchors[anchor_idx] > self.ignore_iou_thresh:
                    targets[scale_idx][anchor_on_scale, i, j, 4] = -1                                # :190 ignore
        return targets

    def compute_loss(self, preds, targets, anchors, balance):
        """reference loss.py:195-246 for ONE scale -> (loss, logs or None): the same native kernels with
        the other two scales absent. `anchors` selects the scale (must be one of self.anchors_d[i])."""
        L = _lib.lib()
        i = [k for k in range(3) if torch.equal(anchors.reshape(-1, 2).to(self.anchors_d.device), self.anchors_d[k])]
        if not i:
            raise _lib.Y5MError("compute_loss: anchors must be one of the model's per-scale anchor sets")
        i = i[0]
        p = preds if (preds.dtype == torch.float32 and preds.is_contiguous()) else preds.float().contiguous()
        dense = targets.to(p.device).float().contiguous()
        plist, dlist = [None] * 3, [None] * 3
        plist[i], dlist[i] = p, dense
        bal = [0.0, 0.0, 0.0]
        bal[i] = float(balance)
        saved, self.balance = self.balance, bal
        try:
            rows_max = max(1, int((dense[..., 4] == 1).sum()))
            loss = _DenseLossFnSingle.apply(self, dlist, rows_max, i, p)
        finally:
            self.balance = saved
        logs = self.last_components.reshape(1, 3) if self.save_logs else None
        return loss, logs


class _DenseLossFnSingle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, dense, rows_max, i, p):
        L = _lib.lib()
        B, naxs = p.shape[0], p.shape[1]
        ny, nx = [1, 1, 1], [1, 1, 1]
        ny[i], nx[i] = p.shape[2], p.shape[3]
        nyc, nxc = _lib.int_array(ny), _lib.int_array(nx)
        plist = [None] * 3
        plist[i] = p
        grads = [None] * 3
        if ctx.needs_input_grad[4]:
            grads[i] = torch.empty_like(p)
        wsb = L.y5m_compute_loss_dense_workspace_bytes(B, naxs, nyc, nxc, rows_max)
        ws = torch.empty(wsb, dtype=torch.uint8, device=p.device)
        out = torch.zeros(4, dtype=torch.float32, device=p.device)
        _lib.check(L.y5m_compute_loss_dense(_lib.ptr_array(plist), _lib.ptr_array(grads), _lib.ptr_array(dense), B, naxs,
                                            nyc, nxc, owner.nc, _lib.ptr(owner.anchors_d), rows_max,
                                            _lib.float_array(owner.balance), float(owner.lambda_box),
                                            float(owner.lambda_obj), float(owner.lambda_class), _lib.ptr(out),
                                            _lib.ptr(ws), wsb, _lib.stream_ptr()), "y5m_compute_loss_dense")
        ctx.g = grads[i]
        owner.last_components = out[1:4]
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        return None, None, None, None, (ctx.g * gout if ctx.g is not None else None)
