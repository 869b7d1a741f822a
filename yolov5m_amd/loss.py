"""YOLO_LOSS -- same signature as the reference's loss.py:20-246.

  * build_targets (loss.py:101-192: a Python double loop over boxes x 9 anchors per image on the host, the
    reference's documented hot loop #2) is ONE native launch for the whole batch (y5m_yolo_build_targets: one
    workgroup per image, float64 box arithmetic, dense targets written on the device), including the
    reference's stateful defect: utils/bboxes_utils.py:18 divides the loss object's anchors by 640 IN PLACE
    for every box, so only the first box ever sees the real anchors. Parity is defined on that
    behaviour, bit for bit; the anchor state lives on the device and `anchors` reads it back.
  * compute_loss (loss.py:195-246) for the 3 scales and its autograd backward run on the MI355X in the
    dense-target variant of the native loss kernels (y5m_compute_loss_dense).
"""
import csv
import os

import numpy as np
import torch

from . import _lib, config


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


class _DenseWorkspace:
    """Device buffers of the dense-target loss for one PLAN of the fused train step (NativeTrainStep with YOLO_LOSS): the dense
    targets of the 3 scales, the loss workspace (row tables + per-cell planes) and the 4 loss values. A captured graph holds
    these addresses, so the workspace lives and dies with the plan (Engine._loss_ws)."""

    def __init__(self, device, B, naxs, shapes, rows_max):
        L = _lib.lib()
        self.key = ("yolo", B, naxs, tuple(shapes), rows_max)
        self.ny = _lib.int_array([s[0] for s in shapes])
        self.nx = _lib.int_array([s[1] for s in shapes])
        self.dense = [torch.zeros((B, naxs, ny, nx, 6), dtype=torch.float32, device=device) for (ny, nx) in shapes]
        self.loss_ws_bytes = L.y5m_compute_loss_dense_workspace_bytes(B, naxs, self.ny, self.nx, rows_max)
        self.loss_ws = torch.empty(self.loss_ws_bytes, dtype=torch.uint8, device=device)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=device)
        self.owner_ptrs = None


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
        # :40 `self.anchors` (decays in place!): device-resident state, two buffers (the kernel reads [0] and writes [1]; [1] is copied back)
        self._anc = [self.anchors_d.clone().float().contiguous(), torch.zeros_like(self.anchors_d, dtype=torch.float32)]
        self.na = self.anchors_d.reshape(9, 2).shape[0]
        self.num_anchors_per_scale = self.na // 3
        self.S = model.head.stride
        self.ignore_iou_thresh = 0.5
        self.save_logs = save_logs
        self.filename = filename
        self.last_components = None
        _lib.require_cuda(self.anchors_d)                     # (the model must live on the GPU: no CPU fallback)
        if self.save_logs and not resume:                                             # :51-62
            folder = os.path.join("train_eval_metrics", filename)
            os.makedirs(folder, exist_ok=True)
            with open(os.path.join(folder, "loss.csv"), "w") as f:
                csv.writer(f).writerow(["epoch", "batch_idx", "box_loss", "object_loss", "class_loss"])

    @property
    def anchors(self):
        """the reference's CPU `self.anchors` (loss.py:40) after every in-place decay so far"""
        return self._anc[0].detach().to("cpu")

    def _build_targets_native(self, shapes, boxes_list):
        """dense targets [(B,3,ny,nx,6)]*3 on the device for a list of per-image (n_i,5) float64 arrays"""
        L = _lib.lib()
        dev = self.anchors_d.device
        B = len(boxes_list)
        counts = [int(len(b)) for b in boxes_list]
        off = np.zeros(B + 1, np.int32)
        off[1:] = np.cumsum(counts)
        flat = (np.concatenate([np.asarray(b, np.float64).reshape(-1, 5) for b in boxes_list], 0) if off[-1] else
                np.zeros((1, 5), np.float64))
        d_boxes = torch.from_numpy(np.ascontiguousarray(flat)).to(dev, non_blocking=True)
        d_off = torch.from_numpy(off).to(dev, non_blocking=True)
        dense = [torch.empty((B, self.num_anchors_per_scale, ny, nx, 6), dtype=torch.float32, device=dev) for (ny, nx) in shapes]
        ny = _lib.int_array([sh[0] for sh in shapes])
        nx = _lib.int_array([sh[1] for sh in shapes])
        st = _lib.int_array([int(v) for v in self.S])
        _lib.check(L.y5m_yolo_build_targets(_lib.ptr(d_boxes), _lib.ptr(d_off), B, ny, nx, st, _lib.ptr(self._anc[0]),
                                            _lib.ptr(self._anc[1]), float(self.ignore_iou_thresh), _lib.ptr_array(dense),
                                            _lib.stream_ptr()), "y5m_yolo_build_targets")
        # the state after this call goes back into _anc[0] (72 bytes, same stream): _anc[0] is ALWAYS the state buffer, so a captured
        # NativeTrainStep graph of this loss object (which holds both addresses) and eager calls of __call__ can be mixed freely
        self._anc[0].copy_(self._anc[1])
        self._keep = (d_boxes, d_off)                 # until the launch has run
        return dense

    def __call__(self, preds, targets, pred_size, batch_idx=None, epoch=None):
        """reference loss.py:64-99. preds: 3 logits tensors; targets: tuple of per-image ndarrays (n_i,5)
        [cls, x, y, w, h] (reference collate_fn, dataset.py:199-202)."""
        self.batch_idx, self.epoch = batch_idx, epoch
        shapes = [(preds[i].shape[2], preds[i].shape[3]) for i in range(len(self.S))]
        dense = self._build_targets_native(shapes, list(targets))                                    # :68-74
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
        """reference loss.py:101-192 for ONE image -> list of 3 CPU tensors (3, ny, nx, 6) (the reference's return
        value; __call__ keeps the batch on the device instead)."""
        shapes = [(input_tensor[i].shape[2], input_tensor[i].shape[3]) for i in range(len(self.S))]
        dense = self._build_targets_native(shapes, [np.asarray(bboxes, np.float64).reshape(-1, 5)])
        return [d[0].to("cpu") for d in dense]

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
