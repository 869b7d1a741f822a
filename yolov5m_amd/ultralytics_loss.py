"""ComputeLoss -- same signature as the reference's ultralytics_loss.py:17-311, HIP underneath.

build_targets is one launch (bit-exact row order / integers / fp32 bits), the loss forward AND its
analytic backward are four launches (y5m_compute_loss); autograd sees a single Function whose
backward just scales the gradients the forward already produced.
"""
import csv
import os

import torch

from . import _lib, config


class _Workspace:
    """Device buffers for one (B, shapes, nt_max) configuration (caller-owned, reused across steps)."""

    def __init__(self, device, B, naxs, shapes, nt_max):
        L = _lib.lib()
        self.key = (B, naxs, tuple(shapes), nt_max)
        self.cap = 5 * naxs * max(nt_max, 1)
        self.ny = _lib.int_array([s[0] for s in shapes])
        self.nx = _lib.int_array([s[1] for s in shapes])
        i32, f32 = torch.int32, torch.float32
        self.count = [torch.zeros(1, dtype=i32, device=device) for _ in range(3)]
        self.bagg = [torch.zeros((self.cap, 4), dtype=i32, device=device) for _ in range(3)]
        self.tbox = [torch.zeros((self.cap, 4), dtype=f32, device=device) for _ in range(3)]
        self.anch = [torch.zeros((self.cap, 2), dtype=f32, device=device) for _ in range(3)]
        self.tcls = [torch.zeros((self.cap,), dtype=i32, device=device) for _ in range(3)]
        self.tg = (_lib.Targets * 3)()
        for s in range(3):
            self.tg[s].count = self.count[s].data_ptr()
            self.tg[s].bagg = self.bagg[s].data_ptr()
            self.tg[s].tbox = self.tbox[s].data_ptr()
            self.tg[s].anch = self.anch[s].data_ptr()
            self.tg[s].tcls = self.tcls[s].data_ptr()
        self.bt_ws_bytes = L.y5m_build_targets_workspace_bytes(nt_max, naxs)
        self.bt_ws = torch.empty(self.bt_ws_bytes, dtype=torch.uint8, device=device)
        self.loss_ws_bytes = L.y5m_compute_loss_workspace_bytes(B, naxs, self.ny, self.nx, nt_max)
        self.loss_ws = torch.empty(self.loss_ws_bytes, dtype=torch.uint8, device=device)
        self.loss_out = torch.zeros(4, dtype=f32, device=device)
        self.d_nt = torch.zeros(1, dtype=i32, device=device)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, targets, ws, p0, p1, p2):
        L = _lib.lib()
        p = (p0, p1, p2)
        B, naxs = p0.shape[0], p0.shape[1]
        need_grad = any(ctx.needs_input_grad[3:6])
        grads = [torch.empty_like(t) for t in p] if need_grad else [None, None, None]
        st = _lib.stream_ptr()
        nt = targets.shape[0]
        _lib.check(L.y5m_build_targets(_lib.ptr(targets), nt, None, ws.key[3], _lib.ptr(owner.anchors), naxs,
                                       ws.ny, ws.nx, float(owner.anchor_t), ws.tg, _lib.ptr(ws.bt_ws),
                                       ws.bt_ws_bytes, st), "y5m_build_targets")
        _lib.check(L.y5m_compute_loss(_lib.ptr_array(p), _lib.ptr_array(grads), B, naxs, ws.ny, ws.nx,
                                      owner.nc, ws.tg, ws.key[3], _lib.float_array(owner.balance),
                                      float(owner.lambda_box), float(owner.lambda_obj),
                                      float(owner.lambda_class), _lib.ptr(ws.loss_out), _lib.ptr(ws.loss_ws),
                                      ws.loss_ws_bytes, st), "y5m_compute_loss")
        ctx.grads = grads
        out = ws.loss_out.clone()
        owner.last_components = out[1:4]
        return out[0:1]

    @staticmethod
    def backward(ctx, gout):
        g = ctx.grads
        if g[0] is None:
            return None, None, None, None, None, None
        return None, None, None, g[0] * gout, g[1] * gout, g[2] * gout


class ComputeLoss:
    """reference ultralytics_loss.py:17-120 (constructor :21-58, __call__ :60-120)."""
    sort_obj_iou = False

    def __init__(self, model, save_logs=False, filename=None, resume=False):
        device = next(model.parameters()).device
        self.lambda_class = 0.5 * (model.head.nc / 80 * 3 / model.head.nl)           # :31
        self.lambda_obj = 1 * ((config.IMAGE_SIZE / 640) ** 2 * 3 / model.head.nl)   # :32
        self.lambda_box = 0.05 * (3 / model.head.nl)                                 # :33
        self.anchor_t = 4.0                                                          # :35
        self.balance = [4.0, 1.0, 0.4]                                               # :37
        self.na = model.head.naxs
        self.nc = model.head.nc
        self.nl = model.head.nl
        self.anchors = model.head.anchors
        self.device = device
        self.save_logs = save_logs
        self.filename = filename
        self.last_components = None
        self._ws = None
        if self.nl != 3:
            raise _lib.Y5MError("the native loss supports nl == 3 detection layers")
        if self.save_logs and not resume:                                            # :47-58
            folder = os.path.join("train_eval_metrics", filename)
            os.makedirs(folder, exist_ok=True)
            with open(os.path.join(folder, "loss.csv"), "w") as f:
                csv.writer(f).writerow(["epoch", "batch_idx", "box_loss", "object_loss", "class_loss"])

    # -- helpers ---------------------------------------------------------------------------------
    def _workspace(self, p, nt):
        B, naxs = p[0].shape[0], p[0].shape[1]
        shapes = [(t.shape[2], t.shape[3]) for t in p]
        ws = self._ws
        if ws is None or ws.key[:3] != (B, naxs, tuple(shapes)) or ws.key[3] < nt:
            nt_max = max(nt, 8)
            if ws is not None and ws.key[:3] == (B, naxs, tuple(shapes)):
                nt_max = max(nt_max, 2 * ws.key[3])
            ws = self._ws = _Workspace(p[0].device, B, naxs, shapes, nt_max)
        return ws

    def _prep(self, p, targets):
        anchors = self.anchors
        _lib.require_cuda(anchors)                            # (the model must live on the GPU: no CPU fallback)
        targets = torch.as_tensor(targets).to(anchors.device, non_blocking=True)     # :63
        targets = targets.float().reshape(-1, 6).contiguous()
        p = [t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous() for t in p]
        _lib.require_cuda(*p)
        return p, targets

    # -- API -------------------------------------------------------------------------------------
    def __call__(self, p, targets, pred_size=None, batch_idx=None, epoch=None):
        p, targets = self._prep(p, targets)
        ws = self._workspace(p, targets.shape[0])
        loss = _LossFn.apply(self, targets, ws, p[0], p[1], p[2])
        if self.save_logs and batch_idx is not None and batch_idx % 100 == 0:        # :108-116
            lb, lo, lc = self.last_components.tolist()
            with open(os.path.join("train_eval_metrics", self.filename, "loss.csv"), "a") as f:
                csv.writer(f).writerow([epoch, batch_idx, lb, lo, lc])
        return loss

    def build_targets(self, p, targets):
        """reference ultralytics_loss.py:122-311 -> (tcls, tbox, indices, anch) lists of 3."""
        L = _lib.lib()
        p, targets = self._prep(p, targets)
        ws = self._workspace(p, targets.shape[0])
        naxs = p[0].shape[1]
        _lib.check(L.y5m_build_targets(_lib.ptr(targets), targets.shape[0], None, ws.key[3],
                                       _lib.ptr(self.anchors), naxs, ws.ny, ws.nx, float(self.anchor_t),
                                       ws.tg, _lib.ptr(ws.bt_ws), ws.bt_ws_bytes, _lib.stream_ptr()),
                   "y5m_build_targets")
        counts = torch.cat(ws.count).tolist()
        tcls, tbox, indices, anch = [], [], [], []
        for s in range(3):
            n = counts[s]
            bagg = ws.bagg[s][:n].long()
            indices.append((bagg[:, 0].clone(), bagg[:, 1].clone(), bagg[:, 2].clone(), bagg[:, 3].clone()))
            tbox.append(ws.tbox[s][:n].clone())
            anch.append(ws.anch[s][:n].clone())
            tcls.append(ws.tcls[s][:n].long())
        return tcls, tbox, indices, anch
