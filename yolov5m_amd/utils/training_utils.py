"""Train-step mechanics -- the callers of the hot path in the reference's utils/training_utils.py.

Two entry points:
  * train_loop(...)       : same signature and control flow as the reference (:81-132): autograd through
                            the native model + native loss, torch.optim step. AMP's GradScaler is a no-op
                            here: the native path computes bf16 activations with f32 accumulation and f32
                            gradients of the loss, so no loss scaling is needed.
  * NativeTrainStep       : the whole step (forward, build-targets + loss, backward, global-norm clip,
                            Adam with L2 decay) as one native launch list over flat buffers, optionally
                            captured into a hipGraph. This is what bench.py measures and what the
                            data-parallel driver (yolov5m_amd/parallel.py) wraps.
"""
import ctypes
import math
import os
import random

import torch
import torch.nn as nn

from .. import _lib, config


# Captured graphs that contain FORKED branches (the weight gradients on the side stream) are never destroyed. Destroying
# such a graph (torch.cuda.CUDAGraph.__del__ -> hipGraphExecDestroy / hipGraphDestroy, ROCm 7.2 + torch 2.10) corrupts the
# host heap: the process then dies in glibc ("free(): invalid pointer", "corrupted size vs. prev_size"), segfaults inside a
# later hipGraphLaunch, replays ANOTHER graph with damaged kernel arguments (wrong results on its first replay) or faults the
# GPU ("Memory access fault ... Reason: Unknown") -- the round-2 "graph replay fault". tools/graph_first_replay.py
# reproduces it within seconds (5 rounds x 15 captured plans: 3 of 3 runs clean with the graphs kept, every run dead without);
# linear graphs (Y5M_OVERLAP=0) are destroyed cleanly (120 create / destroy cycles). So: a forked graph gets one extra
# reference that nobody drops (it is not destroyed at interpreter shutdown either), and once the graphs of Y5M_GRAPH_KEEP_MAX
# (64) captured PLANS are being kept, further plans are captured LINEARLY (weight gradients inline, ~5 % slower per step) so that a long multi_scale run
# which keeps evicting and re-capturing its largest plans cannot leak without bound. NOTES.md has the hunt.
_KEPT_GRAPHS = []
_KEPT_PLANS = 0                  # captures (plans) that keep forked graphs: the cap counts these, not the 1-5 graphs of a plan
_WARNED_LINEAR = False


def _keep_forever(g):
    import ctypes as _c
    _KEPT_GRAPHS.append(g)
    _c.pythonapi.Py_IncRef(_c.py_object(g))
    return g


def _forked_capture_allowed():
    """False once Y5M_GRAPH_KEEP_MAX forked graphs are being kept alive: further plans are captured linearly. Says so once --
    a long multi_scale / data-parallel run that reaches the cap runs its re-captured sizes ~5 % slower from then on."""
    global _WARNED_LINEAR, _KEPT_PLANS
    ok = _KEPT_PLANS < int(os.environ.get("Y5M_GRAPH_KEEP_MAX", "64"))
    if ok:
        _KEPT_PLANS += 1         # (the caller captures one plan: a whole-step graph, or one graph per backward segment)
    elif not _WARNED_LINEAR:
        _WARNED_LINEAR = True
        import warnings
        warnings.warn(f"the captured graphs of {_KEPT_PLANS} plans ({len(_KEPT_GRAPHS)} graphs with forked branches) are being kept "
                      "alive (Y5M_GRAPH_KEEP_MAX): plans captured from now on run their weight gradients inline (about 5 % "
                      "slower per step)")
    return ok


def multi_scale(img, target_shape, max_stride):
    """reference utils/training_utils.py:11-28 (random size in [0.5x, 1x+stride) rounded to the stride,
    bilinear). Pure resampling of the input batch on whatever device it lives."""
    sz = random.randrange(int(target_shape * 0.5), int(target_shape + max_stride)) // max_stride * max_stride
    sf = sz / max(img.shape[2:])
    h, w = img.shape[2:]
    ns = [math.ceil(i * sf / max_stride) * max_stride for i in [h, w]]
    return nn.functional.interpolate(img, size=ns, mode="bilinear", align_corners=False)


def multi_scale_size(h, w, target_shape, max_stride):
    """the (new_h, new_w) multi_scale would resize an (h, w) batch to (same RNG draw as the reference)"""
    sz = random.randrange(int(target_shape * 0.5), int(target_shape + max_stride)) // max_stride * max_stride
    sf = sz / max(h, w)
    return tuple(math.ceil(i * sf / max_stride) * max_stride for i in (h, w))


def preprocess_u8(images_u8, out_hw=None, out=None):
    """Device-side input stage: uint8 (B,3,Hs,Ws) on the GPU -> float32 (B,3,H,W) in [0,1], bilinear-resized
    when out_hw differs from the source size: reference `images.float()/255` (:98) + multi_scale (:11-28) in
    ONE native kernel (y5m_preprocess_u8); only the uint8 batch crosses PCIe."""
    _lib.require_cuda(images_u8)
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[1] != 3:
        raise _lib.Y5MError("preprocess_u8: expected a uint8 (B,3,H,W) tensor")
    img = images_u8.contiguous()
    B, _, Hs, Ws = img.shape
    H, W = out_hw if out_hw is not None else (Hs, Ws)
    if out is None:
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=img.device)
    _lib.check(_lib.lib().y5m_preprocess_u8(_lib.ptr(img), B, Hs, Ws, _lib.ptr(out), H, W, _lib.stream_ptr()),
               "y5m_preprocess_u8")
    return out


def train_loop(model, loader, optim, loss_fn, scaler=None, epoch=0, num_epochs=1, multi_scale_training=True):
    """reference utils/training_utils.py:81-132 (tqdm/printing dropped; returns the mean loss).
    `optim` is what train.py:61 builds -- a torch optimizer: autograd through the native model + loss, then optim.step() -- or a
    NativeTrainStep(model, loss_fn, ...): the same epoch (same accumulation rule, same forced step on the last batch, same input
    stage) with every batch as ONE fused native step (forward + build-targets + loss + backward [+ clip + Adam] as captured graphs)."""
    if isinstance(optim, NativeTrainStep):
        return _train_loop_fused(model, loader, optim, loss_fn, epoch, multi_scale_training)
    nbs = 64                                                   # :87 nominal batch size
    batch_size = len(next(iter(loader))[0])                    # :88
    accumulate = max(round(nbs / batch_size), 1)               # :89
    last_opt_step = -1
    loss_epoch = 0.0
    nb = len(loader)
    optim.zero_grad()
    for idx, (images, bboxes) in enumerate(loader):
        if images.dtype == torch.uint8:
            # same arithmetic as :98-102, on the device: ship the uint8 batch, /255 + bilinear resize natively
            hw = multi_scale_size(images.shape[2], images.shape[3], 640, 32) if multi_scale_training else None
            images = preprocess_u8(images.to(config.DEVICE, non_blocking=True), hw)
        else:
            images = images.float() / 255                      # :98
            if multi_scale_training:
                images = multi_scale(images, target_shape=640, max_stride=32)
            images = images.to(config.DEVICE, non_blocking=True)   # :102
        out = model(images)                                    # :107
        loss = loss_fn(out, bboxes, pred_size=images.shape[2:4], batch_idx=idx, epoch=epoch)
        loss_epoch += float(loss.detach())
        loss.backward()                                        # :114
        if idx - last_opt_step >= accumulate or (idx == nb - 1):
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=10.0)   # :118
            optim.step()
            optim.zero_grad(set_to_none=True)
            last_opt_step = idx
    return loss_epoch / max(nb, 1)


def _train_loop_fused(model, loader, step, loss_fn, epoch, multi_scale_training):
    """train_loop with a NativeTrainStep in the optimizer's place. Accumulation: the reference steps when `idx - last_opt_step >=
    accumulate` (:116, last_opt_step = -1 at the start of every epoch) or on the epoch's last batch, i.e. after every `accumulate`-th
    micro-batch plus one forced step at the end: NativeTrainStep(accumulate=k) + flush(). The mean loss is summed on the device and
    read once at the epoch's end (the reference reads it every 10th batch for its progress bar, :124-127)."""
    if step.model is not model or step.loss_fn is not loss_fn:
        raise _lib.Y5MError("train_loop: the NativeTrainStep passed as `optim` was built for another model / loss object")
    nbs = 64                                                   # :87
    batch_size = len(next(iter(loader))[0])                    # :88
    step.set_accumulate(max(round(nbs / batch_size), 1))       # :89
    dev = model.flat_params.device
    loss_sum, nb = None, len(loader)
    for idx, (images, bboxes) in enumerate(loader):
        if images.dtype == torch.uint8:
            # :98-102 on the device, straight into the plan's static input buffer: no further copy of the batch in step()
            h, w = images.shape[2:4]
            hw = multi_scale_size(h, w, 640, 32) if multi_scale_training else (h, w)
            images = preprocess_u8(images.to(dev, non_blocking=True), hw, out=step.input_buffer(images.shape[0], *hw))
        else:
            images = images.float() / 255                      # :98
            if multi_scale_training:
                images = multi_scale(images, target_shape=640, max_stride=32)
            images = images.to(dev, non_blocking=True)         # :102
        lo = step.step(images, bboxes)                         # :107-121
        loss_sum = lo[0].clone() if loss_sum is None else loss_sum.add_(lo[0])
        if getattr(loss_fn, "save_logs", False) and idx % 100 == 0:
            # the loss objects' own log line (ultralytics_loss.py:108-116, loss.py:82-90): [epoch, batch, box, object, class] every
            # 100th batch -- the fused step does not go through loss_fn.__call__, so it is written here (one host read per 100 batches)
            import csv
            with open(os.path.join("train_eval_metrics", loss_fn.filename, "loss.csv"), "a") as f:
                csv.writer(f).writerow([epoch, idx] + lo[1:4].tolist())
    step.flush()                                               # `idx == nb - 1`
    return float(loss_sum) / max(nb, 1) if loss_sum is not None else 0.0


class NativeTrainStep:
    """Fused native train step on flat buffers.

    step(images, targets): images (B,3,H,W) f32 in [0,1] on the device, targets (nt,6)
    [img, cls, x, y, w, h] (reference collate_fn_ultra format, dataset.py:204-209).
    Optimizer = Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay) with L2-in-gradient (train.py:61)
    after clip_grad_norm_(max_norm) (training_utils.py:118).
    grad_hook: optional callable(flat_grads) run between backward and the optimizer (the data-parallel
    all-reduce plugs in here)."""

    def __init__(self, model, loss_fn, lr=config.LEARNING_RATE, weight_decay=config.WEIGHT_DECAY, max_norm=10.0,
                 betas=(0.9, 0.999), eps=1e-8, nt_max=1024, use_graph=False, grad_hook=None, accumulate=1, overlap=True):
        """accumulate: micro-batches per optimizer step (reference train_loop :87-89, `nbs=64 / batch_size`):
        gradients of `accumulate` consecutive step() calls are SUMMED (the loss is already scaled by the batch
        size, ultralytics_loss.py:118) and clip + Adam run on the sum; flush() forces the step at an epoch end."""
        from ..loss import YOLO_LOSS
        from ..ultralytics_loss import ComputeLoss
        # the two losses of the reference (train.py:102-106: YOLO_LOSS unless --ultralytics_loss): each has its own native
        # build-targets + loss launches; anything else cannot be enqueued into the fused step and is refused HERE, by type
        if isinstance(loss_fn, ComputeLoss):
            self.loss_kind = "ultralytics"
        elif isinstance(loss_fn, YOLO_LOSS):
            self.loss_kind = "yolo"
        else:
            raise _lib.Y5MError(f"NativeTrainStep: loss_fn must be a yolov5m_amd ComputeLoss or YOLO_LOSS, got {type(loss_fn).__name__} "
                                "(an arbitrary callable has no native launch list; use train_loop, which goes through autograd)")
        self.model, self.loss_fn = model, loss_fn
        self.accumulate = max(int(accumulate), 1)
        self._micro = 0
        self.gacc = None
        self.lr, self.wd, self.max_norm, self.betas, self.eps = lr, weight_decay, max_norm, betas, eps
        self.nt_max = nt_max
        self.use_graph = use_graph
        self._overlap_req = overlap
        self.overlap = overlap and accumulate == 1          # bucketed exchange under the backward pass (grad_hook with .launch/.wait)
        self.grad_hook = grad_hook
        model.train()
        model.flatten_parameters()
        dev = model.flat_params.device
        n = model.flat_params.numel()
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.d_step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        L = _lib.lib()
        self.aws_bytes = L.y5m_adam_workspace_bytes()
        self.aws = torch.zeros(self.aws_bytes, dtype=torch.uint8, device=dev)
        self.targets = torch.zeros((nt_max, 6), dtype=torch.float32, device=dev)
        self.d_nt = torch.zeros(1, dtype=torch.int32, device=dev)
        # YOLO_LOSS: the batch's boxes (reference collate_fn arrays, dataset.py:199-202: float64 [cls, x, y, w, h]) back to back
        # and the per-image row ranges, in static buffers filled by load_inputs OUTSIDE any capture
        self.boxes = torch.zeros((nt_max, 5), dtype=torch.float64, device=dev) if self.loss_kind == "yolo" else None
        self._img_off = {}
        self.loss_out = None
        # captured graphs: one forward+loss+backward graph PER PLAN (keyed by the plan's (B, H, W, dtype, mode) key, held
        # together with a strong reference to the plan: id(plan) can be reused after a plan is freed), one optimizer graph
        self._fb_graphs = {}
        self._opt_graph = None
        if self.accumulate > 1:
            self.gacc = torch.zeros(n, dtype=torch.float32, device=dev)

    # forward + build-targets + loss (+ d loss / d logits) + backward: everything before the optimizer
    def _enqueue_fb(self, eng, timeline=None, bwd_upto=None):
        eng._run(eng.pack, timeline)
        eng._run(eng.fwd, timeline)
        loss_ops = self._loss_ops_yolo(eng) if self.loss_kind == "yolo" else self._loss_ops_ultralytics(eng)
        loss_ops.kind = "loss"
        eng._run([(loss_ops, ())], timeline)
        self.loss_out = eng._loss_ws.loss_out
        eng._run(eng.bwd if bwd_upto is None else eng.bwd[:bwd_upto], timeline)
        if bwd_upto is not None:
            eng.join_all()

    def _plan_workspace(self, eng, key, make):
        """The loss workspace belongs to the PLAN: a captured graph holds its addresses, so one workspace shared by all
        plans and re-created whenever the shapes change (what this did until round 2) left the graphs of every other
        resident size replaying on freed memory -- and step() returning another plan's loss tensor -- as soon as
        multi_scale came back to a size. It is dropped with the plan (Engine.release)."""
        ws = getattr(eng, "_loss_ws", None)
        if ws is None or ws.key != key:
            dev = eng.outs[0].device
            m0 = torch.cuda.memory_allocated(dev)
            ws = eng._loss_ws = make(dev)
            # (the plan cache budgets by Engine.nbytes: the workspace belongs to the plan)
            eng.nbytes = getattr(eng, "nbytes", 0) + max(torch.cuda.memory_allocated(dev) - m0, 0)
        return ws

    # ComputeLoss (ultralytics_loss.py:60-120): build-targets + loss (+ d loss / d logits)
    def _loss_ops_ultralytics(self, eng):
        from ..ultralytics_loss import _Workspace
        L = _lib.lib()
        lf = self.loss_fn
        st = _lib.stream_ptr()
        outs = eng.outs
        B = eng.B
        shapes = [(o.shape[2], o.shape[3]) for o in outs]
        ws = self._plan_workspace(eng, (B, eng.naxs, tuple(shapes), self.nt_max),
                                  lambda dev: _Workspace(dev, B, eng.naxs, shapes, self.nt_max))
        grads = eng.head_grad_buffers()
        sparse = os.environ.get("Y5M_SPARSE_HEAD", "1") != "0"
        if sparse and getattr(ws, "owner_ptrs", None) is None:
            own, gob = (ctypes.c_void_p * 3)(), (ctypes.c_void_p * 3)()
            _lib.check(L.y5m_compute_loss_owner_ptrs(_lib.ptr(ws.loss_ws), B, eng.naxs, ws.ny, ws.nx, self.nt_max, own, gob),
                       "y5m_compute_loss_owner_ptrs")
            ws.owner_ptrs = [(int(own[i]), int(gob[i]), ws.bagg[i].data_ptr(), ws.count[i].data_ptr(), ws.cap) for i in range(3)]
        # sparse: the loss writes only the target rows of the dense gradient + a compact objectness plane, and the head
        # backward packs from those (y5m_head_grad_pack_sparse) instead of walking 85 floats per cell
        eng.head_owner = ws.owner_ptrs if sparse else None
        loss_call = L.y5m_compute_loss_sparse if sparse else L.y5m_compute_loss

        def loss_ops():
            _lib.check(L.y5m_build_targets(_lib.ptr(self.targets), 0, _lib.ptr(self.d_nt), self.nt_max,
                                           _lib.ptr(lf.anchors), eng.naxs, ws.ny, ws.nx, float(lf.anchor_t), ws.tg,
                                           _lib.ptr(ws.bt_ws), ws.bt_ws_bytes, st), "y5m_build_targets")
            _lib.check(loss_call(_lib.ptr_array(outs), _lib.ptr_array(grads), B, eng.naxs, ws.ny, ws.nx,
                                          lf.nc, ws.tg, self.nt_max, _lib.float_array(lf.balance),
                                          float(lf.lambda_box), float(lf.lambda_obj), float(lf.lambda_class),
                                          _lib.ptr(ws.loss_out), _lib.ptr(ws.loss_ws), ws.loss_ws_bytes, st),
                       "y5m_compute_loss")
        return loss_ops

    # YOLO_LOSS, the reference's DEFAULT loss (train.py:102-106; loss.py:64-99): dense targets for the whole batch in one launch
    # (loss.py:101-192 is a host loop per image in the reference), then the dense-target loss + d loss / d logits
    def _loss_ops_yolo(self, eng):
        from ..loss import _DenseWorkspace
        L = _lib.lib()
        lf = self.loss_fn
        st = _lib.stream_ptr()
        outs = eng.outs
        B = eng.B
        shapes = [(o.shape[2], o.shape[3]) for o in outs]
        rows_max = max(self.nt_max, 1)                  # positives per scale <= boxes of the batch (one slot per box and scale)
        ws = self._plan_workspace(eng, ("yolo", B, eng.naxs, tuple(shapes), rows_max),
                                  lambda dev: _DenseWorkspace(dev, B, eng.naxs, shapes, rows_max))
        grads = eng.head_grad_buffers()
        sparse = os.environ.get("Y5M_SPARSE_HEAD", "1") != "0"
        if sparse and ws.owner_ptrs is None:
            own, gob, bag, cnt = ((ctypes.c_void_p * 3)() for _ in range(4))
            cap = ctypes.c_int(0)
            _lib.check(L.y5m_compute_loss_dense_owner_ptrs(_lib.ptr(ws.loss_ws), B, eng.naxs, ws.ny, ws.nx, rows_max, own, gob, bag,
                                                           cnt, ctypes.byref(cap)), "y5m_compute_loss_dense_owner_ptrs")
            ws.owner_ptrs = [(int(own[i]), int(gob[i]), int(bag[i]), int(cnt[i]), cap.value) for i in range(3)]
        eng.head_owner = ws.owner_ptrs if sparse else None
        loss_call = L.y5m_compute_loss_dense_sparse if sparse else L.y5m_compute_loss_dense
        off = self._img_off[B]
        stride = _lib.int_array([int(v) for v in lf.S])
        anc = lf._anc                                    # [state read by the launch, state written by it]: same two buffers for every plan

        def loss_ops():
            _lib.check(L.y5m_yolo_build_targets(_lib.ptr(self.boxes), _lib.ptr(off), B, ws.ny, ws.nx, stride, _lib.ptr(anc[0]),
                                                _lib.ptr(anc[1]), float(lf.ignore_iou_thresh), _lib.ptr_array(ws.dense), st),
                       "y5m_yolo_build_targets")
            # the reference's in-place anchor decay (bboxes_utils.py:18) as device state: YOLO_LOSS.__call__ swaps its two buffers
            # on the host, which a captured graph cannot replay -- here the new state is copied back (72 bytes, a graph node)
            anc[0].copy_(anc[1])
            _lib.check(loss_call(_lib.ptr_array(outs), _lib.ptr_array(grads), _lib.ptr_array(ws.dense), B, eng.naxs, ws.ny, ws.nx,
                                 lf.nc, _lib.ptr(lf.anchors_d), rows_max, _lib.float_array(lf.balance), float(lf.lambda_box),
                                 float(lf.lambda_obj), float(lf.lambda_class), _lib.ptr(ws.loss_out), _lib.ptr(ws.loss_ws),
                                 ws.loss_ws_bytes, st), "y5m_compute_loss_dense")
        return loss_ops

    def _optimizer(self, timeline=None):
        L = _lib.lib()
        model = self.model
        n = model.flat_params.numel()
        st = _lib.stream_ptr()

        grads = self.gacc if self.gacc is not None else model.flat_grads

        def opt_ops():
            self.d_step.add_(1)
            _lib.check(L.y5m_grad_norm(_lib.ptr(grads), n, _lib.ptr(self.gnorm), _lib.ptr(self.aws),
                                       self.aws_bytes, st), "y5m_grad_norm")
            _lib.check(L.y5m_adam_step(_lib.ptr(model.flat_params), _lib.ptr(grads), _lib.ptr(self.m),
                                       _lib.ptr(self.v), n, _lib.ptr(self.gnorm), float(self.max_norm),
                                       float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                       float(self.wd), _lib.ptr(self.d_step), st), "y5m_adam_step")
            if self.gacc is not None:
                self.gacc.zero_()
        opt_ops.kind = "optimizer"
        from ..engine import Engine
        Engine._run([(opt_ops, ())], timeline)

    def algorithmic_bytes(self, eng, B=None):
        """ALGORITHMIC HBM bytes of one whole train step on plan `eng` (Engine.algorithmic_bytes + the loss and the optimizer),
        optionally rescaled to batch B (the activation bytes are proportional to the batch, the parameter bytes are not).
        Loss (sparse head gradient): one objectness logit per cell read and one objectness gradient written, f32, plus the
        85-float rows of the matched cells (about three per label); optimizer: the gradient once for the norm, then p, g, m, v
        read and p, m, v written, f32."""
        eb = eng.algorithmic_bytes(sparse_head=os.environ.get("Y5M_SPARSE_HEAD", "1") != "0")     # (the condition _loss_ops_* launch on)
        scale = (B / eng.B) if B else 1.0
        cells = sum(o.shape[1] * o.shape[2] * o.shape[3] for o in eng.outs) * eng.B
        rows = 3 * self.nt_max
        n = self.model.flat_params.numel()
        lists = {k: dict(eb[k]) for k in ("pack", "forward", "backward")}
        lists["loss"] = {"act_read": cells * 4 + rows * 85 * 4, "act_written": cells * 4 + rows * 85 * 4, "par_read": 0, "par_written": 0,
                         "launches": 1, "by_kind": {}}
        if self.loss_kind == "yolo":                       # + the dense targets (6 floats per cell) written by build-targets, read by the loss
            lists["loss"]["act_read"] += cells * 24
            lists["loss"]["act_written"] += cells * 24
        lists["optimizer"] = {"act_read": 0, "act_written": 0, "par_read": 5 * 4 * n, "par_written": 3 * 4 * n, "launches": 2, "by_kind": {}}
        act = sum(v["act_read"] + v["act_written"] for v in lists.values())
        par = sum(v["par_read"] + v["par_written"] for v in lists.values())
        by_kind = {}
        for v in lists.values():
            for k, r in v["by_kind"].items():
                row = by_kind.setdefault(k, [0.0, 0])
                row[0] += (r[0] + r[1]) * scale + r[2] + r[3]
                row[1] += r[4]
        by_kind["loss"] = [(lists["loss"]["act_read"] + lists["loss"]["act_written"]) * scale, 1]
        by_kind["optimizer"] = [float(lists["optimizer"]["par_read"] + lists["optimizer"]["par_written"]), 2]
        return {"B": B or eng.B, "activation_bytes": act * scale, "parameter_bytes": float(par), "total_bytes": act * scale + par,
                "lists": {k: (v["act_read"] + v["act_written"]) * scale + v["par_read"] + v["par_written"] for k, v in lists.items()},
                "by_kind": by_kind}

    def load_inputs(self, images, targets):
        eng = self.model._engine_for(images)
        if images.data_ptr() != eng.x_in.data_ptr():       # a loader that fills input_buffer() directly skips this copy
            eng.x_in.copy_(images, non_blocking=True)
        if self.loss_kind == "yolo":
            self._load_boxes(eng.B, targets)
            return eng
        nt = int(targets.shape[0])
        if nt > self.nt_max:
            raise _lib.Y5MError(f"nt={nt} exceeds nt_max={self.nt_max}")
        if nt:
            self.targets[:nt].copy_(targets.reshape(-1, 6).float(), non_blocking=True)
        self.d_nt.fill_(nt)
        return eng

    def _load_boxes(self, B, targets):
        """YOLO_LOSS targets -> the static float64 box rows + per-image row ranges. Accepts the reference's collate_fn format
        (dataset.py:199-202: a tuple of B per-image arrays (n_i, 5) [cls, x, y, w, h]) and, for callers that hold labels the
        ComputeLoss way, one (nt, 6) array / tensor [img, cls, x, y, w, h] whose rows are grouped by ascending image index
        (collate_fn_ultra, dataset.py:204-209) -- box ORDER is part of the contract (first come first served, loss.py:160-190)."""
        import numpy as np
        if torch.is_tensor(targets) or (isinstance(targets, np.ndarray) and targets.ndim == 2 and targets.shape[1] == 6):
            t = torch.as_tensor(targets).detach().to("cpu").double().reshape(-1, 6).numpy()
            img = t[:, 0].astype(np.int64)
            if img.size and (np.any(np.diff(img) < 0) or img[0] < 0 or img[-1] >= B):
                raise _lib.Y5MError("YOLO_LOSS targets as one (nt, 6) array must be grouped by ascending image index in [0, B)")
            counts = np.bincount(img, minlength=B)
            flat = np.ascontiguousarray(t[:, 1:])
        else:
            per = [np.asarray(b, np.float64).reshape(-1, 5) for b in targets]
            if len(per) != B:
                raise _lib.Y5MError(f"YOLO_LOSS targets: {len(per)} per-image box arrays for a batch of {B}")
            counts = np.array([len(b) for b in per], np.int64)
            flat = np.concatenate(per, 0) if per else np.zeros((0, 5))
        nt = int(counts.sum())
        if nt > self.nt_max:
            raise _lib.Y5MError(f"nt={nt} exceeds nt_max={self.nt_max}")
        off = np.zeros(B + 1, np.int32)
        off[1:] = np.cumsum(counts)
        d_off = self._img_off.get(B)
        if d_off is None:
            d_off = self._img_off[B] = torch.zeros(B + 1, dtype=torch.int32, device=self.boxes.device)
        if nt:
            self.boxes[:nt].copy_(torch.from_numpy(np.ascontiguousarray(flat)), non_blocking=True)
        d_off.copy_(torch.from_numpy(off), non_blocking=True)

    def input_buffer(self, B, H, W):
        """the engine's static (B,3,H,W) float32 input tensor for this shape: a data loader (or preprocess_u8 with
        out=...) can write the batch straight into it; step() then makes no device-to-device copy of the images"""
        probe = torch.empty((B, 3, H, W), dtype=torch.float32, device="meta")
        return self.model._engine_for(probe).x_in

    def step(self, images, targets):
        """Returns the device tensor [loss*bs, lbox, lobj, lcls] of THIS step (no host sync).
        With use_graph the step is two captured hipGraphs (forward+loss+backward | optimizer) replayed
        back to back; grad_hook (the RCCL all-reduce) runs between them on the same stream."""
        eng = self.load_inputs(images, targets)
        self.model._nbt += 1
        self._check_hyper()                              # (before the accumulation branch too: its optimizer graph bakes lr in)
        if self.accumulate > 1:
            return self._step_accumulate(eng)
        if self.overlap and hasattr(self.grad_hook, "launch") and getattr(self.grad_hook, "active", True):
            return self._step_overlapped(eng)
        if not self.use_graph:
            self._enqueue_fb(eng)
            if self.grad_hook is not None:
                self.grad_hook(self.model.flat_grads)
            self._optimizer()
            return self.loss_out
        g1 = self._graph_for(eng)
        if g1 is None:
            # one eager warm-up step (module loading, kernel attributes), then capture
            self._enqueue_fb(eng)
            if self.grad_hook is not None:
                self.grad_hook(self.model.flat_grads)
            self._optimizer()
            self._capture(eng, lambda: self._enqueue_fb(eng))
            return self.loss_out                         # this call WAS the eager step: one call = one step
        g1.replay()
        self.loss_out = eng._loss_ws.loss_out            # (the replayed plan's, not the most recently captured one's)
        if self.grad_hook is not None:
            self.grad_hook(self.model.flat_grads)
        self._opt_graph.replay()
        return self.loss_out

    def _step_overlapped(self, eng):
        """Data-parallel step with the gradient exchange overlapped with the backward pass (north_star: "RCCL all-reduce of
        gradients over xGMI overlapped with the backward convs"). The backward launch list is cut at Engine.grad_cuts()
        into segments; after a segment has been enqueued (eager, or replayed as its own hipGraph), the all-reduce of the
        flat-buffer range that segment finished is launched asynchronously (GradAllReduce.launch: the collective runs on
        the backend's own stream, ordered after everything enqueued so far) and proceeds under the next segment's data and
        weight gradients. Buckets are few and large -- xGMI rings are per-link bound -- and in backward order, head first.
        All exchanges are waited for right before the optimizer."""
        hook, flat = self.grad_hook, self.model.flat_grads
        cuts = getattr(eng, "_cuts", None)
        if cuts is None:
            cuts = eng._cuts = eng.grad_cuts()
        n = flat.numel()
        ks = [k for k, _ in cuts] + [len(eng.bwd)]
        los = [lo for _, lo in cuts] + [0]
        his = [n] + los[:-1]

        def seg(i):                                        # enqueue segment i of forward+loss+backward
            if i == 0:
                self._enqueue_fb(eng, bwd_upto=ks[0])
            else:
                eng._run(eng.bwd[ks[i - 1]:ks[i]])
                if i < len(ks) - 1:
                    eng.join_all()

        graphs = None
        if self.use_graph:
            ent = self._fb_graphs.get(eng.key)
            for k in [k for k, e in self._fb_graphs.items() if e[0].released]:
                del self._fb_graphs[k]
            graphs = ent[1] if (ent is not None and ent[0] is eng and ent[2] == "segments"
                                and self._opt_graph is not None) else None
        if self.use_graph and graphs is None:
            # eager warm-up step (this call's step), then capture one graph per segment
            for i in range(len(ks)):
                seg(i)
                hook.launch(flat, los[i], his[i])
            hook.wait()
            self._optimizer()
            torch.cuda.synchronize()
            forked, saved = eng.overlap and _forked_capture_allowed(), eng.overlap
            try:
                eng.overlap = forked
                gs = []
                for i in range(len(ks)):
                    g = torch.cuda.CUDAGraph()
                    if forked:
                        _keep_forever(g)                 # (before the capture: a capture that fails half-way is kept too)
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        seg(i)
                    gs.append(g)
                if self._opt_graph is None:
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                        self._optimizer()
                    self._opt_graph = g2
                self._fb_graphs[eng.key] = (eng, gs, "segments")
            except Exception as e:
                import warnings
                warnings.warn(f"hipGraph capture failed ({type(e).__name__}: {e}); running the step eagerly")
                torch.cuda.synchronize()
                self.use_graph = False
                eng._pending.clear()
            finally:
                eng.overlap = saved
            return self.loss_out
        for i in range(len(ks)):
            if graphs is not None:
                graphs[i].replay()
            else:
                seg(i)
            hook.launch(flat, los[i], his[i])
        if graphs is not None:
            self.loss_out = eng._loss_ws.loss_out
        hook.wait()
        if graphs is not None:
            self._opt_graph.replay()
        else:
            self._optimizer()
        return self.loss_out

    def _check_hyper(self):
        """the captured optimizer graph holds lr / betas / eps / weight decay / max_norm as kernel arguments: drop it (and the
        plans' graphs, which are only handed out together with it) when any of them was changed on this object"""
        h = (self.lr, tuple(self.betas), self.eps, self.wd, self.max_norm)
        if getattr(self, "_hyper", h) != h:
            self._fb_graphs, self._opt_graph = {}, None
        self._hyper = h

    def _graph_for(self, eng):
        """the captured forward+loss+backward graph of this plan, or None. Entries of plans the model has evicted
        (Engine.released) are dropped here: their buffers are gone."""
        for k in [k for k, e in self._fb_graphs.items() if e[0].released]:
            del self._fb_graphs[k]
        ent = self._fb_graphs.get(eng.key)
        if ent is not None and ent[0] is eng and ent[2] == "whole" and self._opt_graph is not None:
            return ent[1]
        return None

    def _capture(self, eng, enqueue):
        torch.cuda.synchronize()
        forked, saved = eng.overlap and _forked_capture_allowed(), eng.overlap
        try:
            eng.overlap = forked
            # thread_local: a RCCL watchdog / other thread touching the HIP runtime must not abort the capture
            g1 = torch.cuda.CUDAGraph()
            if forked:
                _keep_forever(g1)                        # (before the capture: a capture that fails half-way is kept too)
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                enqueue()
            if self._opt_graph is None:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                    self._optimizer()
                self._opt_graph = g2
            self._fb_graphs[eng.key] = (eng, g1, "whole")
        except Exception as e:                           # capture is an optimisation, never a requirement
            import warnings
            warnings.warn(f"hipGraph capture failed ({type(e).__name__}: {e}); running the step eagerly")
            torch.cuda.synchronize()
            self.use_graph = False
            eng._pending.clear()
        finally:
            eng.overlap = saved

    def set_accumulate(self, k):
        """change the number of micro-batches per optimizer step (train_loop derives it from the loader's batch size, :87-89). A
        pending partial accumulation is stepped first; the captured graphs go (the accumulating graph adds into `gacc`)."""
        k = max(int(k), 1)
        if k == self.accumulate:
            return
        self.flush()
        self.accumulate, self._micro = k, 0
        self.overlap = self._overlap_req and k == 1
        n = self.model.flat_params.numel()
        self.gacc = torch.zeros(n, dtype=torch.float32, device=self.model.flat_params.device) if k > 1 else None
        self._fb_graphs, self._opt_graph = {}, None

    def _step_accumulate(self, eng):
        """micro-batch: forward/backward (+ graph replay) then gacc += grads; every `accumulate`-th call the
        all-reduce hook, clip + Adam on the sum, and gacc = 0"""
        g1 = self._graph_for(eng) if self.use_graph else None
        if self.use_graph and g1 is None:
            self._enqueue_fb(eng)                                  # eager warm-up == this call's micro-batch
            self.gacc.add_(self.model.flat_grads)

            def enqueue():
                self._enqueue_fb(eng)
                self.gacc.add_(self.model.flat_grads)
            self._capture(eng, enqueue)
        elif self.use_graph:
            g1.replay()
            self.loss_out = eng._loss_ws.loss_out
        else:
            self._enqueue_fb(eng)
            self.gacc.add_(self.model.flat_grads)
        self._micro += 1
        if self._micro >= self.accumulate:
            self.flush()
        return self.loss_out

    def flush(self):
        """optimizer step on whatever has been accumulated (reference: `idx == nb-1`, the epoch's last batch)"""
        if self.accumulate <= 1 or self._micro == 0:
            return
        self._check_hyper()
        if self.grad_hook is not None:
            self.grad_hook(self.gacc)
        if self.use_graph and self._opt_graph is not None:
            self._opt_graph.replay()
        else:
            self._optimizer()
        self._micro = 0

    # ---- optimizer state <-> torch.optim.Adam (the reference's checkpoint format, utils/utils.py:56-82) ----
    def optimizer_state_dict(self):
        """{"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]} exactly as
        torch.optim.Adam(model.parameters(), lr, weight_decay).state_dict() would hold it: parameter i is the
        i-th entry of model.parameters(); the flat m / v buffers are split back into per-parameter tensors."""
        params = list(self.model.parameters())
        step = int(self.d_step.item())
        state, off = {}, 0
        for i, p in enumerate(params):
            k = p.numel()
            if step > 0:
                state[i] = {"step": torch.tensor(float(step)),
                            "exp_avg": self.m[off:off + k].view_as(p).clone(),
                            "exp_avg_sq": self.v[off:off + k].view_as(p).clone()}
            off += k
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        """inverse of optimizer_state_dict(); accepts a torch.optim.Adam state_dict of the same model"""
        params = list(self.model.parameters())
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.wd = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        off, step = 0, 0
        self.m.zero_(); self.v.zero_()
        for i, p in enumerate(params):
            k = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                self.m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                step = max(step, int(float(st["step"])))
            off += k
        self.d_step.fill_(step)
        self._fb_graphs, self._opt_graph = {}, None   # captured scalars (lr, betas) may have changed

    def profile_step(self, images, targets, detail=False, overlapped=False):
        """One EAGER step with a HIP event pair around every launch (recorded on the launch stream).
        Returns {kernel family: (total ms, launches)}; with detail=True also the list of (ms, y5m_conv_args) of
        every forward-conv / data-gradient launch, for per-shape rooflines.
        overlapped=False: the forked weight gradients run inline (serialised), so the per-family times add up.
        overlapped=True: the step's own schedule -- weight gradients on the forked stream next to the main stream's
        BatchNorm backward / data gradients, each launch timed by events on the stream it runs on (what a launch costs
        INSIDE the step; the durations then overlap and do not add up to the step time)."""
        eng = self.load_inputs(images, targets)
        self.model._nbt += 1
        tl = []
        saved = eng.overlap
        side_tl = [] if overlapped else None
        if not overlapped:
            eng.overlap = False                      # serialise the side stream so per-family times add up
        eng._tl_side = side_tl
        try:
            self._enqueue_fb(eng, tl)
            self._optimizer(tl)
            torch.cuda.synchronize()
        finally:
            eng.overlap = saved
            eng._tl_side = None
        if side_tl:
            # a forked launch's main-stream pair only brackets its ENQUEUE: take the pair recorded on the forked stream
            by_fn = {id(fn): (e0, e1) for fn, e0, e1 in side_tl}
            items_ = list(eng.pack) + list(eng.fwd) + [None] + list(eng.bwd) + [None]
            tl = [(kind, *by_fn.get(id(item[0]), (e0, e1))) if item is not None else (kind, e0, e1)
                  for (kind, e0, e1), item in zip(tl, items_)]
        fam = {}
        for kind, e0, e1 in tl:
            ms, n = fam.get(kind, (0.0, 0))
            fam[kind] = (ms + e0.elapsed_time(e1), n + 1)
        if not detail:
            return fam
        # the timeline is pack + fwd + [loss] + bwd + [optimizer], one entry per op
        items = list(eng.pack) + list(eng.fwd) + [None] + list(eng.bwd) + [None]
        convs = []
        for (kind, e0, e1), item in zip(tl, items):
            if item is None or kind != "conv_igemm":
                continue
            d = getattr(item[0], "__defaults__", None)
            if d and hasattr(d[0], "_length_"):            # y5m_conv_multi: one launch, several problems (time split by work)
                ms, parts = e0.elapsed_time(e1), list(d[0])
                tot = float(sum(a.K for a in parts))
                convs.extend((ms * a.K / tot, a) for a in parts)
            elif d:
                convs.append((e0.elapsed_time(e1), d[0]))
        self.last_bwd_pw = [(e0.elapsed_time(e1), item[0].bp) for (kind, e0, e1), item in zip(tl, items)
                            if item is not None and kind == "bwd_pw"]
        if detail != "kernels":
            return fam, convs
        # per-kernel view: (kernel instantiation name, ms, algorithmic flop) of every conv / data-gradient / weight-gradient
        # launch (y5m_*_kernel_name asks the library which instantiation it dispatches to)
        L = _lib.lib()
        buf = ctypes.create_string_buffer(192)
        kern = []
        for (kind, e0, e1), item in zip(tl, items):
            if item is None:
                continue
            ms = e0.elapsed_time(e1)
            if kind == "conv_igemm":
                d = getattr(item[0], "__defaults__", None)
                if d and hasattr(d[0], "_length_"):
                    arr = d[0]
                    _lib.check(L.y5m_conv_multi_kernel_name(arr, len(arr), eng.dtype, buf, 192), "kernel_name")
                    kern.append((buf.value.decode(), ms, sum(2.0 * a.M * a.N * a.K for a in arr)))
                elif d:
                    _lib.check(L.y5m_conv_kernel_name(ctypes.byref(d[0]), eng.dtype, buf, 192), "kernel_name")
                    kern.append((buf.value.decode(), ms, 2.0 * d[0].M * d[0].N * d[0].K))
            elif kind == "bwd_pw":
                a = item[0].bp          # fused pointwise backward: data gradient + weight gradient flops; HBM-bound (bench.py by_class)
                r4 = (L.y5m_r4_kernel_forms() >> 3) & 1                               # (the form launch_bp picks, as the library parsed it)
                kern.append((f"bwd_pw_kernel<{a.C},{int(bool(a.accumulate or a.res))},{r4}>", ms, 4.0 * a.M * a.N * a.C,
                             float(a.M) * (2 * a.N + (3 if (a.accumulate or a.res) else 2) * a.C) * 2))     # + algorithmic bytes
            elif kind == "wgrad" and getattr(item[0], "wa", None) is not None:
                wa = item[0].wa
                _lib.check(L.y5m_wgrad_kernel_name(ctypes.byref(wa), eng.dtype, buf, 192), "kernel_name")
                kern.append((buf.value.decode() + " (+ unpack)", ms, 2.0 * wa.M * wa.N * wa.th * wa.tw * wa.C))
        return fam, convs, kern
