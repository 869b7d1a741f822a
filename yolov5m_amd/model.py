"""YOLOV5m -- same constructor / attributes / state_dict surface as the reference model.py:178-239,
executed by the native gfx950 engine (yolov5m_amd/engine.py) instead of ATen.

The sub-modules (CBL, Bottleneck, C3, SPPF, HEADS) carry the parameters under the reference's
names (481 state_dict keys) with PyTorch's default initialisation; their tensors are
re-pointed into ONE flat f32 parameter buffer (and one flat gradient buffer) the first time the model
runs on the GPU, which is what the fused optimizer and the RCCL gradient all-reduce operate on.
YOLOV5m.forward is a single autograd Function around the native forward / backward plan (the hot path).
The sub-modules' own forward() (reference model.py:26, :49, :90, :106, :165 -- `model.backbone[i](x)`) also executes, WITH
autograd: every CBL is one autograd node (`_CBLFn`) whose forward is the native conv (+ batch statistics) + BatchNorm + SiLU
launch sequence and whose backward is the native BatchNorm / SiLU backward, data gradient and weight gradient, through the
op-level C-ABI wrappers (yolov5m_amd/ops.py: NCHW in / out with a layout conversion per call, `compute_dtype` "f32" unless
set); the SPPF pool cascade (`_SppfPoolFn`) and the head convs (`_ConvBiasFn`) likewise. The glue between them (residual add,
channel concat, the head's view / permute) is torch tensor arithmetic on the GPU, differentiated by torch. This is the
module-level API surface, not the hot path: the whole model runs as ONE autograd node over the engine's plan (`_ModelFn`).
"""
import os

import torch
import torch.nn as nn

from . import _lib
from .arch import cbl_list


class CBL(nn.Module):
    """reference model.py:12-28 (parameter container; BN eps=1e-3, momentum=0.03)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding):
        super().__init__()
        conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=False)
        bn = nn.BatchNorm2d(out_channels, eps=1e-3, momentum=0.03)
        self.cbl = nn.Sequential(conv, bn, nn.SiLU(inplace=True))

    compute_dtype = "f32"        # sub-module execution: "f32" (parity) or "bf16"; per instance or on the class

    def forward(self, x):
        """reference model.py:26-28: train mode uses batch statistics and updates the running ones (momentum 0.03, eps 1e-3),
        eval mode the running statistics; differentiable wrt the input and the parameters (one autograd node, `_CBLFn`)"""
        _lib.require_cuda(x)
        conv, bn = self.cbl[0], self.cbl[1]
        return _CBLFn.apply(x.float(), conv.weight, bn.weight, bn.bias, self)


class _CBLFn(torch.autograd.Function):
    """conv + BatchNorm + SiLU of one CBL as ONE autograd node over the native op-level launches (reference model.py:12-28)"""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, mod):
        from . import ops
        conv, bn = mod.cbl[0], mod.cbl[1]
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        dtype = mod.compute_dtype
        ch = 8 if dtype == "bf16" else 4
        cin = x.shape[1]
        xd, wd = x.detach().float(), w.detach().float()
        if cin % ch:                                          # the stem's 3 input channels: zero channels up to a 16-byte piece
            padc = ch - cin % ch
            xd = torch.nn.functional.pad(xd, (0, 0, 0, 0, 0, padc))
            wd = torch.nn.functional.pad(wd, (0, 0, 0, 0, 0, padc))
        ctx.geom = (k, s, p, dtype, cin, tuple(x.shape[2:]))
        ctx.training = mod.training
        if mod.training:
            y, z, sc, sh, mean, invstd, rm, rv = ops.conv_forward_bn_fused(
                xd, wd, s, p, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var, bn.momentum, bn.eps, dtype)
            bn.running_mean.copy_(rm)
            bn.running_var.copy_(rv)
            bn.num_batches_tracked += 1
        else:
            invstd = 1.0 / torch.sqrt(bn.running_var.float() + bn.eps)
            mean = bn.running_mean.float()
            sc = gamma.detach().float() * invstd
            sh = beta.detach().float() - mean * sc
            need_y = any(ctx.needs_input_grad[:4])
            y = ops.conv_forward(xd, wd, s, p, dtype) if need_y else None      # (raw conv output: only the backward reads it)
            z = ops.bn_act(y, sc, sh, dtype) if need_y else ops.conv_forward(xd, wd, s, p, dtype, scale=sc, shift=sh, act=True)
        ctx.save_for_backward(xd, wd, y, sc, sh, mean, invstd)
        return z

    @staticmethod
    def backward(ctx, dz):
        from . import ops
        xd, wd, y, sc, sh, mean, invstd = ctx.saved_tensors
        k, s, p, dtype, cin, hw = ctx.geom
        dz = dz.contiguous().float()
        if ctx.training:
            dy, dgamma, dbeta = ops.bn_silu_backward(dz, y, sc, sh, mean, invstd, dtype)
        else:
            # running statistics are constants: elementwise glue (the eval-mode backward is not a training path)
            t = y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
            sg = torch.sigmoid(t)
            dt = dz * (sg * (1 + t * (1 - sg)))
            dy = dt * sc.view(1, -1, 1, 1)
            dbeta = dt.sum((0, 2, 3))
            dgamma = (dt * (y - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)).sum((0, 2, 3))
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv_dgrad(dy, wd, hw, s, p, dtype)[:, :cin]
        if ctx.needs_input_grad[1]:
            dw = ops.conv_wgrad(dy, xd, k, s, p, dtype)[:, :cin]
        return dx, dw, dgamma, dbeta, None


class Bottleneck(nn.Module):
    """reference model.py:32-50"""

    def __init__(self, in_channels, out_channels, width_multiple=1):
        super().__init__()
        c_ = int(width_multiple * in_channels)
        self.c1 = CBL(in_channels, c_, kernel_size=1, stride=1, padding=0)
        self.c2 = CBL(c_, out_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        """reference model.py:49-50"""
        return self.c2(self.c1(x)) + x.float()


class C3(nn.Module):
    """reference model.py:54-92"""

    def __init__(self, in_channels, out_channels, width_multiple=1, depth=1, backbone=True):
        super().__init__()
        c_ = int(width_multiple * in_channels)
        self.c1 = CBL(in_channels, c_, kernel_size=1, stride=1, padding=0)
        self.c_skipped = CBL(in_channels, c_, kernel_size=1, stride=1, padding=0)
        if backbone:
            self.seq = nn.Sequential(*[Bottleneck(c_, c_, width_multiple=1) for _ in range(depth)])
        else:
            self.seq = nn.Sequential(*[nn.Sequential(CBL(c_, c_, 1, 1, 0), CBL(c_, c_, 3, 1, 1)) for _ in range(depth)])
        self.c_out = CBL(c_ * 2, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x):
        """reference model.py:90-92"""
        return self.c_out(torch.cat([self.seq(self.c1(x)), self.c_skipped(x)], dim=1))


class SPPF(nn.Module):
    """reference model.py:96-112"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        c_ = int(in_channels // 2)
        self.c1 = CBL(in_channels, c_, 1, 1, 0)
        self.pool = nn.MaxPool2d(kernel_size=5, stride=1, padding=2)
        self.c_out = CBL(c_ * 4, out_channels, 1, 1, 0)

    def forward(self, x):
        """reference model.py:106-112: the three cascaded 5x5 max-pools are ONE native launch (y5m_sppf_pool)"""
        x = self.c1(x)
        p1, p2, p3 = _SppfPoolFn.apply(x, self.c1.compute_dtype)
        return self.c_out(torch.cat([x, p1, p2, p3], dim=1))


class _SppfPoolFn(torch.autograd.Function):
    """the three cascaded MaxPool2d(5, 1, 2) of SPPF (reference model.py:108-110) as one autograd node: forward
    y5m_sppf_pool, backward three accumulating y5m_maxpool5_bwd launches"""

    @staticmethod
    def forward(ctx, x, dtype):
        from . import ops
        p1, p2, p3 = ops.sppf_pool(x.detach().float(), dtype)
        ctx.save_for_backward(x.detach().float(), p1, p2)
        ctx.dtype = dtype
        return p1, p2, p3

    @staticmethod
    def backward(ctx, g1, g2, g3):
        from . import ops
        x, p1, p2 = ctx.saved_tensors
        z = torch.zeros_like(x)
        g = [z] + [t.contiguous().float() if t is not None else z for t in (g1, g2, g3)]
        return ops.sppf_pool_backward(x, p1, p2, g, ctx.dtype), None


class HEADS(nn.Module):
    """reference model.py:143-175"""

    def __init__(self, nc=80, anchors=(), ch=()):
        super().__init__()
        self.nc = nc
        self.nl = len(anchors)
        self.naxs = len(anchors[0])
        self.stride = [8, 16, 32]
        anchors_ = torch.tensor(anchors).float().view(self.nl, -1, 2) / \
            torch.tensor(self.stride).repeat(6, 1).T.reshape(3, 3, 2)
        self.register_buffer("anchors", anchors_)
        self.out_convs = nn.ModuleList()
        for in_channels in ch:
            self.out_convs += [nn.Conv2d(in_channels=in_channels, out_channels=(5 + self.nc) * self.naxs, kernel_size=1)]

    compute_dtype = "f32"

    def forward(self, x):
        """reference model.py:165-175: per scale a 1x1 conv with bias, viewed as (B, naxs, ny, nx, 5 + nc); differentiable
        (`_ConvBiasFn`: native forward, data gradient and weight gradient)"""
        out = []
        for i in range(self.nl):
            conv = self.out_convs[i]
            _lib.require_cuda(x[i])
            y = _ConvBiasFn.apply(x[i].float(), conv.weight, conv.bias, self.compute_dtype)
            bs, _, ny, nx = y.shape
            out.append(y.view(bs, self.naxs, 5 + self.nc, ny, nx).permute(0, 1, 3, 4, 2).contiguous())
        return out


class _ConvBiasFn(torch.autograd.Function):
    """1x1 conv + bias of a detection head (reference model.py:157-160, :170) over the native op-level launches"""

    @staticmethod
    def forward(ctx, x, w, b, dtype):
        from . import ops
        ncout = w.shape[0]
        padn = (-ncout) % 4                                            # 255 -> 256 output channels: 16-byte rows
        xd = x.detach().float()
        wd = torch.nn.functional.pad(w.detach().float(), (0, 0, 0, 0, 0, 0, 0, padn))
        bd = torch.nn.functional.pad(b.detach().float(), (0, padn))
        y = ops.conv_forward(xd, wd, 1, 0, dtype, scale=torch.ones_like(bd), shift=bd, act=False)[:, :ncout]
        ctx.save_for_backward(xd, wd)
        ctx.meta = (dtype, ncout, padn)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        xd, wd = ctx.saved_tensors
        dtype, ncout, padn = ctx.meta
        dy = dy.contiguous().float()
        dyp = torch.nn.functional.pad(dy, (0, 0, 0, 0, 0, padn))
        dx = ops.conv_dgrad(dyp, wd, tuple(xd.shape[2:]), 1, 0, dtype) if ctx.needs_input_grad[0] else None
        dw = ops.conv_wgrad(dyp, xd, 1, 1, 0, dtype)[:ncout] if ctx.needs_input_grad[1] else None
        db = dy.sum((0, 2, 3)) if ctx.needs_input_grad[2] else None
        return dx, dw, db, None


class _ModelFn(torch.autograd.Function):
    """One autograd node for the whole network: forward = native forward plan, backward = native
    backward plan; parameter gradients come back as views of the flat gradient buffer."""

    @staticmethod
    def forward(ctx, model, x, *params):
        eng = model._engine_for(x)
        outs = eng.forward(x)
        ctx.model, ctx.eng = model, eng
        if model.static_outputs:
            return tuple(outs)
        return tuple(o.clone() for o in outs)

    @staticmethod
    def backward(ctx, g0, g1, g2):
        model, eng = ctx.model, ctx.eng
        grads = []
        for g, o in zip((g0, g1, g2), eng.outs):
            grads.append(g.contiguous().float() if g is not None else torch.zeros_like(o))
        eng.backward(grads)
        return (None, None) + tuple(model._grad_views())


class YOLOV5m(nn.Module):
    """reference model.py:178-239"""

    def __init__(self, first_out, nc=80, anchors=(), ch=(), inference=False):
        super().__init__()
        self.inference = inference
        self.first_out = first_out
        f = first_out
        self.backbone = nn.ModuleList()
        self.backbone += [
            CBL(in_channels=3, out_channels=f, kernel_size=6, stride=2, padding=2),
            CBL(in_channels=f, out_channels=f * 2, kernel_size=3, stride=2, padding=1),
            C3(in_channels=f * 2, out_channels=f * 2, width_multiple=0.5, depth=2),
            CBL(in_channels=f * 2, out_channels=f * 4, kernel_size=3, stride=2, padding=1),
            C3(in_channels=f * 4, out_channels=f * 4, width_multiple=0.5, depth=4),
            CBL(in_channels=f * 4, out_channels=f * 8, kernel_size=3, stride=2, padding=1),
            C3(in_channels=f * 8, out_channels=f * 8, width_multiple=0.5, depth=6),
            CBL(in_channels=f * 8, out_channels=f * 16, kernel_size=3, stride=2, padding=1),
            C3(in_channels=f * 16, out_channels=f * 16, width_multiple=0.5, depth=2),
            SPPF(in_channels=f * 16, out_channels=f * 16),
        ]
        self.neck = nn.ModuleList()
        self.neck += [
            CBL(in_channels=f * 16, out_channels=f * 8, kernel_size=1, stride=1, padding=0),
            C3(in_channels=f * 16, out_channels=f * 8, width_multiple=0.25, depth=2, backbone=False),
            CBL(in_channels=f * 8, out_channels=f * 4, kernel_size=1, stride=1, padding=0),
            C3(in_channels=f * 8, out_channels=f * 4, width_multiple=0.25, depth=2, backbone=False),
            CBL(in_channels=f * 4, out_channels=f * 4, kernel_size=3, stride=2, padding=1),
            C3(in_channels=f * 8, out_channels=f * 8, width_multiple=0.5, depth=2, backbone=False),
            CBL(in_channels=f * 8, out_channels=f * 8, kernel_size=3, stride=2, padding=1),
            C3(in_channels=f * 16, out_channels=f * 16, width_multiple=0.5, depth=2, backbone=False),
        ]
        self.head = HEADS(nc=nc, anchors=anchors, ch=ch)
        # native-engine state
        self.compute_dtype = "bf16"        # "bf16" (throughput) | "f32" (parity: exact-f32 MFMA)
        self.static_outputs = False        # True: forward returns the engine's own output buffers
        # True: an inference plan packs its weights once per weight VERSION instead of on every forward (Engine.forward; frozen
        # deployed weights -- writers that torch's version counters do not see must call mark_weights_changed())
        self.pack_once = os.environ.get("Y5M_PACK_ONCE", "0") == "1"
        self.flat_params = None
        self.flat_grads = None
        self.pslices = None
        self._engines = {}
        self._flat_device = None

    # ------------------------------------------------------------------ flat parameter storage
    def _named_units(self):
        """(unit name, conv module, bn module or None) in state_dict order"""
        mods = dict(self.named_modules())
        units = [(c.name, mods[c.name + ".cbl.0"], mods[c.name + ".cbl.1"]) for c in cbl_list(self.first_out)]
        for i in range(self.head.nl):
            units.append((f"head.out_convs.{i}", mods[f"head.out_convs.{i}"], None))
        return units

    def flatten_parameters(self):
        """Re-point every parameter into one flat f32 buffer (and gradients into another). Idempotent
        per device. Running BN statistics are flattened too (one f32 buffer + one int64 counter buffer)."""
        dev = next(self.parameters()).device
        if self._flat_device == dev and self.flat_params is not None:
            return
        _lib.require_cuda_device(dev)
        # (a first forward under torch.inference_mode() must not turn the flat buffers -- which live as long as the model and are
        #  trained in place later -- into inference tensors)
        with torch.inference_mode(False):
            self._flatten(dev)

    def _flatten(self, dev):
        params = list(self.parameters())
        n = sum(p.numel() for p in params)
        flat = torch.empty(n, dtype=torch.float32, device=dev)
        gflat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._param_list, self._grad_list = [], []
        off = 0
        pmap = {}
        with torch.no_grad():
            for p in params:
                k = p.numel()
                flat[off:off + k].copy_(p.detach().reshape(-1).float())
                p.data = flat[off:off + k].view(p.shape)
                pmap[id(p)] = gflat[off:off + k].view(p.shape)
                self._param_list.append(p)
                self._grad_list.append(pmap[id(p)])
                off += k
        units = self._named_units()
        nstat = sum(bn.num_features for _, _, bn in units if bn is not None)
        sflat = torch.empty(2 * nstat, dtype=torch.float32, device=dev)
        nbt = torch.zeros(sum(1 for _, _, bn in units if bn is not None), dtype=torch.int64, device=dev)
        so, bi = 0, 0
        self.pslices = {}
        with torch.no_grad():
            for name, conv, bn in units:
                d = {"w": conv.weight.data, "gw": pmap[id(conv.weight)]}
                if bn is not None:
                    c = bn.num_features
                    sflat[so:so + c].copy_(bn.running_mean)
                    sflat[nstat + so:nstat + so + c].copy_(bn.running_var)
                    nbt[bi] = bn.num_batches_tracked
                    bn.running_mean = sflat[so:so + c]
                    bn.running_var = sflat[nstat + so:nstat + so + c]
                    bn.num_batches_tracked = nbt[bi]
                    d.update(g=bn.weight.data, b=bn.bias.data, rm=bn.running_mean, rv=bn.running_var,
                             gg=pmap[id(bn.weight)], gb=pmap[id(bn.bias)])
                    so += c
                    bi += 1
                else:
                    d.update(b=conv.bias.data, gb=pmap[id(conv.bias)])
                self.pslices[name] = d
        self.flat_params, self.flat_grads, self._flat_stats, self._nbt = flat, gflat, sflat, nbt
        self._flat_device = dev
        self._engines = {}

    def _grad_views(self):
        return self._grad_list

    def _check_flat(self):
        """parameters must still alias the flat buffer (a .to()/.half() after flattening breaks it)"""
        p0 = self._param_list[0]
        if p0.data_ptr() != self.flat_params.data_ptr():
            self.flat_params = None
            self.flatten_parameters()

    def _engine_for(self, x):
        from .engine import Engine
        self.flatten_parameters()
        self._check_flat()
        B, _, H, W = x.shape
        dt = _lib.BF16 if self.compute_dtype == "bf16" else _lib.F32
        key = (B, H, W, dt, self.training)
        eng = self._engines.pop(key, None)
        if eng is None:
            # Resident plans are bounded by HBM, not by a small count: multi_scale training (reference
            # utils/training_utils.py:11-28) draws one of 11 sizes (320..640 step 32) per batch, and a plan that is evicted
            # and rebuilt costs seconds (allocation + graph capture) against a 30 ms step. One training plan at B=64 /
            # 640^2 holds ~40 GB and the footprint scales with B*H*W, so the plans of ALL sizes up to 576 (~180 GB) stay
            # resident in the default budget (60 % of the device's memory; Y5M_ENGINE_CACHE_GB) and only the two largest
            # take turns. Least-recently-used eviction with an explicit release, so the evicted plan's HBM is free BEFORE
            # the new one allocates (a plan is a reference cycle -- its launch closures capture it -- and would otherwise
            # live until the cyclic GC runs) and a captured graph can never be replayed on freed buffers (Engine.released).
            # Y5M_ENGINE_CACHE caps the COUNT (default 16).
            cap = max(1, int(os.environ.get("Y5M_ENGINE_CACHE", "16")))
            gb = float(os.environ.get("Y5M_ENGINE_CACHE_GB", "0"))
            dev = self.flat_params.device                  # (x may be a meta tensor: NativeTrainStep.input_buffer probes with one)
            budget = gb * 2**30 if gb > 0 else 0.6 * torch.cuda.get_device_properties(dev).total_memory
            per_px = max((e.nbytes / max(e.key[0] * e.key[1] * e.key[2], 1) for e in self._engines.values()
                          if e.key[3:] == key[3:]), default=0.0)
            if per_px == 0.0:
                # the first plan of this dtype / mode: estimate from ANY resident plan (an eval plan holds about a third of a
                # training plan per pixel, an f32 one twice a bf16 one: the larger figure errs towards evicting)
                per_px = max((e.nbytes / max(e.key[0] * e.key[1] * e.key[2], 1) for e in self._engines.values()), default=0.0)
            need = per_px * B * H * W                     # estimate from a resident plan (of the same dtype / mode when there is one)
            while self._engines and (len(self._engines) >= cap or
                                     sum(e.nbytes for e in self._engines.values()) + need > budget):
                self._engines.pop(next(iter(self._engines))).release()
            # (collect first: a dead plan of ANOTHER model is a reference cycle, and if the cyclic GC frees its tens of GB
            #  while this plan is being built the allocator delta below comes out as zero)
            import gc
            gc.collect()
            m0 = torch.cuda.memory_allocated(dev)
            eng, retry = None, False
            try:
                with torch.inference_mode(False):      # (a plan outlives the inference_mode block that first asked for it)
                    eng = Engine(self, B, H, W, dtype=dt, training=self.training)
            except torch.OutOfMemoryError:
                # the estimate was too small (or there was nothing to estimate from). Only NOTE it here: while this handler
                # runs, the exception's traceback keeps the frames of the failed Engine.__init__ -- and with them the partly
                # built plan's tens of GB -- alive, so nothing freed in here would make room for the retry
                retry = True
            if retry:
                while self._engines:                      # drop EVERY resident plan, then retry once
                    self._engines.pop(next(iter(self._engines))).release()
                gc.collect()                              # (the dead partial plan is a reference cycle through its launch closures)
                torch.cuda.empty_cache()
                m0 = torch.cuda.memory_allocated(dev)
                eng = Engine(self, B, H, W, dtype=dt, training=self.training)
            eng.key = key
            eng.nbytes = max(torch.cuda.memory_allocated(dev) - m0, 0)
            while self._engines and sum(e.nbytes for e in self._engines.values()) + eng.nbytes > budget:
                self._engines.pop(next(iter(self._engines))).release()
        self._engines[key] = eng               # (re-)insert at the most-recently-used end
        return eng

    def mark_weights_changed(self):
        """With `pack_once` (opt-in; Engine.forward) inference plans pack their bf16 weight rows and fold BatchNorm once per weight
        version, read from torch's in-place counters (Engine._weights_key). Code that writes parameters or running statistics behind
        torch's back -- a custom kernel through data_ptr(), a collective (dist.broadcast / all_reduce bump no counter), a write through
        a fresh `.data` view -- calls this to make the next inference forward pack again. Without `pack_once` every forward packs."""
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1

    # ------------------------------------------------------------------ reference API
    def forward(self, x):
        assert x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0, "Width and Height aren't divisible by 32!"
        _lib.require_cuda(x)
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        self.flatten_parameters()
        if self.training:
            self._nbt += 1                      # BatchNorm2d.num_batches_tracked
        if torch.is_grad_enabled() and self.training:
            outs = _ModelFn.apply(self, x, *self._param_list)
        else:
            eng = self._engine_for(x)
            outs = eng.forward(x)
            if not self.static_outputs:
                outs = [o.clone() for o in outs]
        return list(outs)
