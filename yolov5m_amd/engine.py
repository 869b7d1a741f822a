"""Native execution plan for the YOLOv5m train / inference step.

The topology (reference model.py:210-239) is static for a given (B, H, W, dtype, mode), so it is
expanded ONCE into two flat launch lists -- forward and backward -- of pre-bound C-ABI calls over
pre-allocated HBM buffers. Running a list enqueues HIP kernels on the current stream; nothing is
allocated, nothing synchronises, so a whole step can be captured into one hipGraph and replayed.

Memory plan (all resident in HBM, sized for 288 GB):
  * activations are NHWC (pixel-major, channel-minor) in the compute dtype; every concat of the
    reference (C3 :91, SPPF :112, PANet joins :226/:230) is a wider buffer whose producers write their
    channel slice directly (ptr + offset, ld) -- torch.cat never materialises;
  * training keeps, per CBL, the raw conv output (input of BN) for the backward pass;
  * gradients mirror the activation buffers; one scratch holds dy of the layer being differentiated;
  * weights: f32 masters (reference layout) -> packed K-contiguous copies in the compute dtype per
    step (forward rows + data-gradient rows), weight gradients accumulate in packed f32.
"""
import ctypes

import os
import torch

from . import _lib
from .arch import BN_EPS, BN_MOMENTUM, blocks

import os as _os
_TRACE = _os.environ.get("Y5M_TRACE", "0") == "1"
from ._lib import (ConvArgs, WgradArgs, EPI_RAW_STATS, EPI_AFFINE_ACT, EPI_HEAD, EPI_DGRAD, ACT_NONE, ACT_SILU,
                   F32, BF16)

_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}


def _kind(fn, kind, traffic=None):
    """tag a launch closure with its kernel family (used by the live per-family timing in bench.py) and, where the plan knows
    them, its algorithmic HBM bytes (see _traffic)"""
    fn.kind = kind
    if traffic is not None:
        fn.traffic = tuple(int(v) for v in traffic)
    return fn


def _traffic(fn, act_r=0, act_w=0, par_r=0, par_w=0):
    """ALGORITHMIC HBM bytes of one launch-list entry -- every operand moved once, whatever the kernel re-reads through L2:
    (activation bytes read, activation bytes written: both proportional to the batch; parameter / gradient / workspace bytes
    read, written: independent of it). Engine.algorithmic_bytes() sums them; tests/test_traffic_cpu.py pins the step's total."""
    fn.traffic = (int(act_r), int(act_w), int(par_r), int(par_w))
    return fn


def _as_list(op):
    return op if isinstance(op, list) else [op]


def _rup(x, m):
    return (x + m - 1) // m * m


class Act:
    """A (ptr, ld) view of an NHWC activation: B x H x W pixels, C channels at channel offset `off`
    inside a buffer whose pixel stride is `ld` elements."""

    def __init__(self, buf, B, H, W, C, ld=None, off=0):
        self.buf, self.B, self.H, self.W, self.C = buf, B, H, W, C
        self.ld = ld if ld is not None else C
        self.off = off
        self.grad = None          # Act over the gradient buffer (training)
        self.gw = False           # plan-time flag: gradient already written in this backward pass
        self.lazy = None          # plan-time: a gradient (Act) still to be added into .grad (see Engine._flush_lazy)
        self.children = []        # channel slices of a concat buffer
        self.parent, self.c0 = None, 0     # the concat buffer this view is a slice of, first channel in it
        self.n_cons = 0           # plan-time: ops whose backward writes this tensor's gradient (counted on the root)
        self.n_written = 0        # plan-time: how many of them have been emitted so far in this backward plan
        self.producer = None      # the CBL (_Layer with BatchNorm) whose output z this view is

    @property
    def M(self):
        return self.B * self.H * self.W

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.off * self.buf.element_size()

    def slice(self, c0, C):
        s = Act(self.buf, self.B, self.H, self.W, C, self.ld, self.off + c0)
        if self.grad is not None:
            s.grad = Act(self.grad.buf, self.B, self.H, self.W, C, self.grad.ld, self.grad.off + c0)
        self.children.append(s)
        s.parent, s.c0 = self, c0
        return s

    def as_nchw_f32(self):
        """debug/test helper: dense (B,C,H,W) f32 copy"""
        flat = torch.as_strided(self.buf.view(-1), (self.B, self.H, self.W, self.C),
                                (self.H * self.W * self.ld, self.W * self.ld, self.ld, 1), self.off)
        return flat.permute(0, 3, 1, 2).float().contiguous()


class _Layer:
    """Per-CBL (or head conv) static data."""
    pass


class Engine:
    def __init__(self, model, B, H, W, dtype=BF16, training=True, nt_max=0):
        assert H % 32 == 0 and W % 32 == 0, "Width and Height aren't divisible by 32!"   # model.py:211
        self.L = _lib.lib()
        self.released = False
        self.key = None                 # the model's plan-cache key (set by YOLOV5m._engine_for)
        self.model = model
        self.B, self.H, self.W = B, H, W
        self.dtype = dtype
        self.tdt = _TORCH_DT[dtype]
        self.training = training
        self.dev = model.flat_params.device
        self.CH = 8 if dtype == BF16 else 4
        self.BK = 64 if dtype == BF16 else 32
        self.esz = 2 if dtype == BF16 else 4            # bytes of an activation element
        self.nc, self.naxs = model.head.nc, model.head.naxs
        self.nch = 5 + self.nc
        self.fwd, self.bwd = [], []
        self.head_owner = None          # per scale (owner, objectness-gradient plane, target rows, row count, capacity) of the loss, see _head
        self._pending = {}
        import os
        self.overlap = os.environ.get("Y5M_OVERLAP", "1") != "0"   # wgrad on a forked stream (see _side_op)
        # dy scratch ring: the main stream only waits for the weight gradient that used a slot nslots layers ago,
        # so it runs ahead of the side stream instead of ping-ponging with it (each cross-stream wait costs
        # ~10-15 us of dependency latency inside a hipGraph); 0.63 GB per slot at B=64 / 640^2; with the weight gradient forked after the data gradient 3 slots measure
        # 0.35 ms/step better than 2 (and than 4)
        self.nslots = max(2, int(os.environ.get("Y5M_SLOTS", "3")))
        self.lazy_residual = os.environ.get("Y5M_LAZY_RES", "1") != "0"
        self.merge_c3 = os.environ.get("Y5M_MERGE_C3", "1") != "0"
        # fused backward of the pointwise CBLs with 48 / 96 / 192 channels in and out (csrc/y5m_bwd_pw.hip): BatchNorm apply +
        # data gradient + weight gradient in one launch after the reduce pass. Y5M_BWD_PW=0: the three-launch form
        self.fused_pw = os.environ.get("Y5M_BWD_PW", "1") != "0"
        # ... only for launches with at least this many pixels: the kernel owns its CU (eight waves, 130 KB of LDS at C = 192) and
        # pays ~30 us of prologue + weight-gradient atomics per launch, so on the 40x40 layers (102 400 pixels at B = 64) the
        # three-launch form next to the forked weight gradients is the faster one (tools/bwd_pw_bench.py, tools/ab_step.sh)
        self.fused_pw_min_m = int(os.environ.get("Y5M_BWD_PW_MIN_M", "200000"))
        self.fused_stem = os.environ.get("Y5M_BWD_STEM", "1") != "0"         # the stem's bn apply + weight gradient as ONE launch (y5m_bwd_stem)
        self.wgrad_direct = os.environ.get("Y5M_WGRAD_DIRECT", "1") != "0"    # 1x1 layers: atomics straight into flat_grads
        self._direct_wgrads = 0          # C3: c1 + c_skipped as one GEMM (_cbl_pair)
        self._bwd_stack = []
        self.layers = []
        self._scratch_elems = 0
        self._stats_floats = 0
        self._bnws_bytes = 0
        # BatchNorm statistics through f64 accumulator rows (csrc/y5m_bnfuse.h): producers add, consumers derive their
        # coefficients in a prologue -- no partial rows and no finalise launches (158 per step). Y5M_BN_FUSE: 1 (default)
        # forward and backward, 2 / 3 forward / backward only, 0 the three-launch form
        mode = int(self.L.y5m_bn_fuse_enabled()) if training else 0
        self.fuse_f, self.fuse_b = mode in (1, 2), mode in (1, 3)
        self._slots = int(self.L.y5m_bn_acc_slots())
        self._accf = self._accb = 0          # doubles of the forward / backward accumulator rows handed out so far
        self._acc_users = []                 # (conv descriptor, offset): bn_acc is set once the buffer exists
        self._build()

    def release(self):
        """drop every launch closure and tensor of this plan (called by the model's plan cache on eviction). The closures
        capture `self`, so without this the plan is a reference cycle whose HBM only returns when Python's cyclic GC
        runs; `released` tells holders of captured graphs (NativeTrainStep) that replaying them would touch freed memory."""
        self.released = True
        keep = {"released", "key"}
        for k in list(self.__dict__):
            if k not in keep:
                del self.__dict__[k]

    # ------------------------------------------------------------------ allocation helpers
    def _new_act(self, B, H, W, C, need_grad=True):
        buf = torch.zeros((B * H * W * C,), dtype=self.tdt, device=self.dev)
        a = Act(buf, B, H, W, C)
        if self.training and need_grad:
            a.grad = Act(torch.zeros_like(buf), B, H, W, C)
        return a

    def _call(self, lst, fn, *args):
        lst.append((fn, args))

    # ------------------------------------------------------------------ conv descriptors
    def _conv_args(self, x, w, out_ptr, Ho, Wo, k, s, p, N, ldout, epi, Kp, cin=None, **kw):
        a = ConvArgs()
        a.zeros = _lib.zero_page(self.dev).data_ptr()
        cin = cin if cin is not None else x.C
        a.inp, a.w, a.out = x.ptr, w.data_ptr(), out_ptr
        a.B, a.Hin, a.Win, a.ldin = x.B, x.H, x.W, x.ld
        a.Hg, a.Wg, a.sy, a.sx = Ho, Wo, s, s
        a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = k, k, -p, 1, -p, 1
        a.Cin, a.K, a.Kp, a.N, a.M = cin, k * k * cin, Kp, N, x.B * Ho * Wo
        a.Hout, a.Wout, a.ldout, a.osy, a.osx, a.ooy, a.oox = Ho, Wo, ldout, 1, 1, 0, 0
        a.epi, a.act, a.accumulate = epi, ACT_NONE, 0
        a.Np = _rup(N, self.L.y5m_conv_tile_n(N))
        for key, v in kw.items():
            setattr(a, key, v)
        return a

    def _run_conv(self, lst, a):
        L, dt = self.L, self.dtype

        def fn(a=a):
            _lib.check(L.y5m_conv(ctypes.byref(a), dt, _lib.stream_ptr()), "y5m_conv")
        fn.kind = "conv_igemm"
        _traffic(fn, *self._conv_traffic([a]))
        lst.append((fn, ()))

    def _conv_traffic(self, descs):
        """algorithmic bytes of one y5m_conv / y5m_conv_multi launch: the input tensor once (the problems of a multi launch --
        the parity classes of a stride-2 data gradient -- read the SAME input), every output element once (+ once more where
        the epilogue accumulates onto it or takes a residual / lazy accumulation source), the packed weights once"""
        esz = self.esz
        a0 = descs[0]
        act_r = a0.B * a0.Hin * a0.Win * a0.Cin * esz
        act_w = par_r = 0
        for a in descs:
            out = a.M * a.N * (4 if a.epi == EPI_HEAD else esz)
            act_w += out
            if a.accumulate or a.res:
                act_r += a.M * a.N * esz
            par_r += a.N * a.K * esz
        return act_r, act_w, par_r, 0

    # ------------------------------------------------------------------ building blocks
    def _cbl(self, name, x, cout, k, s, p, dest=None, res=None, stem=False):
        """reference model.py:12-28 (+ Bottleneck residual :50 when res is given)."""
        L, dt, st = self.L, self.dtype, _lib.stream_ptr
        P = self.model.pslices[name]          # dict of flat-buffer views: w, g, b, rm, rv (+ grads)
        lay = _Layer()
        lay.name, lay.x, lay.res, lay.stem = name, x, res, stem
        lay.cin_real = 3 if stem else x.C
        lay.cout, lay.k, lay.s, lay.p = cout, k, s, p
        kk, ss, pp = (3, 1, 1) if stem else (k, s, p)     # stem runs as 3x3/s1 on the s2d input
        Ho = (x.H + 2 * pp - kk) // ss + 1
        Wo = (x.W + 2 * pp - kk) // ss + 1
        lay.Ho, lay.Wo = Ho, Wo
        M = x.B * Ho * Wo
        K = kk * kk * x.C
        Kp = _rup(K, self.BK)
        BN = L.y5m_conv_tile_n(cout)
        Np = _rup(cout, BN)
        lay.K, lay.Kp, lay.Np, lay.kk, lay.ss, lay.pp, lay.M = K, Kp, Np, kk, ss, pp, M
        lay.wf = torch.zeros((Np, Kp), dtype=self.tdt, device=self.dev)
        if dest is None:
            dest = self._new_act(x.B, Ho, Wo, cout)
        lay.z = dest
        bn = torch.zeros((4, cout), dtype=torch.float32, device=self.dev)   # scale, shift, mean, invstd
        lay.bn = bn
        mode = 2 if stem else 0
        # ---- per-step weight pack (masters may have changed)
        self._call(self.pack, L.y5m_pack_weights, _lib.ptr(P["w"]), cout, lay.cin_real, k, k, mode, 0, 1, kk, 0, 1,
                   kk, _lib.ptr(lay.wf), Np, Kp, 0, dt)
        if self.training:
            lay.y = torch.zeros((M * cout,), dtype=self.tdt, device=self.dev)
            lay.y_ptr, lay.y_ld = lay.y.data_ptr(), cout
            lay.y_view = (lay.y, 0, cout)         # (tensor, first channel, pixel stride): the raw conv output for tests / tools
            dest.producer = lay
            self._consume(x)
            self._consume(res)
            a = self._conv_args(x, lay.wf, lay.y.data_ptr(), Ho, Wo, kk, ss, pp, cout, cout, EPI_RAW_STATS, Kp)
            lay.fwd_args = a
            if self.fuse_f:
                acc_off = self._accf
                self._accf += self._slots * 2 * Np
                self._acc_users.append((a, acc_off))
                if L.y5m_conv_stages_stats(ctypes.byref(a), dt):
                    # the persistent halo-patch kernel stages its tiles' sums in partial rows and adds them once per workgroup
                    self._stats_floats = max(self._stats_floats, L.y5m_conv_stats_rows(ctypes.byref(a), dt) * 2 * Np)
                    self._stat_users.append(a)
                self._run_conv(self.fwd, a)

                def apply(lay=lay, P=P, bn=bn, dest=dest, res=res, M=M, Np=Np, acc_off=acc_off):
                    _lib.check(L.y5m_bn_act_fused(_lib.ptr(lay.y), lay.cout, self.accf.data_ptr() + 8 * acc_off, Np, M,
                                                  _lib.ptr(P["g"]), _lib.ptr(P["b"]), _lib.ptr(P["rm"]), _lib.ptr(P["rv"]),
                                                  BN_MOMENTUM, BN_EPS, 1, bn[0].data_ptr(), bn[1].data_ptr(),
                                                  bn[2].data_ptr(), bn[3].data_ptr(),
                                                  res.ptr if res is not None else None, res.ld if res is not None else 0,
                                                  dest.ptr, dest.ld, M, lay.cout, ACT_SILU, dt, st()), "y5m_bn_act_fused")
            else:
                tiles_m = L.y5m_conv_stats_rows(ctypes.byref(a), dt)      # partial rows this launch writes (kernel dependent)
                self._stats_floats = max(self._stats_floats, tiles_m * 2 * Np)
                self._stat_users.append(a)
                self._run_conv(self.fwd, a)

                def finalize(lay=lay, P=P, tiles_m=tiles_m, Np=Np, M=M, bn=bn):
                    _lib.check(L.y5m_bn_finalize(_lib.ptr(self.stats), tiles_m, Np, lay.cout, M, _lib.ptr(P["g"]),
                                                 _lib.ptr(P["b"]), _lib.ptr(P["rm"]), _lib.ptr(P["rv"]),
                                                 BN_MOMENTUM, BN_EPS, bn[0].data_ptr(), bn[1].data_ptr(),
                                                 bn[2].data_ptr(), bn[3].data_ptr(), 1, _lib.ptr(self.finws),
                                                 self.finws.numel(), st()), "y5m_bn_finalize")
                self.fwd.append((finalize, ()))

                def apply(lay=lay, bn=bn, dest=dest, res=res, M=M):
                    _lib.check(L.y5m_bn_act(_lib.ptr(lay.y), lay.cout, bn[0].data_ptr(), bn[1].data_ptr(),
                                            res.ptr if res is not None else None, res.ld if res is not None else 0,
                                            dest.ptr, dest.ld, M, lay.cout, ACT_SILU, dt, st()), "y5m_bn_act")
            _traffic(apply, M * cout * self.esz * (2 if res is not None else 1), M * cout * self.esz, 8 * cout, 16 * cout)
            apply.kind = "apply_fused" if self.fuse_f else "apply"
            self.fwd.append((apply, ()))
            self._cbl_backward(lay, P)
        else:
            # eval mode: BatchNorm folded into (scale, shift); all layers in ONE launch (see _batch_packs)
            self._fold_jobs.append((P["g"].data_ptr(), P["b"].data_ptr(), P["rm"].data_ptr(), P["rv"].data_ptr(),
                                    bn[0].data_ptr(), bn[1].data_ptr(), lay.cout))
            a = self._conv_args(x, lay.wf, dest.ptr, Ho, Wo, kk, ss, pp, cout, dest.ld, EPI_AFFINE_ACT, Kp,
                                act=ACT_SILU, scale=bn[0].data_ptr(), shift=bn[1].data_ptr(),
                                res=res.ptr if res is not None else None, ldres=res.ld if res is not None else 0)
            lay.fwd_args = a
            self._run_conv(self.fwd, a)
        self.layers.append(lay)
        return dest

    # backward of one CBL, pushed on a stack (executed in reverse order of the forward)
    def _cbl_backward(self, lay, P):
        L, dt, st = self.L, self.dtype, _lib.stream_ptr
        M, cout = lay.M, lay.cout
        self._scratch_elems = max(self._scratch_elems, M * cout)
        self._bnws_bytes = max(self._bnws_bytes, L.y5m_bn_bwd_workspace_bytes(M, cout))
        lay.accb_off = self._accb
        self._accb += self._slots * 2 * cout
        ntap = lay.kk * lay.kk
        lay.gw_off = self._gw_floats
        lay.ldgw = ntap * lay.x.C
        self._gw_floats += lay.cout * lay.ldgw
        # data-gradient weights
        x = lay.x
        need_dx = x.grad is not None
        lay.wd = []
        if need_dx:
            if lay.ss == 1:
                Kd = ntap * cout
                wd = torch.zeros((_rup(x.C, L.y5m_conv_tile_n(x.C)), _rup(Kd, self.BK)), dtype=self.tdt, device=self.dev)
                lay.wd.append((wd, 0, 0, lay.kk, 0, 1, lay.pp, 0, 1, lay.pp))   # (buf, py, px, th, kh0, khs, dh0, kw0, kws, dw0)
                self._call(self.pack, L.y5m_pack_weights, _lib.ptr(P["w"]), cout, x.C, lay.k, lay.k, 1, 0, 1, lay.kk,
                           0, 1, lay.kk, _lib.ptr(wd), wd.shape[0], wd.shape[1], 0, dt)
            else:
                assert lay.ss == 2
                for py in range(2):
                    kh0 = (py + lay.pp) % 2
                    th = len(range(kh0, lay.kk, 2))
                    dh0 = (py + lay.pp - kh0) // 2
                    for px in range(2):
                        kw0 = (px + lay.pp) % 2
                        tw = len(range(kw0, lay.kk, 2))
                        dw0 = (px + lay.pp - kw0) // 2
                        Kd = th * tw * cout
                        wd = torch.zeros((_rup(x.C, L.y5m_conv_tile_n(x.C)), _rup(Kd, self.BK)), dtype=self.tdt,
                                         device=self.dev)
                        lay.wd.append((wd, py, px, (th, tw), kh0, 2, dh0, kw0, 2, dw0))
                        self._call(self.pack, L.y5m_pack_weights, _lib.ptr(P["w"]), cout, x.C, lay.k, lay.k, 1, kh0, 2,
                                   th, kw0, 2, tw, _lib.ptr(wd), wd.shape[0], wd.shape[1], 0, dt)

        def backward(lay=lay, P=P, need_dx=need_dx):
            if self.fused_pw and self.fuse_b and need_dx and lay.kk == 1 and lay.ss == 1 and lay.res is None and not lay.stem:
                fops = self._bwd_pw_ops(lay.x, lay.y_ptr, lay.y_ld, lay.wd[0][0], [(lay, P)], lay.M, lay.cout)
                if fops is not None:
                    return fops
            if lay.stem and self.fused_stem and self.fuse_b and not need_dx:
                fops = self._bwd_stem_ops(lay, P)
                if fops is not None:
                    return fops
            ops = []
            z = lay.z
            dz = z.grad
            ops.extend(self._flush_lazy(z))           # (a pending residual copy nobody fused: make it now)
            bn = lay.bn
            slot = self._next_slot()
            scratch = self.scratch2[slot]
            ops.append((self._join_op(slot), ()))     # previous user of this dy buffer (its wgrad) must be done
            # residual branch: d(res) (+)= dz   (Bottleneck add, model.py:50)
            if lay.res is not None and lay.res.grad is not None:
                acc = 1 if lay.res.gw else 0
                lay.res.gw = True
                rg = lay.res.grad
                self._written(lay.res)
                if acc == 0 and self.lazy_residual:
                    # first writer of d(res): a plain copy of dz. Do not make it -- the next writer (the data
                    # gradient of the bottleneck's first conv) reads dz as its accumulation source instead:
                    # d(res) = dz + dgrad, one pass less over the tensor (see _flush_lazy for every other writer)
                    lay.res.lazy = dz
                else:
                    ops.extend(self._flush_lazy(lay.res))
                    ops.append((_kind(lambda rg=rg, dz=dz, acc=acc: _lib.check(
                        L.y5m_add(dz.ptr, dz.ld, rg.ptr, rg.ld, lay.M, lay.cout, acc, dt, st()), "y5m_add"), "add",
                        (lay.M * lay.cout * self.esz * (2 if acc else 1), lay.M * lay.cout * self.esz, 0, 0)), ()))
            # BN + SiLU backward -> dy (scratch), dgamma, dbeta
            ops.extend(_as_list(self._bn_backward_op(lay, P, dz, scratch.data_ptr(), lay.cout)))
            # weight gradient (packed f32, atomics into the zeroed gw buffer)
            wa = WgradArgs()
            wa.zeros = _lib.zero_page(self.dev).data_ptr()
            wa.dy, wa.x = scratch.data_ptr(), lay.x.ptr
            wa.dwgt = self.gw.data_ptr() + 4 * lay.gw_off
            wa.B, wa.Hin, wa.Win, wa.ldx = lay.x.B, lay.x.H, lay.x.W, lay.x.ld
            wa.Hg, wa.Wg, wa.sy, wa.sx = lay.Ho, lay.Wo, lay.ss, lay.ss
            wa.th, wa.tw, wa.dh0, wa.dhs, wa.dw0, wa.dws = lay.kk, lay.kk, -lay.pp, 1, -lay.pp, 1
            wa.C, wa.N, wa.M, wa.lddy, wa.lddw, wa.ksplit = lay.x.C, lay.cout, lay.M, lay.cout, lay.ldgw, 0
            lay.wgrad_args = wa
            # weight gradient + its unpack (packed f32 -> reference-layout flat gradient) run on the SIDE
            # stream, concurrently with this layer's data gradient and the next layer's BN backward:
            # wgrad is HBM/atomic-bound, dgrad MFMA-bound, and neither fills the chip alone
            mode = 2 if lay.stem else 0
            ops.append((self._side_op(self._wgrad_ops(lay, wa, [(0, lay.cout, lay.cin_real, lay.k, mode, P["gw"].data_ptr())]),
                                      slot, wa), ()))
            self._grad_done.append((lay.name, P["gw"].data_ptr()))
            # data gradient
            if need_dx:
                xg = lay.x.grad
                acc = 1 if lay.x.gw else 0
                lay.x.gw = True
                for c in lay.x.children:
                    c.gw = True
                lazy = lay.x.lazy                    # pending residual gradient: fused as the accumulation source
                lay.x.lazy = None
                lay.dgrad_args = []
                dyA = Act(scratch, lay.x.B, lay.Ho, lay.Wo, lay.cout)
                for (wd, py, px, tht, kh0, khs, dh0, kw0, kws, dw0) in lay.wd:
                    th, tw = (tht, tht) if isinstance(tht, int) else tht
                    a = ConvArgs()
                    a.zeros = _lib.zero_page(self.dev).data_ptr()
                    a.inp, a.w, a.out = dyA.ptr, wd.data_ptr(), xg.ptr
                    a.B, a.Hin, a.Win, a.ldin = dyA.B, dyA.H, dyA.W, dyA.ld
                    if lay.ss == 1:
                        a.Hg, a.Wg, a.osy, a.osx, a.ooy, a.oox = lay.x.H, lay.x.W, 1, 1, 0, 0
                    else:
                        a.Hg, a.Wg, a.osy, a.osx, a.ooy, a.oox = lay.x.H // 2, lay.x.W // 2, 2, 2, py, px
                    a.sy, a.sx = 1, 1
                    a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = th, tw, dh0, -1, dw0, -1
                    a.Cin, a.K, a.Kp = lay.cout, th * tw * lay.cout, wd.shape[1]
                    a.N, a.M = lay.x.C, lay.x.B * a.Hg * a.Wg
                    a.Hout, a.Wout, a.ldout = lay.x.H, lay.x.W, xg.ld
                    a.epi, a.act, a.accumulate = EPI_DGRAD, ACT_NONE, acc
                    if lazy is not None:
                        assert acc == 1
                        a.res, a.ldres = lazy.ptr, lazy.ld
                    a.Np = wd.shape[0]
                    lay.dgrad_args.append(a)
                if len(lay.dgrad_args) > 1:
                    # stride 2: the parity classes read the same dy -- one launch with their tiles interleaved (y5m_conv_multi)
                    arr = (ConvArgs * len(lay.dgrad_args))()
                    for i, a in enumerate(lay.dgrad_args):
                        ctypes.memmove(ctypes.byref(arr[i]), ctypes.byref(a), ctypes.sizeof(ConvArgs))
                    lay.dgrad_multi = arr
                    ops.append((_kind(lambda arr=arr: _lib.check(L.y5m_conv_multi(arr, len(arr), dt, st()), "y5m_conv_multi(dgrad)"),
                                      "conv_igemm", self._conv_traffic(lay.dgrad_args)), ()))
                else:
                    for a in lay.dgrad_args:
                        ops.append((_kind(lambda a=a: _lib.check(L.y5m_conv(ctypes.byref(a), dt, st()), "y5m_conv(dgrad)"), "conv_igemm",
                                          self._conv_traffic([a])), ()))
                self._written(lay.x)
            return ops
        self._bwd_stack.append(backward)

    # ---- gradient-writer bookkeeping (plan time): who finishes d(activation)? ---------------------------
    @staticmethod
    def _root(act):
        return act.parent if act.parent is not None else act

    def _consume(self, act):
        """forward build: one more op whose backward will write (part of) act's gradient"""
        if act is not None and act.grad is not None:
            self._root(act).n_cons += 1

    def _is_last_writer(self, act):
        r = self._root(act)
        return r.n_written + 1 == r.n_cons

    def _written(self, act):
        self._root(act).n_written += 1

    def _bn_backward_op(self, lay, P, dz, scratch_ptr, lddy):
        """the BatchNorm + SiLU backward launch list entry of one CBL: reduce + apply (dy into the scratch slot)"""
        L, dt, st = self.L, self.dtype, _lib.stream_ptr
        bn = lay.bn
        if self.fuse_b:
            accp = self.gw.data_ptr() + 4 * self._accb_base + 8 * lay.accb_off       # zeroed with gw at the start of the pass
            return (_kind(lambda: _lib.check(
                L.y5m_bn_bwd_fused(dz.ptr, dz.ld, lay.y_ptr, lay.y_ld, bn[0].data_ptr(), bn[1].data_ptr(), bn[2].data_ptr(),
                                   bn[3].data_ptr(), lay.M, lay.cout, ACT_SILU, _lib.ptr(P["gg"]), _lib.ptr(P["gb"]), 0,
                                   scratch_ptr, lddy, accp, dt, st()), "y5m_bn_bwd_fused"), "bn_bwd(reduce + apply)",
                          (4 * lay.M * lay.cout * self.esz, lay.M * lay.cout * self.esz, 16 * lay.cout, 8 * lay.cout)), ())
        return (_kind(lambda: _lib.check(
            L.y5m_bn_bwd(dz.ptr, dz.ld, lay.y_ptr, lay.y_ld, bn[0].data_ptr(), bn[1].data_ptr(), bn[2].data_ptr(),
                         bn[3].data_ptr(), lay.M, lay.cout, ACT_SILU, _lib.ptr(P["gg"]), _lib.ptr(P["gb"]), 0, scratch_ptr,
                         lddy, _lib.ptr(self.bnws), self._bnws_bytes, dt, st()), "y5m_bn_bwd"), "bn_bwd(reduce + apply)",
                      (4 * lay.M * lay.cout * self.esz, lay.M * lay.cout * self.esz, 16 * lay.cout, 8 * lay.cout)), ())

    def _bwd_pw_ops(self, x, y_ptr, y_ld, wd, segs, M, N):
        """launch list entries of the fused pointwise backward (y5m_bwd_pw) of one 1x1 CBL -- or of a merged C3 pair: `segs` is
        [(layer, flat-buffer views P)] side by side on the N output channels of the launch. Per segment the BatchNorm
        reduction (phase 1 of y5m_bn_bwd_fused_phase) fills the layer's accumulator rows, then ONE launch forms dy in
        registers, writes d(x) (store / accumulate / lazy residual source, same bookkeeping as the separate data gradient)
        and adds dW straight into the flat gradient buffer. Returns None when the arguments do not qualify."""
        from ._lib import BwdPwArgs
        L, dt, st = self.L, self.dtype, _lib.stream_ptr
        a = BwdPwArgs()
        xg = x.grad
        a.y, a.x, a.wd, a.dx = y_ptr, x.ptr, wd.data_ptr(), xg.ptr
        a.M, a.ldy, a.ldx, a.Kp, a.lddx, a.lddw = M, y_ld, x.ld, wd.shape[1], xg.ld, x.C
        a.N, a.C, a.act, a.nseg = N, x.C, ACT_SILU, len(segs)
        c0 = 0
        for i, (lay, P) in enumerate(segs):
            sg, dz, bn = a.seg[i], lay.z.grad, lay.bn
            sg.c0, sg.cn, sg.dz, sg.lddz = c0, lay.cout, dz.ptr, dz.ld
            sg.acc = self.gw.data_ptr() + 4 * self._accb_base + 8 * lay.accb_off
            sg.scale, sg.shift, sg.mean, sg.invstd = (bn[k].data_ptr() for k in range(4))
            sg.dgamma, sg.dbeta, sg.dw = P["gg"].data_ptr(), P["gb"].data_ptr(), P["gw"].data_ptr()
            c0 += lay.cout
        # the launch's accumulation mode and lazy residual source are part of what the library checks: fill them in from a
        # PEEK at the plan-time flags, and commit the bookkeeping only once the launch is known to qualify
        acc = 1 if x.gw else 0
        lazy = x.lazy
        if lazy is not None:
            assert acc == 1
            a.res, a.ldres = lazy.ptr, lazy.ld
        a.accumulate = acc
        if M < self.fused_pw_min_m or not L.y5m_bwd_pw_eligible(ctypes.byref(a), dt):
            return None
        ops = []
        for lay, P in segs:
            ops.extend(self._flush_lazy(lay.z))
        x.gw = True
        for c in x.children:
            c.gw = True
        x.lazy = None
        for i, (lay, P) in enumerate(segs):
            sg = a.seg[i]
            def reduce(lay=lay, sg=sg):
                bn, dz = lay.bn, lay.z.grad
                _lib.check(L.y5m_bn_bwd_fused_phase(dz.ptr, dz.ld, lay.y_ptr, lay.y_ld, bn[0].data_ptr(), bn[1].data_ptr(),
                                                    bn[2].data_ptr(), bn[3].data_ptr(), M, lay.cout, ACT_SILU, None, None, 0,
                                                    None, 0, sg.acc, dt, st(), 1), "y5m_bn_bwd_fused_phase(reduce)")
            ops.append((_kind(reduce, "bn_reduce", (2 * M * lay.cout * self.esz, 0, 12 * lay.cout, 0)), ()))
            self._grad_done.append((lay.name, P["gw"].data_ptr()))
        def fused(a=a):
            _lib.check(L.y5m_bwd_pw(ctypes.byref(a), dt, st()), "y5m_bwd_pw")
        fused.bp = a
        # dz and y (N channels each) and x read, dx written (+ its accumulation source read); the transposed weights read,
        # dW added into the flat gradient
        ops.append((_kind(fused, "bwd_pw", (M * (2 * N + x.C * (2 if (a.accumulate or a.res) else 1)) * self.esz, M * x.C * self.esz,
                                            N * x.C * self.esz, N * x.C * 4)), ()))
        self._direct_wgrads += 1
        self._written(x)
        return ops

    def _bwd_stem_ops(self, lay, P):
        """launch list entries of the fused stem backward (y5m_bwd_stem): the BatchNorm reduction into the layer's accumulator
        rows, ONE launch that forms dy in registers and adds the 9-tap weight gradient into the packed buffer, and the unpack
        into the reference layout -- all on the main stream (the stem is the last unit of the backward pass: a forked weight
        gradient would run alone behind it). Returns None when the arguments do not qualify."""
        from ._lib import BwdStemArgs
        L, dt, st = self.L, self.dtype, _lib.stream_ptr
        if self.tdt != torch.bfloat16:
            return None
        dz, bn, x = lay.z.grad, lay.bn, lay.x
        a = BwdStemArgs()
        a.dz, a.y, a.x = dz.ptr, lay.y_ptr, x.ptr
        a.dwgt = self.gw.data_ptr() + 4 * lay.gw_off
        a.B, a.H, a.W, a.lddz, a.ldy, a.ldx, a.lddw = x.B, x.H, x.W, dz.ld, lay.y_ld, x.ld, lay.ldgw
        a.N, a.C, a.act = lay.cout, x.C, ACT_SILU
        a.acc = self.gw.data_ptr() + 4 * self._accb_base + 8 * lay.accb_off
        a.scale, a.shift, a.mean, a.invstd = (bn[k].data_ptr() for k in range(4))
        a.dgamma, a.dbeta = P["gg"].data_ptr(), P["gb"].data_ptr()
        if lay.Ho != x.H or lay.Wo != x.W or not L.y5m_bwd_stem_eligible(ctypes.byref(a)):
            return None
        ops = list(self._flush_lazy(lay.z))

        def reduce():
            _lib.check(L.y5m_bn_bwd_fused_phase(dz.ptr, dz.ld, lay.y_ptr, lay.y_ld, bn[0].data_ptr(), bn[1].data_ptr(),
                                                bn[2].data_ptr(), bn[3].data_ptr(), lay.M, lay.cout, ACT_SILU, None, None, 0,
                                                None, 0, a.acc, dt, st(), 1), "y5m_bn_bwd_fused_phase(reduce)")

        def fused():
            _lib.check(L.y5m_bwd_stem(ctypes.byref(a), st()), "y5m_bwd_stem")
        fused.bs = a

        def unpack():
            _lib.check(L.y5m_unpack_wgrad(a.dwgt, lay.cout, lay.cin_real, lay.k, lay.k, 2, lay.ldgw, P["gw"].data_ptr(), st()),
                       "y5m_unpack_wgrad")
        dw = lay.cout * lay.ldgw * 4
        ops += [(_kind(reduce, "bn_reduce", (2 * lay.M * lay.cout * self.esz, 0, 12 * lay.cout, 0)), ()),
                (_kind(fused, "bwd_stem", (lay.M * (2 * lay.cout + x.C) * self.esz, 0, 20 * lay.cout, dw)), ()),
                (_kind(unpack, "unpack", (0, 0, dw, dw)), ())]
        self._grad_done.append((lay.name, P["gw"].data_ptr()))
        return ops

    def _wgrad_ops(self, lay, wa, unpacks):
        """launch closures [wgrad, unpack...] of one weight gradient. unpacks: [(row offset, Cout, Cin, k, mode,
        dst flat-gradient pointer)]. 3x3 layers accumulate with atomics into the zeroed packed buffer self.gw and are
        unpacked into the reference layout; a 1x1 layer's packed gradient IS the reference layout."""
        L, dt, st = self.L, self.dtype, _lib.stream_ptr
        fs = [lambda: _lib.check(L.y5m_wgrad(ctypes.byref(wa), dt, st()), "y5m_wgrad")]
        if (self.wgrad_direct and wa.th == 1 and wa.tw == 1 and len(unpacks) == 1 and unpacks[0][0] == 0
                and unpacks[0][1] == wa.N and unpacks[0][2] == wa.lddw and unpacks[0][4] == 0):
            # a 1x1 layer's packed gradient [Cout][Cin] IS the reference layout [Cout][Cin][1][1]: accumulate straight
            # into the flat gradient buffer (zeroed at the start of the backward list), no unpack launch
            wa.dwgt = unpacks[0][5]
            self._direct_wgrads += 1
            return fs
        for (row0, cout, cin, k, mode, dst) in unpacks:
            fs.append(lambda row0=row0, cout=cout, cin=cin, k=k, mode=mode, dst=dst: _lib.check(
                L.y5m_unpack_wgrad(wa.dwgt + 4 * row0 * wa.lddw, cout, cin, k, k, mode, wa.lddw, dst, st()),
                "y5m_unpack_wgrad"))
        return fs

    def _flush_lazy(self, act):
        """ops that materialise a pending lazy gradient of `act` (grad = lazy source): a copy. Called by every
        gradient writer / reader that cannot take the source as a fused accumulation operand."""
        if act is None or act.lazy is None:
            return []
        L, dt, st = self.L, self.dtype, _lib.stream_ptr
        src, g = act.lazy, act.grad
        act.lazy = None
        nb = act.B * act.H * act.W * act.C * self.esz
        return [(_kind(lambda: _lib.check(L.y5m_add(src.ptr, src.ld, g.ptr, g.ld, act.B * act.H * act.W, act.C, 0, dt, st()),
                                          "y5m_add(lazy)"), "add", (nb, nb, 0, 0)), ())]

    def _c3(self, name, x, cout, width, depth, backbone, dest=None):
        """reference model.py:54-92"""
        c_ = int(width * x.C)
        merged = self.training and self.merge_c3
        if self.merge_c3 and not self.training:
            # eval: [seq output | c_skipped output | c1 output] side by side in ONE buffer; c_skipped + c1 are one folded conv
            # that writes channels [c_, 3 c_), c_out reads channels [0, 2 c_) -- x is read once, one launch less per C3
            wide = self._new_act(x.B, x.H, x.W, 3 * c_)
            cat, s0 = wide.slice(0, 2 * c_), wide.slice(0, c_)
            t = self._cbl_pair_eval(f"{name}.c_skipped", f"{name}.c1", x, c_, wide.slice(c_, 2 * c_), wide.slice(2 * c_, c_))
        else:
            cat = self._new_act(x.B, x.H, x.W, 2 * c_)
            s0, s1 = cat.slice(0, c_), cat.slice(c_, c_)
            t = self._cbl_pair(f"{name}.c1", f"{name}.c_skipped", x, c_, s1) if merged else self._cbl(f"{name}.c1", x, c_, 1, 1, 0)
        for d in range(depth):
            last = d == depth - 1
            if backbone:
                u = self._cbl(f"{name}.seq.{d}.c1", t, c_, 1, 1, 0)
                t = self._cbl(f"{name}.seq.{d}.c2", u, c_, 3, 1, 1, dest=s0 if last else None, res=t)
            else:
                u = self._cbl(f"{name}.seq.{d}.0", t, c_, 1, 1, 0)
                t = self._cbl(f"{name}.seq.{d}.1", u, c_, 3, 1, 1, dest=s0 if last else None)
        if not merged and (self.training or not self.merge_c3):
            self._cbl(f"{name}.c_skipped", x, c_, 1, 1, 0, dest=s1)
        return self._cbl(f"{name}.c_out", cat, cout, 1, 1, 0, dest=dest)

    def _cbl_pair_eval(self, nameA, nameB, x, cout, dest2, destB):
        """eval mode: two 1x1 CBLs on the same input (C3's c_skipped and c1, reference model.py:77, :69) as ONE conv with
        N = 2*cout and the folded (scale, shift) of both side by side; `dest2` is the 2*cout-channel view both halves land
        in, `destB` its second half (returned: the input of the bottleneck chain)."""
        L, dt = self.L, self.dtype
        esz = 2 if self.tdt == torch.bfloat16 else 4
        N2, K = 2 * cout, x.C
        Kp = _rup(K, self.BK)
        Np2 = _rup(N2, L.y5m_conv_tile_n(N2))
        wf = torch.zeros((Np2, Kp), dtype=self.tdt, device=self.dev)
        bn2 = torch.zeros((2, N2), dtype=torch.float32, device=self.dev)      # scale, shift of both halves
        lay = _Layer()
        lay.name, lay.x, lay.res, lay.stem = nameA + "+" + nameB.rsplit(".", 1)[-1], x, None, False
        lay.cin_real, lay.cout, lay.k, lay.s, lay.p = x.C, N2, 1, 1, 0
        lay.kk, lay.ss, lay.pp, lay.M, lay.Ho, lay.Wo = 1, 1, 0, x.M, x.H, x.W
        lay.K, lay.Kp, lay.Np, lay.z, lay.wf, lay.bn = K, Kp, Np2, dest2, wf, bn2
        for name, off in ((nameA, 0), (nameB, cout)):
            P = self.model.pslices[name]
            rows = cout if off == 0 else Np2 - cout          # the second job also zero-fills the row padding
            self._call(self.pack, L.y5m_pack_weights, _lib.ptr(P["w"]), cout, x.C, 1, 1, 0, 0, 1, 1, 0, 1, 1,
                       ctypes.c_void_p(wf.data_ptr() + off * Kp * esz), rows, Kp, 0, dt)
            self._fold_jobs.append((P["g"].data_ptr(), P["b"].data_ptr(), P["rm"].data_ptr(), P["rv"].data_ptr(),
                                    bn2[0].data_ptr() + 4 * off, bn2[1].data_ptr() + 4 * off, cout))
        a = self._conv_args(x, wf, dest2.ptr, x.H, x.W, 1, 1, 0, N2, dest2.ld, EPI_AFFINE_ACT, Kp,
                            act=ACT_SILU, scale=bn2[0].data_ptr(), shift=bn2[1].data_ptr())
        lay.fwd_args = a
        self._run_conv(self.fwd, a)
        self.layers.append(lay)
        return destB

    def _cbl_pair(self, nameA, nameB, x, cout, destB):
        """C3's two 1x1 CBLs on the SAME input (reference model.py:69 c1 and :77 c_skipped) as ONE GEMM with
        N = 2*cout: x is read once by the forward conv and once by the weight gradient, and the data gradient
        is one K = 2*cout GEMM that writes dx once (instead of a second read-modify-write pass). BatchNorm
        statistics, parameters, activations and gradients stay per layer: the two halves of the merged raw
        output / dy buffers are (ptr, ld) views. Training mode only (eval folds BN into two plain convs)."""
        L, dt, st = self.L, self.dtype, _lib.stream_ptr
        esz = 2 if self.tdt == torch.bfloat16 else 4
        N2, M, K = 2 * cout, x.B * x.H * x.W, x.C
        Kp = _rup(K, self.BK)
        Np2 = _rup(N2, L.y5m_conv_tile_n(N2))
        destA = self._new_act(x.B, x.H, x.W, cout)
        wf = torch.zeros((Np2, Kp), dtype=self.tdt, device=self.dev)
        y2 = torch.zeros((M * N2,), dtype=self.tdt, device=self.dev)
        halves = []
        for name, off, dest in ((nameA, 0, destA), (nameB, cout, destB)):
            P = self.model.pslices[name]
            lay = _Layer()
            lay.name, lay.x, lay.res, lay.stem = name, x, None, False
            lay.cin_real, lay.cout, lay.k, lay.s, lay.p = x.C, cout, 1, 1, 0
            lay.kk, lay.ss, lay.pp, lay.M, lay.Ho, lay.Wo = 1, 1, 0, M, x.H, x.W
            lay.z, lay.off = dest, off
            lay.y_ptr, lay.y_ld = y2.data_ptr() + off * esz, N2
            lay.y_view = (y2, off, N2)
            dest.producer = lay
            lay.bn = torch.zeros((4, cout), dtype=torch.float32, device=self.dev)
            rows = cout if off == 0 else Np2 - cout          # the second job also zero-fills the row padding
            self._call(self.pack, L.y5m_pack_weights, _lib.ptr(P["w"]), cout, x.C, 1, 1, 0, 0, 1, 1, 0, 1, 1,
                       ctypes.c_void_p(wf.data_ptr() + off * Kp * esz), rows, Kp, 0, dt)
            halves.append((lay, P))
            self.layers.append(lay)
        halves[0][0].pair_buffers = (wf, y2)             # the launch descriptors hold raw pointers: keep the tensors alive
        self._consume(x)
        a = self._conv_args(x, wf, y2.data_ptr(), x.H, x.W, 1, 1, 0, N2, N2, EPI_RAW_STATS, Kp)
        halves[0][0].fwd_args = a
        acc_off = tiles_m = 0
        if self.fuse_f:
            acc_off = self._accf
            self._accf += self._slots * 2 * Np2
            self._acc_users.append((a, acc_off))
        else:
            tiles_m = L.y5m_conv_stats_rows(ctypes.byref(a), dt)
            self._stats_floats = max(self._stats_floats, tiles_m * 2 * Np2)
            self._stat_users.append(a)
        self._run_conv(self.fwd, a)
        for lay, P in halves:
            def finalize(lay=lay, P=P):
                bn = lay.bn
                _lib.check(L.y5m_bn_finalize(self.stats.data_ptr() + 4 * lay.off, tiles_m, Np2, cout, M, _lib.ptr(P["g"]),
                                             _lib.ptr(P["b"]), _lib.ptr(P["rm"]), _lib.ptr(P["rv"]), BN_MOMENTUM, BN_EPS,
                                             bn[0].data_ptr(), bn[1].data_ptr(), bn[2].data_ptr(), bn[3].data_ptr(), 1,
                                             _lib.ptr(self.finws), self.finws.numel(), st()), "y5m_bn_finalize")

            def apply(lay=lay):
                bn = lay.bn
                _lib.check(L.y5m_bn_act(y2.data_ptr() + lay.off * esz, N2, bn[0].data_ptr(), bn[1].data_ptr(), None, 0,
                                        lay.z.ptr, lay.z.ld, M, cout, ACT_SILU, dt, st()), "y5m_bn_act")

            def apply_fused(lay=lay, P=P):
                bn = lay.bn        # this half's channels start at column lay.off of the merged launch's accumulator rows
                _lib.check(L.y5m_bn_act_fused(y2.data_ptr() + lay.off * esz, N2, self.accf.data_ptr() + 8 * (acc_off + lay.off),
                                              Np2, M, _lib.ptr(P["g"]), _lib.ptr(P["b"]), _lib.ptr(P["rm"]), _lib.ptr(P["rv"]),
                                              BN_MOMENTUM, BN_EPS, 1, bn[0].data_ptr(), bn[1].data_ptr(), bn[2].data_ptr(),
                                              bn[3].data_ptr(), None, 0, lay.z.ptr, lay.z.ld, M, cout, ACT_SILU, dt, st()),
                           "y5m_bn_act_fused")
            for f_ in (apply, apply_fused):
                _traffic(f_, M * cout * self.esz, M * cout * self.esz, 8 * cout, 16 * cout)
            if self.fuse_f:
                self.fwd.append((apply_fused, ()))
            else:
                self.fwd.append((finalize, ()))
                self.fwd.append((apply, ()))
        # ---- backward: pushed at c1's place, i.e. executed after the whole C3 body, when both dz are final
        self._scratch_elems = max(self._scratch_elems, M * N2)
        self._bnws_bytes = max(self._bnws_bytes, L.y5m_bn_bwd_workspace_bytes(M, cout))
        for lay, _P in halves:
            lay.accb_off = self._accb
            self._accb += self._slots * 2 * cout
        gw_off = self._gw_floats
        self._gw_floats += N2 * K
        need_dx = x.grad is not None
        wd = None
        if need_dx:
            wd = torch.zeros((_rup(K, L.y5m_conv_tile_n(K)), _rup(N2, self.BK)), dtype=self.tdt, device=self.dev)
            halves[1][0].pair_buffers = (wd,)
            for lay, P in halves:
                cols = cout if lay.off == 0 else wd.shape[1] - cout
                self._call(self.pack, L.y5m_pack_weights, _lib.ptr(P["w"]), cout, K, 1, 1, 1, 0, 1, 1, 0, 1, 1,
                           ctypes.c_void_p(wd.data_ptr() + lay.off * esz), wd.shape[0], cols, 0, dt, wd.shape[1])

        def backward():
            if self.fused_pw and self.fuse_b and need_dx:
                fops = self._bwd_pw_ops(x, y2.data_ptr(), N2, wd, halves, M, N2)
                if fops is not None:
                    return fops
            ops = []
            slot = self._next_slot()
            scratch = self.scratch2[slot]
            ops.append((self._join_op(slot), ()))
            for lay, P in halves:
                ops.extend(self._flush_lazy(lay.z))
                ops.extend(_as_list(self._bn_backward_op(lay, P, lay.z.grad, scratch.data_ptr() + lay.off * esz, N2)))
            wa = WgradArgs()
            wa.zeros = _lib.zero_page(self.dev).data_ptr()
            wa.dy, wa.x, wa.dwgt = scratch.data_ptr(), x.ptr, self.gw.data_ptr() + 4 * gw_off
            wa.B, wa.Hin, wa.Win, wa.ldx = x.B, x.H, x.W, x.ld
            wa.Hg, wa.Wg, wa.sy, wa.sx = x.H, x.W, 1, 1
            wa.th, wa.tw, wa.dh0, wa.dhs, wa.dw0, wa.dws = 1, 1, 0, 1, 0, 1
            wa.C, wa.N, wa.M, wa.lddy, wa.lddw, wa.ksplit = K, N2, M, N2, K, 0
            fs = self._wgrad_ops(halves[0][0], wa, [(lay.off, cout, K, 1, 0, P["gw"].data_ptr()) for lay, P in halves])
            ops.append((self._side_op(fs, slot, wa), ()))
            for lay, P in halves:
                self._grad_done.append((lay.name, P["gw"].data_ptr()))
            if need_dx:
                pre = self._flush_lazy(x)
                ops.extend(pre)
                acc = 1 if x.gw else 0
                x.gw = True
                for c in x.children:
                    c.gw = True
                g = ConvArgs()
                g.zeros = _lib.zero_page(self.dev).data_ptr()
                g.inp, g.w, g.out = scratch.data_ptr(), wd.data_ptr(), x.grad.ptr
                g.B, g.Hin, g.Win, g.ldin = x.B, x.H, x.W, N2
                g.Hg, g.Wg, g.sy, g.sx = x.H, x.W, 1, 1
                g.th, g.tw, g.dh0, g.dhs, g.dw0, g.dws = 1, 1, 0, -1, 0, -1
                g.Cin, g.K, g.Kp, g.N, g.M = N2, N2, wd.shape[1], K, M
                g.Hout, g.Wout, g.ldout, g.osy, g.osx, g.ooy, g.oox = x.H, x.W, x.grad.ld, 1, 1, 0, 0
                g.epi, g.act, g.accumulate, g.Np = EPI_DGRAD, ACT_NONE, acc, wd.shape[0]
                self._written(x)
                ops.append((_kind(lambda g=g: _lib.check(L.y5m_conv(ctypes.byref(g), dt, st()), "y5m_conv(pair dgrad)"),
                                  "conv_igemm", self._conv_traffic([g])), ()))
            return ops
        self._bwd_stack.append(backward)
        return destA

    def _sppf(self, name, x, cout):
        """reference model.py:96-112"""
        L, dt, st = self.L, self.dtype, _lib.stream_ptr
        c_ = x.C // 2
        cat = self._new_act(x.B, x.H, x.W, 4 * c_)
        sl = [cat.slice(i * c_, c_) for i in range(4)]
        self._cbl(f"{name}.c1", x, c_, 1, 1, 0, dest=sl[0])
        sppfws = torch.zeros((L.y5m_sppf_pool_workspace_bytes(x.B, x.H, x.W, c_),), dtype=torch.uint8, device=self.dev)
        self.fwd.append((_kind(lambda: _lib.check(L.y5m_sppf_pool(sl[0].ptr, cat.ld, x.B, x.H, x.W, c_, sl[1].ptr, sl[2].ptr,
                                                                 sl[3].ptr, _lib.ptr(sppfws), sppfws.numel(), dt, st()),
                                                  "y5m_sppf_pool"), "pool", (x.M * c_ * self.esz, 3 * x.M * c_ * self.esz, 0, 0)), ()))
        if self.training:
            poolws = torch.zeros((L.y5m_maxpool5_bwd_workspace_bytes(x.B, x.H, x.W, c_),), dtype=torch.uint8, device=self.dev)
            for _ in range(3):
                self._consume(cat)            # the three pool backward passes accumulate into slices of d(cat)

            def backward():
                for _ in range(3):
                    self._written(cat)
                g = [s.grad for s in sl]
                # g2 += bwd(p2; g3) ; g1 += bwd(p1; g2) ; g0 += bwd(x; g1)   (cascade of model.py:108-110) as ONE call: one launch
                # where the library's LDS-tiled form applies (Y5M_POOL_TILE), else the three per-level launches inside it
                tiled = bool(L.y5m_sppf_pool_tiled(x.H, x.W, c_, dt))
                nb = x.M * c_ * self.esz

                def pool_bwd():
                    _lib.check(L.y5m_sppf_pool_bwd(sl[0].ptr, sl[1].ptr, sl[2].ptr, cat.ld, g[0].ptr, g[1].ptr, g[2].ptr, g[3].ptr,
                                                   g[0].ld, x.B, x.H, x.W, c_, _lib.ptr(poolws), poolws.numel(), dt, st()),
                               "y5m_sppf_pool_bwd")
                # tiled: z0..z2, g3 and the three accumulation targets read, the three targets written; per level: z, g and the
                # target read, the target written
                return [(_kind(pool_bwd, "pool", ((7 if tiled else 9) * nb, 3 * nb, 0, 0)), ())]
            self._bwd_stack.append(backward)
        return self._cbl(f"{name}.c_out", cat, cout, 1, 1, 0)

    def _upsample_into(self, x, dst):
        """reference model.py:225 (nearest x2), written straight into its concat slice"""
        L, dt, st = self.L, self.dtype, _lib.stream_ptr
        self.fwd.append((_kind(lambda: _lib.check(L.y5m_upsample2x(x.ptr, x.ld, x.B, x.H, x.W, x.C, dst.ptr, dst.ld, dt, st()),
                                                  "y5m_upsample2x"), "upsample", (x.M * x.C * self.esz, 4 * x.M * x.C * self.esz, 0, 0)), ()))
        if self.training:
            self._consume(x)

            def backward():
                pre = self._flush_lazy(x)
                self._written(x)
                acc = 1 if x.gw else 0
                x.gw = True
                return pre + [(_kind(lambda: _lib.check(L.y5m_upsample2x_bwd(dst.grad.ptr, dst.grad.ld, x.B, x.H, x.W, x.C,
                                                                       x.grad.ptr, x.grad.ld, acc, dt, st()),
                                                  "y5m_upsample2x_bwd"), "upsample",
                                     ((4 + acc) * x.M * x.C * self.esz, x.M * x.C * self.esz, 0, 0)), ())]
            self._bwd_stack.append(backward)

    def _head(self, i, x):
        """reference model.py:162-175: 1x1 conv + bias, stored permuted as (B,naxs,ny,nx,5+nc) f32"""
        L, dt, st = self.L, self.dtype, _lib.stream_ptr
        P = self.model.pslices[f"head.out_convs.{i}"]
        N = self.naxs * self.nch
        lay = _Layer()
        lay.name, lay.x = f"head.out_convs.{i}", x
        Kp = _rup(x.C, self.BK)
        Np = _rup(N, L.y5m_conv_tile_n(N))
        lay.wf = torch.zeros((Np, Kp), dtype=self.tdt, device=self.dev)
        out = torch.zeros((x.B, self.naxs, x.H, x.W, self.nch), dtype=torch.float32, device=self.dev)
        lay.out = out
        self._call(self.pack, L.y5m_pack_weights, _lib.ptr(P["w"]), N, x.C, 1, 1, 0, 0, 1, 1, 0, 1, 1, _lib.ptr(lay.wf),
                   Np, Kp, 0, dt)
        a = self._conv_args(x, lay.wf, out.data_ptr(), x.H, x.W, 1, 1, 0, N, N, EPI_HEAD, Kp,
                            scale=P["b"].data_ptr(), naxs=self.naxs, nch=self.nch)
        lay.fwd_args = a
        self._run_conv(self.fwd, a)
        if self.training:
            self._consume(x)
            ldp = _rup(N, 16)                      # 255 -> 256: 16-byte rows for the MFMA operand loads
            M = x.M
            lay.gout = torch.zeros_like(out)       # d(loss)/d(logits), filled by the loss / autograd
            self._scratch_elems = max(self._scratch_elems, M * ldp)
            lay.gw_off = self._gw_floats
            lay.ldgw = x.C
            self._gw_floats += ldp * x.C
            wd = torch.zeros((_rup(x.C, L.y5m_conv_tile_n(x.C)), _rup(ldp, self.BK)), dtype=self.tdt, device=self.dev)
            self._call(self.pack, L.y5m_pack_weights, _lib.ptr(P["w"]), N, x.C, 1, 1, 1, 0, 1, 1, 0, 1, 1, _lib.ptr(wd),
                       wd.shape[0], wd.shape[1], ldp, dt)

            def backward(lay=lay, x=x, P=P, wd=wd, ldp=ldp, M=M, N=N, i=i):
                ops = []
                slot = self._next_slot()
                scratch = self.scratch2[slot]
                ops.append((self._join_op(slot), ()))
                def pack(lay=lay, x=x, P=P, scratch=scratch, ldp=ldp, i=i):
                    # head_owner: set by NativeTrainStep when lay.gout was written by y5m_compute_loss (zero outside the
                    # objectness channel and the target rows): the sparse pack reads ~1/85 of it
                    ow = self.head_owner[i] if self.head_owner is not None else None
                    if ow:
                        _lib.check(L.y5m_head_grad_pack_sparse(_lib.ptr(lay.gout), ctypes.c_void_p(ow[0]), ctypes.c_void_p(ow[1]),
                                                               ctypes.c_void_p(ow[2]), ctypes.c_void_p(ow[3]), ow[4],
                                                               x.B, self.naxs, x.H, x.W, self.nch, _lib.ptr(scratch), ldp,
                                                               _lib.ptr(P["gb"]), dt, st()), "y5m_head_grad_pack_sparse")
                    else:
                        _lib.check(L.y5m_head_grad_pack(_lib.ptr(lay.gout), x.B, self.naxs, x.H, x.W, self.nch, _lib.ptr(scratch),
                                                        ldp, _lib.ptr(P["gb"]), dt, st()), "y5m_head_grad_pack")
                # sparse form (the loss wrote the target rows + a compact objectness plane): one f32 per cell read; dense form: the
                # whole f32 gradient; either way the bf16 operand rows of the two GEMMs behind it are written in full
                # (two figures: which one applies is decided where the launch decides -- `head_owner` set or not -- when the plan is
                #  asked, algorithmic_bytes below)
                pack.traffic_sparse = (M * self.naxs * 4, M * ldp * self.esz, 0, 4 * N)
                ops.append((_kind(pack, "head_pack", (M * self.naxs * 4 * self.nch, M * ldp * self.esz, 0, 4 * N)), ()))
                wa = WgradArgs()
                wa.zeros = _lib.zero_page(self.dev).data_ptr()
                wa.dy, wa.x, wa.dwgt = scratch.data_ptr(), x.ptr, self.gw.data_ptr() + 4 * lay.gw_off
                wa.B, wa.Hin, wa.Win, wa.ldx = x.B, x.H, x.W, x.ld
                wa.Hg, wa.Wg, wa.sy, wa.sx = x.H, x.W, 1, 1
                wa.th, wa.tw, wa.dh0, wa.dhs, wa.dw0, wa.dws = 1, 1, 0, 1, 0, 1
                wa.C, wa.N, wa.M, wa.lddy, wa.lddw, wa.ksplit = x.C, ldp, M, ldp, x.C, 0
                lay.wgrad_args = wa
                ops.append((self._side_op(self._wgrad_ops(lay, wa, [(0, N, x.C, 1, 0, P["gw"].data_ptr())]), slot, wa), ()))
                self._grad_done.append((lay.name, P["gw"].data_ptr()))
                self._written(x)
                acc = 1 if x.gw else 0
                x.gw = True
                a = ConvArgs()
                a.zeros = _lib.zero_page(self.dev).data_ptr()
                a.inp, a.w, a.out = scratch.data_ptr(), wd.data_ptr(), x.grad.ptr
                a.B, a.Hin, a.Win, a.ldin = x.B, x.H, x.W, ldp
                a.Hg, a.Wg, a.sy, a.sx = x.H, x.W, 1, 1
                a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = 1, 1, 0, -1, 0, -1
                a.Cin, a.K, a.Kp, a.N, a.M = ldp, ldp, wd.shape[1], x.C, M
                a.Hout, a.Wout, a.ldout, a.osy, a.osx, a.ooy, a.oox = x.H, x.W, x.grad.ld, 1, 1, 0, 0
                a.epi, a.act, a.accumulate, a.Np = EPI_DGRAD, ACT_NONE, acc, wd.shape[0]
                lay.dgrad_args = [a]
                ops.append((_kind(lambda: _lib.check(L.y5m_conv(ctypes.byref(a), dt, st()), "y5m_conv(head dgrad)"), "conv_igemm",
                                  self._conv_traffic([a])), ()))
                return ops
            self._bwd_stack.append(backward)
        self.heads.append(lay)
        return out

    # ------------------------------------------------------------------ whole-model plan
    def _build(self):
        L, dt = self.L, self.dtype
        B, H, W = self.B, self.H, self.W
        self.pack, self.heads, self._stat_users = [], [], []
        self._fold_jobs = []
        self._gw_floats = 0
        self._grad_done = []
        first_out = self.model.first_out
        backbone, neck = blocks(first_out)
        # input: NCHW f32 images -> space-to-depth NHWC (no gradient)
        self.x_in = torch.zeros((B, 3, H, W), dtype=torch.float32, device=self.dev)
        s2d = self._new_act(B, H // 2, W // 2, 16, need_grad=False)
        self.fwd.append((_kind(lambda: _lib.check(L.y5m_s2d_input(_lib.ptr(self.x_in), B, H, W, s2d.ptr, dt, _lib.stream_ptr()),
                                                  "y5m_s2d_input"), "input", (B * 3 * H * W * 4, B * (H // 2) * (W // 2) * 16 * self.esz, 0, 0)), ()))
        f = first_out
        # concat buffers of the PANet joins (model.py:226, :230); producers write their slices in place
        cat1 = self._new_act(B, H // 16, W // 16, 16 * f)     # [up(neck0) | backbone6]
        cat2 = self._new_act(B, H // 8, W // 8, 8 * f)        # [up(neck2) | backbone4]
        cat3 = self._new_act(B, H // 16, W // 16, 8 * f)      # [neck4 | neck2]
        cat4 = self._new_act(B, H // 32, W // 32, 16 * f)     # [neck6 | neck0]
        c1a, c1b = cat1.slice(0, 8 * f), cat1.slice(8 * f, 8 * f)
        c2a, c2b = cat2.slice(0, 4 * f), cat2.slice(4 * f, 4 * f)
        c3a, c3b = cat3.slice(0, 4 * f), cat3.slice(4 * f, 4 * f)
        c4a, c4b = cat4.slice(0, 8 * f), cat4.slice(8 * f, 8 * f)
        dests = {4: c2b, 6: c1b}
        x = s2d
        for idx, (kind, a) in enumerate(backbone):
            name = f"backbone.{idx}"
            if kind == "cbl":
                x = self._cbl(name, x, a["cout"], a["k"], a["s"], a["p"], stem=(idx == 0))
            elif kind == "c3":
                x = self._c3(name, x, a["cout"], a["width"], a["depth"], a["backbone"], dest=dests.get(idx))
            else:
                x = self._sppf(name, x, a["cout"])
        n0 = self._cbl("neck.0", x, 8 * f, 1, 1, 0, dest=c4b)
        self._upsample_into(n0, c1a)
        n1 = self._c3("neck.1", cat1, 8 * f, 0.25, 2, False)
        n2 = self._cbl("neck.2", n1, 4 * f, 1, 1, 0, dest=c3b)
        self._upsample_into(n2, c2a)
        n3 = self._c3("neck.3", cat2, 4 * f, 0.25, 2, False)
        self._cbl("neck.4", n3, 4 * f, 3, 2, 1, dest=c3a)
        n5 = self._c3("neck.5", cat3, 8 * f, 0.5, 2, False)
        self._cbl("neck.6", n5, 8 * f, 3, 2, 1, dest=c4a)
        n7 = self._c3("neck.7", cat4, 16 * f, 0.5, 2, False)
        self.outs = [self._head(0, n3), self._head(1, n5), self._head(2, n7)]
        self._batch_packs()
        # shared scratch
        self.stats = torch.zeros((max(self._stats_floats, 1),), dtype=torch.float32, device=self.dev)
        self.finws = torch.zeros((L.y5m_bn_finalize_workspace_bytes(16 * first_out + 96),), dtype=torch.uint8, device=self.dev)
        for a in self._stat_users:
            a.stats = self.stats.data_ptr()
        if self._acc_users:
            # forward accumulator rows of every layer, zeroed by ONE fill at the start of the pass
            self.accf = torch.zeros((self._accf,), dtype=torch.float64, device=self.dev)
            for a, off in self._acc_users:
                a.bn_acc = self.accf.data_ptr() + 8 * off
            self.fwd.insert(0, (_kind(lambda: self.accf.zero_(), "fill", (0, 0, 0, 8 * self._accf)), ()))
        if self.training:
            self.scratch2 = [torch.zeros((self._scratch_elems,), dtype=self.tdt, device=self.dev) for _ in range(self.nslots)]
            self.scratch = self.scratch2[0]
            self.bnws = torch.zeros((self._bnws_bytes,), dtype=torch.uint8, device=self.dev)
            # (the backward accumulator rows of the BatchNorm reductions ride behind the packed weight gradients: one fill)
            self._accb_base = (self._gw_floats + 1) // 2 * 2
            self.gw = torch.zeros((self._accb_base + (2 * self._accb if self.fuse_b else 0),), dtype=torch.float32, device=self.dev)
            # expand the backward stack in reverse order; plan-time gradient-written flags
            self.bwd.append((_kind(lambda: self.gw.zero_(), "fill", (0, 0, 0, 4 * self.gw.numel())), ()))
            self.bwd.append((_kind(lambda: self.model.flat_grads.zero_() if self._direct_wgrads else None, "fill",
                                   (0, 0, 0, 4 * self.model.flat_grads.numel())), ()))
            # bwd_marks[i] = (op index after which unit i's parameter gradients are final, unit name,
            # device address of its weight gradient inside the flat buffer) -- in backward order. The marks are taken from the
            # FINAL order of the list (after the reorder below): a unit's mark is one past the last of ITS OWN ops, wherever
            # the reorder put them, so a cut (grad_cuts) can never fall in front of a launch that still writes the unit's
            # gradients.
            units = []                                     # (ops of the unit, [(name, address)])
            for mk in reversed(self._bwd_stack):
                n0, i0 = len(self._grad_done), len(self.bwd)
                self.bwd.extend(mk())
                if len(self._grad_done) > n0:
                    units.append((self.bwd[i0:], self._grad_done[n0:]))
            for sl in range(self.nslots):
                self.bwd.append((self._join_op(sl, final=True), ()))
            if os.environ.get("Y5M_WGRAD_AFTER_DGRAD", "1") == "1":
                # fork the weight gradient AFTER the layer's data-gradient launches: it then runs next to the following
                # layer's HBM-bound BatchNorm backward instead of next to the MFMA-bound data gradient it would only
                # share the matrix cores with (measured 30.83 -> 30.14 ms/step). The dy slot it reads is not rewritten
                # before the join of that slot, which precedes the BatchNorm backward two layers further on.
                out, i = [], 0
                while i < len(self.bwd):
                    op = self.bwd[i]
                    if getattr(op[0], "kind", None) == "wgrad":
                        j = i + 1
                        while j < len(self.bwd) and getattr(self.bwd[j][0], "kind", None) == "conv_igemm":
                            out.append(self.bwd[j])
                            j += 1
                        out.append(op)
                        i = j
                    else:
                        out.append(op)
                        i += 1
                self.bwd = out
            pos = {id(op[0]): i for i, op in enumerate(self.bwd)}
            assert len(pos) == len(self.bwd), "launch closures must be distinct objects"
            self.bwd_marks = []
            for ops, done in units:
                k = 1 + max(pos[id(op[0])] for op in ops)
                for name, addr in done:
                    self.bwd_marks.append((k, name, addr))
            self._check_grad_writers(units, pos)

    # kinds of launch-list entries that write parameter gradients (weights, gamma, beta, head bias)
    _GRAD_WRITER_KINDS = ("wgrad", "bwd_pw", "bwd_stem", "unpack", "bn_bwd(reduce + apply)", "bn_reduce", "head_pack")

    def _check_grad_writers(self, units, pos):
        """plan-time invariant of the overlapped gradient exchange: every launch that writes a unit's parameter gradients sits
        in front of that unit's mark, and the marks do not decrease along the backward order (a bucket is exchanged after the
        cut that follows its last unit)."""
        marks = {}
        for k, name, _ in self.bwd_marks:
            marks[name] = k
        last = 0
        for ops, done in units:
            k = marks[done[0][0]]
            assert all(marks[name] == k for name, _ in done)
            assert k >= last, ("unit marks must not decrease", done[0][0], k, last)
            last = k
            for op in ops:
                if getattr(op[0], "kind", None) in self._GRAD_WRITER_KINDS:
                    assert pos[id(op[0])] < k, ("gradient writer behind its unit's mark", done[0][0], op[0].kind)
        n_writers = sum(1 for op in self.bwd if getattr(op[0], "kind", None) in self._GRAD_WRITER_KINDS)
        n_in_units = sum(1 for ops, _ in units for op in ops if getattr(op[0], "kind", None) in self._GRAD_WRITER_KINDS)
        assert n_writers == n_in_units, ("a gradient-writing launch belongs to no unit", n_writers, n_in_units)

    # ------------------------------------------------------------------ side-stream overlap (backward)
    def _next_slot(self):
        self._slot_seq = getattr(self, "_slot_seq", -1) + 1
        return self._slot_seq % self.nslots

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        return self._side

    def _side_op(self, fns, slot, wa=None):
        """Run `fns` on the side stream, ordered after everything enqueued so far on the current stream
        (fork). The completion event is kept per dy-buffer slot for the matching join. Inside a captured
        hipGraph this becomes a parallel branch. Y5M_OVERLAP=0 runs them inline."""
        def run():
            if not self.overlap:                    # (read at RUN time: a capture may switch the fork off, NativeTrainStep._capture)
                for f in fns:
                    f()
                return
            main = torch.cuda.current_stream()
            side = self._side_stream()
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            tls = getattr(self, "_tl_side", None)      # profile_step(overlapped=True): time the launch on ITS stream
            with torch.cuda.stream(side):
                if tls is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                for f in fns:
                    f()
                if tls is not None:
                    e1.record(side)
                    tls.append((run, e0, e1))
                done = torch.cuda.Event()
                done.record(side)
            self._pending[slot] = done
        run.kind = "wgrad"
        run.wa = wa                     # the y5m_wgrad_args of this launch (bench.py: per-kernel roofline)
        if wa is not None:
            # dy and x once, the packed f32 gradient written (atomics), each unpack launch behind it reads and writes it once
            dw = wa.N * wa.th * wa.tw * wa.C * 4
            u = 1 if len(fns) > 1 else 0
            _traffic(run, (wa.M * wa.N + wa.B * wa.Hin * wa.Win * wa.C) * self.esz, 0, dw * u, dw * (1 + u))
        return run

    def _join_op(self, slot, final=False):
        def run():
            ev = self._pending.pop(slot, None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
        run.kind = "join"
        return run

    def join_all(self):
        """make the current stream wait for every weight gradient still running on the side stream (used where a captured
        segment of the backward list ends: a capture may not end with forked work outstanding)"""
        for slot in list(self._pending):
            torch.cuda.current_stream().wait_event(self._pending.pop(slot))

    def grad_cuts(self, fractions=(0.5, 0.9)):
        """Cut points of the backward launch list for an overlapped gradient exchange. Returns [(op index k, element
        offset lo)], ascending in k / descending in lo: after self.bwd[:k] has run (and join_all()), every parameter
        gradient at flat offset >= lo is final. The backward pass finishes the head first and the stem last while the flat
        buffer is in model order, so the finished part is a growing SUFFIX of the buffer; a cut is taken at the first unit
        boundary where that suffix holds at least the given fraction of all gradient elements (most parameters sit in the
        deep layers, which finish early: half of the elements are final after about a fifth of the backward time)."""
        base, numel = self.model.flat_grads.data_ptr(), self.model.flat_grads.numel()
        ranges = []                                          # (lo, hi, op index after which final)
        for k, name, _addr in self.bwd_marks:
            P = self.model.pslices[name]
            for key in ("gw", "gg", "gb"):
                if key in P:
                    lo = (P[key].data_ptr() - base) // 4
                    ranges.append((lo, lo + P[key].numel(), k))
        ranges.sort()
        assert ranges[0][0] == 0 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])) and ranges[-1][1] == numel, \
            "the unit marks must tile the flat gradient buffer"
        cuts = []
        for f in fractions:
            for k in sorted({r[2] for r in ranges}):
                lo = numel
                for r in reversed(ranges):
                    if r[2] > k:
                        break
                    lo = r[0]
                if numel - lo >= f * numel:
                    if 0 < lo and (not cuts or (k > cuts[-1][0] and lo < cuts[-1][1])):
                        cuts.append((k, lo))
                    break
        return cuts

    def _batch_packs(self):
        """Fold every y5m_pack_weights entry of self.pack into ONE launch over a device job table."""
        L, dt = self.L, self.dtype
        jobs, rest, total = [], [], 0
        for fn, args in self.pack:
            if fn is L.y5m_pack_weights:
                j = _lib.PackJob()
                (src, j.Cout, j.Cin, j.KH, j.KW, j.mode, j.kh0, j.khs, j.th, j.kw0, j.kws, j.tw, dst, j.rows_p, j.Kp,
                 j.cstride, _dt) = args[:17]
                j.ldd = args[17] if len(args) > 17 else 0       # (engine-only extension: column window of wider rows)
                j.src, j.dst, j.start = src.value, dst.value, total
                total += j.rows_p * j.Kp
                jobs.append(j)
            else:
                rest.append((fn, args))
        if not jobs:
            return
        arr = (_lib.PackJob * len(jobs))(*jobs)
        raw = bytes(arr)
        self._pack_table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
        n, tot = len(jobs), total

        def pack_all():
            _lib.check(L.y5m_pack_weights_batched(_lib.ptr(self._pack_table), n, tot, dt, _lib.stream_ptr()),
                       "y5m_pack_weights_batched")
        pack_all.kind = "pack_weights"
        # every master weight a job reads (f32) and every packed element it writes
        masters = {j.src: j.Cout * j.Cin * j.KH * j.KW * 4 for j in jobs}
        _traffic(pack_all, 0, 0, sum(masters.values()), tot * self.esz)
        self.pack = [(pack_all, ())] + rest
        if self._fold_jobs:
            fj, start = [], 0
            for (g_, b_, rm, rv, sc, sh, C) in self._fold_jobs:
                j = _lib.FoldJob()
                j.gamma, j.beta, j.running_mean, j.running_var, j.scale, j.shift, j.C, j.start = g_, b_, rm, rv, sc, sh, C, start
                start += C
                fj.append(j)
            raw = bytes((_lib.FoldJob * len(fj))(*fj))
            self._fold_table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
            nf, totc = len(fj), start

            def fold_all():
                _lib.check(L.y5m_bn_fold_batched(_lib.ptr(self._fold_table), nf, totc, BN_EPS, _lib.stream_ptr()),
                           "y5m_bn_fold_batched")
            self.pack.append((fold_all, ()))

    # ------------------------------------------------------------------ execution
    @staticmethod
    def _run(lst, timeline=None):
        """Enqueue a launch list. timeline: optional list that receives (kind, start_event, end_event)
        per op (events recorded on the current stream = the stream the kernels are launched on)."""
        trace = _TRACE
        for fn, args in lst:
            if trace:                                       # Y5M_TRACE=1: name every launch-list entry and wait for it (fault hunting)
                print("[y5m]", getattr(fn, "kind", getattr(fn, "__name__", "op")), flush=True)
            if timeline is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if args:
                rc = fn(*args, _lib.stream_ptr())
                if rc != 0:
                    raise _lib.Y5MError(f"native call failed rc={rc}: {_lib.lib().y5m_last_error().decode()}")
            else:
                fn()
            if trace:
                torch.cuda.synchronize()
            if timeline is not None:
                e1.record()
                timeline.append((getattr(fn, "kind", getattr(fn, "__name__", "other")), e0, e1))

    def algorithmic_bytes(self, sparse_head=None):
        """The plan's ALGORITHMIC HBM traffic: every operand of every launch-list entry moved once (see _traffic), summed per
        list and per kernel family. Returns {"pack" | "forward" | "backward": {"act_read", "act_written", "par_read",
        "par_written", "launches", "by_kind": {kind: [act_read, act_written, par_read, par_written, launches]}}, "B": batch}.
        The act_* bytes are proportional to the batch, the par_* bytes (weights, weight gradients, BatchNorm rows, fills) do
        not depend on it: total(B') = act * B' / B + par. What L2 absorbs or re-reads is NOT in here -- the PMC counters'
        job (tools/pmc_bench.sh); this is the floor a launch list can be held against. The head-gradient pack is counted in the
        form its launch would take NOW: sparse once NativeTrainStep's loss has set `head_owner`, dense (85x the read) on a plan
        driven through autograd (`backward(grads)`)."""
        if sparse_head is None:                           # (NativeTrainStep.algorithmic_bytes passes the form ITS loss produces)
            sparse_head = self.head_owner is not None
        out = {"B": self.B}
        for name, lst in (("pack", self.pack), ("forward", self.fwd), ("backward", self.bwd if self.training else [])):
            tot, by = [0, 0, 0, 0, 0], {}
            for fn, _args in lst:
                t = getattr(fn, "traffic", None)
                if sparse_head:
                    t = getattr(fn, "traffic_sparse", t)
                kind = getattr(fn, "kind", getattr(fn, "__name__", "other"))
                if t is None:
                    if kind not in ("join", "finalize", "fold_all"):
                        raise AssertionError(f"launch-list entry of kind {kind!r} carries no traffic figure")
                    t = (0, 0, 0, 0)
                row = by.setdefault(kind, [0, 0, 0, 0, 0])
                for i in range(4):
                    row[i] += t[i]
                    tot[i] += t[i]
                row[4] += 1
                tot[4] += 1
            out[name] = {"act_read": tot[0], "act_written": tot[1], "par_read": tot[2], "par_written": tot[3], "launches": tot[4],
                         "by_kind": by}
        return out

    def conv_flops(self):
        """Algorithmic FLOPs (2*MAC on the REAL layer shapes) of one forward
        pass of all convs at this engine's batch and resolution."""
        total = 0
        for lay in self.layers:
            total += 2 * lay.M * lay.cout * lay.cin_real * lay.k * lay.k
        for lay in self.heads:
            total += 2 * lay.x.M * (self.naxs * self.nch) * lay.x.C
        return total

    def forward(self, images=None):
        """images (B,3,H,W) f32 on the device (or already copied into self.x_in). Returns the 3 logits
        buffers (B,naxs,ny,nx,5+nc) f32 -- engine-owned, overwritten by the next forward."""
        if images is not None:
            self.x_in.copy_(images)
        # Every forward packs the bf16 weight rows (and an inference plan folds BatchNorm) from the f32 masters: one launch reading
        # 85 MB, ~4 % of an eval forward at B = 32 @ 640x640 -- and always correct, whoever wrote the masters (the native optimizer and
        # BatchNorm kernels write through raw pointers, `dist.broadcast` / `all_reduce` and a fresh `.data` view bump no counter).
        # OPT-IN (`model.pack_once = True`, or Y5M_PACK_ONCE=1: a deployed detector whose weights are frozen): an inference plan packs
        # once per weight VERSION -- torch's in-place counters of the parameters and the running-statistics buffers plus
        # `model.mark_weights_changed()` for the writers those counters do not see.
        key = self._weights_key() if (not self.training and getattr(self.model, "pack_once", False)) else None
        if key is None or key != getattr(self, "_packed_key", None):
            self._run(self.pack)
            self._packed_key = key
        self._run(self.fwd)
        return self.outs

    def _weights_key(self):
        m = self.model
        # (a Parameter is `p.data = view of the flat buffer`: it keeps its OWN version counter, so every parameter is asked; the
        #  running statistics are plain views of their flat buffer and share its counter; counters only grow, so the sum moves
        #  whenever any of them does). Inference tensors (buffers created under torch.inference_mode) have no counter: no key,
        #  pack on every forward.
        try:
            return (m.flat_params.data_ptr(), sum(p._version for p in m._param_list) + m.flat_params._version,
                    m._flat_stats._version, m._nbt._version, getattr(m, "_weights_epoch", 0))
        except RuntimeError:
            return None

    def backward(self, grads=None):
        """grads: 3 tensors d(loss)/d(logits) (or None if already written into head gout buffers).
        Fills the model's flat gradient buffer (reference parameter layout)."""
        assert self.training
        if grads is not None:
            self.head_owner = None               # arbitrary upstream gradients: dense head pack
            for lay, g in zip(self.heads, grads):
                lay.gout.copy_(g)
        self._run(self.bwd)

    def head_grad_buffers(self):
        return [lay.gout for lay in self.heads]
