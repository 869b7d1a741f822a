"""Functional single-op wrappers over the C ABI (NCHW f32 in / out, layout conversion by torch).

These are the op-level surface the parity tests drive (the model itself goes through engine.py with
resident NHWC buffers and never converts layouts). dtype: "f32" or "bf16".
"""
import ctypes

import torch

from . import _lib
from ._lib import ConvArgs, WgradArgs, EPI_DGRAD, EPI_AFFINE_ACT, EPI_RAW_STATS, ACT_NONE, ACT_SILU, F32, BF16

_DT = {"f32": (F32, torch.float32, 4, 32), "bf16": (BF16, torch.bfloat16, 8, 64)}


LAST_WGRAD_KERNEL = None
LAST_KERNEL = None      # which kernel family the most recent y5m_conv launch of this module took (tests assert on it)


def _rup(x, m):
    return (x + m - 1) // m * m


def _conv(a, dt, what="y5m_conv"):
    global LAST_KERNEL
    L = _lib.lib()
    name = ctypes.create_string_buffer(192)
    L.y5m_conv_kernel_name(ctypes.byref(a), dt, name, 192)
    LAST_KERNEL = ("halo" if L.y5m_conv_is_halo(ctypes.byref(a), dt) else
                   "gemm8" if name.value.startswith(b"conv_gemm8") else
                   "pointwise" if L.y5m_conv_is_pointwise(ctypes.byref(a), dt) else "tiled")
    _lib.check(L.y5m_conv(ctypes.byref(a), dt, _lib.stream_ptr()), what)


def to_nhwc(x, tdt, cpad=None):
    B, C, H, W = x.shape
    Cp = cpad or C
    out = torch.zeros((B, H, W, Cp), dtype=tdt, device=x.device)
    out[..., :C] = x.permute(0, 2, 3, 1).to(tdt)
    return out


def from_nhwc(x, C=None):
    return x[..., : (C or x.shape[-1])].permute(0, 3, 1, 2).float().contiguous()


def pack_fwd(w, dtype):
    L = _lib.lib()
    dt, tdt, CH, BK = _DT[dtype]
    Cout, Cin, k, _ = w.shape
    Kp = _rup(k * k * Cin, BK)
    Np = _rup(Cout, L.y5m_conv_tile_n(Cout))
    wf = torch.zeros((Np, Kp), dtype=tdt, device=w.device)
    _lib.check(L.y5m_pack_weights(_lib.ptr(w.contiguous().float()), Cout, Cin, k, k, 0, 0, 1, k, 0, 1, k, _lib.ptr(wf),
                                  Np, Kp, 0, dt, _lib.stream_ptr()), "y5m_pack_weights")
    return wf, Kp, Np


def conv_forward(x, w, stride, pad, dtype="f32", scale=None, shift=None, act=False, res=None):
    """x (B,Cin,H,W), w (Cout,Cin,k,k) -> (B,Cout,Ho,Wo). Optional fused affine + SiLU + residual."""
    L = _lib.lib()
    dt, tdt, CH, BK = _DT[dtype]
    _lib.require_cuda(x, w)
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    assert Cin % CH == 0 and Cout % 4 == 0
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xn = to_nhwc(x, tdt)
    wf, Kp, Np = pack_fwd(w, dtype)
    out = torch.zeros((B, Ho, Wo, Cout), dtype=tdt, device=x.device)
    a = ConvArgs()
    a.zeros = _lib.zero_page(x.device).data_ptr()
    a.inp, a.w, a.out = xn.data_ptr(), wf.data_ptr(), out.data_ptr()
    a.B, a.Hin, a.Win, a.ldin = B, H, W, Cin
    a.Hg, a.Wg, a.sy, a.sx = Ho, Wo, stride, stride
    a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = k, k, -pad, 1, -pad, 1
    a.Cin, a.K, a.Kp, a.N, a.M = Cin, k * k * Cin, Kp, Cout, B * Ho * Wo
    a.Hout, a.Wout, a.ldout, a.osy, a.osx, a.ooy, a.oox = Ho, Wo, Cout, 1, 1, 0, 0
    a.Np = Np
    keep = []
    if scale is not None:
        sc, sh = scale.float().contiguous(), shift.float().contiguous()
        keep += [sc, sh]
        a.epi, a.act = EPI_AFFINE_ACT, (ACT_SILU if act else ACT_NONE)
        a.scale, a.shift = sc.data_ptr(), sh.data_ptr()
        if res is not None:
            rn = to_nhwc(res, tdt)
            keep.append(rn)
            a.res, a.ldres = rn.data_ptr(), Cout
    else:
        a.epi = EPI_DGRAD
    _conv(a, dt)
    torch.cuda.synchronize()
    return from_nhwc(out)


def conv_forward_stats(x, w, stride, pad, dtype="f32"):
    """raw conv output + the per-channel (sum, sumsq) the training epilogue emits"""
    L = _lib.lib()
    dt, tdt, CH, BK = _DT[dtype]
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xn = to_nhwc(x, tdt)
    wf, Kp, Np = pack_fwd(w, dtype)
    out = torch.zeros((B, Ho, Wo, Cout), dtype=tdt, device=x.device)
    M = B * Ho * Wo
    a = ConvArgs()
    a.zeros = _lib.zero_page(x.device).data_ptr()
    a.inp, a.w, a.out = xn.data_ptr(), wf.data_ptr(), out.data_ptr()
    a.B, a.Hin, a.Win, a.ldin = B, H, W, Cin
    a.Hg, a.Wg, a.sy, a.sx = Ho, Wo, stride, stride
    a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = k, k, -pad, 1, -pad, 1
    a.Cin, a.K, a.Kp, a.N, a.M = Cin, k * k * Cin, Kp, Cout, M
    a.Hout, a.Wout, a.ldout, a.osy, a.osx, a.ooy, a.oox = Ho, Wo, Cout, 1, 1, 0, 0
    a.Np, a.epi = Np, EPI_RAW_STATS
    stats = torch.zeros((L.y5m_conv_stats_rows(ctypes.byref(a), dt), 2, Np), dtype=torch.float32, device=x.device)
    a.stats = stats.data_ptr()
    _conv(a, dt)
    torch.cuda.synchronize()
    s = stats.sum(0)
    return from_nhwc(out), s[0, :Cout], s[1, :Cout]


def conv_forward_bn_fused(x, w, stride, pad, gamma, beta, running_mean, running_var, momentum, eps, dtype="f32", split=None,
                          repeats=1):
    """training-mode CBL forward WITHOUT partial rows / y5m_bn_finalize: the conv adds its channel sums into f64 accumulator
    rows (y5m_conv_args.bn_acc), y5m_bn_act_fused derives the batch statistics from them and normalises + SiLU. Returns
    (raw conv output, activated output, scale, shift, mean, invstd, running_mean, running_var).
    split: channel index where a second BatchNorm layer starts (the merged C3 pair: one conv, two y5m_bn_act_fused calls on
    column ranges of the same rows). repeats: the whole sequence that many times (running statistics move each time)."""
    L = _lib.lib()
    dt, tdt, CH, BK = _DT[dtype]
    _lib.require_cuda(x, w)
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xn = to_nhwc(x, tdt)
    wf, Kp, Np = pack_fwd(w, dtype)
    out = torch.zeros((B, Ho, Wo, Cout), dtype=tdt, device=x.device)
    z = torch.zeros_like(out)
    M = B * Ho * Wo
    a = ConvArgs()
    a.zeros = _lib.zero_page(x.device).data_ptr()
    a.inp, a.w, a.out = xn.data_ptr(), wf.data_ptr(), out.data_ptr()
    a.B, a.Hin, a.Win, a.ldin = B, H, W, Cin
    a.Hg, a.Wg, a.sy, a.sx = Ho, Wo, stride, stride
    a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = k, k, -pad, 1, -pad, 1
    a.Cin, a.K, a.Kp, a.N, a.M = Cin, k * k * Cin, Kp, Cout, M
    a.Hout, a.Wout, a.ldout, a.osy, a.osx, a.ooy, a.oox = Ho, Wo, Cout, 1, 1, 0, 0
    a.Np, a.epi = Np, EPI_RAW_STATS
    g, b = gamma.float().contiguous(), beta.float().contiguous()
    rm, rv = running_mean.float().clone().contiguous(), running_var.float().clone().contiguous()
    res = torch.zeros((4, Cout), dtype=torch.float32, device=x.device)      # scale, shift, mean, invstd
    acc = torch.zeros((L.y5m_bn_acc_slots(), 2, Np), dtype=torch.float64, device=x.device)
    # (the halo-patch kernel stages its tiles' sums in partial rows before it adds them)
    stats = torch.zeros((L.y5m_conv_stats_rows(ctypes.byref(a), dt), 2, Np), dtype=torch.float32, device=x.device)
    a.bn_acc, a.stats = acc.data_ptr(), stats.data_ptr()
    bounds = [0, Cout] if split is None else [0, split, Cout]
    esz = 2 if dtype == "bf16" else 4
    for _ in range(repeats):
        acc.zero_()
        _conv(a, dt)
        for c0, c1 in zip(bounds[:-1], bounds[1:]):
            _lib.check(L.y5m_bn_act_fused(out.data_ptr() + c0 * esz, Cout, acc.data_ptr() + 8 * c0, Np, M,
                                          g.data_ptr() + 4 * c0, b.data_ptr() + 4 * c0, rm.data_ptr() + 4 * c0,
                                          rv.data_ptr() + 4 * c0, momentum, eps, 1, res[0].data_ptr() + 4 * c0,
                                          res[1].data_ptr() + 4 * c0, res[2].data_ptr() + 4 * c0, res[3].data_ptr() + 4 * c0,
                                          None, 0, z.data_ptr() + c0 * esz, Cout, M, c1 - c0, ACT_SILU, dt, _lib.stream_ptr()),
                       "y5m_bn_act_fused")
    torch.cuda.synchronize()
    return from_nhwc(out), from_nhwc(z), res[0], res[1], res[2], res[3], rm, rv


def conv_dgrad(dy, w, in_hw, stride, pad, dtype="f32", init=None, src=None):
    """dy (B,Cout,Ho,Wo), w (Cout,Cin,k,k) -> dx (B,Cin,H,W): autograd of conv2d wrt its input.
    init (B,Cin,H,W): accumulate the gradient onto it (the engine's fan-out accumulation).
    src  (B,Cin,H,W): dx = src + gradient with src read from its OWN tensor (the fused residual gradient)."""
    L = _lib.lib()
    dt, tdt, CH, BK = _DT[dtype]
    B, Cout, Ho, Wo = dy.shape
    _, Cin, k, _ = w.shape
    H, W = in_hw
    dyn = to_nhwc(dy, tdt)
    dx = torch.zeros((B, H, W, Cin), dtype=tdt, device=dy.device) if init is None else to_nhwc(init, tdt)
    srcn = to_nhwc(src, tdt) if src is not None else None
    if src is not None:
        dx.fill_(7.0)                # must be overwritten, not accumulated
    wsrc = w.contiguous().float()
    classes = [(0, 0)] if stride == 1 else [(py, px) for py in range(2) for px in range(2)]
    for (py, px) in classes:
        if stride == 1:
            kh0, khs, th, dh0 = 0, 1, k, pad
            kw0, kws, tw, dw0 = 0, 1, k, pad
        else:
            kh0 = (py + pad) % 2; th = len(range(kh0, k, 2)); dh0 = (py + pad - kh0) // 2; khs = 2
            kw0 = (px + pad) % 2; tw = len(range(kw0, k, 2)); dw0 = (px + pad - kw0) // 2; kws = 2
        Kd = th * tw * Cout
        rows = _rup(Cin, L.y5m_conv_tile_n(Cin))
        wd = torch.zeros((rows, _rup(Kd, BK)), dtype=tdt, device=dy.device)
        _lib.check(L.y5m_pack_weights(_lib.ptr(wsrc), Cout, Cin, k, k, 1, kh0, khs, th, kw0, kws, tw, _lib.ptr(wd),
                                      wd.shape[0], wd.shape[1], 0, dt, _lib.stream_ptr()), "y5m_pack_weights")
        a = ConvArgs()
        a.zeros = _lib.zero_page(dy.device).data_ptr()
        a.inp, a.w, a.out = dyn.data_ptr(), wd.data_ptr(), dx.data_ptr()
        a.B, a.Hin, a.Win, a.ldin = B, Ho, Wo, Cout
        if stride == 1:
            a.Hg, a.Wg, a.osy, a.osx, a.ooy, a.oox = H, W, 1, 1, 0, 0
        else:
            a.Hg, a.Wg, a.osy, a.osx, a.ooy, a.oox = H // 2, W // 2, 2, 2, py, px
        a.sy, a.sx = 1, 1
        a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = th, tw, dh0, -1, dw0, -1
        a.Cin, a.K, a.Kp, a.N, a.M = Cout, Kd, wd.shape[1], Cin, B * a.Hg * a.Wg
        a.Hout, a.Wout, a.ldout = H, W, Cin
        a.epi, a.Np = EPI_DGRAD, wd.shape[0]
        a.accumulate = 0 if (init is None and src is None) else 1
        if srcn is not None:
            a.res, a.ldres = srcn.data_ptr(), Cin
        _conv(a, dt, "y5m_conv(dgrad)")
        torch.cuda.synchronize()
    return from_nhwc(dx)


def conv_wgrad(dy, x, k, stride, pad, dtype="f32", ksplit=0):
    """dy (B,Cout,Ho,Wo), x (B,Cin,H,W) -> dw (Cout,Cin,k,k): autograd of conv2d wrt its weight."""
    L = _lib.lib()
    dt, tdt, CH, BK = _DT[dtype]
    B, Cout, Ho, Wo = dy.shape
    _, Cin, H, W = x.shape
    dyn, xn = to_nhwc(dy, tdt), to_nhwc(x, tdt)
    gp = torch.zeros((Cout, k * k * Cin), dtype=torch.float32, device=dy.device)
    a = WgradArgs()
    a.zeros = _lib.zero_page(dy.device).data_ptr()
    a.dy, a.x, a.dwgt = dyn.data_ptr(), xn.data_ptr(), gp.data_ptr()
    a.B, a.Hin, a.Win, a.ldx = B, H, W, Cin
    a.Hg, a.Wg, a.sy, a.sx = Ho, Wo, stride, stride
    a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = k, k, -pad, 1, -pad, 1
    a.C, a.N, a.M, a.lddy, a.lddw, a.ksplit = Cin, Cout, B * Ho * Wo, Cout, k * k * Cin, ksplit
    out = torch.zeros((Cout, Cin, k, k), dtype=torch.float32, device=dy.device)
    global LAST_WGRAD_KERNEL
    _buf = ctypes.create_string_buffer(192)
    _lib.check(L.y5m_wgrad_kernel_name(ctypes.byref(a), dt, _buf, 192), "y5m_wgrad_kernel_name")
    LAST_WGRAD_KERNEL = _buf.value.decode()
    _lib.check(L.y5m_wgrad(ctypes.byref(a), dt, _lib.stream_ptr()), "y5m_wgrad")
    _lib.check(L.y5m_unpack_wgrad(_lib.ptr(gp), Cout, Cin, k, k, 0, k * k * Cin, _lib.ptr(out), _lib.stream_ptr()),
               "y5m_unpack_wgrad")
    torch.cuda.synchronize()
    return out


def bwd_pw(dz, y, x, w, scale, shift, mean, invstd, split=None, init=None, src=None):
    """Fused backward of a pointwise CBL (y5m_bwd_pw; bf16, Cin == Cout in {48, 96, 192}): dz, y (B,C,H,W), x (B,C,H,W),
    w (C,C,1,1), per-channel forward statistics scale = gamma*invstd, shift, mean, invstd (C,). The BatchNorm reduction is
    accumulated first (y5m_bn_bwd_fused_phase, phase 1), per segment when `split` cuts the output channels in two (the merged
    C3 pair). init: accumulate dx onto it; src: dx = src + gradient. Returns dx, dW (C,C,1,1), dgamma, dbeta."""
    from ._lib import BwdPwArgs
    L = _lib.lib()
    dt, tdt, CH, BK = _DT["bf16"]
    B, C, H, W = dz.shape
    M = B * H * W
    dev = dz.device
    dzn, yn, xn = to_nhwc(dz, tdt), to_nhwc(y, tdt), to_nhwc(x, tdt)
    dx = torch.zeros((B, H, W, C), dtype=tdt, device=dev) if init is None else to_nhwc(init, tdt)
    srcn = to_nhwc(src, tdt) if src is not None else None
    if src is not None:
        dx.fill_(7.0)
    wd = torch.zeros((_rup(C, L.y5m_conv_tile_n(C)), _rup(C, BK)), dtype=tdt, device=dev)
    _lib.check(L.y5m_pack_weights(_lib.ptr(w.contiguous().float()), C, C, 1, 1, 1, 0, 1, 1, 0, 1, 1, _lib.ptr(wd),
                                  wd.shape[0], wd.shape[1], 0, dt, _lib.stream_ptr()), "y5m_pack_weights")
    segs = [(0, C)] if split is None else [(0, split), (split, C - split)]
    slots = int(L.y5m_bn_acc_slots())
    dw = torch.zeros((C, C), dtype=torch.float32, device=dev)
    dgamma = torch.full((C,), float("nan"), dtype=torch.float32, device=dev)
    dbeta = torch.full((C,), float("nan"), dtype=torch.float32, device=dev)
    st = [t.contiguous().float() for t in (scale, shift, mean, invstd)]
    a = BwdPwArgs()
    a.y, a.x, a.wd, a.dx = yn.data_ptr(), xn.data_ptr(), wd.data_ptr(), dx.data_ptr()
    a.res = srcn.data_ptr() if srcn is not None else None
    a.M, a.ldy, a.ldx, a.Kp, a.lddx, a.ldres, a.lddw = M, C, C, wd.shape[1], C, C, C
    a.N, a.C, a.accumulate, a.act, a.nseg = C, C, 1 if (init is not None or src is not None) else 0, ACT_SILU, len(segs)
    keep = []
    for i, (c0, cn) in enumerate(segs):
        acc = torch.zeros((slots, 2, cn), dtype=torch.float64, device=dev)
        sl = [t[c0:c0 + cn].contiguous() for t in st]
        keep += [acc] + sl
        _lib.check(L.y5m_bn_bwd_fused_phase(dzn.data_ptr() + 2 * c0, C, yn.data_ptr() + 2 * c0, C, _lib.ptr(sl[0]), _lib.ptr(sl[1]),
                                            _lib.ptr(sl[2]), _lib.ptr(sl[3]), M, cn, ACT_SILU, None, None, 0, None, 0,
                                            _lib.ptr(acc), dt, _lib.stream_ptr(), 1), "y5m_bn_bwd_fused_phase(reduce)")
        s = a.seg[i]
        s.c0, s.cn, s.acc = c0, cn, acc.data_ptr()
        s.dz, s.lddz = dzn.data_ptr() + 2 * c0, C
        s.scale, s.shift, s.mean, s.invstd = (t.data_ptr() for t in sl)
        s.dgamma, s.dbeta = dgamma.data_ptr() + 4 * c0, dbeta.data_ptr() + 4 * c0
        s.dw = dw.data_ptr() + 4 * c0 * C
    assert L.y5m_bwd_pw_eligible(ctypes.byref(a), dt) == 1
    _lib.check(L.y5m_bwd_pw(ctypes.byref(a), dt, _lib.stream_ptr()), "y5m_bwd_pw")
    torch.cuda.synchronize()
    return from_nhwc(dx), dw.view(C, C, 1, 1).clone(), dgamma, dbeta


def bwd_stem(dz, y, x, scale, shift, mean, invstd):
    """Fused backward of the stem CBL in its executed form (y5m_bwd_stem; bf16): dz, y (B,48,H,W) gradient wrt / raw output of a
    3x3 / stride 1 / pad 1 conv over x (B,16,H,W) (the space-to-depth image); per-channel forward statistics as in bwd_pw. The
    BatchNorm reduction is accumulated first (y5m_bn_bwd_fused_phase, phase 1). Returns dW (48,16,3,3), dgamma, dbeta."""
    from ._lib import BwdStemArgs
    L = _lib.lib()
    dt, tdt, CH, BK = _DT["bf16"]
    B, N, H, W = dz.shape
    C = x.shape[1]
    M = B * H * W
    dev = dz.device
    dzn, yn, xn = to_nhwc(dz, tdt), to_nhwc(y, tdt), to_nhwc(x, tdt)
    slots = int(L.y5m_bn_acc_slots())
    acc = torch.zeros((slots, 2, N), dtype=torch.float64, device=dev)
    st = [t.contiguous().float() for t in (scale, shift, mean, invstd)]
    _lib.check(L.y5m_bn_bwd_fused_phase(dzn.data_ptr(), N, yn.data_ptr(), N, _lib.ptr(st[0]), _lib.ptr(st[1]), _lib.ptr(st[2]),
                                        _lib.ptr(st[3]), M, N, ACT_SILU, None, None, 0, None, 0, _lib.ptr(acc), dt,
                                        _lib.stream_ptr(), 1), "y5m_bn_bwd_fused_phase(reduce)")
    gp = torch.zeros((N, 9 * C), dtype=torch.float32, device=dev)
    dgamma = torch.full((N,), float("nan"), dtype=torch.float32, device=dev)
    dbeta = torch.full((N,), float("nan"), dtype=torch.float32, device=dev)
    a = BwdStemArgs()
    a.dz, a.y, a.x, a.dwgt = dzn.data_ptr(), yn.data_ptr(), xn.data_ptr(), gp.data_ptr()
    a.B, a.H, a.W, a.lddz, a.ldy, a.ldx, a.lddw, a.N, a.C, a.act = B, H, W, N, N, C, 9 * C, N, C, ACT_SILU
    a.acc = acc.data_ptr()
    a.scale, a.shift, a.mean, a.invstd = (t.data_ptr() for t in st)
    a.dgamma, a.dbeta = dgamma.data_ptr(), dbeta.data_ptr()
    assert L.y5m_bwd_stem_eligible(ctypes.byref(a)) == 1
    _lib.check(L.y5m_bwd_stem(ctypes.byref(a), _lib.stream_ptr()), "y5m_bwd_stem")
    out = torch.zeros((N, C, 3, 3), dtype=torch.float32, device=dev)
    _lib.check(L.y5m_unpack_wgrad(_lib.ptr(gp), N, C, 3, 3, 0, 9 * C, _lib.ptr(out), _lib.stream_ptr()), "y5m_unpack_wgrad")
    torch.cuda.synchronize()
    return out, dgamma, dbeta


def sppf_pool(x, dtype="f32"):
    """x (B,C,H,W) -> the three cascaded MaxPool2d(5,1,2) outputs (reference model.py:108-110) in one native launch"""
    L = _lib.lib()
    dt, tdt, CH, BK = _DT[dtype]
    _lib.require_cuda(x)
    B, C, H, W = x.shape
    xn = to_nhwc(x, tdt)
    outs = [torch.zeros_like(xn) for _ in range(3)]
    wsb = L.y5m_sppf_pool_workspace_bytes(B, H, W, C)
    ws = torch.zeros(wsb, dtype=torch.uint8, device=x.device)
    _lib.check(L.y5m_sppf_pool(xn.data_ptr(), C, B, H, W, C, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                               _lib.ptr(ws), wsb, dt, _lib.stream_ptr()), "y5m_sppf_pool")
    torch.cuda.synchronize()
    return tuple(from_nhwc(o) for o in outs)


def bn_act(y, scale, shift, dtype="f32", act=True):
    """z = act(y * scale + shift) per channel (y5m_bn_act): y (B,C,H,W), scale / shift (C,)"""
    L = _lib.lib()
    dt, tdt, CH, BK = _DT[dtype]
    _lib.require_cuda(y)
    B, C, H, W = y.shape
    yn = to_nhwc(y, tdt)
    out = torch.zeros_like(yn)
    sc, sh = scale.float().contiguous(), shift.float().contiguous()
    _lib.check(L.y5m_bn_act(_lib.ptr(yn), C, _lib.ptr(sc), _lib.ptr(sh), None, 0, _lib.ptr(out), C, B * H * W, C,
                            ACT_SILU if act else ACT_NONE, dt, _lib.stream_ptr()), "y5m_bn_act")
    torch.cuda.synchronize()
    return from_nhwc(out)


def bn_silu_backward(dz, y, scale, shift, mean, invstd, dtype="f32"):
    """autograd of z = silu(batch_norm(y)) with BATCH statistics wrt y, gamma, beta (y5m_bn_bwd_fused: reduce into f64
    accumulator rows + apply): dz, y (B,C,H,W); scale = gamma * invstd, shift, mean, invstd (C,) as the forward produced them.
    Returns dy (B,C,H,W) f32, dgamma, dbeta."""
    L = _lib.lib()
    dt, tdt, CH, BK = _DT[dtype]
    _lib.require_cuda(dz, y)
    B, C, H, W = dz.shape
    M = B * H * W
    dzn, yn = to_nhwc(dz, tdt), to_nhwc(y, tdt)
    dy = torch.zeros_like(dzn)
    dg = torch.zeros(C, dtype=torch.float32, device=dz.device)
    db = torch.zeros(C, dtype=torch.float32, device=dz.device)
    acc = torch.zeros((L.y5m_bn_acc_slots(), 2, C), dtype=torch.float64, device=dz.device)
    st = [t.float().contiguous() for t in (scale, shift, mean, invstd)]
    _lib.check(L.y5m_bn_bwd_fused(_lib.ptr(dzn), C, _lib.ptr(yn), C, _lib.ptr(st[0]), _lib.ptr(st[1]), _lib.ptr(st[2]),
                                  _lib.ptr(st[3]), M, C, ACT_SILU, _lib.ptr(dg), _lib.ptr(db), 0, _lib.ptr(dy), C,
                                  acc.data_ptr(), dt, _lib.stream_ptr()), "y5m_bn_bwd_fused")
    torch.cuda.synchronize()
    return from_nhwc(dy), dg, db


def sppf_pool_backward(x, p1, p2, g, dtype="f32"):
    """autograd of the SPPF pool cascade p1 = pool(x), p2 = pool(p1), p3 = pool(p2) (reference model.py:108-110) wrt x:
    g = (g_x, g_p1, g_p2, g_p3), the gradients arriving at the four concat slices (B,C,H,W). One native call
    (y5m_sppf_pool_bwd, accumulating): g_p2 += bwd(p2; g_p3), g_p1 += bwd(p1; g_p2), g_x += bwd(x; g_p1). Returns d(x)."""
    L = _lib.lib()
    dt, tdt, CH, BK = _DT[dtype]
    _lib.require_cuda(x)
    B, C, H, W = x.shape
    src = [to_nhwc(t, tdt) for t in (x, p1, p2)]
    gn = [to_nhwc(t, tdt) for t in g]
    wsb = L.y5m_maxpool5_bwd_workspace_bytes(B, H, W, C)
    ws = torch.zeros(max(wsb, 1), dtype=torch.uint8, device=x.device)
    _lib.check(L.y5m_sppf_pool_bwd(_lib.ptr(src[0]), _lib.ptr(src[1]), _lib.ptr(src[2]), C, _lib.ptr(gn[0]), _lib.ptr(gn[1]),
                                   _lib.ptr(gn[2]), _lib.ptr(gn[3]), C, B, H, W, C, _lib.ptr(ws), wsb, dt, _lib.stream_ptr()),
               "y5m_sppf_pool_bwd")
    torch.cuda.synchronize()
    return from_nhwc(gn[0])
