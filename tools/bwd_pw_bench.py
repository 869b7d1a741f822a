#!/usr/bin/env python
"""The fused pointwise backward (y5m_bwd_pw) alone on the chip against the three launches it replaces (bn_bwd apply, pointwise
data gradient, pointwise weight gradient), per layer shape of the B=64 @ 640^2 step. A cache-flushing copy runs between the
timed launches (the step's operands are cold). usage: bwd_pw_bench.py [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5m_amd import _lib
from yolov5m_amd._lib import BwdPwArgs, ConvArgs, WgradArgs, EPI_DGRAD, BF16, ACT_SILU

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = _lib.lib()
dev = "cuda"
st = _lib.stream_ptr
flush_src = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
flush_dst = torch.empty_like(flush_src)
rup = lambda x, m: (x + m - 1) // m * m


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters):
        flush_dst.copy_(flush_src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3


for (C, HW, acc) in ((192, 40, 0), (192, 40, 1), (96, 80, 0), (192, 80, 0), (96, 160, 0), (48, 160, 0), (48, 160, 1)):
    M = B * HW * HW
    dz = torch.randn(M * C, device=dev).bfloat16()
    y = torch.randn(M * C, device=dev).bfloat16()
    x = torch.randn(M * C, device=dev).bfloat16()
    dx = torch.zeros(M * C, device=dev).bfloat16()
    dy = torch.zeros(M * C, device=dev).bfloat16()
    wd = (torch.randn(rup(C, L.y5m_conv_tile_n(C)) * rup(C, 64), device=dev) * 0.05).bfloat16()
    Kp = rup(C, 64)
    dw = torch.zeros(C * C, device=dev)
    slots = int(L.y5m_bn_acc_slots())
    acc_rows = torch.zeros(slots * 2 * C, dtype=torch.float64, device=dev)
    scale, shift, mean, invstd = [torch.rand(C, device=dev) + 0.5 for _ in range(4)]
    gg, gb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    a = BwdPwArgs()
    a.y, a.x, a.wd, a.dx = y.data_ptr(), x.data_ptr(), wd.data_ptr(), dx.data_ptr()
    a.M, a.ldy, a.ldx, a.Kp, a.lddx, a.lddw, a.N, a.C, a.accumulate, a.act, a.nseg = M, C, C, Kp, C, C, C, C, acc, ACT_SILU, 1
    s = a.seg[0]
    s.c0, s.cn, s.dz, s.lddz, s.acc = 0, C, dz.data_ptr(), C, acc_rows.data_ptr()
    s.scale, s.shift, s.mean, s.invstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    s.dgamma, s.dbeta, s.dw = gg.data_ptr(), gb.data_ptr(), dw.data_ptr()
    assert L.y5m_bwd_pw_eligible(ctypes.byref(a), BF16)
    red = lambda: _lib.check(L.y5m_bn_bwd_fused_phase(dz.data_ptr(), C, y.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                                    invstd.data_ptr(), M, C, ACT_SILU, None, None, 0, None, 0, acc_rows.data_ptr(), BF16, st(), 1), "reduce")
    app = lambda: _lib.check(L.y5m_bn_bwd_fused_phase(dz.data_ptr(), C, y.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                                    invstd.data_ptr(), M, C, ACT_SILU, gg.data_ptr(), gb.data_ptr(), 0, dy.data_ptr(), C,
                                                    acc_rows.data_ptr(), BF16, st(), 2), "apply")
    fused = lambda: _lib.check(L.y5m_bwd_pw(ctypes.byref(a), BF16, st()), "bwd_pw")
    g = ConvArgs()
    g.zeros = _lib.zero_page(dev).data_ptr()
    g.inp, g.w, g.out = dy.data_ptr(), wd.data_ptr(), dx.data_ptr()
    g.B, g.Hin, g.Win, g.ldin, g.Hg, g.Wg, g.sy, g.sx = B, HW, HW, C, HW, HW, 1, 1
    g.th, g.tw, g.dh0, g.dhs, g.dw0, g.dws = 1, 1, 0, -1, 0, -1
    g.Cin, g.K, g.Kp, g.N, g.M = C, C, Kp, C, M
    g.Hout, g.Wout, g.ldout, g.osy, g.osx, g.epi, g.accumulate = HW, HW, C, 1, 1, EPI_DGRAD, acc
    g.Np = rup(C, L.y5m_conv_tile_n(C))
    dgrad = lambda: _lib.check(L.y5m_conv(ctypes.byref(g), BF16, st()), "dgrad")
    wa = WgradArgs()
    wa.zeros = _lib.zero_page(dev).data_ptr()
    wa.dy, wa.x, wa.dwgt = dy.data_ptr(), x.data_ptr(), dw.data_ptr()
    wa.B, wa.Hin, wa.Win, wa.ldx, wa.Hg, wa.Wg, wa.sy, wa.sx = B, HW, HW, C, HW, HW, 1, 1
    wa.th, wa.tw, wa.dh0, wa.dhs, wa.dw0, wa.dws = 1, 1, 0, 1, 0, 1
    wa.C, wa.N, wa.M, wa.lddy, wa.lddw, wa.ksplit = C, C, M, C, C, 0
    wgrad = lambda: _lib.check(L.y5m_wgrad(ctypes.byref(wa), BF16, st()), "wgrad")
    t = {n: timeit(f) for n, f in (("reduce", red), ("fused", fused), ("apply", app), ("dgrad", dgrad), ("wgrad", wgrad))}
    bytes_f = M * C * 2 * (4 + acc)
    print(f"C={C:3d} {HW}x{HW} acc={acc}: fused {t['fused']:7.1f} us ({bytes_f / t['fused'] / 1e3:6.0f} GB/s algorithmic)   "
          f"apply {t['apply']:6.1f} + dgrad {t['dgrad']:6.1f} = {t['apply'] + t['dgrad']:6.1f} us on the main stream, wgrad {t['wgrad']:6.1f} us forked   "
          f"(reduce {t['reduce']:6.1f} us in both)", flush=True)
    if hasattr(L, "y5m_debug_bp_timing") or os.environ.get("Y5M_LIB"):
        try:
            buf = (ctypes.c_ulonglong * 16)()
            fused(); torch.cuda.synchronize()
            L.y5m_debug_bp_timing.restype = ctypes.c_int
            if L.y5m_debug_bp_timing(buf) == 0:
                for w in range(2):
                    t = [buf[w * 8 + k] for k in range(4)]
                    print(f"      wg {'0' if w == 0 else 'last'}: prologue {(t[1]-t[0])/100:.1f}  loop {(t[2]-t[1])/100:.1f}  atomics+drain {(t[3]-t[2])/100:.1f}  (us at 100 MHz s_memtime)")
        except Exception as e:
            print("      (no timing build)", e)
    del dz, y, x, dx, dy
