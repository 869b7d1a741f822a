#!/usr/bin/env python
"""Micro-benchmark of single conv / wgrad launches (for rocprofv3 --pmc runs).
usage: conv_bench.py KIND B Cin H W Cout k s [iters]   KIND in fwd|dgrad|wgrad"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5m_amd import _lib
from yolov5m_amd._lib import ConvArgs, WgradArgs, EPI_RAW_STATS, EPI_DGRAD, BF16

kind = sys.argv[1]
B, Cin, H, W, Cout, k, s = [int(v) for v in sys.argv[2:9]]
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 50
p = k // 2
L = _lib.lib()
dev = "cuda"
Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
M = B * Ho * Wo
rup = lambda x, m: (x + m - 1) // m * m
x = torch.randn(B * H * W * Cin, device=dev).bfloat16()
y = torch.randn(M * Cout, device=dev).bfloat16()
z = _lib.zero_page(dev)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if kind in ("fwd", "dgrad"):
    K = k * k * Cin
    Kp = rup(K, 64)
    BN = L.y5m_conv_tile_n(Cout)
    Np = rup(Cout, BN)
    w = (torch.randn(Np * Kp, device=dev) * 0.05).bfloat16()
    a = ConvArgs()
    a.zeros = z.data_ptr()
    a.inp, a.w, a.out = x.data_ptr(), w.data_ptr(), y.data_ptr()
    a.B, a.Hin, a.Win, a.ldin, a.Hg, a.Wg, a.sy, a.sx = B, H, W, Cin, Ho, Wo, s, s
    a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = k, k, -p, 1, -p, 1
    a.Cin, a.K, a.Kp, a.N, a.M = Cin, K, Kp, Cout, M
    a.Hout, a.Wout, a.ldout, a.osy, a.osx = Ho, Wo, Cout, 1, 1
    a.Np = Np
    if kind == "fwd":
        a.epi = EPI_RAW_STATS
        stats = torch.zeros(L.y5m_conv_stats_rows(ctypes.byref(a), BF16) * 2 * Np, device=dev)
        a.stats = stats.data_ptr()
    else:
        a.epi = EPI_DGRAD
    run = lambda: _lib.check(L.y5m_conv(ctypes.byref(a), BF16, _lib.stream_ptr()), "conv")
    flops = 2.0 * M * Cout * K
else:
    g = torch.zeros(Cout * k * k * Cin, device=dev)
    a = WgradArgs()
    a.zeros = z.data_ptr()
    a.dy, a.x, a.dwgt = y.data_ptr(), x.data_ptr(), g.data_ptr()
    a.B, a.Hin, a.Win, a.ldx, a.Hg, a.Wg, a.sy, a.sx = B, H, W, Cin, Ho, Wo, s, s
    a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = k, k, -p, 1, -p, 1
    a.C, a.N, a.M, a.lddy, a.lddw, a.ksplit = Cin, Cout, M, Cout, k * k * Cin, 0
    run = lambda: _lib.check(L.y5m_wgrad(ctypes.byref(a), BF16, _lib.stream_ptr()), "wgrad")
    flops = 2.0 * M * Cout * k * k * Cin
for _ in range(5):
    run()
torch.cuda.synchronize()
ev0.record()
for _ in range(iters):
    run()
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / iters
print(f"{kind} B={B} Cin={Cin} {H}x{W} Cout={Cout} k={k} s={s}: {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TF/s")
