#!/usr/bin/env python
"""bandwidth of the BN elementwise kernels vs a plain device copy (same box, same sizes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5m_amd import _lib
from yolov5m_amd._lib import BF16, ACT_SILU
L = _lib.lib()
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for (M, C) in ((1638400, 48), (1638400, 96), (409600, 96), (409600, 192), (102400, 192), (102400, 384), (25600, 768)):
    y = torch.randn(M * C, device=dev).bfloat16(); dz = torch.randn(M * C, device=dev).bfloat16()
    out = torch.empty_like(y)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
    mu, inv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    wsb = L.y5m_bn_bwd_workspace_bytes(M, C); ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
    st = _lib.stream_ptr()
    t_copy = timeit(lambda: out.copy_(y))
    t_act = timeit(lambda: L.y5m_bn_act(_lib.ptr(y), C, _lib.ptr(sc), _lib.ptr(sh), None, 0, _lib.ptr(out), C, M, C, ACT_SILU, BF16, st))
    t_bwd = timeit(lambda: L.y5m_bn_bwd(_lib.ptr(dz), C, _lib.ptr(y), C, _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(mu), _lib.ptr(inv), M, C, ACT_SILU,
                                        _lib.ptr(dg), _lib.ptr(db), 0, _lib.ptr(out), C, _lib.ptr(ws), wsb, BF16, st))
    part = torch.zeros(512 * 2 * C, device=dev)
    t_app = timeit(lambda: L.y5m_bn_bwd_from_partials(_lib.ptr(part), 512, C, _lib.ptr(dz), C, _lib.ptr(y), C, _lib.ptr(sc), _lib.ptr(sh),
                                                      _lib.ptr(mu), _lib.ptr(inv), M, C, ACT_SILU, _lib.ptr(dg), _lib.ptr(db), 0,
                                                      _lib.ptr(out), C, _lib.ptr(ws), wsb, BF16, st))
    b = M * C * 2
    print(f"M={M:8d} C={C:4d} ({b/1e6:6.1f} MB/tensor): copy {2*b/t_copy/1e12:5.2f} TB/s ({t_copy*1e6:6.1f} us) | bn_act {2*b/t_act/1e12:5.2f} TB/s ({t_act*1e6:6.1f} us) | bn_bwd(10B/elem) {5*b/t_bwd/1e12:5.2f} TB/s ({t_bwd*1e6:6.1f} us) | reduce alone {2*b/(t_bwd-t_app)/1e12:5.2f} TB/s ({(t_bwd-t_app)*1e6:6.1f} us)")
