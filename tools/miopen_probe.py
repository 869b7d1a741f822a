#!/usr/bin/env python
"""Calibration: torch's conv2d (MIOpen) forward / input gradient / weight gradient on the step's 3x3 shapes (bf16, channels_last,
B=64, alone on the chip). Not used by the product path. usage: miopen_probe.py [benchmark 0|1]"""
import sys, time, torch
import torch.nn.functional as F
dev = "cuda:0"
torch.backends.cudnn.benchmark = len(sys.argv) > 1 and sys.argv[1] == "1"
shapes = [(192, 192, 40, 1), (96, 96, 80, 1), (384, 384, 20, 1), (48, 48, 160, 1), (48, 96, 320, 2), (96, 192, 160, 2), (192, 384, 80, 2),
          (384, 768, 40, 2)]
B = 64
for cin, cout, hw, s in shapes:
    x = torch.randn(B, cin, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, device=dev, dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ho = (hw + 2 - 3) // s + 1
    flops = 2.0 * B * ho * ho * cout * cin * 9
    t0 = time.time()
    y = F.conv2d(x, w, None, s, 1)
    gy = torch.randn_like(y)
    torch.autograd.grad(y, (x, w), gy)
    torch.cuda.synchronize()
    setup = time.time() - t0
    res = []
    for what in ("fwd", "dgrad", "wgrad"):
        def run():
            if what == "fwd":
                with torch.no_grad():
                    return F.conv2d(x, w, None, s, 1)
            if what == "dgrad":
                return torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])
            return torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        res.append(f"{what} {us:8.1f} us {flops / us / 1e6:7.1f} TF/s")
    print(f"{cin:4d} -> {cout:4d} 3x3 s{s} @ {ho}x{ho} (first call {setup:5.1f} s): " + " | ".join(res), flush=True)
