#!/usr/bin/env python
"""pure-read HBM bandwidth reference on this box (torch reductions over tensors of the decode leg's size)"""
import torch
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
x = torch.randn(128 * 3 * 160 * 160 * 85, device="cuda")
b = x.numel() * 4
print(f"sum   {b / t(lambda: x.sum()) / 1e12:.2f} TB/s  ({b/1e9:.2f} GB)")
print(f"amax  {b / t(lambda: x.amax()) / 1e12:.2f} TB/s")
y = torch.empty_like(x)
print(f"copy  {2 * b / t(lambda: y.copy_(x)) / 1e12:.2f} TB/s (read+write)")
