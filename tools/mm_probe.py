#!/usr/bin/env python
"""Calibration: what torch.mm (hipBLASLt / rocBLAS) reaches on the 1x1-conv GEMM shapes of the step (out[M,N] = a[M,K] @ w[N,K]^T,
bf16, alone on the chip), next to a large square GEMM. Not used by the product path."""
import torch
dev = "cuda:0"
shapes = [(8192, 8192, 8192), (102400, 384, 384), (102400, 192, 384), (102400, 384, 192), (102400, 192, 768), (102400, 768, 384),
          (25600, 768, 768), (25600, 384, 768), (25600, 768, 1536), (25600, 1536, 768), (25600, 384, 384), (409600, 192, 384),
          (409600, 384, 192), (102400, 192, 192), (409600, 96, 96)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        o = torch.mm(a, w.t())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        o = torch.mm(a, w.t())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"M={M:7d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  {(M * (N + K) + N * K) * 2 / us / 1e6:6.2f} TB/s algorithmic", flush=True)
