"""GPU parity, op level: the implicit-GEMM conv kernel (forward / fused epilogue / dgrad), the weight
gradient kernel and the NN helpers, through the C ABI, against torch fp32 on the CPU (floating-point
kernels -> a plain fp32 reference is the oracle form; tolerances stated per test)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"

TOL = {"f32": 2e-5, "bf16": 2e-2}     # max |err| relative to max |ref|


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * (hi - lo) + lo)


def _relerr(got, ref):
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


def _q(t, dtype):
    """operands as the kernel sees them (bf16-rounded in bf16 mode) so the check isolates the kernel"""
    return t.bfloat16().float() if dtype == "bf16" else t


CONV_CASES = [
    # B, Cin, H, W, Cout, k, s, p
    (2, 48, 16, 20, 48, 3, 1, 1),
    (2, 48, 16, 16, 96, 3, 2, 1),
    (1, 96, 12, 12, 96, 1, 1, 0),
    (3, 96, 10, 14, 192, 3, 2, 1),
    (2, 192, 8, 8, 192, 3, 1, 1),
    (2, 384, 6, 6, 96, 1, 1, 0),
    (1, 16, 24, 24, 48, 3, 1, 1),       # stem geometry (space-to-depth input)
    (2, 768, 4, 4, 384, 1, 1, 0),
]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward(case, dtype):
    from yolov5m_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    x = _q(_rand((B, Cin, H, W), 1), dtype)
    w = _q(_rand((Cout, Cin, k, k), 2, -0.2, 0.2), dtype)
    ref = F.conv2d(x, w, None, s, p)
    got = ops.conv_forward(x.to(DEV), w.to(DEV), s, p, dtype).cpu()
    assert got.shape == ref.shape
    assert _relerr(got, ref) < TOL[dtype], (case, dtype, _relerr(got, ref))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_conv_fused_epilogue(dtype):
    from yolov5m_amd import ops
    B, Cin, H, W, Cout, k, s, p = 2, 96, 10, 10, 96, 3, 1, 1
    x = _q(_rand((B, Cin, H, W), 3), dtype)
    w = _q(_rand((Cout, Cin, k, k), 4, -0.1, 0.1), dtype)
    sc, sh = _rand((Cout,), 5, 0.5, 1.5), _rand((Cout,), 6, -0.2, 0.2)
    res = _q(_rand((B, Cout, H, W), 7), dtype)
    y = F.conv2d(x, w, None, s, p) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    ref = F.silu(y) + res
    got = ops.conv_forward(x.to(DEV), w.to(DEV), s, p, dtype, scale=sc.to(DEV), shift=sh.to(DEV), act=True,
                           res=res.to(DEV)).cpu()
    assert _relerr(got, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_conv_stats_epilogue(dtype):
    from yolov5m_amd import ops
    B, Cin, H, W, Cout, k, s, p = 3, 48, 18, 18, 96, 3, 2, 1      # M = 243: partial pixel tile
    x = _q(_rand((B, Cin, H, W), 8), dtype)
    w = _q(_rand((Cout, Cin, k, k), 9, -0.2, 0.2), dtype)
    ref = F.conv2d(x, w, None, s, p)
    got, s1, s2 = ops.conv_forward_stats(x.to(DEV), w.to(DEV), s, p, dtype)
    assert _relerr(got.cpu(), ref) < TOL[dtype]
    np.testing.assert_allclose(s1.cpu().numpy(), ref.sum((0, 2, 3)).numpy(), rtol=1e-3, atol=1e-2)
    np.testing.assert_allclose(s2.cpu().numpy(), (ref * ref).sum((0, 2, 3)).numpy(), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES[:6])
def test_conv_dgrad(case, dtype):
    from yolov5m_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    x = _rand((B, Cin, H, W), 11).requires_grad_(True)
    w = _q(_rand((Cout, Cin, k, k), 12, -0.2, 0.2), dtype)
    y = F.conv2d(x, w, None, s, p)
    dy = _q(_rand(tuple(y.shape), 13), dtype)
    y.backward(dy)
    got = ops.conv_dgrad(dy.to(DEV), w.to(DEV), (H, W), s, p, dtype).cpu()
    assert _relerr(got, x.grad) < TOL[dtype], (case, dtype, _relerr(got, x.grad))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_wgrad(case, dtype):
    from yolov5m_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    x = _q(_rand((B, Cin, H, W), 21), dtype)
    w = _rand((Cout, Cin, k, k), 22, -0.2, 0.2).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = _q(_rand(tuple(y.shape), 23), dtype)
    y.backward(dy)
    got = ops.conv_wgrad(dy.to(DEV), x.to(DEV), k, s, p, dtype).cpu()
    assert _relerr(got, w.grad) < TOL[dtype], (case, dtype, _relerr(got, w.grad))
