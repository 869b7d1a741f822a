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


# short-K pointwise layers take the barrier-free streaming kernel (y5m_conv_pw.hip) in bf16: every
# (channel chunk, K steps) instance, several chunks per pixel stream, a stream count that does not divide
# the group count, and one pixel count (4606) that is not a multiple of the 16-pixel group (that layer
# must come out right through the tiled kernel)
PW_CASES = [
    # B, Cin, H, W, Cout
    (2, 48, 48, 50, 48),
    (2, 96, 48, 50, 48),
    (2, 192, 48, 50, 48),
    (2, 48, 48, 50, 96),
    (2, 96, 48, 50, 96),
    (2, 192, 48, 50, 192),
    (2, 96, 48, 50, 384),
    (2, 192, 48, 50, 144),
    (2, 96, 47, 49, 96),
    (5, 96, 80, 80, 96),
]


@pytest.mark.parametrize("case", PW_CASES)
def test_pointwise_forward_stats(case):
    from yolov5m_amd import ops
    B, Cin, H, W, Cout = case
    x = _q(_rand((B, Cin, H, W), 31), "bf16")
    w = _q(_rand((Cout, Cin, 1, 1), 32, -0.2, 0.2), "bf16")
    ref = F.conv2d(x, w)
    got, s1, s2 = ops.conv_forward_stats(x.to(DEV), w.to(DEV), 1, 0, "bf16")
    assert _relerr(got.cpu(), ref) < TOL["bf16"]
    np.testing.assert_allclose(s1.cpu().numpy(), ref.sum((0, 2, 3)).numpy(), rtol=1e-3, atol=2e-2)
    np.testing.assert_allclose(s2.cpu().numpy(), (ref * ref).sum((0, 2, 3)).numpy(), rtol=1e-3, atol=2e-2)


@pytest.mark.parametrize("res", [False, True])
@pytest.mark.parametrize("case", PW_CASES)
def test_pointwise_fused_epilogue(case, res):
    from yolov5m_amd import ops
    B, Cin, H, W, Cout = case
    x = _q(_rand((B, Cin, H, W), 33), "bf16")
    w = _q(_rand((Cout, Cin, 1, 1), 34, -0.2, 0.2), "bf16")
    sc, sh = _rand((Cout,), 35, 0.5, 1.5), _rand((Cout,), 36, -0.2, 0.2)
    r = _q(_rand((B, Cout, H, W), 37), "bf16") if res else None
    ref = F.silu(F.conv2d(x, w) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    if res:
        ref = ref + r
    got = ops.conv_forward(x.to(DEV), w.to(DEV), 1, 0, "bf16", scale=sc.to(DEV), shift=sh.to(DEV), act=True,
                           res=r.to(DEV) if res else None).cpu()
    assert _relerr(got, ref) < TOL["bf16"]


@pytest.mark.parametrize("acc", [False, True])
@pytest.mark.parametrize("case", PW_CASES)
def test_pointwise_dgrad(case, acc):
    """data gradient of a 1x1 conv = pointwise conv with the transposed weights, optionally accumulated
    onto an existing gradient (fan-out in the graph)"""
    from yolov5m_amd import ops
    B, Cout, H, W, Cin = case            # roles swapped: the kernel sees K = Cout of the forward conv
    x = _rand((B, Cin, H, W), 41).requires_grad_(True)
    w = _q(_rand((Cout, Cin, 1, 1), 42, -0.2, 0.2), "bf16")
    y = F.conv2d(x, w)
    dy = _q(_rand(tuple(y.shape), 43), "bf16")
    y.backward(dy)
    init = _q(_rand((B, Cin, H, W), 44), "bf16") if acc else None
    ref = x.grad + (init if acc else 0)
    got = ops.conv_dgrad(dy.to(DEV), w.to(DEV), (H, W), 1, 0, "bf16", init=init.to(DEV) if acc else None).cpu()
    assert _relerr(got, ref) < TOL["bf16"]


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
