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


# 3x3 / stride-1 layers with >= 64 input channels and N % 96 == 0 run on the persistent halo-patch kernel
# (y5m_conv_halo.hip) in bf16: 32-channel last slab (C = 96), one and several 64-channel slabs, one and two channel
# tiles, image borders on all sides with several images per launch, a pixel count that is not a multiple of the
# 256-pixel tile, fewer pixels than one tile, and more tiles than workgroups (the persistent loop's second round).
HALO_CASES = [
    # B, Cin, H, W, Cout
    (2, 96, 10, 14, 96),
    (3, 64, 9, 11, 192),
    (2, 192, 20, 20, 192),
    (1, 192, 12, 12, 384),
    (2, 384, 8, 8, 384),
    (1, 96, 80, 80, 96),
    (1, 128, 6, 7, 96),
    (2, 96, 400, 96, 96),
]


def _halo_expected(Cin, N, W=0):
    """the library's dispatch rule (y5m_conv_halo.hip halo_geom): 192-channel tiles (Y5M_CONV_HALO=0 turns the kernel off); images
    wider than 44 pixels only with the two-stage weight ring (Y5M_CONV_HALO_NS2=1, up to 88)"""
    import os
    lvl = int(os.environ.get("Y5M_CONV_HALO", "1"))
    wmax = 88 if os.environ.get("Y5M_CONV_HALO_NS2") == "1" else 44
    return "halo" if lvl >= 1 and N % 192 == 0 and Cin % 64 == 0 and W <= wmax else None


@pytest.mark.parametrize("case", HALO_CASES)
def test_halo_forward_stats_and_epilogue(case):
    from yolov5m_amd import ops
    B, Cin, H, W, Cout = case
    want = _halo_expected(Cin, Cout, W)
    x = _q(_rand((B, Cin, H, W), 61), "bf16")
    w = _q(_rand((Cout, Cin, 3, 3), 62, -0.1, 0.1), "bf16")
    ref = F.conv2d(x, w, None, 1, 1)
    got, s1, s2 = ops.conv_forward_stats(x.to(DEV), w.to(DEV), 1, 1, "bf16")
    assert want is None or ops.LAST_KERNEL == want
    assert _relerr(got.cpu(), ref) < TOL["bf16"], _relerr(got.cpu(), ref)
    np.testing.assert_allclose(s1.cpu().numpy(), ref.sum((0, 2, 3)).numpy(), rtol=2e-3, atol=0.02 * float(ref.abs().max()) * 8)
    np.testing.assert_allclose(s2.cpu().numpy(), (ref * ref).sum((0, 2, 3)).numpy(), rtol=2e-3, atol=0.05)
    sc, sh = _rand((Cout,), 63, 0.5, 1.5), _rand((Cout,), 64, -0.2, 0.2)
    res = _q(_rand((B, Cout, H, W), 65), "bf16")
    ref2 = F.silu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + res
    got2 = ops.conv_forward(x.to(DEV), w.to(DEV), 1, 1, "bf16", scale=sc.to(DEV), shift=sh.to(DEV), act=True,
                            res=res.to(DEV)).cpu()
    assert want is None or ops.LAST_KERNEL == want
    assert _relerr(got2, ref2) < TOL["bf16"]


# images 45..88 pixels wide (the 80x80 stage of a 1280x1280 model): two patch buffers + THREE weight stages exceed the LDS, so
# these shapes run on the tiled kernel unless Y5M_CONV_HALO_NS2=1 selects the halo kernel's two-stage ring (round 5, default off)
HALO_WIDE_CASES = [(1, 192, 10, 80, 192), (2, 64, 9, 56, 192), (1, 384, 6, 88, 384), (3, 192, 13, 47, 192)]


def test_halo_two_stage_ring_wide_images_subprocess():
    """Y5M_CONV_HALO_NS2=1 in a child (the knob is read once): forward with statistics, the fused inference epilogue and the three
    data-gradient modes of HALO_WIDE_CASES on conv_halo_kernel<6,*,ns2>; Y5M_PERSIST_CUS=3 makes every workgroup walk several
    tiles, so the ring's parity is carried across slabs and tiles"""
    import os, subprocess, sys
    if os.environ.get("Y5M_CONV_HALO_NS2") == "1":
        pytest.skip("already the child")
    env = dict(os.environ, Y5M_CONV_HALO_NS2="1", Y5M_PERSIST_CUS="3")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-k", "halo_wide"], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("case", HALO_WIDE_CASES)
def test_halo_wide_forward_and_dgrad(case):
    """the wide-image cases through the same checks as HALO_CASES; which kernel runs them depends on Y5M_CONV_HALO_NS2 (the child
    of test_halo_two_stage_ring_wide_images_subprocess asserts it is the halo kernel's two-stage form)"""
    import os
    from yolov5m_amd import ops
    ns2 = os.environ.get("Y5M_CONV_HALO_NS2") == "1"
    B, Cin, H, W, Cout = case
    x = _q(_rand((B, Cin, H, W), 61), "bf16")
    w = _q(_rand((Cout, Cin, 3, 3), 62, -0.1, 0.1), "bf16")
    ref = F.conv2d(x, w, None, 1, 1)
    got, s1, s2 = ops.conv_forward_stats(x.to(DEV), w.to(DEV), 1, 1, "bf16")
    assert (ops.LAST_KERNEL == "halo") == ns2, ops.LAST_KERNEL
    assert _relerr(got.cpu(), ref) < TOL["bf16"], _relerr(got.cpu(), ref)
    np.testing.assert_allclose(s1.cpu().numpy(), ref.sum((0, 2, 3)).numpy(), rtol=2e-3, atol=0.02 * float(ref.abs().max()) * 8)
    np.testing.assert_allclose(s2.cpu().numpy(), (ref * ref).sum((0, 2, 3)).numpy(), rtol=2e-3, atol=0.05)
    sc, sh = _rand((Cout,), 63, 0.5, 1.5), _rand((Cout,), 64, -0.2, 0.2)
    res = _q(_rand((B, Cout, H, W), 65), "bf16")
    ref2 = F.silu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + res
    got2 = ops.conv_forward(x.to(DEV), w.to(DEV), 1, 1, "bf16", scale=sc.to(DEV), shift=sh.to(DEV), act=True, res=res.to(DEV)).cpu()
    assert _relerr(got2, ref2) < TOL["bf16"]
    for mode in ("plain", "init", "src"):
        test_halo_dgrad(case, mode)


@pytest.mark.parametrize("mode", ["plain", "init", "src"])
@pytest.mark.parametrize("case", HALO_CASES[:6])
def test_halo_dgrad(case, mode):
    """data gradient of the same layers (mirrored taps, transposed weights): plain store, accumulation onto an existing
    gradient, and the fused residual source"""
    from yolov5m_amd import ops
    B, Cin, H, W, Cout = case
    dy = _q(_rand((B, Cout, H, W), 66), "bf16")
    w = _q(_rand((Cout, Cin, 3, 3), 67, -0.1, 0.1), "bf16")
    ref = F.conv_transpose2d(dy, w, None, 1, 1)
    kw = {}
    if mode != "plain":
        extra = _q(_rand((B, Cin, H, W), 68), "bf16")
        ref = ref + extra
        kw = {"init": extra.to(DEV)} if mode == "init" else {"src": extra.to(DEV)}
    want = _halo_expected(Cout, Cin, W)            # the data gradient has Cout input and Cin output channels
    got = ops.conv_dgrad(dy.to(DEV), w.to(DEV), (H, W), 1, 1, "bf16", **kw).cpu()
    assert want is None or ops.LAST_KERNEL == want
    assert _relerr(got, ref) < TOL["bf16"], _relerr(got, ref)


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


@pytest.mark.parametrize("shape", [(2, 40, 36), (1, 64, 64), (3, 24, 40)])
def test_stem_streaming_kernel(shape):
    """the stem geometry (3x3 / stride 1 / pad 1 over the 16-channel space-to-depth image, 48 output channels) runs on the
    tapped mode of the streaming kernel in bf16: training statistics epilogue and the folded-BN inference epilogue,
    image borders on every side, several images per launch"""
    from yolov5m_amd import ops
    B, H, W = shape
    x = _q(_rand((B, 16, H, W), 41), "bf16")
    x[:, 12:] = 0                                             # the 4 padding channels of the space-to-depth input
    w = _q(_rand((48, 16, 3, 3), 42, -0.2, 0.2), "bf16")
    ref = F.conv2d(x, w, None, 1, 1)
    got, s1, s2 = ops.conv_forward_stats(x.to(DEV), w.to(DEV), 1, 1, "bf16")
    assert _relerr(got.cpu(), ref) < TOL["bf16"]
    np.testing.assert_allclose(s1.cpu().numpy(), ref.sum((0, 2, 3)).numpy(), rtol=2e-3, atol=5e-2)
    np.testing.assert_allclose(s2.cpu().numpy(), (ref * ref).sum((0, 2, 3)).numpy(), rtol=2e-3, atol=5e-2)
    sc, sh = _rand((48,), 43, 0.5, 1.5), _rand((48,), 44, -0.2, 0.2)
    ref2 = F.silu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    got2 = ops.conv_forward(x.to(DEV), w.to(DEV), 1, 1, "bf16", scale=sc.to(DEV), shift=sh.to(DEV), act=True).cpu()
    assert _relerr(got2, ref2) < TOL["bf16"]


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


# ---- 1x1 layers with >= 384 input channels and N % 192 == 0 run on the persistent two-phase GEMM kernel (y5m_conv_gemm.hip)
# in bf16 once a launch has enough 256-pixel x 192-channel work items for one workgroup per CU (Y5M_CONV_GEMM8_MIN, default
# 192): the last case below does by itself (876 items: several tiles per workgroup, two channel tiles); the small ones --
# pixel counts that are not a multiple of the tile, an odd number of 64-channel units, one / two / four channel tiles --
# reach it in the child process of test_gemm8_small_shapes_subprocess, which lowers the threshold to 1.
GEMM8_CASES = [
    # B, Cin, H, W, Cout
    (3, 384, 9, 11, 192),
    (2, 448, 20, 20, 384),
    (1, 768, 16, 16, 768),
    (70, 384, 40, 40, 384),
]


def _gemm8_expected(case, dgrad=False):
    """the library's dispatch rule (y5m_conv_gemm.hip gemm8_geom): Y5M_CONV_GEMM8 = 2 (default) takes the forward
    epilogues only, 1 also the data gradients (the child process of test_gemm8_small_shapes_subprocess)"""
    import os
    B, Cin, H, W, Cout = case
    mode = int(os.environ.get("Y5M_CONV_GEMM8", "2"))
    if mode == 0 or (mode == 2 and dgrad) or (mode == 3 and not dgrad):
        return None
    items = (B * H * W + 255) // 256 * (Cout // 192)
    return "gemm8" if items >= int(os.environ.get("Y5M_CONV_GEMM8_MIN", "192")) else None


def test_gemm8_small_shapes_subprocess():
    import os, subprocess, sys
    if os.environ.get("Y5M_CONV_GEMM8_MIN") == "1":
        pytest.skip("already the child")
    env = dict(os.environ, Y5M_CONV_GEMM8_MIN="1", Y5M_CONV_GEMM8="1")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-k", "gemm8 and not subprocess"], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("case", GEMM8_CASES)
def test_gemm8_forward_epilogues(case):
    """plain store, folded BN + SiLU + residual, and raw + statistics (partial rows) against torch fp32"""
    from yolov5m_amd import ops
    B, Cin, H, W, Cout = case
    x = _q(_rand((B, Cin, H, W), 61), "bf16")
    w = _q(_rand((Cout, Cin, 1, 1), 62, -0.1, 0.1), "bf16")
    ref = F.conv2d(x, w)
    exp = _gemm8_expected(case)
    got = ops.conv_forward(x.to(DEV), w.to(DEV), 1, 0, "bf16").cpu()          # (plain store = the data-gradient epilogue)
    assert _gemm8_expected(case, dgrad=True) is None or ops.LAST_KERNEL == "gemm8", ops.LAST_KERNEL
    assert _relerr(got, ref) < TOL["bf16"]
    sc, sh = _rand((Cout,), 63, 0.5, 1.5), _rand((Cout,), 64, -0.2, 0.2)
    r = _q(_rand((B, Cout, H, W), 65), "bf16")
    ref2 = F.silu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + r
    got2 = ops.conv_forward(x.to(DEV), w.to(DEV), 1, 0, "bf16", scale=sc.to(DEV), shift=sh.to(DEV), act=True, res=r.to(DEV)).cpu()
    assert exp is None or ops.LAST_KERNEL == exp, ops.LAST_KERNEL
    assert _relerr(got2, ref2) < TOL["bf16"]
    got3, s1, s2 = ops.conv_forward_stats(x.to(DEV), w.to(DEV), 1, 0, "bf16")
    assert exp is None or ops.LAST_KERNEL == exp, ops.LAST_KERNEL
    assert _relerr(got3.cpu(), ref) < TOL["bf16"]
    np.testing.assert_allclose(s1.cpu().numpy(), ref.sum((0, 2, 3)).numpy(), rtol=2e-3, atol=5e-2)
    np.testing.assert_allclose(s2.cpu().numpy(), (ref * ref).sum((0, 2, 3)).numpy(), rtol=2e-3, atol=5e-2)


@pytest.mark.parametrize("case", GEMM8_CASES)
def test_gemm8_bn_accumulator_rows(case):
    from yolov5m_amd import ops
    from yolov5m_amd.arch import BN_EPS, BN_MOMENTUM
    B, Cin, H, W, Cout = case
    x = _q(_rand((B, Cin, H, W), 66), "bf16")
    w = _q(_rand((Cout, Cin, 1, 1), 67, -0.1, 0.1), "bf16")
    gamma, beta = _rand((Cout,), 68, 0.5, 1.5), _rand((Cout,), 69, -0.3, 0.3)
    ref = F.conv2d(x.double(), w.double())
    mean, var = ref.mean((0, 2, 3)), ref.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + BN_EPS)
    y, z, g_scale, g_shift, g_mean, g_invstd, _rm, _rv = ops.conv_forward_bn_fused(
        x.to(DEV), w.to(DEV), 1, 0, gamma.to(DEV), beta.to(DEV), torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV),
        BN_MOMENTUM, BN_EPS, "bf16", repeats=2)
    exp = _gemm8_expected(case)
    assert exp is None or ops.LAST_KERNEL == exp, ops.LAST_KERNEL
    assert _relerr(y.cpu(), ref.float()) < TOL["bf16"]
    np.testing.assert_allclose(g_mean.cpu().numpy(), mean.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(g_invstd.cpu().numpy(), invstd.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(g_scale.cpu().numpy(), (gamma.double() * invstd).numpy(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("mode", ["plain", "init", "src"])
@pytest.mark.parametrize("case", GEMM8_CASES)
def test_gemm8_dgrad(case, mode):
    """data gradient of a 1x1 conv Cout -> Cin seen from the kernel: K = Cout (>= 384), N = Cin; store, accumulate in place,
    and the fused residual source"""
    from yolov5m_amd import ops
    B, Cin, H, W, Cout = case                 # roles swapped: the FORWARD conv maps Cout_k <- Cin_k with Cin_k = Cout here
    fw_cin, fw_cout = Cout, Cin               # forward conv: fw_cin -> fw_cout; its dgrad has K = fw_cout, N = fw_cin
    x = _rand((B, fw_cin, H, W), 71).requires_grad_(True)
    w = _q(_rand((fw_cout, fw_cin, 1, 1), 72, -0.1, 0.1), "bf16")
    dy = _q(_rand((B, fw_cout, H, W), 73), "bf16")
    F.conv2d(x, w).backward(dy)
    ref = x.grad
    extra = _q(_rand((B, fw_cin, H, W), 74), "bf16")
    kw = {"init": extra.to(DEV)} if mode == "init" else {"src": extra.to(DEV)} if mode == "src" else {}
    got = ops.conv_dgrad(dy.to(DEV), w.to(DEV), (H, W), 1, 0, "bf16", **kw).cpu()
    if fw_cout >= 384 and fw_cin % 192 == 0 and _gemm8_expected((B, fw_cout, H, W, fw_cin), dgrad=True):
        assert ops.LAST_KERNEL == "gemm8", ops.LAST_KERNEL
    assert _relerr(got, ref + (extra if mode != "plain" else 0)) < TOL["bf16"]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", [CONV_CASES[0], CONV_CASES[1], CONV_CASES[4], (2, 96, 48, 50, 96, 1, 1, 0)])
def test_conv_dgrad_fused_residual_source(case, dtype):
    """dx = src + dgrad with src in its own tensor (bottleneck residual gradient fused into the first conv's
    data gradient), stride 1 and the four parity classes of stride 2"""
    from yolov5m_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    x = _rand((B, Cin, H, W), 71).requires_grad_(True)
    w = _q(_rand((Cout, Cin, k, k), 72, -0.2, 0.2), dtype)
    y = F.conv2d(x, w, None, s, p)
    dy = _q(_rand(tuple(y.shape), 73), dtype)
    y.backward(dy)
    src = _q(_rand((B, Cin, H, W), 74), dtype)
    got = ops.conv_dgrad(dy.to(DEV), w.to(DEV), (H, W), s, p, dtype, src=src.to(DEV)).cpu()
    assert _relerr(got, x.grad + src) < TOL[dtype]


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


# ---- wide-layer weight gradients: the 192 x 96 and 96 x 96 block tiles of wgrad_kernel. Ragged pixel counts (tail chunk),
# widths that are not a multiple of the 64-pixel chunk (rows and images change inside a chunk), stride 2, channel counts
# that leave a partial channel tile, and enough pixels for several chunks per split-K range.
WGRAD_WIDE_CASES = [
    # B, Cin, H, W, Cout, k, s, p
    (2, 96, 12, 12, 96, 3, 1, 1),
    (3, 192, 20, 20, 192, 3, 1, 1),
    (1, 384, 7, 9, 192, 3, 2, 1),
    (5, 96, 40, 40, 96, 3, 1, 1),
    (2, 144, 10, 14, 384, 3, 1, 1),
    (16, 192, 40, 40, 192, 3, 1, 1),
    (2, 192, 9, 13, 384, 3, 2, 1),
    (3, 384, 10, 10, 384, 3, 1, 1),
]


@pytest.mark.parametrize("case", WGRAD_WIDE_CASES)
def test_conv_wgrad_wide_layers(case):
    from yolov5m_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    x = _q(_rand((B, Cin, H, W), 21), "bf16")
    w = _rand((Cout, Cin, k, k), 22, -0.2, 0.2).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = _q(_rand(tuple(y.shape), 23), "bf16")
    y.backward(dy)
    got = ops.conv_wgrad(dy.to(DEV), x.to(DEV), k, s, p, "bf16").cpu()
    assert ops.LAST_WGRAD_KERNEL.startswith("wgrad_kernel"), ops.LAST_WGRAD_KERNEL
    assert _relerr(got, w.grad) < TOL["bf16"], (case, ops.LAST_WGRAD_KERNEL, _relerr(got, w.grad))


# ---- wgrad_rows_kernel: 3x3 weight gradients of the layers with 48 input channels, one kernel row per block, runs of 32 output
# pixels of one row. Widths that are not a multiple of 32 (masked run tails), a single partial run per row, stride 2 with odd
# input sizes, 48 and 96 output channels (4 x 1 and 2 x 2 wave arrangements), enough pixels for several chunks per pixel range.
WGRAD_ROWS_CASES = [
    # B, Cin, H, W, Cout, k, s, p
    (2, 48, 12, 40, 48, 3, 1, 1),
    (1, 48, 9, 17, 48, 3, 1, 1),
    (3, 48, 20, 64, 48, 3, 1, 1),
    (2, 48, 24, 50, 96, 3, 2, 1),
    (1, 48, 11, 13, 96, 3, 2, 1),
    (4, 48, 64, 64, 96, 3, 2, 1),
    (2, 48, 16, 96, 96, 3, 1, 1),
    (2, 48, 33, 70, 48, 3, 2, 1),
    (8, 48, 80, 80, 96, 3, 2, 1),
    (8, 48, 40, 40, 48, 3, 1, 1),
]


@pytest.mark.parametrize("case", WGRAD_ROWS_CASES)
def test_conv_wgrad_rows_kernel(case):
    from yolov5m_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    x = _q(_rand((B, Cin, H, W), 31), "bf16")
    w = _rand((Cout, Cin, k, k), 32, -0.2, 0.2).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = _q(_rand(tuple(y.shape), 33), "bf16")
    y.backward(dy)
    got = ops.conv_wgrad(dy.to(DEV), x.to(DEV), k, s, p, "bf16").cpu()
    assert ops.LAST_WGRAD_KERNEL.startswith("wgrad_rows_kernel"), ops.LAST_WGRAD_KERNEL
    assert _relerr(got, w.grad) < TOL["bf16"], (case, ops.LAST_WGRAD_KERNEL, _relerr(got, w.grad))
    # per tap: an error confined to one tap (a wrong row offset / stride) must not hide in the norm of the other eight
    for t in range(9):
        assert _relerr(got[:, :, t // 3, t % 3], w.grad[:, :, t // 3, t % 3]) < TOL["bf16"], (case, t)


# ---- fused pointwise backward (csrc/y5m_bwd_pw.hip): BatchNorm+SiLU backward apply + data gradient + weight gradient of a
# 1x1 CBL in one launch, against torch autograd of the same layer on the same bf16 operands. Channel counts 48 / 96 / 192
# (the three tile geometries), pixel counts that leave a tile tail and that give some workgroups no tile at all, two
# BatchNorm segments (the merged C3 pair), plain / accumulate-onto / separate-source epilogues.
BWD_PW_CASES = [
    # B, C, H, W, split
    (2, 48, 20, 24, None),
    (3, 96, 17, 19, None),
    (2, 192, 12, 10, None),
    (5, 192, 40, 40, 96),
    (4, 96, 80, 80, 48),
    (2, 48, 160, 160, None),
    (64, 192, 20, 20, None),
]


@pytest.mark.parametrize("mode", ["plain", "init", "src"])
@pytest.mark.parametrize("case", BWD_PW_CASES)
def test_bwd_pw_fused(case, mode):
    from yolov5m_amd import ops
    B, C, H, W, split = case
    x = _q(_rand((B, C, H, W), 31, -1, 1), "bf16")
    w = _rand((C, C, 1, 1), 32, -0.25, 0.25)
    wq = _q(w, "bf16")
    gamma, beta = _rand((C,), 33, 0.5, 1.5), _rand((C,), 34, -0.3, 0.3)
    yq = _q(F.conv2d(x, wq), "bf16").requires_grad_(True)          # the raw conv output as the engine stores it
    dz = _q(_rand((B, C, H, W), 35, -1, 1), "bf16")
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.silu(F.batch_norm(yq, None, None, g_, b_, True, 0.03, 1e-3))
    z.backward(dz)
    dyq = _q(yq.grad, "bf16")                                       # dy reaches both GEMMs rounded to bf16
    dx_ref = F.conv_transpose2d(dyq, wq)
    dw_ref = torch.nn.grad.conv2d_weight(x, w.shape, dyq)
    extra = _q(_rand((B, C, H, W), 36, -1, 1), "bf16") if mode != "plain" else None
    if extra is not None:
        dx_ref = dx_ref + extra
    mean = yq.detach().mean((0, 2, 3))
    var = yq.detach().var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-3)
    scale = gamma * invstd
    shift = beta - mean * scale
    dev = lambda t: t.to(DEV)
    dx, dw, dg, db = ops.bwd_pw(dev(dz), dev(yq.detach()), dev(x), dev(w), dev(scale), dev(shift), dev(mean), dev(invstd), split=split,
                                init=dev(extra) if mode == "init" else None, src=dev(extra) if mode == "src" else None)
    assert _relerr(dx.cpu(), dx_ref) < TOL["bf16"], (case, mode, _relerr(dx.cpu(), dx_ref))
    assert _relerr(dw.cpu(), dw_ref) < TOL["bf16"], (case, mode, _relerr(dw.cpu(), dw_ref))
    np.testing.assert_allclose(dg.cpu().numpy(), g_.grad.numpy(), rtol=2e-3, atol=2e-3 * float(g_.grad.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), b_.grad.numpy(), rtol=2e-3, atol=2e-3 * float(b_.grad.abs().max()))


# ---- fused stem backward (csrc/y5m_bwd_stem.hip): BatchNorm+SiLU backward apply + 9-tap weight gradient of the stem in its executed
# form (3x3 / stride 1 / pad 1 over a 16-channel image, 48 output channels), against torch autograd on the same bf16 operands.
# Widths that are / are not multiples of the 32-pixel run, fewer chunks than workgroups and several chunks per workgroup, a
# single image row, and per-tap comparison (a wrong row segment / pixel offset must not hide in the other taps' norm).
BWD_STEM_CASES = [(2, 24, 40), (1, 9, 17), (3, 1, 70), (4, 96, 96), (64, 64, 64), (1, 5, 32)]


@pytest.mark.parametrize("case", BWD_STEM_CASES)
def test_bwd_stem_fused(case):
    from yolov5m_amd import ops
    B, H, W = case
    N, C = 48, 16
    x = _q(_rand((B, C, H, W), 61, -1, 1), "bf16")
    w = _rand((N, C, 3, 3), 62, -0.25, 0.25)
    gamma, beta = _rand((N,), 63, 0.5, 1.5), _rand((N,), 64, -0.3, 0.3)
    yq = _q(F.conv2d(x, _q(w, "bf16"), None, 1, 1), "bf16").requires_grad_(True)
    dz = _q(_rand((B, N, H, W), 65, -1, 1), "bf16")
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.silu(F.batch_norm(yq, None, None, g_, b_, True, 0.03, 1e-3))
    z.backward(dz)
    dyq = _q(yq.grad, "bf16")                                       # dy reaches the MFMAs rounded to bf16
    dw_ref = torch.nn.grad.conv2d_weight(x, w.shape, dyq, 1, 1)
    mean = yq.detach().mean((0, 2, 3))
    var = yq.detach().var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-3)
    scale = gamma * invstd
    shift = beta - mean * scale
    dev = lambda t: t.to(DEV)
    dw, dg, db = ops.bwd_stem(dev(dz), dev(yq.detach()), dev(x), dev(scale), dev(shift), dev(mean), dev(invstd))
    assert _relerr(dw.cpu(), dw_ref) < TOL["bf16"], (case, _relerr(dw.cpu(), dw_ref))
    for t in range(9):
        assert _relerr(dw.cpu()[:, :, t // 3, t % 3], dw_ref[:, :, t // 3, t % 3]) < TOL["bf16"], (case, t)
    np.testing.assert_allclose(dg.cpu().numpy(), g_.grad.numpy(), rtol=2e-3, atol=2e-3 * float(g_.grad.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), b_.grad.numpy(), rtol=2e-3, atol=2e-3 * float(b_.grad.abs().max()))


# ---- BatchNorm statistics through accumulator rows (y5m_conv_args.bn_acc + y5m_bn_act_fused, csrc/y5m_bnfuse.h): the conv
# launch adds its channel sums as f64 atomics, the normalise launch derives scale / shift / mean / invstd and the running
# statistics itself (reference model.py:17 BatchNorm2d(eps=1e-3, momentum=0.03) in train mode, :20 SiLU); against torch
# batch statistics of the same conv. One case per kernel family (tiled 3x3 stride 2, halo-patch 3x3, pointwise streaming),
# pixel counts that are not a multiple of any tile, two channel ranges on one launch (the merged C3 pair), more tiles than
# workgroups, and the whole sequence twice (running statistics updated twice).
BNFUSE_CASES = [
    # B, Cin, H, W, Cout, k, s, p, dtype, split, expected kernel family
    (3, 48, 18, 18, 96, 3, 2, 1, "f32", None, "tiled"),
    (3, 48, 18, 18, 96, 3, 2, 1, "bf16", None, "tiled"),
    (2, 192, 20, 20, 192, 3, 1, 1, "bf16", None, "halo"),
    (3, 64, 9, 11, 192, 3, 1, 1, "bf16", None, "halo"),
    (2, 96, 48, 50, 96, 1, 1, 0, "bf16", 48, "pointwise"),
    (5, 96, 80, 80, 192, 1, 1, 0, "bf16", 96, "pointwise"),
    (2, 768, 4, 4, 384, 1, 1, 0, "bf16", None, "tiled"),
    (40, 192, 20, 20, 192, 3, 1, 1, "bf16", None, "halo"),      # more tiles than workgroups: several tiles per block
]


@pytest.mark.parametrize("case", BNFUSE_CASES)
def test_conv_bn_accumulator_rows(case):
    from yolov5m_amd import ops
    from yolov5m_amd.arch import BN_EPS, BN_MOMENTUM
    B, Cin, H, W, Cout, k, s, p, dtype, split, family = case
    x = _q(_rand((B, Cin, H, W), 51), dtype)
    w = _q(_rand((Cout, Cin, k, k), 52, -0.2, 0.2), dtype)
    gamma, beta = _rand((Cout,), 53, 0.5, 1.5), _rand((Cout,), 54, -0.3, 0.3)
    rm0, rv0 = _rand((Cout,), 55, -0.5, 0.5), _rand((Cout,), 56, 0.5, 2.0)
    ref = F.conv2d(x.double(), w.double(), None, s, p)
    n = ref.numel() // Cout
    mean, var = ref.mean((0, 2, 3)), ref.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + BN_EPS)
    scale = gamma.double() * invstd
    shift = beta.double() - mean * scale
    unb = var * n / (n - 1)
    rm, rv = rm0.double(), rv0.double()
    for _ in range(2):
        rm = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
        rv = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unb
    y, z, g_scale, g_shift, g_mean, g_invstd, g_rm, g_rv = ops.conv_forward_bn_fused(
        x.to(DEV), w.to(DEV), s, p, gamma.to(DEV), beta.to(DEV), rm0.to(DEV), rv0.to(DEV), BN_MOMENTUM, BN_EPS, dtype,
        split=split, repeats=2)
    assert ops.LAST_KERNEL == family, (ops.LAST_KERNEL, family)
    assert _relerr(y.cpu(), ref.float()) < TOL[dtype]
    # the sums are taken over the f32 accumulators (before the bf16 store), so the statistics are f32-accurate in both modes
    tol = dict(rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(g_mean.cpu().numpy(), mean.numpy(), **tol)
    np.testing.assert_allclose(g_invstd.cpu().numpy(), invstd.numpy(), **tol)
    np.testing.assert_allclose(g_scale.cpu().numpy(), scale.numpy(), **tol)
    np.testing.assert_allclose(g_shift.cpu().numpy(), shift.numpy(), **tol)
    np.testing.assert_allclose(g_rm.cpu().numpy(), rm.numpy(), **tol)
    np.testing.assert_allclose(g_rv.cpu().numpy(), rv.numpy(), **tol)
    # the activated output, from the raw output AS STORED (bf16-rounded in bf16 mode) and the reference coefficients
    zref = F.silu(_q(y.cpu(), dtype).double() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    assert _relerr(z.cpu(), zref.float()) < (1e-2 if dtype == "bf16" else 2e-5)


def test_bn_unfused_engine_subprocess():
    """Y5M_BN_FUSE=0 keeps the engine's three-launch form (partial rows + y5m_bn_finalize / y5m_bn_bwd) alive for A/B runs:
    the f32 train-step golden runs that way in a child process"""
    import os, subprocess, sys
    if os.environ.get("Y5M_BN_FUSE") == "0":
        pytest.skip("already the child")
    env = dict(os.environ, Y5M_BN_FUSE="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(os.path.dirname(__file__), "test_gpu_model.py"), "-q", "-x",
                        "-k", "train_step_grads_f32_golden and default"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ---- training-mode BatchNorm helpers (y5m_bn_finalize / y5m_bn_act / y5m_bn_bwd) against torch fp32 ----
BN_CASES = [
    # M (pixels), C, ld (row pitch >= C: concat-slice views)
    (1000, 48, 48),
    (5000, 96, 192),
    (4099, 192, 192),
    (777, 384, 768),
    (300, 768, 768),
    (70000, 48, 48),
    (400000, 48, 48),       # more pixel rows than the grid covers in one pass: the unrolled grid-stride loops
    (250000, 96, 192),
]


def _bn_inputs(M, C, ld, seed):
    y = _rand((M, ld), seed, -2.0, 2.0)
    dz = _rand((M, ld), seed + 1, -1.0, 1.0)
    gamma, beta = _rand((C,), seed + 2, 0.5, 1.5), _rand((C,), seed + 3, -0.3, 0.3)
    return y, dz, gamma, beta


@pytest.mark.parametrize("case", BN_CASES)
def test_bn_finalize_from_partials(case):
    """per-tile (sum, sumsq) partials [rows][2][Np] -> mean/var/scale/shift + running stats (momentum 0.03,
    unbiased running variance), as nn.BatchNorm2d in train mode. Several different inputs go through the SAME
    workspace back to back: the cross-workgroup hand-off (stage rows + ticket counters) must never serve a
    stale value and must leave the counters at zero."""
    from yolov5m_amd import _lib
    M, C, _ = case
    L = _lib.lib()
    Np = (C + 95) // 96 * 96
    rows = (M + 127) // 128
    d = lambda t: t.to(DEV).contiguous()
    wsb = L.y5m_bn_finalize_workspace_bytes(Np)
    ws = torch.zeros(wsb, dtype=torch.uint8, device=DEV)
    out = torch.zeros((4, C), device=DEV)
    for seed in (51, 151, 251, 351):
        y = _rand((M, C), seed, -2.0, 2.0) * (1.0 + seed / 100.0) + seed / 200.0
        pad = rows * 128 - M
        yp = torch.cat([y, torch.zeros((pad, C))]) if pad else y
        blk = yp.view(rows, 128, C)
        part = torch.zeros((rows, 2, Np))
        part[:, 0, :C] = blk.sum(1)
        part[:, 1, :C] = (blk * blk).sum(1)
        gamma, beta = _rand((C,), seed + 1, 0.5, 1.5), _rand((C,), seed + 2, -0.3, 0.3)
        rm, rv = _rand((C,), seed + 3), _rand((C,), seed + 4, 0.5, 1.5)
        bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.03)
        with torch.no_grad():
            bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
        bn.train()
        bn(y.t().reshape(1, C, M, 1))
        mean = y.double().mean(0)
        var = y.double().var(0, unbiased=False)
        partd, g_, b_, rm_, rv_ = d(part), d(gamma), d(beta), d(rm), d(rv)
        _lib.check(L.y5m_bn_finalize(_lib.ptr(partd), rows, Np, C, M, _lib.ptr(g_), _lib.ptr(b_), _lib.ptr(rm_), _lib.ptr(rv_),
                                     0.03, 1e-3, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                     1, _lib.ptr(ws), wsb, _lib.stream_ptr()), "bn_finalize")
        torch.cuda.synchronize()
        invstd = 1.0 / torch.sqrt(var + 1e-3)
        np.testing.assert_allclose(out[2].cpu().numpy(), mean.float().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(out[3].cpu().numpy(), invstd.float().numpy(), rtol=1e-4)
        np.testing.assert_allclose(out[0].cpu().numpy(), (gamma * invstd.float()).numpy(), rtol=1e-4)
        np.testing.assert_allclose(out[1].cpu().numpy(), (beta - mean.float() * gamma * invstd.float()).numpy(), rtol=2e-4, atol=1e-4)
        np.testing.assert_allclose(rm_.cpu().numpy(), bn.running_mean.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(rv_.cpu().numpy(), bn.running_var.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", BN_CASES)
def test_bn_act_and_backward(case, dtype):
    """z = silu(y*scale + shift) and its autograd (dy, dgamma, dbeta) with batch statistics, through
    strided (ptr, ld) views"""
    from yolov5m_amd import _lib
    from yolov5m_amd._lib import F32, BF16, ACT_SILU
    M, C, ld = case
    L = _lib.lib()
    tdt, dt = (torch.float32, F32) if dtype == "f32" else (torch.bfloat16, BF16)
    y, dz, gamma, beta = _bn_inputs(M, C, ld, 61)
    y, dz = _q(y, dtype), _q(dz, dtype)
    yc = y[:, :C].clone().double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mean, var = yc.mean(0), yc.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-3)
    z = F.silu((yc - mean) * invstd * g64 + b64)
    z.backward(dz[:, :C].double())
    scale = (gamma.double() * invstd.detach()).float()
    shift = (beta.double() - mean.detach() * gamma.double() * invstd.detach()).float()
    d = lambda t: t.to(DEV).contiguous()
    yd, dzd = d(y.to(tdt)), d(dz.to(tdt))
    sc, sh, mu, is_ = d(scale), d(shift), d(mean.detach().float()), d(invstd.detach().float())
    # forward
    out = torch.zeros((M, ld), dtype=tdt, device=DEV)
    _lib.check(L.y5m_bn_act(_lib.ptr(yd), ld, _lib.ptr(sc), _lib.ptr(sh), None, 0, _lib.ptr(out), ld, M, C, ACT_SILU, dt,
                            _lib.stream_ptr()), "bn_act")
    assert _relerr(out[:, :C].float().cpu(), z.detach().float()) < (1e-5 if dtype == "f32" else 1e-2)
    assert float(out[:, C:].float().abs().max()) == 0.0 if ld > C else True      # neighbours of the view untouched
    # backward
    dy = torch.zeros((M, C), dtype=tdt, device=DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    wsb = L.y5m_bn_bwd_workspace_bytes(M, C)
    ws = torch.zeros(wsb, dtype=torch.uint8, device=DEV)
    for _ in range(2):
        _lib.check(L.y5m_bn_bwd(_lib.ptr(dzd), ld, _lib.ptr(yd), ld, _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(mu), _lib.ptr(is_),
                                M, C, ACT_SILU, _lib.ptr(dg), _lib.ptr(db), 0, _lib.ptr(dy), C, _lib.ptr(ws), wsb, dt,
                                _lib.stream_ptr()), "bn_bwd")
        torch.cuda.synchronize()
        tol = 2e-4 if dtype == "f32" else 2e-2
        assert _relerr(dy.float().cpu(), yc.grad.float()) < tol
        assert _relerr(dg.cpu(), g64.grad.float()) < 2e-4
        assert _relerr(db.cpu(), b64.grad.float()) < 2e-4
    # the two-launch form (y5m_bn_bwd_fused): the reduce adds into f64 accumulator rows, the apply derives its coefficients
    acc = torch.zeros((L.y5m_bn_acc_slots(), 2, C), dtype=torch.float64, device=DEV)
    for _ in range(2):
        dy.zero_(); dg.zero_(); db.zero_(); acc.zero_()
        _lib.check(L.y5m_bn_bwd_fused(_lib.ptr(dzd), ld, _lib.ptr(yd), ld, _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(mu),
                                      _lib.ptr(is_), M, C, ACT_SILU, _lib.ptr(dg), _lib.ptr(db), 0, _lib.ptr(dy), C,
                                      acc.data_ptr(), dt, _lib.stream_ptr()), "bn_bwd_fused")
        torch.cuda.synchronize()
        assert _relerr(dy.float().cpu(), yc.grad.float()) < (2e-4 if dtype == "f32" else 2e-2)
        assert _relerr(dg.cpu(), g64.grad.float()) < 2e-4
        assert _relerr(db.cpu(), b64.grad.float()) < 2e-4


def test_bn_shared_workspace_across_widths():
    """the engine shares ONE BatchNorm workspace between layers of different widths: wide layer, narrow
    layer, wide layer again through the same buffer must all come out right (the ticket counters may not live
    where another width keeps its stage rows)"""
    from yolov5m_amd import _lib
    L = _lib.lib()
    wsb = max(L.y5m_bn_finalize_workspace_bytes(768), L.y5m_bn_finalize_workspace_bytes(96))
    ws = torch.zeros(wsb, dtype=torch.uint8, device=DEV)
    for it, C in enumerate((768, 48, 192, 96, 768, 48)):
        M, Np = 3000, (C + 95) // 96 * 96
        rows = (M + 127) // 128
        y = _rand((M, C), 81 + it, -2.0, 2.0) * (1 + it)
        yp = torch.cat([y, torch.zeros((rows * 128 - M, C))])
        blk = yp.view(rows, 128, C)
        part = torch.zeros((rows, 2, Np))
        part[:, 0, :C] = blk.sum(1)
        part[:, 1, :C] = (blk * blk).sum(1)
        ones, zeros = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        out = torch.zeros((4, C), device=DEV)
        partd = part.to(DEV)
        _lib.check(L.y5m_bn_finalize(_lib.ptr(partd), rows, Np, C, M, _lib.ptr(ones), _lib.ptr(zeros), _lib.ptr(rm), _lib.ptr(rv),
                                     0.03, 1e-3, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                     1, _lib.ptr(ws), wsb, _lib.stream_ptr()), "bn_finalize")
        torch.cuda.synchronize()
        np.testing.assert_allclose(out[2].cpu().numpy(), y.double().mean(0).float().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(out[3].cpu().numpy(), (1.0 / torch.sqrt(y.double().var(0, unbiased=False) + 1e-3)).float().numpy(),
                                   rtol=1e-4)


def test_conv_multi_equals_separate_launches():
    """y5m_conv_multi (the 4 parity classes of a stride-2 data gradient as ONE launch with interleaved tiles) against the
    same 4 problems launched one by one: bit-identical output (same tiles, same arithmetic). Through ops.conv_dgrad with
    Y5M_CONV_MULTI toggled in a fresh library instance is not possible in-process, so the engine-level entry is driven
    directly: a stride-2 3x3 layer's backward at two sizes."""
    import ctypes
    from yolov5m_amd import _lib, config
    from yolov5m_amd.model import YOLOV5m
    L = _lib.lib()
    torch.manual_seed(0)
    m = YOLOV5m(48, 80, config.ANCHORS, (192, 384, 768)).to(DEV); m.compute_dtype = "bf16"; m.train()
    x = torch.rand((2, 3, 128, 160), device=DEV)
    eng = m._engine_for(x)
    st = _lib.stream_ptr()
    checked = 0
    for lay in eng.layers:
        arr = getattr(lay, "dgrad_multi", None)
        if arr is None:
            continue
        n = len(arr)
        a0 = arr[0]
        dy = torch.randn(a0.B * a0.Hin * a0.Win * a0.ldin, device=DEV).bfloat16()
        outs = []
        for mode in ("multi", "single"):
            out = torch.full((a0.B * a0.Hout * a0.Wout * a0.ldout,), 3.0, device=DEV).bfloat16()
            tmp = (type(a0) * n)()
            for i in range(n):
                ctypes.memmove(ctypes.byref(tmp[i]), ctypes.byref(arr[i]), ctypes.sizeof(a0))
                tmp[i].inp, tmp[i].out, tmp[i].accumulate, tmp[i].res = dy.data_ptr(), out.data_ptr(), 0, None
            if mode == "multi":
                _lib.check(L.y5m_conv_multi(tmp, n, _lib.BF16, st), "multi")
            else:
                for i in range(n):
                    _lib.check(L.y5m_conv(ctypes.byref(tmp[i]), _lib.BF16, st), "single")
            outs.append(out)
        assert torch.equal(outs[0], outs[1]), lay.name
        written = outs[0].float().view(-1, a0.ldout)[:, :a0.N]                # (ldout > N: the gradient of a concat slice)
        assert float((written == 3.0).float().mean()) < 0.01                  # every output pixel was written
        checked += 1
    assert checked >= 5                                                       # backbone 1/3/5/7 + the two neck downsamples


def test_tap48_streaming_kernel():
    """48-channel 3x3 layers on the tapped streaming kernel: stride-1 forward with statistics (48 -> 48), the inference
    epilogue with residual, and the stride-1 data gradient with / without accumulation (default on); the stride-2
    48 -> 96 forward takes it only with Y5M_CONV_PW_TAP48=7 and the tiled kernel otherwise"""
    from yolov5m_amd import ops
    B, H, W = 2, 40, 48
    x = _q(_rand((B, 48, H, W), 51), "bf16")
    w = _q(_rand((48, 48, 3, 3), 52, -0.1, 0.1), "bf16")
    ref = F.conv2d(x, w, None, 1, 1)
    got, s1, s2 = ops.conv_forward_stats(x.to(DEV), w.to(DEV), 1, 1, "bf16")
    assert _relerr(got.cpu(), ref) < TOL["bf16"]
    np.testing.assert_allclose(s1.cpu().numpy(), ref.sum((0, 2, 3)).numpy(), rtol=2e-3, atol=5e-2)
    w2 = _q(_rand((96, 48, 3, 3), 53, -0.1, 0.1), "bf16")
    ref2 = F.conv2d(x, w2, None, 2, 1)
    got2, _, _ = ops.conv_forward_stats(x.to(DEV), w2.to(DEV), 2, 1, "bf16")
    assert _relerr(got2.cpu(), ref2) < TOL["bf16"]
    sc, sh = _rand((48,), 54, 0.5, 1.5), _rand((48,), 55, -0.2, 0.2)
    res = _q(_rand((B, 48, H, W), 56), "bf16")
    ref3 = F.silu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + res
    got3 = ops.conv_forward(x.to(DEV), w.to(DEV), 1, 1, "bf16", scale=sc.to(DEV), shift=sh.to(DEV), act=True, res=res.to(DEV)).cpu()
    assert _relerr(got3, ref3) < TOL["bf16"]
    dy = _q(_rand((B, 48, H, W), 57), "bf16")
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, 1, 1).backward(dy)
    got4 = ops.conv_dgrad(dy.to(DEV), w.to(DEV), (H, W), 1, 1, "bf16").cpu()
    assert _relerr(got4, xr.grad) < TOL["bf16"]
    init = _q(_rand((B, 48, H, W), 58), "bf16")
    got5 = ops.conv_dgrad(dy.to(DEV), w.to(DEV), (H, W), 1, 1, "bf16", init=init.to(DEV)).cpu()
    assert _relerr(got5, xr.grad + init) < TOL["bf16"]
