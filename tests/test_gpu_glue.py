"""GPU parity, op level, for the one-launch glue kernels around the convolutions (space-to-depth input, nearest
upsample and its backward, gradient add, the head's output-gradient packing, gradient norm + clip + Adam), each through
the C ABI against the torch CPU op the reference uses at that place. Data movement is compared bit for bit; the
optimizer within f32 rounding of torch.optim.Adam (reference train.py:61, :118)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dt(dtype):
    from yolov5m_amd._lib import F32, BF16
    return (torch.float32, F32) if dtype == "f32" else (torch.bfloat16, BF16)


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(1, 64, 64), (3, 96, 160), (2, 322, 34)])
def test_s2d_input_is_the_2x2_space_to_depth_of_the_nchw_image(shape, dtype):
    """reference model.py:210: the (B,3,H,W) image feeds a 6x6 / stride-2 conv; here it becomes the (B,H/2,W/2,16) NHWC
    tensor the 3x3 stem GEMM reads, channel = (dy*2+dx)*3 + c, channels 12..15 zero"""
    from yolov5m_amd import _lib
    B, H, W = shape
    tdt, dt = _dt(dtype)
    img = _rand((B, 3, H, W), 3)
    out = torch.full((B, H // 2, W // 2, 16), 7.0, dtype=tdt, device=DEV)
    imgd = img.to(DEV)
    _lib.check(_lib.lib().y5m_s2d_input(_lib.ptr(imgd), B, H, W, _lib.ptr(out), dt, _lib.stream_ptr()), "s2d")
    ref = torch.zeros((B, H // 2, W // 2, 16))
    for dy in range(2):
        for dx in range(2):
            for c in range(3):
                ref[..., (dy * 2 + dx) * 3 + c] = img[:, c, dy::2, dx::2]
    assert torch.equal(out.float().cpu(), ref.to(tdt).float())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 5, 7, 96, 96, 192), (1, 20, 20, 384, 384, 768), (3, 1, 1, 8, 16, 8)])
def test_upsample2x_and_backward_strided_views(case, dtype):
    """nn.Upsample(scale_factor=2, nearest) (reference model.py:225) written straight into a channel slice of the concat
    buffer, and its backward (sum of the 4 children), plain and accumulating"""
    from yolov5m_amd import _lib
    B, H, W, C, ldin, ldout = case
    tdt, dt = _dt(dtype)
    L = _lib.lib()
    x = _rand((B, H, W, ldin), 5).to(tdt)
    out = torch.full((B, 2 * H, 2 * W, ldout), 3.0, dtype=tdt, device=DEV)
    xd = x.to(DEV)
    _lib.check(L.y5m_upsample2x(_lib.ptr(xd), ldin, B, H, W, C, _lib.ptr(out), ldout, dt, _lib.stream_ptr()), "up")
    ref = F.interpolate(x[..., :C].float().permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
    got = out.float().cpu()
    assert torch.equal(got[..., :C], ref)
    if ldout > C:
        assert float((got[..., C:] - 3.0).abs().max()) == 0.0          # the neighbouring slice is untouched
    # backward: gin[b,y,x,c] (+)= sum of gout over the 2x2 children
    g = _rand((B, 2 * H, 2 * W, ldout), 6).to(tdt)
    gd = g.to(DEV)
    gin0 = _rand((B, H, W, ldin), 7).to(tdt)
    gf = g[..., :C].float()
    s4 = gf[:, 0::2, 0::2] + gf[:, 0::2, 1::2] + gf[:, 1::2, 0::2] + gf[:, 1::2, 1::2]
    for acc in (0, 1):
        gin = gin0.clone().to(DEV)
        _lib.check(L.y5m_upsample2x_bwd(_lib.ptr(gd), ldout, B, H, W, C, _lib.ptr(gin), ldin, acc, dt, _lib.stream_ptr()), "upb")
        want = s4 + (gin0[..., :C].float() if acc else 0.0)
        got = gin.float().cpu()
        tol = 0.0 if dtype == "f32" and not acc else (1e-6 if dtype == "f32" else 2e-2)
        assert float((got[..., :C] - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
        if ldin > C:
            assert torch.equal(got[..., C:], gin0[..., C:].float())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_add_views(dtype):
    """dst (+)= src on (ptr, ld) views: the residual / concat gradient plumbing (reference model.py:50, :91 backward)"""
    from yolov5m_amd import _lib
    tdt, dt = _dt(dtype)
    M, C, lds, ldd = 1237, 96, 192, 96
    src, dst0 = _rand((M, lds), 9).to(tdt), _rand((M, ldd), 10).to(tdt)
    sd = src.to(DEV)
    for acc in (0, 1):
        dst = dst0.clone().to(DEV)
        _lib.check(_lib.lib().y5m_add(_lib.ptr(sd), lds, _lib.ptr(dst), ldd, M, C, acc, dt, _lib.stream_ptr()), "add")
        want = (src[:, :C].float() + (dst0[:, :C].float() if acc else 0.0)).to(tdt).float()
        assert torch.equal(dst.float().cpu()[:, :C], want)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 8, 8, 85, 256), (3, 5, 7, 85, 256), (1, 20, 20, 6, 32)])
def test_head_grad_pack_dense(case, dtype):
    """d(loss)/d(logits) arrives as (B,3,ny,nx,5+nc) f32 -- the permuted view of reference model.py:173 -- and the head
    conv's gradient kernels want [pixel][3*(5+nc)] rows in the compute dtype plus the bias gradient (column sums)"""
    from yolov5m_amd import _lib
    B, ny, nx, nch, ldp = case
    tdt, dt = _dt(dtype)
    dl = _rand((B, 3, ny, nx, nch), 11)
    dld = dl.to(DEV)
    dyp = torch.full((B * ny * nx, ldp), 5.0, dtype=tdt, device=DEV)
    db = torch.full((3 * nch,), 9.0, device=DEV)
    _lib.check(_lib.lib().y5m_head_grad_pack(_lib.ptr(dld), B, 3, ny, nx, nch, _lib.ptr(dyp), ldp, _lib.ptr(db), dt,
                                             _lib.stream_ptr()), "pack")
    ref = dl.permute(0, 2, 3, 1, 4).reshape(B * ny * nx, 3 * nch)
    got = dyp.float().cpu()
    assert torch.equal(got[:, :3 * nch], ref.to(tdt).float())
    assert float(got[:, 3 * nch:].abs().max()) == 0.0                   # the K padding of the gradient GEMMs is zero
    np.testing.assert_allclose(db.cpu().numpy(), ref.double().sum(0).float().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("n", [7, 4096, 1_000_003])
@pytest.mark.parametrize("scale", [0.01, 30.0])
def test_grad_norm_clip_adam_matches_torch(n, scale):
    """clip_grad_norm_(max_norm=10) + Adam(lr 5e-4, weight_decay 5e-4) (reference train.py:61, utils/training_utils.py:118)
    over one flat buffer, three steps with the device-side step counter; scale 30 makes the clip engage, 0.01 not.
    (n = 7 and 1 000 003: the scalar tail of the 4-wide kernel)"""
    from yolov5m_amd import _lib
    L = _lib.lib()
    p0 = _rand((n,), 21)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_ref], lr=5e-4, weight_decay=5e-4)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    ws = torch.zeros(L.y5m_adam_workspace_bytes(), dtype=torch.uint8, device=DEV)
    norm = torch.zeros(1, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    for it in range(3):
        g = _rand((n,), 30 + it) * scale
        p_ref.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([p_ref], 10.0)
        opt.step()
        gd = g.to(DEV)
        step += 1
        _lib.check(L.y5m_grad_norm(_lib.ptr(gd), n, _lib.ptr(norm), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "norm")
        _lib.check(L.y5m_adam_step(_lib.ptr(p), _lib.ptr(gd), _lib.ptr(m), _lib.ptr(v), n, _lib.ptr(norm), 10.0, 5e-4, 0.9, 0.999,
                                   1e-8, 5e-4, _lib.ptr(step), _lib.stream_ptr()), "adam")
        torch.cuda.synchronize()
        np.testing.assert_allclose(float(norm[0]), float(tn), rtol=1e-5)
        # an update is lr * mhat / (sqrt(vhat) + eps) = O(lr): compare the parameters to a fraction of one update. The few
        # elements whose clipped gradient cancels against weight_decay * p are ill-conditioned in mhat / sqrt(vhat) (a
        # 1e-6 relative difference of the norm moves them by percents of an update -- |g*clip + wd*p| ~ eps = 1e-8 puts the ratio anywhere in (-1, 1)): counted, and bounded by the
        # size of an update
        d = (p.cpu() - p_ref.detach()).abs()
        assert int((d > 2e-3 * 5e-4 + 1e-7).sum()) <= max(1, n // 100000), it     # (measured: 1 element of 1 000 003, 18 % of lr)
        assert float(d.max()) <= 2 * 5e-4, it
    st = opt.state[p_ref]
    # (when the clip engages, the f32 norm torch computes over n elements and the f64-accumulated one here differ by ~1e-6
    #  relative, and so do all clipped gradients)
    tight = scale < 1.0
    np.testing.assert_allclose(m.cpu().numpy(), st["exp_avg"].numpy(), rtol=1e-5 if tight else 1e-4,
                               atol=(1e-6 if tight else 2e-5) * float(st["exp_avg"].abs().max()))
    np.testing.assert_allclose(v.cpu().numpy(), st["exp_avg_sq"].numpy(), rtol=1e-5 if tight else 1e-4,
                               atol=(1e-7 if tight else 2e-5) * float(st["exp_avg_sq"].abs().max()))
