"""Which kernel instantiation the library dispatches each YOLOv5m layer family to (y5m_conv_kernel_name /
y5m_wgrad_kernel_name run the dispatch and launch nothing, so this needs no GPU). Pins the defaults the measurements in
DESIGN.md sections 4-5 refer to (B=64 @ 640^2 shapes, bf16), and the knob-selected weight-gradient forms in a child process
(every knob is read once per process). Layer shapes: reference model.py:12-28 (CBL), :60-118 (C3 / Bottleneck)."""
import ctypes
import os
import subprocess
import sys

import pytest

from yolov5m_amd import _lib
from yolov5m_amd._lib import BF16, EPI_DGRAD, EPI_RAW_STATS, ConvArgs, WgradArgs

_rup = lambda x, m: (x + m - 1) // m * m
_PTR = 16          # any non-null value: nothing is dereferenced in name-only mode


def _conv_name(B, Cin, H, W, Cout, k, s, epi):
    L = _lib.lib()
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    a = ConvArgs()
    a.zeros = a.inp = a.w = a.out = a.stats = _PTR
    a.B, a.Hin, a.Win, a.ldin, a.Hg, a.Wg, a.sy, a.sx = B, H, W, Cin, Ho, Wo, s, s
    a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = k, k, -p, 1, -p, 1
    K = k * k * Cin
    a.Cin, a.K, a.N, a.M = Cin, K, Cout, B * Ho * Wo
    a.Kp = _rup(K, 64) + (64 * ((Cin + 63) // 64) if k == 3 else 0)      # room for the halo kernel's last weight unit
    a.Hout, a.Wout, a.ldout, a.osy, a.osx = Ho, Wo, Cout, 1, 1
    a.Np = _rup(Cout, 192)
    a.epi = epi
    buf = ctypes.create_string_buffer(192)
    _lib.check(L.y5m_conv_kernel_name(ctypes.byref(a), BF16, buf, 192), "y5m_conv_kernel_name")
    return buf.value.decode()


def _wgrad_name(B, Cin, H, W, Cout, k, s):
    L = _lib.lib()
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    a = WgradArgs()
    a.zeros = a.dy = a.x = a.dwgt = _PTR
    a.B, a.Hin, a.Win, a.ldx, a.Hg, a.Wg, a.sy, a.sx = B, H, W, Cin, Ho, Wo, s, s
    a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = k, k, -p, 1, -p, 1
    a.C, a.N, a.M, a.lddy, a.lddw, a.ksplit = Cin, Cout, B * Ho * Wo, Cout, k * k * Cin, 0
    buf = ctypes.create_string_buffer(192)
    _lib.check(L.y5m_wgrad_kernel_name(ctypes.byref(a), BF16, buf, 192), "y5m_wgrad_kernel_name")
    return buf.value.decode()


_DEFAULT_ENV = not any(k.startswith(("Y5M_CONV", "Y5M_WGRAD", "Y5M_R4")) for k in os.environ)

CONV_DEFAULTS = [
    # B, Cin, H, W, Cout, k, s -> forward (raw + statistics), data gradient
    ((64, 192, 40, 40, 192, 3, 1), "conv_halo_kernel<6,0>", "conv_halo_kernel<6,3>"),
    ((64, 384, 20, 20, 384, 3, 1), "conv_halo_kernel<6,0>", "conv_halo_kernel<6,3>"),
    ((64, 96, 80, 80, 96, 3, 1), "conv_igemm_kernel<bf16,2,2,4,3,0>", "conv_igemm_kernel<bf16,2,2,4,3,0>"),
    ((64, 96, 160, 160, 192, 3, 2), "conv_igemm_kernel<bf16,2,4,4,3,0>", None),
    ((64, 96, 80, 80, 96, 1, 1), "conv_pw_kernel<6,3,0,0,0>", "conv_pw_kernel<6,3,3,0,0>"),
    ((64, 48, 160, 160, 48, 3, 1), "conv_pw_kernel<3,14,0,0,48>", "conv_pw_kernel<3,14,3,0,48>"),
    ((64, 384, 40, 40, 384, 1, 1), "conv_gemm8_kernel<0>", "conv_igemm_kernel<bf16,2,4,4,3,0>"),
    ((64, 768, 20, 20, 768, 1, 1), "conv_gemm8_kernel<0>", "conv_igemm_kernel<bf16,2,4,4,3,0>"),
]

WGRAD_DEFAULTS = [
    ((64, 192, 40, 40, 192, 3, 1), "wgrad_kernel<bf16,2,2,1,3,1,6,1>"),           # the 192 x 96 block tile (wave 96 x 48)
    ((64, 384, 40, 40, 768, 3, 2), "wgrad_kernel<bf16,2,2,1,3,1,6,1>"),
    ((64, 96, 80, 80, 192, 3, 2), "wgrad_kernel<bf16,2,2,1,3,1,6,1>"),            # 192 x 96 block tile
    ((64, 96, 80, 80, 96, 3, 1), "wgrad_kernel<bf16,2,2,1,3,1,3,1>"),
    ((64, 48, 320, 320, 96, 3, 2), "wgrad_rows_kernel<2,2,2,0>"),                   # 48 input channels: one kernel row per block
    ((64, 48, 160, 160, 48, 3, 1), "wgrad_rows_kernel<1,4,1,0>"),
    ((64, 16, 320, 320, 48, 3, 1), "wgrad_kernel<bf16,1,1,4,9,9,3,0>"),           # stem: all nine taps in one block
    ((64, 384, 20, 20, 384, 1, 1), "wgrad_kernel<bf16,2,2,1,3,1,6,1>"),           
]


@pytest.mark.skipif(not _DEFAULT_ENV, reason="a Y5M_CONV* / Y5M_WGRAD* knob is set")
@pytest.mark.parametrize("case,fwd,dgrad", CONV_DEFAULTS)
def test_conv_default_dispatch(case, fwd, dgrad):
    assert _conv_name(*case, EPI_RAW_STATS) == fwd
    if dgrad is not None:
        assert _conv_name(*case, EPI_DGRAD) == dgrad


@pytest.mark.skipif(not _DEFAULT_ENV, reason="a Y5M_CONV* / Y5M_WGRAD* knob is set")
@pytest.mark.parametrize("case,want", WGRAD_DEFAULTS)
def test_wgrad_default_dispatch(case, want):
    assert _wgrad_name(*case) == want


@pytest.mark.skipif(not _DEFAULT_ENV, reason="a Y5M_CONV* / Y5M_WGRAD* knob is set")
def test_configs4_80x80_stage_dispatch_is_the_documented_gap():
    """BASELINE.json configs[4] (B = 128 @ 1280x1280 inference): the ten 192 -> 192 3x3 layers of the 80x80 stage do not fit the halo
    kernel's LDS image with a three-stage weight ring (2 x (258 + 2 W) x 128 B + 3 x 24 KB = 182 KB at W = 80) and run on the tiled
    kernel by default; Y5M_CONV_HALO_NS2=1 (round 5, unmeasured) gives them the two-stage ring. The 40x40 stage (384 channels) and
    every halo-eligible layer of the 640x640 plans fit the three-stage form. (tools/plan_dispatch.py lists whole plans.)"""
    EPI_AFFINE_ACT = 1
    assert _conv_name(128, 192, 80, 80, 192, 3, 1, EPI_AFFINE_ACT) == "conv_igemm_kernel<bf16,2,4,4,3,0>"
    assert _conv_name(128, 384, 40, 40, 384, 3, 1, EPI_AFFINE_ACT) == "conv_halo_kernel<6,1>"
    for S in range(320, 704, 32):                       # every multi_scale size of the training loop (training_utils.py:11-28)
        for C, div in ((192, 16), (384, 32)):
            assert _conv_name(64, C, S // div, S // div, C, 3, 1, EPI_RAW_STATS) == "conv_halo_kernel<6,0>", (S, C)
    child = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
             "import test_dispatch_cpu as D\n"
             "assert D._conv_name(128, 192, 80, 80, 192, 3, 1, 1) == 'conv_halo_kernel<6,1,ns2>'\n"
             "assert D._conv_name(64, 192, 40, 40, 192, 3, 1, 0) == 'conv_halo_kernel<6,0>'\n"
             "assert D._conv_name(2, 192, 12, 96, 192, 3, 1, 0).startswith('conv_igemm_kernel')\n"      # 96 wide: no ring fits
             ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, Y5M_CONV_HALO_NS2="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]


def _wgrad_geometry(B, Cin, H, W, Cout, k, s):
    L = _lib.lib()
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    a = WgradArgs()
    a.zeros = a.dy = a.x = a.dwgt = _PTR
    a.B, a.Hin, a.Win, a.ldx, a.Hg, a.Wg, a.sy, a.sx = B, H, W, Cin, Ho, Wo, s, s
    a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = k, k, -p, 1, -p, 1
    a.C, a.N, a.M, a.lddy, a.lddw, a.ksplit = Cin, Cout, B * Ho * Wo, Cout, k * k * Cin, 0
    g = (ctypes.c_int32 * 8)()
    _lib.check(L.y5m_wgrad_geometry(ctypes.byref(a), BF16, g), "y5m_wgrad_geometry")
    return list(g)


@pytest.mark.skipif(not _DEFAULT_ENV, reason="a Y5M_CONV* / Y5M_WGRAD* knob is set: the defaults are what is pinned here")
def test_weight_gradient_launch_geometry_and_the_xcd_deal_it_implies():
    """y5m_wgrad_geometry (round 6; tools/wgrad_traffic.py reads it): {n tiles, c tiles, tap groups, ranges, workgroups, dY channels per
    block, X channels per block and tap, pixels per chunk}. Pins the dominant class of the step -- 192 -> 192 3x3 @ 40x40, B = 64: 18
    blocks per pixel range, 14 ranges, 252 blocks = 31.5 per XCD, so the kernel's 8 contiguous pieces cut inside 7 of the 14 ranges
    (profiles/r06_wgrad_traffic.txt: 1.38x of the algorithmic bytes get past the XCDs' L2s; NOTES.md round 6) -- and the
    kernel-row form of the 48-channel layers."""
    g = _wgrad_geometry(64, 192, 40, 40, 192, 3, 1)
    assert g == [1, 2, 9, 14, 252, 192, 96, 64], g
    G = g[0] * g[1] * g[2]
    q8, r8 = g[4] >> 3, g[4] & 7                              # the kernel's deal: XCD x takes q8 (+ 1 if x < r8) consecutive logical blocks
    starts = [x * (q8 + 1) if x < r8 else r8 * (q8 + 1) + (x - r8) * q8 for x in range(1, 8)]
    assert (G, sum(1 for b in starts if b % G != 0)) == (18, 7), starts
    assert _wgrad_geometry(64, 48, 160, 160, 48, 3, 1)[:5] == [1, 1, 3, 170, 510]       # wgrad_rows_kernel: one kernel row per block
    g1 = _wgrad_geometry(64, 192, 40, 40, 192, 1, 1)                                    # pointwise: ~224 blocks, 2 per range
    assert g1[:3] == [1, 2, 1] and g1[3] * 2 == g1[4] == 224
