"""CPU: the train step's ALGORITHMIC HBM bytes are a deterministic function of the plan (Engine.algorithmic_bytes /
NativeTrainStep.algorithmic_bytes: every operand of every launch-list entry moved once). Pinned here at BASELINE.json
configs[2] (B = 64 @ 640x640, bf16): 77.06 GB per step -- the floor the PMC-measured traffic (profiles/r03_pmc_bench.json: ~96 GB)
is held against, and the number a fusion that drops a pass over a tensor must lower. The plan is built at B = 2 on the CPU
executor's library (tests/emu) and rescaled: activation bytes are proportional to the batch, parameter bytes do not depend on it
(checked: B = 1 and B = 2 give the same B = 64 figure)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.skipif(not os.path.exists(os.environ.get("Y5M_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")),
                                reason="host clang of the ROCm image not present (tests/emu compiles the kernel sources for its CPU executor with it)")
sys.path.insert(0, HERE)

_DEFAULT_ENV = not any(k.startswith("Y5M_") and k not in ("Y5M_EMU_THREADS",) for k in os.environ)


def _step(B, monkeypatch, **env):
    # the fused pointwise backward is chosen by pixel count (>= 200 000 at B = 64): scale the threshold with the batch so that the
    # small plan makes the B = 64 choices
    monkeypatch.setenv("Y5M_BWD_PW_MIN_M", str(200000 * B // 64))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from yolov5m_amd import config
    from yolov5m_amd.model import YOLOV5m
    from yolov5m_amd.ultralytics_loss import ComputeLoss
    from yolov5m_amd.utils.training_utils import NativeTrainStep
    m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
    m.compute_dtype = "bf16"
    m.train()
    m.flatten_parameters()
    st = NativeTrainStep(m, ComputeLoss(m), nt_max=8 * B)
    eng = m._engine_for(torch.empty((B, 3, 640, 640), device="meta"))
    return m, st, eng


@pytest.mark.skipif(not _DEFAULT_ENV, reason="a Y5M_* knob is set")
def test_step_bytes_at_config2(monkeypatch):
    from emu.harness import emulated
    with emulated():
        m, st, eng = _step(2, monkeypatch)
        r = st.algorithmic_bytes(eng, B=64)
        assert int(r["total_bytes"]) == 77_056_144_008, r["total_bytes"]
        assert int(r["parameter_bytes"]) == 1_324_155_528
        gb = {k: v / 1e9 for k, v in r["lists"].items()}
        assert abs(gb["forward"] - 26.572) < 1e-3 and abs(gb["backward"] - 49.623) < 1e-3 and abs(gb["optimizer"] - 0.678) < 1e-3
        kinds = {k: (round(b / 1e9, 3), n) for k, (b, n) in r["by_kind"].items()}
        # forward BatchNorm + SiLU: every conv output read and written once more (47.45 M elements per image, + the residual rows)
        assert kinds["apply_fused"] == (12.505, 79)
        assert kinds["conv_igemm"][1] == 134 and kinds["wgrad"][1] == 60 and kinds["bwd_pw"][1] == 13 and kinds["bwd_stem"][1] == 1
        e = eng.algorithmic_bytes()
        assert e["forward"]["launches"] == len(eng.fwd) and e["backward"]["launches"] == len(eng.bwd)
        m._engines = {}
        m1, st1, eng1 = _step(1, monkeypatch)
        assert int(st1.algorithmic_bytes(eng1, B=64)["total_bytes"]) == 77_056_144_008


@pytest.mark.skipif(not _DEFAULT_ENV, reason="a Y5M_* knob is set")
def test_three_launch_batchnorm_backward_costs_what_the_fusion_saves(monkeypatch):
    """Y5M_BWD_PW=0: the 13 fused pointwise-backward launches become BatchNorm backward + data gradient + weight gradient again; the
    counter must see exactly the passes that come back (dy written by the apply pass, read by the data gradient and the weight
    gradient; x read by the weight gradient instead of inside the fused kernel)"""
    from emu.harness import emulated
    with emulated():
        m, st, eng = _step(2, monkeypatch, Y5M_BWD_PW="0")
        r = st.algorithmic_bytes(eng, B=64)
        assert "bwd_pw" not in r["by_kind"]
        assert r["total_bytes"] > 77_056_144_008 + 5e9, r["total_bytes"]


def test_eval_plan_has_a_traffic_figure_on_every_launch():
    from emu.harness import emulated
    from yolov5m_amd import config
    from yolov5m_amd.model import YOLOV5m
    with emulated():
        m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
        m.compute_dtype = "bf16"
        m.eval()
        m.flatten_parameters()
        eng = m._engine_for(torch.empty((1, 3, 64, 64), device="meta"))
        e = eng.algorithmic_bytes()
        assert e["backward"]["launches"] == 0 and e["forward"]["launches"] == len(eng.fwd)
        assert e["forward"]["act_written"] > 0 and e["pack"]["par_written"] > 0


@pytest.mark.skipif(not _DEFAULT_ENV, reason="a Y5M_* knob is set")
def test_inference_forward_is_bandwidth_bound_by_its_own_bytes():
    """BASELINE.json configs[1] (forward only, B = 32 @ 640x640, bf16, inference mode): 7.51 GB of algorithmic HBM traffic = 1.5 ms at
    the 5 TB/s a device copy reaches, against 1.56 TFLOP = 0.63 ms at the dense bf16 MFMA peak: the forward legs of bench.py are
    held against the HBM roofline, not the MFMA one"""
    from emu.harness import emulated
    from yolov5m_amd import config
    from yolov5m_amd.model import YOLOV5m
    with emulated():
        m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
        m.compute_dtype = "bf16"
        m.eval()
        m.flatten_parameters()
        eng = m._engine_for(torch.empty((2, 3, 640, 640), device="meta"))
        f = eng.algorithmic_bytes()["forward"]
        gb = ((f["act_read"] + f["act_written"]) * 16 + f["par_read"] + f["par_written"]) / 1e9
        assert abs(gb - 7.51) < 0.02, gb
        t_hbm, t_mfma = gb / 5e3, 32 * 48.872e9 / 2.5e15
        assert t_hbm > 2 * t_mfma
