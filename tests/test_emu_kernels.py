"""CPU: the HIP kernel SOURCES of yolov5m_amd/csrc executed lane by lane on this machine (tests/emu: every GPU thread a
fiber, wavefront collectives -- MFMA, ds_read_b64_tr_b16, DPP, readlane, ballot -- as rendezvous of 64 fibers) and checked by
the SAME test functions the -m gpu suite runs on the MI355X: tests/test_gpu_conv.py, test_gpu_detect_loss.py,
test_gpu_glue.py and the small cases of test_gpu_model.py are imported, pointed at host tensors and called case by case.

What this proves on a CPU-only container: the kernels' index arithmetic, tile / tap / halo logic, epilogues, statistics,
atomics and the engine's launch lists produce the oracle's numbers at HEAD. What it cannot see: timing, s_waitcnt / cache
coherence / LDS-bank behaviour, stream and graph ordering -- the -m gpu suite stays the parity gate (tests/emu/include/emu_rt.h).
The library under test here is build/emu/liby5m_emu.so, test infrastructure like oracle/; the product has no CPU path
(tests/test_abi.py asserts that)."""
import importlib
import inspect
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.skipif(not os.path.exists(os.environ.get("Y5M_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")),
                                reason="host clang of the ROCm image not present (tests/emu compiles the kernel sources for its CPU executor with it)")
sys.path.insert(0, HERE)
from emu.run_gpu_tests import expand  # noqa: E402


def _elems(case):
    """rough size of a conv-like case tuple (B, C, H, W, ...): used to keep the slow ones out of the CPU suite"""
    try:
        B, C, H, W = case[:4]
        return B * C * H * W * (case[4] if len(case) > 4 and isinstance(case[4], int) else 1)
    except Exception:
        return 0


# test function -> predicate on its keyword arguments: True = run on the emulator. Functions not listed run in full.
_MAX = 3.5e8
SELECT = {
    "test_gpu_conv": {
        # subprocess tests start a fresh interpreter on the real device; conv_multi builds a model on "cuda"
        "test_gemm8_small_shapes_subprocess": None, "test_bn_unfused_engine_subprocess": None,
        "test_halo_96_channel_tile_subprocess": None, "test_halo_two_stage_ring_wide_images_subprocess": None,
        "test_gemm8_forward_epilogues": lambda kw: _elems(kw["case"]) < _MAX,
        "test_gemm8_bn_accumulator_rows": lambda kw: _elems(kw["case"]) < _MAX,
        "test_gemm8_dgrad": lambda kw: _elems(kw["case"]) < _MAX,
        "test_halo_forward_stats_and_epilogue": lambda kw: _elems(kw["case"]) < _MAX,
        "test_conv_bn_accumulator_rows": lambda kw: _elems(kw["case"]) < _MAX,
        "test_bn_act_and_backward": lambda kw: kw["case"][0] <= 100000,
    },
    "test_gpu_detect_loss": {
        "test_nms_b128_config4_shape_bit_exact": None,        # 128 images x 25 200 boxes: minutes of fibers
        "test_sparse_head_gradient_pack16_subprocess": None,  # (a child interpreter on the real device; executor twin: test_emu_checks)
        "test_decode_vs_oracle_640": None,
        "test_nms_large_random": lambda kw: kw["N"] <= 25200,
    },
    "test_gpu_glue": {},
    "test_gpu_model": {
        "test_forward_f32_golden": lambda kw: (kw["tag"], kw["mode"]) in (("s64", "eval"), ("s96x128", "train")),
        "test_train_step_grads_f32_golden": lambda kw: kw["variant"] == "default",
        "test_sppf_pool_forward_backward_bit_exact": lambda kw: True,
        "test_submodule_forward_matches_torch": lambda kw: True,
        "test_submodule_backward_matches_torch": lambda kw: True,
        "test_eval_merged_c3_pair_equals_two_convs": lambda kw: kw["dtype"] == "bf16",
        # the fused step on the reference's default loss: the captured form (the anchor state advances inside the recording)
        # (sparse native gradient against the dense autograd one is part of it; the target-format twin is GPU-only: its host logic
        #  is tests/test_model_cpu.py::test_yolo_loss_target_staging_ragged_empty_and_grouped_forms)
        "test_native_train_step_yolo_loss_matches_autograd": None,      # (ran green here as a one-off, profiles/r06_emu_extra.txt; the suite
        #  holds the fused YOLO_LOSS step against the REAL reference instead: g16 below, and g17 = the reference's train_loop epoch)
        "test_train_loop_epoch_reference_golden": lambda kw: (kw["loss_kind"], kw["optimizer"]) == ("yolo", "fused"),
        "test_native_train_step_yolo_loss_vs_oracle": None,      # (GPU-only: the oracle is pinned by g16, the native step by the next line)
        "test_native_yolo_steps_reference_golden": lambda kw: True,
        "test_detect_driver_is_the_reference_detect_flow": lambda kw: True,
        # (test_native_train_step_matches_torch_adam / test_input_stage_u8_golden bound rounding-level noise -- 8 near-cancelling
        #  Adam elements of 21 M, 2e-6 absolute -- that the host's un-contracted arithmetic moves: 12 elements / 2.1e-6 here)
        "*": None,                                           # everything else of this module: full-size / graphs / subprocesses
    },
}


def _collect():
    items = []
    for modname, rules in SELECT.items():
        mod = importlib.import_module(modname)
        for name in sorted(n for n in dir(mod) if n.startswith("test_")):
            fn = getattr(mod, name)
            if not callable(fn):
                continue
            rule = rules.get(name, rules.get("*", True))
            if rule is None:
                continue
            for i, kw in enumerate(expand(fn)):
                if rule is True or rule(kw):
                    items.append(pytest.param(modname, name, kw, id=f"{modname[9:]}::{name[5:]}[{i}]"))
    return items


@pytest.fixture(scope="module")
def emu():
    from emu.harness import emulated
    with emulated() as L:
        yield L


@pytest.mark.parametrize("modname,name,kw", _collect())
def test_kernel_sources_on_cpu_executor(emu, golden, monkeypatch, modname, name, kw):
    mod = importlib.import_module(modname)
    monkeypatch.setattr(mod, "DEV", "cpu", raising=False)
    fn = getattr(mod, name)
    kw = dict(kw)
    for pn in inspect.signature(fn).parameters:
        if pn == "golden":
            kw[pn] = golden
        elif pn == "monkeypatch":
            kw[pn] = monkeypatch
        elif pn not in kw:
            fx = getattr(mod, pn, None)
            if fx is not None and hasattr(fx, "_get_wrapped_function"):
                kw[pn] = fx._get_wrapped_function()()
    fn(**kw)


def test_emulator_detects_divergent_collectives_and_deadlocks():
    """the executor must fail loudly, not hang or pass, when a wave-wide operation sits in divergent control flow (a child
    process: the failure is an abort)"""
    import subprocess
    import textwrap
    src = textwrap.dedent('''
        #include <hip/hip_runtime.h>
        __global__ void bad(int* out) { if (threadIdx.x < 32) out[threadIdx.x] = __shfl(1, 0, 64); __syncthreads(); }
        int main() { int out[64]; hipLaunchKernelGGL(bad, dim3(1), dim3(64), 0, 0, out); return 0; }
    ''')
    import tempfile
    from emu import build as B
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "bad.cpp")
        with open(p, "w") as f:
            f.write(src)
        exe = os.path.join(d, "bad")
        subprocess.check_call([B.CXX] + B.FLAGS + [p, os.path.join(B.HERE, "emu_rt.cpp"), "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "deadlock" in r.stderr, (r.returncode, r.stderr[-500:])
