"""CPU: the executor's two hazard models (tests/emu/include/emu_rt.h, round 5) catch what they are there for.

Each test builds a DELIBERATELY BROKEN variant of a real kernel source (a textual edit, compiled into build/emu/liby5m_emu_<tag>.so;
the product source is untouched), runs the op-level GPU test of that kernel on it in a child process and expects the executor to
object -- and the unbroken library to pass the same case. TEST INFRASTRUCTURE."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.skipif(not os.path.exists(os.environ.get("Y5M_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")),
                                reason="host clang of the ROCm image not present (tests/emu compiles the kernel sources for its CPU executor with it)")
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)

CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from emu.harness import emulated
import test_gpu_conv as T
T.DEV = "cpu"
with emulated():
    %s
print("CHILD-PASSED")
'''


def _child(body, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    e.setdefault("Y5M_EMU_THREADS", "4")
    return subprocess.run([sys.executable, "-c", CHILD % (ROOT, HERE, body)], env=e, capture_output=True, text=True, timeout=900)


def _variant(tag, edits):
    from emu import build as B
    if not os.path.exists(B.CXX):
        pytest.skip("host clang of the ROCm image not present")
    return B.build_variant(tag, edits)


def test_readfirstlane_of_a_nonuniform_value_is_caught():
    """wgrad_rows_kernel's round-4 form keeps the run state in scalar registers: (row, run) of the first chunk go through
    readfirstlane. A variant that passes a LANE-DEPENDENT value there computes garbage on the GPU (every lane gets lane 0's
    value) but the right answer on a plain per-lane executor -- the uniformity check must abort on it."""
    lib = _variant("rfl", {"y5m_conv_wgrad.hip": lambda s: s.replace(
        "j = __builtin_amdgcn_readfirstlane(j);", "j = __builtin_amdgcn_readfirstlane(j + (lane >> 5)) - (lane >> 5);")})
    body = "T.test_conv_wgrad_rows_kernel((2, 48, 24, 50, 96, 3, 2, 1))"
    bad = _child(body, Y5M_EMU_LIB=lib, Y5M_R4_KERNELS=1)
    assert bad.returncode != 0 and "NOT wave-uniform" in bad.stderr and "y5m_conv_wgrad.hip" in bad.stderr, (bad.returncode, bad.stderr[-800:])
    # the same broken library with the check off: the per-lane arithmetic is "right", the test passes -- which is the blind spot
    blind = _child(body, Y5M_EMU_LIB=lib, Y5M_R4_KERNELS=1, Y5M_EMU_CHECK_UNIFORM=0)
    assert blind.returncode == 0 and "CHILD-PASSED" in blind.stdout, blind.stderr[-800:]
    good = _child(body, Y5M_R4_KERNELS=1)
    assert good.returncode == 0 and "CHILD-PASSED" in good.stdout, good.stderr[-800:]


def test_missing_wait_behind_lds_dma_reads_stale_lds():
    """conv_halo_kernel's prologue stages the first patch and two weight slices with LDS-DMA loads and waits for them with
    `s_waitcnt vmcnt(0)` in front of the barrier. Without that wait the first unit's fragments are read before the data has landed:
    with deferred completion the executor's LDS still holds what was there before, and the forward result is wrong."""
    def drop_wait(s):
        i = s.index("// ---- prologue: first patch, weights of units 0 and 1")
        j = s.index('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', i)
        return s[:j] + s[j + len('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");'):]
    lib = _variant("nowait", {"y5m_conv_halo.hip": drop_wait})
    body = "T.test_halo_forward_stats_and_epilogue((2, 192, 20, 20, 192))"
    bad = _child(body, Y5M_EMU_LIB=lib)
    assert bad.returncode != 0 and "AssertionError" in bad.stderr, (bad.returncode, bad.stderr[-800:])
    # loads that land at once (the executor before round 5) cannot see it
    blind = _child(body, Y5M_EMU_LIB=lib, Y5M_EMU_DEFER_DMA=0)
    assert blind.returncode == 0 and "CHILD-PASSED" in blind.stdout, blind.stderr[-800:]
    good = _child(body)
    assert good.returncode == 0 and "CHILD-PASSED" in good.stdout, good.stderr[-800:]


def test_too_weak_wait_behind_lds_dma_is_caught():
    """the same prologue with `vmcnt(2)`: the two newest loads of every wave may still be in flight at the barrier"""
    def weaken(s):
        i = s.index("// ---- prologue: first patch, weights of units 0 and 1")
        j = s.index('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', i)
        return s[:j] + 'asm volatile("s_waitcnt vmcnt(2)" ::: "memory");' + s[j + len('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");'):]
    lib = _variant("weakwait", {"y5m_conv_halo.hip": weaken})
    bad = _child("T.test_halo_forward_stats_and_epilogue((2, 192, 20, 20, 192))", Y5M_EMU_LIB=lib)
    assert bad.returncode != 0 and "AssertionError" in bad.stderr, (bad.returncode, bad.stderr[-800:])


def test_round4_kernel_forms_green_on_the_executor():
    """the five round-4 kernel rewrites sit behind Y5M_R4_KERNELS (default 0 = the round-3 forms that ran on hardware; csrc/
    y5m_common.h). The regular executor suite runs the default; this runs the op-level cases of those five kernels with all five
    round-4 forms selected (a child: the mask is read once per process), under the uniformity check and deferred LDS-DMA."""
    e = dict(os.environ, Y5M_R4_KERNELS="31")
    r = subprocess.run([sys.executable, os.path.join(HERE, "emu", "run_gpu_tests.py"), "test_gpu_conv",
                        "--only=wgrad_rows|bwd_pw_fused|bwd_stem_fused|bn_act_and_backward"], env=e, capture_output=True, text=True, timeout=1500)
    lines = [l for l in r.stdout.splitlines() if " test_" in l]
    assert r.returncode == 0 and len(lines) >= 40, (r.returncode, r.stderr[-800:])
    bad = [l for l in lines if not l.endswith(" ok")]
    assert not bad, bad[:3]
    # the names the dispatch reports carry the form: a case that silently ran the round-3 form would prove nothing
    c = _child('from yolov5m_amd import ops\n    T.test_conv_wgrad_rows_kernel((1, 48, 11, 13, 96, 3, 2, 1))\n    assert ops.LAST_WGRAD_KERNEL.endswith(",1>"), ops.LAST_WGRAD_KERNEL',
               Y5M_R4_KERNELS=31)
    assert c.returncode == 0 and "CHILD-PASSED" in c.stdout, c.stderr[-800:]
    c = _child('from yolov5m_amd import ops\n    T.test_conv_wgrad_rows_kernel((1, 48, 11, 13, 96, 3, 2, 1))\n    assert ops.LAST_WGRAD_KERNEL.endswith(",0>"), ops.LAST_WGRAD_KERNEL')
    assert c.returncode == 0 and "CHILD-PASSED" in c.stdout, c.stderr[-800:]


def test_bench_untimed_legs_execute_on_the_executor():
    """bench.py's whole main() -- timed loop, then the roofline (an event pair around every launch of profile_step), forward, detect and
    cpu_baseline legs -- at a tiny size on the CPU executor (tests/emu/dry_run_bench.py; stand-ins for HIP events / streams): a Python
    error in a leg the driver only meets at round end shows up here. And a leg that does fail costs its own object, not the line."""
    import json
    r = subprocess.run([sys.executable, os.path.join(HERE, "emu", "dry_run_bench.py"), "1", "--no-smoke"],
                       env=dict(os.environ, Y5M_EMU_THREADS="4"), capture_output=True, text=True, timeout=1500)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stderr[-1500:])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "forward", "detect", "kernel_forms"):
        assert k in d, k
    for leg in ("roofline", "forward", "detect", "cpu_baseline"):
        assert "error" not in d[leg], (leg, d[leg])
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["cpu_baseline"]["kind"] == "port"
    assert "BASELINE.json" not in d["config"]["workload"].split("(")[-1] or "NOT a BASELINE" in d["config"]["workload"]
    sys.path.insert(0, ROOT)
    import bench
    g = bench._guarded("broken", lambda: 1 / 0)
    assert g == {"error": "broken leg failed: ZeroDivisionError: division by zero"}


def test_pool_tiled_forms_bit_identical_on_the_executor(tmp_path):
    """Y5M_POOL_TILE=1 (default 0, csrc/y5m_nn.hip: one image x a slab of channels per workgroup in LDS; the backward cascade of the
    three SPPF pools as ONE launch): the f32 forms against ATen bit-exactly and the bf16 forms bit-identical to the separable
    kernels -- tests/test_gpu_model.py::test_sppf_pool_tiled_forms_subprocess does this on the GPU, this is its executor twin."""
    import numpy as np
    outs = {}
    for tag, env in (("sep", {}), ("tiled", {"Y5M_POOL_TILE": "1"})):
        body = ("import test_gpu_model as M\n    M.DEV = 'cpu'\n    from yolov5m_amd import _lib\n"
                f"    assert _lib.lib().y5m_sppf_pool_tiled(20, 20, 384, _lib.BF16) == {int(tag == 'tiled')}\n"
                f"    assert _lib.lib().y5m_sppf_pool_tiled(40, 40, 384, _lib.BF16) == {int(tag == 'tiled')}\n"
                "    for s in ((4, 96, 10, 13), (2, 384, 20, 20), (1, 64, 40, 40)):\n        M.test_sppf_pool_forward_backward_bit_exact(s)\n"
                "    for i, s in enumerate(((2, 384, 20, 20), (1, 64, 40, 40), (3, 40, 7, 9))):\n"
                f"        a, b = M._pool_bf16_run(*s); np.savez({str(tmp_path / tag)!r} + str(i), a=a, b=b)")
        c = _child(body, **env)
        assert c.returncode == 0 and "CHILD-PASSED" in c.stdout, c.stderr[-1500:]
        outs[tag] = [np.load(str(tmp_path / f"{tag}{i}.npz")) for i in range(3)]
    for s, t in zip(outs["sep"], outs["tiled"]):
        assert np.array_equal(s["a"], t["a"]) and np.array_equal(s["b"], t["b"])
        assert np.abs(s["b"]).max() > 0


def test_halo_two_stage_ring_green_on_the_executor():
    """Y5M_CONV_HALO_NS2=1 (default 0; csrc/y5m_conv_halo.hip): images 45..88 pixels wide -- the 80x80 stage of a 1280x1280 model --
    on the halo kernel with a TWO-stage weight ring (three stages + two patch buffers exceed the LDS, so these shapes ran tiled).
    The executor twin of tests/test_gpu_conv.py::test_halo_two_stage_ring_wide_images_subprocess: three "CUs", so every workgroup walks
    several tiles and the ring's parity is carried across slabs and tiles; waves visited in random order."""
    body = ("import test_dispatch_cpu as D\n    from yolov5m_amd._lib import EPI_RAW_STATS, EPI_DGRAD\n"
            "    assert D._conv_name(128, 192, 80, 80, 192, 3, 1, EPI_RAW_STATS) == 'conv_halo_kernel<6,0,ns2>'\n"
            "    assert D._conv_name(128, 192, 80, 80, 192, 3, 1, EPI_DGRAD) == 'conv_halo_kernel<6,3,ns2>'\n"
            "    assert D._conv_name(64, 192, 40, 40, 192, 3, 1, EPI_RAW_STATS) == 'conv_halo_kernel<6,0>'\n"
            "    for case in T.HALO_WIDE_CASES:\n        T.test_halo_wide_forward_and_dgrad(case)")
    c = _child(body, Y5M_CONV_HALO_NS2=1, Y5M_EMU_CUS=3, Y5M_EMU_WAVE_ORDER=7)
    assert c.returncode == 0 and "CHILD-PASSED" in c.stdout, c.stderr[-1500:]
    off = _child("import test_dispatch_cpu as D\n    from yolov5m_amd._lib import EPI_RAW_STATS\n"
                 "    assert D._conv_name(128, 192, 80, 80, 192, 3, 1, EPI_RAW_STATS).startswith('conv_igemm_kernel')")
    assert off.returncode == 0 and "CHILD-PASSED" in off.stdout, off.stderr[-800:]


def test_head_pack16_bit_identical_on_the_executor(tmp_path):
    """Y5M_HEAD_PACK16=1 (default 0): the sparse head-gradient pack's objectness rows, two rows per store instruction in 16-byte pieces.
    The three scales' packed rows (the dy slots) and bias gradients of a train step's head backward (engine, 2 x 64 x 96: 96 / 24 / 6
    pixels per image and scale -- partial waves, odd row counts) are bit-identical to the 8-byte kernel's, and the sparse-vs-dense op
    test holds."""
    import numpy as np
    body = ("import test_gpu_detect_loss as D\n    D.DEV = 'cpu'\n"
            "    from yolov5m_amd import config\n    anchors = torch.tensor(config.ANCHORS).float().view(3, -1, 2) / torch.tensor([8., 16., 32.]).view(3, 1, 1)\n"
            "    for dt in ('bf16', 'f32'):\n        D.test_sparse_head_gradient_path_equals_dense(anchors, dt)\n"
            "    from yolov5m_amd.model import YOLOV5m\n    from yolov5m_amd.ultralytics_loss import ComputeLoss\n"
            "    from yolov5m_amd.utils.synth import synth_images, synth_labels, synth_state_dict\n    from yolov5m_amd.utils.training_utils import NativeTrainStep\n"
            "    m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768)); m.load_state_dict(synth_state_dict(), strict=True)\n"
            "    m.compute_dtype = 'bf16'; m.train(); m.flatten_parameters()\n"
            "    st = NativeTrainStep(m, ComputeLoss(m), nt_max=64)\n"
            "    eng = st.load_inputs(synth_images(2, 64, 96, seed='hp/i'), synth_labels(2, 5, seed='hp/l'))\n"
            "    kinds = [getattr(op[0], 'kind', None) for op in eng.bwd]\n"
            "    k = max(i for i, kd in enumerate(kinds) if kd == 'head_pack') + 1\n"
            "    assert kinds.count('head_pack') == 3\n"
            "    st._enqueue_fb(eng, bwd_upto=k)                  # forward, loss, the backward list up to the third head pack\n"
            "    np.savez(OUT, g=m.flat_grads.numpy(), **{f's{i}': t.float().numpy() for i, t in enumerate(eng.scratch2)})")
    outs = []
    for tag, env in (("p8", {}), ("p16", {"Y5M_HEAD_PACK16": "1"})):
        f = str(tmp_path / (tag + ".npz"))
        c = _child(body.replace("OUT", repr(f)), Y5M_EMU_THREADS=1, **env)          # one OS thread: the f32 atomics of the step in one order
        assert c.returncode == 0 and "CHILD-PASSED" in c.stdout, c.stderr[-1500:]
        outs.append(dict(np.load(f)))
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert all(np.abs(outs[0][f"s{i}"]).max() > 0 for i in range(3)) and np.abs(outs[0]["g"]).max() > 0
