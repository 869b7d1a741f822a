"""CPU: the engine's launch lists and the data-parallel step, executed on the CPU lane-level executor (tests/emu) -- the
product's own engine.py / training_utils.py / parallel.py drive the product's own kernel sources on host tensors.

  * plan invariants of the overlapped gradient exchange (Engine.bwd_marks / grad_cuts / _check_grad_writers), with and
    without the weight-gradient reorder;
  * the property the exchange relies on, by execution: once the backward list has run up to a cut, no later launch writes the
    part of the flat gradient buffer that cut declares final (bit-exact before / after);
  * two gloo ranks: NativeTrainStep's overlapped schedule (segments + bucket launches + wait) lands on the all-reduce of the
    ranks' plain gradients, parameters stay bit-identical across ranks.
What this cannot cover: streams, events, captured graphs, RCCL (launches are synchronous here) -- tests/test_gpu_zz_dp.py."""
import os
import socket
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.skipif(not os.path.exists(os.environ.get("Y5M_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")),
                                reason="host clang of the ROCm image not present (tests/emu compiles the kernel sources for its CPU executor with it)")
sys.path.insert(0, HERE)


def _model(dtype="bf16"):
    from yolov5m_amd import config
    from yolov5m_amd.model import YOLOV5m
    from yolov5m_amd.utils.synth import synth_state_dict
    m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
    m.load_state_dict(synth_state_dict(), strict=True)
    m.compute_dtype = dtype
    m.train()
    m.flatten_parameters()
    return m


@pytest.mark.parametrize("after_dgrad", ["1", "0"])
def test_plan_marks_and_cuts(after_dgrad, monkeypatch):
    from emu.harness import emulated
    from yolov5m_amd import _lib
    from yolov5m_amd.engine import Engine
    monkeypatch.setenv("Y5M_WGRAD_AFTER_DGRAD", after_dgrad)
    with emulated():
        m = _model()
        eng = Engine(m, 2, 64, 64, dtype=_lib.BF16, training=True)        # (_check_grad_writers runs inside)
        n = m.flat_grads.numel()
        ks = [k for k, _, _ in eng.bwd_marks]
        assert ks == sorted(ks) and ks[-1] <= len(eng.bwd)
        # every parameter-gradient writer of the list sits in front of its unit's mark; the reorder moved weight gradients
        kinds = [getattr(op[0], "kind", None) for op in eng.bwd]
        if after_dgrad == "1":
            assert any(a == "conv_igemm" and b == "wgrad" for a, b in zip(kinds, kinds[1:]))
        cuts = eng.grad_cuts()
        assert len(cuts) == 2
        (k1, lo1), (k2, lo2) = cuts
        assert 0 < k1 < k2 < len(eng.bwd) and n > lo1 > lo2 > 0
        assert n - lo1 >= 0.5 * n and n - lo2 >= 0.9 * n
        marks = {k for k, _, _ in eng.bwd_marks}
        assert k1 in marks and k2 in marks
        # a cut never separates a unit's gradient writers from its mark
        for k, _ in cuts:
            assert kinds[k - 1] in Engine._GRAD_WRITER_KINDS + ("conv_igemm", "join")


@pytest.mark.parametrize("min_m", ["200000", "0"])      # default dispatch at this size | every eligible 1x1 CBL on bwd_pw_kernel
def test_gradients_behind_a_cut_are_final(min_m, monkeypatch):
    """run the backward list cut by cut: what a cut declares final must not change afterwards (bit-exact)"""
    monkeypatch.setenv("Y5M_BWD_PW_MIN_M", min_m)
    from emu.harness import emulated
    from yolov5m_amd.ultralytics_loss import ComputeLoss
    from yolov5m_amd.utils.synth import synth_images, synth_labels
    from yolov5m_amd.utils.training_utils import NativeTrainStep
    with emulated():
        m = _model()
        st = NativeTrainStep(m, ComputeLoss(m), nt_max=64)
        eng = st.load_inputs(synth_images(2, 64, 64, seed="cut/img"), synth_labels(2, 4, seed="cut/lab"))
        (k1, lo1), (k2, lo2) = eng.grad_cuts()
        kinds = [getattr(op[0], "kind", None) for op in eng.bwd]
        assert ("bwd_pw" in kinds) == (min_m == "0") and "bwd_stem" in kinds
        flat = m.flat_grads
        st._enqueue_fb(eng, bwd_upto=k1)
        s1 = flat[lo1:].clone()
        assert float(s1.abs().max()) > 0
        eng._run(eng.bwd[k1:k2])
        eng.join_all()
        assert torch.equal(flat[lo1:], s1), "a launch behind the first cut wrote gradients that cut declared final"
        s2 = flat[lo2:].clone()
        eng._run(eng.bwd[k2:])
        assert torch.equal(flat[lo2:], s2), "a launch behind the second cut wrote gradients that cut declared final"
        assert float(flat[:lo2].abs().max()) > 0                      # (the last segment did write its own part)


def test_per_layer_backward_helper_of_the_full_size_gpu_test(monkeypatch):
    """tests/test_gpu_model.py::test_bf16_per_layer_backward_full_size_default_dispatch checks one layer per shape class at
    B = 64 @ 640x640 on the GPU; here its helper runs at 2 x 64 x 64 on the CPU executor with every eligible 1x1 CBL on the
    fused pointwise kernel, so that the selection logic and the layer-local restatement are themselves tested on this machine"""
    monkeypatch.setenv("Y5M_BWD_PW_MIN_M", "0")
    from emu.harness import emulated
    import test_gpu_model as T
    monkeypatch.setattr(T, "DEV", "cpu")
    with emulated():
        worst, checked_dx, names, kinds = T._per_layer_backward(2, 64, one_per_shape=True)
    assert kinds.count("bwd_pw") >= 10 and kinds.count("bwd_stem") == 1
    assert checked_dx >= 8 and max(worst.values()) <= 2e-2
    assert any(n.startswith("wgrad_rows_kernel") for n in names) and any(n.startswith("conv_halo_kernel<6,3>") for n in names), names


_SLAB_CHILD = r"""
import sys, os, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from emu.harness import emulated
from yolov5m_amd import config
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.utils.synth import synth_images, synth_state_dict
with emulated():
    m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
    m.load_state_dict(synth_state_dict(), strict=True)
    m.compute_dtype = "bf16"; m.eval()
    with torch.no_grad():
        o = m(synth_images(5, 64, 96, seed="slab"))
    np.savez(sys.argv[1], *[t.numpy() for t in o])
"""


def test_conv_slab_path_equals_one_launch(tmp_path):
    """BASELINE.json configs[4] runs every layer whose input view exceeds 2 GiB in slabs of whole images (y5m_conv); the only
    GPU test of that loop needs B = 128 @ 1280x1280. Y5M_CONV_SLAB_BYTES lowers the limit (read once: child processes): the
    eval forward of 5 images with a 40 KB limit -- one to three images per slab depending on the layer, a ragged last slab,
    residual and head epilogues included -- equals the one-launch forward bit for bit."""
    import subprocess
    import numpy as np
    root = os.path.dirname(HERE)
    outs = []
    for tag, env in (("one", {}), ("slabs", {"Y5M_CONV_SLAB_BYTES": "40960"})):
        f = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, "-c", _SLAB_CHILD % (root, HERE), f], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(f))
    for k in outs[0].files:
        assert np.array_equal(outs[0][k], outs[1][k]), k
        assert np.isfinite(outs[0][k]).all() and np.abs(outs[0][k]).max() > 0


def test_plan_cache_retries_after_oom_outside_the_handler(monkeypatch):
    """the retry after an out-of-memory plan build must run AFTER the except block (the traceback keeps the failed
    plan alive inside it), with every resident plan released first"""
    import gc
    from emu.harness import emulated
    import yolov5m_amd.engine as E
    with emulated():
        m = _model()
        x = torch.zeros((1, 3, 64, 64))
        first = m._engine_for(x)
        real, calls, state = E.Engine, [], {}

        def flaky(*a, **k):
            calls.append(sys.exc_info()[0])              # the exception being handled at the time of the call, if any
            if len(calls) == 1:
                state["resident_at_first_try"] = len(m._engines)
                raise torch.OutOfMemoryError("synthetic")
            state["resident_at_retry"], state["first_released"] = len(m._engines), first.released
            return real(*a, **k)
        monkeypatch.setattr(E, "Engine", flaky)
        eng = m._engine_for(torch.zeros((1, 3, 96, 64)))
        assert len(calls) == 2 and calls[1] is None, "the retry ran inside the exception handler"
        assert state["resident_at_first_try"] == 1 and state["resident_at_retry"] == 0 and state["first_released"]
        assert not eng.released and list(m._engines.values()) == [eng]
        gc.collect()


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), Y5M_EMU_THREADS=str(max(1, 8 // world)))
    torch.set_num_threads(max(1, 8 // world))
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from emu.harness import emulated
    from yolov5m_amd import parallel
    from yolov5m_amd.ultralytics_loss import ComputeLoss
    from yolov5m_amd.utils.synth import synth_images, synth_labels
    from yolov5m_amd.utils.training_utils import NativeTrainStep
    try:
        with emulated():
            r, _, w = parallel.init_from_env(backend="gloo")
            x, t = synth_images(2, 64, 64, seed=f"dp/img{rank}"), synth_labels(2, 4, seed=f"dp/lab{rank}")
            graphs = world <= 2
            ref = None
            if graphs:
                # reference: this rank's plain gradient (a SEPARATE backward pass), summed over the ranks with ONE collective
                m0 = _model()
                s0 = NativeTrainStep(m0, ComputeLoss(m0), nt_max=64)
                s0._enqueue_fb(s0.load_inputs(x, t))
                ref = m0.flat_grads.clone()
                dist.all_reduce(ref)
            # the overlapped schedule: segments of the backward list, one bucket launched behind each, waited for once
            m = _model()
            parallel.broadcast_parameters(m)
            hook = parallel.GradAllReduce(world)
            # use_graph=True: the harness records the per-segment captures, the second step REPLAYS them with a bucket behind each
            # (world 8: one eager overlapped step -- eight ranks share this machine's eight cores)
            # (world 8: eight ranks share this machine's eight cores, so no second backward pass -- the reference is the local
            #  gradient of every bucket as it stands when the bucket is launched, summed over the ranks afterwards; that a bucket
            #  is FINAL at that point is test_gradients_behind_a_cut_are_final's statement, bit-exact)
            launched, local = [], {}
            real_launch = hook.launch

            def spy(flat, lo, hi):
                launched.append((lo, hi))
                local[(lo, hi)] = flat[lo:hi].clone()
                return real_launch(flat, lo, hi)
            hook.launch = spy
            step = NativeTrainStep(m, ComputeLoss(m), nt_max=64, use_graph=graphs, grad_hook=hook, overlap=True)
            p0 = m.flat_params.clone()
            step.step(x, t)
            cuts = m._engines[next(iter(m._engines))]._cuts
            # three buckets in backward order (head first), contiguous, covering the whole flat buffer: 84.76 MB
            n = m.flat_grads.numel()
            assert len(launched) == 3 and launched[0][1] == n and launched[-1][0] == 0, launched
            assert all(a[0] == b[1] for a, b in zip(launched, launched[1:])) and sum(hi - lo for lo, hi in launched) * 4 == 84762228, launched
            if ref is None:
                ref = torch.empty_like(m.flat_grads)
                for (lo, hi), v in local.items():
                    ref[lo:hi] = v
                dist.all_reduce(ref)
            gerr = float((m.flat_grads - ref).abs().max() / ref.abs().max())
            if graphs:
                segs = step._fb_graphs[next(iter(step._fb_graphs))]
                assert segs[2] == "segments" and len(segs[1]) == 3 and all(len(g.ops) > 0 for g in segs[1]) and len(step._opt_graph.ops) == 3
                p1 = m.flat_params.clone()
                step.step(x, t)                                  # replayed segments
                assert float((m.flat_params - p1).abs().max()) > 0 and int(step.d_step) == 2
            mine = m.flat_params.clone()
            others = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(others, mine)
            same = all(torch.equal(o, others[0]) for o in others)
            moved = float((mine - p0).abs().max())
            q.put((rank, len(cuts), gerr, same, moved, None))
            dist.barrier()
            dist.destroy_process_group()
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, 0, 0.0, False, 0.0, traceback.format_exc()[-2000:]))
        raise


@pytest.mark.parametrize("world", [2, 8])
def test_dp_overlapped_schedule_gloo_ranks(world):
    """world 2: eager warm-up + captured segments + a replayed step. world 8 (the driver's SCALE run is the first time the
    data-parallel path meets eight ranks on hardware): make_buckets / grad_cuts / broadcast_parameters / the bucketed exchange with
    EIGHT gloo ranks, one overlapped step each on its own batch."""
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    from emu.harness import emu_lib_path
    emu_lib_path()                                     # build once, before the ranks race for it
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=1500) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, ncuts, gerr, same, moved, err in res:
        assert err is None, err
        assert ncuts == 2
        # overlapped == plain up to the order of the f32 atomic adds of two separate backward passes (bf16 operands; a bucket
        # exchanged before its gradients were final, a wrong range or a missing wait would be O(1))
        assert gerr < 1e-4, gerr
        assert same, "parameters differ across ranks after the optimizer"
        assert moved > 0
    assert all(p.exitcode == 0 for p in procs)


def test_eval_plan_packs_once_per_weight_version(monkeypatch):
    """an inference plan packs its weights / folds BatchNorm once per weight VERSION (Engine.forward): the second forward of the same
    weights launches no pack; an in-place edit of a parameter, of a running statistic, a load_state_dict and a native train step
    (which writes the masters through raw pointers) each make the next inference forward pack again -- and its logits equal those of
    a model built fresh from the same state"""
    from emu.harness import emulated
    from yolov5m_amd import config
    from yolov5m_amd.engine import Engine
    from yolov5m_amd.model import YOLOV5m
    from yolov5m_amd.ultralytics_loss import ComputeLoss
    from yolov5m_amd.utils.synth import synth_images, synth_labels, synth_state_dict
    from yolov5m_amd.utils.training_utils import NativeTrainStep
    packs = []
    real = Engine._run

    def counting(lst, timeline=None):
        if lst and getattr(lst[0][0], "kind", None) == "pack_weights":
            packs.append(1)
        return real(lst, timeline)
    monkeypatch.setattr(Engine, "_run", staticmethod(counting))

    def fresh(sd):
        f = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
        f.load_state_dict(sd, strict=True)
        f.compute_dtype = "bf16"
        f.eval()
        with torch.no_grad():
            return [o.clone() for o in f(x)]
    with emulated():
        x = synth_images(1, 64, 64, seed="ver/img")
        # default: every forward packs (always correct, whoever wrote the masters) -- also when the FIRST forward of the model runs
        # under torch.inference_mode() (the flat buffers must not become inference tensors, which have no version counter)
        m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
        m.load_state_dict(synth_state_dict(), strict=True)
        m.compute_dtype = "bf16"
        m.eval()
        with torch.inference_mode():
            oi = [o.clone() for o in m(x)]
            m(x)
        assert len(packs) == 2 and not m.flat_params.is_inference()
        with torch.no_grad():
            m.backbone[0].cbl[0].weight.data.mul_(0.5)              # a writer torch's counters do not see: still picked up
            assert not torch.equal(m(x)[0], oi[0]) and len(packs) == 3
        m.pack_once = True
        with torch.inference_mode():                                 # (the opt-in works under inference_mode too)
            m(x); m(x)
        assert len(packs) == 4
        packs.clear()
        m = _model()
        m.eval()
        m.pack_once = True                                           # the opt-in: once per weight version
        with torch.no_grad():
            o1 = [o.clone() for o in m(x)]
            n1 = len(packs)
            o2 = [o.clone() for o in m(x)]
        assert n1 == 1 and len(packs) == 1 and all(torch.equal(a, b) for a, b in zip(o1, o2)) and all(torch.equal(a, b) for a, b in zip(o1, oi))
        with torch.no_grad():
            m.backbone[0].cbl[0].weight.mul_(0.5)                    # in-place edit of a parameter
            o3 = [o.clone() for o in m(x)]
        assert len(packs) == 2 and not torch.equal(o3[0], o1[0])
        with torch.no_grad():
            m.backbone[1].cbl[1].running_var.mul_(2.0)               # ... of a running statistic
            o4 = [o.clone() for o in m(x)]
        assert len(packs) == 3 and not torch.equal(o4[0], o3[0])
        with torch.no_grad():
            m.backbone[0].cbl[0].weight.data.mul_(2.0)              # behind torch's back (a fresh .data view has its own counter) ...
            n = len(packs)
            m(x)
            assert len(packs) == n                                   # ... is invisible (documented) until the model is told
            m.mark_weights_changed()
            m(x)
            assert len(packs) == n + 1
        packs.clear()
        m.load_state_dict(synth_state_dict(), strict=True)              # load_state_dict
        with torch.no_grad():
            o5 = [o.clone() for o in m(x)]
        assert len(packs) >= 1 and all(torch.equal(a, b) for a, b in zip(o5, o1))
        # a native train step: masters, running statistics and num_batches_tracked written by kernels
        m.train()
        st = NativeTrainStep(m, ComputeLoss(m), nt_max=64)
        st.step(synth_images(2, 64, 64, seed="ver/t"), synth_labels(2, 4, seed="ver/l"))
        m.eval()
        packs.clear()
        with torch.no_grad():
            o6 = [o.clone() for o in m(x)]
        assert len(packs) == 1 and not torch.equal(o6[0], o1[0])
        ref = fresh({k: v.clone() for k, v in m.state_dict().items()})
        assert all(torch.equal(a, b) for a, b in zip(o6, ref))
