#!/usr/bin/env python
"""Data-parallel parity, run under torch.distributed.run with WORLD_SIZE ranks (2 in the test; backend from
Y5M_DIST_BACKEND: gloo lets both ranks share one GPU, nccl = RCCL needs one GPU per rank).
  1. one DP step (f32, bucketed exchange overlapped with the backward segments): the exchanged flat gradient equals the SUM of
     the single-replica gradients of every rank's batch, computed here without any collective;
  2. three more steps with captured per-segment graphs: parameters bit-identical on all ranks;
  3. the same schedule with the un-overlapped exchange (one all-reduce after the backward pass): same first-step gradient;
  4. REPLAYED steps (the captured per-segment graphs) of both schedules from identical parameters at lr = 0: the exchanged
     gradients agree to the noise of the f32 atomic order. (Until round 4 this compared the parameters after four Adam steps
     of both schedules to 2e-2; that bound sits inside its own noise -- profiles/r04_adam_noise_emu.txt: a gradient noise of
     2e-7 already moves the 4-step update by 2.5e-3, Adam's first updates being +-lr whatever the gradient's size -- and was
     the likely source of round 3's one-in-ten red run. It is still printed, with a bound only a gross error reaches.)
Y5M_DP_EMU=1 runs the script's logic on the CPU lane-level executor of tests/emu (host tensors, recorded graphs, Y5M_DP_SHAPE=B,H,W)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import contextlib
import numpy as np
import torch, torch.distributed as dist
EMU = os.environ.get("Y5M_DP_EMU") == "1"
if EMU:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emu.harness import emulated
    _ctx = emulated()
    _ctx.__enter__()
from yolov5m_amd import config, parallel
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.ultralytics_loss import ComputeLoss
from yolov5m_amd.utils.training_utils import NativeTrainStep
from yolov5m_amd.utils.synth import synth_images, synth_labels, synth_state_dict

rank, local, world = parallel.init_from_env(backend="gloo" if EMU else None)
assert world > 1, "run under torch.distributed.run with --nproc-per-node >= 2"
dev = "cpu" if EMU else f"cuda:{torch.cuda.current_device()}"
GRAPH = True                     # (on the CPU executor the harness records captures: tests/emu/harness.py)
SHAPE = tuple(int(v) for v in os.environ.get("Y5M_DP_SHAPE", "2,96,128").split(","))
FAILED = []


def check(name, value, bound):
    """every bound of this script goes through here: the value is printed on every rank (so a soak gives its distribution)
    and a violation is recorded and reported at the end instead of hiding the later checks behind the first assert"""
    ok = bool(value < bound)
    print(f"CHECK rank{rank} {name} value {value:.3e} bound {bound:.1e} {'ok' if ok else 'FAIL'}", flush=True)
    if not ok:
        FAILED.append((name, value, bound))


def model():
    m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
    m.load_state_dict(synth_state_dict(), strict=True)
    m = m.to(dev); m.compute_dtype = "f32"; m.train(); m.flatten_parameters()
    return m


def batch(r):
    return synth_images(*SHAPE, seed=f"dp/img{r}").to(dev), synth_labels(SHAPE[0], 4, seed=f"dp/lab{r}").to(dev)


# ---- 1. summed-gradient parity ------------------------------------------------------------------------------------
ref = None
for r in range(world):
    m = model()
    st = NativeTrainStep(m, ComputeLoss(m), nt_max=64)
    x, t = batch(r)
    eng = st.load_inputs(x, t)
    st._enqueue_fb(eng)
    torch.cuda.synchronize()
    ref = m.flat_grads.clone() if ref is None else ref + m.flat_grads
    m._engines.clear()
m = model()
parallel.broadcast_parameters(m)
hook = parallel.GradAllReduce(world, timing=True)
step = NativeTrainStep(m, ComputeLoss(m), nt_max=64, use_graph=GRAPH, grad_hook=hook, overlap=True)
x, t = batch(rank)
lo = step.step(x, t)
torch.cuda.synchronize()
cuts = step.model._engines[next(iter(step.model._engines))]._cuts
assert len(cuts) >= 1, cuts
err = float((m.flat_grads - ref).abs().max() / ref.abs().max())
check("summed_gradient_vs_replica_sum", err, 1e-4)
g_overlapped = m.flat_grads.clone()

# ---- 2. identical parameters after captured steps ------------------------------------------------------------------
for _ in range(3):
    lo = step.step(x, t)
torch.cuda.synchronize()
assert not GRAPH or (step._opt_graph is not None and isinstance(step._fb_graphs[next(iter(step._fb_graphs))][1], list))
mine = m.flat_params.clone()
gathered = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(gathered, mine)
pdiff = max(float((g - gathered[0]).abs().max()) for g in gathered)
check("parameters_differ_across_ranks_maxabs", pdiff, 1e-30)          # bit-identical: any difference fails
assert bool(torch.isfinite(mine).all()) and float(lo[0]) == float(lo[0])
st_ = hook.stats()
assert EMU or (st_ is not None and len(st_["buckets"]) == len(cuts) + 1 and st_["allreduce_exposed_ms"] >= 0.0), st_

# ---- 3. un-overlapped exchange, same schedule -----------------------------------------------------------------------
m2 = model()
parallel.broadcast_parameters(m2)
step2 = NativeTrainStep(m2, ComputeLoss(m2), nt_max=64, use_graph=GRAPH, grad_hook=parallel.GradAllReduce(world), overlap=False)
step2.step(x, t)
torch.cuda.synchronize()
# both schedules sum the same per-rank gradients of the same parameters: equal up to the order of the f32 atomic adds inside
# the weight-gradient kernels (a bucket exchanged before its gradients were final, a wrong cut or a missed join would be O(1))
gerr = float((m2.flat_grads - g_overlapped).abs().max() / g_overlapped.abs().max())
check("overlapped_vs_plain_exchanged_gradient", gerr, 2e-5)
for _ in range(3):
    step2.step(x, t)
torch.cuda.synchronize()
p0 = torch.cat([p.detach().reshape(-1) for p in model().parameters()])
d1, d2 = (mine - p0).cpu().numpy(), (m2.flat_params - p0).cpu().numpy()
rel = np.linalg.norm(d1 - d2) / np.linalg.norm(d2)
# (four Adam steps of two schedules: sign-like updates amplify the atomic-order noise of the gradients -- 2.5e-3 at a gradient noise
#  of 2e-7, profiles/r04_adam_noise_emu.txt; only a gross error -- a bucket exchanged before its gradients were final in every step --
#  reaches this bound)
check("overlapped_vs_plain_update_after_4_adam_steps(informational)", rel, 2e-1)
if rel > 5e-2:      # (the round-3 bound: kept as a WARN line -- tools/dp_soak.sh collects them -- until a GPU soak has recorded the distribution)
    print(f"WARN rank{rank} overlapped_vs_plain_update_after_4_adam_steps {rel:.3e} exceeds the round-3 bound 5e-2", flush=True)

# ---- 4. replayed steps of both schedules from identical parameters, lr = 0 ---------------------------------------------
m2.flat_params.copy_(m.flat_params)
step.lr = step2.lr = 0.0                                 # (_check_hyper drops the graphs: next call = eager step + capture, then replays)
for _ in range(3):
    step.step(x, t)
    step2.step(x, t)
torch.cuda.synchronize()
assert not GRAPH or (step._opt_graph is not None and step2._opt_graph is not None)
assert torch.equal(m2.flat_params, m.flat_params), "lr = 0 must leave the parameters where they were"
g_rep, g_rep2 = m.flat_grads, m2.flat_grads              # exchanged gradients of the third (second replayed) step of each schedule
gerr_rep = float((g_rep - g_rep2).abs().max() / g_rep2.abs().max())
check("replayed_overlapped_vs_replayed_plain_exchanged_gradient", gerr_rep, 2e-5)
dist.barrier()
if FAILED:
    print(f"dp parity FAILED on rank {rank}: {FAILED}", flush=True)
    dist.destroy_process_group()
    sys.exit(1)
if rank == 0:
    print(f"dp parity ok: world {world}, cuts {cuts}, grad err {err:.2e}, overlap-vs-plain gradient diff {gerr:.2e}, "
          f"replayed {gerr_rep:.2e}, update diff {rel:.2e}")
dist.destroy_process_group()
if EMU:
    del step, step2, m, m2
    _ctx.__exit__(None, None, None)
