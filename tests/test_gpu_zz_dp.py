"""GPU: the data-parallel step. (1) a REAL RCCL collective (backend "nccl" = RCCL) with a single rank, in a subprocess:
process-group init, parameter broadcast, the bucketed exchange issued from the communication stream between the captured
per-segment hipGraphs; (2) two ranks sharing this box's one GPU over gloo (RCCL refuses two ranks on one device): summed
gradient parity, identical parameters, overlapped == plain schedule; (3) `python bench.py --gpus 2` WITHOUT
torch.distributed.run in front, the way the driver calls it: bench.py starts the ranks itself and its line says so.
RCCL with more than one rank needs more than one GPU: the driver's 8-GPU run is its first execution."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (file name: collected AFTER test_gpu_model.py, so that `-x` never hides the model goldens behind a subprocess test)


def _keep_log(tag, r):
    """a failing child's complete stdout / stderr goes to gpurun_out/ (merged back from the GPU box): the assertion message
    alone truncates it"""
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, f"{tag}_{os.getpid()}.log")
    with open(path, "w") as f:
        f.write(f"returncode {r.returncode}\n---- stdout ----\n{r.stdout}\n---- stderr ----\n{r.stderr}\n")
    return path


def test_rccl_allreduce_between_graphs_single_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_smoke.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "dp smoke ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_dp_two_ranks_gradient_sum_and_identical_parameters():
    """SURVEY 8e parity for data parallelism, two ranks sharing this GPU over gloo (RCCL refuses two ranks on one
    device; the collective calls are the same): the exchanged gradient == sum of the single-replica gradients, the
    bucketed exchange is overlapped with the backward segments (per-segment hipGraphs), parameters stay bit-identical
    across ranks, and the overlapped schedule lands where the plain one does. tools/dp_parity.py holds the checks."""
    import socket
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", Y5M_DIST_BACKEND="gloo")
    with socket.socket() as sk:                     # a free rendezvous port (a fixed one can still sit in TIME_WAIT from an earlier run)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "dp_parity.py")], env=env,
                       capture_output=True, text=True, timeout=900)
    if r.returncode != 0 or "dp parity ok" not in r.stdout:
        path = _keep_log("dp_parity", r)
        checks = [l for l in r.stdout.splitlines() if l.startswith("CHECK") or "FAILED" in l]
        raise AssertionError((path, checks, r.stdout[-3000:], r.stderr[-3000:]))


def test_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2 ...` (no torchrun): bench.py re-executes itself under torch.distributed.run with two ranks
    (gloo here, folded onto the one GPU) and rank 0's JSON line reports n_gpus == rccl_ranks == 2, the global batch of both
    ranks, and the exchange statistics (three buckets in backward order + the exposed wait)."""
    import json
    env = dict(os.environ, Y5M_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--batch", "2", "--size", "320", "--no-roofline", "--no-detect", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["dist_backend"] == "gloo" and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    ex = d["exchange"]["last_step_max_over_ranks"]
    assert len(ex["buckets"]) == 3 and abs(sum(b["MB"] for b in ex["buckets"]) - 84.76) < 0.1, ex
    assert ex["allreduce_exposed_ms"] >= 0.0
    assert len(d["ranks"]) == 2 and all(r["persistent_cus"] == 240 for r in d["ranks"]) and d["ranks"][1]["rank"] == 1
    assert d["exchange"]["plain_exchange_same_run"]["ms_per_step"] > 0 and d["exchange"]["overlap_gain_ms_per_step_approx"] is not None
    assert d["value"] > 0 and abs(d["value"] - 4 * 3 / (d["ms_per_step"] * 3e-3)) < 0.02 * d["value"]


def test_bench_refuses_more_ranks_than_gpus():
    """--gpus N with fewer than N visible devices (and no gloo folding) must fail loudly, never print an N-GPU line"""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "Y5M_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())
