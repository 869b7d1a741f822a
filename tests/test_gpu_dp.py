"""GPU: the data-parallel step with a REAL RCCL collective (backend "nccl" = RCCL) -- single rank, in a subprocess:
process-group init, parameter broadcast, and a SUM all_reduce of the flat gradient buffer between the two captured
hipGraphs of every step. (World size > 1 needs more than one GPU: the algorithm is covered on CPU by
tests/test_parallel_cpu.py with gloo, world size 2.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", Y5M_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tools", "dp_parity.py")], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "dp parity ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
