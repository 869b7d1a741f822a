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
