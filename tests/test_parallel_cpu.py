"""CPU (gloo, world_size 2): the data-parallel pieces -- bucket partition of the flat gradient buffer,
SUM all-reduce (whole and bucket-by-bucket, async) and parameter broadcast -- behave as the N>1 bench
path assumes. No GPU: the same code runs with backend "nccl" (= RCCL) on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolov5m_amd import parallel
from yolov5m_amd.arch import state_dict_spec


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, l, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # data-parallel runs cap the persistent (one-workgroup-per-CU) kernels at 240 CUs: the default is set by init_from_env
    # before the library reads it, and the library reports what it will use (256 CUs assumed where no device answers)
    from yolov5m_amd import _lib
    assert os.environ.get("Y5M_PERSIST_CUS") == "240" and _lib.lib().y5m_persistent_cu_count() == 240
    n = 100003
    g = torch.Generator().manual_seed(1234 + rank)
    flat = torch.rand(n, generator=g)
    ref = sum(torch.rand(n, generator=torch.Generator().manual_seed(1234 + k)) for k in range(world))
    ar = parallel.GradAllReduce(world)
    whole = ar(flat.clone())
    ok1 = torch.allclose(whole, ref, rtol=1e-6, atol=1e-6)
    # bucketed + asynchronous, in backward order
    b = flat.clone()
    bounds = list(range(0, n, 7919))
    buckets = parallel.make_buckets(n, bounds, target_bytes=64 << 10)
    for lo, hi in buckets:
        ar.launch(b, lo, hi)
    ar.wait()
    ok2 = torch.allclose(b, ref, rtol=1e-6, atol=1e-6)

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.full((5,), float(rank + 1)))
            self.register_buffer("rm", torch.full((3,), float(rank + 10)))
    m = M()
    parallel.broadcast_parameters(m, src=0)
    ok3 = bool((m.w == 1).all()) and bool((m.rm == 10).all())
    q.put((rank, ok1, ok2, ok3))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_and_broadcast_gloo_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, ok3 in res:
        assert ok1 and ok2 and ok3, (rank, ok1, ok2, ok3)


def test_buckets_cover_flat_buffer_in_backward_order():
    offs, off = [], 0
    for _, shape, kind in state_dict_spec():
        if kind in ("conv", "bn_w", "bn_b", "head_w", "head_b"):
            offs.append(off)
            off += int(np.prod(shape)) if len(shape) else 1
    n = off
    assert n == 21190557
    bk = parallel.make_buckets(n, offs, target_bytes=16 << 20)
    assert bk[0][1] == n and bk[-1][0] == 0                       # head first, stem last
    for (lo, hi), (lo2, hi2) in zip(bk[:-1], bk[1:]):
        assert lo == hi2 and lo2 < hi2                            # contiguous, descending
    assert all(lo in offs or lo == 0 for lo, _ in bk)             # cut only at layer-unit starts
    assert sum(hi - lo for lo, hi in bk) == n
    assert 4 <= len(bk) <= 12


def test_bench_gpus_n_refuses_without_devices():
    """`python bench.py --gpus 2` where fewer than 2 GPUs are visible (here: none) must fail loudly -- exit code != 0, the
    reason on stderr, no JSON line -- instead of reporting a 2-GPU number from one rank (the round-2 behaviour)"""
    import subprocess, sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two or more GPUs visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "Y5M_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())
