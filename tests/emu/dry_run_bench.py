"""Developer tool: bench.py's main() and smoke.run() executed on the CPU lane-level executor at a tiny size (eager, no graphs), so that a
Python-level mistake in the driver-facing entry points shows up on this CPU-only container instead of in the driver's round-end run.
The torch.cuda calls bench.py makes are replaced by stand-ins; "cuda:N" device strings are redirected to the host.
    python tests/emu/dry_run_bench.py                       # smoke.run() + bench.py --gpus 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29713 tests/emu/dry_run_bench.py 2
TEST INFRASTRUCTURE (not collected by pytest: two minutes of fibers)."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("Y5M_DIST_BACKEND", "gloo")
os.environ.setdefault("Y5M_EMU_THREADS", "4")
import torch  # noqa: E402
from emu.harness import emulated  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
with emulated():
    class P:
        total_memory = 64 << 30; name = "emu"; gcnArchName = "gfx950"; pci_bus_id = 3; pci_device_id = 0; pci_domain_id = 0; uuid = "emu"
    torch.cuda.get_device_properties = lambda *a, **k: P()
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.current_device = lambda: 0
    torch.cuda.device_count = lambda: max(n, 1)
    torch.cuda.empty_cache = lambda: None
    fix = lambda x: "cpu" if (isinstance(x, str) and x.startswith("cuda")) else x
    ot, om, tt = torch.Tensor.to, torch.nn.Module.to, torch.tensor
    torch.Tensor.to = lambda self, *a, **k: ot(self, *[fix(x) for x in a], **{kk: fix(v) for kk, v in k.items()})
    torch.nn.Module.to = lambda self, *a, **k: om(self, *[fix(x) for x in a], **k)
    torch.tensor = lambda *a, **k: tt(*a, **{kk: fix(v) for kk, v in k.items()})
    for name in ("randn", "rand", "zeros", "ones", "empty", "full", "arange", "randint"):
        setattr(torch, name, (lambda f: lambda *a, **k: f(*a, **{kk: fix(v) for kk, v in k.items()}))(getattr(torch, name)))
    import torch.distributed.tensor  # noqa: F401  (its annotations evaluate `torch.Generator | None` at import: before the stand-in)
    og = torch.Generator
    torch.Generator = lambda device="cpu": og(device=fix(device))

    # stand-ins of HIP events and streams (launches are synchronous here): an event is the host clock at record()
    import time

    class Ev:
        def __init__(self, enable_timing=False): self.t = 0.0
        def record(self, stream=None): self.t = time.perf_counter()
        def elapsed_time(self, other): return max((other.t - self.t) * 1e3, 1e-6)
        def synchronize(self): pass
        def wait(self, *a): pass
        def query(self): return True

    class St:
        cuda_stream = 0
        def __init__(self, *a, **k): pass
        def wait_event(self, e): pass
        def wait_stream(self, s): pass
        def synchronize(self): pass
        def record_event(self, e=None):
            e = e or Ev(); e.record(); return e
    import contextlib
    torch.cuda.Event, torch.cuda.Stream = Ev, St
    torch.cuda.current_stream = lambda *a, **k: St()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    if n == 1 and "--no-smoke" not in sys.argv:
        import yolov5m_amd.smoke as S
        S.run()
    # every untimed leg of bench.py too (roofline = profile_step with an event pair per launch, forward, detect, cpu_baseline), each
    # at a tiny size: a Python error in any of them shows here, not in the driver's round-end run
    B_, S_ = os.environ.get("Y5M_DRY_SHAPE", "2x64").split("x")
    sys.argv = ["bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "1", "--leg-iters", "1", "--batch", B_, "--size", S_, "--no-graph",
                "--fwd-shape", f"{B_}x{S_}", "--detect-shape", f"{B_}x{S_}", "--cpu-shape", f"1x{S_}"] + os.environ.get("Y5M_DRY_ARGS", "").split()      # (e.g. Y5M_DRY_ARGS="--loss yolo --no-detect")
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
