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
    if n == 1:
        import yolov5m_amd.smoke as S
        S.run()
    sys.argv = ["bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "2", "--size", "64", "--no-graph", "--no-roofline",
                "--no-detect", "--no-cpu-baseline"]
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
