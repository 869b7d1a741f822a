#!/usr/bin/env python
"""Timing-only ablation of the train step: the named C-ABI entry points are replaced by no-ops (results are WRONG by
construction), to see how much of the step each family really costs once everything else overlaps around it.
usage: step_ablation.py [name,name,...] [B] [S]     e.g. step_ablation.py y5m_wgrad,y5m_unpack_wgrad"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5m_amd import _lib, config
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.ultralytics_loss import ComputeLoss
from yolov5m_amd.utils.training_utils import NativeTrainStep
from yolov5m_amd.utils.synth import synth_images, synth_labels

names = [n for n in (sys.argv[1] if len(sys.argv) > 1 else "").split(",") if n and n != "none"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = int(sys.argv[3]) if len(sys.argv) > 3 else 640
L = _lib.lib()
for n in names:
    getattr(L, n)                      # must exist
    setattr(L, n, lambda *a, **k: 0)
dev = "cuda:0"
torch.manual_seed(0)
model = YOLOV5m(first_out=config.FIRST_OUT, nc=80, anchors=config.ANCHORS,
                ch=(config.FIRST_OUT * 4, config.FIRST_OUT * 8, config.FIRST_OUT * 16)).to(dev)
model.compute_dtype = "bf16"
model.train()
model.flatten_parameters()
step = NativeTrainStep(model, ComputeLoss(model), nt_max=B * 8, use_graph=True)
images = step.input_buffer(B, S, S)
images.copy_(synth_images(B, S, S, seed="img/rank0").to(dev))
targets = synth_labels(B, 8, seed="lab/rank0").to(dev)
for _ in range(3):
    step.step(images, targets)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step.step(images, targets)
torch.cuda.synchronize()
print(f"stubbed={','.join(names) or 'none':60s} {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms/step", flush=True)
if os.environ.get("PER_STEP"):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    evs[0].record()
    for i in range(60):
        step.step(images, targets)
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(60)]
    print("per-step ms:", " ".join(f"{t:.2f}" for t in ts))
    print(f"min {min(ts):.3f} median {sorted(ts)[30]:.3f} mean {sum(ts)/60:.3f} max {max(ts):.3f}")
