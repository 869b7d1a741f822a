#!/usr/bin/env python
"""multi_scale training with MORE sizes than the plan cache holds: every step may evict a plan (Engine.release + its
captured graphs destroyed), build + capture another and replay resident ones -- the churn a B=64 run sees when the two
largest sizes take turns. usage: multiscale_churn.py [cache] [rounds] [batch] [graph 0|1]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cache = sys.argv[1] if len(sys.argv) > 1 else "6"
os.environ["Y5M_ENGINE_CACHE"] = cache
import torch
from yolov5m_amd import config
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.ultralytics_loss import ComputeLoss
from yolov5m_amd.utils.training_utils import NativeTrainStep
from yolov5m_amd.utils.synth import synth_images, synth_labels

rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
graph = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
torch.manual_seed(0)
m = YOLOV5m(48, 80, config.ANCHORS, (192, 384, 768)).to("cuda"); m.compute_dtype = "bf16"; m.train()
step = NativeTrainStep(m, ComputeLoss(m), nt_max=B * 8, use_graph=graph)
sizes = list(range(320, 641, 32))
rng = random.Random(0)
data = {s: (synth_images(B, s, s, seed=f"mc{s}").to("cuda"), synth_labels(B, 4, seed=f"mcl{s}")) for s in sizes}
import time
tim = {"plan": 0.0, "capture": 0.0, "nplan": 0, "ncap": 0}
_ef, _cap = m._engine_for, step._capture
def ef(x):
    n0 = len(getattr(m, "_built", []))
    t0 = time.time(); keys = set(m._engines); e = _ef(x); torch.cuda.synchronize(); dt = time.time() - t0
    if e.key not in keys:
        tim["plan"] += dt; tim["nplan"] += 1
    return e
def cap(eng, enq):
    t0 = time.time(); _cap(eng, enq); torch.cuda.synchronize(); tim["capture"] += time.time() - t0; tim["ncap"] += 1
m._engine_for, step._capture = ef, cap
n = 0
for r in range(rounds):
    order = sizes[:]
    rng.shuffle(order)
    for s in order:
        x, t = data[s]
        l = float(step.step(x, t)[0])
        n += 1
        print(f"step {n:3d} size {s} loss {l:.4f} plans {len(m._engines)} graphs {len(step._fb_graphs)} plan_s {tim['plan']:.1f}/{tim['nplan']} capture_s {tim['capture']:.1f}/{tim['ncap']}", flush=True)
print("ok", {k: round(v, 2) if isinstance(v, float) else v for k, v in tim.items()})
