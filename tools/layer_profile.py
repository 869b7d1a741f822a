#!/usr/bin/env python
"""Per-launch timing of the conv / wgrad kernels inside one eager train step (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5m_amd import config, _lib
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.ultralytics_loss import ComputeLoss
from yolov5m_amd.utils.training_utils import NativeTrainStep
from yolov5m_amd.utils.synth import synth_images, synth_labels

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 640
torch.manual_seed(0)
m = YOLOV5m(48, 80, config.ANCHORS, (192, 384, 768)).to("cuda"); m.compute_dtype = "bf16"; m.train()
step = NativeTrainStep(m, ComputeLoss(m), nt_max=B * 8)
x = synth_images(B, S, S).to("cuda"); t = synth_labels(B, 8).to("cuda")
for _ in range(2):
    step.step(x, t)
eng = step.load_inputs(x, t)
tl = []
step._enqueue_fb(eng, tl)
torch.cuda.synchronize()
lists = list(eng.pack) + list(eng.fwd) + [None] + list(eng.bwd)
rows = []
for (kind, e0, e1), item in zip(tl, lists):
    ms = e0.elapsed_time(e1)
    if item is None:
        continue
    fn = item[0]
    d = getattr(fn, "__defaults__", None)
    if kind == "conv_igemm" and d and hasattr(d[0], "_length_"):
        parts = list(d[0])
        fl = sum(2.0 * a.M * a.N * a.K for a in parts)
        a = parts[-1]
        rows.append((ms, f"conv  M={a.M:8d} N={a.N:4d} K={sum(q.K for q in parts):5d} taps=multi x{len(parts)} s=2 epi={a.epi} acc={a.accumulate}  {fl/ms/1e9:7.1f} TF/s"))
    elif kind == "conv_igemm" and d:
        a = d[0]
        fl = 2.0 * a.M * a.N * a.K
        rows.append((ms, f"conv  M={a.M:8d} N={a.N:4d} K={a.K:5d} taps={a.th}x{a.tw} s={a.sy} epi={a.epi} acc={a.accumulate}  {fl/ms/1e9:7.1f} TF/s"))
    elif kind == "wgrad" and d:
        a = d[0]
        fl = 2.0 * a.M * a.N * a.C * a.th * a.tw
        rows.append((ms, f"wgrad M={a.M:8d} N={a.N:4d} C={a.C:5d} taps={a.th}x{a.tw} s={a.sy}  {fl/ms/1e9:7.1f} TF/s"))
    elif ms > 0.15:
        rows.append((ms, f"{kind}"))
tot = sum(r[0] for r in rows)
agg = {}
for ms, s in rows:
    k = s.split("  ")[0] if s.startswith(("conv", "wgrad")) else s
    a = agg.setdefault(s if not s.startswith(("conv", "wgrad")) else s.rsplit("  ", 1)[0], [0.0, 0, s])
    a[0] += ms; a[1] += 1; a[2] = s
for k, (ms, n, s) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{ms:8.3f} ms x{n:3d}  {s}")
print("total listed", tot)
