#!/usr/bin/env python
"""Per-launch time of every weight-gradient launch (+ its unpack) of one eager, serialised train step (HIP events on
the launch stream), grouped by layer shape: which layers the serial cost of the weight gradients consists of.
usage: python tools/wgrad_layers.py [B] [size]"""
import ctypes, os, sys
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
saved, eng.overlap = eng.overlap, False
agg = {}
L = _lib.lib()
buf = ctypes.create_string_buffer(192)
for rep in range(3):
    tl = []
    step._enqueue_fb(eng, tl)
    torch.cuda.synchronize()
    items = list(eng.pack) + list(eng.fwd) + [None] + list(eng.bwd)
    for (kind, e0, e1), item in zip(tl, items):
        if item is None or kind != "wgrad" or getattr(item[0], "wa", None) is None:
            continue
        wa = item[0].wa
        L.y5m_wgrad_kernel_name(ctypes.byref(wa), eng.dtype, buf, 192)
        key = (wa.M, wa.N, wa.C, wa.th * wa.tw, wa.sy, buf.value.decode())
        a = agg.setdefault(key, [0.0, 0])
        a[0] += e0.elapsed_time(e1); a[1] += 1
eng.overlap = saved
tot = 0.0
print("  ms/step  launches  us/launch  TFLOP/s   M        N    C  taps s  kernel")
for (M, N, C, taps, s, name), (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    per_step, n_step = ms / 3, n // 3
    fl = 2.0 * M * N * C * taps * n_step
    tot += per_step
    print(f"{per_step:9.3f} {n_step:9d} {per_step / n_step * 1e3:10.1f} {fl / per_step / 1e9:8.1f}  {M:8d} {N:4d} {C:4d} {taps:3d}  {s}  {name}")
print(f"total {tot:.3f} ms/step (weight gradient + unpack, serialised)")
