#!/usr/bin/env python
"""Where does the native step's first Adam update differ from autograd + torch.optim.Adam? (f32 atomics reorder the
gradient sums; Adam's first update lr*g/(|g|+eps) amplifies that for |g| ~ eps = 1e-8)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yolov5m_amd import config
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.ultralytics_loss import ComputeLoss
from yolov5m_amd.utils.training_utils import NativeTrainStep
from yolov5m_amd.utils.synth import synth_images, synth_labels, synth_state_dict
DEV = "cuda"
def model():
    m = YOLOV5m(48, 80, config.ANCHORS, (192, 384, 768))
    m.load_state_dict(synth_state_dict()); m = m.to(DEV); m.compute_dtype = "f32"; m.train(); return m
x = synth_images(2, 96, 128).to(DEV); t = synth_labels(2, 5, seed="lab3")
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    m1 = model()
    opt = torch.optim.Adam(m1.parameters(), lr=config.LEARNING_RATE, weight_decay=config.WEIGHT_DECAY)
    ComputeLoss(m1)(m1(x), t, None).backward()
    g1 = torch.cat([p.grad.reshape(-1) for p in m1.parameters()]).cpu().numpy()
    torch.nn.utils.clip_grad_norm_(m1.parameters(), max_norm=10.0); opt.step()
    m2 = model(); step = NativeTrainStep(m2, ComputeLoss(m2), nt_max=64); step.step(x, t)
    g2 = m2.flat_grads.cpu().numpy()
    p0 = torch.cat([p.detach().reshape(-1) for p in model().parameters()]).cpu().numpy()
    d1 = torch.cat([p.detach().reshape(-1) for p in m1.parameters()]).cpu().numpy() - p0
    d2 = m2.flat_params.cpu().numpy() - p0
    err = np.abs(d1 - d2)
    idx = np.argsort(-err)[:5]
    names, off = [], 0
    for n, p in m1.named_parameters():
        names.append((off, off + p.numel(), n)); off += p.numel()
    def nm(i):
        return next(n for a, b, n in names if a <= i < b)
    print(f"rep {rep}: max {err.max():.3e} ({err.max() / np.abs(d1).max():.3%} of the update), > 1% of lr: {(err > 5e-6).sum()} of {err.size}")
    for i in idx:
        print(f"    {nm(i):45s} g_autograd {g1[i]: .3e} g_native {g2[i]: .3e}  update {d1[i]: .3e} vs {d2[i]: .3e}")
