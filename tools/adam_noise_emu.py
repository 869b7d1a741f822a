#!/usr/bin/env python
# (tools/: run with the repository root as the working directory; needs tests/emu, no GPU)
"""noise floor of dp_parity's check 4 on the CPU executor: the same 4 Adam steps (f32, 2x96x128, single rank) run twice;
the only difference between the runs is the order of the f32 atomic adds (8 OS threads take workgroups in varying order)"""
import sys, time, torch, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from emu.harness import emulated
from yolov5m_amd import config
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.ultralytics_loss import ComputeLoss
from yolov5m_amd.utils.training_utils import NativeTrainStep
from yolov5m_amd.utils.synth import synth_images, synth_labels, synth_state_dict
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def run():
    m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192,384,768))
    m.load_state_dict(synth_state_dict(), strict=True)
    m.compute_dtype = "f32"; m.train(); m.flatten_parameters()
    p0 = m.flat_params.clone()
    st = NativeTrainStep(m, ComputeLoss(m), nt_max=64, use_graph=False)
    x, t = synth_images(2, 96, 128, seed="dp/img0"), synth_labels(2, 4, seed="dp/lab0")
    g1 = None
    for i in range(4):
        st.step(x, t)
        if i == 0: g1 = m.flat_grads.clone()
    return g1, (m.flat_params - p0).numpy()
with emulated():
    t0 = time.time()
    base_g, base_d = run()
    print("one run %.1fs" % (time.time() - t0), flush=True)
    for k in range(N):
        g, d = run()
        gerr = float((g - base_g).abs().max() / base_g.abs().max())
        rel = float(np.linalg.norm(d - base_d) / np.linalg.norm(base_d))
        nflip = int((np.abs(d - base_d) > 1e-4).sum())
        print(f"pair {k}: first-step gradient diff {gerr:.3e}  update diff after 4 Adam steps {rel:.3e}  elements moved > 1e-4: {nflip}", flush=True)
