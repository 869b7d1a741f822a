#!/usr/bin/env python
"""print the loss of the first N native train steps (B=64 @ 640, bf16): divergence / NaN hunting"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5m_amd import config
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.ultralytics_loss import ComputeLoss
from yolov5m_amd.utils.training_utils import NativeTrainStep
from yolov5m_amd.utils.synth import synth_images, synth_labels
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = int(sys.argv[3]) if len(sys.argv) > 3 else 640
dev = "cuda"
torch.manual_seed(0)
model = YOLOV5m(first_out=config.FIRST_OUT, nc=80, anchors=config.ANCHORS,
                ch=(config.FIRST_OUT * 4, config.FIRST_OUT * 8, config.FIRST_OUT * 16)).to(dev)
model.compute_dtype = os.environ.get("DT", "bf16")
model.train()
model.flatten_parameters()
step = NativeTrainStep(model, ComputeLoss(model), nt_max=B * 8, use_graph=os.environ.get("GRAPH", "1") == "1")
images = synth_images(B, S, S, seed="img/rank0").to(dev)
targets = synth_labels(B, 8, seed="lab/rank0").to(dev)
out = []
for i in range(N):
    lo = step.step(images, targets)
    torch.cuda.synchronize()
    out.append([round(float(v), 4) for v in lo])
print(out)
