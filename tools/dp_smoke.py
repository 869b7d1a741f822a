#!/usr/bin/env python
"""single-rank RCCL smoke of the data-parallel step: real nccl all_reduces of the three gradient buckets, issued from the
communication stream between the captured per-segment graphs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from yolov5m_amd import config, parallel
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.ultralytics_loss import ComputeLoss
from yolov5m_amd.utils.training_utils import NativeTrainStep
from yolov5m_amd.utils.synth import synth_images, synth_labels
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768)).to("cuda"); m.compute_dtype = "bf16"; m.train()
m.flatten_parameters(); parallel.broadcast_parameters(m)
# the bench's own exchange object, forced on with one rank: three buckets issued from the communication stream between
# the captured per-segment graphs (NativeTrainStep._step_overlapped), over a real RCCL communicator
hook = parallel.GradAllReduce(1, timing=True, force=True)
calls = hook._evs
step = NativeTrainStep(m, ComputeLoss(m), nt_max=64, use_graph=True, grad_hook=hook, overlap=True)
x = synth_images(4, 320, 320).to("cuda"); t = synth_labels(4, 8).to("cuda")
losses = [float(step.step(x, t)[0]) for _ in range(6)]
torch.cuda.synchronize()
st = hook.stats()
print("losses", [round(v, 3) for v in losses], "exchange", st, "graph", step._opt_graph is not None)
ent = step._fb_graphs[next(iter(step._fb_graphs))]
assert all(v == v for v in losses) and losses[-1] < losses[0] and step._opt_graph is not None
assert ent[2] == "segments" and len(ent[1]) == 3 and len(st["buckets"]) == 3, (ent[2], st)
dist.destroy_process_group()
print("dp smoke ok")
