#!/usr/bin/env python
"""Experiment: confine the forked weight-gradient stream to a subset of the CUs (hipExtStreamCreateWithCUMask) so that the
main stream's data gradients / BatchNorm passes and the weight gradients stop taking each other's CUs. A CU mask is a
property of a STREAM (an HSA queue), not of a captured kernel node, so this only exists in EAGER mode: every line below is
an eager step (NativeTrainStep(use_graph=False)) except the first, the graph-replayed default for reference.
usage: cu_mask_probe.py [B] [S]        masks come from MASKS below (8 x 32-bit words = 256 CUs)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5m_amd import config
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.ultralytics_loss import ComputeLoss
from yolov5m_amd.utils.training_utils import NativeTrainStep
from yolov5m_amd.utils.synth import synth_images, synth_labels

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 640
dev = "cuda:0"
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    r = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    assert r == 0, f"hipExtStreamCreateWithCUMask -> {r}"
    return torch.cuda.ExternalStream(st.value, device=dev)


def popcount(words):
    return sum(bin(w).count("1") for w in words)


def probe_width(stream, label):
    """how much of the chip does a stream see: a bandwidth-bound copy and an MFMA-bound matmul, on `stream`"""
    a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); y = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for _ in range(2):
            b.copy_(a); torch.mm(x, y)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(stream)
        for _ in range(5):
            b.copy_(a)
        e[1].record(stream)
        for _ in range(5):
            torch.mm(x, y)
        e[2].record(stream)
    torch.cuda.synchronize()
    print(f"  [{label}] copy {2 * a.numel() * 5 / e[0].elapsed_time(e[1]) / 1e9:6.2f} TB/s   mm {2 * 8192 ** 3 * 5 / e[1].elapsed_time(e[2]) / 1e9:7.1f} TFLOP/s", flush=True)


def run(label, use_graph, side_words=None, main_words=None):
    torch.manual_seed(0)
    model = YOLOV5m(first_out=config.FIRST_OUT, nc=80, anchors=config.ANCHORS,
                    ch=(config.FIRST_OUT * 4, config.FIRST_OUT * 8, config.FIRST_OUT * 16)).to(dev)
    model.compute_dtype = "bf16"
    model.train()
    model.flatten_parameters()
    step = NativeTrainStep(model, ComputeLoss(model), nt_max=B * 8, use_graph=use_graph)
    images = step.input_buffer(B, S, S)
    images.copy_(synth_images(B, S, S, seed="img/rank0").to(dev))
    targets = synth_labels(B, 8, seed="lab/rank0").to(dev)
    eng = step.load_inputs(images, targets)
    if side_words is not None:
        eng._side = masked_stream(side_words)
    main = masked_stream(main_words) if main_words is not None else torch.cuda.current_stream()
    with torch.cuda.stream(main):
        for _ in range(3):
            lo = step.step(images, targets)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            lo = step.step(images, targets)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{label:58s} {ms:7.3f} ms/step   loss {float(lo[0]):.4f}", flush=True)
    model._engines = {}
    del step, model, eng
    torch.cuda.empty_cache()


ALL = [0xFFFFFFFF] * 8
# Only masks that leave CUs on every XCD under BOTH plausible bit orders (bit i -> XCD i % 8, or bit i -> XCD i / 32): an XCD
# without CUs could leave workgroups undispatchable.
MASKS = {
    "0x000000FF x 8 (64 CUs)": [0x000000FF] * 8,
    "0x0000FFFF x 8 (128 CUs)": [0x0000FFFF] * 8,
    "0x00FFFFFF x 8 (192 CUs)": [0x00FFFFFF] * 8,
}
print("what a masked stream sees (alone on the chip):")
probe_width(torch.cuda.current_stream(), "unmasked")
for name, w in MASKS.items():
    probe_width(masked_stream(w), f"{name}: {popcount(w)} CUs")
print()
run("graph replay (default)", True)
run("eager, unmasked side stream", False)
for name, w in MASKS.items():
    run(f"eager, side stream on {name}", False, side_words=w)
    comp = [(~x) & 0xFFFFFFFF for x in w]
    run(f"eager, side on {name}, main on the complement", False, side_words=w, main_words=comp)
