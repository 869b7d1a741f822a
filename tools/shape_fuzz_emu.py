#!/usr/bin/env python
"""Developer tool (no GPU): the product's engine + kernel sources on the CPU lane-level executor (tests/emu) at UNUSUAL input shapes --
odd batches, non-square images, the smallest grids -- against the torch-CPU oracle (oracle/model_ref.py, loss_ref.py), f32 mode:
eval logits (1e-4 of the largest logit), train-mode first-step ComputeLoss and YOLO_LOSS values (2e-4), and the whole flat gradient of the
fused step (5e-3 of its largest element: train-mode BatchNorm over a handful of samples amplifies f32 round-off; the reference's own f32
path sits 2e-5 .. 2e-3 from its fp64 evaluation at such sizes, tests/test_gpu_model.py). Index arithmetic, tile / tap / halo edges,
dispatch thresholds and slab logic are what this exercises; the GPU suite covers 64x64 .. 640x640 squares and 96x128.
usage: python tools/shape_fuzz_emu.py [BxHxW ...]      (default: 3x32x64 2x96x160 5x64x64 1x128x96 2x160x64)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from emu.harness import emulated  # noqa: E402
from oracle import loss_ref, model_ref  # noqa: E402
from yolov5m_amd import config  # noqa: E402
from yolov5m_amd.loss import YOLO_LOSS  # noqa: E402
from yolov5m_amd.model import YOLOV5m  # noqa: E402
from yolov5m_amd.ultralytics_loss import ComputeLoss  # noqa: E402
from yolov5m_amd.utils.synth import synth_images, synth_labels, synth_state_dict  # noqa: E402
from yolov5m_amd.utils.training_utils import NativeTrainStep  # noqa: E402


def model():
    m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
    m.load_state_dict(synth_state_dict(), strict=True)
    m.compute_dtype = "f32"
    return m


def main():
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(3, 32, 64), (2, 96, 160), (5, 64, 64), (1, 128, 96), (2, 160, 64)]
    bad = 0
    with emulated():
        for (B, H, W) in shapes:
            x = synth_images(B, H, W, seed=f"fuzz/{B}x{H}x{W}")
            t = synth_labels(B, 3, seed=f"fuzz/lab/{B}x{H}x{W}")
            sd = synth_state_dict()
            # eval logits
            m = model(); m.eval()
            with torch.no_grad():
                o = m(x)
                r = model_ref.forward(sd, x, training=False)
            e_eval = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(o, r))
            # train step, ComputeLoss: loss + every gradient
            params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k and "anchors" not in k}
            full = dict(sd); full.update(params)
            out = model_ref.forward(full, x, training=True)
            l, _ = loss_ref.compute_loss_ultra(out, t, sd["head.anchors"])
            l.backward()
            m = model(); m.train()
            st = NativeTrainStep(m, ComputeLoss(m), lr=0.0, nt_max=max(8, 3 * B))
            lo = st.step(x, t)
            gref = torch.cat([params[k].grad.reshape(-1) for k, _ in m.named_parameters()])
            g = m.flat_grads
            e_loss = abs(float(lo[0]) - float(l)) / abs(float(l))
            e_grad = float((g - gref).abs().max() / gref.abs().max())
            # YOLO_LOSS through the fused step against the oracle's restatement of loss.py on the oracle's train-mode logits
            tn = t.numpy().astype(np.float64)
            per = tuple(tn[tn[:, 0] == b][:, 1:] for b in range(B))
            m2 = model(); m2.train()
            st2 = NativeTrainStep(m2, YOLO_LOSS(m2, rect_training=False), lr=0.0, nt_max=max(8, 3 * B))
            lo2 = st2.step(x, per)
            yl = loss_ref.YoloLossRef(sd["head.anchors"]) if hasattr(loss_ref, "YoloLossRef") else None
            e_yolo = None
            if yl is not None:
                ly = yl([o_.detach() for o_ in out], per)
                e_yolo = abs(float(lo2[0]) - float(ly)) / abs(float(ly))
            ok = e_eval <= 1e-4 and e_loss <= 2e-4 and e_grad <= 5e-3 and (e_yolo is None or e_yolo <= 2e-4)
            bad += not ok
            print(f"{B}x{H}x{W}: eval logits {e_eval:.1e}  ComputeLoss {float(lo[0]):.5f} vs {float(l):.5f} ({e_loss:.1e})  gradient {e_grad:.1e}"
                  f"  YOLO_LOSS {float(lo2[0]):.5f}" + (f" ({e_yolo:.1e})" if e_yolo is not None else "") + ("" if ok else "   <-- OUT OF BOUNDS"), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
