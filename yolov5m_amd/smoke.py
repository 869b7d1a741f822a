"""smoke(): tiny end-to-end invocation of the hot path on cuda:0, checked against the CPU oracle.
(The oracle is imported here as the CHECKER only -- allowed for smoke(), tests and bench cpu_baseline.)"""
import numpy as np
import torch


def run():
    from oracle import loss_ref, model_ref
    from . import config
    from .model import YOLOV5m
    from .ultralytics_loss import ComputeLoss
    from .utils.synth import synth_state_dict, synth_images, synth_labels
    from .utils.plot_utils import cells_to_bboxes
    from .utils.bboxes_utils import nms_batched
    from .utils.training_utils import NativeTrainStep

    dev = "cuda:0"
    sd = synth_state_dict()
    B, H, W = 2, 64, 96
    x = synth_images(B, H, W)
    t = synth_labels(B, 4, seed="smoke")

    # ---- one train step through the module API: forward + ComputeLoss + backward (f32 parity mode)
    m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    m.compute_dtype = "f32"
    m.train()
    out = m(x.to(dev))
    loss = ComputeLoss(m)(out, t, None)
    loss.backward()
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and "anchors" not in k else v)
           for k, v in sd.items()}
    oref = model_ref.forward(sdr, x, training=True)
    lref, _ = loss_ref.compute_loss_ultra(oref, t, sd["head.anchors"])
    lref.backward()
    assert abs(float(loss.detach()) - float(lref.detach())) <= 1e-3 * abs(float(lref.detach())), (float(loss.detach()), float(lref.detach()))
    gk = "backbone.4.c_out.cbl.0.weight"
    got = dict(m.named_parameters())[gk].grad.cpu().numpy()
    ref = sdr[gk].grad.numpy()
    assert np.abs(got - ref).max() <= 2e-2 * np.abs(ref).max(), "weight gradient mismatch"

    # ---- the fused native step in bf16 (what bench.py times)
    m2 = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
    m2.load_state_dict(sd, strict=True)
    m2 = m2.to(dev)
    step = NativeTrainStep(m2, ComputeLoss(m2), nt_max=64)
    l0 = float(step.step(x.to(dev), t.to(dev))[0])
    l1 = float(step.step(x.to(dev), t.to(dev))[0])
    assert np.isfinite(l0) and np.isfinite(l1)

    # ---- detect path: eval forward -> decode -> NMS, index sets vs the oracle on the same boxes
    m.eval()
    with torch.no_grad():
        o = m(x.to(dev))
    dec = cells_to_bboxes(o, m.head.anchors, m.head.stride, is_pred=True, to_list=False)
    rows, idx, cnt = nms_batched(dec, 0.6, 0.01, 300)
    ref = loss_ref.non_max_suppression(dec.cpu(), 0.6, 0.01, 300)
    for b in range(B):
        assert int(cnt[b]) == len(ref[b][1])
        assert np.array_equal(idx[b, :int(cnt[b])].cpu().numpy(), ref[b][1])
    print(f"smoke ok: loss={float(loss.detach()):.5f} (oracle {float(lref.detach()):.5f}); native bf16 step loss {l0:.4f}->{l1:.4f}; "
          f"nms kept={cnt.tolist()}")
