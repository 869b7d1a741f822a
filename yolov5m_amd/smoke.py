"""smoke(): tiny end-to-end invocation of the hot path, checked against the CPU oracle.
(The oracle is imported here as the CHECKER only -- allowed for smoke(), tests and bench cpu_baseline.)"""
import types

import numpy as np
import torch


def run():
    from oracle import loss_ref
    from .utils.synth import synth_state_dict, synth_labels, uniform
    from .utils.plot_utils import cells_to_bboxes
    from .utils.bboxes_utils import nms_batched
    from .ultralytics_loss import ComputeLoss

    dev = "cuda:0"
    anchors = synth_state_dict()["head.anchors"]

    class Stub:
        head = types.SimpleNamespace(nc=80, nl=3, naxs=3, anchors=anchors.to(dev), stride=[8, 16, 32])

        def parameters(self):
            return iter([torch.nn.Parameter(torch.zeros(1, device=dev))])

    shapes = [(8, 8), (4, 4), (2, 2)]
    B = 2
    p = [uniform(f"smoke/{i}", (B, 3, ny, nx, 85), -3.0, 3.0) for i, (ny, nx) in enumerate(shapes)]
    t = synth_labels(B, 6, seed="smoke")
    # loss fwd+bwd
    pg = [x.to(dev).requires_grad_(True) for x in p]
    loss = ComputeLoss(Stub())(pg, t, None)
    loss.backward()
    pc = [x.clone().requires_grad_(True) for x in p]
    lref, _ = loss_ref.compute_loss_ultra(pc, t, anchors)
    lref.backward()
    assert abs(float(loss) - float(lref)) <= 1e-4 * abs(float(lref)), (float(loss), float(lref))
    for a, b in zip(pg, pc):
        assert np.abs(a.grad.cpu().numpy() - b.grad.numpy()).max() <= 1e-4 * np.abs(b.grad.numpy()).max() + 1e-9
    # decode + NMS
    dec = cells_to_bboxes([x.to(dev) for x in p], anchors.to(dev), [8, 16, 32], is_pred=True, to_list=False)
    rows, idx, cnt = nms_batched(dec, 0.6, 0.01, 300)
    ref = loss_ref.non_max_suppression(dec.cpu(), 0.6, 0.01, 300)
    for b in range(B):
        assert int(cnt[b]) == len(ref[b][1])
        assert np.array_equal(idx[b, :int(cnt[b])].cpu().numpy(), ref[b][1])
    print(f"smoke ok: loss={float(loss):.6f} (oracle {float(lref):.6f}), nms kept={cnt.tolist()}")
