"""GPU parity: HIP decode / NMS / IoU / build_targets / ComputeLoss (through the C ABI) against the
CPU oracle and the golden fixtures generated from the real reference."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import loss_ref
from yolov5m_amd.utils.synth import synth_state_dict, synth_labels, uniform

DEV = "cuda"


@pytest.fixture(scope="module")
def anchors():
    return synth_state_dict()["head.anchors"]


class _StubModel:
    """what ComputeLoss reads from a model (reference ultralytics_loss.py:22,31-41)."""

    def __init__(self, anchors, nc=80):
        self.head = types.SimpleNamespace(nc=nc, nl=3, naxs=3, anchors=anchors.to(DEV), stride=[8, 16, 32])
        self._p = torch.nn.Parameter(torch.zeros(1, device=DEV))

    def parameters(self):
        return iter([self._p])


# ---------------------------------------------------------------------------------------------- decode
def test_decode_golden(golden, anchors):
    from yolov5m_amd.utils.plot_utils import cells_to_bboxes
    g = golden("g6_decode_nms")
    p = [torch.from_numpy(g[f"dec/p{i}"]).to(DEV) for i in range(3)]
    out = cells_to_bboxes(p, anchors.to(DEV), [8, 16, 32], is_pred=True, to_list=False).cpu().numpy()
    ref = g["dec/out"]
    assert np.array_equal(out[..., 0], ref[..., 0])            # class index exact (incl. saturated ties)
    np.testing.assert_allclose(out[..., 1:], ref[..., 1:], rtol=1e-4, atol=1e-6)   # SURVEY B.4 tolerance
    t = [torch.from_numpy(g[f"dect/t{i}"]).to(DEV) for i in range(3)]
    out = cells_to_bboxes(t, anchors.to(DEV), [8, 16, 32], is_pred=False, to_list=False).cpu().numpy()
    assert np.array_equal(out, g["dect/out"])


def test_decode_vs_oracle_640(anchors):
    from yolov5m_amd.utils.plot_utils import cells_to_bboxes
    shapes = [(80, 80), (40, 40), (20, 20)]
    p = [uniform(f"dec640/{i}", (3, 3, ny, nx, 85), -8.0, 8.0) for i, (ny, nx) in enumerate(shapes)]
    ref = loss_ref.cells_to_bboxes(p, anchors, [8, 16, 32], is_pred=True).numpy()
    out = cells_to_bboxes([t.to(DEV) for t in p], anchors.to(DEV), [8, 16, 32], is_pred=True, to_list=False)
    out = out.cpu().numpy()
    assert out.shape == (3, 25200, 6)
    assert (out[..., 0] != ref[..., 0]).mean() < 1e-4        # argmax may flip only on near-ties
    np.testing.assert_allclose(out[..., 1:], ref[..., 1:], rtol=1e-4, atol=1e-5)


def test_decode_sigmoid_ties_and_ragged_tail(anchors):
    """Class logits that saturate sigmoid in fp32 (several classes at exactly 1.0 or 0.0): the first
    maximum in SIGMOID space must win, as torch.argmax does (plot_utils.py:27). B=1 leaves a ragged last
    tile on every scale (19200, 4800, 1200 cells are not all multiples of 64), served by the generic kernel."""
    from yolov5m_amd.utils.plot_utils import cells_to_bboxes
    shapes = [(80, 80), (40, 40), (20, 20)]
    p = [uniform(f"dectie/{i}", (1, 3, ny, nx, 85), -4.0, 4.0) for i, (ny, nx) in enumerate(shapes)]
    rng = np.random.default_rng(7)
    for t in p:
        flat = t.view(-1, 85)
        n = flat.shape[0]
        rows = torch.from_numpy(rng.permutation(n)[: n // 2])
        for k, r in enumerate(rows.tolist()):
            cls = np.sort(rng.permutation(80)[:3]) + 5
            if k % 3 == 0:      # three saturated classes, the LAST holds the largest logit
                flat[r, cls] = torch.tensor([20.0, 30.0, 40.0])
            elif k % 3 == 1:    # every class saturates to 0.0
                flat[r, 5:] = -120.0 - torch.arange(80, dtype=torch.float32).flip(0)
            else:               # exact duplicates of the maximum logit
                flat[r, cls] = 9.5
    ref = loss_ref.cells_to_bboxes(p, anchors, [8, 16, 32], is_pred=True).numpy()
    out = cells_to_bboxes([t.to(DEV) for t in p], anchors.to(DEV), [8, 16, 32], is_pred=True, to_list=False)
    out = out.cpu().numpy()
    # saturated / duplicated rows have ties only by construction: they must match exactly
    sig = 1.0 / (1.0 + np.exp(-np.concatenate([t.view(-1, 85)[:, 5:].numpy() for t in p]).astype(np.float64)))
    top2 = np.sort(sig, axis=1)[:, -2:]
    tie_rows = (top2[:, 1] - top2[:, 0]) < 1e-6
    constructed = np.concatenate([(t.view(-1, 85)[:, 5:].abs().max(1).values > 9.0).numpy() for t in p])
    assert constructed.sum() > 1000
    assert np.array_equal(out[0, constructed, 0], ref[0, constructed, 0])
    assert np.array_equal(out[0, ~tie_rows, 0], ref[0, ~tie_rows, 0])
    np.testing.assert_allclose(out[..., 1:], ref[..., 1:], rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------------------------------------- NMS
def _check_nms(bx, thr, iou, max_det=300):
    from yolov5m_amd.utils.bboxes_utils import nms_batched, non_max_suppression
    ref = loss_ref.non_max_suppression(bx, iou, thr, max_det)
    rows, idx, cnt = nms_batched(torch.from_numpy(bx).to(DEV), iou, thr, max_det)
    rows, idx, cnt = rows.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()
    for b in range(bx.shape[0]):
        rr, ri = ref[b]
        assert cnt[b] == len(ri), (b, cnt[b], len(ri))
        assert np.array_equal(idx[b, :cnt[b]], ri), f"image {b}: kept index set differs"
        assert np.array_equal(rows[b, :cnt[b]].view(np.uint32), rr.view(np.uint32)), f"image {b}: rows differ"
    return ref


def test_nms_golden_bit_exact(golden):
    g = golden("g6_decode_nms")
    for name in g["nms/names"]:
        bx = g[f"nms/{name}/in"]
        thr, iou = g[f"nms/{name}/thr"]
        if bx.shape[1] == 0:
            continue
        ref = _check_nms(bx, float(thr), float(iou))
        for b in range(bx.shape[0]):     # and the reference's own output rows
            assert np.array_equal(ref[b][0].view(np.uint32), g[f"nms/{name}/out{b}"].view(np.uint32))


def test_nms_api_shapes(golden):
    from yolov5m_amd.utils.bboxes_utils import non_max_suppression
    g = golden("g6_decode_nms")
    bx = g["nms/n1000_t0.25_i0.45/in"]
    lst = non_max_suppression(torch.from_numpy(bx).to(DEV), 0.45, 0.25, 300, tolist=True)
    assert isinstance(lst, list) and len(lst) == 2 and all(len(r) == 6 for r in lst[0])
    assert np.array_equal(np.array(lst[0], np.float32), g["nms/n1000_t0.25_i0.45/out0"])
    cat = non_max_suppression(torch.from_numpy(bx).to(DEV), 0.45, 0.25, 300, tolist=False)
    assert cat.shape == (len(lst[0]) + len(lst[1]), 6)


def test_nms_kernel_pinned_by_reference_aladdin(golden):
    """nms_kernel against the kept lists the REAL reference's `non_max_suppression_aladdin` produced (g14: tie-free,
    single class, IoUs clear of the threshold -- where it and torchvision.ops.nms define the same set; see
    tests/test_oracle_golden.py::test_nms_restatement_pinned_by_reference_aladdin): index sets bit-exact, rows = the
    reference wrapper's corner conversion of the kept boxes."""
    from yolov5m_amd.utils.bboxes_utils import nms_batched
    g = golden("g14_nms_crosspin")
    for name in g["names"].tolist():
        thr, iou = g[f"{name}/par"].tolist()
        mid, corners, keep = g[f"{name}/mid"], g[f"{name}/corners"], g[f"{name}/keep"]
        rows, idx, cnt = nms_batched(torch.from_numpy(mid[None]).to(DEV), float(iou), float(thr), 1024)
        k = int(cnt[0])
        keep = keep[:1024]               # (the kernel's max_detections bound: the wrapper keeps the first max_detections, :202-203)
        assert k == len(keep) and idx[0, :k].cpu().tolist() == keep.tolist(), name
        assert np.array_equal(rows[0, :k, 2:].cpu().numpy(), corners[keep]), name


@pytest.mark.parametrize("N,thr,iou", [(25200, 0.01, 0.6), (25200, 0.25, 0.45), (100800, 0.01, 0.6)])
def test_nms_large_random(N, thr, iou):
    """> NMS_CAP candidates: exercises the radix-select rounds."""
    rng = np.random.default_rng(N)
    B = 2
    bx = np.zeros((B, N, 6), np.float32)
    bx[..., 0] = rng.integers(0, 80, (B, N))
    bx[..., 1] = rng.uniform(0, 1, (B, N))
    bx[..., 2:4] = rng.uniform(0, 1280, (B, N, 2))
    bx[..., 4:6] = rng.uniform(8, 120, (B, N, 2))
    _check_nms(bx, thr, iou)


def test_nms_heavy_overlap_many_rounds():
    """few survivors among many candidates: every round of the select/sort/greedy loop runs."""
    rng = np.random.default_rng(5)
    N = 20000
    bx = np.zeros((1, N, 6), np.float32)
    centers = rng.uniform(100, 500, (8, 2))
    cid = rng.integers(0, 8, N)
    bx[0, :, 0] = 0
    bx[0, :, 1] = rng.uniform(0.3, 1, N)
    bx[0, :, 2:4] = centers[cid] + rng.normal(0, 2, (N, 2))
    bx[0, :, 4:6] = rng.uniform(70, 80, (N, 2))
    _check_nms(bx, 0.25, 0.45)


def test_nms_edge_cases():
    rng = np.random.default_rng(8)
    # nothing passes / one passes / all identical boxes with tied scores (stable: lowest index first)
    bx = np.zeros((3, 50, 6), np.float32)
    bx[..., 4:6] = 10
    bx[1, 7, 1] = 0.9
    bx[2, :, 1] = 0.5
    bx[2, :, 2:4] = 100
    _check_nms(bx, 0.25, 0.45)
    # tied scores on distinct boxes + max_det truncation
    bx = np.zeros((1, 700, 6), np.float32)
    bx[0, :, 1] = np.repeat(rng.uniform(0.3, 1, 70), 10)
    bx[0, :, 2:4] = rng.uniform(0, 5000, (700, 2))
    bx[0, :, 4:6] = 5
    _check_nms(bx, 0.25, 0.45, max_det=300)
    _check_nms(bx, 0.25, 0.45, max_det=17)


# ---------------------------------------------------------------------------------------------- IoU
def test_iou_golden(golden):
    from yolov5m_amd.utils.bboxes_utils import intersection_over_union
    g = golden("g1_giou")
    a, b = torch.from_numpy(g["a"]).to(DEV), torch.from_numpy(g["b"]).to(DEV)
    for flag, key in ((False, "iou"), (True, "giou")):
        out = intersection_over_union(a, b, GIoU=flag).cpu().numpy()
        assert out.shape == g[key].shape
        np.testing.assert_allclose(out, g[key], rtol=1e-5, atol=1e-6)


def test_iou_backward_vs_autograd():
    from yolov5m_amd.utils.bboxes_utils import intersection_over_union
    rng = np.random.default_rng(3)
    a = torch.from_numpy(rng.uniform(0.5, 6, (500, 4)).astype(np.float32))
    b = torch.from_numpy(rng.uniform(0.5, 6, (500, 4)).astype(np.float32))
    w = torch.from_numpy(rng.uniform(-1, 1, (500, 1)).astype(np.float32))
    ac, bc = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    (loss_ref.giou(ac, bc, GIoU=True) * w).sum().backward()
    ag, bg = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    (intersection_over_union(ag, bg, GIoU=True) * w.to(DEV)).sum().backward()
    np.testing.assert_allclose(ag.grad.cpu().numpy(), ac.grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(bg.grad.cpu().numpy(), bc.grad.numpy(), rtol=1e-4, atol=1e-6)


# ---------------------------------------------------------------------------------------------- targets
def test_build_targets_golden_bit_exact(golden, anchors):
    from yolov5m_amd.ultralytics_loss import ComputeLoss
    g = golden("g2_build_targets")
    lf = ComputeLoss(_StubModel(anchors))
    for name in g["names"]:
        shapes = [tuple(s) for s in g[f"{name}/shapes"]]
        p = [torch.zeros(4, 3, ny, nx, 85, device=DEV) for (ny, nx) in shapes]
        tcls, tbox, indices, anch = lf.build_targets(p, torch.from_numpy(g[f"{name}/targets"]))
        for i in range(3):
            b, a, gj, gi = [t.cpu().numpy() for t in indices[i]]
            assert np.array_equal(b, g[f"{name}/{i}/b"]), (name, i)
            assert np.array_equal(a, g[f"{name}/{i}/a"]), (name, i)
            assert np.array_equal(gj, g[f"{name}/{i}/gj"]), (name, i)
            assert np.array_equal(gi, g[f"{name}/{i}/gi"]), (name, i)
            assert np.array_equal(tcls[i].cpu().numpy(), g[f"{name}/{i}/tcls"]), (name, i)
            assert np.array_equal(tbox[i].cpu().numpy().view(np.uint32), g[f"{name}/{i}/tbox"].view(np.uint32)), (name, i)
            assert np.array_equal(anch[i].cpu().numpy().view(np.uint32), g[f"{name}/{i}/anch"].view(np.uint32)), (name, i)


def test_build_targets_large_vs_oracle(anchors):
    from yolov5m_amd.ultralytics_loss import ComputeLoss
    lf = ComputeLoss(_StubModel(anchors))
    t = synth_labels(64, 8, seed="bt_large")          # nt = 512 (configs[2])
    shapes = [(80, 80), (40, 40), (20, 20)]
    p = [torch.zeros(64, 3, ny, nx, 85, device=DEV) for (ny, nx) in shapes]
    tcls, tbox, indices, anch = lf.build_targets(p, t)
    ref = loss_ref.build_targets_ultra(shapes, t.numpy(), anchors.numpy())
    for i in range(3):
        for k, v in zip(("b", "a", "gj", "gi"), indices[i]):
            assert np.array_equal(v.cpu().numpy(), ref[i][k])
        assert np.array_equal(tbox[i].cpu().numpy().view(np.uint32), ref[i]["tbox"].view(np.uint32))
        assert np.array_equal(tcls[i].cpu().numpy(), ref[i]["tcls"])


# ---------------------------------------------------------------------------------------------- loss
def test_compute_loss_golden(golden, anchors):
    from yolov5m_amd.ultralytics_loss import ComputeLoss
    g = golden("g3_compute_loss")
    lf = ComputeLoss(_StubModel(anchors))
    for name in g["names"]:
        shapes = [tuple(s) for s in g[f"{name}/shapes"]]
        B = int(g[f"{name}/B"])
        if name == "b4_640":
            p = [uniform(f"g3/{name}/{i}", (B, 3, ny, nx, 85), -3.0, 3.0) for i, (ny, nx) in enumerate(shapes)]
        else:
            p = [torch.from_numpy(g[f"{name}/p{i}"]) for i in range(3)]
        p = [t.to(DEV).requires_grad_(True) for t in p]
        loss = lf(p, torch.from_numpy(g[f"{name}/targets"]), None)
        assert loss.shape == (1,)
        loss.backward()
        np.testing.assert_allclose(loss.detach().cpu().numpy(), g[f"{name}/loss"], rtol=1e-4)   # north_star tol
        for i in range(3):
            got = p[i].grad.cpu().numpy()
            if name == "b4_640":
                np.testing.assert_allclose(got.astype(np.float64).sum(), g[f"{name}/g{i}_sum"], rtol=1e-3)
                np.testing.assert_allclose(np.abs(got.astype(np.float64)).sum(), g[f"{name}/g{i}_abs"], rtol=1e-4)
                np.testing.assert_allclose(got[..., 4].reshape(-1)[::97], g[f"{name}/g{i}_obj_sample"], rtol=1e-4, atol=1e-9)
            else:
                ref = g[f"{name}/g{i}"]
                assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-9, (name, i)


def test_compute_loss_b64_vs_oracle(anchors):
    """configs[2] label recipe (nt=512) at reduced grid so the CPU oracle finishes in seconds."""
    from yolov5m_amd.ultralytics_loss import ComputeLoss
    lf = ComputeLoss(_StubModel(anchors))
    B = 64
    shapes = [(20, 20), (10, 10), (5, 5)]
    t = synth_labels(B, 8, seed="loss_b64")
    p = [uniform(f"lb64/{i}", (B, 3, ny, nx, 85), -4.0, 4.0) for i, (ny, nx) in enumerate(shapes)]
    pc = [x.clone().requires_grad_(True) for x in p]
    lref, _ = loss_ref.compute_loss_ultra(pc, t, anchors)
    lref.backward()
    pg = [x.to(DEV).requires_grad_(True) for x in p]
    l = lf(pg, t, None)
    l.backward()
    np.testing.assert_allclose(l.detach().cpu().numpy(), lref.detach().numpy(), rtol=1e-4)
    for i in range(3):
        ref = pc[i].grad.numpy()
        got = pg[i].grad.cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-9


# ---------------------------------------------------------------------------------------------- YOLO_LOSS
def test_yolo_loss_golden_pinned_sequence(golden, anchors):
    """reference loss.py on a FRESH object with a pinned call sequence (anchor decay state)"""
    from yolov5m_amd.loss import YOLO_LOSS
    g = golden("g4_yolo_loss")
    B = int(g["B"])
    lf = YOLO_LOSS(_StubModel(anchors), rect_training=False)
    # dense targets of the first image of a fresh object: exact
    fresh = YOLO_LOSS(_StubModel(anchors), rect_training=False)
    shapes = [tuple(s) for s in g["shapes"]]
    p0 = [torch.zeros(1, 3, ny, nx, 85, device=DEV) for (ny, nx) in shapes]
    tg = fresh.build_targets(p0, g["0/boxes0"], (64, 64))
    for i in range(3):
        assert np.array_equal(tg[i].numpy(), g[f"bt/t{i}"])
    for call in range(2):
        assert np.array_equal(lf.anchors.numpy(), g[f"{call}/anchors_before"])
        p = [torch.from_numpy(g[f"{call}/p{i}"]).to(DEV).requires_grad_(True) for i in range(3)]
        boxes = tuple(g[f"{call}/boxes{b}"] for b in range(B))
        loss = lf(p, boxes, pred_size=(64, 64))
        loss.backward()
        np.testing.assert_allclose(float(loss.detach()), float(g[f"{call}/loss"]), rtol=1e-4)
        assert np.array_equal(lf.anchors.numpy(), g[f"{call}/anchors_after"])
        for i in range(3):
            ref = g[f"{call}/g{i}"]
            got = p[i].grad.cpu().numpy()
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-9, (call, i)


def test_yolo_build_targets_native_batch_golden(golden):
    """y5m_yolo_build_targets (one launch per batch) against the REAL reference's per-image Python loop: two calls of 16
    images x 8 boxes on one loss object -- 256 boxes walk the in-place anchor decay through real anchors, denormals and
    exact zeros (nine-way ties in the anchor ranking), boxes sharing a cell, ignore cells. Dense targets (all three
    scales, every float) and the anchor state after each call: bit-exact. Also the empty image and the all-empty batch."""
    from yolov5m_amd.loss import YOLO_LOSS
    g = golden("g12_yolo_build_targets")
    B = int(g["B"])
    shapes = [tuple(int(v) for v in s) for s in g["shapes"]]
    lf = YOLO_LOSS(_StubModel(torch.from_numpy(g["anchors0"])), rect_training=False)
    for call in range(2):
        boxes = [g[f"{call}/boxes{b}"] for b in range(B)]
        dense = lf._build_targets_native(shapes, boxes)
        for i in range(3):
            ref = np.zeros(tuple(dense[i].shape), np.float32)
            ref[tuple(g[f"{call}/nz{i}"].T)] = g[f"{call}/val{i}"]
            got = dense[i].cpu().numpy()
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (call, i, int((got != ref).sum()))
        assert np.array_equal(lf.anchors.numpy().view(np.uint32), g[f"{call}/anchors_after"].view(np.uint32))
    # images without boxes leave zeros and do not advance the anchor state
    before = lf.anchors.numpy().copy()
    dense = lf._build_targets_native(shapes, [np.zeros((0, 5)), g["0/boxes0"][:0]])
    assert all(float(d.abs().sum()) == 0.0 for d in dense)
    assert np.array_equal(lf.anchors.numpy(), before)
    # a fresh object, mixed empty / non-empty images == the oracle, image by image
    lf2 = YOLO_LOSS(_StubModel(torch.from_numpy(g["anchors0"])), rect_training=False)
    ref2 = loss_ref.YoloLossRef(g["anchors0"])
    boxes = [g["0/boxes0"], np.zeros((0, 5)), g["0/boxes1"][:3], np.zeros((0, 5)), g["0/boxes2"]]
    dense = lf2._build_targets_native(shapes, boxes)
    tg = [ref2.build_targets(shapes, b) for b in boxes]
    for i in range(3):
        assert np.array_equal(dense[i].cpu().numpy(), torch.stack([t[i] for t in tg], 0).numpy())
    assert np.array_equal(lf2.anchors.numpy(), ref2.anchors.numpy())


def test_yolo_loss_single_scale_and_nan(anchors):
    from yolov5m_amd.loss import YOLO_LOSS
    from oracle.loss_ref import YoloLossRef
    lf = YOLO_LOSS(_StubModel(anchors), rect_training=False)
    ref = YoloLossRef(anchors)
    shapes = [(8, 8), (4, 4), (2, 2)]
    boxes = (np.array([[3, 0.5, 0.5, 0.3, 0.4], [7, 0.2, 0.7, 0.1, 0.2]]), np.array([[1, 0.6, 0.4, 0.5, 0.5]]))
    p = [uniform(f"yl/{i}", (2, 3, ny, nx, 85), -3.0, 3.0) for i, (ny, nx) in enumerate(shapes)]
    tg = [ref.build_targets(shapes, b) for b in boxes]
    for i in range(3):
        dense = torch.stack([t[i] for t in tg], 0)
        lr = ref.compute_loss(p[i].clone(), dense.clone(), ref.anchors_d[i], loss_ref.BALANCE[i])
        lg, logs = lf.compute_loss(p[i].to(DEV), dense.clone().to(DEV), lf.anchors_d[i], lf.balance[i])
        np.testing.assert_allclose(float(lg), float(lr), rtol=1e-4)
    # no boxes at all -> NaN, as the reference (loss.py:212)
    l = lf([t.to(DEV) for t in p], (np.zeros((0, 5)), np.zeros((0, 5))), pred_size=(64, 64))
    assert bool(torch.isnan(l))


def test_nms_aladdin_golden_and_oracle(golden):
    """non_max_suppression_aladdin through the C ABI (y5m_nms_aladdin): the kept boxes are the SAME list objects
    in the same order as the real reference kept (golden), and as the C oracle keeps on larger lists (bit-exact
    index sets: integer / compare work)"""
    import numpy as np
    from oracle import cnative
    from yolov5m_amd.utils.bboxes_utils import non_max_suppression_aladdin
    g = golden("g9_nms_aladdin")
    for name in g["names"].tolist():
        thr, iou, mid, md = g[f"{name}/par"].tolist()
        lst = [[float(v) for v in row] for row in g[f"{name}/in"]]
        kept = non_max_suppression_aladdin(lst, iou, thr, box_format="midpoint" if mid else "corners", max_detections=int(md))
        pos = {id(r): i for i, r in enumerate(lst)}
        assert [pos[id(r)] for r in kept] == g[f"{name}/keep"].tolist(), name
    rng = np.random.default_rng(5)
    for N, md, fmt in ((25200, 300, "corners"), (6000, 1000, "midpoint"), (5000, 17, "corners")):
        bx = np.zeros((N, 6), np.float32)
        bx[:, 0] = rng.integers(0, 80, N)
        bx[:, 1] = rng.uniform(0, 1, N).astype(np.float32)
        c, wh = rng.uniform(0, 640, (N, 2)).astype(np.float32), rng.uniform(4, 200, (N, 2)).astype(np.float32)
        bx[:, 2:4], bx[:, 4:6] = (c, wh) if fmt == "midpoint" else (c - wh / 2, c + wh / 2)
        lst = bx.tolist()
        kept = non_max_suppression_aladdin(lst, 0.45, 0.25, box_format=fmt, max_detections=md)
        pos = {id(r): i for i, r in enumerate(lst)}
        assert [pos[id(r)] for r in kept] == cnative.nms_aladdin(bx, 0.45, 0.25, fmt, md).tolist(), (N, md, fmt)
    assert non_max_suppression_aladdin([], 0.5, 0.5) == []


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_sparse_head_gradient_path_equals_dense(anchors, dtype):
    """The native step's head-gradient path -- y5m_compute_loss_sparse (writes only the rows of the cells a target hit +
    a compact objectness-gradient plane) followed by y5m_head_grad_pack_sparse -- against y5m_compute_loss +
    y5m_head_grad_pack: same loss, packed rows bit-identical, bias gradient equal up to summation order. Targets include
    duplicates of one cell (their rows accumulate) and scales whose pixel count is not a multiple of 64."""
    import ctypes
    from yolov5m_amd import _lib
    from yolov5m_amd.ultralytics_loss import ComputeLoss, _Workspace
    L = _lib.lib()
    B, nt_max = 3, 64
    shapes = [(20, 12), (10, 6), (5, 3)]
    p = [uniform(f"shp/{i}", (B, 3, ny, nx, 85), -3.0, 3.0).to(DEV) for i, (ny, nx) in enumerate(shapes)]
    t = synth_labels(B, 6, seed="shp")
    t = torch.cat([t, t[:4]]).to(DEV)                          # duplicated targets: several rows on the same cells
    lf = ComputeLoss(_StubModel(anchors))
    st = _lib.stream_ptr()
    res = {}
    for mode, fn in (("dense", L.y5m_compute_loss), ("sparse", L.y5m_compute_loss_sparse)):
        ws = _Workspace(DEV, B, 3, shapes, nt_max)
        grads = [torch.full_like(q, 123.0) for q in p]          # the sparse variant leaves most of it untouched
        _lib.check(L.y5m_build_targets(_lib.ptr(t), t.shape[0], None, nt_max, _lib.ptr(lf.anchors), 3, ws.ny, ws.nx,
                                       float(lf.anchor_t), ws.tg, _lib.ptr(ws.bt_ws), ws.bt_ws_bytes, st), "bt")
        _lib.check(fn(_lib.ptr_array(p), _lib.ptr_array(grads), B, 3, ws.ny, ws.nx, lf.nc, ws.tg, nt_max,
                      _lib.float_array(lf.balance), float(lf.lambda_box), float(lf.lambda_obj), float(lf.lambda_class),
                      _lib.ptr(ws.loss_out), _lib.ptr(ws.loss_ws), ws.loss_ws_bytes, st), mode)
        own, gob = (ctypes.c_void_p * 3)(), (ctypes.c_void_p * 3)()
        _lib.check(L.y5m_compute_loss_owner_ptrs(_lib.ptr(ws.loss_ws), B, 3, ws.ny, ws.nx, nt_max, own, gob), "owner")
        tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
        code = _lib.BF16 if dtype == "bf16" else _lib.F32
        packed, bias = [], []
        for i, (ny, nx) in enumerate(shapes):
            d = torch.full((B * ny * nx, 256), 7.0, dtype=tdt, device=DEV)
            b = torch.zeros(255, device=DEV)
            if mode == "dense":
                _lib.check(L.y5m_head_grad_pack(_lib.ptr(grads[i]), B, 3, ny, nx, 85, _lib.ptr(d), 256, _lib.ptr(b), code, st), "dense")
                assert int((grads[i][..., 5:].abs().sum(-1) > 0).sum()) > 0      # there ARE target rows on this scale
            else:
                _lib.check(L.y5m_head_grad_pack_sparse(_lib.ptr(grads[i]), ctypes.c_void_p(own[i]), ctypes.c_void_p(gob[i]),
                                                       _lib.ptr(ws.bagg[i]), _lib.ptr(ws.count[i]), ws.cap, B, 3,
                                                       ny, nx, 85, _lib.ptr(d), 256, _lib.ptr(b), code, st), "sparse")
                assert float((grads[i] == 123.0).float().mean()) > 0.5           # untouched outside the target rows
            packed.append(d); bias.append(b)
        res[mode] = (ws.loss_out.clone(), packed, bias)
    assert torch.equal(res["dense"][0], res["sparse"][0])
    for i in range(3):
        d0, d1 = res["dense"][1][i].float(), res["sparse"][1][i].float()
        # identical arithmetic; only cells hit by SEVERAL target rows accumulate (f32 atomics) in a run-dependent order
        assert float((d0 == d1).float().mean()) > 0.999, i
        np.testing.assert_allclose(d1.cpu().numpy(), d0.cpu().numpy(), rtol=1e-2 if dtype == "bf16" else 1e-5, atol=1e-9)
        np.testing.assert_allclose(res["sparse"][2][i].cpu().numpy(), res["dense"][2][i].cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_sparse_head_gradient_pack16_subprocess():
    """Y5M_HEAD_PACK16=1 (default 0; round 5): the objectness rows of the sparse head-gradient pack written two rows per store
    instruction in 16-byte pieces (head_grad_pack_obj16_kernel) -- the sparse-vs-dense test above in a child with the knob on (pixel
    counts 720 / 180 / 45: whole waves, a partial wave, an odd last row)"""
    import os, subprocess, sys
    if os.environ.get("Y5M_HEAD_PACK16") == "1":
        pytest.skip("already the child")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-k", "sparse_head_gradient_path_equals_dense"],
                       env=dict(os.environ, Y5M_HEAD_PACK16="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("shape", [(2, 3, 5, 7), (1, 3, 20, 20), (3, 3, 1, 1)])
def test_class_obj_accuracy_counts_bit_exact(shape):
    """y5m_class_obj_accuracy against the reference's masked counting (utils/validation_utils.py:58-68) on random logits with
    TIED class maxima (torch.argmax takes the first), cell counts that are not multiples of a wave, no object at all in one
    case, and the objectness read from channel 0 as the reference does (:66)."""
    from yolov5m_amd import _lib
    B, A, ny, nx = shape
    g = torch.Generator().manual_seed(B * 100 + ny)
    out = torch.randn((B, A, ny, nx, 85), generator=g)
    out[..., 5:] = torch.round(out[..., 5:] * 2) / 2                      # many equal class logits: ties
    y = torch.zeros((B, A, ny, nx, 6))
    obj = torch.rand((B, A, ny, nx), generator=g) < (0.0 if shape[2] == 1 else 0.3)
    y[..., 4] = obj.float()
    y[..., 5] = torch.randint(0, 80, (B, A, ny, nx), generator=g).float()
    # half of the object cells get the label their argmax will produce
    am = torch.argmax(out[..., 5:], dim=-1).float()
    take = torch.rand((B, A, ny, nx), generator=g) < 0.5
    y[..., 5] = torch.where(take, am, y[..., 5])
    conf = 0.6
    m = y[..., 4] == 1
    want = [int(m.sum()), int((torch.argmax(out[..., 5:][m], dim=-1) == y[..., 5][m]).sum()),
            int(((torch.sigmoid(out[..., 0]) > conf)[m] == y[..., 4][m]).sum())]
    od, yd = out.to(DEV).contiguous(), y.to(DEV).contiguous()
    counts = torch.zeros(3, dtype=torch.int64, device=DEV)
    for _ in range(2):                                                    # accumulates: the second call doubles every counter
        _lib.check(_lib.lib().y5m_class_obj_accuracy(_lib.ptr(od), _lib.ptr(yd), B * A * ny * nx, 85, conf, _lib.ptr(counts),
                                                     _lib.stream_ptr()), "y5m_class_obj_accuracy")
    torch.cuda.synchronize()
    assert counts.cpu().tolist() == [2 * v for v in want], (counts.cpu().tolist(), want)


def test_dense_targets_builder_matches_dataset_algorithm(anchors):
    """utils/validation_utils.DenseTargets (the datasets' dense target builder, dataset.py:337-414 -- the same
    algorithm and the same in-place anchor decay as YOLO_LOSS.build_targets) for two consecutive batches against the oracle's
    restatement run image by image (oracle pinned to the real reference's loss.build_targets by g12): bit-exact, anchor
    state included; an image without boxes; non-square images."""
    from yolov5m_amd import config
    from yolov5m_amd.utils.validation_utils import DenseTargets
    dt = DenseTargets(config.ANCHORS, device=DEV)
    ref = loss_ref.YoloLossRef(anchors)
    assert torch.equal(dt.anchors, ref.anchors)
    rng = np.random.RandomState(5)
    for hw in ((96, 128), (64, 64)):
        labels = []
        for b in range(3):
            n = [4, 0, 7][b]
            lab = np.zeros((n, 5), np.float64)
            lab[:, 0] = rng.randint(0, 80, n)
            lab[:, 1:3] = rng.uniform(0.05, 0.95, (n, 2))
            lab[:, 3:5] = rng.uniform(0.02, 0.6, (n, 2))
            labels.append(lab)
        got = dt(labels, hw)
        shapes = [(hw[0] // s, hw[1] // s) for s in (8, 16, 32)]
        want = [ref.build_targets(shapes, lab) for lab in labels]
        for i in range(3):
            w = torch.stack([t[i] for t in want], 0)
            assert tuple(got[i].shape) == tuple(w.shape)
            assert torch.equal(got[i].cpu(), w), (hw, i, float((got[i].cpu() - w).abs().max()))
        assert torch.equal(dt.anchors, ref.anchors)
