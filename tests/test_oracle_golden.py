"""CPU: the oracle restatement (oracle/) against the golden vectors produced by the REAL reference
(tests/golden/make_golden.py). This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import cnative, loss_ref, model_ref
from yolov5m_amd.utils.synth import synth_state_dict, synth_images, uniform


def test_g1_giou_bit_exact(golden):
    g = golden("g1_giou")
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    assert np.array_equal(loss_ref.giou(a, b, GIoU=False).numpy(), g["iou"], equal_nan=True)
    assert np.array_equal(loss_ref.giou(a, b, GIoU=True).numpy(), g["giou"], equal_nan=True)


def test_g2_build_targets_bit_exact(golden):
    g = golden("g2_build_targets")
    anchors = synth_state_dict()["head.anchors"].numpy()
    for name in g["names"]:
        shapes = [tuple(s) for s in g[f"{name}/shapes"]]
        res = loss_ref.build_targets_ultra(shapes, g[f"{name}/targets"], anchors)
        for i in range(3):
            for k in ("b", "a", "gj", "gi", "tcls"):
                assert np.array_equal(res[i][k], g[f"{name}/{i}/{k}"]), (name, i, k)
            # fp32 bits
            assert np.array_equal(res[i]["tbox"].view(np.uint32), g[f"{name}/{i}/tbox"].view(np.uint32)), (name, i)
            assert np.array_equal(res[i]["anch"].view(np.uint32), g[f"{name}/{i}/anch"].view(np.uint32)), (name, i)


def test_g3_compute_loss(golden):
    g = golden("g3_compute_loss")
    anchors = synth_state_dict()["head.anchors"]
    for name in g["names"]:
        shapes = [tuple(s) for s in g[f"{name}/shapes"]]
        B = int(g[f"{name}/B"])
        if name == "b4_640":
            p = [uniform(f"g3/{name}/{i}", (B, 3, ny, nx, 85), -3.0, 3.0) for i, (ny, nx) in enumerate(shapes)]
        else:
            p = [torch.from_numpy(g[f"{name}/p{i}"]) for i in range(3)]
        p = [t.clone().requires_grad_(True) for t in p]
        loss, _ = loss_ref.compute_loss_ultra(p, g[f"{name}/targets"], anchors)
        loss.backward()
        np.testing.assert_allclose(loss.detach().numpy(), g[f"{name}/loss"], rtol=1e-6)
        for i in range(3):
            if name == "b4_640":
                gi = p[i].grad.numpy()
                np.testing.assert_allclose(gi.astype(np.float64).sum(), g[f"{name}/g{i}_sum"], rtol=1e-5)
                np.testing.assert_allclose(gi[..., 4].reshape(-1)[::97], g[f"{name}/g{i}_obj_sample"], rtol=1e-6)
            else:
                np.testing.assert_allclose(p[i].grad.numpy(), g[f"{name}/g{i}"], rtol=1e-5, atol=1e-9)


def test_g4_yolo_loss_pinned_sequence(golden):
    g = golden("g4_yolo_loss")
    anchors = synth_state_dict()["head.anchors"]
    shapes = [tuple(s) for s in g["shapes"]]
    B = int(g["B"])
    # dense targets, fresh object, first image: exact
    fresh = loss_ref.YoloLossRef(anchors)
    tg = fresh.build_targets(shapes, g["0/boxes0"])
    for i in range(3):
        assert np.array_equal(tg[i].numpy(), g[f"bt/t{i}"])
    lf = loss_ref.YoloLossRef(anchors)
    for call in range(2):
        assert np.array_equal(lf.anchors.numpy(), g[f"{call}/anchors_before"])
        p = [torch.from_numpy(g[f"{call}/p{i}"]).clone().requires_grad_(True) for i in range(3)]
        boxes = tuple(g[f"{call}/boxes{b}"] for b in range(B))
        loss = lf(p, boxes)
        loss.backward()
        np.testing.assert_allclose(loss.detach().numpy(), g[f"{call}/loss"], rtol=1e-6)
        assert np.array_equal(lf.anchors.numpy(), g[f"{call}/anchors_after"])
        for i in range(3):
            np.testing.assert_allclose(p[i].grad.numpy(), g[f"{call}/g{i}"], rtol=1e-5, atol=1e-9)


def _g12_dense(g, call, i):
    B = int(g["B"])
    ny, nx = [tuple(s) for s in g["shapes"]][i]
    d = np.zeros((B, 3, ny, nx, 6), np.float32)
    nz = g[f"{call}/nz{i}"]
    d[tuple(nz.T)] = g[f"{call}/val{i}"]
    return d


def test_g12_yolo_build_targets_batch_sequence(golden):
    """the oracle's YOLO_LOSS.build_targets against the real reference over 256 boxes in a row (the whole anchor decay:
    distinct IoUs, denormals, nine-way ties): dense targets and the anchor state, bit for bit"""
    g = golden("g12_yolo_build_targets")
    lf = loss_ref.YoloLossRef(g["anchors0"])
    shapes = [tuple(s) for s in g["shapes"]]
    B = int(g["B"])
    for call in range(2):
        tg = [lf.build_targets(shapes, g[f"{call}/boxes{b}"]) for b in range(B)]
        for i in range(3):
            got = torch.stack([t[i] for t in tg], 0).numpy()
            assert np.array_equal(got, _g12_dense(g, call, i)), (call, i)
        assert np.array_equal(lf.anchors.numpy(), g[f"{call}/anchors_after"])


@pytest.mark.parametrize("tag,shape", [("s64", (1, 64, 64)), ("s96x128", (2, 96, 128))])
def test_g5_model_forward(golden, tag, shape):
    g = golden("g5_model")
    sd = synth_state_dict()
    x = synth_images(*shape)
    for mode in ("eval", "train"):
        ns = {} if mode == "train" else None
        with torch.no_grad():
            o = model_ref.forward(sd, x, training=(mode == "train"), new_stats=ns)
        for i in range(3):
            flat = o[i].reshape(-1).numpy()
            step = int(g[f"{tag}/{mode}/o{i}_step"])
            np.testing.assert_allclose(flat[::step][:4096], g[f"{tag}/{mode}/o{i}_sample"], rtol=1e-4, atol=1e-5)
        if mode == "train":
            for k in ("backbone.0.cbl.1.running_mean", "backbone.0.cbl.1.running_var",
                      "neck.7.c_out.cbl.1.running_var"):
                np.testing.assert_allclose(ns[k].numpy(), g[f"{tag}/train/{k}"], rtol=1e-5, atol=1e-7)


def test_g5_train_step_grads(golden):
    g = golden("g5_model")
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and "anchors" not in k else v)
          for k, v in synth_state_dict().items()}
    x = synth_images(2, 96, 128)
    o = model_ref.forward(sd, x, training=True)
    loss, _ = loss_ref.compute_loss_ultra(o, g["step/targets"], sd["head.anchors"])
    loss.backward()
    np.testing.assert_allclose(loss.detach().numpy(), g["step/loss"], rtol=1e-5)
    for key in g.files:
        if key.startswith("step/grad/"):
            k = key[len("step/grad/"):]
            ref = g[key]
            got = sd[k].grad.numpy()
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-7, k


def test_g6_decode(golden):
    g = golden("g6_decode_nms")
    sd = synth_state_dict()
    anchors = sd["head.anchors"]
    p = [torch.from_numpy(g[f"dec/p{i}"]) for i in range(3)]
    out = loss_ref.cells_to_bboxes(p, anchors, [8, 16, 32], is_pred=True)
    assert np.array_equal(out.numpy(), g["dec/out"])
    t = [torch.from_numpy(g[f"dect/t{i}"]) for i in range(3)]
    out = loss_ref.cells_to_bboxes(t, anchors, [8, 16, 32], is_pred=False)
    assert np.array_equal(out.numpy(), g["dect/out"])


def test_g6_nms_rows_bit_exact(golden):
    g = golden("g6_decode_nms")
    for name in g["nms/names"]:
        bx = g[f"nms/{name}/in"]
        thr, iou = g[f"nms/{name}/thr"]
        res = loss_ref.non_max_suppression(bx, float(iou), float(thr), 300)
        for b in range(bx.shape[0]):
            rows, idx = res[b]
            assert np.array_equal(rows.view(np.uint32), g[f"nms/{name}/out{b}"].view(np.uint32)), (name, b)
            assert rows.shape[0] <= 300
            # src indices point at rows with the same class/score
            if len(idx):
                assert np.array_equal(bx[b][idx][:, :2], rows[:, :2])


def test_nms_restatement_pinned_by_reference_aladdin(golden):
    """A12's inner kernel, pinned by reference-held code: torchvision.ops.nms 0.12 is absent from /root/reference and from
    this image, but the reference's OWN greedy NMS (`non_max_suppression_aladdin`, utils/bboxes_utils.py:129-173) defines
    the same kept set in the same order on tie-free single-class inputs with max_detections >= N and every pairwise IoU
    >= 1e-4 away from the threshold (g14, produced by the REAL reference function in make_golden.py). The C restatement
    of torchvision's algorithm, the restated wrapper (midpoint rows -> corners in the reference's order) and the C
    restatement of aladdin must all keep exactly that list."""
    g = golden("g14_nms_crosspin")
    for name in g["names"].tolist():
        thr, iou = g[f"{name}/par"].tolist()
        mid, corners, keep = g[f"{name}/mid"], g[f"{name}/corners"], g[f"{name}/keep"]
        N = mid.shape[0]
        assert cnative.nms_tv012(corners, mid[:, 1], iou).tolist() == keep.tolist(), name
        rows, idx = loss_ref.non_max_suppression(mid[None], float(iou), float(thr), N)[0]
        assert idx.tolist() == keep.tolist(), name
        assert np.array_equal(rows[:, 2:], corners[keep]) and np.array_equal(rows[:, 1], mid[keep, 1])
        al = np.concatenate([mid[:, :2], corners], 1)
        assert cnative.nms_aladdin(al, iou, thr, "corners", N).tolist() == keep.tolist(), name


def test_nms_restatement_vs_independent_numpy_greedy():
    """the torchvision restatement against an independent numpy greedy written here (same published algorithm: areas,
    stable descending order, max(0,.) clamps, `ovr > double(thr)`) on tie-free single-class boxes"""
    rng = np.random.default_rng(9)
    N = 400
    xy = rng.uniform(0, 300, (N, 2)).astype(np.float32)
    wh = rng.uniform(20, 90, (N, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    scores = rng.permutation(N).astype(np.float32) / N
    keep = cnative.nms_tv012(boxes, scores, 0.5)
    order = np.argsort(-scores, kind="stable")
    alive = np.ones(N, bool)
    ref = []
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for i in order:
        if not alive[i]:
            continue
        ref.append(i)
        xx1 = np.maximum(boxes[i, 0], boxes[:, 0]); yy1 = np.maximum(boxes[i, 1], boxes[:, 1])
        xx2 = np.minimum(boxes[i, 2], boxes[:, 2]); yy2 = np.minimum(boxes[i, 3], boxes[:, 3])
        inter = np.maximum(0, xx2 - xx1) * np.maximum(0, yy2 - yy1)
        ovr = inter / (area[i] + area - inter)
        alive &= ~(ovr.astype(np.float64) > 0.5)
    assert list(keep) == ref


def test_oracle_nms_aladdin_matches_reference(golden):
    """non_max_suppression_aladdin (reference utils/bboxes_utils.py:129-173): the C restatement against the kept
    lists the real reference produced (corners / midpoint, score ties, truncation before suppression)"""
    from oracle import cnative
    g = golden("g9_nms_aladdin")
    for name in g["names"].tolist():
        thr, iou, mid, md = g[f"{name}/par"].tolist()
        keep = cnative.nms_aladdin(g[f"{name}/in"], iou, thr, "midpoint" if mid else "corners", int(md))
        assert keep.tolist() == g[f"{name}/keep"].tolist(), name


def test_g10_config0_reference_recipe(golden):
    """BASELINE.json configs[0] (the reference's own CPU-runnable case, my_loss_vs_ultra_loss.py:26-33: seed 355,
    4x3x640x640 uniform images, 12 labels) through the oracle: train-mode logits + ComputeLoss, then the detect
    path (eval forward, decode, NMS at 0.01 / 0.6 / 300), against values produced by the real reference."""
    g = golden("g10_config0")
    torch.manual_seed(355)
    images = torch.rand((4, 3, 640, 640))
    np.testing.assert_allclose(images.reshape(-1)[::4801].numpy(), g["img_sample"], rtol=0, atol=0)
    sd = synth_state_dict()
    labels = torch.from_numpy(g["labels"])
    with torch.no_grad():
        o = model_ref.forward(sd, images, training=True)
        for i in range(3):
            got = o[i].reshape(-1).numpy()[::int(g[f"o{i}_step"])][:4096]
            ref = g[f"o{i}_sample"]
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (i, np.abs(got - ref).max())
        loss, _ = loss_ref.compute_loss_ultra(o, labels, sd["head.anchors"])
        np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-5)
        oe = model_ref.forward(sd, images, training=False)
        boxes = loss_ref.cells_to_bboxes(oe, sd["head.anchors"], [8, 16, 32], is_pred=True)
        np.testing.assert_allclose(boxes[0, ::97, 1].numpy(), g["eval_obj_sample"], rtol=1e-5, atol=1e-6)
        kept = loss_ref.non_max_suppression(boxes.numpy(), 0.6, 0.01, 300)
    assert [len(k[1]) for k in kept] == g["eval_nms_counts"].tolist()


def test_g7_large_batch_first_step(golden):
    """B=16 @ 320x320 train-mode logits + ComputeLoss of the first step through the oracle, against the real
    reference (the fixture the GPU large-batch tests are checked with)."""
    g = golden("g7_large_step")
    B, H, W = [int(v) for v in g["shape"]]
    x = synth_images(B, H, W, seed="img/rank0")
    sd = synth_state_dict()
    with torch.no_grad():
        o = model_ref.forward(sd, x, training=True)
        for i in range(3):
            got = o[i].reshape(-1).numpy()[::int(g[f"o{i}_step"])][:4096]
            ref = g[f"o{i}_sample"]
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (i, np.abs(got - ref).max())
        loss, _ = loss_ref.compute_loss_ultra(o, torch.from_numpy(g["targets"]), sd["head.anchors"])
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-5)


def test_g16_yolo_loss_model_level_sequence(golden):
    """the REAL reference's train-mode forward + a fresh YOLO_LOSS over a pinned two-call sequence at the model level (g16): the
    oracle's restatements (model_ref.forward + loss_ref.YoloLossRef) give the same loss values (1e-5) and the same decayed anchors
    (bit for bit) after every call"""
    g = golden("g16_yolo_train_steps")
    B, H, W = [int(v) for v in g["shape"]]
    sd = synth_state_dict()
    ref = loss_ref.YoloLossRef(sd["head.anchors"])
    assert np.array_equal(ref.anchors.numpy(), g["anchors_start"])
    for call in range(2):
        x = synth_images(B, H, W, seed=f"g16/img{call}")
        boxes = tuple(g[f"{call}/boxes{b}"] for b in range(B))
        with torch.no_grad():
            l = ref(model_ref.forward(sd, x, training=True), boxes)
        np.testing.assert_allclose(float(l), float(g[f"{call}/loss"]), rtol=1e-5)
        assert np.array_equal(ref.anchors.numpy(), g[f"{call}/anchors_after"])
