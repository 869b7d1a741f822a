"""Generate golden vectors by IMPORTING THE REAL REFERENCE (/root/reference) in the build container.

Run:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)

Fixtures are data only (inputs + the reference's outputs); no reference source is copied. The
reference cannot travel to the GPU box, these files can. Inputs are generated from the counter-based
generator in yolov5m_amd/utils/synth.py or from seeded numpy, and are stored in the fixture so tests
never depend on RNG reproducibility.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import ref_import                      # noqa: E402
from yolov5m_amd.utils.synth import synth_state_dict, synth_images, synth_labels, uniform  # noqa: E402

R = ref_import.load()
torch.set_num_threads(8)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def ref_model(nc=80):
    m = R.YOLOV5m(first_out=48, nc=nc, anchors=R.config.ANCHORS, ch=(192, 384, 768))
    return m


# ------------------------------------------------------------------------------------------------
def g1_giou():
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 10, (1000, 4)).astype(np.float32)
    b = rng.uniform(0, 10, (1000, 4)).astype(np.float32)
    # degenerate rows: zero-size, identical, disjoint, contained
    a[0] = [1, 1, 0, 0]; b[0] = [1, 1, 0, 0]
    a[1] = [5, 5, 2, 2]; b[1] = [5, 5, 2, 2]
    a[2] = [1, 1, 1, 1]; b[2] = [8, 8, 1, 1]
    a[3] = [5, 5, 6, 6]; b[3] = [5, 5, 1, 1]
    a[4] = [0, 0, 0, 3]; b[4] = [0, 0, 3, 0]
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    iou = R.intersection_over_union(ta, tb, GIoU=False).numpy()
    giou = R.intersection_over_union(ta, tb, GIoU=True).numpy()
    save("g1_giou", a=a, b=b, iou=iou, giou=giou)


# ------------------------------------------------------------------------------------------------
def _bt_cases():
    rng = np.random.default_rng(2)
    cases = []
    # (name, targets (nt,6), shapes)
    s640 = [(80, 80), (40, 40), (20, 20)]
    t = synth_labels(4, 8).numpy()
    cases.append(("synth_b4", t, s640))
    t = synth_labels(16, 8, seed="lab2").numpy()
    cases.append(("synth_b16", t, s640))
    # config-0 recipe style: randint/100 boxes incl. zero w/h
    t = np.zeros((12, 6), np.float32)
    t[:, 0] = np.repeat(np.arange(4), 3); t[:, 1] = np.repeat(np.arange(4), 3)
    t[:, 2:] = rng.integers(0, 50, (12, 4)) / 100
    cases.append(("cfg0_style", t.astype(np.float32), s640))
    # exact-integer grid coordinates (both +/- neighbours fire), borders, ratio == 4.0 boundary
    t = np.array([
        [0, 1, 0.5, 0.5, 0.1, 0.1],          # gx = 40.0, 20.0, 10.0 exactly
        [0, 2, 0.25, 0.75, 0.2, 0.05],
        [1, 3, 1.0, 1.0, 0.05, 0.05],        # on the far border: gx = nx -> clamp
        [1, 4, 0.0, 0.0, 0.05, 0.05],        # on the near border
        [1, 5, 0.00625, 0.99, 0.03, 0.04],   # < 1 cell from border
        [2, 6, 0.5, 0.5, 4.0 * 1.25 / 80, 1.625 / 80],   # w ratio == 4.0 exactly on scale 0 anchor 0
        [2, 7, 0.5, 0.5, 1.25 / 80 / 4.0, 1.625 / 80],   # 1/r == 4.0
        [2, 8, 0.3, 0.3, 0.0, 0.1],          # zero width -> inf ratio
        [3, 9, 0.7123, 0.2877, 0.11, 0.23],
        [3, 9, 0.7123, 0.2877, 0.11, 0.23],  # duplicate target -> duplicate cells
        [3, 10, 0.7125, 0.2875, 0.12, 0.22],  # same cell, different box
    ], np.float32)
    cases.append(("edges", t, s640))
    cases.append(("rect", synth_labels(2, 5, seed="lab3").numpy(), [(48, 80), (24, 40), (12, 20)]))
    cases.append(("tiny", synth_labels(2, 6, seed="lab4").numpy(), [(8, 8), (4, 4), (2, 2)]))
    cases.append(("empty", np.zeros((0, 6), np.float32), s640))
    cases.append(("one", np.array([[0, 17, 0.51, 0.49, 0.3, 0.4]], np.float32), s640))
    return cases


def g2_build_targets(model):
    lf = R.ComputeLoss(model)
    out = {}
    names = []
    for name, t, shapes in _bt_cases():
        p = [torch.zeros(4, 3, ny, nx, 85) for (ny, nx) in shapes]
        tcls, tbox, indices, anch = lf.build_targets(p, torch.from_numpy(t))
        names.append(name)
        out[f"{name}/targets"] = t
        out[f"{name}/shapes"] = np.array(shapes, np.int64)
        for i in range(3):
            b, a, gj, gi = indices[i]
            out[f"{name}/{i}/b"] = b.numpy().astype(np.int64).reshape(-1)
            out[f"{name}/{i}/a"] = a.numpy().astype(np.int64).reshape(-1)
            out[f"{name}/{i}/gj"] = gj.numpy().astype(np.int64).reshape(-1)
            out[f"{name}/{i}/gi"] = gi.numpy().astype(np.int64).reshape(-1)
            out[f"{name}/{i}/tbox"] = tbox[i].numpy().astype(np.float32).reshape(-1, 4)
            out[f"{name}/{i}/anch"] = anch[i].numpy().astype(np.float32).reshape(-1, 2)
            out[f"{name}/{i}/tcls"] = tcls[i].numpy().astype(np.int64).reshape(-1)
    out["names"] = np.array(names)
    save("g2_build_targets", **out)


# ------------------------------------------------------------------------------------------------
def g3_compute_loss(model):
    lf = R.ComputeLoss(model)
    out = {}
    names = []
    cases = [
        ("tiny_b2", 2, [(8, 8), (4, 4), (2, 2)], synth_labels(2, 6, seed="lab4").numpy()),
        ("mid_b3", 3, [(16, 24), (8, 12), (4, 6)], synth_labels(3, 7, seed="lab5").numpy()),
        ("empty_b2", 2, [(8, 8), (4, 4), (2, 2)], np.zeros((0, 6), np.float32)),
        ("dups_b2", 2, [(8, 8), (4, 4), (2, 2)], np.array(
            [[0, 3, 0.5, 0.5, 0.4, 0.5], [0, 3, 0.5, 0.5, 0.4, 0.5], [1, 5, 0.51, 0.52, 0.5, 0.6],
             [1, 7, 0.52, 0.51, 0.7, 0.4], [1, 7, 0.25, 0.75, 0.3, 0.3]], np.float32)),
        ("b4_640", 4, [(80, 80), (40, 40), (20, 20)], synth_labels(4, 8).numpy()),
    ]
    for name, B, shapes, t in cases:
        p = [(uniform(f"g3/{name}/{i}", (B, 3, ny, nx, 85), -3.0, 3.0)).requires_grad_(True)
             for i, (ny, nx) in enumerate(shapes)]
        loss = lf(p, torch.from_numpy(t), None)
        loss.backward()
        names.append(name)
        out[f"{name}/targets"] = t
        out[f"{name}/shapes"] = np.array(shapes, np.int64)
        out[f"{name}/B"] = np.array(B)
        out[f"{name}/loss"] = loss.detach().numpy()
        for i in range(3):
            if name != "b4_640":
                out[f"{name}/p{i}"] = p[i].detach().numpy()
                out[f"{name}/g{i}"] = p[i].grad.numpy()
            else:   # too big to store: inputs are regenerated from the counter generator; keep checksums
                g = p[i].grad.numpy()
                out[f"{name}/g{i}_sum"] = np.array(g.astype(np.float64).sum())
                out[f"{name}/g{i}_abs"] = np.array(np.abs(g.astype(np.float64)).sum())
                out[f"{name}/g{i}_obj_sample"] = g[..., 4].reshape(-1)[::97].copy()
    out["names"] = np.array(names)
    save("g3_compute_loss", **out)


# ------------------------------------------------------------------------------------------------
def g4_yolo_loss(model):
    """Fresh YOLO_LOSS object, pinned call sequence (anchors decay in place per box)."""
    lf = R.YOLO_LOSS(model, rect_training=False)
    out = {}
    shapes = [(8, 8), (4, 4), (2, 2)]
    B = 2
    rng = np.random.default_rng(4)
    calls = []
    for call in range(2):
        p = [uniform(f"g4/{call}/{i}", (B, 3, ny, nx, 85), -3.0, 3.0).requires_grad_(True)
             for i, (ny, nx) in enumerate(shapes)]
        boxes = []
        for b in range(B):
            n = 3
            arr = np.zeros((n, 5), np.float64)
            arr[:, 0] = rng.integers(0, 80, n)
            arr[:, 1:3] = rng.uniform(0.1, 0.9, (n, 2))
            arr[:, 3:5] = rng.uniform(0.05, 0.6, (n, 2))
            boxes.append(arr)
        # dense targets as the reference builds them (before compute_loss mutates them)
        anchors_before = lf.anchors.clone()
        loss = lf(p, tuple(boxes), pred_size=(64, 64))
        loss.backward()
        out[f"{call}/anchors_before"] = anchors_before.numpy()
        out[f"{call}/anchors_after"] = lf.anchors.clone().numpy()
        out[f"{call}/loss"] = loss.detach().numpy()
        for b in range(B):
            out[f"{call}/boxes{b}"] = boxes[b]
        for i in range(3):
            out[f"{call}/p{i}"] = p[i].detach().numpy()
            out[f"{call}/g{i}"] = p[i].grad.numpy()
        calls.append(call)
    # dense targets for a fresh object, first image only (exact)
    lf2 = R.YOLO_LOSS(model, rect_training=False)
    p = [torch.zeros(1, 3, ny, nx, 85) for (ny, nx) in shapes]
    tg = lf2.build_targets(p, out["0/boxes0"], (64, 64))
    for i in range(3):
        out[f"bt/t{i}"] = tg[i].numpy()
    out["shapes"] = np.array(shapes, np.int64)
    out["B"] = np.array(B)
    save("g4_yolo_loss", **out)


# ------------------------------------------------------------------------------------------------
def g5_model(model):
    sd = synth_state_dict()
    out = {}
    for tag, (B, H, W) in {"s64": (1, 64, 64), "s96x128": (2, 96, 128), "s320": (2, 320, 320)}.items():
        x = synth_images(B, H, W)
        for mode in ("eval", "train"):
            model.load_state_dict(sd, strict=True)
            model.train(mode == "train")
            with torch.no_grad():
                o = model(x.clone())
            for i in range(3):
                flat = o[i].reshape(-1).numpy()
                step = max(1, flat.size // 4096)
                out[f"{tag}/{mode}/o{i}_sample"] = flat[::step][:4096].copy()
                out[f"{tag}/{mode}/o{i}_step"] = np.array(step)
                out[f"{tag}/{mode}/o{i}_sum"] = np.array(flat.astype(np.float64).sum())
                out[f"{tag}/{mode}/o{i}_abs"] = np.array(np.abs(flat.astype(np.float64)).sum())
                if tag == "s64":
                    out[f"{tag}/{mode}/o{i}"] = o[i].numpy()
            if mode == "train":
                nsd = model.state_dict()
                for k in ("backbone.0.cbl.1.running_mean", "backbone.0.cbl.1.running_var",
                          "neck.7.c_out.cbl.1.running_mean", "neck.7.c_out.cbl.1.running_var",
                          "backbone.9.c_out.cbl.1.running_var"):
                    out[f"{tag}/train/{k}"] = nsd[k].numpy().copy()
    # one full training step's gradients at a small size: ComputeLoss + backward
    B, H, W = 2, 96, 128
    model.load_state_dict(sd, strict=True)
    model.train(True)
    model.zero_grad()
    x = synth_images(B, H, W)
    t = synth_labels(B, 5, seed="lab3")
    lf = R.ComputeLoss(model)
    o = model(x.clone())
    loss = lf(o, t, None)
    loss.backward()
    out["step/loss"] = loss.detach().numpy()
    out["step/targets"] = t.numpy()
    named = dict(model.named_parameters())
    for k in ("backbone.0.cbl.0.weight", "backbone.0.cbl.1.weight", "backbone.0.cbl.1.bias",
              "backbone.2.seq.1.c2.cbl.0.weight", "backbone.4.c_out.cbl.0.weight",
              "backbone.9.c1.cbl.0.weight", "backbone.9.c_out.cbl.1.weight", "neck.0.cbl.0.weight",
              "neck.3.seq.1.1.cbl.0.weight", "neck.4.cbl.0.weight", "neck.7.c_out.cbl.0.weight",
              "head.out_convs.0.weight", "head.out_convs.1.bias", "head.out_convs.2.weight"):
        out[f"step/grad/{k}"] = named[k].grad.numpy().copy()
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters()))
    out["step/grad_norm"] = np.array(float(gn))
    save("g5_model", **out)


# ------------------------------------------------------------------------------------------------
def g6_decode_nms(model):
    out = {}
    anchors = model.head.anchors
    strides = model.head.stride
    # decode on small grids
    shapes = [(8, 12), (4, 6), (2, 3)]
    B = 2
    p = [uniform(f"g6/dec/{i}", (B, 3, ny, nx, 85), -6.0, 6.0) for i, (ny, nx) in enumerate(shapes)]
    p[0][0, 0, 0, 0, 5:] = 30.0      # fp32-saturated sigmoid tie -> lowest class index wins
    p[0][0, 0, 0, 1, 5:] = -30.0
    dec = R.cells_to_bboxes([t.clone() for t in p], anchors, strides, is_pred=True, to_list=False)
    for i in range(3):
        out[f"dec/p{i}"] = p[i].numpy()
    out["dec/out"] = dec.numpy()
    # is_pred=False branch (dense targets -> boxes)
    tg = [uniform(f"g6/dect/{i}", (B, 3, ny, nx, 6), 0.0, 1.0) for i, (ny, nx) in enumerate(shapes)]
    dect = R.cells_to_bboxes([t.clone() for t in tg], anchors, strides, is_pred=False, to_list=False)
    for i in range(3):
        out[f"dect/t{i}"] = tg[i].numpy()
    out["dect/out"] = dect.numpy()
    # NMS cases: (B,N,6) inputs -> reference wrapper (around the restated torchvision nms)
    rng = np.random.default_rng(6)
    names = []
    for N in (0, 1, 64, 65, 1000, 5000):
        for (thr, iou) in ((0.01, 0.6), (0.25, 0.45)):
            Bn = 2
            bx = np.zeros((Bn, N, 6), np.float32)
            bx[..., 0] = rng.integers(0, 80, (Bn, N))
            bx[..., 1] = rng.uniform(0, 1, (Bn, N)) ** 2
            bx[..., 2:4] = rng.uniform(0, 640, (Bn, N, 2))
            bx[..., 4:6] = rng.uniform(4, 200, (Bn, N, 2))
            name = f"n{N}_t{thr}_i{iou}"
            names.append(name)
            res = R.non_max_suppression(torch.from_numpy(bx.copy()), iou_threshold=iou, threshold=thr,
                                        max_detections=300, tolist=True)
            out[f"nms/{name}/in"] = bx
            out[f"nms/{name}/thr"] = np.array([thr, iou])
            for b in range(Bn):
                out[f"nms/{name}/out{b}"] = np.array(res[b], np.float32).reshape(-1, 6)
    # clustered boxes (heavy overlap, few survivors) + many survivors > 300
    N = 3000
    bx = np.zeros((1, N, 6), np.float32)
    centers = rng.uniform(100, 500, (12, 2))
    cid = rng.integers(0, 12, N)
    bx[0, :, 0] = rng.integers(0, 3, N)
    bx[0, :, 1] = rng.uniform(0.02, 1, N)
    bx[0, :, 2:4] = centers[cid] + rng.normal(0, 4, (N, 2))
    bx[0, :, 4:6] = rng.uniform(60, 80, (N, 2))
    res = R.non_max_suppression(torch.from_numpy(bx.copy()), iou_threshold=0.45, threshold=0.25,
                                max_detections=300, tolist=True)
    names.append("clustered")
    out["nms/clustered/in"] = bx
    out["nms/clustered/thr"] = np.array([0.25, 0.45])
    out["nms/clustered/out0"] = np.array(res[0], np.float32).reshape(-1, 6)
    out["nms/names"] = np.array(names)
    save("g6_decode_nms", **out)


def g9_nms_aladdin():
    """reference non_max_suppression_aladdin (utils/bboxes_utils.py:129-173) on seeded lists: kept indices"""
    from utils import bboxes_utils as RB
    rng = np.random.default_rng(9)
    out, names = {}, []
    def run(name, bx, thr, iou, fmt, md):
        lst = [[float(v) for v in row] for row in bx]
        # kept rows are identified by object identity (the reference returns the list objects it was given)
        plain = [r[:6] for r in lst]
        kept = RB.non_max_suppression_aladdin(plain, iou, thr, box_format=fmt, max_detections=md)
        pos = {id(r): i for i, r in enumerate(plain)}
        out[f"{name}/in"] = bx
        out[f"{name}/par"] = np.array([thr, iou, 1.0 if fmt == "midpoint" else 0.0, md])
        out[f"{name}/keep"] = np.array([pos[id(r)] for r in kept], dtype=np.int64)
        names.append(name)
    for N in (0, 1, 50, 400, 1500):
        for fmt in ("corners", "midpoint"):
            bx = np.zeros((N, 6), np.float32)
            bx[:, 0] = rng.integers(0, 4, N)
            bx[:, 1] = (rng.uniform(0, 1, N) ** 2).astype(np.float32)
            if N >= 50:
                bx[::7, 1] = bx[3, 1]                  # score ties: the stable order decides
            c = rng.uniform(50, 400, (N, 2)).astype(np.float32)
            wh = rng.uniform(20, 120, (N, 2)).astype(np.float32)
            if fmt == "corners":
                bx[:, 2:4], bx[:, 4:6] = c - wh / 2, c + wh / 2
            else:
                bx[:, 2:4], bx[:, 4:6] = c, wh
            run(f"n{N}_{fmt}_a", bx, 0.05, 0.5, fmt, 300)
            run(f"n{N}_{fmt}_b", bx, 0.3, 0.3, fmt, 40)
    out["names"] = np.array(names)
    save("g9_nms_aladdin", **out)


def g14_nms_crosspin():
    """Cross-pin of the torchvision-0.12 NMS restatement (absent from /root/reference and from this image) against NMS
    arithmetic the reference DOES hold: its own `non_max_suppression_aladdin` (utils/bboxes_utils.py:129-173). On tie-free,
    single-class inputs with max_detections >= N and every pairwise IoU at least 1e-4 away from the threshold, the two
    define the same kept set in the same order: aladdin keeps a box iff IoU(+1e-7 in the union) < thr against every kept
    box of its class, torchvision suppresses iff IoU > thr -- the 1e-7 epsilon and the </> boundary cannot matter at that
    margin. Boxes are midpoint rows [cls, score, x, y, w, h] whose corner conversion is EXACT in float32 (x1, y1 multiples
    of 1/4, w, h multiples of 1/2), so the reference's wrapper order x1=x-w/2; y1=y-h/2; y2=h+y1; x2=w+x1 (:190-193) and
    the corners handed to aladdin are the same float32 numbers. Stored: the midpoint rows, the corners, the kept indices."""
    from utils import bboxes_utils as RB
    out, names = {}, []
    for N, iou_thr, seed in ((64, 0.45, 1), (65, 0.6, 2), (1000, 0.45, 3), (1000, 0.6, 4), (5000, 0.45, 5), (5000, 0.6, 6)):
        rng = np.random.default_rng(1400 + seed)
        span = {64: 200, 65: 200, 1000: 400, 5000: 420}[N]      # (dense: the reference's greedy loop is O(N * kept) torch calls)
        def draw(n):
            x1 = np.floor(rng.uniform(0, span, n) * 4) / 4
            y1 = np.floor(rng.uniform(0, span, n) * 4) / 4
            w = np.floor(rng.uniform(30, 110, n) * 2) / 2
            h = np.floor(rng.uniform(30, 110, n) * 2) / 2
            return np.stack([x1, y1, w, h], 1)
        geo = draw(N)
        while True:
            # every pairwise IoU (float64) clear of the threshold by 1e-4: boxes of offending pairs are re-drawn until none is
            # left (with 12.5 M pairs at N = 5000 a fresh draw of the whole set never passes); chunked to bound memory
            c = np.stack([geo[:, 0], geo[:, 1], geo[:, 0] + geo[:, 2], geo[:, 1] + geo[:, 3]], 1).astype(np.float64)
            area = (c[:, 2] - c[:, 0]) * (c[:, 3] - c[:, 1])
            bad = set()
            for i0 in range(0, N, 500):
                a = c[i0:i0 + 500, None, :]
                iw = np.clip(np.minimum(a[..., 2], c[None, :, 2]) - np.maximum(a[..., 0], c[None, :, 0]), 0, None)
                ih = np.clip(np.minimum(a[..., 3], c[None, :, 3]) - np.maximum(a[..., 1], c[None, :, 1]), 0, None)
                inter = iw * ih
                iou = inter / (area[i0:i0 + 500, None] + area[None, :] - inter)
                ii, jj = np.nonzero(np.abs(iou - iou_thr) < 1e-4)
                bad.update(int(max(i + i0, j)) for i, j in zip(ii, jj) if i + i0 != j)
            if not bad:
                break
            idx = np.array(sorted(bad))
            geo[idx] = draw(len(idx))
        x1, y1, w, h = geo[:, 0], geo[:, 1], geo[:, 2], geo[:, 3]
        score = (rng.permutation(N).astype(np.float32) + 1.0) / np.float32(N + 1)      # tie-free, all > the 1e-4 score threshold
        assert len(np.unique(score)) == N and score.min() > 1e-4
        mid = np.zeros((N, 6), np.float32)
        mid[:, 1] = score
        mid[:, 2], mid[:, 3], mid[:, 4], mid[:, 5] = x1 + w / 2, y1 + h / 2, w, h
        xa = mid[:, 2] - mid[:, 4] / 2; ya = mid[:, 3] - mid[:, 5] / 2          # the wrapper's float32 arithmetic, in its order
        yb = mid[:, 5] + ya; xb = mid[:, 4] + xa
        corners = np.stack([xa, ya, xb, yb], 1).astype(np.float32)
        assert np.array_equal(corners.astype(np.float64), c), "corner conversion must be exact"
        lst = [[0.0, float(s)] + [float(v) for v in cc] for s, cc in zip(score, corners)]
        kept = RB.non_max_suppression_aladdin(lst, iou_thr, 1e-4, box_format="corners", max_detections=N)
        pos = {id(r): i for i, r in enumerate(lst)}
        keep = np.array([pos[id(r)] for r in kept], dtype=np.int64)
        name = f"n{N}_i{iou_thr}"
        out[f"{name}/mid"], out[f"{name}/corners"], out[f"{name}/keep"] = mid, corners, keep
        out[f"{name}/par"] = np.array([1e-4, iou_thr])
        names.append(name)
        print(name, "kept", len(keep))
    out["names"] = np.array(names)
    save("g14_nms_crosspin", **out)


def g10_config0(model):
    """BASELINE.json configs[0] (the reference's own CPU-runnable case, ultralytics_files/my_loss_vs_ultra_loss.py
    :26-33): torch.manual_seed(355); images = rand(4,3,640,640); 12 labels [img, cls, x, y, w, h] with
    bboxes = randint(0,50,(12,4))/100. Weights: the counter-based generator (the script's are unseeded).
    Stored: inputs' labels, train-mode logits samples, ComputeLoss, decode+NMS counts at (0.01, 0.6, 300)."""
    sd = synth_state_dict()
    torch.manual_seed(355)
    images = torch.rand((4, 3, 640, 640))
    img_idx = torch.arange(4).repeat(3, 1).T.reshape(12, 1)
    classes = torch.arange(4).repeat(3, 1).T.reshape(12, 1)
    bboxes = torch.randint(low=0, high=50, size=(12, 4)) / 100
    labels = torch.cat([img_idx, classes, bboxes], dim=-1).float()
    model.load_state_dict(sd, strict=True)
    model.train(True)
    lf = R.ComputeLoss(model)
    with torch.no_grad():
        o = model(images.clone())
        loss = lf(o, labels, None)
    out = {"labels": labels.numpy(), "loss": np.array(float(loss)), "img_sum": np.array(float(images.double().sum())),
           "img_sample": images.reshape(-1)[::4801].numpy().copy()}
    for i in range(3):
        flat = o[i].reshape(-1).numpy()
        step = max(1, flat.size // 4096)
        out[f"o{i}_sample"] = flat[::step][:4096].copy()
        out[f"o{i}_step"] = np.array(step)
    model.train(False)
    model.load_state_dict(sd, strict=True)
    with torch.no_grad():
        oe = model(images.clone())
        boxes = R.cells_to_bboxes(oe, model.head.anchors, model.head.stride, is_pred=True, to_list=False)
        kept = R.non_max_suppression(boxes, iou_threshold=0.6, threshold=0.01, max_detections=300, tolist=True)
    out["eval_nms_counts"] = np.array([len(k) for k in kept])
    out["eval_nms_first"] = np.array(kept[0][:20], np.float32).reshape(-1, 6)
    out["eval_obj_sample"] = boxes[0, ::97, 1].numpy().copy()
    save("g10_config0", **out)


def g11_eval_path(model):
    """reference YOLO_EVAL (utils/validation_utils.py) on a two-batch synthetic loader: class / obj accuracies and the
    (preds, targets) lists handed to MeanAveragePrecision.update (captured from the stub). Dense targets come from the
    reference's YOLO_LOSS.build_targets on a FRESH object per batch item order (anchor-decay state, App. C.1)."""
    import tempfile
    from utils import validation_utils as RV
    sd = synth_state_dict()
    model.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(11)
    B, S = 2, 96
    loss_obj = R.YOLO_LOSS(model, rect_training=False)
    batches, out = [], {}
    for bi in range(2):
        img = torch.from_numpy(rng.integers(0, 256, (B, 3, S, S), dtype=np.uint8))
        labs = []
        for b in range(B):
            n = 3
            lab = np.zeros((n, 5))
            lab[:, 0] = rng.integers(0, 80, n)
            lab[:, 1:3] = rng.uniform(0.1, 0.9, (n, 2))
            lab[:, 3:5] = rng.uniform(0.05, 0.4, (n, 2))
            labs.append(lab)
        model.eval()
        with torch.no_grad():
            o = model(img.float() / 255)
        tg = [loss_obj.build_targets(o, lab, (S, S)) for lab in labs]
        dense = [torch.stack([t[i] for t in tg], dim=0) for i in range(3)]
        batches.append((img, dense))
        out[f"b{bi}/img"] = img.numpy()
        for i in range(3):
            out[f"b{bi}/dense{i}"] = dense[i].numpy()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            ev = RV.YOLO_EVAL(save_logs=False, conf_threshold=0.01, nms_iou_thresh=0.6, map_iou_thresh=0.5, device="cpu",
                              filename="g11", resume=False)
            # the accuracies are only printed by the reference: recompute them through its own code path by
            # enabling save_logs on the instance after construction (rounded to 3 decimals there)
            ev.save_logs = True
            ev.check_class_accuracy(model, [(im.clone(), [d.clone() for d in dn]) for im, dn in batches])
            out["class_accuracy"] = np.array(ev.class_accuracy)
            out["obj_accuracy"] = np.array(ev.obj_accuracy)
            ev.save_logs = False
            RV.MeanAveragePrecision.reset_mock()
            ev.map_pr_rec(model, [(im.clone(), [d.clone() for d in dn]) for im, dn in batches], model.head.anchors, 0)
            preds, targets = RV.MeanAveragePrecision.return_value.update.call_args[0]
        finally:
            os.chdir(cwd)
    for bi in range(2):
        out[f"b{bi}/pred_n"] = np.array(preds[bi]["boxes"].shape[0])
        out[f"b{bi}/pred_scores"] = preds[bi]["scores"].numpy()[:64].copy()
        out[f"b{bi}/true_boxes"] = targets[bi]["boxes"].numpy().copy()
        out[f"b{bi}/true_labels"] = targets[bi]["labels"].numpy().copy()
    save("g11_eval_path", **out)


def g7_large_step(model):
    """first train-mode forward + ComputeLoss of the reference at a batch large enough that every layer width
    runs its multi-workgroup reductions (B=16 @ 320x320): the loss and its 3 components, and sampled logits"""
    sd = synth_state_dict()
    B, S = 16, 320
    model.load_state_dict(sd, strict=True)
    model.train(True)
    x = synth_images(B, S, S, seed="img/rank0")
    t = synth_labels(B, 8, seed="lab/rank0")
    lf = R.ComputeLoss(model)
    with torch.no_grad():
        o = model(x.clone())
        loss = lf(o, t, None)
    out = {"loss": np.array(float(loss)), "targets": t.numpy(), "shape": np.array([B, S, S])}
    for i in range(3):
        flat = o[i].reshape(-1).numpy()
        step = max(1, flat.size // 4096)
        out[f"o{i}_sample"] = flat[::step][:4096].copy()
        out[f"o{i}_step"] = np.array(step)
    save("g7_large_step", **out)


def g16_yolo_train_steps(model):
    """The reference's DEFAULT recipe at the model level (train.py:102-106: YOLO_LOSS): its own train-mode forward on the synthetic
    weights + a FRESH YOLO_LOSS object over a pinned sequence of two calls (the anchors decay in place with every box) on
    two batches of 2 x 96 x 128 with ragged box counts; no optimizer step in between (the weights stay the synthetic ones, so the
    sequence isolates the loss object's state). Stored: inputs (boxes), loss per call, anchors after each call. Pins the fused
    NativeTrainStep(model, YOLO_LOSS) against the real reference (tests/test_gpu_model.py::test_native_yolo_steps_reference_golden)."""
    sd = synth_state_dict()
    model.load_state_dict(sd, strict=True)
    model.train(True)
    lf = R.YOLO_LOSS(model, rect_training=False)
    B, H, W = 2, 96, 128
    out = {"shape": np.array([B, H, W]), "anchors_start": lf.anchors.clone().numpy()}
    for call in range(2):
        x = synth_images(B, H, W, seed=f"g16/img{call}")
        t = synth_labels(B, 4, seed=f"g16/lab{call}").numpy().astype(np.float64)
        boxes = tuple(t[t[:, 0] == b][: 2 + b + call, 1:].copy() for b in range(B))       # 2 + 3, then 3 + 4 boxes
        with torch.no_grad():
            o = model(x.clone())
            loss = lf(o, boxes, pred_size=(H, W))
        out[f"{call}/loss"] = np.array(float(loss))
        out[f"{call}/anchors_after"] = lf.anchors.clone().numpy()
        for b in range(B):
            out[f"{call}/boxes{b}"] = boxes[b]
    # (running statistics moved by the two train-mode forwards: not part of the fixture; the native step is run with lr = 0)
    save("g16_yolo_train_steps", **out)


def g17_train_loop_epoch(model):
    """The REAL reference's train_loop (utils/training_utils.py:81-132) for one epoch on this CPU -- 3 uint8 batches of 2 x 64 x 96,
    multi_scale off, Adam(lr, weight_decay) as train.py:61 builds it, GradScaler / autocast inert on a CPU -- for both losses of
    train.py:102-106: batch 2 -> accumulate = 32 -> ONE forced optimizer step on the last batch (clip 10, Adam with L2 decay) on
    the gradients of the three batches. Stored: the batches, every batch's loss value, and the epoch's parameter update at 8192 strided
    positions (+ its absolute sum). Pins train_loop's mechanics -- with a torch optimizer AND with the fused NativeTrainStep in its
    place -- against the real thing (tests/test_gpu_model.py::test_train_loop_epoch_reference_golden)."""
    import contextlib
    import io
    from utils.training_utils import train_loop as ref_train_loop          # the reference's (oracle.ref_import put it on sys.path)
    g = torch.Generator().manual_seed(17)
    imgs = [torch.randint(0, 256, (2, 3, 64, 96), generator=g, dtype=torch.uint8) for _ in range(3)]
    labs = [synth_labels(2, 3, seed=f"g17/lab{i}") for i in range(3)]
    out = {"images": torch.stack(imgs).numpy(), "labels": torch.stack(labs).numpy(),
           "lr": np.array(R.config.LEARNING_RATE), "weight_decay": np.array(R.config.WEIGHT_DECAY)}
    for kind in ("ultralytics", "yolo"):
        model.load_state_dict(synth_state_dict(), strict=True)
        model.train(True)
        p0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
        opt = torch.optim.Adam(model.parameters(), lr=R.config.LEARNING_RATE, weight_decay=R.config.WEIGHT_DECAY)
        lf = R.ComputeLoss(model) if kind == "ultralytics" else R.YOLO_LOSS(model, rect_training=False)
        losses = []

        def rec(*a, _lf=lf, **k):
            l = _lf(*a, **k)
            losses.append(float(l))
            return l
        if kind == "ultralytics":
            loader = list(zip(imgs, labs))
        else:
            loader = [(im, tuple(lb.numpy().astype(np.float64)[lb[:, 0] == b][:, 1:] for b in range(2))) for im, lb in zip(imgs, labs)]
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            ref_train_loop(model, loader, opt, rec, torch.cuda.amp.GradScaler(), 0, 1, multi_scale_training=False)
        d = torch.cat([p.detach().reshape(-1) for p in model.parameters()]) - p0
        step = d.numel() // 8192
        out[f"{kind}/losses"] = np.array(losses)
        out[f"{kind}/update_sample"] = d[::step][:8192].numpy().copy()
        out[f"{kind}/update_step"] = np.array(step)
        out[f"{kind}/update_abs_sum"] = np.array(float(d.double().abs().sum()))
        print(kind, losses, float(d.abs().max()))
    save("g17_train_loop_epoch", **out)


def g18_train_loop_accumulation(model):
    """The REAL reference's train_loop with SEVERAL optimizer steps in the epoch: 7 uint8 batches of 22 x 32 x 32 -> accumulate =
    round(64 / 22) = 3 -> steps after batches 3 and 6 (`idx - last_opt_step >= accumulate`, :116) and the forced one on batch 7 alone;
    ComputeLoss. Stored: batches, per-batch losses, the parameter update at 8192 strided positions + its absolute sum, how many times
    optim.step ran -- and the reference's OWN sensitivity: the same epoch from weights perturbed by 1e-7 / 1e-6 relative (one f32 ulp /
    ten) moves that update by 2.4 % / 3.9 % in relative L2 (train-mode BatchNorm over 22 samples at 1 x 1, three Adam steps): the
    yardstick an implementation with another summation order is held to. Pins the accumulation rule of train_loop."""
    import contextlib
    import io
    from utils.training_utils import train_loop as ref_train_loop
    g = torch.Generator().manual_seed(18)
    nb, B = 7, 22
    imgs = [torch.randint(0, 256, (B, 3, 32, 32), generator=g, dtype=torch.uint8) for _ in range(nb)]
    labs = [synth_labels(B, 2, seed=f"g18/lab{i}") for i in range(nb)]
    loader = list(zip(imgs, labs))

    def epoch(eps):
        sd = synth_state_dict()
        if eps:
            gen = torch.Generator().manual_seed(1)
            sd = {k: (v * (1 + eps * torch.randn(v.shape, generator=gen)) if v.is_floating_point() and "anchors" not in k else v) for k, v in sd.items()}
        model.load_state_dict(sd, strict=True)
        model.train(True)
        p0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
        opt = torch.optim.Adam(model.parameters(), lr=R.config.LEARNING_RATE, weight_decay=R.config.WEIGHT_DECAY)
        steps, losses = [], []
        real_step = opt.step
        opt.step = lambda *a, **k: (steps.append(1), real_step(*a, **k))[1]
        lf = R.ComputeLoss(model)

        def rec(*a, **k):
            l = lf(*a, **k)
            losses.append(float(l))
            return l
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            ref_train_loop(model, loader, opt, rec, torch.cuda.amp.GradScaler(), 0, 1, multi_scale_training=False)
        return torch.cat([p.detach().reshape(-1) for p in model.parameters()]) - p0, losses, len(steps)
    d, losses, nsteps = epoch(0.0)
    step = d.numel() // 8192
    ref = d[::step][:8192].numpy().copy()
    sens = {}
    for eps in (1e-7, 1e-6):
        dp = epoch(eps)[0][::step][:8192].numpy()
        sens[eps] = float(np.linalg.norm(dp - ref) / np.linalg.norm(ref))
    print("optimizer steps", nsteps, "losses", losses, "max |update|", float(d.abs().max()), "sensitivity", sens)
    save("g18_train_loop_accumulation", images=torch.stack(imgs).numpy(), labels=torch.stack(labs).numpy(), losses=np.array(losses),
         optimizer_steps=np.array(nsteps), update_sample=ref, update_step=np.array(step),
         update_abs_sum=np.array(float(d.double().abs().sum())), lr=np.array(R.config.LEARNING_RATE), weight_decay=np.array(R.config.WEIGHT_DECAY),
         update_rel_l2_weights_1e7=np.array(sens[1e-7]), update_rel_l2_weights_1e6=np.array(sens[1e-6]))


def g8_input_stage():
    """reference input stage (utils/training_utils.py:98-100): images.float()/255 then multi_scale with a pinned
    `random` seed: the chosen sizes for several seeds, and sampled output values for two of them"""
    import random
    from utils import training_utils as RT
    rng = np.random.default_rng(8)
    img = torch.from_numpy(rng.integers(0, 256, (2, 3, 96, 160), dtype=np.uint8))
    out = {"img": img.numpy()}
    sizes = []
    for seed in range(12):
        random.seed(seed)
        o = RT.multi_scale(img.float() / 255, target_shape=640, max_stride=32)
        sizes.append([seed, o.shape[2], o.shape[3]])
        if seed in (0, 5):
            flat = o.reshape(-1).numpy()
            step = max(1, flat.size // 8192)
            out[f"seed{seed}/sample"] = flat[::step][:8192].copy()
            out[f"seed{seed}/step"] = np.array(step)
    out["sizes"] = np.array(sizes)
    save("g8_input_stage", **out)


# ------------------------------------------------------------------------------------------------
def _sample_idx(n, k=256):
    step = max(1, n // k)
    return np.arange(0, n, step)[:k]


def g13_fp64_and_full_gradients(model):
    """Precision yardsticks from the REAL reference (VERDICT r1 "weak" 1-3):
      * train-mode logits of the g5 / g7 shapes with the reference model in float64: tests bound the HIP f32 path's
        distance to these by a multiple of the reference's OWN f32 distance (the g5 / g7 samples), instead of a
        hand-picked tolerance -- BatchNorm over few samples amplifies f32 round-off, and this measures by how much;
      * one full training step at B = 16 @ 320x320 (every kernel variant fires): ComputeLoss and the gradient of ALL
        243 parameter tensors, in float32 and in float64: per tensor the L2 norm and 256 sampled values;
      * BASELINE.json configs[2]: the first-step ComputeLoss at B = 64 @ 640x640 on bench.py's inputs and initial weights
        (torch.manual_seed(0) initialisation of the module, loaded into the reference model)."""
    import copy
    sd = synth_state_dict()
    out = {}
    m64 = copy.deepcopy(model).double()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    for tag, (B, H, W, seed) in {"s64": (1, 64, 64, None), "s96x128": (2, 96, 128, None), "s320": (2, 320, 320, None),
                                 "b16_320": (16, 320, 320, "img/rank0")}.items():
        x = synth_images(B, H, W) if seed is None else synth_images(B, H, W, seed=seed)
        m64.load_state_dict(sd64, strict=True)
        m64.train(True)
        with torch.no_grad():
            o = m64(x.double())
        for i in range(3):
            flat = o[i].reshape(-1).numpy()
            step = max(1, flat.size // 4096)
            out[f"{tag}/train64/o{i}_sample"] = flat[::step][:4096].copy()
            out[f"{tag}/train64/o{i}_step"] = np.array(step)
    # ---- full gradients, B=16 @ 320^2, f32 and f64
    B, S = 16, 320
    x = synth_images(B, S, S, seed="img/rank0")
    t = synth_labels(B, 8, seed="lab/rank0")
    out["grad/targets"] = t.numpy()
    out["grad/shape"] = np.array([B, S, S])
    names = None
    for prec, mdl, state, cast in (("f32", model, sd, lambda v: v), ("f64", m64, sd64, lambda v: v.double())):
        mdl.load_state_dict(state, strict=True)
        mdl.train(True)
        mdl.zero_grad()
        lf = R.ComputeLoss(mdl)
        o = mdl(cast(x.clone()))
        loss = lf(o, t, None)
        loss.backward()
        out[f"grad/{prec}/loss"] = np.array(float(loss))
        named = list(mdl.named_parameters())
        names = [k for k, _ in named]
        norms, samples = [], []
        for k, p in named:
            g = p.grad.reshape(-1).double().numpy()
            norms.append(np.sqrt((g * g).sum()))
            v = np.zeros(256)
            idx = _sample_idx(g.size)
            v[:idx.size] = g[idx]
            samples.append(v)
        out[f"grad/{prec}/norm"] = np.array(norms)
        out[f"grad/{prec}/sample"] = np.array(samples).astype(np.float64 if prec == "f64" else np.float32)
    out["grad/names"] = np.array(names)
    # ---- configs[2] first-step loss on bench.py's inputs and initial weights
    from yolov5m_amd.model import YOLOV5m
    from yolov5m_amd import config as C
    torch.manual_seed(0)
    mine = YOLOV5m(first_out=C.FIRST_OUT, nc=80, anchors=C.ANCHORS, ch=(C.FIRST_OUT * 4, C.FIRST_OUT * 8, C.FIRST_OUT * 16))
    model.load_state_dict(mine.state_dict(), strict=True)
    model.train(True)
    B, S = 64, 640
    x = synth_images(B, S, S, seed="img/rank0")
    t = synth_labels(B, 8, seed="lab/rank0")
    lf = R.ComputeLoss(model)
    with torch.no_grad():
        o = model(x)
        loss = lf(o, t, None)
    out["b64_640/loss"] = np.array(float(loss))
    out["b64_640/obj_logit_mean"] = np.array([float(oi[..., 4].mean()) for oi in o])
    save("g13_precision", **out)


# ------------------------------------------------------------------------------------------------
def g15_full_size_backward(model):
    """configs[2] at FULL size: the real reference's f32 forward + ComputeLoss + backward at B = 64 @ 640x640 on bench.py's
    own inputs and initial weights (torch.manual_seed(0) default init, synth images / labels of rank 0). Stored: the loss,
    the L2 norm of every one of the 243 parameter gradients, the total norm, and 256 sampled elements of 32 tensors spread
    over the network. Memory: autograd would hold ~40 GB of saved activations at this size, so every top-level block of
    the backbone and neck is wrapped in torch.utils.checkpoint (the block's own forward is re-run during backward: the
    gradients are those of the un-wrapped model up to f32 re-association; BatchNorm batch statistics are recomputed from
    the same inputs). The reference code itself is untouched."""
    from torch.utils.checkpoint import checkpoint
    from yolov5m_amd.model import YOLOV5m
    from yolov5m_amd import config as C
    torch.manual_seed(0)
    mine = YOLOV5m(first_out=C.FIRST_OUT, nc=80, anchors=C.ANCHORS, ch=(C.FIRST_OUT * 4, C.FIRST_OUT * 8, C.FIRST_OUT * 16))
    model.load_state_dict(mine.state_dict(), strict=True)
    model.train(True)
    model.zero_grad()
    for seq in (model.backbone, model.neck):
        for blk in seq:
            blk.forward = (lambda x, f=blk.forward: checkpoint(f, x, use_reentrant=False))
    B, S = 64, 640
    x = synth_images(B, S, S, seed="img/rank0").requires_grad_(True)     # (a checkpointed block needs a grad-requiring input)
    t = synth_labels(B, 8, seed="lab/rank0")
    lf = R.ComputeLoss(model)
    o = model(x)
    loss = lf(o, t, None)
    loss.backward()
    named = list(model.named_parameters())
    out = {"loss": np.array(float(loss)), "names": np.array([k for k, _ in named])}
    norms = np.array([float(p.grad.double().norm()) for _, p in named])
    out["norm"] = norms
    out["total_norm"] = np.array(float(np.sqrt((norms ** 2).sum())))
    pick = sorted(set(np.linspace(0, len(named) - 1, 32).round().astype(int).tolist()))
    out["sampled"] = np.array(pick, dtype=np.int64)
    samples = []
    for i in pick:
        g = named[i][1].grad.reshape(-1).numpy()
        v = np.zeros(256, np.float32)
        idx = _sample_idx(g.size)
        v[:idx.size] = g[idx]
        samples.append(v)
    out["sample"] = np.array(samples)
    print("g15: loss", float(loss), "total grad norm", float(out["total_norm"]))
    save("g15_full_size_backward", **out)


def g12_yolo_build_targets(model):
    """YOLO_LOSS.build_targets of the REAL reference at batch scale: a fresh loss object, two calls of 16 images x 8
    boxes at 320x320 (cell grids 40 / 20 / 10). 256 boxes in a row walk the loss object's anchors through the whole
    in-place decay: real anchors, shrinking anchors, fp32 denormals, exact zeros -- i.e. distinct IoUs, then
    nine-way ties in the anchor ranking. Includes boxes that share a cell (slot already taken) and a pair of wide boxes
    per image (ignore cells). Stored sparsely: the non-zero cells of every scale and the anchor state after each call."""
    lf = R.YOLO_LOSS(model, rect_training=False)
    shapes = [(40, 40), (20, 20), (10, 10)]
    B, nb = 16, 8
    rng = np.random.default_rng(12)
    out = {"shapes": np.array(shapes, np.int64), "B": np.array(B), "anchors0": lf.anchors.clone().numpy()}
    p = [torch.zeros(1, 3, ny, nx, 85) for (ny, nx) in shapes]
    for call in range(2):
        dense = [[] for _ in range(3)]
        for b in range(B):
            arr = np.zeros((nb, 5), np.float64)
            arr[:, 0] = rng.integers(0, 80, nb)
            arr[:, 1:3] = rng.uniform(0.02, 0.98, (nb, 2))
            arr[:, 3:5] = rng.uniform(0.02, 0.7, (nb, 2))
            arr[1, 1:3] = arr[0, 1:3]                        # same cell as box 0 on every scale
            arr[2, 3:5] = rng.uniform(0.5, 0.95, 2)          # wide boxes: several anchors above the ignore threshold
            arr[3, 3:5] = arr[2, 3:5] * rng.uniform(0.9, 1.1, 2)
            arr[3, 1:3] = np.clip(arr[2, 1:3] + rng.uniform(-0.01, 0.01, 2), 0.02, 0.98)
            out[f"{call}/boxes{b}"] = arr
            tg = lf.build_targets(p, arr, (320, 320))
            for i in range(3):
                dense[i].append(tg[i].numpy())
        out[f"{call}/anchors_after"] = lf.anchors.clone().numpy()
        for i in range(3):
            d = np.stack(dense[i], 0)                        # (B,3,ny,nx,6)
            nz = np.argwhere(np.any(d != 0, axis=-1)).astype(np.int32)
            out[f"{call}/nz{i}"] = nz
            out[f"{call}/val{i}"] = d[tuple(nz.T)].astype(np.float32)
    save("g12_yolo_build_targets", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g16", "g17", "g18"]
    torch.manual_seed(0)
    model = ref_model()
    if "g1" in which: g1_giou()
    if "g2" in which: g2_build_targets(model)
    if "g3" in which: g3_compute_loss(model)
    if "g4" in which: g4_yolo_loss(model)
    if "g6" in which: g6_decode_nms(model)
    if "g5" in which: g5_model(model)
    if "g7" in which: g7_large_step(model)
    if "g8" in which: g8_input_stage()
    if "g9" in which: g9_nms_aladdin()
    if "g10" in which: g10_config0(model)
    if "g11" in which: g11_eval_path(model)
    if "g12" in which: g12_yolo_build_targets(model)
    if "g13" in which: g13_fp64_and_full_gradients(model)
    if "g14" in which: g14_nms_crosspin()
    if "g16" in which: g16_yolo_train_steps(model)
    if "g17" in which: g17_train_loop_epoch(model)
    if "g18" in which: g18_train_loop_accumulation(model)
    if "g15" in which: g15_full_size_backward(model)        # (not in the default list: ~10 min and ~25 GB of host memory)
