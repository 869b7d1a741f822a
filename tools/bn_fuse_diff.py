#!/usr/bin/env python
"""Where do the two BatchNorm statistics forms first diverge? (total bf16 gradient norm 187.9 with
Y5M_BN_FUSE=0 -- partial rows + finalise launches -- against 199-202 with the default accumulator rows; quantised oracle 192.)
Runs ONE native bf16 step (B=16 @ 320x320, the precision tests' batch) in two child processes that differ only in
Y5M_BN_FUSE (read once per process), dumps per-layer quantities and prints, in forward order for the forward quantities
and in backward order for the backward ones, the relative difference of each -- and the first layer where it exceeds 1e-3.
usage: python tools/bn_fuse_diff.py            (parent)      |  python tools/bn_fuse_diff.py child <out.npz>"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def child(path, seed="rank0"):
    import torch
    from yolov5m_amd import config
    from yolov5m_amd.model import YOLOV5m
    from yolov5m_amd.ultralytics_loss import ComputeLoss
    from yolov5m_amd.utils.training_utils import NativeTrainStep
    from yolov5m_amd.utils.synth import synth_images, synth_labels, synth_state_dict
    B, S = 16, 320
    m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
    m.load_state_dict(synth_state_dict(), strict=True)
    m = m.to("cuda"); m.compute_dtype = "bf16"; m.train()
    step = NativeTrainStep(m, ComputeLoss(m), nt_max=B * 8, use_graph=False)
    x = synth_images(B, S, S, seed=f"img/{seed}").to("cuda")
    t = synth_labels(B, 8, seed=f"lab/{seed}").to("cuda")
    eng = step.load_inputs(x, t)
    step._enqueue_fb(eng)
    torch.cuda.synchronize()
    out = {"names": np.array([l.name for l in eng.layers]), "loss": step.loss_out.float().cpu().numpy(),
           "gnorm": np.array(float(m.flat_grads.double().norm()))}
    def nrm(act):
        return float(act.as_nchw_f32().double().norm())
    for i, lay in enumerate(eng.layers):
        P = m.pslices[lay.name]
        ybuf, yoff, yld = lay.y_view
        yv = torch.as_strided(ybuf.view(-1), (lay.M, lay.cout), (yld, 1), yoff).double()
        out[f"{i}/bn"] = lay.bn.double().cpu().numpy()                  # scale, shift, mean, invstd
        out[f"{i}/ysum"] = yv.sum(0).cpu().numpy()                      # per channel, from the STORED bf16 y
        out[f"{i}/ysq"] = (yv * yv).sum(0).cpu().numpy()
        out[f"{i}/fw"] = np.array([float(yv.norm()), nrm(lay.z)])
        out[f"{i}/bw"] = np.array([nrm(lay.z.grad), float(P["gw"].double().norm()), float(P["gg"].double().norm()),
                                   float(P["gb"].double().norm())])
        out[f"{i}/gg"] = P["gg"].double().cpu().numpy()
        out[f"{i}/gb"] = P["gb"].double().cpu().numpy()
        out[f"{i}/M"] = np.array(lay.M)
    np.savez(path, **out)


def main():
    tmp = os.path.join(ROOT, "gpurun_out", "bn_fuse_diff")
    os.makedirs(tmp, exist_ok=True)
    # the same comparison on other batches first: is the sign of (fuse1 - fuse0) a property of the form or of the batch?
    print("total gradient norm per batch (seed), Y5M_BN_FUSE=0 / 1, and a repeat of form 1 (run-to-run noise):")
    for seed in ("rank0", "rank1", "rank2", "rank3", "rank4"):
        g = []
        for f in ("0", "1", "1"):
            path = os.path.join(tmp, f"seed_{seed}_{f}.npz")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", path, seed], env=dict(os.environ, Y5M_BN_FUSE=f),
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-3000:]
            d = np.load(path)
            g.append((float(d["gnorm"]), float(d["loss"][0])))
            os.remove(path)
        print(f"  {seed}: fuse0 {g[0][0]:9.3f}  fuse1 {g[1][0]:9.3f} ({(g[1][0] / g[0][0] - 1) * 100:+5.1f} %)  fuse1 again {g[2][0]:9.3f}   "
              f"loss {g[0][1]:.4f} / {g[1][1]:.4f}", flush=True)
    print()
    runs = {}
    for f in ("0", "1"):
        path = os.path.join(tmp, f"fuse{f}.npz")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], env=dict(os.environ, Y5M_BN_FUSE=f),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        runs[f] = np.load(path)
    a, b = runs["0"], runs["1"]
    names = a["names"].tolist()
    rel = lambda u, v: float(np.abs(u - v).max() / max(np.abs(u).max(), 1e-30))
    print(f"loss  fuse0 {a['loss'][0]:.5f}  fuse1 {b['loss'][0]:.5f}     total grad norm  fuse0 {float(a['gnorm']):.3f}  fuse1 {float(b['gnorm']):.3f}")
    print("\nforward order: rel. difference of (scale, shift, mean, invstd), |y|, |z|; and each form's own statistics against the"
          " statistics of its STORED bf16 y (mean_stored - mean_used, in units of the channel's std)")
    first = None
    for i, n in enumerate(names):
        d_bn = [rel(a[f"{i}/bn"][k], b[f"{i}/bn"][k]) for k in range(4)]
        d_fw = [abs(a[f"{i}/fw"][k] - b[f"{i}/fw"][k]) / a[f"{i}/fw"][k] for k in range(2)]
        M = float(a[f"{i}/M"])
        own = []
        for r in (a, b):
            mean_s = r[f"{i}/ysum"] / M
            var_s = np.maximum(r[f"{i}/ysq"] / M - mean_s ** 2, 1e-30)
            own.append(float(np.abs((mean_s - r[f"{i}/bn"][2]) / np.sqrt(var_s)).max()))
        flag = max(d_bn + d_fw) > 1e-3
        if flag and first is None:
            first = n
        print(f"{i:3d} {n:28s} bn {' '.join(f'{v:.1e}' for v in d_bn)}  |y| {d_fw[0]:.1e} |z| {d_fw[1]:.1e}   "
              f"own-mean offset fuse0 {own[0]:.1e} fuse1 {own[1]:.1e} {'  <-- > 1e-3' if flag else ''}")
    print("first forward divergence > 1e-3:", first)
    print("\nbackward order: rel. difference of |dz|, |dW|, dgamma, dbeta")
    firstb = None
    for i in reversed(range(len(names))):
        d = [abs(a[f"{i}/bw"][k] - b[f"{i}/bw"][k]) / max(a[f"{i}/bw"][k], 1e-30) for k in range(2)]
        d += [rel(a[f"{i}/gg"], b[f"{i}/gg"]), rel(a[f"{i}/gb"], b[f"{i}/gb"])]
        flag = max(d) > 1e-3
        if flag and firstb is None:
            firstb = names[i]
        print(f"{i:3d} {names[i]:28s} |dz| {d[0]:.1e} |dW| {d[1]:.1e} dgamma {d[2]:.1e} dbeta {d[3]:.1e}   "
              f"|dW| fuse0 {a[f'{i}/bw'][1]:.4f} fuse1 {b[f'{i}/bw'][1]:.4f}{'  <-- > 1e-3' if flag else ''}")
    print("first backward divergence > 1e-3:", firstb)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2], *(sys.argv[3:4]))
    else:
        main()
