#!/usr/bin/env python
"""Achieved numerical errors of the HIP path against the float64 reference (tests/golden/g13_precision.npz), next to the
reference's own float32 error, and of the bf16 path against the quantisation-aware oracle. Prints one line per check."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yolov5m_amd import config
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.ultralytics_loss import ComputeLoss
from yolov5m_amd.utils.synth import synth_images, synth_labels, synth_state_dict
from oracle import model_ref, loss_ref
G = lambda n: np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", n + ".npz"))
g13, g5, g7 = G("g13_precision"), G("g5_model"), G("g7_large_step")
DEV = "cuda"

def model(dt):
    m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
    m.load_state_dict(synth_state_dict(), strict=True)
    m = m.to(DEV); m.compute_dtype = dt
    return m

for tag, (B, H, W, seed) in {"s64": (1, 64, 64, None), "s96x128": (2, 96, 128, None), "s320": (2, 320, 320, None),
                             "b16_320": (16, 320, 320, "img/rank0")}.items():
    x = (synth_images(B, H, W) if seed is None else synth_images(B, H, W, seed=seed)).to(DEV)
    m = model("f32"); m.train()
    with torch.no_grad():
        o = m(x)
    for i in range(3):
        r64 = g13[f"{tag}/train64/o{i}_sample"]; step = int(g13[f"{tag}/train64/o{i}_step"])
        hip = o[i].reshape(-1).cpu().numpy()[::step][:4096]
        r32 = g7[f"o{i}_sample"] if tag == "b16_320" else g5[f"{tag}/train/o{i}_sample"]
        sc = np.abs(r64).max()
        print(f"logits {tag} o{i}: hip-f64 {np.abs(hip - r64).max() / sc:.2e}  ref32-f64 {np.abs(r32 - r64).max() / sc:.2e}  hip-ref32 {np.abs(hip - r32).max() / sc:.2e}")
# full gradients
B, H, W = [int(v) for v in g13["grad/shape"]]
x = synth_images(B, H, W, seed="img/rank0").to(DEV)
t = torch.from_numpy(g13["grad/targets"])
m = model("f32"); m.train()
loss = ComputeLoss(m)(m(x), t, None); loss.backward()
print("loss hip", float(loss), "ref32", float(g13["grad/f32/loss"]), "ref64", float(g13["grad/f64/loss"]))
names = list(g13["grad/names"]); named = dict(m.named_parameters())
worst = []
for j, k in enumerate(names):
    gh = named[k].grad.reshape(-1).double().cpu().numpy()
    step = max(1, gh.size // 256); idx = np.arange(0, gh.size, step)[:256]
    s64 = g13["grad/f64/sample"][j][:idx.size]; s32 = g13["grad/f32/sample"][j][:idx.size].astype(np.float64)
    sc = np.abs(s64).max() + 1e-30
    eh, er = np.abs(gh[idx] - s64).max() / sc, np.abs(s32 - s64).max() / sc
    nh = np.sqrt((gh * gh).sum()); n64 = g13["grad/f64/norm"][j]; n32 = g13["grad/f32/norm"][j]
    worst.append((eh, er, abs(nh - n64) / n64, abs(n32 - n64) / n64, k))
worst.sort(reverse=True)
print("grad samples: max hip-f64 %.2e (ref32-f64 there %.2e) [%s]; median hip %.2e ref %.2e" % (worst[0][0], worst[0][1], worst[0][4], np.median([w[0] for w in worst]), np.median([w[1] for w in worst])))
print("grad norms: max hip-f64 %.2e, max ref32-f64 %.2e" % (max(w[2] for w in worst), max(w[3] for w in worst)))
print("tensors with hip err > 1e-4:", sum(w[0] > 1e-4 for w in worst), " ref32 err > 1e-4:", sum(w[1] > 1e-4 for w in worst), "of", len(worst))
for w in worst[:6]: print("   %.2e %.2e %s" % (w[0], w[1], w[4]))
# bf16 vs quantisation-aware oracle
torch.set_num_threads(64)
x = synth_images(16, 320, 320, seed="img/rank0"); t = torch.from_numpy(g13["grad/targets"])
sd = synth_state_dict()
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k and "anchors" not in k}
for quant in (True, False):
    full = dict(sd); full.update(params)
    for p in params.values(): p.grad = None
    out = model_ref.forward(full, x, training=True, quant=quant)
    l, _ = loss_ref.compute_loss_ultra(out, t, sd["head.anchors"])
    l.backward()
    if quant: oq, lq, gq = [o.detach() for o in out], float(l), {k: p.grad.clone() for k, p in params.items()}
    else: of, lf_, gf = [o.detach() for o in out], float(l), {k: p.grad.clone() for k, p in params.items()}
m = model("bf16"); m.train()
o = m(x.to(DEV)); loss = ComputeLoss(m)(o, t, None); loss.backward()
print("bf16 loss hip", float(loss), "quant-oracle", lq, "f32-oracle", lf_)
for i in range(3):
    a = o[i].detach().cpu()
    print(f"bf16 logits o{i}: hip-quant {float((a - oq[i]).norm() / oq[i].norm()):.3e}  quant-f32 {float((oq[i] - of[i]).norm() / of[i].norm()):.3e}  hip-f32 {float((a - of[i]).norm() / of[i].norm()):.3e}")
named = dict(m.named_parameters())
eq, ef = [], []
for k in gq:
    gh = named[k].grad.detach().cpu()
    eq.append((float((gh - gq[k]).norm() / (gq[k].norm() + 1e-30)), k)); ef.append(float((gq[k] - gf[k]).norm() / (gf[k].norm() + 1e-30)))
eq.sort(reverse=True)
print("bf16 grads rel-L2 hip-quant: max %.3e [%s] median %.3e ; quant-f32: max %.3e median %.3e" % (eq[0][0], eq[0][1], np.median([e[0] for e in eq]), max(ef), np.median(ef)))
tot_h = torch.sqrt(sum((named[k].grad.detach().cpu().double() ** 2).sum() for k in gq)); tot_q = torch.sqrt(sum((gq[k].double() ** 2).sum() for k in gq))
print("bf16 total grad norm hip %.5g quant %.5g" % (float(tot_h), float(tot_q)))
for e in eq[:5]: print("   %.3e %s" % e)
