#!/usr/bin/env python
"""The train step's ALGORITHMIC HBM bytes per kernel family at BASELINE.json configs[2] (B = 64 @ 640x640, bf16), from the plan
(Engine.algorithmic_bytes: every operand of every launch moved once; no GPU involved -- the plan is built at B = 2 on the CPU
executor's library and rescaled, activation bytes being proportional to the batch), next to the PMC-measured traffic of the last
committed profiles/r*_pmc_bench.json, and the forward BatchNorm passes a consumer-side fusion could drop.
Last section: per-launch roofline FLOORS of the train and inference plans -- sum over launches of max(bytes / 5 TB/s, FLOP / 2.5 PFLOP/s)
(5 TB/s = what a device copy reaches on these boxes, tools/read_bw.py; 2.5 PFLOP/s = dense bf16 MFMA peak).
usage: python tools/traffic_report.py > profiles/rNN_step_bytes.txt"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("Y5M_BWD_PW_MIN_M", str(200000 * 2 // 64))
import torch  # noqa: E402
from emu.harness import emulated  # noqa: E402
from yolov5m_amd import _lib, config  # noqa: E402
from yolov5m_amd.model import YOLOV5m  # noqa: E402
from yolov5m_amd.ultralytics_loss import ComputeLoss  # noqa: E402
from yolov5m_amd.utils.training_utils import NativeTrainStep  # noqa: E402

# PMC kernel name -> launch-list kinds it serves
PMC = {"bn_act_kernel": ["apply_fused", "apply"], "bn_bwd_reduce_kernel + bn_bwd_apply_kernel": ["bn_bwd(reduce + apply)", "bn_reduce"],
       "conv_pw_kernel + conv_igemm_kernel + conv_igemm_multi_kernel + conv_gemm8_kernel + conv_halo_kernel": ["conv_igemm"],
       "wgrad_kernel + wgrad_rows_kernel + unpack_wgrad_kernel": ["wgrad", "unpack"], "bwd_pw_kernel": ["bwd_pw"], "bwd_stem_kernel": ["bwd_stem"],
       "adam_kernel + sumsq_kernel": ["optimizer"], "pack_batched_kernel": ["pack_weights"], "s2d_input_kernel": ["input"],
       "upsample2x_kernel + upsample2x_bwd_kernel": ["upsample"],
       "sppf_colmax_kernel + sppf_rowmax_kernel + maxpool5_gather_kernel + maxpool5_argmax_kernel": ["pool"]}

with emulated():
    B = 2
    m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
    m.compute_dtype = "bf16"
    m.train()
    m.flatten_parameters()
    st = NativeTrainStep(m, ComputeLoss(m), nt_max=8 * B)
    eng = m._engine_for(torch.empty((B, 3, 640, 640), device="meta"))
    r = st.algorithmic_bytes(eng, B=64)
    print(f"# tools/traffic_report.py: algorithmic HBM bytes of one train step, B = 64 @ 640x640, bf16 (plan built at B = {B}, rescaled)")
    print(f"total {r['total_bytes'] / 1e9:.3f} GB = activations {r['activation_bytes'] / 1e9:.3f} + parameters / gradients / workspaces {r['parameter_bytes'] / 1e9:.3f}")
    print("lists: " + ", ".join(f"{k} {v / 1e9:.3f}" for k, v in r["lists"].items()))
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bench.json")))
    pmc, steps = ({}, 1)
    if files:
        d = json.load(open(files[-1]))
        pmc = d["kernels"]
        steps = pmc.get("adam_kernel", {}).get("launches", 1)
        print(f"measured column: {os.path.relpath(files[-1], ROOT)} ({steps} eager steps; plan-construction fills excluded)")
    print(f"\n{'launch-list kind':28s} {'launches':>8s} {'algorithmic GB':>15s}")
    for k, (b, n) in sorted(r["by_kind"].items(), key=lambda kv: -kv[1][0]):
        print(f"{k:28s} {n:8d} {b / 1e9:15.3f}")
    if pmc:
        print(f"\n{'kernels (PMC)':100s} {'measured GB/step':>17s} {'algorithmic':>12s} {'ratio':>6s}")
        tm = ta = 0.0
        for names, kinds in PMC.items():
            meas = sum(pmc[n]["hbm_bytes_per_launch"] * pmc[n]["launches"] for n in names.split(" + ") if n in pmc) / steps / 1e9
            alg = sum(r["by_kind"].get(k, [0, 0])[0] for k in kinds) / 1e9
            tm += meas
            ta += alg
            print(f"{names:100s} {meas:17.3f} {alg:12.3f} {meas / max(alg, 1e-9):6.2f}")
        print(f"{'sum of the rows above':100s} {tm:17.3f} {ta:12.3f} {tm / ta:6.2f}")
        # the conv family split by the kernel each launch dispatches to at B = 64 (y5m_conv_kernel_name on the rescaled arguments):
        # which conv kernel re-reads most (round 6)
        import ctypes
        import re
        L = _lib.lib()
        buf = ctypes.create_string_buffer(192)
        byk = {}
        for lst in (eng.fwd, eng.bwd):
            for fn, _ in lst:
                d = getattr(fn, "__defaults__", None)
                if getattr(fn, "kind", None) != "conv_igemm" or not d:
                    continue
                t = fn.traffic
                by = (t[0] + t[1]) * 64 / B + t[2] + t[3]
                if hasattr(d[0], "_length_"):
                    name = "conv_igemm_multi_kernel"
                else:
                    q = type(d[0])()
                    ctypes.memmove(ctypes.byref(q), ctypes.byref(d[0]), ctypes.sizeof(q))
                    q.B, q.M = q.B * 64 // B, q.M * 64 // B
                    _lib.check(L.y5m_conv_kernel_name(ctypes.byref(q), eng.dtype, buf, 192), "y5m_conv_kernel_name")
                    name = re.sub(r"<.*", "", buf.value.decode())
                e = byk.setdefault(name, [0.0, 0])
                e[0] += by; e[1] += 1
        print(f"\n{'conv + data-gradient launches by kernel (PMC name)':60s} {'launches':>8s} {'measured GB/step':>17s} {'algorithmic':>12s} {'ratio':>6s}")
        for name, (by, n) in sorted(byk.items(), key=lambda kv: -kv[1][0]):
            meas = pmc[name]["hbm_bytes_per_launch"] * pmc[name]["launches"] / steps / 1e9 if name in pmc else float("nan")
            print(f"{name:60s} {n:8d} {meas:17.3f} {by / 1e9:12.3f} {meas / (by / 1e9):6.2f}")
    # forward BatchNorm + SiLU passes (read y, write z: 4 B per element) whose output is read ONLY by 1x1 convolutions (+ their weight
    # gradients): the passes a consumer-side "normalise in the loader" fusion could drop -- each needs the loader change in every
    # kernel that reads the tensor (conv_pw / conv_gemm8 forward, wgrad_kernel or bwd_pw_kernel backward)
    by_z = {}
    for lay in eng.layers:
        by_z.setdefault(id(lay.z.buf), []).append(lay)
    cons = {}
    for lay in eng.layers + eng.heads:
        x = lay.x
        root = x.parent if x.parent is not None else x
        cons.setdefault(id(root.buf), []).append(lay)
    drop = tot = 0
    rows = []
    for lay in eng.layers:
        nb = lay.M * lay.cout * 2 * 2 * 64 // B
        tot += nb
        users = [c for c in cons.get(id(lay.z.buf), []) if c is not lay]
        # consumers of the (possibly wider, concatenated) buffer this layer writes into
        ok = bool(users) and all(getattr(c, "kk", 1) == 1 and getattr(c, "ss", 1) == 1 for c in users) and lay.res is None
        # residual users (a later bottleneck adds this tensor) and pool / upsample readers are not 1x1 convolutions
        if ok:
            rows.append((lay.name, nb, [c.name for c in users]))
            drop += nb
    print(f"\nforward BatchNorm + SiLU passes: {tot / 1e9:.3f} GB per step; output read only by 1x1 convolutions (upper bound of a loader fusion, "
          f"residual / pool / upsample readers not checked): {drop / 1e9:.3f} GB = {100.0 * drop / r['total_bytes']:.1f} % of the step")
    for name, nb, users in rows:
        print(f"  {name:28s} {nb / 1e9:6.3f} GB  -> {', '.join(users)}")


# ---- per-launch roofline floors ------------------------------------------------------------------------------------------------
HBM, MFMA = 5.0e12, 2.5e15
# the same floors against the guide's achievable 6.3 TB/s and the 8 TB/s peak bench.py divides by, side by side ("at
# its floor" is relative to the 5 TB/s a device copy reaches on these boxes)
BW_ALT = (6.3e12, 8.0e12)


def floor(B_plan, S, B, training):
    with emulated():
        mm = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
        mm.compute_dtype = "bf16"
        mm.train(training)
        mm.flatten_parameters()
        e = mm._engine_for(torch.empty((B_plan, 3, S, S), device="meta"))
        sc = B / B_plan
        tt = tb = tf = 0.0
        alt = [0.0, 0.0]
        by_kind = {}
        for lst in ([e.fwd, e.bwd] if training else [e.fwd]):
            for fn, _ in lst:
                t = getattr(fn, "traffic_sparse", None) or getattr(fn, "traffic", (0, 0, 0, 0))     # (the fused step's sparse head-gradient pack)
                by = (t[0] + t[1]) * sc + t[2] + t[3]
                fl, kind, d = 0.0, getattr(fn, "kind", ""), getattr(fn, "__defaults__", None)
                if kind == "conv_igemm" and d:
                    arr = list(d[0]) if hasattr(d[0], "_length_") else [d[0]]
                    fl = sum(2.0 * a.M * a.N * a.K for a in arr) * sc
                elif kind == "wgrad" and getattr(fn, "wa", None) is not None:
                    fl = 2.0 * fn.wa.M * fn.wa.N * fn.wa.th * fn.wa.tw * fn.wa.C * sc
                elif kind == "bwd_pw":
                    fl = 4.0 * fn.bp.M * fn.bp.N * fn.bp.C * sc
                tt += max(by / HBM, fl / MFMA)
                for i, bw in enumerate(BW_ALT):
                    alt[i] += max(by / bw, fl / MFMA)
                tb += by
                tf += fl
                r_ = by_kind.setdefault(kind or getattr(fn, "__name__", "other"), [0.0, 0.0, 0.0, 0, 0.0, 0.0])
                r_[0] += max(by / HBM, fl / MFMA); r_[1] += by; r_[2] += fl; r_[3] += 1
                r_[4] += max(by / BW_ALT[0], fl / MFMA); r_[5] += max(by / BW_ALT[1], fl / MFMA)
        floor.by_kind = by_kind
        floor.alt = alt
        return tt, tb, tf


print("\nper-launch roofline floors (sum over launches of max(bytes / 5 TB/s, FLOP / 2.5 PFLOP/s)); measured = round 3")
for Bp, S, Bq, tr, meas, what in ((2, 640, 64, True, 25.1, "train forward + backward (loss / optimizer not included), configs[2]"),
                                  (2, 640, 32, False, 3.69, "inference forward, configs[1]"), (1, 1280, 128, False, 50.3, "inference forward, configs[4]")):
    t_, b_, f_ = floor(Bp, S, Bq, tr)
    print(f"  B={Bq:3d} @ {S:4d} {what}: {b_ / 1e9:7.2f} GB, {f_ / 1e12:6.2f} TFLOP; floor {t_ * 1e3:6.2f} ms (pure HBM {b_ / HBM * 1e3:6.2f}, pure MFMA "
          f"{f_ / MFMA * 1e3:5.2f}); measured {meas} ms = {meas / (t_ * 1e3):.2f}x the floor"
          f" [floor @ 6.3 TB/s {floor.alt[0] * 1e3:.2f} ms = {meas / (floor.alt[0] * 1e3):.2f}x, @ 8.0 TB/s {floor.alt[1] * 1e3:.2f} ms = {meas / (floor.alt[1] * 1e3):.2f}x]")
    if tr:
        # per family: floor against the SERIALISED eager family times of the last committed bench line (forked stream run inline, so the
        # families add up; the graph-replayed step overlaps the weight gradients with the main stream and is shorter than their sum)
        lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_b64_640_line*.json")))
        fam = json.load(open(lines[-1]))["roofline"]["family_ms_per_step"] if lines else {}
        alias = {"apply_fused": ["apply_fused", "apply"], "head_pack": ["head_pack", "pack"]}
        print(f"    per family (floor from the plan; measured = serialised eager family time, {os.path.basename(lines[-1]) if lines else 'n/a'}):")
        print(f"    {'family':26s} {'launches':>8s} {'GB':>8s} {'TFLOP':>7s} {'floor ms':>9s} {'measured ms':>12s} {'x floor':>8s} {'x @6.3':>7s} {'x @8.0':>7s}   (floor = 5.0 TB/s)")
        for k, (ft, fb, ff, n, f63, f80) in sorted(floor.by_kind.items(), key=lambda kv: -kv[1][0]):
            meas_k = sum(fam.get(a, 0.0) for a in alias.get(k, [k]))
            if ft * 1e3 < 0.005 and not meas_k:
                continue
            print(f"    {k:26s} {n:8d} {fb / 1e9:8.2f} {ff / 1e12:7.2f} {ft * 1e3:9.2f} {meas_k:12.2f} {(meas_k / (ft * 1e3)) if ft > 0 and meas_k else 0:8.2f} {(meas_k / (f63 * 1e3)) if f63 > 0 and meas_k else 0:7.2f} {(meas_k / (f80 * 1e3)) if f80 > 0 and meas_k else 0:7.2f}")

