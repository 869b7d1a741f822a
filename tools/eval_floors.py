#!/usr/bin/env python
"""Per-launch roofline floors of the two INFERENCE plans BASELINE.json names -- configs[1] (B = 32 @ 640x640) and configs[4] (B = 128 @
1280x1280), bf16, BatchNorm folded -- from the plan (no GPU): every launch-list entry's algorithmic HBM bytes (Engine._traffic) and FLOPs,
grouped by (kernel instantiation, shape), floor = max(bytes / BW, FLOP / 2.5 PFLOP/s) with BW = 5.0 TB/s (what a device copy reaches on the
pool's boxes, tools/read_bw.py), 6.3 TB/s (the guide's achievable figure) and 8.0 TB/s (the HBM3E peak bench.py divides by), side by side
(VERDICT r5 weak 9 / 10, next 6). No per-launch timing of an eval plan was ever recorded (tools/eval_layers.py prints one on a GPU; its output
was not kept), so the `est ms` column is an ESTIMATE: the round-3 duration of the TRAIN-mode forward launch of the same layer at B = 64 @
640x640 (profiles/r03_layers_b64_640.txt: same kernel family, statistics epilogue instead of the folded BatchNorm + SiLU one) scaled by the
pixel ratio. The measured totals (3.69 ms, 50.3 ms: profiles/r03_bench_b64_640_line_final.json) say how far the estimate can be trusted.
usage: python tools/eval_floors.py > profiles/rNN_eval_floors.txt"""
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from emu.harness import emulated  # noqa: E402
from yolov5m_amd import _lib, config  # noqa: E402
from yolov5m_amd.model import YOLOV5m  # noqa: E402

BWS = (5.0e12, 6.3e12, 8.0e12)
MFMA = 2.5e15


def r03_forward_times():
    """(M at B = 64 @ 640, N, K, taps, stride) -> ms per launch of the train-mode forward conv (epi=0 rows), round 3"""
    out = {}
    p = os.path.join(ROOT, "profiles", "r03_layers_b64_640.txt")
    if not os.path.exists(p):
        return out
    for ln in open(p):
        m = re.match(r"\s*([\d.]+) ms x\s*(\d+)\s+conv\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) taps=(\S+)(?: x\d+)? s=(\d) epi=(\d)", ln)
        if m and m.group(8) in ("0", "2"):              # statistics epilogue (CBL forward) | bias epilogue (head)
            ms, n, M, N, K, taps, s = float(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), m.group(6), int(m.group(7))
            out[(M, N, K, taps, s)] = ms / n
    return out


def plan_rows(Bp, S, B):
    with emulated():
        mm = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768))
        mm.compute_dtype = "bf16"
        mm.eval()
        mm.flatten_parameters()
        e = mm._engine_for(torch.empty((Bp, 3, S, S), device="meta"))
        L = _lib.lib()
        sc = B / Bp
        buf = ctypes.create_string_buffer(192)
        rows = {}
        for fn, _ in list(e.pack) + list(e.fwd):
            t = getattr(fn, "traffic", (0, 0, 0, 0))
            by = (t[0] + t[1]) * sc + t[2] + t[3]
            kind, d = getattr(fn, "kind", getattr(fn, "__name__", "other")), getattr(fn, "__defaults__", None)
            fl, key = 0.0, (kind,)
            if kind == "conv_igemm" and d and isinstance(d[0], _lib.ConvArgs):
                a = d[0]
                q = type(a)()                              # the dispatch depends on the launch size: ask for the REPORTED batch, not the plan's
                ctypes.memmove(ctypes.byref(q), ctypes.byref(a), ctypes.sizeof(a))
                q.B, q.M = int(a.B * sc), int(a.M * sc)
                _lib.check(L.y5m_conv_kernel_name(ctypes.byref(q), e.dtype, buf, 192), "y5m_conv_kernel_name")
                fl = 2.0 * a.M * a.N * a.K * sc
                key = (buf.value.decode(), int(a.M * sc), a.N, a.K, f"{a.th}x{a.tw}", a.sy)
            r = rows.setdefault(key, [0, 0.0, 0.0])
            r[0] += 1; r[1] += by; r[2] += fl
        return rows


def main():
    t03 = r03_forward_times()
    for Bp, S, B, meas, what in ((2, 640, 32, 3.69, "configs[1]: inference forward B = 32 @ 640x640"),
                                 (1, 1280, 128, 50.3, "configs[4]: inference forward B = 128 @ 1280x1280")):
        rows = plan_rows(Bp, S, B)
        tb = sum(r[1] for r in rows.values()); tf = sum(r[2] for r in rows.values())
        fl = [sum(max(r[1] / bw, r[2] / MFMA) for r in rows.values()) * 1e3 for bw in BWS]
        print(f"\n## {what}: {sum(r[0] for r in rows.values())} launches, {tb / 1e9:.2f} GB, {tf / 1e12:.2f} TFLOP; measured (round 3) {meas} ms")
        print(f"   floor (sum over launches of max(bytes / BW, FLOP / 2.5 PF)): {fl[0]:.2f} ms @ 5.0 TB/s ({meas / fl[0]:.2f}x), {fl[1]:.2f} ms @ 6.3 TB/s ({meas / fl[1]:.2f}x), "
              f"{fl[2]:.2f} ms @ 8.0 TB/s ({meas / fl[2]:.2f}x); pure MFMA {tf / MFMA * 1e3:.2f} ms")
        print(f"   {'launch group':40s} {'M':>9s} {'N':>4s} {'K':>5s} {'taps':>4s} {'s':>1s} {'n':>3s} {'GB':>7s} {'TFLOP':>6s} {'bound':>5s} | floor ms @ {'5.0':>6s} {'6.3':>6s} {'8.0':>6s} | {'est ms':>7s} {'x 5.0':>6s} {'x 8.0':>6s}")
        est_tot = est_floor5 = 0.0
        out = []
        scale_pix = (S / 640.0) ** 2 * B / 64.0
        for key, (n, by, f) in rows.items():
            fls = [max(by / bw, f / MFMA) * 1e3 for bw in BWS]
            est = None
            if len(key) == 6:
                name, M, N, K, taps, s = key
                M64 = int(round(M / scale_pix))
                t = t03.get((M64, N, K, taps, s))
                if t is not None:
                    est = t * scale_pix * n
            out.append((fls[0], key, n, by, f, fls, est))
        for f5, key, n, by, f, fls, est in sorted(out, key=lambda r: -r[0]):
            if f5 < 0.002 * fl[0]:
                continue
            if len(key) == 6:
                name, M, N, K, taps, s = key
                lab = f"   {name:40s} {M:9d} {N:4d} {K:5d} {taps:>4s} {s:1d}"
            else:
                lab = f"   {key[0]:40s} {'':9s} {'':4s} {'':5s} {'':4s} {'':1s}"
            bound = "mfma" if f / MFMA > by / BWS[0] else "hbm"
            e = f"{est:7.3f} {est / fls[0]:6.2f} {est / fls[2]:6.2f}" if est is not None else f"{'-':>7s} {'':6s} {'':6s}"
            if est is not None:
                est_tot += est; est_floor5 += fls[0]
            print(f"{lab} {n:3d} {by / 1e9:7.3f} {f / 1e12:6.2f} {bound:>5s} |            {fls[0]:6.3f} {fls[1]:6.3f} {fls[2]:6.3f} | {e}")
        print(f"   launches with an estimate: floor {est_floor5:.2f} ms @ 5.0 TB/s, estimated {est_tot:.2f} ms ({est_tot / max(est_floor5, 1e-9):.2f}x); "
              f"all launches: floor {fl[0]:.2f}, measured {meas} -- the estimate covers {est_tot / meas * 100:.0f} % of the measured time")


if __name__ == "__main__":
    main()
