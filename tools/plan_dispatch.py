#!/usr/bin/env python
"""Which kernel instantiation every conv of the YOLOv5m plan dispatches to at a given (batch, size, mode), with its FLOPs -- from the
library's name queries (y5m_conv_kernel_name runs the dispatch and launches nothing: no GPU needed). Forward launches (statistics
epilogue in training, folded BatchNorm + SiLU in inference) and, for training, the stride-1 data gradients. Made after this listing
showed (round 5) that BASELINE configs[4]'s largest layer group ran on the tiled kernel because the halo kernel's LDS image did not fit.
usage: python tools/plan_dispatch.py [B S mode]...     e.g.  tools/plan_dispatch.py 64 640 train 32 640 eval 128 1280 eval"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dispatch_cpu as T  # noqa: E402
from yolov5m_amd._lib import EPI_DGRAD, EPI_RAW_STATS  # noqa: E402
from yolov5m_amd.arch import blocks  # noqa: E402

EPI_AFFINE_ACT = 1


def layers(S):
    """(tag, Cin, Cout, k, stride, input size) of every CBL launch of the plan (C3's c1 + c_skipped as the merged pair)"""
    def c3(cin, cout, width, depth, hw):
        c_ = int(width * cin)
        L = [("c3.pair", cin, 2 * c_, 1, 1, hw)]
        for _ in range(depth):
            L += [("c3.seq.1x1", c_, c_, 1, 1, hw), ("c3.seq.3x3", c_, c_, 3, 1, hw)]
        return L + [("c3.c_out", 2 * c_, cout, 1, 1, hw)]
    bb, _ = blocks(48)
    out, hw = [], S
    for idx, (kind, a) in enumerate(bb):
        if kind == "cbl":
            if idx == 0:
                out.append(("stem(3x3 on s2d)", 16, 48, 3, 1, hw // 2))
            else:
                out.append(("down 3x3 s2", a["cin"], a["cout"], 3, 2, hw))
            hw //= 2
        elif kind == "c3":
            out += c3(a["cin"], a["cout"], a["width"], a["depth"], hw)
        else:
            out += [("sppf.c1", a["cin"], a["cin"] // 2, 1, 1, hw), ("sppf.c_out", 2 * a["cin"], a["cout"], 1, 1, hw)]
    h = hw
    out.append(("neck.0", 768, 384, 1, 1, h)); out += c3(768, 384, 0.25, 2, 2 * h)
    out.append(("neck.2", 384, 192, 1, 1, 2 * h)); out += c3(384, 192, 0.25, 2, 4 * h)
    out.append(("neck.4 3x3 s2", 192, 192, 3, 2, 4 * h)); out += c3(384, 384, 0.5, 2, 2 * h)
    out.append(("neck.6 3x3 s2", 384, 384, 3, 2, 2 * h)); out += c3(768, 768, 0.5, 2, h)
    return out


def report(B, S, mode):
    tot, cnt, det = collections.defaultdict(float), collections.Counter(), collections.defaultdict(list)
    for tag, cin, cout, k, s, hin in layers(S):
        ho = hin // s
        fl = 2.0 * B * ho * ho * cout * cin * k * k / 1e9
        key = ("fwd", T._conv_name(B, cin, hin, hin, cout, k, s, EPI_RAW_STATS if mode == "train" else EPI_AFFINE_ACT))
        tot[key] += fl; cnt[key] += 1; det[key].append(f"{tag} {cin}->{cout} @{hin}")
        if mode == "train" and s == 1 and not tag.startswith("stem"):
            key = ("dgrad", T._conv_name(B, cout, ho, ho, cin, k, 1, EPI_DGRAD))
            tot[key] += fl; cnt[key] += 1; det[key].append(f"{tag} {cin}->{cout} @{hin}")
    total = sum(tot.values())
    print(f"\n== B = {B} @ {S}x{S}, {mode}: {total / 1e3:.2f} TFLOP in the listed launches"
          + (" (stride-2 data gradients = conv_igemm_multi_kernel, eligible 1x1 backward = bwd_pw_kernel: not listed)" if mode == "train" else ""))
    for key, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"  {key[0]:5s} {key[1]:42s} {cnt[key]:3d} launches {v:9.1f} GFLOP {100 * v / total:5.1f} %")
        if "igemm" in key[1]:
            seen = collections.Counter(det[key])
            print("          tiled: " + "; ".join(f"{n} x {d}" for d, n in seen.items()))


if __name__ == "__main__":
    a = sys.argv[1:] or ["64", "640", "train", "32", "640", "eval", "128", "1280", "eval"]
    print("# tools/plan_dispatch.py: conv launches -> kernel instantiation (library name queries; environment knobs apply: "
          + (", ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("Y5M_")) or "defaults") + ")")
    for i in range(0, len(a), 3):
        report(int(a[i]), int(a[i + 1]), a[i + 2])
