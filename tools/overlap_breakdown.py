#!/usr/bin/env python
"""Where the graph-replayed step's wall time goes, from a rocprofv3 --kernel-trace CSV: time with only main-stream kernels running,
only forked weight-gradient kernels (wgrad_kernel / unpack_wgrad_kernel), both, or nothing; and the main stream's largest idle
windows (what it was waiting for). usage: overlap_breakdown.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
side = lambda n: "wgrad_kernel" in n or "unpack_wgrad" in n
tot = {"main only": 0, "side only": 0, "both": 0, "idle": 0}
nsteps = 0
for k in range(len(adam) - 6, len(adam) - 1):
    seg = ev[adam[k] + 1: adam[k + 1] + 1]
    pts = []
    for s, e, n in seg:
        w = 1 if side(n) else 0
        pts.append((s, 0, w)); pts.append((e, 1, w))
    pts.sort()
    cnt = [0, 0]
    last = pts[0][0]
    for t, kind, w in pts:
        dt = t - last
        key = "both" if cnt[0] and cnt[1] else "main only" if cnt[0] else "side only" if cnt[1] else "idle"
        tot[key] += dt
        last = t
        cnt[w] += 1 if kind == 0 else -1
    nsteps += 1
print(f"average over {nsteps} replayed steps (ms): " + ", ".join(f"{k} {v / nsteps / 1e6:.3f}" for k, v in tot.items()),
      f"| wall {sum(tot.values()) / nsteps / 1e6:.3f}")
# the last step: windows where NO main-stream kernel runs, and what starts next on the main stream
seg = ev[adam[-2] + 1: adam[-1] + 1]
main = [(s, e, n) for s, e, n in seg if not side(n)]
gaps, cur_e = [], main[0][1]
for s, e, n in main[1:]:
    if s > cur_e:
        running = [m for m in seg if side(m[2]) and m[0] < s and m[1] > cur_e]
        gaps.append((s - cur_e, n, len(running)))
    cur_e = max(cur_e, e)
gaps.sort(reverse=True)
print(f"main-stream idle windows in the last step: {len(gaps)} totalling {sum(g[0] for g in gaps) / 1e6:.3f} ms; > 5 us: {sum(1 for g in gaps if g[0] > 5000)} totalling {sum(g[0] for g in gaps if g[0] > 5000) / 1e6:.3f} ms")
for g, n, r in gaps[:15]:
    print(f"   {g / 1e3:8.1f} us before {n[:70]}  ({r} forked kernels running meanwhile)")
