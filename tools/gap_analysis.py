#!/usr/bin/env python
"""GPU idle time inside graph replays: union of kernel [start,end] intervals vs wall, from a rocprofv3 kernel trace CSV"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# steps are delimited by adam_kernel launches
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
print("steps found:", len(adam))
for k in range(len(adam) - 6, len(adam) - 1):
    seg = ev[adam[k] + 1: adam[k + 1] + 1]
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _ in seg)
    print(f"step {k}: wall {(t1-t0)/1e6:.3f} ms, busy(union) {busy/1e6:.3f} ms, idle {(t1-t0-busy)/1e6:.3f} ms, kernels {len(seg)}, sum of durations {ksum/1e6:.3f} ms")
# gap histogram of the last step
seg = ev[adam[-2] + 1: adam[-1] + 1]
gaps = []
cur_e = seg[0][1]
for s, e, n in seg[1:]:
    if s > cur_e: gaps.append((s - cur_e, n))
    cur_e = max(cur_e, e)
gaps.sort(reverse=True)
print("largest gaps (us):", [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:12]])
print("number of gaps:", len(gaps), "mean gap us:", round(sum(g for g, _ in gaps) / max(len(gaps), 1) / 1e3, 2))
