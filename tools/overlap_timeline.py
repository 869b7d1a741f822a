#!/usr/bin/env python
"""Where the weight-gradient stream runs relative to the main chain inside one graph-replayed step, from a rocprofv3
kernel-trace CSV: per 1-ms bin of the step, the busy time of main-chain kernels and of the weight-gradient kernels, plus the
exposed tail (time after the last main-chain kernel during which only weight-gradient work runs)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
k = len(adam) - 2
seg = ev[adam[k] + 1: adam[k + 1] + 1]
t0 = seg[0][0]
side = lambda n: "wgrad_kernel" in n or "unpack_wgrad" in n
main = [(s - t0, e - t0, n) for s, e, n in seg if not side(n)]
sd = [(s - t0, e - t0, n) for s, e, n in seg if side(n)]
T = max(e for _, e, _ in main + sd)
print(f"step wall {T/1e6:.3f} ms; main kernels {len(main)} sum {sum(e-s for s,e,_ in main)/1e6:.2f} ms; wgrad-stream kernels {len(sd)} sum {sum(e-s for s,e,_ in sd)/1e6:.2f} ms")
first_bwd = min(s for s, e, n in sd)
print(f"first weight gradient starts at {first_bwd/1e6:.2f} ms, last one ends at {max(e for _,e,_ in sd)/1e6:.2f} ms")
# main-chain kernels excluding the optimizer tail (sumsq/adam): last kernel that is part of backward
bw = [x for x in main if not any(t in x[2] for t in ("adam_kernel", "sumsq"))]
print(f"last non-optimizer main kernel ends at {max(e for _,e,_ in bw)/1e6:.2f} ms ({max(bw, key=lambda x: x[1])[2][:50]})")
def busy(evs, a, b):
    return sum(max(0, min(e, b) - max(s, a)) for s, e, _ in evs)
print(" ms   main-busy  wgrad-busy (fraction of the bin; >1 = several kernels at once)")
step = 1_000_000
t = 0
while t < T:
    print(f"{t/1e6:4.0f}   {busy(main, t, t+step)/step:8.2f}  {busy(sd, t, t+step)/step:8.2f}")
    t += step
# exposed: time when NO main kernel runs but a wgrad-stream kernel does
pts = sorted(set([0, T] + [x for s, e, _ in main + sd for x in (s, e)]))
exp = idle = 0
for a, b in zip(pts, pts[1:]):
    m = any(s <= a and e >= b for s, e, _ in main)
    w = any(s <= a and e >= b for s, e, _ in sd)
    if not m and w: exp += b - a
    if not m and not w: idle += b - a
print(f"only weight-gradient work running: {exp/1e6:.2f} ms; nothing running: {idle/1e6:.2f} ms")
