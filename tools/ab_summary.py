#!/usr/bin/env python
"""Summary of tools/ab_step.sh output (lines `round R  NAME  [ENV]  X ms/step`): per variant the rounds' values, their mean and
minimum, and the difference of the means to the FIRST variant of the file (the baseline of that A/B). A variant is a win when its
mean is lower than the baseline's by more than the baseline's own spread (max - min over its rounds) -- the rule the round-4/5
knobs are decided by (what is not faster inside the step stays off).
usage: python tools/ab_summary.py gpurun_out/ab_step/ab.txt [more files]"""
import re
import sys


def main():
    for path in sys.argv[1:]:
        rows = {}
        order = []
        for line in open(path):
            m = re.match(r"round\s+(\d+)\s+(\S+)\s+\[(.*?)\]\s+([\d.]+)\s+ms/step", line.strip())
            if not m:
                continue
            name = m.group(2)
            if name not in rows:
                rows[name] = (m.group(3), [])
                order.append(name)
            rows[name][1].append(float(m.group(4)))
        if not order:
            print(f"{path}: no `round .. ms/step` lines")
            continue
        base = rows[order[0]][1]
        bmean, bspread = sum(base) / len(base), max(base) - min(base)
        print(f"# {path}: baseline `{order[0]}` mean {bmean:.3f} ms/step, spread {bspread:.3f}")
        print(f"{'variant':16s} {'env':34s} {'values':28s} {'mean':>8s} {'min':>8s} {'d(mean)':>9s}  verdict")
        for name in order:
            env, vals = rows[name]
            mean = sum(vals) / len(vals)
            d = mean - bmean
            verdict = "baseline" if name == order[0] else ("WIN" if d < -max(bspread, 0.02) else ("loss" if d > max(bspread, 0.02) else "noise"))
            print(f"{name:16s} {env[:34]:34s} {' '.join(f'{v:.3f}' for v in vals)[:28]:28s} {mean:8.3f} {min(vals):8.3f} {d:+9.3f}  {verdict}")


if __name__ == "__main__":
    main()
