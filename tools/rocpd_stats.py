#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database into a per-kernel stats table (calls, total, avg, %)
-- the same content as `rocprofv3 --stats`'s kernel_stats CSV, for committing under profiles/."""
import re
import sqlite3
import sys


def main(path, out=None, skip_first_steps=0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in kcols else ("kernel_name" if "kernel_name" in kcols else "name")
    q = (f"select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d "
         f"join rocpd_info_kernel_symbol s on d.kernel_id = s.id")
    rows = cur.execute(q).fetchall()
    agg = {}
    for name, st, en in rows:
        name = re.sub(r"\s+", " ", str(name))
        name = re.sub(r"\(.*$", "", name) if len(name) > 120 else name
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        dur = en - st
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"\"{name}\",{a[0]},{a[1]},{a[1] / a[0]:.1f},{100.0 * a[1] / tot:.2f},{a[2]},{a[3]}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
