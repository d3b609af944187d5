#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into the same table `--stats` prints:
per-kernel calls / total / average / min / max duration.  Usage: rocpd_summary.py results.db [out.md]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    short = re.sub(r"\(.*", "", name)
    short = re.sub(r"^void ", "", short)
    a = agg.setdefault(short, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values()) or 1
lines = ["| kernel | calls | total_us | avg_us | min_us | max_us | % |", "|---|---|---|---|---|---|---|"]
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"| {k} | {a[0]} | {a[1]/1e3:.1f} | {a[1]/a[0]/1e3:.2f} | {a[2]/1e3:.2f} | {a[3]/1e3:.2f} | {100*a[1]/tot:.1f} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "a").write(out + "\n")
