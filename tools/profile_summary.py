#!/usr/bin/env python3
"""Turn one tools/profile_round.sh output directory into the markdown tables kept under profiles/.
usage: profile_summary.py gpurun_out/prof_<tag>
 * kernel-trace stats: rocprofv3's own *_kernel_stats.csv
 * PMC passes: FETCH_SIZE / WRITE_SIZE (KiB) per launch, averaged over the launches of each kernel's most frequent grid
   size (bench.py also launches small variable-base MSMs in its folded-batch probe; mixing them in would understate the
   per-launch traffic of the 2^16 fixed-base shape)."""
import collections
import csv
import glob
import os
import sys


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


def stats_table(path):
    rows = list(csv.DictReader(open(path)))
    out = ["| kernel | calls | total_ms | avg_us | min_us | max_us | % |", "|---|---|---|---|---|---|---|"]
    for r in rows[:20]:
        out.append(f"| {short(r['Name'])} | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
                   f"{int(r['MinNs']) / 1e3:.1f} | {int(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
    return "\n".join(out)


def pmc(path):
    per = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> grid -> values
    for r in csv.DictReader(open(path)):
        per[short(r["Kernel_Name"])][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    res = {}
    for k, grids in per.items():
        g = max(grids, key=lambda g: (len(grids[g]), g))         # the most frequent grid; ties and near-ties aside, the set-up's smaller launches must not win:
        big = max(grids)                                          # prefer the LARGEST grid when it was launched at least 10 times (the steps' launches; round 5: the set-up hashes
        if len(grids[big]) >= 10: g = big                         # 16 384 chains state by state -- 17 small launches of pstate_hash_kernel against 14 of the step's)
        v = grids[g]
        res[k] = (sum(v) / len(v), len(v), g)
    return res


def main():
    d = sys.argv[1]
    st = glob.glob(os.path.join(d, "trace", "*kernel_stats.csv"))
    print("## kernel-trace stats\n")
    print(stats_table(st[0]) if st else "(no kernel stats found)")
    f = glob.glob(os.path.join(d, "fetch", "*counter_collection.csv"))
    w = glob.glob(os.path.join(d, "write", "*counter_collection.csv"))
    if f and w:
        F, W = pmc(f[0]), pmc(w[0])
        print("\n## PMC passes, KiB per launch (most frequent grid size of each kernel)\n")
        print("| kernel | grid | launches | FETCH_SIZE | WRITE_SIZE | bytes/launch |\n|---|---|---|---|---|---|")
        tot = {k: (F[k][0] + W.get(k, (0,))[0]) * 1024 for k in F}
        for k in sorted(tot, key=tot.get, reverse=True)[:12]:
            print(f"| {k} | {F[k][2]} | {F[k][1]} | {F[k][0]:.0f} | {W.get(k, (0,))[0]:.0f} | {tot[k] / 1e6:.1f} MB |")


if __name__ == "__main__":
    main()
