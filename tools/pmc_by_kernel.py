"""per-kernel sums of whatever counters a set of `rocprofv3 --pmc` passes collected: usage  python tools/pmc_by_kernel.py DIR [DIR ...]   (each DIR holds one pass's *counter_collection.csv)
Prints one row per kernel: launches, and per counter the total and the value per launch -- tools/gpu_round6.sh tcc reads the accumulate kernels' L2 hit rate from it."""
import collections, csv, glob, os, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").replace("mb::", "").split("(")[0]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
ctrs = sorted({c for k in tot for c in tot[k]})
print("| kernel | launches | " + " | ".join(ctrs) + " |\n|---|---|" + "---|" * len(ctrs))
for k in sorted(tot, key=lambda k: -max(tot[k].values())):
    print(f"| {k} | {max(n[k].values())} | " + " | ".join(f"{tot[k].get(c, 0):.4g}" for c in ctrs) + " |")
hit = {k: (tot[k].get("TCC_HIT_sum", 0), tot[k].get("TCC_MISS_sum", 0)) for k in tot if "TCC_HIT_sum" in tot[k]}
if hit:
    print("\n| kernel | L2 (TCC) hit rate |\n|---|---|")
    for k, (h, m) in sorted(hit.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:12]:
        if h + m: print(f"| {k} | {h / (h + m):.3f} ({h:.3g} hits, {m:.3g} misses) |")
