#!/bin/bash
# BASELINE config C2 alone: the VALU side of the MSM's roofline -- SQ_INSTS_VALU / SQ_WAVES per kernel of tools/c2_rate.py on ONE lane (kernels do not overlap),
# own --pmc pass with --kernel-trace only; prints instructions, waves, duration and cycles per wave-instruction per SIMD of every MSM kernel
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_c2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/sq -o s -- python $GRAFT_REPO_ROOT/tools/c2_rate.py 1 24 > $OUT/c2_sq.log 2>&1
cd $GRAFT_REPO_ROOT
python - $OUT/sq <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/s_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    if r["Counter_Name"] == "SQ_WAVES": dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("| kernel | launches | us / launch | waves / launch | M VALU instr / launch | cycles per wave-instr per SIMD (2.4 GHz) |\n|---|---|---|---|---|---|")
for k in sorted(dur, key=dur.get, reverse=True)[:10]:
    L = len(n[k]); us = dur[k] / L; iv = agg[k]["SQ_INSTS_VALU"] / L
    print(f"| {k} | {L} | {us:.1f} | {agg[k]['SQ_WAVES'] / L:.0f} | {iv / 1e6:.2f} | {us * 1e-6 * 2.4e9 * 1024 / iv if iv else 0:.2f} |")
PY
