#!/bin/bash
# BASELINE config C2 alone: the VALU side of the MSM's roofline -- SQ_INSTS_VALU / SQ_WAVES per kernel of tools/c2_rate.py with the 16-LANE kernel forms (round 5: the timed
# run uses them; round 4 counted the single-lane forms, whose quad-cooperative bucket reduction issues a third more instructions -- under --pmc the dispatches are serialised either way),
# own --pmc pass with --kernel-trace only; prints instructions, waves, duration and cycles per wave-instruction per SIMD of every MSM kernel
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_c2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $OUT/sq -o s -- python $GRAFT_REPO_ROOT/tools/c2_rate.py 16 24 > $OUT/c2_sq.log 2>&1
cd $GRAFT_REPO_ROOT
python - $OUT/sq $TAG <<'PY'
import collections, csv, glob, json, os, sys
f = glob.glob(sys.argv[1] + "/**/s_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    if r["Counter_Name"] == "SQ_WAVES": dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
CALLS = 32 + 24                                              # tools/c2_rate.py 16 24: 32 warm-up calls + 24 timed, 8 checks each
per_call = {k: v for k, v in agg.items() if len(n[k]) % CALLS == 0 and len(n[k]) >= CALLS}      # one-time kernels (SRS tables, accumulator minting) are not part of a check
print("| kernel | launches | us / launch | waves / launch | M VALU instr / launch | cycles per wave-instr per SIMD (2.4 GHz) |\n|---|---|---|---|---|---|")
for k in sorted(per_call, key=lambda k: dur[k], reverse=True)[:12]:
    L = len(n[k]); us = dur[k] / L; iv = agg[k]["SQ_INSTS_VALU"] / L
    print(f"| {k} | {L} | {us:.1f} | {agg[k]['SQ_WAVES'] / L:.0f} | {iv / 1e6:.2f} | {us * 1e-6 * 2.4e9 * 1024 / iv if iv else 0:.2f} |")
total = sum(v["SQ_INSTS_VALU"] for v in per_call.values()) / CALLS / 8
out = {"source": f"tools/profile_c2_sq.sh {sys.argv[2]}: rocprofv3 --pmc SQ_INSTS_VALU, tools/c2_rate.py 16 24 (the 16-lane kernel forms of the timed run; 8 un-folded 2^16 Vesta accumulator checks per call, {CALLS} calls; one-time kernels left out)",
       "valu_wave_instructions_per_check": total,
       "kernels": {k: {"launches_per_call": len(n[k]) // CALLS, "M_wave_instr_per_call": round(agg[k]["SQ_INSTS_VALU"] / CALLS / 1e6, 2), "us_per_launch_alone": round(dur[k] / len(n[k]), 1)}
                   for k in sorted(per_call, key=lambda k: agg[k]["SQ_INSTS_VALU"], reverse=True)[:10]}}
json.dump(out, open(os.path.join(os.path.dirname(sys.argv[1]), "msm_valu.json"), "w"), indent=1)
print(f"\n{total / 1e6:.2f} M wave-instructions per check -> {os.path.join(os.path.dirname(sys.argv[1]), 'msm_valu.json')}")
PY
