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
per = collections.defaultdict(lambda: collections.defaultdict(dict))      # kernel -> dispatch -> counter -> value (+ "us")
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d = per[k][r["Dispatch_Id"]]
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
CALLS = 32 + 24                                              # tools/c2_rate.py 16 24: 32 warm-up calls + 24 timed, 8 checks each
med = lambda xs: sorted(xs)[len(xs) // 2]
rows = {}
for k, ds in per.items():
    if len(ds) < CALLS: continue                             # one-time kernels (SRS generation, window tables) are not part of a check
    # the set-up also launches some of the check's kernels a few times on single MSMs (minting the accumulators): the MEDIAN dispatch is a call's
    rows[k] = {"per_call": round(len(ds) / CALLS), "instr": med([d.get("SQ_INSTS_VALU", 0.0) for d in ds.values()]), "waves": med([d.get("SQ_WAVES", 0.0) for d in ds.values()]), "us": med([d["us"] for d in ds.values()])}
print("| kernel | launches per call | us / launch (alone) | waves / launch | M VALU instr / launch | cycles per wave-instr per SIMD (2.4 GHz) |\n|---|---|---|---|---|---|")
for k in sorted(rows, key=lambda k: rows[k]["instr"] * rows[k]["per_call"], reverse=True)[:14]:
    r = rows[k]
    print(f"| {k} | {r['per_call']} | {r['us']:.1f} | {r['waves']:.0f} | {r['instr'] / 1e6:.2f} | {r['us'] * 1e-6 * 2.4e9 * 1024 / r['instr'] if r['instr'] else 0:.2f} |")
total = sum(r["instr"] * r["per_call"] for r in rows.values()) / 8
out = {"source": f"tools/profile_c2_sq.sh {sys.argv[2]}: rocprofv3 --pmc SQ_INSTS_VALU, tools/c2_rate.py 16 24 (the 16-lane kernel forms of the timed run; 8 un-folded 2^16 Vesta accumulator checks per call, {CALLS} calls; median dispatch of every kernel launched at least once per call)",
       "valu_wave_instructions_per_check": total,
       "kernels": {k: {"launches_per_call": rows[k]["per_call"], "M_wave_instr_per_call": round(rows[k]["instr"] * rows[k]["per_call"] / 1e6, 2), "us_per_launch_alone": round(rows[k]["us"], 1)}
                   for k in sorted(rows, key=lambda k: rows[k]["instr"] * rows[k]["per_call"], reverse=True)[:12]}}
json.dump(out, open(os.path.join(os.path.dirname(sys.argv[1]), "msm_valu.json"), "w"), indent=1)
print(f"\n{total / 1e6:.2f} M wave-instructions per check -> {os.path.join(os.path.dirname(sys.argv[1]), 'msm_valu.json')}")
PY
