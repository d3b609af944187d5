#!/usr/bin/env python3
"""profiles/<tag>_rocprof.md from one round's raw rocprofv3 output (tools/profile_round.sh + tools/profile_sq.sh) and the bench line.
usage: profile_report.py <tag> <bench.json>     reads gpurun_out/prof_<tag>/, prints markdown"""
import collections
import csv
import json
import os
import sys

from profile_summary import pmc, short, stats_table

tag, bench = sys.argv[1], sys.argv[2]
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "prof_" + tag)
b = json.loads(open(bench).read().strip().split("\n")[-1])
STEPS = 14                                                   # --steps 10 --warmup 4 of the single-lane passes
rnd = tag[1:3].lstrip("0")
print(f"# Round {rnd} profile ({tag}) -- `python bench.py` (default mode `{b['config'].get('mode', 'prepared')}`: {b['config']['proofs_per_step']} proofs per step, "
      f"{b['config']['pipeline_lanes']} lanes) on one MI355X\n")
print(f"Raw rocprofv3 output: `gpurun_out/prof_{tag}/` (scratch).  Commands: `tools/profile_round.sh {tag}` (kernel-trace + stats of the default bench command; "
      f"FETCH_SIZE and WRITE_SIZE in separate `--pmc` passes with `--pipeline 1` so that kernels do not overlap) and `tools/profile_sq.sh {tag}` (SQ counters, own "
      f"passes).  Bench line of the same build: `profiles/{tag}_bench.json` ({b['value'] / 1e3:.1f} k proofs/s, {b['ms_per_step']:.1f} ms per step).\n")
print(f"## kernel-trace stats of the default command ({b['config']['pipeline_lanes']} lanes in flight: durations include time-sharing of the CUs)\n")
print(stats_table(os.path.join(d, "trace", "t_kernel_stats.csv")))
F, W = pmc(os.path.join(d, "fetch", "f_counter_collection.csv")), pmc(os.path.join(d, "write", "w_counter_collection.csv"))
print("\n## PMC passes, KiB per launch (most frequent grid size of each kernel)\n")
print("| kernel | grid | launches | FETCH_SIZE | WRITE_SIZE | bytes/launch |\n|---|---|---|---|---|---|")
tot = {k: (F[k][0] + W.get(k, (0,))[0]) * 1024 for k in F}
for k in sorted(tot, key=tot.get, reverse=True)[:14]:
    print(f"| {k} | {F[k][2]} | {F[k][1]} | {F[k][0]:.0f} | {W.get(k, (0,))[0]:.0f} | {tot[k] / 1e6:.1f} MB |")
# single-lane step budget from the FETCH pass's kernel trace
rows = list(csv.DictReader(open(os.path.join(d, "fetch", "f_kernel_trace.csv"))))
# round 6 (VERDICT r05 next #3c): a (kernel, grid) pair launched a whole number of times per step belongs to the steps; everything else -- the set-up's launches of the
# same kernels on other grids (the 16 384 chains are hashed state by state before the first step: 17 launches of pstate_hash_kernel) -- gets a row of its own, so that
# the per-launch average of a step's kernels can be read off this table
gsz = lambda r: r.get("Grid_Size") or (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])      # the kernel-trace CSV names the three dimensions, the counter CSV their product
gcount = collections.Counter((short(r["Kernel_Name"]), gsz(r)) for r in rows)
dur, cnt = collections.defaultdict(float), collections.Counter()
for r in rows:
    n = short(r["Kernel_Name"]); g_ = gcount[(n, gsz(r))]
    if g_ < STEPS or g_ % STEPS: n += " [set-up]"
    dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; cnt[n] += 1
step_sum = sum(v for k, v in dur.items() if not k.endswith("[set-up]"))
print(f"\n## Single-lane step (`--pipeline 1`, `dev_fork = 0`: ONE stream, kernels back to back): time per step of {b['config']['proofs_per_step']} proofs "
      f"(sum over the steps' kernels {step_sum / STEPS / 1e3:.1f} ms; set-up launches on rows of their own, per run)\n")
print("| kernel | launches/step | us/step | us/launch |\n|---|---|---|---|")
for k in sorted(dur, key=dur.get, reverse=True)[:30]:
    if k.endswith("[set-up]"): print(f"| {k} | ({cnt[k]} per run) | ({dur[k]:.0f} per run) | {dur[k] / cnt[k]:.0f} |")
    else: print(f"| {k} | {cnt[k] / STEPS:.1f} | {dur[k] / STEPS:.0f} | {dur[k] / cnt[k]:.0f} |")
# VALU instruction budget from the SQ pass
sq = os.path.join(d, "sq", "s_counter_collection.csv")
if os.path.exists(sq):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    rows_sq = list(csv.DictReader(open(sq)))
    # the set-up launches some of the step's kernels too, on other grid sizes (round 5: the 16 384 distinct chains are hashed
    # state by state, 17 launches of pstate_hash_kernel on 16 384 states each -- one step's worth of hashes that is not a step) and is left out
    grids = collections.defaultdict(collections.Counter)
    for r in rows_sq:
        if r["Counter_Name"] == "SQ_WAVES": grids[short(r["Kernel_Name"])][r["Grid_Size"]] += 1
    for r in rows_sq:                                        # a (kernel, grid) pair belongs to the steps when it was launched a whole number of times per step
        n_ = grids[short(r["Kernel_Name"])][r["Grid_Size"]]
        if n_ < STEPS or n_ % STEPS: continue
        agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"]); disp[short(r["Kernel_Name"])].add(r["Dispatch_Id"])
    # one-time kernels (SRS tables, the Lagrange basis' FFT stages: fewer launches than steps, or launches that come in one burst of a setup) are not part of a step
    setup = {k for k in agg if len(disp[k]) < STEPS or k.startswith("mb::lagrange_stage_kernel") or k.startswith("mb::msm_build_table_kernel") or k.startswith("mb::lagrange_digit_table_kernel")}
    setup_total = sum(agg[k]["SQ_INSTS_VALU"] for k in setup) / 1e9
    for k in setup: del agg[k]
    total = sum(v["SQ_INSTS_VALU"] for v in agg.values()) / STEPS
    MIX = 4.40                                               # cycles per instruction of the Poseidon rounds' mix (774 multiply-accumulates : ~190 simple), 8 waves per SIMD:
    floor = total * MIX / (1024 * 2.4e9) * 1e3               # profiles/r04_microbench_ratio.jsonl -- 86 % of a step's instructions are sponge kernels
    print(f"\n## VALU instruction budget per step (SQ_INSTS_VALU, single lane; one-time setup kernels -- {setup_total:.2f} G in all -- left out)\n\nTotal {total / 1e9:.2f} G wave-instructions per step; at the MEASURED {MIX} cycles per wave64 "
          f"instruction of the sponge kernels' mix (profiles/r04_valu_roofline.md) on 1024 SIMDs and 2.4 GHz that is **{floor:.1f} ms** per step -- the pipelined step takes "
          f"{b['ms_per_step']:.1f} ms (= {floor / b['ms_per_step'] * 100:.0f} % of that issue rate; the 'floor ms' column below prices every instruction the same way).\n")
    json.dump({"source": f"tools/profile_sq.sh {tag} + tools/profile_report.py: rocprofv3 --pmc SQ_INSTS_VALU over `bench.py --pipeline 1` ({STEPS} steps), one-time setup kernels left out",
               "proofs_per_step": b["config"]["proofs_per_step"], "valu_wave_instructions_per_step": total, "cycles_per_wave_instruction": MIX,
               "top_kernels_G_per_step": {k: round(agg[k]["SQ_INSTS_VALU"] / STEPS / 1e9, 3) for k in sorted(agg, key=lambda k: agg[k]["SQ_INSTS_VALU"], reverse=True)[:8]}},
              open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "step_valu.json"), "w"), indent=1)
    print("| kernel | G instr/step | floor ms | waves/step |\n|---|---|---|---|")
    for k in sorted(agg, key=lambda k: agg[k]["SQ_INSTS_VALU"], reverse=True)[:18]:
        v = agg[k]; iv = v["SQ_INSTS_VALU"] / STEPS
        print(f"| {k} | {iv / 1e9:.3f} | {iv * MIX / (1024 * 2.4e9) * 1e3:.2f} | {v['SQ_WAVES'] / STEPS:.0f} |")
r, rv = b["roofline"], b.get("roofline_valu", {})
print(f"\n## Dominant kernel `{r['kernel']}` (one launch = {r['states_per_launch']} protocol-state hashes)\n")
print(f"* HIP events in bench.py: {r['avg_launch_us']:.0f} us isolated, {r['avg_launch_us_in_timed_region']:.0f} us inside the timed region (lanes time-share the CUs; "
      f"rocprofv3's average over the same region is in the first table).")
h = r.get("hbm", r)                                          # round 5: `roofline` is the VALU bound, the HBM view rides inside it
print(f"* algorithmic bytes per launch {h['algorithmic_bytes_per_launch']} B -> {h['achieved']:.2f} GB/s = {h['frac'] * 100:.3f} % of the 8 TB/s HBM peak; PMC traffic "
      f"(FETCH x 2 for 16-B-per-lane loads + WRITE): see the table above; the kernel is integer-multiply bound.")
if rv:
    print(f"* multiply-accumulate issue: {rv['permutations_per_launch']} permutations x {rv.get('limb_macs_per_permutation', 0)} limb MACs per launch = {rv['achieved']:.1f} {rv['unit']} of a {rv['peak']:.1f} "
          f"{rv['unit']} issue rate ({rv['bound']}) = **{rv['frac']:.2f}**" + (f"; against the measured pure v_mad_u64_u32 peak of {rv['pure_mac_peak']:.1f}: **{rv['frac_of_pure_mac_peak']:.2f}** (`roofline.frac`)." if "pure_mac_peak" in rv else "."))
