"""summary of tools/profile_c2.sh: HBM-side traffic of ONE 2^16 accumulator check (= one 2^16-base fixed-base MSM + its b_poly coefficients), summed over
every kernel of the check, KiB units.  FETCH_SIZE is corrected per access pattern: x 2 for the kernels that stream coalesced 16-B-per-lane loads
(MI355X_MICROARCH.md: on gfx950 the counter tallies their 128-B requests at 64 B; the same correction profiles/pstate_hash_traffic.json applies), x 1 for
the accumulate kernels, whose traffic is random 64-B gathers from the window table: CALIBRATED on a known byte count (tools/calibrate_fetch.sh: the probe's
gather kernel moves 570.5 MB per launch by construction, FETCH_SIZE reports 577.6 MB).  Calls of c2_rate.py carry 8 checks each: launches after the warm-up are
counted per kernel name and divided by the number of checks they served.  usage: python tools/c2_traffic.py <dir with fetch/ write/ trace/>"""
import collections, csv, glob, json, os, sys
d = sys.argv[1]


def per_kernel(sub, counter):
    f = glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
# c2_rate.py 1 24: 32 warm-up calls + 24 timed calls + the set-up MSMs of make_accumulators (8 single MSMs, other kernel instances): use the LAST 24 calls' launches
CALLS, PER_CALL = 24, 8
rows, tot_f, tot_w = [], 0.0, 0.0
for k in sorted(set(fetch) | set(write)):
    if not any(t in k for t in ("msm_", "bpoly", "challenge_to_field", "accumulator", "points_", "field_")):
        continue
    fv, wv = fetch.get(k, []), write.get(k, [])
    n = min(len(fv), len(wv))
    per_call = max(1, round(n / (CALLS + 32)))                   # launches of this kernel per call (the warm-up calls are the same shape)
    last = CALLS * per_call
    if n < last:
        continue
    f_kib, w_kib = sum(fv[-last:]) / (CALLS * PER_CALL), sum(wv[-last:]) / (CALLS * PER_CALL)
    factor = 1.0 if ("accumulate" in k or "heavy" in k) else 2.0      # random 64-B gathers (calibrated) / coalesced 16-B-per-lane streams (the guide's correction)
    rows.append({"kernel": k.replace("void ", "").replace("mb::", ""), "launches_per_call": per_call, "fetch_kib_per_check": round(f_kib, 1), "write_kib_per_check": round(w_kib, 1),
                 "fetch_factor": factor, "hbm_bytes_per_check": round((factor * f_kib + w_kib) * 1024)})
    tot_f += f_kib; tot_w += w_kib; tot_b = (tot_b if "tot_b" in dir() else 0.0) + (factor * f_kib + w_kib) * 1024
alg = 65536 * 96 + 96
out = {"source": f"tools/profile_c2.sh -> {os.path.basename(d)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, one lane, 8 checks per call, last {CALLS} calls)",
       "fetch_size_kib_per_check": round(tot_f, 1), "write_size_kib_per_check": round(tot_w, 1), "hbm_bytes_per_check": round(tot_b),
       "hbm_bytes_per_check_uncorrected": round((tot_f + tot_w) * 1024), "hbm_bytes_per_check_all_fetch_doubled": round((2 * tot_f + tot_w) * 1024),
       "algorithmic_bytes_per_msm": alg, "ratio_to_algorithmic": round(tot_b / alg, 2),
       "note": "hbm_bytes = sum over kernels of (fetch_factor x FETCH_SIZE + WRITE_SIZE): factor 2 for coalesced 16-B-per-lane streams (the guide's gfx950 correction, as for "
               "pstate_hash), factor 1 for the accumulate kernels' random 64-B gathers (calibrated: tools/calibrate_fetch.sh); every kernel of a check, not the accumulate kernel alone",
       "kernels": sorted(rows, key=lambda r: -r["hbm_bytes_per_check"])}
print(json.dumps(out, indent=1))
