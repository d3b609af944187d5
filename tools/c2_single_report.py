"""summary of `tools/gpu_round6.sh <tag> c2single`: BASELINE C2 as written -- ONE un-folded 2^16 Vesta accumulator check per call, alone on the chip (tools/c2_single.py 64).
Per check, from the LAST 64 calls' launches (a call ends with xyzz_eq_affine_kernel): sum of kernel durations (rocprofv3 --kernel-trace), HBM-side bytes (FETCH_SIZE / WRITE_SIZE in
separate --pmc passes; FETCH x 2 for the coalesced 16-B-per-lane streams -- MI355X_MICROARCH.md's gfx950 correction -- x 1 for the accumulate kernels' random 64-B gathers, calibrated:
tools/calibrate_fetch.sh).   usage: python tools/c2_single_report.py gpurun_out/<tag>   -> JSON (the `single_check` entry of profiles/msm_traffic.json)"""
import collections, csv, glob, json, os, sys
d = sys.argv[1]
CALLS = 64
END = "xyzz_eq_affine_kernel"


def find(sub, pat):
    f = glob.glob(os.path.join(d, sub, "**", pat), recursive=True)
    return f[0] if f else None


def last_calls(rows, key_start):
    rows = sorted(rows, key=key_start)
    ends = [i for i, r in enumerate(rows) if END in r["Kernel_Name"]]
    assert len(ends) > CALLS, f"only {len(ends)} checks in the trace"
    return rows[ends[-CALLS - 1] + 1: ends[-1] + 1]


short = lambda n: n.replace("void ", "").replace("mb::", "").split("(")[0]
tr = last_calls(list(csv.DictReader(open(find("c2single_trace", "*kernel_trace.csv")))), lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(float); cnt = collections.Counter()
for r in tr:
    k = short(r["Kernel_Name"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; cnt[k] += 1
span = (max(int(r["End_Timestamp"]) for r in tr) - min(int(r["Start_Timestamp"]) for r in tr)) / 1e3 / CALLS
pm = {}
for sub, ctr in (("c2single_fetch", "FETCH_SIZE"), ("c2single_write", "WRITE_SIZE")):
    rows = [r for r in csv.DictReader(open(find(sub, "*counter_collection.csv"))) if r["Counter_Name"] == ctr]
    rows = last_calls(rows, lambda r: int(r["Dispatch_Id"]))
    acc = collections.defaultdict(float)
    for r in rows: acc[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    pm[ctr] = acc
kern = []
tot = 0.0
for k in sorted(dur, key=dur.get, reverse=True):
    f_kib, w_kib = pm["FETCH_SIZE"].get(k, 0.0) / CALLS, pm["WRITE_SIZE"].get(k, 0.0) / CALLS
    factor = 1.0 if ("accumulate" in k or "heavy" in k) else 2.0
    b = (factor * f_kib + w_kib) * 1024; tot += b
    kern.append({"kernel": k, "launches_per_check": cnt[k] / CALLS, "us_per_check": round(dur[k] / CALLS, 2), "fetch_kib": round(f_kib, 1), "write_kib": round(w_kib, 1), "fetch_factor": factor,
                 "hbm_bytes": round(b)})
alg = 65536 * 96 + 96
ksum = sum(dur.values()) / CALLS
print(json.dumps({"source": f"tools/gpu_round6.sh c2single -> {os.path.basename(d.rstrip('/'))}: rocprofv3 --kernel-trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/c2_single.py 64, last {CALLS} checks",
                  "kernel_us_sum_per_check": round(ksum, 1), "first_launch_to_last_end_us_per_check": round(span, 1), "hbm_bytes_per_check": round(tot), "algorithmic_bytes_per_msm": alg,
                  "ratio_to_algorithmic": round(tot / alg, 2), "algorithmic_GBps_kernels": round(alg / ksum / 1e3, 2), "traffic_GBps_kernels": round(tot / ksum / 1e3, 1),
                  "frac_of_hbm_peak_algorithmic": round(alg / ksum / 1e3 / 8000, 5), "frac_of_hbm_peak_traffic": round(tot / ksum / 1e3 / 8000, 4), "kernels": kern}, indent=1))
