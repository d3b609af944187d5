# dev tool: kernel timeline of the LAST boundary call in a rocprofv3 --kernel-trace CSV (run: tools/boundary_ab.py SIZE 2 under rocprofv3)
# usage: python tools/call_timeline.py KERNEL_TRACE.csv [WINDOW_MS] [MIN_MS]      (MIN_MS: shortest kernel listed, default 0.25)
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 70.0
min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
end = max(r["e"] for r in rows)
last = [r for r in rows if r["s"] >= end - win * 1e6]
# the last call: everything after the previous call's verdict words left the GPU (words_out_kernel, or the verdict kernel of older builds)
last.sort(key=lambda r: r["s"])
ends = [r["e"] for r in last if "words_out_kernel" in r["Kernel_Name"]] or [r["e"] for r in last if "state_job_verdict_kernel" in r["Kernel_Name"]]
if len(ends) >= 2:
    last = [r for r in last if r["s"] >= ends[-2] and r["s"] <= ends[-1]]
t0 = last[0]["s"]
def short(n): return n.replace("mb::", "").split("(")[0][:44]
print(f"{'start':>8} {'dur':>8}  queue  kernel")
for r in last:
    d = (r["e"] - r["s"]) / 1e6
    if d >= min_ms: print(f"{(r['s'] - t0) / 1e6:8.3f} {d:8.3f}  {r.get('Queue_Id', '?'):>5}  {short(r['Kernel_Name'])}")
print(f"total {(max(r['e'] for r in last) - t0) / 1e6:.2f} ms, {len(last)} kernels")
