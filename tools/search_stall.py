# dev tool: what one bad opening per call costs the other callers of a device (bench.py boundary_leg `one_bad_opening_per_call`), under the tuning of $MINA_TUNE
# (e.g. MINA_TUNE=search_ctx=0: the round-4 search on the device's one context, drained and locked).   usage: python tools/search_stall.py [proofs_per_call]
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import mina_bridge_amd as m
import bench
m.lib.tune_from_string(os.environ.get("MINA_TUNE", ""))
r = bench.boundary_leg(m, "0", int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 2.0)
print(json.dumps({"tune": os.environ.get("MINA_TUNE", ""), "lone": round(r["value"]), "two_callers": round(r["two_caller_threads"]["value"]), "four_callers": round(r["four_caller_threads"]["value"]),
                  "c5_before_ms": round(r["c5_4096_per_call"]["ms_per_call"], 1), "c5_after_searches_ms": r["c5_4096_per_call_after_culprit_searches"] and round(r["c5_4096_per_call_after_culprit_searches"]["ms_per_call"], 1),
                  "one_bad_opening_per_call": {k: v for k, v in r["one_bad_opening_per_call"].items() if k != "note"}}))
