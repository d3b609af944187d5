#!/bin/bash
# round 6: legs of a device-resident job on their own streams (mina_verify_tuning.dev_fork) -- one fresh process per setting "<pipeline>:<jobs>:<MINA_TUNE or ->"
# usage: tools/dev_fork_sweep.sh <tag> [--probes] "4:16384:dev_fork=1" "20:16384:dev_fork=0" ...   -> gpurun_out/<tag>/sweep.jsonl (one line per setting)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
TAG=$1; shift
EXTRA="--no-probes"; if [ "$1" = "--probes" ]; then EXTRA=""; shift; fi
O=gpurun_out/$TAG; mkdir -p $O
for cfg in "$@"; do
  IFS=: read P J T <<< "$cfg"
  [ "$T" = "-" ] && T=""
  MINA_TUNE="$T" timeout 600 python bench.py --no-boundary --no-cpu-baseline $EXTRA --steps ${STEPS:-30} --warmup 3 --pipeline $P --jobs $J > $O/run.json 2>> $O/err.log
  python - "$cfg" $O/run.json <<'PY' | tee -a $O/sweep.jsonl
import json, sys
cfg, path = sys.argv[1:3]
d = None
for l in open(path):
    if l.startswith('{"metric"'): d = json.loads(l)
if d is None: print(json.dumps({"cfg": cfg, "error": "no line"})); raise SystemExit
iso = (d.get("stage_us") or {}).get("isolated") or {}
print(json.dumps({"cfg": cfg, "value": round(d["value"]), "ms_per_step": round(d["ms_per_step"], 2), "sustained": round(d["sustained"]["value"]) if d.get("sustained") else None,
                  "hbm_GiB": d["config"]["hbm_in_use_GiB"], "call_latency_ms": d.get("call_latency_ms") and round(d["call_latency_ms"], 1),
                  "c5": d.get("c5_4096_total_strong") and round(d["c5_4096_total_strong"]["value"]), "sclk": (d.get("power") or {}).get("sclk_mhz_avg"), "w": (d.get("power") or {}).get("socket_power_w_avg"),
                  "pstate_hash_us": iso.get("pstate_hash") and round(iso["pstate_hash"])}))
PY
done
