#!/bin/bash
# dev tool: pipeline / hw-queue sweep
for q in 4 8 16; do for p in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 400 --warmup 20 --pipeline $p --no-cpu-baseline 2>&1 | tail -1 > /tmp/o.json
  python - <<PY
import json
d=json.load(open('/tmp/o.json'))
print("queues",$q,"lanes",d["config"]["pipeline_lanes"], round(d["value"],1), "proofs/s", round(d["ms_per_step"],3),"ms/step acc_us", round(d["roofline"]["avg_launch_us"],1))
PY
done; done
