#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "20 20 24" "20 24 24" "20 32 24" "80 24 24" "80 32 24" "20 20 32" "80 20 32" "20 32 40" "80 32 40"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$3 timeout 600 python bench.py --steps $1 --warmup 3 --pipeline $2 --no-probes --no-boundary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
b = json.loads([l for l in sys.stdin if l.startswith('{\"metric')][-1]); print('steps $1 lanes $2 queues $3:', round(b['value']), round(b['ms_per_step'], 2))"
done
