#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "20 32768 20" "20 24576 20" "20 16384 16" "20 16384 24" "80 16384 20"; do set -- $cfg
  timeout 900 python bench.py --steps $1 --warmup 3 --jobs $2 --pipeline $3 --no-probes --no-boundary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
b = json.loads([l for l in sys.stdin if l.startswith('{\"metric')][-1]); print('steps $1 jobs $2 lanes $3:', round(b['value']), round(b['ms_per_step'], 2))"
done
