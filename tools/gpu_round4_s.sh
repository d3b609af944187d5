#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python __graft_entry__.py --smoke 2>&1 | tail -2
for cfg in "20 16384 20" "20 8192 20" "10 16384 10"; do set -- $cfg
  timeout 900 python bench.py --steps $1 --warmup 3 --jobs $2 --pipeline $3 --no-probes --no-boundary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
b = json.loads([l for l in sys.stdin if l.startswith('{\"metric')][-1]); print('steps $1 jobs $2 lanes $3:', round(b['value']), round(b['ms_per_step'], 2))"
done
