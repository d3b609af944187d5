#!/bin/bash
# round 4, call C: the whole -m gpu suite on the current build + the batch-affine probe (K1 question) + the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 600 tools/probes/bin/batch_affine_probe > $O/batch_affine_probe.jsonl 2>&1; cat $O/batch_affine_probe.jsonl
( time timeout 3000 python -m pytest tests -m gpu -q --durations=5 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/r04c/bench.json') if l.startswith('{"metric')][-1])
print({k:b[k] for k in ('value','ms_per_step')}, b['roofline_valu']['frac'], b['roofline_valu'].get('waves_per_simd_in_launch'), b['cpu_baseline']['value'], b.get('cpu_baseline_folded',{}).get('value'))
PY
