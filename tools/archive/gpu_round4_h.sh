#!/bin/bash
# round 4, call H: MSM tests of the fp29 group law, the whole gpu suite, C2 rate A/B (msm_fp29 1 / 0), the bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_msm.py -q -x ) > $O/pytest_msm.log 2>&1; tail -5 $O/pytest_msm.log
for t in 1 0; do MINA_TUNE=msm_fp29=$t timeout 300 python tools/c2_rate.py 16 400 2>/dev/null | tail -1; done
MINA_TUNE=msm_fp29=1 timeout 300 python tools/c2_rate.py 1 200 2>/dev/null | tail -1; MINA_TUNE=msm_fp29=0 timeout 300 python tools/c2_rate.py 1 200 2>/dev/null | tail -1
( time timeout 3000 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/r04h/bench.json') if l.startswith('{"metric')][-1])
print({k:b[k] for k in ('value','ms_per_step')}, 'c2', b['c2_accumulator_only']['value'], 'c5', b['c5_4096_total_strong']['value'], 'boundary', b['boundary_bytes_to_bools']['value'], b['stage_us']['isolated'])
PY
