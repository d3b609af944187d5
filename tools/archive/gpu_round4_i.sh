#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_msm.py -q -x ) > $O/pytest_msm.log 2>&1; tail -3 $O/pytest_msm.log
for t in 1 0 1 0; do MINA_TUNE=msm_fp29=$t timeout 300 python tools/c2_rate.py 16 400 2>/dev/null | tail -1; done
MINA_TUNE=msm_fp29=1 timeout 300 python tools/c2_rate.py 1 200 2>/dev/null | tail -1; MINA_TUNE=msm_fp29=0 timeout 300 python tools/c2_rate.py 1 200 2>/dev/null | tail -1
