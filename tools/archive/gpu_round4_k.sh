#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 2 3 4 8; do echo "waves $w"; MINA_X_WAVES=$w MINA_TUNE=msm_fp29=1 timeout 300 python tools/c2_rate.py 16 400 2>/dev/null | tail -1; MINA_X_WAVES=$w MINA_TUNE=msm_fp29=1 timeout 300 python tools/c2_rate.py 1 200 2>/dev/null | tail -1; done
timeout 600 python -m pytest tests/test_gpu_msm.py -q -x 2>&1 | tail -2
