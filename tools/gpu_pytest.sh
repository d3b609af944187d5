#!/bin/bash
# the whole -m gpu suite on the box (no -x: every failure is listed)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pytest}; mkdir -p $O
( time timeout 3000 python -m pytest tests -m gpu -q --durations=15 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
