#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
for i in 1 2 3; do ( timeout 1500 python -m pytest tests/test_sharded_state_job.py tests/test_sharded_gloo.py -q -m gpu ) > $O/pytest_$i.log 2>&1; tail -2 $O/pytest_$i.log; done
