#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O
( timeout 1500 python -m pytest tests/test_bench_multi.py tests/test_sharded_state_job.py -q -m gpu ) > $O/pytest.log 2>&1; tail -15 $O/pytest.log
