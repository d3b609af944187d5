#!/bin/bash
# kernel timeline of ONE single-proof mina_verify_state_batch call (the reference's call pattern: one proof per call)
O=$GRAFT_REPO_ROOT/gpurun_out/prof_single; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d $O -o t -- python $GRAFT_REPO_ROOT/tools/boundary_ab.py 1 6 > $O/log.txt 2>&1
tail -2 $O/log.txt
cd $GRAFT_REPO_ROOT; python tools/call_timeline.py $(find $O -name "*kernel_trace.csv" | head -1) 40 0.02 > $O/timeline.txt; tail -3 $O/timeline.txt; wc -l $O/timeline.txt
