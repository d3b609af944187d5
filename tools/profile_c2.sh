#!/bin/bash
# BASELINE config C2 alone (tools/c2_rate.py: 8 un-folded 2^16 Vesta accumulator checks per call) under rocprofv3: per-kernel time with 16 lanes in flight,
# then FETCH_SIZE / WRITE_SIZE in separate --pmc passes on ONE lane (kernels do not overlap) -> tools/c2_traffic.py -> profiles/msm_traffic.json
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_c2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/tools/c2_rate.py 16 400 > $OUT/c2_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o f -- python $GRAFT_REPO_ROOT/tools/c2_rate.py 1 24 > $OUT/c2_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/write -o w -- python $GRAFT_REPO_ROOT/tools/c2_rate.py 1 24 > $OUT/c2_write.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $OUT/c2_trace.log | cut -c1-200
python tools/c2_traffic.py $OUT > $OUT/msm_traffic.json; cat $OUT/msm_traffic.json | head -60
