#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the default bench command, then FETCH_SIZE / WRITE_SIZE in
# separate --pmc passes (MI355X_MICROARCH.md: they do not fit one pass), single lane and ONE stream per job (MINA_TUNE=dev_fork=0, round 6) so kernels do not overlap.
TAG=${1:-r01b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --no-probes --no-cpu-baseline --no-boundary > $OUT/bench_trace.log 2>&1
MINA_TUNE=dev_fork=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 4 --pipeline 1 --no-probes --no-cpu-baseline --no-boundary > $OUT/bench_fetch.log 2>&1
MINA_TUNE=dev_fork=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 4 --pipeline 1 --no-probes --no-cpu-baseline --no-boundary > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | head -20
grep -h '"metric"' $OUT/bench_trace.log | cut -c1-200
