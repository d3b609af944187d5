#!/bin/bash
# round 6: ONE script for the GPU-box calls; stages named on the command line, outputs under gpurun_out/<tag>/.
#   tools/gpu_round6.sh <tag> stage [stage ...]     stages: pytest bench bench20 profile c2 c2single timeline tcc gpus2 gpus8 preflight soak callers
cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for st in "$@"; do
  case $st in
    pytest)   ( time timeout 3000 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log ;;
    bench)    ( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json | tail -1 ;;
    bench20)  ( time timeout 900 python bench.py --steps 20 --warmup 3 ) > $O/bench_steps20.json 2>> $O/bench.err; echo "bench20 rc=$?" ;;
    profile)  timeout 900 bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log | cut -c1-200
              timeout 900 bash tools/profile_sq.sh $TAG > $O/profile_sq.log 2>&1
              python tools/profile_report.py $TAG $O/bench.json > $O/${TAG}_rocprof.md 2> $O/report.err; wc -l $O/${TAG}_rocprof.md ;;
    c2)       timeout 900 bash tools/profile_c2.sh $TAG > $O/profile_c2.log 2>&1; tail -3 $O/profile_c2.log
              timeout 900 bash tools/profile_c2_sq.sh $TAG > $O/profile_c2_sq.log 2>&1; tail -4 $O/profile_c2_sq.log
              timeout 300 python tools/c2_rate.py 16 400 2>/dev/null | tail -1 ;;
    c2single) # BASELINE C2 as written: ONE 2^16 check per call, alone: wall time, then kernel stats and the FETCH / WRITE passes of the same command
              timeout 300 python tools/c2_single.py 200 2>/dev/null | tail -1 | tee $O/c2_single_wall.json
              ( cd /tmp && export TMPDIR=/tmp
                rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/c2single_trace -o t -- python $GRAFT_REPO_ROOT/tools/c2_single.py 64 > $GRAFT_REPO_ROOT/$O/c2single_trace.log 2>&1
                rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $GRAFT_REPO_ROOT/$O/c2single_fetch -o f -- python $GRAFT_REPO_ROOT/tools/c2_single.py 64 > $GRAFT_REPO_ROOT/$O/c2single_fetch.log 2>&1
                rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $GRAFT_REPO_ROOT/$O/c2single_write -o w -- python $GRAFT_REPO_ROOT/tools/c2_single.py 64 > $GRAFT_REPO_ROOT/$O/c2single_write.log 2>&1 )
              python tools/c2_single_report.py $O > $O/c2_single.json 2> $O/c2_single_report.err; cat $O/c2_single.json | cut -c1-600 ;;
    tcc)      # K1: L2 (TCC) hit / miss and TCP counters of the accumulate kernels in the 8-checks-per-call form (profiles/r06_k1.md)
              ( cd /tmp && export TMPDIR=/tmp
                for pm in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
                  tag=$(echo $pm | cut -d' ' -f1)
                  rocprofv3 --kernel-trace --pmc $pm -f csv -d $GRAFT_REPO_ROOT/$O/tcc_$tag -o c -- python $GRAFT_REPO_ROOT/tools/c2_rate.py 1 24 > $GRAFT_REPO_ROOT/$O/tcc_$tag.log 2>&1
                done )
              python tools/pmc_by_kernel.py $O/tcc_* > $O/tcc_by_kernel.txt 2>&1; head -40 $O/tcc_by_kernel.txt ;;
    timeline) # a lone forked call of 16384 proofs, kernel by kernel (profiles/r06_dev_fork.md)
              ( cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$O/tl -o t -- python $GRAFT_REPO_ROOT/tools/dev_fork_rate.py 16384 "1:dev_fork=1" > $GRAFT_REPO_ROOT/$O/tl.log 2>&1 )
              python tools/call_timeline.py $(find $O/tl -name "*kernel_trace.csv" | head -1) 100 0.2 > $O/timeline_lone_fork.txt; rm -rf $O/tl; tail -3 $O/timeline_lone_fork.txt ;;
    gpus2)    ( time timeout 1200 python bench.py --gpus 2 ) > $O/bench_gpus2_shared.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?" ;;
    gpus8)    ( time timeout 2400 python bench.py --gpus 8 --steps 10 ) > $O/bench_gpus8_shared.json 2> $O/bench_gpus8.err; echo "gpus8 rc=$?" ;;
    preflight) ( timeout 1200 python bench.py --gpus 8 --preflight ) > $O/preflight_gpus8.json 2> $O/preflight.err; echo "preflight rc=$?"; cut -c1-400 $O/preflight_gpus8.json ;;
    callers)  timeout 600 python tools/concurrent_callers.py 6 > $O/concurrent_callers.log 2>&1; tail -12 $O/concurrent_callers.log ;;
    soak)     for t in "soak.py 120" "soak_ipa.py 90" "soak_sponge.py 120" "soak_lanes.py 120" "soak_fork.py 120" "soak_verifier.py 180" "soak_boundary.py 300"; do set -- $t
                secs=$(( $2 * ${SOAK_SCALE:-1} )); timeout $(( secs + 600 )) python tools/$1 $secs > $O/${1%.py}.log 2>&1; echo "$1 rc=$?" | tee -a $O/soak_rc.log; tail -1 $O/${1%.py}.log | cut -c1-400
              done ;;
    *) echo "unknown stage $st" ;;
  esac
done
