#!/bin/bash
# round 5: ONE script for every GPU-box call; stages named on the command line, outputs under gpurun_out/<tag>/.
#   tools/gpu_round5.sh <tag> stage [stage ...]     stages: fe52 sharded pytest msm c2ab abr04 multi bench bench20 gpus2 microbench profile c2 account callers soak
cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for st in "$@"; do
  case $st in
    fe52)     ( timeout 120 tools/probes/bin/fe52_probe --values | python tools/probes/fe52_check.py ) > $O/fe52_check.json 2>&1; cat $O/fe52_check.json
              timeout 300 tools/probes/bin/fe52_probe > $O/fe52_probe.jsonl 2>&1; tail -12 $O/fe52_probe.jsonl ;;
    sharded)  ( time timeout 1500 python -m pytest tests/test_sharded_state_job.py -m gpu -q -x ) > $O/pytest_sharded.log 2>&1; tail -5 $O/pytest_sharded.log ;;
    pytest)   ( time timeout 3000 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log ;;
    bench)    ( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-1500 $O/bench.json ;;
    bench20)  ( time timeout 900 python bench.py --steps 20 --warmup 3 ) > $O/bench_steps20.json 2>> $O/bench.err; echo "bench20 rc=$?" ;;
    gpus2)    ( time timeout 1200 python bench.py --gpus 2 ) > $O/bench_gpus2_shared.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?" ;;
    microbench) timeout 300 mina_bridge_amd/microbench > $O/microbench.jsonl 2>&1; timeout 300 mina_bridge_amd/microbench --ratio > $O/microbench_ratio.jsonl 2>&1 ;;
    profile)  timeout 900 bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log | cut -c1-200
              timeout 900 bash tools/profile_sq.sh $TAG > $O/profile_sq.log 2>&1
              python tools/profile_report.py $TAG $O/bench.json > $O/${TAG}_rocprof.md 2> $O/report.err; wc -l $O/${TAG}_rocprof.md ;;
    gpus8)    ( time timeout 2400 python bench.py --gpus 8 --steps 10 ) > $O/bench_gpus8_shared.json 2> $O/bench_gpus8.err; echo "gpus8 rc=$?" ;;
    preflight) ( timeout 1200 python bench.py --gpus 8 --preflight ) > $O/preflight_gpus8.json 2> $O/preflight.err; echo "preflight rc=$?"; cut -c1-400 $O/preflight_gpus8.json ;;
    c2)       timeout 900 bash tools/profile_c2.sh $TAG > $O/profile_c2.log 2>&1; tail -3 $O/profile_c2.log
              timeout 900 bash tools/profile_c2_sq.sh $TAG > $O/profile_c2_sq.log 2>&1; tail -4 $O/profile_c2_sq.log
              timeout 300 python tools/c2_rate.py 16 400 2>/dev/null | tail -1 ;;
    msm)      ( time timeout 2400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_large_srs.py tests/test_gpu_ipa.py tests/test_gpu_sponge_ipa.py -m gpu -q -x ) > $O/pytest_msm.log 2>&1; tail -5 $O/pytest_msm.log ;;
    c2ab)     for rep in 1 2 3; do for t in ${C2AB:-3 2 1 0}; do echo -n "msm_fp29=$t "; MINA_TUNE=msm_fp29=$t timeout 300 python tools/c2_rate.py 16 400 2>/dev/null | tail -1; done; done | tee $O/c2_ab.log ;;
    multi)    ( time timeout 3000 python -m pytest tests/test_bench_multi.py tests/test_sharded_state_job.py -m gpu -q ) > $O/pytest_multi.log 2>&1; tail -6 $O/pytest_multi.log ;;
    abr04)    bash tools/ab_c2.sh tools/probes/bin/lib_r04.so 3 2>&1 | tee $O/ab_c2_r04.log ;;   # A = this build, B = the round-4 library (tools/probes/bin/lib_r04.so, built from 4076b66), same box
    account)  timeout 600 python tools/c4_rate.py > $O/c4_rate.log 2>&1; tail -8 $O/c4_rate.log ;;
    callers)  timeout 600 python tools/concurrent_callers.py 6 > $O/concurrent_callers.log 2>&1; tail -12 $O/concurrent_callers.log ;;
    law29)    # kernel time of the direct commitments and the variable-base MSM with the 29-bit law (msm_fp29=1) and without (=0): rocprofv3 kernel stats of tools/probes/law29_ab.py
              ( cd /tmp && export TMPDIR=/tmp && for t in 1 0; do MINA_TUNE=msm_fp29=$t rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/law29_$t -o t -- python $GRAFT_REPO_ROOT/tools/probes/law29_ab.py > $GRAFT_REPO_ROOT/$O/law29_$t.log 2>&1; done )
              for t in 1 0; do echo "msm_fp29=$t $(tail -1 $O/law29_$t.log)"; f=$(find $O/law29_$t -name "*kernel_stats.csv" | head -1); grep -i "pubcomm\|msm_accumulate\|msm_table29" $f | cut -d, -f1-5 | cut -c1-160; done | tee $O/law29_ab.txt ;;
    lawtests) ( time timeout 2400 python -m pytest tests/test_lagrange.py tests/test_gpu_msm.py tests/test_kimchi.py tests/test_state_job.py tests/test_native_composite.py tests/test_verify_boundary.py -m gpu -q -x ) > $O/pytest_law.log 2>&1; tail -5 $O/pytest_law.log ;;
    soak)     # randomised differential soaks of the build against the oracle, fresh seeds; SOAK_SCALE multiplies the seconds (default 1: 19 min in all)
              for t in "soak.py 180" "soak_ipa.py 120" "soak_sponge.py 180" "soak_lanes.py 180" "soak_verifier.py 180" "soak_boundary.py 300"; do set -- $t
                secs=$(( $2 * ${SOAK_SCALE:-1} )); timeout $(( secs + 600 )) python tools/$1 $secs > $O/${1%.py}.log 2>&1; echo "$1 rc=$?" | tee -a $O/soak_rc.log; tail -1 $O/${1%.py}.log | cut -c1-400
              done ;;
    *) echo "unknown stage $st" ;;
  esac
done
