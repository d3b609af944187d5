#!/bin/bash
# round 4: the evidence run of the final build -- gpu suite, bench lines, rocprofv3 passes, microbench
cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 3000 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
( time timeout 900 python bench.py --steps 20 --warmup 3 ) > $O/bench_driver_shape.json 2>> $O/bench.err; echo "bench(driver shape) rc=$?"
( time timeout 1200 python bench.py --gpus 2 ) > $O/bench_gpus2_shared.json 2> $O/bench_gpus2.err; echo "bench --gpus 2 rc=$?"
timeout 300 mina_bridge_amd/microbench > $O/microbench.jsonl 2>&1
timeout 900 bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log | cut -c1-200
timeout 900 bash tools/profile_sq.sh $TAG > $O/profile_sq.log 2>&1
timeout 900 bash tools/profile_c2.sh $TAG > $O/profile_c2.log 2>&1
python tools/profile_report.py $TAG $O/bench.json > $O/${TAG}_rocprof.md 2> $O/report.err; wc -l $O/${TAG}_rocprof.md
for t in 1 0; do MINA_TUNE=msm_fp29=$t timeout 300 python tools/c2_rate.py 16 400 2>/dev/null | tail -1; done
timeout 600 python tools/concurrent_callers.py 6 > $O/concurrent_callers.log 2>&1; tail -12 $O/concurrent_callers.log
