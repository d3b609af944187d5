#!/bin/bash
# round 4, call A: the -m gpu suite, the default bench line, bench --gpus 2 on the 1-GPU box, the VALU microbench, SQ counters of the dominant kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
( time timeout 1200 python bench.py --gpus 2 --steps 10 --warmup 4 ) > $O/bench_gpus2.json 2> $O/bench_gpus2.err; echo "bench2 rc=$?"; cut -c1-300 $O/bench_gpus2.json; tail -5 $O/bench_gpus2.err
timeout 300 mina_bridge_amd/microbench > $O/microbench.jsonl 2>&1; grep -c probe $O/microbench.jsonl; grep "mix" $O/microbench.jsonl
timeout 900 bash tools/profile_sq.sh r04a > $O/profile_sq.log 2>&1; tail -3 $O/profile_sq.log
