#!/bin/bash
# what clock and socket power the chip runs at UNDER the bench's load (the "2.4 GHz" every cycle figure in profiles/ is nominal):
# rocm-smi sampled every ~0.3 s beside (1) microbench --sustain (the pure v_mad_u64_u32 stream and the round's mix held for 8 s each), (2) the default bench step.  usage: tools/clock_sample.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
sample() {   # $1 = label, samples until the file $O/.stop exists
  while [ ! -e $O/.stop ]; do
    echo "{\"t\": $(date +%s.%N), \"leg\": \"$1\", \"smi\": $(rocm-smi -c -P --json 2>/dev/null | tr -d '\n')}"
    sleep 0.3
  done
}
rocm-smi -c -P --json > $O/idle.json 2>&1
rocm-smi --showmaxpower --showperflevel --showclkfrq > $O/caps.txt 2>&1
rm -f $O/.stop; sample microbench > $O/samples_microbench.jsonl & S=$!
timeout 300 mina_bridge_amd/microbench --sustain 8 > $O/microbench_sustain.jsonl 2>&1
touch $O/.stop; wait $S
rm -f $O/.stop; sample bench > $O/samples_bench.jsonl & S=$!
timeout 600 python bench.py --no-cpu-baseline --no-boundary --steps 200 --warmup 8 > $O/bench.json 2> $O/bench.err
touch $O/.stop; wait $S; rm -f $O/.stop
wc -l $O/samples_*.jsonl; tail -4 $O/microbench_sustain.jsonl
