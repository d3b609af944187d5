#!/bin/bash
# dev tool: same-box A/B of two builds of the library -- usage: tools/ab_libs.sh <other.so> [rounds]; alternates the tracked build (A) and <other.so> (B)
# under `bench.py --no-cpu-baseline --no-boundary --steps 20 --warmup 3` and prints value / isolated kernel times of each run.
# NOASSERT=1: B is a timing experiment that computes WRONG values -- the verdict asserts are cut out of a temporary copy of bench.py and the probes are skipped
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
B=$1; N=${2:-2}; L=mina_bridge_amd/libminaverify.so
cp $L /tmp/lib_A.so
mkdir -p gpurun_out/ab
BENCH=bench.py; EXTRA=""
if [ -n "$NOASSERT" ]; then sed 's/^\(\s*\)assert verdicts_ok/\1verdicts_ok/; s/^\(\s*\)assert all(int(g.sum/\1all(int(g.sum/' bench.py > _bench_na.py; BENCH=_bench_na.py; EXTRA="--no-probes"; fi
for i in $(seq $N); do for v in A B; do
  if [ $v = A ]; then cp /tmp/lib_A.so $L; else cp $B $L; fi
  timeout 400 python $BENCH $EXTRA --no-cpu-baseline --no-boundary --steps 40 --warmup 3 > gpurun_out/ab/$v$i.json 2>> gpurun_out/ab/err.log
  python - $v $i <<PY
import json,sys
for l in open("gpurun_out/ab/%s%s.json" % (sys.argv[1], sys.argv[2])):
    if l.startswith('{"metric"'):
        d=json.loads(l); print(sys.argv[1], round(d["value"]), round(d["sustained"]["value"]) if d.get("sustained") else None, {k: round(v) for k,v in d["stage_us"]["isolated"].items() if v})
PY
done; done
cp /tmp/lib_A.so $L
rm -f _bench_na.py
