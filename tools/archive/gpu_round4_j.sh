#!/bin/bash
# per-kernel durations of a C2 call (8 checks, ONE lane: no overlap) with the accumulate kernels on 29-bit limbs and on 8 x 32
cd /tmp && export TMPDIR=/tmp
for t in 1 0; do
  O=$GRAFT_REPO_ROOT/gpurun_out/prof_c2_fp29_$t; mkdir -p $O
  MINA_TUNE=msm_fp29=$t rocprofv3 --kernel-trace --stats -f csv -d $O -o t -- python $GRAFT_REPO_ROOT/tools/c2_rate.py 1 100 > $O/log.txt 2>&1
  python - <<PY
import csv, glob
f = glob.glob('$O/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows if int(r['Calls']) >= 100)
print('msm_fp29=$t: sum of kernel time per call (us):', round(tot / 132 / 1e3, 1))
for r in rows[:12]:
    if int(r['Calls']) >= 100: print('  ', r['Name'][:58].ljust(58), r['Calls'], 'avg us', round(float(r['AverageNs']) / 1e3, 1), round(100 * float(r['TotalDurationNs']) / tot, 1), '%')
PY
done
