#!/bin/bash
# calibration of FETCH_SIZE for the MSM's access pattern (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern"):
# tools/probes/batch_affine_probe's xyzz_kernel gathers exactly one 64-B point per addition at random from a 64 MiB table (8 388 608 additions per launch
# = 536.9 MB of points + 33.6 MB of references)
OUT=$GRAFT_REPO_ROOT/gpurun_out/calib_fetch; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o f -- $GRAFT_REPO_ROOT/tools/probes/bin/batch_affine_probe > $OUT/probe.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/calib_fetch/fetch/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'FETCH_SIZE': acc[(r['Kernel_Name'].split('(')[0], r['Grid_Size'])].append(float(r['Counter_Value']))
for k, v in acc.items(): print(k, len(v), 'FETCH_SIZE KiB avg', sum(v)/len(v), '-> MB', sum(v)/len(v)*1024/1e6)
PY
