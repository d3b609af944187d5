#!/bin/bash
# SQ counters for the dominant kernel (single lane so launches do not overlap); own pass, --kernel-trace only
TAG=${1:-r01c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -f csv -d $OUT/sq -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 4 --pipeline 1 --no-probes --no-cpu-baseline --no-boundary > $OUT/bench_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/sq2 -o s2 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 4 --pipeline 1 --no-probes --no-cpu-baseline --no-boundary > $OUT/bench_sq2.log 2>&1
ls $OUT/sq $OUT/sq2; tail -3 $OUT/bench_sq.log | cut -c1-300
