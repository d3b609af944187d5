#!/bin/bash
# dev tool: bytes -> bools rate of mina_verify_state_batch (tools/boundary_rate.py) under the pipeline's knobs; one process per setting
# usage: tools/boundary_sweep.sh SIZE "MINA_TUNE=chunk=4096,slots=8" "MINA_TUNE=early_sub=512" ...   (fields of mina_verify_tuning; process-level settings such as GPU_MAX_HW_QUEUES as plain env)
size=$1; shift
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python tools/boundary_rate.py $size 2>&1 | grep proofs_per_call
done
