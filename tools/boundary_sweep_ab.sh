#!/bin/bash
# dev tool: tools/boundary_ab.py under settings that are fixed per process (stream masks, pool size); one process per setting, medians of ROUNDS calls
# usage: tools/boundary_sweep_ab.sh SIZE ROUNDS "MINA_TUNE=chunk=4096,slots=8" "MINA_TUNE=early_sub=512" ...   (fields of mina_verify_tuning; process-level settings such as GPU_MAX_HW_QUEUES as plain env)
size=$1; rounds=$2; shift; shift
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python tools/boundary_ab.py $size $rounds 2>&1 | grep config
done
