#!/bin/bash
# dev tool: bytes -> bools rate of mina_verify_state_batch (tools/boundary_rate.py) under the pipeline's knobs; one process per setting
# usage: tools/boundary_sweep.sh SIZE "ENV1=a ENV2=b" "ENV1=c" ...
size=$1; shift
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python tools/boundary_rate.py $size 2>&1 | grep proofs_per_call
done
