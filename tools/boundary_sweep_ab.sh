#!/bin/bash
# dev tool: tools/boundary_ab.py under settings that are fixed per process (stream masks, pool size); one process per setting, medians of ROUNDS calls
# usage: tools/boundary_sweep_ab.sh SIZE ROUNDS "ENV1=a ENV2=b" "ENV1=c" ...
size=$1; rounds=$2; shift; shift
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python tools/boundary_ab.py $size $rounds 2>&1 | grep config
done
