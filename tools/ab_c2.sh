#!/bin/bash
# dev tool: same-box A/B of two builds of the library on BASELINE config C2 (tools/c2_rate.py, 16 lanes): usage tools/ab_c2.sh <other.so> [rounds]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
B=$1; N=${2:-3}; L=mina_bridge_amd/libminaverify.so
cp $L /tmp/lib_A.so
for i in $(seq $N); do for v in A B; do
  if [ $v = A ]; then cp /tmp/lib_A.so $L; else cp $B $L; fi
  echo -n "$v "; timeout 300 python tools/c2_rate.py 16 400 2>/dev/null | tail -1
done; done
cp /tmp/lib_A.so $L
