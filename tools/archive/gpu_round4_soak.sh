#!/bin/bash
# randomised differential soaks of the round-4 build against the oracle (fresh seeds)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04soak; mkdir -p $O
for t in "soak.py 240" "soak_ipa.py 120" "soak_lanes.py 120" "soak_sponge.py 90" "soak_verifier.py 180" "soak_boundary.py 300"; do set -- $t
  timeout $(( $2 + 600 )) python tools/$1 $2 > $O/${1%.py}.log 2>&1; echo "$1 rc=$?"; tail -1 $O/${1%.py}.log | cut -c1-400
done
