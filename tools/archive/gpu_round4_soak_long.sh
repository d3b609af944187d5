#!/bin/bash
# a longer randomised differential soak of the final build (lazy products in every 29-bit lane form) against the oracle, fresh seeds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04soak_long; mkdir -p $O
for t in "soak_sponge.py 300" "soak_lanes.py 300" "soak_verifier.py 300" "soak_boundary.py 600"; do set -- $t
  timeout $(( $2 + 600 )) python tools/$1 $2 > $O/${1%.py}.log 2>&1; echo "$1 rc=$?"; tail -1 $O/${1%.py}.log | cut -c1-400
done
