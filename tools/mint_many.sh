#!/bin/bash
# Mint the distinct complete wrap proofs of tests/golden/statement_k15_many.npz with the repo's CPU oracle prover (tests/golden/gen_statement_fixture.py,
# ~6 min per proof per core), WORKERS processes at nice 19, proofs 4 .. 4 + WORKERS*PER - 1 (0 .. 3 are tests/golden/statement_k15.json).  Parts go to
# .mint_parts/ (ignored); `python tests/golden/encode_statement_fixture.py --many` packs whatever parts exist.
cd "$(dirname "$0")/.."
WORKERS=${1:-6}; PER=${2:-42}
mkdir -p .mint_parts
for w in $(seq 0 $((WORKERS - 1))); do
  S=$((4 + w * PER))
  nohup nice -n 19 python tests/golden/gen_statement_fixture.py $PER --start $S --out .mint_parts/part_$S.json > .mint_parts/part_$S.log 2>&1 &
  echo "worker $w: proofs $S .. $((S + PER - 1)) pid $!"
done
