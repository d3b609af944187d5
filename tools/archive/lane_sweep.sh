# dev: proofs/s of the full job vs pipeline lanes at a given proofs-per-call (default 256)
B=${1:-256}
for p in 1 2 4 8 16; do
  echo -n "jobs $B lanes $p: "
  python bench.py --no-cpu-baseline --no-probes --jobs $B --pipeline $p --steps 60 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d["value"]), round(d["ms_per_step"],2))'
done
