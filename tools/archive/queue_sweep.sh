# dev: proofs/s of the full job at a small proofs-per-call vs GPU_MAX_HW_QUEUES (16 pipeline lanes)
B=${1:-256}
for q in 12 16 18 20 24 32; do
  echo -n "jobs $B queues $q: "
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-probes --jobs $B --pipeline 16 --steps 64 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d["value"]), round(d["ms_per_step"],2))'
done
