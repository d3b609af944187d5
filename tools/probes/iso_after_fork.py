# probe (round 6): is an isolated state-hash launch slower in a process that has run forked jobs?  (bench.py's isolated figure read 37 - 40 ms in such a process, 33.8 ms otherwise)
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
import bench
B = 16384
ctx = m.MinaContext(0)
for f in (0, 1):
    ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(1, 1 << 16); ctx.srs_create(0, 1 << 16)
(hj, keep), kp, _, _ = bench.build_full_job(ctx, m, B, 5, 0)
dev = torch.device("cuda", 0)
dj, dk, tens = bench.device_jobs(m, hj, keep, kp, dev)
ctx.state_jobs_prepare(15, 40)
outs = [torch.zeros(B + 4, dtype=torch.int32, device=dev) for _ in range(8)]
torch.cuda.synchronize()
def call(i): ctx.state_job_batch_dev(dj, outs[i % 8].data_ptr(), outs[i % 8].data_ptr() + 4 * B)
def iso(tag):
    ctx.synchronize(); m.lib.tune_from_string("dev_fork=0"); ctx.set_pipeline(1)
    for i in range(2): call(i)
    ctx.synchronize(); ctx.prof_enable(1 << 11)
    for i in range(6): call(i)
    p = ctx.prof_read(); ctx.prof_enable(0)
    n, ms = p["pstate_hash"]
    print(json.dumps({"when": tag, "pstate_hash_us": round(ms / n * 1e3), "launches": n}), flush=True)
iso("fresh process")
for lanes, tune in ((4, "dev_fork=1"), (6, "dev_fork=1")):
    ctx.synchronize(); m.lib.tune_from_string(tune); ctx.set_pipeline(lanes)
    for i in range(4 * lanes): call(i)
    iso(f"after {lanes} forked lanes")
ctx.synchronize(); m.lib.tune_from_string("dev_fork=0"); ctx.set_pipeline(20)
for i in range(40): call(i)
iso("after 20 plain lanes")
