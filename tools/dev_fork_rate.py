# dev tool (round 6): the device-resident job under the leg fork (mina_verify_tuning.dev_fork / dev_piece_waves) -- ONE process, one job of B proofs built once, then per
# setting "lanes:tune": calls/s with `lanes` calls in flight and the latency of a lone call.  usage: python tools/dev_fork_rate.py B "4:dev_piece_waves=1024" "1:dev_fork=0" ...
# (settings that change a stream's creation mode -- dev_fork bits 1, 2 -- need a fresh process: tools/dev_fork_sweep.sh)
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
import bench
B = int(sys.argv[1])
ctx = m.MinaContext(0)
for f in (0, 1):
    ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(1, 1 << 16); ctx.srs_create(0, 1 << 16)
(hj, keep), kp, _, _ = bench.build_full_job(ctx, m, B, 5, 0)
dev = torch.device("cuda", 0)
dj, dk, tens = bench.device_jobs(m, hj, keep, kp, dev)
ctx.state_jobs_prepare(15, 40)
outs = [torch.zeros(B + 4, dtype=torch.int32, device=dev) for _ in range(32)]
torch.cuda.synchronize()
def call(i): ctx.state_job_batch_dev(dj, outs[i % 32].data_ptr(), outs[i % 32].data_ptr() + 4 * B)
for spec in sys.argv[2:]:
    lanes, _, tune = spec.partition(":")
    lanes = int(lanes)
    ctx.synchronize()
    m.lib.tune_from_string(tune if tune and tune != "-" else "dev_fork=1")
    ctx.set_pipeline(lanes)
    for i in range(2 * lanes): call(i)
    ctx.synchronize()
    reps = max(24, 6 * lanes)
    t0 = time.perf_counter()
    for i in range(reps): call(i)
    ctx.synchronize()
    many = (time.perf_counter() - t0) / reps
    assert outs[(reps - 1) % 32].cpu().numpy().tolist() == [1] * B + [1, 0, 1, 0]
    ctx.set_pipeline(1)
    for i in range(2): call(i)
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(6):
        call(i); ctx.synchronize()
    lone = (time.perf_counter() - t0) / 6
    free, total = torch.cuda.mem_get_info()
    print(json.dumps({"B": B, "lanes": lanes, "tune": tune, "proofs_per_s": round(B / many), "ms_per_call_in_flight": round(many * 1e3, 2), "lone_call_ms": round(lone * 1e3, 2),
                      "hbm_GiB": round((total - free) / 2**30, 1)}), flush=True)
