# dev tool (round 6): BASELINE C2 as written -- ONE un-folded 2^16 Vesta accumulator check per call, one call at a time, nothing else on the chip.
# Run plain for wall time, or under rocprofv3 (--kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE: tools/gpu_round6.sh c2single) for the per-kernel sums.
# usage: python tools/c2_single.py [CALLS]
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
import bench
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = m.MinaContext(0)
for f in (0, 1):
    ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(1, 1 << 16)
dev = torch.device("cuda", 0)
pre, sg = bench.make_accumulators(ctx, 1, 4242)
d_pre = torch.from_numpy(pre.reshape(-1)).to(dev); d_sg = torch.from_numpy(sg.reshape(-1)).to(dev); d_v = torch.zeros(1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
one = lambda: ctx.accumulator_check_dev(1, 16, 1, d_pre.data_ptr(), d_sg.data_ptr(), 0, d_v.data_ptr())
for _ in range(8): one()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    one(); ctx.synchronize()
dt = (time.perf_counter() - t0) / calls
assert int(d_v.cpu()[0]) == 1
print(json.dumps({"single_check_wall_us": round(dt * 1e6, 1), "calls": calls, "warm_calls": 8, "algorithmic_GBps_wall": round((65536 * 96 + 96) / dt / 1e9, 2)}))
