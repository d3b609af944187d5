# dev tool: BASELINE config C2 alone -- un-folded 2^16-base Vesta accumulator checks, 8 per call, over LANES lanes (bench.py's secondary key
# `c2_accumulator_only`); run it under `rocprofv3 --kernel-trace --stats` for the kernel breakdown.   usage: python tools/c2_rate.py [LANES] [CALLS]
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np, torch
import mina_bridge_amd as m
m.lib.tune_from_string(os.environ.get("MINA_TUNE", ""))      # e.g. MINA_TUNE=msm_fp29=0 (fields of mina_verify_tuning)
from bench import make_accumulators, CURVE_VESTA, ACC_K
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda:0")
ctx = m.MinaContext(0)
ctx.poseidon_set_params(0, m.poseidon_params.default_params_bytes(0)); ctx.poseidon_set_params(1, m.poseidon_params.default_params_bytes(1))
ctx.srs_create(CURVE_VESTA, 1 << 16)
if m.lib.verify_tuning_get().msm_fp29 >= 2: ctx.srs_split_table(CURVE_VESTA)
pre8, sg8 = make_accumulators(ctx, 8, 4242)
d_pre8 = torch.from_numpy(pre8.reshape(-1)).to(dev); d_sg8 = torch.from_numpy(sg8.reshape(-1)).to(dev); d_v8 = torch.zeros(8, dtype=torch.int32, device=dev)
ctx.set_pipeline(lanes)
c2 = lambda: ctx.accumulator_check_multi_dev(CURVE_VESTA, ACC_K, 8, d_pre8.data_ptr(), d_sg8.data_ptr(), d_v8.data_ptr())
for _ in range(32): c2()
ctx.synchronize(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(calls): c2()
ctx.synchronize(); dt = time.perf_counter() - t
assert os.environ.get("C2_NOASSERT") or d_v8.cpu().numpy().tolist() == [1] * 8      # C2_NOASSERT: timing probes that compute wrong sums (tools/probes)
print(json.dumps({"lanes": lanes, "checks_per_s": round(8 * calls / dt, 1), "us_per_check": round(dt / (8 * calls) * 1e6, 2)}))
