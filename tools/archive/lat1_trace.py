import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
import bench, ctypes
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_partial_inputs as P
ctx = m.MinaContext(0)
for f in (0, 1):
    ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(1, 1 << 16); ctx.srs_create(0, 1 << 16)
B = 1
(hj, keep), _ = P.build_batch(ctx, m, B, seed=5)
kp, op, acc, _ = P.build_full_section(ctx, m, B)
for name in ("sponge_state", "sponge_pos", "cip", "evalpoints", "evalscale", "polyscale", "comms", "public_inputs"):
    setattr(hj, name, None)
keep = list(keep) + [kp]
for name, arr in list(op.items()) + list(acc.items()):
    arr = np.ascontiguousarray(arr); keep.append(arr); setattr(hj, name, arr.ctypes.data)
hj.n_comms = 47
hj.kimchi = ctypes.addressof(kp[0])
for _ in range(3):
    assert ctx.state_job_batch((hj, keep)).tolist() == [1] * B
