# dev: host time to ISSUE one full Proof-of-State job (mina_state_job_batch_dev returns after queueing) vs the time until it is done
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import ctypes
import numpy as np
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = m.MinaContext(0)
for f in (0, 1):
    ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(1, 1 << 16); ctx.srs_create(0, 1 << 16)
(hj, keep), kp, _ = bench.build_full_job(ctx, m, B, 5)
dev = torch.device("cuda", 0)
tens = []
def up(struct, cls, keepl):
    d = cls(); ctypes.memmove(ctypes.byref(d), ctypes.byref(struct), ctypes.sizeof(cls))
    by = {a.ctypes.data: a for a in keepl if isinstance(a, np.ndarray)}
    for name in cls.POINTER_FIELDS:
        addr = getattr(struct, name)
        if addr and addr in by:
            t = torch.from_numpy(np.array(by[addr].view(np.uint8).reshape(-1))).to(dev); tens.append(t); setattr(d, name, t.data_ptr())
    return d
dj = up(hj, m.lib.StateJobs, keep)
dk = up(kp[0], m.lib.KimchiProofs, kp[1])
hst, hkeep = next(a for a in kp[1] if isinstance(a, tuple))
dst = up(hst, m.lib.PicklesStatements, hkeep)
dk.statements = ctypes.addressof(dst); dj.kimchi = ctypes.addressof(dk)
ctx.state_jobs_prepare(15, 40)
ctx.set_pipeline(16)
outs = [torch.zeros(B + 4, dtype=torch.int32, device=dev) for _ in range(16)]
for i in range(32):
    ctx.state_job_batch_dev(dj, outs[i % 16].data_ptr(), outs[i % 16].data_ptr() + 4 * B)
ctx.synchronize()
t0 = time.perf_counter()
for i in range(64):
    ctx.state_job_batch_dev(dj, outs[i % 16].data_ptr(), outs[i % 16].data_ptr() + 4 * B)
t1 = time.perf_counter()
ctx.synchronize()
t2 = time.perf_counter()
print({"proofs_per_call": B, "host_issue_ms_per_call": round((t1 - t0) / 64 * 1e3, 3), "done_ms_per_call": round((t2 - t0) / 64 * 1e3, 3)})
