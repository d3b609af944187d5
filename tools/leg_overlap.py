# dev: which leg of the Proof-of-State job overlaps across pipeline lanes at small call sizes?  For each leg alone (state hashes,
# accumulator, kimchi + opening) and for the whole job: ms per call when calls are issued one at a time, and with 16 calls in flight.
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import ctypes, json
import numpy as np
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = m.MinaContext(0)
for f in (0, 1):
    ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(1, 1 << 16); ctx.srs_create(0, 1 << 16)
(hj, keep), kp, _, _ = bench.build_full_job(ctx, m, B, 5)
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
outs = [torch.zeros(B + 4, dtype=torch.int32, device=dev) for _ in range(16)]
full = (dj.with_states, dj.with_ipa, dj.with_accumulator, dj.kimchi, dj.npub)
def rate(lanes, reps):
    ctx.set_pipeline(lanes)
    for i in range(2 * lanes): ctx.state_job_batch_dev(dj, outs[i % 16].data_ptr(), outs[i % 16].data_ptr() + 4 * B)
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        ctx.state_job_batch_dev(dj, outs[i % 16].data_ptr(), outs[i % 16].data_ptr() + 4 * B)
        if lanes == 1: ctx.synchronize()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for name, cfg in (("state hashes", (1, 0, 0, 0, 0)), ("accumulator", (0, 0, 1, 0, 0)), ("kimchi + opening", (0, full[1], 0, full[3], full[4])), ("whole job", full)):
    dj.with_states, dj.with_ipa, dj.with_accumulator, dj.kimchi, dj.npub = cfg
    one = rate(1, 12); many = rate(16, 96)
    print(json.dumps({"leg": name, "proofs_per_call": B, "ms_per_call_alone": round(one, 3), "ms_per_call_16_in_flight": round(many, 3), "overlap": round(one / many, 2)}), flush=True)
