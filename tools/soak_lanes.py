# dev tool: stress of the pipelined `_dev` path (what bench.py times): random lane counts, random group sizes, many calls issued
# back to back over the lanes with different inputs -- some tampered -- and every verdict word checked afterwards.  A race
# between lanes (shared scratch, stale workspace) would show up as a wrong verdict.  usage: soak_lanes.py SECONDS
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mina_bridge_amd as m
from oracle import oracle as O
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(time.time()) & 0xffffffff
rng = np.random.Generator(np.random.PCG64(seed))
print("seed", seed)
curve, k, fs = 1, 16, 0
ctx = m.MinaContext(0); ctx.srs_create(curve, 65536)
g = O.srs_create(curve, 65536, threads=os.cpu_count() or 8)[0]
_, endo_r = O.endo(curve)
POOL = []
for j in range(12):                                             # valid instances from the CPU oracle
    pre = rng.integers(0, 256, (k, 16), dtype=np.uint8)
    chals = np.stack([O.challenge_to_field(fs, pre[i].copy(), endo_r) for i in range(k)])
    POOL.append((pre, O.msm_pippenger(curve, g, O.b_poly_coefficients(fs, chals), threads=16)))
t0 = time.time(); rounds = 0; checks = 0
while time.time() - t0 < budget:
    rounds += 1
    lanes = int(rng.choice([2, 3, 8, 16])); ctx.set_pipeline(lanes)
    calls = int(rng.integers(lanes, 4 * lanes + 1))
    bufs, expect = [], []
    for c_ in range(calls):
        G = int(rng.choice([1, 2, 3, 4, 5, 8, 12]))
        pre = np.empty((G, k, 16), np.uint8); sg = np.empty((G, 64), np.uint8); exp = []
        for j in range(G):
            p, s = POOL[int(rng.integers(0, len(POOL)))]
            pre[j], sg[j] = p, s
            ok = 1
            if rng.random() < 0.25:
                if rng.random() < 0.5: pre[j, int(rng.integers(0, k)), int(rng.integers(0, 16))] ^= 1 << int(rng.integers(0, 8))
                else: sg[j] = POOL[(int(rng.integers(0, len(POOL))))][1] if rng.random() < 0.5 else g[int(rng.integers(0, 65536))]
                ok = int((sg[j] == s).all() and (pre[j] == p).all())
            exp.append(ok)
        d_pre = ctx.dev_upload(ctx.dev_malloc(pre.size), pre); d_sg = ctx.dev_upload(ctx.dev_malloc(sg.size), sg)
        d_v = ctx.dev_upload(ctx.dev_malloc(4 * G), np.full(G, 9, np.uint32).view(np.uint8))
        bufs.append((G, d_pre, d_sg, d_v)); expect.append(exp)
    for G, d_pre, d_sg, d_v in bufs:                            # nothing waits between the calls
        if G == 1 and rng.random() < 0.5: ctx.accumulator_check_dev(curve, k, 1, d_pre, d_sg, 0, d_v)
        else: ctx.accumulator_check_multi_dev(curve, k, G, d_pre, d_sg, d_v)
    got = [ctx.dev_download(b[3], 4 * b[0]).view(np.uint32).tolist() for b in bufs]
    for b in bufs:
        for ptr in b[1:]: ctx.dev_free(ptr)
    assert got == expect, ("lanes", seed, rounds, lanes, calls, [i for i, (a, e) in enumerate(zip(got, expect)) if a != e])
    checks += sum(len(e) for e in expect)
ctx.set_pipeline(1)
print(f"lane soak ok: {rounds} rounds, {checks} verdicts in {time.time() - t0:.0f}s")
