# dev tool: randomised differential soak against the CPU oracle (fresh seeds every run): fixed-base / variable-base / multi-
# problem MSMs over random shapes in both context modes, accumulator checks with random tampering.  usage: soak.py SECONDS
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mina_bridge_amd as m
from oracle import oracle as O
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
MOD = {0: Q, 1: P}
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(time.time()) & 0xffffffff
rng = np.random.Generator(np.random.PCG64(seed))
print("seed", seed)
ctx = m.MinaContext(0)
G = {}
for curve in (0, 1):
    ctx.srs_create(curve, 65536)
    G[curve] = O.srs_create(curve, 65536, threads=os.cpu_count() or 8)[0]


def scalars(n, mod, kind):
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8); a[:, 31] &= 0x1f
    if kind == 1: a[:] = a[0]
    elif kind == 2: a[:, 4:] = 0
    elif kind == 3: a[rng.random(n) < 0.7] = 0
    elif kind == 4: a[:] = np.frombuffer((mod - 1).to_bytes(32, "little"), np.uint8)
    elif kind == 5 and n > 1: a[::2] = a[1]
    return a


t0 = time.time(); it = 0
while time.time() - t0 < budget:
    it += 1
    curve = int(rng.integers(0, 2)); mod = MOD[curve]; g = G[curve]
    ctx.set_pipeline(int(rng.choice([1, 1, 2, 5])))
    what = int(rng.integers(0, 4))
    if what == 0:
        n = int(rng.choice([1, 2, 255, 256, 257, 4095, 4096, 30000, 65535, 65536]))
        sc = scalars(n, mod, int(rng.integers(0, 6)))
        assert (ctx.msm_srs(curve, sc) == O.msm_pippenger(curve, g[:n], sc, threads=16)).all(), ("fixed", seed, it)
    elif what == 1:
        n = int(rng.choice([1, 3, 2047, 2048, 2049, 9000, 131071, 131072, 200000]))
        pts = g[rng.integers(0, 65536, n)].copy()
        pts[rng.random(n) < 0.02] = 0
        sc = scalars(n, mod, int(rng.integers(0, 6)))
        assert (ctx.msm(curve, pts, sc) == O.msm_pippenger(curve, pts, sc, threads=16)).all(), ("var", seed, it)
    elif what == 2:
        n = int(rng.choice([1, 100, 256, 1000, 4096, 20000, 65536])); nprob = int(rng.integers(1, 10))
        if n * nprob > 300000: nprob = max(1, 300000 // n)
        sc = np.stack([scalars(n, mod, int(rng.integers(0, 6))) for _ in range(nprob)])
        got = ctx.msm_srs_multi(curve, sc, nprob)
        for j in range(nprob):
            assert (got[j] == O.msm_pippenger(curve, g[:n], sc[j], threads=16)).all(), ("multi", seed, it, j)
    else:
        k = int(rng.choice([4, 9, 16])); cnt = int(rng.integers(1, 12)); fs = 1 if curve == 0 else 0
        _, endo_r = O.endo(curve)
        pre = rng.integers(0, 256, (cnt, k, 16), dtype=np.uint8)
        sg = np.empty((cnt, 64), np.uint8)
        for j in range(cnt):
            chals = np.stack([O.challenge_to_field(fs, pre[j, i].copy(), endo_r) for i in range(k)])
            sg[j] = O.msm_pippenger(curve, g[: 1 << k], O.b_poly_coefficients(fs, chals), threads=16)
        exp = [1] * cnt
        for j in range(cnt):
            if rng.random() < 0.3:
                if rng.random() < 0.5: pre[j, int(rng.integers(0, k)), int(rng.integers(0, 16))] ^= 1 << int(rng.integers(0, 8))
                else: sg[j] = g[int(rng.integers(0, 65536))]
                exp[j] = 0
        assert ctx.accumulator_check_multi(curve, k, pre.reshape(-1), sg).tolist() == exp, ("acc", seed, it)
ctx.set_pipeline(1)
print(f"soak ok: {it} random cases in {time.time() - t0:.0f}s")
