# dev tool: limit check at the largest supported SRS depth (2^20): regeneration, 2^20-base fixed MSM, 700 k-point variable MSM vs the CPU oracle
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import mina_bridge_amd as m
from oracle import oracle as O
from conftest import rand_scalars
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
curve, k = 1, 20
c = m.MinaContext(0)
t0 = time.time(); c.srs_create(curve, 1 << k); print("gpu srs", time.time() - t0)
t0 = time.time(); g, h = O.srs_create(curve, 1 << k, threads=os.cpu_count()); print("cpu srs", time.time() - t0)
assert (c.srs_get_g(curve, (1 << k) - 1000, 1000) == g[-1000:]).all()
sc = rand_scalars(1 << k, P, seed=5)
t0 = time.time(); got = c.msm_srs(curve, sc); print("gpu msm", time.time() - t0)
t0 = time.time(); exp = O.msm_pippenger(curve, g, sc, threads=32); print("cpu msm", time.time() - t0)
assert (got == exp).all()
got = c.msm(curve, g[:700000], sc[:700000]); assert (got == O.msm_pippenger(curve, g[:700000], sc[:700000], threads=32)).all()
print("k=20 ok")
