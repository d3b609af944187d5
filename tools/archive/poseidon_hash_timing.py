# dev tool: mina_poseidon_hash wall time (48-element messages) across the kernel-form thresholds
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import mina_bridge_amd as m
c = m.MinaContext(0); c.poseidon_set_params(0, m.poseidon_params.default_params_bytes(0))
rng = np.random.default_rng(1)
for n in (17408, 60000, 69632, 140000, 300000):
    msgs = rng.integers(0, 256, (n * 48, 32), dtype=np.uint8); msgs[:, 31] &= 0x3f
    c.poseidon_hash(0, msgs, n, 48)
    t0 = time.perf_counter(); c.poseidon_hash(0, msgs, n, 48); dt = time.perf_counter() - t0
    print(f"n={n}: {dt*1e3:.1f} ms, {n*25/dt/1e6:.1f} M perm/s")
