# dev tool: mina_merkle_roots wall time (host-buffer API) for BASELINE config C4's shape and a chip-filling batch
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import mina_bridge_amd as m
ctx = m.MinaContext(0)
ctx.poseidon_set_params(0, m.poseidon_params.default_params_bytes(0))
rng = np.random.default_rng(1)
depth = 35
for n in (256, 4096, 65536):
    leaves = rng.integers(0, 256, (n, 32), dtype=np.uint8); leaves[:, 31] &= 0x3f
    sib = rng.integers(0, 256, (n * depth, 32), dtype=np.uint8); sib[:, 31] &= 0x3f
    dirs = rng.integers(0, 2, n * depth, dtype=np.uint8)
    ctx.merkle_roots(0, leaves, sib, dirs, depth)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): ctx.merkle_roots(0, leaves, sib, dirs, depth)
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n} depth={depth}: {dt*1e3:.2f} ms per batch, {n/dt:.0f} paths/s, {n*depth/dt/1e6:.2f} M permutations/s")
