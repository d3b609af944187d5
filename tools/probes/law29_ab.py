#!/usr/bin/env python3
"""Kernel-time A/B of the two paths round 5 moved onto the 29-bit group law: the direct public-input commitments (lagrange.cuh) and the variable-base MSM.
Run under `rocprofv3 --kernel-trace --stats` once with MINA_TUNE=msm_fp29=1 and once with =0 (tools/gpu_round5.sh law29); prints nothing but a checksum."""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mina_bridge_amd import lib as L

L.tune_from_string(os.environ.get("MINA_TUNE", ""))
ctx = L.MinaContext(0)
curve = L.CURVE_PALLAS
ctx.srs_create(curve, 1 << 15)
rng = np.random.default_rng(5)
B, NPUB = 16384, 40
pub = rng.integers(0, 256, size=(B * NPUB, 32), dtype=np.uint8); pub[:, 31] &= 0x3f
h = hashlib.sha256()
for _ in range(5):
    h.update(ctx.public_input_commitment_batch(curve, 15, pub.reshape(-1), B).tobytes())
g = ctx.srs_lagrange_basis(curve, 15)[: 1 << 15]
n = 1 << 15
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x3f
for _ in range(5):
    h.update(ctx.msm(curve, g.reshape(n, 64), sc).tobytes())
print(h.hexdigest())
