#!/usr/bin/env python3
"""Generates tests/golden/vectors.json with the CPU oracle (run in the build container):
    python tests/golden/gen_golden.py
There is no reference implementation to import (the verifier is Rust in un-vendored crates), so
these vectors pin *regressions* of the restatement and give the GPU path a fixed target; the only vectors that come
from the reference itself are srs_head_tail.json (bytes of srs/*.srs) and the two SRS sha256 digests."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rand_scalars  # noqa: E402
from oracle import oracle as O, pasta_ref as R  # noqa: E402
import mina_bridge_amd.poseidon_params as PP  # noqa: E402

MODS = {0: R.P, 1: R.Q}
out = {"constants": {}, "msm": [], "b_poly": [], "to_group": [], "poseidon": [], "to_field": []}
for curve, name in ((0, "pallas"), (1, "vesta")):
    eq, er = O.endo(curve)
    out["constants"][name] = {
        "endo_q": O.le_to_int(eq), "endo_r": O.le_to_int(er),
        "sqrt_neg3_base": R.BWParams(R.base_modulus(curve)).sqrt_neg_three_u_squared,
    }
srs = {c: O.srs_create(c, 65536, threads=os.cpu_count() or 4) for c in (0, 1)}
for curve in (0, 1):
    r = R.scalar_modulus(curve)
    g = srs[curve][0]
    for n in (1, 2, 31, 32, 1024, 65536):
        for dist in ("uniform", "bits128"):
            seed = 1000 + 7 * n + curve + (0 if dist == "uniform" else 3)
            sc = rand_scalars(n, r, seed=seed, bits=128 if dist == "bits128" else None)
            res = O.msm_pippenger(curve, g[:n], sc, threads=os.cpu_count() or 4)
            out["msm"].append({"curve": curve, "n": n, "dist": dist, "seed": seed, "result": res.tobytes().hex()})
    fs = O.scalar_field_of(curve)
    for k in (3, 10, 16):
        ch = rand_scalars(k, MODS[fs], seed=2000 + k + curve)
        co = O.b_poly_coefficients(fs, ch)
        x = rand_scalars(1, MODS[fs], seed=2100 + k)[0]
        out["b_poly"].append({"field": fs, "k": k, "seed": 2000 + k + curve, "coeffs_sha256": hashlib.sha256(co.tobytes()).hexdigest(),
                              "x_seed": 2100 + k, "eval": O.b_poly(fs, ch, x).tobytes().hex()})
    fb = O.base_field_of(curve)
    t = rand_scalars(16, MODS[fb], seed=3000 + curve)
    out["to_group"].append({"curve": curve, "seed": 3000 + curve, "points_sha256": hashlib.sha256(O.to_group(curve, t).tobytes()).hexdigest(),
                            "first": O.to_group(curve, t)[0].tobytes().hex()})
    _, endo_r = O.endo(curve)
    pre = rand_scalars(8, MODS[fs], seed=3100 + curve)[:, :16]
    out["to_field"].append({"field": fs, "seed": 3100 + curve,
                            "out": [O.challenge_to_field(fs, p.copy(), endo_r).tobytes().hex() for p in pre]})
for field in (0, 1):
    params = PP.default_params_bytes(field)
    st = rand_scalars(12, MODS[field], seed=4000 + field).reshape(4, 96)
    out["poseidon"].append({"field": field, "constants": PP.NAME, "params_sha256": hashlib.sha256(params).hexdigest(), "seed": 4000 + field,
                            "permuted_sha256": hashlib.sha256(O.poseidon_permute(field, params, st).tobytes()).hexdigest(),
                            "hash_of_empty": O.poseidon_hash(field, params, np.zeros(0, np.uint8)).tobytes().hex()})
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.json"), "w"), indent=1)
print("wrote vectors.json:", {k: len(v) for k, v in out.items()})
