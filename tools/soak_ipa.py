# dev tool: randomised differential soak of mina_ipa_batch_check against the CPU restatement's verifier (fresh seeds): random
# shapes, random single-byte tampering of any field -- the two verdicts must agree on every batch.  usage: soak_ipa.py SECONDS
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mina_bridge_amd as m
from oracle import oracle as O, ipa_ref as I, pasta_ref as R
from ipa_helpers import mint, to_abi
from test_ipa_fullsize import to_oracle_entry
import test_ipa_fullsize as TF
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(time.time()) & 0xffffffff
rng = np.random.Generator(np.random.PCG64(seed))
print("seed", seed)
ctx = m.MinaContext(0)
G = {}
for curve in (0, 1):
    ctx.poseidon_set_params(curve, m.poseidon_params.default_params_bytes(curve))
    ctx.srs_create(curve, 1024)
    G[curve] = O.srs_create(curve, 1024, threads=8)
t0 = time.time(); it = 0; accepted = 0; rejected = 0
while time.time() - t0 < budget:
    it += 1
    curve = int(rng.integers(0, 2)); g, h = G[curve]
    k = int(rng.integers(1, 6)); npts = int(rng.integers(1, 4)); batch = int(rng.integers(1, 4))
    ops = []
    for b in range(batch):
        e, sp = mint(curve, g, h, k, n_polys=int(rng.integers(1, 5)), n_points=npts, seed=int(rng.integers(0, 1 << 30)))
        ops.append(to_abi(e, sp))
    if rng.random() < 0.6:                                       # tamper one byte of one field of one opening
        o = ops[int(rng.integers(0, batch))]
        key = str(rng.choice(["lr", "delta", "sg", "z1", "z2", "evalpoints", "comms", "combined_inner_product", "polyscale", "evalscale", "sponge_state"]))
        a = o[key].copy(); a[int(rng.integers(0, a.size))] ^= 1 << int(rng.integers(0, 8)); o[key] = a
    rb = int(rng.integers(1, 1 << 62)); sb = int(rng.integers(1, 1 << 62))
    got = ctx.ipa_batch_check(curve, ops, O.int_to_le(rb), O.int_to_le(sb))
    # the CPU verdict: malformed encodings (non-canonical / off-curve) count as reject, as in upstream's deserialiser
    p_base = R.base_modulus(curve); r_sc = R.scalar_modulus(curve)
    wellformed = True
    for o in ops:
        for key in ("lr", "delta", "sg", "comms"):
            for pt in o[key].reshape(-1, 64):
                x, y = O.le_to_int(pt[:32]), O.le_to_int(pt[32:])
                if x >= p_base or y >= p_base or not (not pt.any() or O.is_on_curve(curve, pt)): wellformed = False
        for key in ("z1", "z2", "evalpoints", "combined_inner_product", "polyscale", "evalscale"):
            for v in o[key].reshape(-1, 32):
                if O.le_to_int(v) >= r_sc: wellformed = False
        for v in o["sponge_state"].reshape(-1, 32):
            if O.le_to_int(v) >= p_base: wellformed = False
    exp = False
    if wellformed:
        TF.FX = {"curve": curve}                                 # to_oracle_entry reads the curve from the module's fixture
        exp = I.ipa_verify_batch(curve, g[: 1 << k], O.bytes_to_point(h), [to_oracle_entry(O, o) for o in ops], rb, sb)
    assert got == exp, ("ipa", seed, it, curve, k, batch, got, exp)
    accepted += got; rejected += (not got)
print(f"ipa soak ok: {it} random batches in {time.time() - t0:.0f}s ({accepted} accepted, {rejected} rejected)")
