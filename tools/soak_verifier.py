#!/usr/bin/env python3
"""Randomised differential soak of the round-2 verifier stages against the CPU oracle, fresh seeds every run:
  * Pickles statement -> 40 public inputs (api_pickles.hip, batch form): random statements of every shape (0..2 previous accumulators,
    optional + chunked evaluations, every installed step domain, both sponge forms via batch sizes around 1024), random malformed ones
    (non-canonical element / unknown domain / bad branch byte) must clear exactly their own flag;
  * kimchi oracles + to_batch rows (api_kimchi.hip): random field content in the shape of a proof (it need not verify: the rows are a
    function of the bytes) on a random synthetic index -- sponge hand-over, ft_eval0, v, u, zeta, combined inner product, every
    commitment row incl. the chunked ft commitment -- equal to oracle/kimchi_ref.py field for field;
  * the batch fold on the matrix cores (bpoly_mfma.cuh) against the VALU fold kernel (MINA_BPOLY_MFMA toggled per context is not
    possible, so: against the oracle) at random (k, batch).
usage: soak_verifier.py SECONDS     prints one JSON line with the case counts; exits non-zero on the first mismatch."""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.setrecursionlimit(10000)
import numpy as np
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
from ipa_helpers import poseidon_pp
from kimchi_helpers import install_index, install_step_index, kimchi_arrays, make_step_index, statements_soa
from oracle import kimchi_ref as K, oracle as O, pasta_ref as R, pickles_ref as PK
from wire_writers import synth_wrap_proof

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(time.time())
rng = random.Random(seed0)
ctx = m.MinaContext(0)
for f in (0, 1):
    ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(0, 1 << 8); ctx.srs_create(1, 1 << 8)
g, h = O.srs_create(0, 1 << 8, threads=4)
hp = O.bytes_to_point(h)
pb, ps = poseidon_pp(0), poseidon_pp(1)
counts = {"statements": 0, "malformed_statements": 0, "kimchi_rows": 0, "fold_outputs": 0}
t_end = time.time() + budget
rnd = 0
while time.time() < t_end:
    rnd += 1
    # ---- a fresh synthetic wrap index + step index
    klog = rng.choice([5, 6, 7]); npub = rng.choice([1, 5, 8, 13])
    circ = K.synthetic_circuit(0, g[: 1 << klog], hp, pb, ps, klog, npub, seed=rng.getrandbits(30))
    install_index(ctx, circ.index)
    step = make_step_index(rng.getrandbits(30))
    install_step_index(ctx, step)
    comms = list(circ.index.sigma_comm) + list(circ.index.coefficients_comm) + list(circ.index.selector_comm)
    # ---- statements
    n_old = rng.randrange(3); count = rng.choice([3, 7, 40])
    optional = [rng.randrange(3) == 0 for _ in range(19)]
    wraps, apps = [], []
    for i in range(count):
        w = synth_wrap_proof(rng, k=klog)
        w["prev_optional"] = [(([rng.randrange(PK.P)], [rng.randrange(PK.P)]) if o else None) for o in optional]
        if rng.randrange(2):
            j = rng.randrange(43); w["prev_evals"][j] = ([rng.randrange(PK.P) for _ in range(2)], [rng.randrange(PK.P) for _ in range(2)])
        w["step_comms"] = w["step_comms"][:n_old]; w["step_old_chals"] = w["step_old_chals"][:n_old]
        wraps.append(w); apps.append(rng.randrange(PK.P))
    want = [PK.statement_public_input(w, step, comms, a, pb, ps)[0] for w, a in zip(wraps, apps)]
    no, ne, sec = statements_soa(wraps, apps)
    reps = rng.choice([1, 1, 1100 // count + 1])            # sometimes past the 8-lane / 3-lane threshold
    big = {k: np.tile(v.reshape(count, -1), (reps, 1)).reshape(-1) if v.size >= count else v for k, v in sec.items()}
    pub, ok = ctx.pickles_public_inputs_batch(ctx.make_pickles_statements(no, ne, big), count * reps)
    assert ok.all(), ("statement flagged", seed0, rnd)
    got = pub.reshape(reps, count, 40, 32)
    for b in range(count):
        assert [O.le_to_int(x) for x in got[0, b]] == want[b], ("statement mismatch", seed0, rnd, b)
    assert (got == got[:1]).all(), ("tiled statements differ", seed0, rnd)
    counts["statements"] += count * reps
    bad = {k: v.copy() for k, v in sec.items()}
    victims = rng.sample(range(count), 2)
    bad["prev_evals"].reshape(count, -1)[victims[0], 32 * rng.randrange(2 * ne):][:32] = 0xff
    if rng.randrange(2):
        bad["misc"].reshape(count, 32)[victims[1], 0] = rng.choice([3, 9, 17, 40])
    else:
        bad["misc"].reshape(count, 32)[victims[1], 1] = rng.choice([3, 7, 255])
    _, ok2 = ctx.pickles_public_inputs_batch(ctx.make_pickles_statements(no, ne, bad), count)
    assert ok2.tolist() == [0 if b in victims else 1 for b in range(count)], ("malformed flags", seed0, rnd, victims, ok2.tolist())
    counts["malformed_statements"] += 2
    # ---- kimchi rows on random proof-shaped content (valid curve points, arbitrary scalars)
    nb = rng.choice([2, 5])
    pts = lambda n: [O.bytes_to_point(g[rng.randrange(1 << klog)]) for _ in range(n)]
    proofs, publics = [], []
    for _ in range(nb):
        chals = [[rng.randrange(R.Q) for _ in range(klog)] for _ in range(2)]
        prf = {"prev": [(ch, cm) for ch, cm in zip(chals, pts(2))], "w_comm": pts(15), "z_comm": pts(1)[0], "t_comm": pts(7),
               "evals": [(rng.randrange(R.Q), rng.randrange(R.Q)) for _ in range(43)], "ft_eval1": rng.randrange(R.Q),
               "opening": {"lr": [(p, q) for p, q in zip(pts(klog), pts(klog))], "delta": pts(1)[0], "sg": pts(1)[0], "z1": rng.randrange(R.Q), "z2": rng.randrange(R.Q)}}
        proofs.append(prf); publics.append([rng.randrange(R.Q) for _ in range(npub)])
    arrays, _ = kimchi_arrays(proofs, publics)
    rows = ctx.kimchi_to_batch(ctx.make_kimchi_proofs(nb, 2, npub, arrays), klog)
    assert not rows["malformed"][0]
    for b in range(nb):
        o, entry = K.oracles_and_batch(circ.index, proofs[b], publics[b], pb, ps, g[: 1 << klog], hp)
        st, mode, cnt = o["sponge_after"].raw()
        assert [O.le_to_int(rows["sponge_state"][b, 32 * i: 32 * i + 32]) for i in range(3)] == st and rows["sponge_pos"][b].tolist() == [mode, cnt], ("sponge", seed0, rnd)
        assert O.le_to_int(rows["ft_eval0"][b]) == o["ft_eval0"] and O.le_to_int(rows["cip"][b]) == o["combined_inner_product"], ("ft/cip", seed0, rnd)
        assert O.le_to_int(rows["polyscale"][b]) == o["v"] and O.le_to_int(rows["evalscale"][b]) == o["u"], ("v/u", seed0, rnd)
        assert [O.bytes_to_point(c) for c in rows["comms"][b]] == o["comms"], ("comms", seed0, rnd)
    counts["kimchi_rows"] += nb
    # ---- the matrix-core fold
    field = rng.randrange(2); kk = rng.choice([2, 3, 5, 8]); batch = rng.choice([256, 257, 300, 511, 1000])
    mod = R.P if field == 0 else R.Q
    chals = np.stack([O.int_to_le(rng.choice([0, 1, mod - 1, rng.randrange(mod), rng.randrange(mod)])) for _ in range(batch * kk)])
    wts = np.stack([O.int_to_le(rng.randrange(mod)) for _ in range(batch)])
    gotf = ctx.b_poly_fold(field, kk, chals, wts)
    acc = [0] * (1 << kk)
    for b in range(batch):
        s = O.b_poly_coefficients(field, chals[b * kk:(b + 1) * kk]); wb = O.le_to_int(wts[b])
        for j in range(1 << kk):
            acc[j] = (acc[j] + wb * int.from_bytes(s[j].tobytes(), "little")) % mod
    assert (gotf == O.ints_to_le(acc)).all(), ("fold", seed0, rnd, field, kk, batch)
    counts["fold_outputs"] += 1 << kk
ctx.close()
print(json.dumps({"seed": seed0, "rounds": rnd, **counts, "seconds": budget}))
