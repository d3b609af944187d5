"""Generates tests/golden/statement_k15.json: COMPLETE wrap proofs at the Pickles wrap size -- a Pickles statement, the wrap proof whose
public input is that statement's 40-element packing, and the step accumulator the statement carries -- for the verifier index of
tests/golden/kimchi_k15.json (same circuit seed) and the synthetic step index `kimchi_helpers.make_step_index(99)`.

Minted by the repo's OWN CPU oracle (oracle/pickles_ref.py derives the public input, oracle/kimchi_ref.py's miniature prover mints the
proof, oracle/state_job_ref.py the accumulator) under the Poseidon constant set named in the file: inputs + expected ACCEPT for the
full Proof-of-State job from raw statements at BASELINE size (bench.py, tests/test_state_job.py).  There is no reference
implementation to import and no real proof offline.
Run:  python tests/golden/gen_statement_fixture.py [count]   (~6 min per proof)
      python tests/golden/gen_statement_fixture.py count --start S --out FILE     proofs S .. S+count-1 (same seeds as a longer single run would use) into FILE:
      the parts of tests/golden/statement_k15_many.npz (tests/golden/encode_statement_fixture.py --many; tools/mint_many.sh runs the parts in parallel)"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.setrecursionlimit(10000)
from oracle import ipa_ref as I, kimchi_ref as K, oracle as O, pasta_ref as R, pickles_ref as PK, state_job_ref as J
from ipa_helpers import poseidon_pp
from kimchi_helpers import load_k15_fixture, make_chain, make_step_index
from wire_writers import synth_wrap_proof
import mina_bridge_amd.poseidon_params as PP

K_LOG2, NPUB, ACC_K = 15, 40, 16
count = int(sys.argv[1]) if len(sys.argv) > 1 else 4
start = int(sys.argv[sys.argv.index("--start") + 1]) if "--start" in sys.argv else 0
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "tests/golden/statement_k15.json")
R.inv = lambda a, m: pow(a, -1, m)                              # same value as the oracle's pow(a, m - 2, m), ~20x faster: the prover inverts 3 x 2^18 times per proof
nthreads = os.cpu_count() or 4
g, h = O.srs_create(0, 1 << K_LOG2, threads=nthreads)
gv, _ = O.srs_create(1, 1 << ACC_K, threads=nthreads)
hp = O.bytes_to_point(h)
pb, ps = poseidon_pp(0), poseidon_pp(1)
hx = lambda p: O.point_to_bytes(p).tobytes().hex()
t0 = time.time()
circ = K.synthetic_circuit(0, g, hp, pb, ps, K_LOG2, NPUB, seed=0xC15)
ix_fixture, _, fx = load_k15_fixture()
assert circ.index.digest == ix_fixture.digest and fx["poseidon_constants"] == PP.NAME, "the wrap index must be the one of kimchi_k15.json"
print("circuit", round(time.time() - t0, 1), "s", flush=True)
ix = circ.index
comms = list(ix.sigma_comm) + list(ix.coefficients_comm) + list(ix.selector_comm)
step = make_step_index(99)
out = {"poseidon_constants": PP.NAME, "wrap_index": "tests/golden/kimchi_k15.json", "step_index": "kimchi_helpers.make_step_index(99)", "proofs": []}
STATEMENT_KEYS = ("alpha", "beta", "gamma", "zeta", "joint_combiner", "feature_flags", "bulletproof_challenges", "proofs_verified", "domain_log2", "sponge_digest",
                  "challenge_polynomial_commitment", "old_bulletproof_challenges", "step_comms", "step_old_chals", "prev_public_input", "prev_evals", "prev_optional", "prev_ft_eval1")
for i in range(start, start + count):
    t0 = time.time()
    rng = random.Random(0x57A7 + i)
    wrap = synth_wrap_proof(rng, k=K_LOG2, lookups=False)
    pres = [[rng.getrandbits(128) for _ in range(15)] for _ in range(2)]
    chals = [[R.challenge_to_field(p, R.endo_r(0), R.Q) for p in row] for row in pres]
    wrap["prev_optional"] = [None] * 19
    wrap["old_bulletproof_challenges"] = pres
    prev_comms = []
    for ch in chals:                                            # the previous wrap accumulators: commitments of b_poly_coefficients(chals)
        sc = [O.le_to_int(x) for x in O.b_poly_coefficients(1, O.ints_to_le(ch))]
        prev_comms.append(I.commit(0, g[: 1 << K_LOG2], hp, sc, 0))
    wrap["step_comms"] = prev_comms
    pre, sg = J.make_accumulator(1, gv, ACC_K, 0xACC + i)
    wrap["bulletproof_challenges"] = [int.from_bytes(pre[j].tobytes(), "little") for j in range(16)]
    wrap["challenge_polynomial_commitment"] = O.bytes_to_point(sg)
    # the application state the statement binds: the hash of the candidate tip (state 15) of the deterministic chain `make_chain(Random(chain_seed))`
    chain_seed = 0xC4A1 + i
    app = make_chain(random.Random(chain_seed), pb)[1][15]
    pubs, dv, mw_, ms_ = PK.statement_public_input(wrap, step, comms, app, pb, ps)
    proof = K.synthetic_proof(circ, g, hp, pb, ps, pubs, seed=9000 + i, prev_chals=chals)
    assert [cm for _, cm in proof["prev"]] == prev_comms
    o, entry = K.oracles_and_batch(ix, proof, pubs, pb, ps, g, hp)
    assert I.ipa_verify_batch(0, g, hp, [entry], 7, 9), "minted proof does not verify"
    assert J.accumulator_ok(1, gv, ACC_K, pre, sg)
    op = proof["opening"]
    enc = lambda v: (None if v is None else [enc(x) for x in v] if isinstance(v, (list, tuple)) else bool(v) if isinstance(v, bool) else str(v))
    out["proofs"].append({"statement": {k: enc(wrap[k]) for k in STATEMENT_KEYS}, "app_state": str(app), "chain_seed": chain_seed, "pubs": [str(x) for x in pubs],
                          "acc_pre": pre.tobytes().hex(), "acc_sg": bytes(sg).hex(),
                          "w_comm": [hx(p) for p in proof["w_comm"]], "z_comm": hx(proof["z_comm"]), "t_comm": [hx(p) for p in proof["t_comm"]],
                          "evals": [[str(a), str(b)] for a, b in proof["evals"]], "ft_eval1": str(proof["ft_eval1"]),
                          "lr": [[hx(l), hx(r)] for l, r in op["lr"]], "delta": hx(op["delta"]), "sg": hx(op["sg"]), "z1": str(op["z1"]), "z2": str(op["z2"])})
    print("proof", i, "ok", round(time.time() - t0, 1), "s", flush=True)
    json.dump(out, open(out_path, "w"), indent=0)
