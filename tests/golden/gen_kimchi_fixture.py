"""Generates tests/golden/kimchi_k15.json: a synthetic verifier index at the Pickles wrap size (domain 2^15 = SRS chunk, 40 public
inputs, 2 recursion challenges) and ACCEPTING kimchi-shaped proofs for it, minted by the repo's OWN CPU oracle
(oracle/kimchi_ref.py: miniature prover) under the Poseidon constant set named in the file.  There is no reference implementation to
import and no real blockchain-snark index offline; these are inputs + expected verdict for the GPU kimchi step at
BASELINE size (bench.py --kimchi, tests/test_kimchi.py).  Run:  python tests/golden/gen_kimchi_fixture.py [count]   (~5 min per proof)"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.setrecursionlimit(10000)
from oracle import ipa_ref as I, kimchi_ref as K, oracle as O, pasta_ref as R
from ipa_helpers import poseidon_pp
import mina_bridge_amd.poseidon_params as PP

K_LOG2, NPUB = 15, 40
count = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g, h = O.srs_create(0, 1 << K_LOG2, threads=os.cpu_count() or 4)
hp = O.bytes_to_point(h)
pb, ps = poseidon_pp(0), poseidon_pp(1)
hx = lambda p: O.point_to_bytes(p).tobytes().hex()
t0 = time.time()
circ = K.synthetic_circuit(0, g, hp, pb, ps, K_LOG2, NPUB, seed=0xC15)
print("circuit", round(time.time() - t0, 1), "s", flush=True)
ix = circ.index
out = {"poseidon_constants": PP.NAME, "log2_domain": K_LOG2, "zk_rows": ix.zk_rows, "perm_alpha_offset": ix.perm_alpha_offset, "npub": NPUB, "n_prev": 2,
       "shifts": [str(x) for x in ix.shifts], "sigma_comm": [hx(p) for p in ix.sigma_comm], "coefficients_comm": [hx(p) for p in ix.coefficients_comm],
       "selector_comm": [hx(p) for p in ix.selector_comm], "constant_term": [[int(x) for x in tok] for tok in ix.constant_term], "digest": str(ix.digest), "proofs": []}
for i in range(count):
    t0 = time.time()
    rng = random.Random(0xA11 + i)
    pubs = [rng.randrange(R.Q) for _ in range(NPUB)]
    proof = K.synthetic_proof(circ, g, hp, pb, ps, pubs, seed=7000 + i)
    o, entry = K.oracles_and_batch(ix, proof, pubs, pb, ps, g, hp)
    assert I.ipa_verify_batch(0, g, hp, [entry], 7, 9), "minted proof does not verify"
    op = proof["opening"]
    out["proofs"].append({"pubs": [str(x) for x in pubs], "prev": [[[str(c) for c in ch], hx(cm)] for ch, cm in proof["prev"]],
                          "w_comm": [hx(p) for p in proof["w_comm"]], "z_comm": hx(proof["z_comm"]), "t_comm": [hx(p) for p in proof["t_comm"]],
                          "evals": [[str(a), str(b)] for a, b in proof["evals"]], "ft_eval1": str(proof["ft_eval1"]),
                          "lr": [[hx(l), hx(r)] for l, r in op["lr"]], "delta": hx(op["delta"]), "sg": hx(op["sg"]), "z1": str(op["z1"]), "z2": str(op["z2"]),
                          "expect": {"ft_eval0": str(o["ft_eval0"]), "cip": str(o["combined_inner_product"]), "v": str(o["v"]), "u": str(o["u"]), "ft_comm": hx(o["ft_comm"])}})
    print("proof", i, "ok", round(time.time() - t0, 1), "s", flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests/golden/kimchi_k15.json"), "w"), indent=0)
