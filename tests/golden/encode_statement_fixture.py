"""tests/golden/statement_k15.json (+ the wrap index of kimchi_k15.json, + kimchi_helpers.make_step_index(99)) -> tests/golden/
statement_k15_encoded.json: the SAME inputs in the byte layouts of the C-ABI (include/mina_verify.h: mina_verifier_index, mina_step_index,
mina_pickles_statements, mina_kimchi_proofs, the opening and accumulator sections of mina_state_jobs), hex per section.
bench.py's default mode reads only this file, so that nothing under oracle/ is imported outside its cpu_baseline leg; the test
tests/test_statement_fixture.py checks that it is what the helpers produce.  Run after gen_statement_fixture.py:
    python tests/golden/encode_statement_fixture.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from kimchi_helpers import STEP_DOMAINS, encode_tokens, kimchi_arrays, load_k15_fixture, load_statement_fixture, make_step_index, pts, statements_soa
from oracle import oracle as O


def encode(items=None, fx=None):
    ix, _, fxk = load_k15_fixture()
    if items is None:
        items, fx = load_statement_fixture()
    step = make_step_index(99)
    hx = lambda a: np.ascontiguousarray(a, dtype=np.uint8).reshape(-1).tobytes().hex()
    out = {"poseidon_constants": fx["poseidon_constants"], "source": "tests/golden/statement_k15.json, tests/golden/kimchi_k15.json, kimchi_helpers.make_step_index(99)",
           "wrap_index": {"log2_domain": ix.log2_domain, "zk_rows": ix.zk_rows, "perm_alpha_offset": ix.perm_alpha_offset, "shifts": hx(O.ints_to_le(ix.shifts)),
                          "sigma_comm": hx(pts(ix.sigma_comm)), "coefficients_comm": hx(pts(ix.coefficients_comm)), "selector_comm": hx(pts(ix.selector_comm)),
                          "constant_term": encode_tokens(ix.constant_term).hex()},
           "step_index": {"zk_rows": step.zk_rows, "domains": STEP_DOMAINS, "shifts": hx(np.concatenate([O.ints_to_le(step.shifts[k]).reshape(-1) for k in STEP_DOMAINS])),
                          "constant_term": encode_tokens(step.constant_term).hex()},
           "proofs": []}
    for it in items:
        n_old, n_evals, sec = statements_soa([it["wrap"]], [it["app"]])
        arrays, op = kimchi_arrays([it["proof"]], [])
        arrays.pop("prev_chals")                                       # the 128-bit prechallenges travel instead (expanded on the GPU)
        pre = b"".join(int(c).to_bytes(16, "little") for row in it["wrap"]["old_bulletproof_challenges"] for c in row)
        out["proofs"].append({"n_old": n_old, "n_evals": n_evals, "statement": {k: hx(v) for k, v in sec.items()},
                              "kimchi": dict({k: hx(v) for k, v in arrays.items() if v is not None}, prev_prechallenges=pre.hex()),
                              "opening": {k: hx(v) for k, v in op.items()}, "acc_prechallenges": hx(it["acc_pre"]), "acc_sg": hx(it["acc_sg"]),
                              "public_inputs": hx(O.ints_to_le(it["pubs"]))})
    return out


def encode_boundary_bytes():
    """the same four proofs as the reference's caller would send them (core/src/aligned.rs:31-58): bincode `MinaStateProof` (wrap proof + the 16
    + 1 protocol states of the deterministic chain whose candidate tip the statement binds) and the 1057-byte `MinaStatePubInputs`, hex.
    bench.py's `boundary_bytes_to_bools` leg reads this file (nothing under oracle/ outside its cpu_baseline leg)."""
    import random
    from ipa_helpers import poseidon_pp
    from kimchi_helpers import make_chain
    from oracle import mina_state_ref as S
    from wire_writers import state_proof_bytes, state_pub_bytes
    items, fx = load_statement_fixture()
    out = {"poseidon_constants": fx["poseidon_constants"], "source": "tests/golden/statement_k15.json through tests/wire_writers.py", "proofs": []}
    for it in items:
        states, hashes = make_chain(random.Random(it["chain_seed"]), poseidon_pp(0))
        assert hashes[15] == it["app"]
        p, ev = it["proof"], it["proof"]["evals"]
        wrap = dict(it["wrap"])
        wrap.update(w_comm=p["w_comm"], z_comm=p["z_comm"], t_comm=p["t_comm"], z_eval=ev[0], selector_eval=ev[1:7], w_eval=ev[7:22], coefficients_eval=ev[22:37], s_eval=ev[37:43],
                    ft_eval1=p["ft_eval1"], lr=p["opening"]["lr"], z1=p["opening"]["z1"], z2=p["opening"]["z2"], delta=p["opening"]["delta"], sg=p["opening"]["sg"])
        out["proofs"].append({"proof": state_proof_bytes(wrap, states).hex(),
                              "pub": state_pub_bytes(True, hashes[16], hashes[:16], [S.snarked_ledger_hash(s) for s in states[:16]]).hex()})
    return out


def encode_many():
    """.mint_parts/part_*.json (tools/mint_many.sh: proofs 4, 5, ... of the same generator, same indexes) -> tests/golden/statement_k15_many.npz: the sections of
    `encode()` as uint8 arrays [n_proofs, bytes_per_proof], one array per section ("statement.<name>", "kimchi.<name>", "opening.<name>", "acc_prechallenges",
    "acc_sg", "public_inputs"), + "chain_seed".  bench.py tiles its headline batch from the 4 proofs of statement_k15_encoded.json + these (config.distinct_inputs)."""
    import glob
    parts = sorted(glob.glob(os.path.join(ROOT, ".mint_parts", "part_*.json")), key=lambda p: int(p.rsplit("_", 1)[1].split(".")[0]))
    cols, seeds, consts = {}, [], None
    for path in parts:
        items, fx = load_statement_fixture(path)
        consts = consts or fx["poseidon_constants"]
        assert fx["poseidon_constants"] == consts
        for rec, it in zip(encode(items, fx)["proofs"], items):
            flat = {"acc_prechallenges": rec["acc_prechallenges"], "acc_sg": rec["acc_sg"], "public_inputs": rec["public_inputs"]}
            for grp in ("statement", "kimchi", "opening"):
                flat.update({grp + "." + k: v for k, v in rec[grp].items()})
            for k, v in flat.items():
                cols.setdefault(k, []).append(np.frombuffer(bytes.fromhex(v), np.uint8))
            seeds.append(it["chain_seed"])
    n = len(seeds)
    assert n and all(len(v) == n for v in cols.values())
    arrays = {k: np.stack(v) for k, v in cols.items()}
    np.savez(os.path.join(ROOT, "tests/golden/statement_k15_many.npz"), chain_seed=np.array(seeds, np.int64), poseidon_constants=np.frombuffer(consts.encode(), np.uint8), **arrays)
    return n


if __name__ == "__main__":
    if "--many" in sys.argv:
        print("wrote tests/golden/statement_k15_many.npz:", encode_many(), "proofs")
        sys.exit(0)
    json.dump(encode(), open(os.path.join(ROOT, "tests/golden/statement_k15_encoded.json"), "w"), indent=0)
    print("wrote tests/golden/statement_k15_encoded.json")
    json.dump(encode_boundary_bytes(), open(os.path.join(ROOT, "tests/golden/state_proofs_k15_bytes.json"), "w"), indent=0)
    print("wrote tests/golden/state_proofs_k15_bytes.json")
