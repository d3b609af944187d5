"""CPU-side pin of tests/golden/statement_k15.json (the full-size input of bench.py's default mode) against oracle/pickles_ref.py"""


def test_statement_fixture_is_what_the_oracle_derives():
    """CPU: the committed statements pack (oracle/pickles_ref.py) to the committed public inputs, and the fixture names the constant set in use"""
    from ipa_helpers import poseidon_pp
    from kimchi_helpers import load_k15_fixture, load_statement_fixture, make_step_index
    from oracle import pickles_ref as PK
    import mina_bridge_amd.poseidon_params as PP
    ix, _, _ = load_k15_fixture()
    items, fx = load_statement_fixture()
    assert fx["poseidon_constants"] == PP.NAME and len(items) >= 4
    comms = list(ix.sigma_comm) + list(ix.coefficients_comm) + list(ix.selector_comm)
    step = make_step_index(99)
    for it in items[:2]:
        assert PK.statement_public_input(it["wrap"], step, comms, it["app"], poseidon_pp(0), poseidon_pp(1))[0] == it["pubs"]


def test_encoded_fixture_is_the_helpers_encoding_of_the_fixture():
    """tests/golden/statement_k15_encoded.json (what bench.py's default mode reads, so that it needs nothing under oracle/) is exactly the
    C-ABI encoding the test helpers produce from statement_k15.json / kimchi_k15.json / make_step_index(99)"""
    import importlib.util, json, os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("encode_statement_fixture", os.path.join(here, "golden", "encode_statement_fixture.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.encode() == json.load(open(os.path.join(here, "golden", "statement_k15_encoded.json")))


def test_the_many_proof_fixture_holds_distinct_proofs_the_cpu_oracle_accepts():
    """tests/golden/statement_k15_many.npz (round 5: 252 more complete wrap proofs minted by the same generator under the same indexes -- bench.py's headline batch is
    tiled from 4 + 252 distinct proofs): every proof is distinct, and a sample of them passes the native CPU composite of the whole Proof-of-State job
    (oracle/composite_oracle.c: statement -> public inputs -> commitment -> kimchi oracles -> k = 15 opening check -> 2^16 accumulator), a tampered one does not.
    The GPU side: tests/test_state_job.py::test_many_distinct_wrap_proofs_in_one_job."""
    import json, os
    import numpy as np
    import mina_bridge_amd.poseidon_params as PP
    from oracle import composite as C, oracle as O
    here = os.path.dirname(os.path.abspath(__file__))
    z = np.load(os.path.join(here, "golden", "statement_k15_many.npz"))
    fx = json.load(open(os.path.join(here, "golden", "statement_k15_encoded.json")))
    assert bytes(z["poseidon_constants"]).decode() == PP.NAME == fx["poseidon_constants"]
    n = len(z["chain_seed"])
    assert n >= 252
    for key in ("kimchi.w_comm", "opening.z1", "statement.sponge_digest", "acc_sg"):
        assert len({bytes(r) for r in z[key]} | {bytes.fromhex(p[key.split(".")[0]][key.split(".")[1]]) if "." in key else bytes.fromhex(p[key]) for p in fx["proofs"]}) == n + 4, key
    srs = {c: O.srs_create(c, 1 << 16, threads=4) for c in (0, 1)}
    C.setup(srs[0], srs[1], PP.default_params_bytes(0), PP.default_params_bytes(1), fx["wrap_index"], fx["step_index"], threads=4)

    def item(i):
        it = {"n_old": fx["proofs"][0]["n_old"], "n_evals": fx["proofs"][0]["n_evals"], "statement": {}, "kimchi": {}, "opening": {},
              "acc_prechallenges": z["acc_prechallenges"][i].tobytes().hex(), "acc_sg": z["acc_sg"][i].tobytes().hex()}
        for k in z.files:
            if "." in k:
                grp, name = k.split(".")
                it[grp][name] = z[k][i].tobytes().hex()
        return it
    recs, nf, exp = np.zeros((17, 2048), np.uint8), np.zeros(17, np.uint32), np.zeros((17, 32), np.uint8)      # no state leg here: the wrap-proof and accumulator legs of the job
    picks = [0, 1, 57, 130, n - 1]
    proofs = [C.make_proof(item(i), recs, nf, exp) for i in picks]
    res = [C.verify_one(p) for p in proofs]
    assert all(r.ipa_ok and r.acc_ok and r.statement_ok for r in res), [(r.ipa_ok, r.acc_ok, r.statement_ok) for r in res]
    bad = item(57); b = bytearray(bytes.fromhex(bad["opening"]["z1"])); b[0] ^= 1; bad["opening"]["z1"] = bytes(b).hex()
    assert not C.verify_one(C.make_proof(bad, recs, nf, exp)).ipa_ok
