"""oracle/composite_oracle.c (the native CPU composite `cpu_baseline` times) against the Python composite it restates, value by value, on
the committed full-size fixture: the statement's 40 public inputs, kimchi's ft_eval0 / v / u / combined inner product, the 17 state hashes,
ACCEPT -- and REJECT for a tamper of every leg."""
import copy
import json
import os
import random

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native(srs_oracle):
    import mina_bridge_amd.poseidon_params as PP
    from ipa_helpers import poseidon_pp
    from kimchi_helpers import load_statement_fixture, make_chain
    from oracle import composite as C, mina_state_ref as S
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "statement_k15_encoded.json")))
    assert fx["poseidon_constants"] == PP.NAME
    C.setup(srs_oracle[0], srs_oracle[1], PP.default_params_bytes(0), PP.default_params_bytes(1), fx["wrap_index"], fx["step_index"], threads=os.cpu_count() or 4)
    items, _ = load_statement_fixture()
    chains = []
    for it in items:
        states, hashes = make_chain(random.Random(it["chain_seed"]), poseidon_pp(0))
        recs = np.zeros((17, 64, 32), np.uint8); nf = np.zeros(17, np.uint32)
        for i, st in enumerate(states):
            f = [st["previous_state_hash"]] + S.body_to_input(st["body"]).to_fields()
            nf[i] = len(f) - 1
            recs[i, : len(f)] = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in f), np.uint8).reshape(len(f), 32)
        exp = np.frombuffer(b"".join(int(h).to_bytes(32, "little") for h in hashes), np.uint8).reshape(17, 32).copy()
        chains.append((recs.reshape(17, -1), nf, exp, hashes))
    return {"C": C, "fx": fx, "items": items, "chains": chains}


def _le(b):
    return int.from_bytes(bytes(b), "little")


def test_native_composite_equals_the_python_composite(native, srs_oracle, oracle):
    from ipa_helpers import poseidon_pp
    from kimchi_helpers import load_k15_fixture, make_step_index
    from oracle import kimchi_ref as K, pickles_ref as PK
    C, fx, items, chains = native["C"], native["fx"], native["items"], native["chains"]
    ix, _, _ = load_k15_fixture()
    comms = list(ix.sigma_comm) + list(ix.coefficients_comm) + list(ix.selector_comm)
    g, h = srs_oracle[0]
    for b in (0, 3):
        recs, nf, exp, hashes = chains[b]
        r = C.verify_one(C.make_proof(fx["proofs"][b], recs, nf, exp))
        assert [_le(r.hashes[32 * i: 32 * i + 32]) for i in range(17)] == hashes
        pubs, dv, _, _ = PK.statement_public_input(items[b]["wrap"], make_step_index(99), comms, items[b]["app"], poseidon_pp(0), poseidon_pp(1))
        assert _le(r.step_cip) == dv["combined_inner_product"] and _le(r.step_b) == dv["b"]
        assert [_le(r.pubs[32 * i: 32 * i + 32]) for i in range(40)] == pubs == items[b]["pubs"]
        o, _ = K.oracles_and_batch(ix, items[b]["proof"], pubs, poseidon_pp(0), poseidon_pp(1), g[: 1 << 15], oracle.bytes_to_point(h))
        assert oracle.bytes_to_point(np.frombuffer(bytes(r.public_comm), np.uint8)) == o["public_comm"]
        assert _le(r.v) == o["v"] and _le(r.u) == o["u"]
        assert _le(r.ft_eval0) == o["ft_eval0"] and _le(r.cip) == o["combined_inner_product"]
        assert (r.chain_ok, r.statement_ok, r.ipa_ok, r.acc_ok, r.verdict) == (1, 1, 1, 1, 1)


def test_native_composite_rejects_a_tamper_of_every_leg(native):
    C, fx, chains = native["C"], native["fx"], native["chains"]
    recs, nf, exp, _ = chains[1]
    good = fx["proofs"][1]

    def flip(hexstr, byte, bit=0):
        b = bytearray(bytes.fromhex(hexstr)); b[byte] ^= 1 << bit; return bytes(b).hex()
    cases = {}
    it = copy.deepcopy(good); it["statement"]["app_state"] = flip(it["statement"]["app_state"], 3); cases["application state"] = (it, recs, exp, "ipa_ok")
    it = copy.deepcopy(good); it["statement"]["misc"] = flip(it["statement"]["misc"], 4); cases["feature flag"] = (it, recs, exp, "ipa_ok")
    it = copy.deepcopy(good); it["opening"]["z1"] = flip(it["opening"]["z1"], 0); cases["opening scalar"] = (it, recs, exp, "ipa_ok")
    it = copy.deepcopy(good); it["kimchi"]["evals"] = flip(it["kimchi"]["evals"], 40 * 64); cases["evaluation"] = (it, recs, exp, "ipa_ok")
    it = copy.deepcopy(good); it["acc_prechallenges"] = flip(it["acc_prechallenges"], 17); cases["accumulator"] = (it, recs, exp, "acc_ok")
    e2 = exp.copy(); e2[4, 0] ^= 1; cases["public hash"] = (good, recs, e2, "chain_ok")
    r2 = recs.copy(); r2[7, 5 * 32] ^= 1; cases["state body"] = (good, r2, exp, "chain_ok")
    for name, (item, rr, ee, leg) in cases.items():
        r = C.verify_one(C.make_proof(item, rr, nf, ee))
        assert r.verdict == 0 and getattr(r, leg) == 0, name
    # threaded across proofs: verdict bytes per proof
    proofs = [C.make_proof(fx["proofs"][i % 4], *chains[i % 4][:3]) for i in range(6)] + [C.make_proof(cases["opening scalar"][0], recs, nf, exp)]
    assert C.verify_many(proofs, threads=4).tolist() == [1] * 6 + [0]


def test_folded_batch_equals_the_per_proof_verdicts(native):
    """oc_verify_folded (bench.py's cpu_baseline_folded: per-proof transcripts, then ONE MSM per curve over the batch, as kimchi batch_verify and the GPU
    job fold it) against the per-proof composite: an accepting batch is accepted under several randomisers; a batch with ONE bad opening, ONE bad
    accumulator or ONE bad public hash is rejected (the fold answers for the batch; the per-proof legs keep their own bits); equal-and-opposite
    errors in two proofs -- z2 + t in one, z2 - t in the other -- cancel under rho = sigma = 1 and are caught under random ones."""
    import time
    C, fx, chains = native["C"], native["fx"], native["chains"]
    good = [C.make_proof(fx["proofs"][i % 4], *chains[i % 4][:3]) for i in range(9)]
    for rand in (None, bytes(range(96)), b"\x01" + bytes(31) + b"\x01" + bytes(31) + b"\x01" + bytes(31)):       # the last: every randomiser = 1
        ok, v = C.verify_folded(good, 4, rand)
        assert ok and v.tolist() == [1] * 9
    assert C.verify_many(good, 4).tolist() == [1] * 9

    def flip(hexstr, byte, bit=0):
        b = bytearray(bytes.fromhex(hexstr)); b[byte] ^= 1 << bit; return bytes(b).hex()
    recs, nf, exp, _ = chains[1]
    it = copy.deepcopy(fx["proofs"][1]); it["opening"]["z1"] = flip(it["opening"]["z1"], 0)
    ok, v = C.verify_folded(good[:5] + [C.make_proof(it, recs, nf, exp)] + good[5:], 3)
    assert not ok and v.tolist() == [0] * 10
    it = copy.deepcopy(fx["proofs"][1]); it["acc_prechallenges"] = flip(it["acc_prechallenges"], 17)
    ok, v = C.verify_folded([C.make_proof(it, recs, nf, exp)] + good, 3)
    assert not ok and not v.any()
    e2 = exp.copy(); e2[4, 0] ^= 1
    ok, v = C.verify_folded(good[:2] + [C.make_proof(fx["proofs"][1], recs, nf, e2)] + good[2:], 2)
    assert ok and v.tolist() == [1, 1, 0] + [1] * 7, "a wrong public hash fails its own proof only: the folds do not depend on it"
    # cancelling pair on z2 (the h term: -rho_b z2_b): proofs a, b with z2_a + t, z2_b - t
    Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
    a, b = copy.deepcopy(fx["proofs"][0]), copy.deepcopy(fx["proofs"][2])
    za, zb = int.from_bytes(bytes.fromhex(a["opening"]["z2"]), "little"), int.from_bytes(bytes.fromhex(b["opening"]["z2"]), "little")
    a["opening"]["z2"] = ((za + 12345) % Q).to_bytes(32, "little").hex(); b["opening"]["z2"] = ((zb - 12345) % Q).to_bytes(32, "little").hex()
    pair = [C.make_proof(a, *chains[0][:3]), C.make_proof(b, *chains[2][:3])]
    assert C.verify_many(pair, 2).tolist() == [0, 0], "each is invalid on its own"
    ok1, _ = C.verify_folded(pair, 2, b"\x01" + bytes(31) + b"\x01" + bytes(31) + b"\x01" + bytes(31))
    assert ok1, "with every randomiser = 1 the two errors cancel: why the randomisers must be unpredictable"
    ok2, v2 = C.verify_folded(pair, 2)
    assert not ok2 and not v2.any()
    t0 = time.perf_counter(); C.verify_folded(good * 4, 4); dt = time.perf_counter() - t0
    print(f"folded: {len(good) * 4 / dt:.1f} proofs/s on 4 threads")
