"""The reference-shaped boundary at the REAL sizes: `mina_verify_state(proof bytes, public-input bytes) -> bool` (Aligned's
`verify_mina_state_ffi` shape, README.md:275-310) on a bincode `MinaStateProof` whose wrap proof lives on the 2^15 domain with 40 public
inputs, 47 commitments, a 15-round opening and the 2^16 step accumulator -- the committed fixture tests/golden/statement_k15.json, whose
statement binds the hash of the candidate tip of a deterministic chain.  Every byte goes through the product's own parsers; the chain,
the statement, the kimchi step, the opening and the accumulator all run on the GPU.  Synthetic verifier / step index (the real ones are
not offline); Poseidon constants as named in the fixture."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ALL = 1 | 2 | 4 | 8 | 16 | 32


@pytest.fixture(scope="module")
def big(oracle):
    import mina_bridge_amd as m
    from ipa_helpers import poseidon_pp
    from kimchi_helpers import install_index, install_step_index, load_k15_fixture, load_statement_fixture, make_chain, make_step_index
    from oracle import mina_state_ref as S
    from wire_writers import state_proof_bytes, state_pub_bytes
    ix, _, _ = load_k15_fixture()
    items, fx = load_statement_fixture()
    m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)      # the tests run on the surrogate Poseidon tables (the real ones are not offline)
    gctx = m.lib.verify_global_ctx()
    install_index(gctx, ix)
    install_step_index(gctx, make_step_index(99))
    cases = []
    for it in items:
        states, hashes = make_chain(random.Random(it["chain_seed"]), poseidon_pp(0))
        assert hashes[15] == it["app"], "the fixture's application state is the hash of its chain's candidate tip"
        p, ev = it["proof"], it["proof"]["evals"]
        wrap = dict(it["wrap"])
        wrap.update(w_comm=p["w_comm"], z_comm=p["z_comm"], t_comm=p["t_comm"], z_eval=ev[0], selector_eval=ev[1:7], w_eval=ev[7:22], coefficients_eval=ev[22:37],
                    s_eval=ev[37:43], ft_eval1=p["ft_eval1"], lr=p["opening"]["lr"], z1=p["opening"]["z1"], z2=p["opening"]["z2"], delta=p["opening"]["delta"],
                    sg=p["opening"]["sg"])
        ledger = [S.snarked_ledger_hash(s) for s in states[:16]]
        cases.append({"wrap": wrap, "states": states, "hashes": hashes, "proof": state_proof_bytes(wrap, states), "pub": state_pub_bytes(True, hashes[16], hashes[:16], ledger)})
    yield {"m": m, "cases": cases}
    m.lib.verify_configure(0)


def test_full_size_state_proofs_accepted_through_the_boundary(big):
    m = big["m"]
    for c in big["cases"][:2]:
        assert m.lib.verify_state(c["proof"], c["pub"]) is True
        assert m.lib.verify_state_checks(c["proof"], c["pub"]) == (ALL, ALL)


def test_full_size_batch_and_one_tamper_per_stage(big):
    """mina_verify_state_batch over the four proofs + tampered copies: a flipped public hash (CHAIN), a changed statement field (KIMCHI
    through the public input), a changed opening scalar (KIMCHI / opening), a changed step prechallenge (ACCUMULATOR and the statement)"""
    import copy
    from wire_writers import state_proof_bytes
    m, cases = big["m"], big["cases"]
    proofs = [c["proof"] for c in cases]; pubs = [c["pub"] for c in cases]
    bad_pub = bytearray(cases[0]["pub"]); bad_pub[40] ^= 1
    w1 = copy.deepcopy(cases[1]["wrap"]); w1["feature_flags"][0] = not w1["feature_flags"][0]
    w2 = copy.deepcopy(cases[2]["wrap"]); w2["z1"] = (w2["z1"] + 1) % (1 << 254)
    w3 = copy.deepcopy(cases[3]["wrap"]); w3["bulletproof_challenges"][5] ^= 1
    proofs += [cases[0]["proof"], state_proof_bytes(w1, cases[1]["states"]), state_proof_bytes(w2, cases[2]["states"]), state_proof_bytes(w3, cases[3]["states"])]
    pubs += [bytes(bad_pub), cases[1]["pub"], cases[2]["pub"], cases[3]["pub"]]
    assert m.lib.verify_state_batch(proofs, pubs).tolist() == [1, 1, 1, 1, 0, 0, 0, 0]
    # the culprits of the failed folded opening check were found from the prepared rows of the batch (slices of them re-checked); the same
    # verdicts when every part repeats its transcripts instead
    with m.lib.tuning(search_full=1):
        assert m.lib.verify_state_batch(proofs, pubs).tolist() == [1, 1, 1, 1, 0, 0, 0, 0]
    # a bigger batch with culprits at both ends and in the middle: two rounds of cuts
    many_p = [proofs[i % 4] for i in range(70)]; many_q = [pubs[i % 4] for i in range(70)]
    for i in (0, 33, 34, 69): many_p[i] = proofs[6]; many_q[i] = pubs[6]
    many_p[50] = proofs[7]; many_q[50] = pubs[7]
    want = [0 if i in (0, 33, 34, 50, 69) else 1 for i in range(70)]
    assert m.lib.verify_state_batch(many_p, many_q).tolist() == want
    passed, ran = m.lib.verify_state_checks(proofs[5], pubs[5])
    assert ran == ALL and passed == ALL & ~32, "a changed statement field fails exactly the kimchi step"
    passed, ran = m.lib.verify_state_checks(proofs[7], pubs[7])
    assert ran == ALL and not passed & 16 and not passed & 32, "a changed step prechallenge fails the accumulator check and, through the statement, the kimchi step"


def test_concurrent_single_proof_calls_are_merged_into_shared_jobs(big):
    """The reference's entry point takes ONE proof and is called from many threads at once.  24 threads call
    mina_verify_state concurrently (good proofs and the tampered ones of the test above, interleaved): every caller gets its own
    verdict, and because calls that arrive while a job runs leave together as the next job, the whole burst takes far less than 24
    single-proof latencies."""
    import copy
    import threading
    import time
    from wire_writers import state_proof_bytes
    m, cases = big["m"], big["cases"]
    w_bad = copy.deepcopy(cases[2]["wrap"]); w_bad["z1"] = (w_bad["z1"] + 1) % (1 << 254)
    bad_pub = bytearray(cases[0]["pub"]); bad_pub[40] ^= 1
    calls = []
    for i in range(24):
        c = cases[i % 4]
        if i % 6 == 4: calls.append((state_proof_bytes(w_bad, cases[2]["states"]), cases[2]["pub"], False))
        elif i % 6 == 5: calls.append((cases[0]["proof"], bytes(bad_pub), False))
        else: calls.append((c["proof"], c["pub"], True))
    assert m.lib.verify_state(*calls[0][:2]) is True               # warm: context, tables, buffers
    t0 = time.perf_counter()
    for p, q, want in calls[:4]: assert m.lib.verify_state(p, q) is want
    one = (time.perf_counter() - t0) / 4
    def burst_of_calls():
        got = [None] * len(calls)
        def worker(i): got[i] = m.lib.verify_state(calls[i][0], calls[i][1])
        th = [threading.Thread(target=worker, args=(i,)) for i in range(len(calls))]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        return got, time.perf_counter() - t0
    got, _ = burst_of_calls()                                       # the first search for culprits sets up its lanes' buffers
    assert got == [c[2] for c in calls]
    got, burst = burst_of_calls()
    assert got == [c[2] for c in calls]
    assert burst < 0.5 * len(calls) * one, f"24 concurrent calls took {burst * 1e3:.1f} ms against {one * 1e3:.1f} ms for one: they were not merged"
