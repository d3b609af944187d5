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
    def burst_of_calls(calls):
        got = [None] * len(calls)
        def worker(i): got[i] = m.lib.verify_state(calls[i][0], calls[i][1])
        th = [threading.Thread(target=worker, args=(i,)) for i in range(len(calls))]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        return got, time.perf_counter() - t0
    for _ in range(2):                                              # the first search for culprits sets up its lanes' buffers
        got, mixed = burst_of_calls(calls)
        assert got == [c[2] for c in calls], "every caller gets the verdict of its own proof"
    # the timing claim is made on the burst WITHOUT tampered proofs (a merged job that fails goes through a culprit search of several jobs, whose
    # length depends on how the arrivals happened to be grouped): 24 good calls leave as one or two jobs
    good = [c for c in calls if c[2]]; good = (good * 2)[:24]
    burst_of_calls(good)
    got, burst = min((burst_of_calls(good) for _ in range(3)), key=lambda r: r[1])
    assert got == [True] * 24
    assert burst < 0.5 * len(good) * one, f"24 concurrent calls took {burst * 1e3:.1f} ms against {one * 1e3:.1f} ms for one: they were not merged"
    print(f"one {one * 1e3:.1f} ms; 24 concurrent good calls {burst * 1e3:.1f} ms; 24 with 8 tampered {mixed * 1e3:.1f} ms")


def test_c5_4096_full_size_proofs_over_two_contexts_equal_one_context_equal_the_oracle(big, srs_oracle):
    """BASELINE config C5 at its size through the boundary: 4096 serialized full-size state proofs in ONE `mina_verify_state_batch` call -- the four
    fixture proofs and three tampered variants (a flipped candidate-chain hash in the public input, a changed opening scalar, a changed step
    prechallenge), the tampered ones placed in both halves of the call.  The expected verdict of every position is the NATIVE CPU ORACLE's verdict
    (oracle/composite_oracle.c) on that distinct instance.  Run on one context, on two logical contexts (two contiguous shards of 2048, each with its
    own folding randomisers and culprit search: the multi-GPU path of SURVEY.md 8e.1, `$MINA_VERIFY_DEVICES=0,0`), and on two contexts with each
    shard cut into chunks of 512: the three verdict vectors are identical and equal to the oracle's."""
    import copy
    import json
    import os
    import mina_bridge_amd.poseidon_params as PP
    from kimchi_helpers import install_index, install_step_index, load_k15_fixture, make_step_index
    from oracle import composite as C, mina_state_ref as S
    from wire_writers import state_proof_bytes
    m, cases = big["m"], big["cases"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = json.load(open(os.path.join(root, "tests", "golden", "statement_k15_encoded.json")))
    assert fx["poseidon_constants"] == PP.NAME

    # ---- the seven distinct instances: bytes for the boundary, (encoded item, records, expected hashes) for the oracle
    def chain_records(c):
        recs = np.zeros((17, 64, 32), np.uint8); nf = np.zeros(17, np.uint32)
        for i, st in enumerate(c["states"]):
            f = [st["previous_state_hash"]] + S.body_to_input(st["body"]).to_fields()
            nf[i] = len(f) - 1
            recs[i, : len(f)] = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in f), np.uint8).reshape(len(f), 32)
        exp = np.frombuffer(b"".join(int(h).to_bytes(32, "little") for h in c["hashes"]), np.uint8).reshape(17, 32).copy()
        return recs.reshape(17, -1), nf, exp
    inst = []                                               # (proof bytes, pub bytes, oracle proof)
    for b, c in enumerate(cases):
        recs, nf, exp = chain_records(c)
        inst.append((c["proof"], c["pub"], C.make_proof(fx["proofs"][b], recs, nf, exp)))
    # (4) a flipped bit in candidate_chain_state_hashes[0] of the public input (byte 33 + 7)
    recs, nf, exp = chain_records(cases[0]); exp[0, 7] ^= 1
    bad_pub = bytearray(cases[0]["pub"]); bad_pub[40] ^= 1
    inst.append((cases[0]["proof"], bytes(bad_pub), C.make_proof(fx["proofs"][0], recs, nf, exp)))
    # (5) opening scalar z1 + 1
    w = copy.deepcopy(cases[2]["wrap"]); w["z1"] = (w["z1"] + 1) % (1 << 254)
    it = copy.deepcopy(fx["proofs"][2]); it["opening"]["z1"] = int(w["z1"]).to_bytes(32, "little").hex()
    inst.append((state_proof_bytes(w, cases[2]["states"]), cases[2]["pub"], C.make_proof(it, *chain_records(cases[2]))))
    # (6) step bulletproof prechallenge 5, low bit: the accumulator's challenges ARE the statement's
    w = copy.deepcopy(cases[3]["wrap"]); w["bulletproof_challenges"][5] ^= 1
    it = copy.deepcopy(fx["proofs"][3])
    bp = bytearray(bytes.fromhex(it["statement"]["bulletproof_challenges"])); bp[5 * 16] ^= 1
    it["statement"]["bulletproof_challenges"] = bytes(bp).hex(); it["acc_prechallenges"] = bytes(bp).hex()
    inst.append((state_proof_bytes(w, cases[3]["states"]), cases[3]["pub"], C.make_proof(it, *chain_records(cases[3]))))

    C.setup(srs_oracle[0], srs_oracle[1], PP.default_params_bytes(0), PP.default_params_bytes(1), fx["wrap_index"], fx["step_index"], threads=os.cpu_count() or 4)
    oracle_verdicts = C.verify_many([x[2] for x in inst], min(len(inst), os.cpu_count() or 4)).tolist()
    assert oracle_verdicts == [1, 1, 1, 1, 0, 0, 0], f"the CPU oracle's verdicts on the distinct instances: {oracle_verdicts}"

    # ---- 4096 positions: the good four tiled, tampered instances in both halves (and at both ends)
    N = 4096
    which = [i % 4 for i in range(N)]
    for pos, k in ((0, 4), (700, 5), (2047, 6), (2048, 5), (3000, 4), (4095, 6)):
        which[pos] = k
    proofs = [inst[k][0] for k in which]; pubs = [inst[k][1] for k in which]
    want = [oracle_verdicts[k] for k in which]
    ix, _, _ = load_k15_fixture()
    keep_dev = os.environ.get("MINA_VERIFY_DEVICES")
    try:
        one = m.lib.verify_state_batch(proofs, pubs).tolist()
        assert one == want, f"one context: {[i for i in range(N) if one[i] != want[i]][:8]}"
        m.lib.verify_shutdown()
        os.environ["MINA_VERIFY_DEVICES"] = "0,0"
        assert m.lib.verify_device_count() == 2
        alld = m.lib.verify_all_devices()
        install_index(alld, ix); install_step_index(alld, make_step_index(99))
        two = m.lib.verify_state_batch(proofs, pubs).tolist()
        assert two == want, f"two contexts: {[i for i in range(N) if two[i] != want[i]][:8]}"
        with m.lib.tuning(single_max=1, chunk=512):
            chunked = m.lib.verify_state_batch(proofs, pubs).tolist()
        assert chunked == want, f"two contexts, chunks of 512: {[i for i in range(N) if chunked[i] != want[i]][:8]}"
    finally:
        if keep_dev is None: os.environ.pop("MINA_VERIFY_DEVICES", None)
        else: os.environ["MINA_VERIFY_DEVICES"] = keep_dev
        m.lib.verify_shutdown()
        gctx = m.lib.verify_global_ctx()
        install_index(gctx, ix); install_step_index(gctx, make_step_index(99))


FORCED_SHAPES = {
    # lane forms of the sponge kernels (ctx.h use_coop*): every form must give the same bits
    "lanes16_everywhere": dict(coop16_max=1 << 30),
    "lanes8_everywhere": dict(coop16_max=0, coop8_max=1 << 30, transcript_coop8_max=1 << 30, ipa_coop8_max=1 << 30, kimchi_coop8_max=1 << 30),
    "lanes3_everywhere": dict(coop16_max=0, coop8_max=0, transcript_coop8_max=1),
    # every shortcut replaced by its slower equivalent
    "shortcuts_off": dict(bpoly_mfma=0, pubcomm_direct=0, ipa_shared_points=0, kimchi_shared_digest=0, ipa_side_stream=0, search_full=1, msm_fp29=0),
    # the bytes -> bools pipeline bent every way its knobs allow
    "pipeline_streamed_entry_by_entry": dict(early_min=1, early_sub=1, head_min=0, hash_piece_waves=1),
    "pipeline_no_forks_no_masks": dict(split_max=0, chain_cus=0, up_stream=0, slots=16, ahead=3, merge_batch_max=0, acc_mask=2),
    "pipeline_small_chunks": dict(single_max=1, chunk=7, window=2, merge=0, search_fan=2, early_min=1, early_sub=3),
}


@pytest.mark.parametrize("shape", sorted(FORCED_SHAPES))
def test_verdicts_do_not_depend_on_the_tuning(big, shape):
    """the full-size batch of test_full_size_batch_and_one_tamper_per_stage, at 300 proofs (the matrix-core fold needs >= 256), under every forced
    lane form / shortcut / pipeline shape of `mina_verify_tuning` -- run by the driver's GPU test tier, not by hand: identical verdicts"""
    import copy
    from wire_writers import state_proof_bytes
    m, cases = big["m"], big["cases"]
    bad_pub = bytearray(cases[0]["pub"]); bad_pub[40] ^= 1
    w2 = copy.deepcopy(cases[2]["wrap"]); w2["z1"] = (w2["z1"] + 1) % (1 << 254)
    w3 = copy.deepcopy(cases[3]["wrap"]); w3["bulletproof_challenges"][5] ^= 1
    bads = {17: (cases[0]["proof"], bytes(bad_pub)), 150: (state_proof_bytes(w2, cases[2]["states"]), cases[2]["pub"]), 299: (state_proof_bytes(w3, cases[3]["states"]), cases[3]["pub"])}
    proofs = [bads[i][0] if i in bads else cases[i % 4]["proof"] for i in range(300)]
    pubs = [bads[i][1] if i in bads else cases[i % 4]["pub"] for i in range(300)]
    want = [0 if i in bads else 1 for i in range(300)]
    with m.lib.tuning(**FORCED_SHAPES[shape]):
        got = m.lib.verify_state_batch(proofs, pubs).tolist()
        assert got == want, (shape, [i for i in range(300) if got[i] != want[i]][:8])
        assert m.lib.verify_state(cases[1]["proof"], cases[1]["pub"]) is True and m.lib.verify_state(*bads[150]) is False
