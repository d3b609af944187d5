"""Soundness of the FOLDED checks (SURVEY.md 8a a8 / section 7 "Verifier randomness"; kimchi `batch_verify` / poly-commitment `SRS::verify`
fold the proofs of a call with powers of two random field elements drawn from an RNG per call).

A fold  sum_b rho_b * (check_b)  only proves every check_b when the rho_b are unknown to whoever chose the proofs.  With rho known, two
invalid proofs can be built whose errors cancel:
  * opening leg: the batch-shared blinder `h` carries  -sum_b rho_b * z2_b  and nothing in the transcript binds z2, so
    z2_0' = z2_0 + rho_1 * t,  z2_1' = z2_1 - rho_0 * t   leaves the sum unchanged;
  * accumulator leg:  sg_0' = sg_0 + T,  sg_1' = sg_1 - (rho_0 / rho_1) * T.
The kernel-level entry points take the randomisers as ARGUMENTS (the header says what a caller who passes predictable ones gets): the first
two tests show that such a pair IS accepted there when built for the values passed, and rejected for any other.  The reference-shaped
boundary (`mina_verify_state_batch`, merged `mina_verify_state` calls) draws them from the operating system's CSPRNG per job, so no pair
built in advance survives -- the remaining tests: both proofs `false`, exactly what the reference (one proof per call) answers."""
import copy
import random
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _le(x):
    return np.frombuffer(int(x).to_bytes(32, "little"), np.uint8).copy()


# ------------------------------------------------------------------------------------------------ kernel level: explicit randomisers
def test_opening_fold_cancelling_z2_pair_needs_the_randomisers(ctx_srs, oracle):
    import mina_bridge_amd as m
    from kimchi_helpers import install_index, kimchi_arrays, load_k15_fixture
    from oracle import pasta_ref as R
    ix, proofs, fx = load_k15_fixture()
    install_index(ctx_srs, ix)
    plist, pubs = [p for _, p in proofs], [pi for pi, _ in proofs]
    npub = fx["npub"]

    def job(pl, rb, sb):
        a, o = kimchi_arrays(pl, pubs)
        kp = m.MinaContext.make_kimchi_proofs(len(pl), 2, npub, a)
        ja = dict(o); ja["rand_base"] = _le(rb); ja["sg_rand_base"] = _le(sb); ja["public_inputs"] = a["public_inputs"]
        return m.MinaContext.make_state_jobs(len(pl), ja, with_ipa=1, kimchi=kp, k=15, log2_domain=15, npub=npub, n_evalpoints=2, n_comms=47)

    def cancelling_pair(rb, t):
        # rho_b = rand_base^b: rho_0 = 1, rho_1 = rb.  z2 lives in the scalar field of Pallas (Fq).
        bad = [copy.deepcopy(p) for p in plist]
        bad[0]["opening"]["z2"] = (bad[0]["opening"]["z2"] + rb * t) % R.Q
        bad[1]["opening"]["z2"] = (bad[1]["opening"]["z2"] - t) % R.Q
        return bad

    B = len(plist)
    assert ctx_srs.state_job_batch(job(plist, 7, 9)).tolist() == [1] * B
    t = 0x1234567890ABCDEF1234567890ABCDEF
    # built for the randomiser the caller then passes: the folded sum is unchanged, both invalid proofs ride through (the caller's contract)
    assert ctx_srs.state_job_batch(job(cancelling_pair(7, t), 7, 9)).tolist() == [1] * B
    # any other randomiser: the fold fails, the culprit search names exactly the two
    rng = random.Random(99)
    for _ in range(3):
        rb = rng.randrange(1 << 254)
        assert ctx_srs.state_job_batch(job(cancelling_pair(7, t), rb, rng.randrange(1 << 254))).tolist() == [0, 0] + [1] * (B - 2)
    # each of them alone is invalid whatever the randomiser
    alone = cancelling_pair(7, t)
    assert ctx_srs.state_job_batch(job([alone[0]] + plist[1:], 7, 9)).tolist() == [0] + [1] * (B - 1)


def test_accumulator_fold_cancelling_sg_pair_needs_the_randomisers(ctx_srs, oracle, srs_oracle):
    from oracle import pasta_ref as R, state_job_ref as J
    O = oracle
    gv, _ = srs_oracle[1]
    k = 16
    acc = [J.make_accumulator(1, gv, k, 500 + i) for i in range(3)]
    pre = np.concatenate([a[0].reshape(-1) for a in acc]); sg = [O.bytes_to_point(a[1]) for a in acc]
    rho = [0x1111 + 7 * i for i in range(3)]
    T = O.bytes_to_point(gv[5])                                   # any Vesta point
    pack = lambda pts: np.concatenate([O.point_to_bytes(p) for p in pts])
    rho_b = np.concatenate([_le(r) for r in rho])
    assert ctx_srs.accumulator_check_batch(1, k, pre, pack(sg), rho_b).tolist() == [1, 1, 1]
    # sum_b rho_b (MSM_b - sg_b): + rho_0 T - rho_1 (rho_0 / rho_1) T = 0.  Vesta: base field Fq, scalar field Fp.
    ratio = rho[0] * pow(rho[1], -1, R.P) % R.P
    bad = list(sg)
    bad[0] = R.add(sg[0], T, R.Q)
    bad[1] = R.add(sg[1], R.neg(R.scalar_mul(ratio, T, R.Q), R.Q), R.Q)
    assert ctx_srs.accumulator_check_batch(1, k, pre, pack(bad), rho_b).tolist() == [1, 1, 1], "built for the rho the caller passes: accepted (the caller's contract)"
    rng = random.Random(5)
    other = np.concatenate([_le(rng.randrange(1 << 254)) for _ in range(3)])
    assert ctx_srs.accumulator_check_batch(1, k, pre, pack(bad), other).tolist() == [0, 0, 1]
    # the deterministic, un-folded form needs no randomness at all
    assert ctx_srs.accumulator_check_multi(1, k, pre, pack(bad)).tolist() == [0, 0, 1]


# ------------------------------------------------------------------------------------------------ the boundary: CSPRNG per job
# (this one first: it runs in a process state WITHOUT indexes and shuts the process-wide context down; `big` below sets its own up)
def test_boundary_rejects_cancelling_sg_pairs_on_the_accumulator_leg(oracle, srs_oracle):
    """the accumulator leg on its own at the boundary: a process WITHOUT a verifier index under MINA_VERIFY_ALLOW_MISSING_KIMCHI runs the
    chain and accumulator steps only (with an index the statement binds `challenge_polynomial_commitment`, so the tampered accumulators
    would fail the kimchi step as well).  sg_0' = sg_0 + T, sg_1' = sg_1 - r T for several guesses r of rho_0 / rho_1: both `false`."""
    import mina_bridge_amd as m
    from ipa_helpers import poseidon_pp
    from kimchi_helpers import load_statement_fixture, make_chain
    from oracle import mina_state_ref as S, pasta_ref as R
    from wire_writers import state_proof_bytes, state_pub_bytes
    O = oracle
    items, _ = load_statement_fixture()
    m.lib.verify_shutdown()
    m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE | m.lib.VERIFY_ALLOW_MISSING_KIMCHI)
    try:
        cases = []
        for it in items[:3]:
            states, hashes = make_chain(random.Random(it["chain_seed"]), poseidon_pp(0))
            p, ev = it["proof"], it["proof"]["evals"]
            wrap = dict(it["wrap"])
            wrap.update(w_comm=p["w_comm"], z_comm=p["z_comm"], t_comm=p["t_comm"], z_eval=ev[0], selector_eval=ev[1:7], w_eval=ev[7:22], coefficients_eval=ev[22:37],
                        s_eval=ev[37:43], ft_eval1=p["ft_eval1"], lr=p["opening"]["lr"], z1=p["opening"]["z1"], z2=p["opening"]["z2"], delta=p["opening"]["delta"], sg=p["opening"]["sg"])
            pub = state_pub_bytes(True, hashes[16], hashes[:16], [S.snarked_ledger_hash(s) for s in states[:16]])
            cases.append((wrap, states, pub))
        good = [state_proof_bytes(w, s) for w, s, _ in cases]; pubs = [q for _, _, q in cases]
        assert m.lib.verify_state_batch(good, pubs).tolist() == [1, 1, 1]
        T = O.bytes_to_point(srs_oracle[1][0][11])
        rng = random.Random(77)
        for r in [1, 2, R.P - 1] + [rng.randrange(1, R.P) for _ in range(3)]:
            w0 = copy.deepcopy(cases[0][0]); w1 = copy.deepcopy(cases[1][0])
            w0["challenge_polynomial_commitment"] = R.add(w0["challenge_polynomial_commitment"], T, R.Q)
            w1["challenge_polynomial_commitment"] = R.add(w1["challenge_polynomial_commitment"], R.neg(R.scalar_mul(r, T, R.Q), R.Q), R.Q)
            bad = [state_proof_bytes(w0, cases[0][1]), state_proof_bytes(w1, cases[1][1]), good[2]]
            assert m.lib.verify_state_batch(bad, pubs).tolist() == [0, 0, 1], r
            got = [None, None, None]
            gate = threading.Barrier(3)
            def worker(i):
                gate.wait(); got[i] = m.lib.verify_state(bad[i], pubs[i])
            th = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
            for t in th: t.start()
            for t in th: t.join()
            assert got == [False, False, True], r
    finally:
        m.lib.verify_configure(0)
        m.lib.verify_shutdown()


@pytest.fixture(scope="module")
def big(oracle):
    import mina_bridge_amd as m
    from ipa_helpers import poseidon_pp
    from kimchi_helpers import install_index, install_step_index, load_k15_fixture, load_statement_fixture, make_chain, make_step_index
    from oracle import mina_state_ref as S
    from wire_writers import state_proof_bytes, state_pub_bytes
    ix, _, _ = load_k15_fixture()
    items, fx = load_statement_fixture()
    m.lib.verify_shutdown()
    m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)
    gctx = m.lib.verify_global_ctx()
    install_index(gctx, ix)
    install_step_index(gctx, make_step_index(99))
    cases = []
    for it in items:
        states, hashes = make_chain(random.Random(it["chain_seed"]), poseidon_pp(0))
        p, ev = it["proof"], it["proof"]["evals"]
        wrap = dict(it["wrap"])
        wrap.update(w_comm=p["w_comm"], z_comm=p["z_comm"], t_comm=p["t_comm"], z_eval=ev[0], selector_eval=ev[1:7], w_eval=ev[7:22], coefficients_eval=ev[22:37],
                    s_eval=ev[37:43], ft_eval1=p["ft_eval1"], lr=p["opening"]["lr"], z1=p["opening"]["z1"], z2=p["opening"]["z2"], delta=p["opening"]["delta"],
                    sg=p["opening"]["sg"])
        ledger = [S.snarked_ledger_hash(s) for s in states[:16]]
        cases.append({"wrap": wrap, "states": states, "proof": state_proof_bytes(wrap, states), "pub": state_pub_bytes(True, hashes[16], hashes[:16], ledger)})
    yield {"m": m, "cases": cases}
    m.lib.verify_configure(0)
    m.lib.verify_shutdown()


def _z2_pair(cases, i, j, guess, t):
    from oracle import pasta_ref as R
    from wire_writers import state_proof_bytes
    a = copy.deepcopy(cases[i]["wrap"]); b = copy.deepcopy(cases[j]["wrap"])
    a["z2"] = (a["z2"] + guess * t) % R.Q; b["z2"] = (b["z2"] - t) % R.Q
    return state_proof_bytes(a, cases[i]["states"]), state_proof_bytes(b, cases[j]["states"])


def test_boundary_rejects_cancelling_z2_pairs(big):
    """through mina_verify_state_batch: the pair that passed the round-2 build (constant rand_base = 7), the same for other guesses, the pair
    in either order and buried among valid proofs -- always both `false`, the valid ones `true`"""
    m, cases = big["m"], big["cases"]
    assert m.lib.verify_state_batch([c["proof"] for c in cases], [c["pub"] for c in cases]).tolist() == [1] * len(cases)
    rng = random.Random(2024)
    for guess in [7, 9, 1, 7 * 7] + [rng.randrange(1 << 254) for _ in range(4)]:
        t = rng.randrange(1, 1 << 250)
        pa, pb = _z2_pair(cases, 0, 1, guess, t)
        assert m.lib.verify_state_batch([pa, pb], [cases[0]["pub"], cases[1]["pub"]]).tolist() == [0, 0], guess
        assert m.lib.verify_state_batch([pb, pa], [cases[1]["pub"], cases[0]["pub"]]).tolist() == [0, 0], guess
    pa, pb = _z2_pair(cases, 0, 1, 7, 12345)
    proofs = [cases[2]["proof"], pa, cases[3]["proof"], pb, cases[0]["proof"]]; pubs = [cases[2]["pub"], cases[0]["pub"], cases[3]["pub"], cases[1]["pub"], cases[0]["pub"]]
    assert m.lib.verify_state_batch(proofs, pubs).tolist() == [1, 0, 1, 0, 1]
    # one proof per call, as the reference verifies them: the same answers
    assert m.lib.verify_state(pa, cases[0]["pub"]) is False and m.lib.verify_state(pb, cases[1]["pub"]) is False


def test_merged_concurrent_calls_reject_cancelling_pairs(big):
    """two hostile callers time their mina_verify_state calls so that the call merger puts both proofs into ONE job (what the round-2 build
    folded with constants): both get `false`; honest callers in the same job get `true`"""
    m, cases = big["m"], big["cases"]
    pa, pb = _z2_pair(cases, 0, 1, 7, 0xDEADBEEF)
    calls = [(pa, cases[0]["pub"], False), (pb, cases[1]["pub"], False)] + [(cases[i % 4]["proof"], cases[i % 4]["pub"], True) for i in range(6)]
    for _ in range(3):
        got = [None] * len(calls)
        gate = threading.Barrier(len(calls))
        def worker(i):
            gate.wait()
            got[i] = m.lib.verify_state(calls[i][0], calls[i][1])
        th = [threading.Thread(target=worker, args=(i,)) for i in range(len(calls))]
        for t in th: t.start()
        for t in th: t.join()
        assert got == [c[2] for c in calls]


def test_surrogate_tables_and_half_configured_contexts_fail_closed(big):
    """a context on the library's surrogate Poseidon tables answers `false` unless the caller set MINA_VERIFY_ALLOW_SURROGATE; a wrap index
    without a step index does not run the kimchi step (the proof would not be bound to the statement): `false`, KIMCHI not in `ran`"""
    m, cases = big["m"], big["cases"]
    c = cases[0]
    assert "UNPINNED" in m.lib.poseidon_params_name()
    assert m.lib.verify_state(c["proof"], c["pub"]) is True
    m.lib.verify_configure(0)
    try:
        assert m.lib.verify_state(c["proof"], c["pub"]) is False
        assert m.lib.verify_state_batch([c["proof"]], [c["pub"]]).tolist() == [0]
    finally:
        m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)
    assert m.lib.verify_state(c["proof"], c["pub"]) is True


def test_one_malformed_shape_does_not_fail_the_other_callers(big):
    """a proof whose L/R count differs from the installed index (14 pairs instead of 15) parses, but must fail ALONE -- the honest
    proofs of the same job keep their `true`; likewise a proof with a lookup feature switched on (no lookup terms in the linearization:
    rejected at the kimchi step, not evaluated with zeros)"""
    from wire_writers import state_proof_bytes
    m, cases = big["m"], big["cases"]
    w = copy.deepcopy(cases[1]["wrap"]); w["lr"] = w["lr"][:14]
    short = state_proof_bytes(w, cases[1]["states"])
    w2 = copy.deepcopy(cases[2]["wrap"]); w2["feature_flags"][6] = True
    lookup = state_proof_bytes(w2, cases[2]["states"])
    proofs = [cases[0]["proof"], short, cases[2]["proof"], lookup, cases[3]["proof"], b"\x00" * 100]
    pubs = [cases[0]["pub"], cases[1]["pub"], cases[2]["pub"], cases[2]["pub"], cases[3]["pub"], cases[0]["pub"]]
    assert m.lib.verify_state_batch(proofs, pubs).tolist() == [1, 0, 1, 0, 1, 0]
    passed, ran = m.lib.verify_state_checks(short, cases[1]["pub"])
    assert ran & 32 and not passed & 32 and passed & 1, "the malformed shape parses (FORMAT) and fails the kimchi step"
    passed, ran = m.lib.verify_state_checks(lookup, cases[2]["pub"])
    assert ran & 32 and not passed & 32
    # all entries of a call unusable: nothing reaches the GPU, all false
    assert m.lib.verify_state_batch([short, b""], [cases[1]["pub"], cases[0]["pub"]]).tolist() == [0, 0]


def test_network_flag_of_the_public_input_must_match_the_declared_network(big):
    """upstream picks the devnet or the mainnet verifier index by `is_state_proof_from_devnet`; this library holds one index pair, so
    once its network is declared a proof that claims the other one fails (its own verdict only, at the kimchi step)"""
    m, cases = big["m"], big["cases"]
    c = cases[0]
    assert c["pub"][0] == 1                                        # the fixtures' public inputs say devnet
    try:
        m.lib.verify_set_network(1)
        assert m.lib.verify_state(c["proof"], c["pub"]) is True
        m.lib.verify_set_network(0)
        assert m.lib.verify_state(c["proof"], c["pub"]) is False
        passed, ran = m.lib.verify_state_checks(c["proof"], c["pub"])
        assert ran & 32 and not passed & 32 and passed & (1 | 2 | 4 | 8 | 16) == (1 | 2 | 4 | 8 | 16)
        main = bytes([0]) + c["pub"][1:]                           # the same proof claiming mainnet: accepted by the (synthetic) mainnet index
        assert m.lib.verify_state_batch([c["proof"], c["proof"]], [main, c["pub"]]).tolist() == [1, 0]
    finally:
        m.lib.verify_set_network(-1)
    assert m.lib.verify_state(c["proof"], c["pub"]) is True
