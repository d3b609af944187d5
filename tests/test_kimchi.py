"""kimchi `oracles` + `to_batch` on the GPU vs oracle/kimchi_ref.py on a synthetic verifier index: the
BatchEvaluationProof the GPU builds equals the oracle's field for field; proofs minted by the oracle's miniature prover are
accepted end to end through the Proof-of-State job's kimchi leg, tampered ones rejected (with the culprit isolated)."""
import copy
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
K_LOG2, NPUB = 6, 5


@pytest.fixture(scope="module")
def circuit(srs_oracle):
    from ipa_helpers import poseidon_pp
    from oracle import kimchi_ref as K, oracle as O
    g, h = srs_oracle[0]
    return K.synthetic_circuit(0, g, O.bytes_to_point(h), poseidon_pp(0), poseidon_pp(1), K_LOG2, NPUB, seed=21)


@pytest.fixture(scope="module")
def proofs(srs_oracle, circuit):
    from ipa_helpers import poseidon_pp
    from oracle import kimchi_ref as K, oracle as O, pasta_ref as R
    g, h = srs_oracle[0]
    out = []
    for s in range(4):
        rng = random.Random(100 + s)
        pubs = [rng.randrange(R.Q) for _ in range(NPUB)]
        out.append((pubs, K.synthetic_proof(circuit, g, O.bytes_to_point(h), poseidon_pp(0), poseidon_pp(1), pubs, seed=200 + s)))
    return out


def _oracle(circuit, srs_oracle, pubs, proof):
    from ipa_helpers import poseidon_pp
    from oracle import kimchi_ref as K, oracle as O
    g, h = srs_oracle[0]
    return K.oracles_and_batch(circuit.index, proof, pubs, poseidon_pp(0), poseidon_pp(1), g, O.bytes_to_point(h))


def test_index_digest_and_batch_rows_match_oracle(ctx_srs, oracle, srs_oracle, circuit, proofs):
    import mina_bridge_amd as m
    from kimchi_helpers import install_index, kimchi_arrays
    install_index(ctx_srs, circuit.index)
    assert oracle.le_to_int(ctx_srs.verifier_index_digest()) == circuit.index.digest
    arrays, _ = kimchi_arrays([p for _, p in proofs], [pi for pi, _ in proofs])
    kp = m.MinaContext.make_kimchi_proofs(len(proofs), 2, NPUB, arrays)
    got = ctx_srs.kimchi_to_batch(kp, K_LOG2)
    assert not got["malformed"][0]
    for b, (pubs, proof) in enumerate(proofs):
        o, entry = _oracle(circuit, srs_oracle, pubs, proof)
        st, mode, cnt = o["sponge_after"].raw()
        assert [oracle.le_to_int(got["sponge_state"][b, 32 * i: 32 * i + 32]) for i in range(3)] == st
        assert got["sponge_pos"][b].tolist() == [mode, cnt]
        assert oracle.le_to_int(got["ft_eval0"][b]) == o["ft_eval0"]
        assert oracle.le_to_int(got["polyscale"][b]) == o["v"] and oracle.le_to_int(got["evalscale"][b]) == o["u"]
        assert [oracle.le_to_int(got["evalpoints"][b, :32]), oracle.le_to_int(got["evalpoints"][b, 32:])] == entry["evalpoints"]
        assert oracle.le_to_int(got["cip"][b]) == o["combined_inner_product"]
        assert [oracle.bytes_to_point(c) for c in got["comms"][b]] == o["comms"]          # includes public_comm and the chunked ft_comm
    # a non-canonical evaluation / an off-curve commitment raise the malformed flag
    bad = dict(arrays); ev = arrays["evals"].copy(); ev[:32] = 0xFF; bad["evals"] = ev
    assert ctx_srs.kimchi_to_batch(m.MinaContext.make_kimchi_proofs(len(proofs), 2, NPUB, bad), K_LOG2)["malformed"][0]
    bad = dict(arrays); tc = arrays["t_comm"].copy(); tc[3 * 64] ^= 1; bad["t_comm"] = tc
    assert ctx_srs.kimchi_to_batch(m.MinaContext.make_kimchi_proofs(len(proofs), 2, NPUB, bad), K_LOG2)["malformed"][0]


def _kimchi_job(m, proofs_list, publics):
    from kimchi_helpers import kimchi_arrays
    from oracle import oracle as O
    B = len(proofs_list)
    arrays, op = kimchi_arrays(proofs_list, publics)
    kp = m.MinaContext.make_kimchi_proofs(B, 2, NPUB, arrays)
    ja = dict(op); ja["rand_base"] = O.int_to_le(7); ja["sg_rand_base"] = O.int_to_le(9); ja["public_inputs"] = arrays["public_inputs"]
    return m.MinaContext.make_state_jobs(B, ja, with_ipa=1, kimchi=kp, k=K_LOG2, log2_domain=K_LOG2, npub=NPUB, n_evalpoints=2, n_comms=47)


def test_kimchi_leg_accepts_minted_proofs_and_isolates_tampered_ones(ctx_srs, oracle, srs_oracle, circuit, proofs):
    import mina_bridge_amd as m
    from kimchi_helpers import install_index
    from oracle import ipa_ref as I, pasta_ref as R
    install_index(ctx_srs, circuit.index)
    plist, pubs = [p for _, p in proofs], [pi for pi, _ in proofs]
    assert ctx_srs.state_job_batch(_kimchi_job(m, plist, pubs)).tolist() == [1, 1, 1, 1]
    assert ctx_srs.state_job_batch(_kimchi_job(m, plist[:1], pubs[:1])).tolist() == [1]
    # tamper: an evaluation (proof 1), a public input (proof 2), ft_eval1 (proof 3); the oracle rejects the same ones
    bad = [copy.deepcopy(p) for p in plist]; bpubs = [list(x) for x in pubs]
    bad[1]["evals"][9] = ((bad[1]["evals"][9][0] + 1) % R.Q, bad[1]["evals"][9][1])
    bpubs[2][0] = (bpubs[2][0] + 1) % R.Q
    bad[3]["ft_eval1"] = (bad[3]["ft_eval1"] + 1) % R.Q
    assert ctx_srs.state_job_batch(_kimchi_job(m, bad, bpubs)).tolist() == [1, 0, 0, 0]
    g, h = srs_oracle[0]
    for b in range(4):
        _, entry = _oracle(circuit, srs_oracle, bpubs[b], bad[b])
        assert I.ipa_verify_batch(0, g[: 1 << K_LOG2], oracle.bytes_to_point(h), [entry], 7, 9) == (b == 0)
    # a commitment swapped for another valid point (w_comm[4] <- w_comm[5]) and a recursion challenge changed
    bad = [copy.deepcopy(p) for p in plist]
    bad[0]["w_comm"][4] = bad[0]["w_comm"][5]
    ch, cm = bad[2]["prev"][1]; bad[2]["prev"][1] = ([(ch[0] + 1) % R.Q] + ch[1:], cm)
    assert ctx_srs.state_job_batch(_kimchi_job(m, bad, pubs)).tolist() == [0, 1, 0, 1]


def test_kimchi_full_size_wrap_domain(ctx_srs, oracle):
    """the Pickles wrap size: domain 2^15 = SRS chunk, 40 public inputs, 2 recursion challenges, 47 commitments (committed fixture minted
    by the oracle's prover): digest, ft_eval0, v, u, combined inner product and the chunked ft_comm equal the oracle's; all four proofs
    are accepted through the job's kimchi leg, a tampered one is isolated"""
    import mina_bridge_amd as m
    from kimchi_helpers import install_index, kimchi_arrays, load_k15_fixture
    ix, proofs, fx = load_k15_fixture()
    install_index(ctx_srs, ix)
    assert oracle.le_to_int(ctx_srs.verifier_index_digest()) == ix.digest
    plist, pubs = [p for _, p in proofs], [pi for pi, _ in proofs]
    arrays, op = kimchi_arrays(plist, pubs)
    B, npub = len(plist), fx["npub"]
    got = ctx_srs.kimchi_to_batch(m.MinaContext.make_kimchi_proofs(B, 2, npub, arrays), 15)
    assert not got["malformed"][0]
    for b, p in enumerate(plist):
        e = p["expect"]
        assert oracle.le_to_int(got["ft_eval0"][b]) == int(e["ft_eval0"]) and oracle.le_to_int(got["cip"][b]) == int(e["cip"])
        assert oracle.le_to_int(got["polyscale"][b]) == int(e["v"]) and oracle.le_to_int(got["evalscale"][b]) == int(e["u"])
        assert got["comms"][b, 3].tobytes().hex() == e["ft_comm"]

    def job(pl, pu):
        a, o = kimchi_arrays(pl, pu)
        kp = m.MinaContext.make_kimchi_proofs(len(pl), 2, npub, a)
        ja = dict(o); ja["rand_base"] = oracle.int_to_le(7); ja["sg_rand_base"] = oracle.int_to_le(9); ja["public_inputs"] = a["public_inputs"]
        return m.MinaContext.make_state_jobs(len(pl), ja, with_ipa=1, kimchi=kp, k=15, log2_domain=15, npub=npub, n_evalpoints=2, n_comms=47)
    assert ctx_srs.state_job_batch(job(plist, pubs)).tolist() == [1] * B
    import copy
    bad = [copy.deepcopy(p) for p in plist]
    bad[2]["evals"][40] = (bad[2]["evals"][40][0], (bad[2]["evals"][40][1] + 1) % (1 << 254))
    assert ctx_srs.state_job_batch(job(bad, pubs)).tolist() == [1, 1, 0, 1]


def test_statement_driven_job_full_size(ctx_srs, oracle):
    """tests/golden/statement_k15.json: complete wrap proofs at the Pickles wrap size -- statement, the proof whose public input is its
    packing, the accumulator it carries.  The job derives the public inputs on the GPU from the statements (mina_pickles_statements),
    runs kimchi + the opening check + the accumulator check, accepts all; a change to ANY statement field (here: one feature flag, the
    application state, one step evaluation) rejects exactly that proof -- every field is bound through the public input."""
    import copy
    import mina_bridge_amd as m
    from kimchi_helpers import install_index, install_step_index, kimchi_arrays, load_k15_fixture, load_statement_fixture, make_step_index, statements_soa
    ix, _, _ = load_k15_fixture()
    install_index(ctx_srs, ix)
    install_step_index(ctx_srs, make_step_index(99))
    items, fx = load_statement_fixture()
    B = len(items)
    # the GPU derivation equals the packing the fixture's proofs were minted for
    n_old, n_evals, sec = statements_soa([it["wrap"] for it in items], [it["app"] for it in items])
    pub, ok = ctx_srs.pickles_public_inputs_batch(m.MinaContext.make_pickles_statements(n_old, n_evals, sec), B)
    assert ok.tolist() == [1] * B
    for b, it in enumerate(items):
        assert [oracle.le_to_int(x) for x in pub[b]] == it["pubs"]

    def job(wraps, apps):
        a, o = kimchi_arrays([it["proof"] for it in items], [])
        no, ne, s = statements_soa(wraps, apps)
        kp = m.MinaContext.make_kimchi_proofs(B, 2, 40, a, statements=m.MinaContext.make_pickles_statements(no, ne, s))
        ja = dict(o); ja["rand_base"] = oracle.int_to_le(7); ja["sg_rand_base"] = oracle.int_to_le(9)
        ja["acc_prechallenges"] = np.concatenate([it["acc_pre"].reshape(-1) for it in items]); ja["acc_sg"] = np.concatenate([it["acc_sg"] for it in items])
        rho = np.random.Generator(np.random.PCG64(5)).integers(0, 256, (B, 32), dtype=np.uint8); rho[:, 31] &= 0x3F
        ja["acc_rho"] = rho.reshape(-1)
        return m.MinaContext.make_state_jobs(B, ja, with_ipa=1, with_accumulator=1, kimchi=kp, k=15, log2_domain=15, npub=40, n_evalpoints=2, n_comms=47, acc_k=16)
    wraps, apps = [it["wrap"] for it in items], [it["app"] for it in items]
    assert ctx_srs.state_job_batch(job(wraps, apps)).tolist() == [1] * B
    bad = copy.deepcopy(wraps); bapps = list(apps)
    bad[0]["feature_flags"][3] = not bad[0]["feature_flags"][3]
    bapps[1] = (bapps[1] + 1) % (1 << 254)
    e = bad[3]["prev_evals"][20]; bad[3]["prev_evals"][20] = ([(e[0][0] + 1) % (1 << 254)], e[1])
    assert ctx_srs.state_job_batch(job(bad, bapps)).tolist() == [0, 0, 1, 0]

    # the recursion challenges left out: the job takes them from the statement (messages_for_next_wrap_proof.old_bulletproof_challenges --
    # the one source a verifier has) and kimchi's digest of them from the statement's own sponge; same verdicts, and a changed challenge in
    # the statement now fails that proof twice over (its digest in the public input, and the challenge polynomial the opening evaluates)
    def job_from_statement(wraps, apps):
        j, keep = job(wraps, apps)
        kp = m.lib.KimchiProofs.from_address(j.kimchi)
        kp.prev_chals = 0; kp.prev_prechallenges = 0
        return j, keep
    assert ctx_srs.state_job_batch(job_from_statement(wraps, apps)).tolist() == [1] * B
    assert ctx_srs.state_job_batch(job_from_statement(bad, bapps)).tolist() == [0, 0, 1, 0]
    bad2 = copy.deepcopy(wraps); bad2[2]["old_bulletproof_challenges"][1][7] ^= 1 << 77
    assert ctx_srs.state_job_batch(job_from_statement(bad2, apps)).tolist() == [1, 1, 0, 1]

