"""The reference-shaped boundary on the GPU: `mina_verify_state` / `mina_verify_account` fed with the
bytes core/src/aligned.rs:31-58 produces (independent Python writers), every step reported by `*_checks`, tamper -> false,
batch and `--save-proof` file forms, and the plain-C consumer."""
import copy
import os
import random
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
K_LOG2 = 6


from kimchi_helpers import make_chain as _chain  # noqa: E402


@pytest.fixture(scope="module")
def world(srs_oracle):
    """global context with a synthetic wrap index installed + everything needed to mint state proofs for it"""
    import mina_bridge_amd as m
    from ipa_helpers import poseidon_pp
    from kimchi_helpers import install_index
    from oracle import kimchi_ref as K, oracle as O
    g, h = srs_oracle[0]
    circ = K.synthetic_circuit(0, g, O.bytes_to_point(h), poseidon_pp(0), poseidon_pp(1), K_LOG2, 40, seed=77)
    m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)      # the tests run on the surrogate Poseidon tables (the real ones are not offline)
    gctx = m.lib.verify_global_ctx()
    install_index(gctx, circ.index)
    step = make_step_index(99)
    install_step_index(gctx, step)
    yield {"circ": circ, "gctx": gctx, "step": step}
    m.lib.verify_configure(0)


from kimchi_helpers import STEP_DOMAINS, install_step_index, make_step_index  # noqa: E402


def mint_state_proof(world, srs_oracle, seed, optional_slots=(), statement_overrides=None):
    from ipa_helpers import poseidon_pp
    from oracle import kimchi_ref as K, oracle as O, pasta_ref as R, state_job_ref as J, mina_state_ref as S
    from wire_writers import synth_wrap_proof
    rng = random.Random(seed)
    g, h = srs_oracle[0]; gv, _ = srs_oracle[1]
    wrap = synth_wrap_proof(rng, k=K_LOG2, lookups=False)
    # recursion challenges of the wrap proof: 128-bit prechallenges on the wire, endo-expanded by the verifier
    pres = [[rng.getrandbits(128) for _ in range(15)] for _ in range(2)]
    chals = [[R.challenge_to_field(p, R.endo_r(0), R.Q) for p in row[:K_LOG2]] for row in pres]
    wrap["prev_optional"] = [(([rng.randrange(R.P)], [rng.randrange(R.P)]) if i in optional_slots else None) for i in range(19)]
    wrap.update(statement_overrides or {})
    wrap["old_bulletproof_challenges"] = pres
    prev_comms = []
    for ch in chals:                                            # the previous wrap accumulators: commitments of b_poly_coefficients(chals)
        sc = [O.le_to_int(x) for x in O.b_poly_coefficients(1, O.ints_to_le(ch))]
        from oracle import ipa_ref as I
        prev_comms.append(I.commit(0, g[: 1 << K_LOG2], O.bytes_to_point(h), sc, 0))
    wrap["step_comms"] = prev_comms
    pre, sg = J.make_accumulator(1, gv, 16, seed + 2)
    wrap["bulletproof_challenges"] = [int.from_bytes(pre[i].tobytes(), "little") for i in range(16)]
    wrap["challenge_polynomial_commitment"] = O.bytes_to_point(sg)
    states, hashes = _chain(rng, poseidon_pp(0))
    # the wrap circuit's public input IS the statement: deferred values recomputed from prev_evals, the two message digests, packing
    from oracle import pickles_ref as PK
    ix = world["circ"].index
    comms = list(ix.sigma_comm) + list(ix.coefficients_comm) + list(ix.selector_comm)
    pubs, dv, mw_, ms_ = PK.statement_public_input(wrap, world["step"], comms, hashes[15], poseidon_pp(0), poseidon_pp(1))
    proof = K.synthetic_proof(world["circ"], g, O.bytes_to_point(h), poseidon_pp(0), poseidon_pp(1), pubs, seed=seed + 1, prev_chals=chals)
    assert [cm for _, cm in proof["prev"]] == prev_comms
    ev = proof["evals"]
    wrap.update(w_comm=proof["w_comm"], z_comm=proof["z_comm"], t_comm=proof["t_comm"],
                z_eval=ev[0], selector_eval=ev[1:7], w_eval=ev[7:22], coefficients_eval=ev[22:37], s_eval=ev[37:43], ft_eval1=proof["ft_eval1"],
                lr=proof["opening"]["lr"], z1=proof["opening"]["z1"], z2=proof["opening"]["z2"], delta=proof["opening"]["delta"], sg=proof["opening"]["sg"])
    return wrap, states, hashes


def to_bytes(wrap, states, hashes, ledger=None):
    from oracle import mina_state_ref as S
    from wire_writers import state_proof_bytes, state_pub_bytes
    ledger = ledger or [S.snarked_ledger_hash(s) for s in states[:16]]
    return state_proof_bytes(wrap, states), state_pub_bytes(True, hashes[16], hashes[:16], ledger)


ALL = 1 | 2 | 4 | 8 | 16 | 32


def test_verify_state_end_to_end(world, srs_oracle, tmp_path):
    import mina_bridge_amd as m
    wrap, states, hashes = mint_state_proof(world, srs_oracle, 1000)
    proof, pub = to_bytes(wrap, states, hashes)
    assert len(pub) == 1057
    assert m.lib.verify_state_checks(proof, pub) == (ALL, ALL)
    assert m.lib.verify_state(proof, pub) is True
    # the --save-proof file form (core/src/aligned.rs:60-69)
    (tmp_path / "mina_state.proof").write_bytes(proof); (tmp_path / "mina_state.pub").write_bytes(pub)
    assert m.lib.verify_state_files(str(tmp_path / "mina_state.proof"), str(tmp_path / "mina_state.pub")) is True
    assert m.lib.verify_state_files(str(tmp_path / "missing"), str(tmp_path / "mina_state.pub")) is False
    # every failure is `false`: garbage, truncation, wrong pub length
    assert m.lib.verify_state(b"", pub) is False and m.lib.verify_state(proof[:-1], pub) is False and m.lib.verify_state(proof, pub[:-1]) is False
    assert m.lib.verify_state_checks(proof[:100], pub) == (0, 1)

    def masks(w=wrap, s=states, hs=hashes, ledger=None):
        p, q = to_bytes(w, s, hs, ledger)
        passed, ran = m.lib.verify_state_checks(p, q)
        assert m.lib.verify_state(p, q) is (passed == ALL)
        return passed
    # LEDGER: a ledger hash of the public input that is not the state's
    from oracle import mina_state_ref as S
    led = [S.snarked_ledger_hash(s) for s in states[:16]]; led[3] ^= 1
    assert masks(ledger=led) == ALL & ~2
    # CHAIN: a state hash of the public input off by one; a broken link
    hs = list(hashes); hs[5] ^= 1
    assert masks(hs=hs) == ALL & ~4
    # CONSENSUS: the bridge tip is the longer chain
    st = copy.deepcopy(states); st[16]["body"]["consensus_state"]["blockchain_length"] = 5000
    from ipa_helpers import poseidon_pp
    hs = list(hashes); hs[16] = S.protocol_state_hash(st[16], poseidon_pp(0))
    assert masks(s=st, hs=hs) == ALL & ~8
    # ACCUMULATOR: one step prechallenge changed (the prechallenges are also part of the statement the kimchi step binds)
    w2 = dict(wrap); bc = list(wrap["bulletproof_challenges"]); bc[7] ^= 1; w2["bulletproof_challenges"] = bc
    assert masks(w=w2) == ALL & ~16 & ~32
    # KIMCHI: an evaluation changed / an opening scalar changed
    w2 = dict(wrap); we = list(wrap["w_eval"]); we[2] = ((we[2][0] + 1) % (1 << 254), we[2][1]); w2["w_eval"] = we
    assert masks(w=w2) == ALL & ~32
    w2 = dict(wrap); w2["z1"] = (wrap["z1"] + 1) % (1 << 254)
    assert masks(w=w2) == ALL & ~32
    # KIMCHI binds the STATEMENT: any statement field that enters the public input (directly or through the deferred values / digests)
    for field, mutate in (("alpha", lambda v: v ^ 1), ("sponge_digest", lambda v: [v[0] ^ 1] + v[1:]), ("prev_ft_eval1", lambda v: (v + 1) % (1 << 254)),
                          ("domain_log2", lambda v: 10 if v != 10 else 11), ("feature_flags", lambda v: [not v[0]] + v[1:]),
                          ("step_old_chals", lambda v: [[v[0][0] ^ 1] + v[0][1:]] + v[1:]), ("prev_evals", lambda v: [([v[0][0][0] ^ 1], v[0][1])] + v[1:])):
        w2 = dict(wrap); w2[field] = mutate(wrap[field])
        assert masks(w=w2) == ALL & ~32, field
    # ... including the application state: a wrap proof made for another tip hash (state chain re-linked, all other steps still pass)
    st2 = copy.deepcopy(states); st2[15]["body"]["consensus_state"]["total_currency"] ^= 1
    hs2 = list(hashes); hs2[15] = S.protocol_state_hash(st2[15], poseidon_pp(0))
    assert masks(s=st2, hs=hs2) == ALL & ~32


def test_verify_state_batch_and_missing_index_policy(world, srs_oracle):
    import mina_bridge_amd as m
    items = [mint_state_proof(world, srs_oracle, 2000 + 10 * i) for i in range(3)]
    pairs = [to_bytes(*it) for it in items]
    bad = dict(items[1][0]); bad["ft_eval1"] = (bad["ft_eval1"] + 1) % (1 << 254)
    pairs[1] = to_bytes(bad, items[1][1], items[1][2])
    v = m.lib.verify_state_batch([p for p, _ in pairs] + [b"junk"], [q for _, q in pairs] + [pairs[0][1]])
    assert v.tolist() == [1, 0, 1, 0]
    assert m.lib.verify_state_batch([], []).tolist() == []


def test_verify_account_end_to_end(world, srs_oracle, ctx, tmp_path):
    """MinaAccountProof + MinaAccountPubInputs bytes (independent writers) through mina_verify_account: ABI cross-check, account hash on the
    GPU == oracle, Merkle fold == ledger hash; config C4 shape (depth 35)"""
    import struct
    import mina_bridge_amd as m
    from ipa_helpers import poseidon_pp
    from oracle import mina_account_ref as A, pasta_ref as R
    pp = poseidon_pp(0)
    rng = random.Random(31)
    accounts = [A.synth_account(rng, zk, timed, deleg, with_vk=vk) for zk, timed, deleg, vk in
                [(False, False, False, True), (True, True, True, True), (True, False, True, False), (False, True, True, True)]]
    # account hash parity, both encodings
    want = [A.account_hash(a, pp) for a in accounts]
    for enc, bp in ((m.lib.ENC_BINPROT, True), (m.lib.ENC_BINCODE, False)):
        got = ctx.account_hash_batch([A.write_account(a, bp) for a in accounts], enc)
        assert [int.from_bytes(x.tobytes(), "little") for x in got] == want
    proofs, pubs = [], []
    for a, leaf in zip(accounts, want):
        path = [(rng.randrange(2), rng.randrange(R.P)) for _ in range(35)]
        root = R.merkle_root(leaf, path, pp)
        enc = A.abi_encode_account(a)
        proofs.append(A.write_account_proof(path, a)); pubs.append(root.to_bytes(32, "little") + struct.pack("<Q", len(enc)) + enc)
    OK = 1 | 64 | 128
    for p, q in zip(proofs, pubs):
        assert m.lib.verify_account_checks(p, q) == (OK, OK) and m.lib.verify_account(p, q) is True
    assert m.lib.verify_account_batch(proofs, pubs).tolist() == [1, 1, 1, 1]
    (tmp_path / "mina_account.proof").write_bytes(proofs[1]); (tmp_path / "mina_account.pub").write_bytes(pubs[1])
    assert m.lib.verify_account_files(str(tmp_path / "mina_account.proof"), str(tmp_path / "mina_account.pub")) is True
    # tampering: the ABI bytes (balance word), the account itself (hash changes -> Merkle fails, ABI fails too), a sibling, the ledger hash
    q = bytearray(pubs[0]); q[40 + 32 + 5 * 32 + 31] ^= 1
    assert m.lib.verify_account_checks(proofs[0], bytes(q)) == (1 | 128, OK)
    a2 = copy.deepcopy(accounts[0]); a2["nonce"] ^= 1
    p2 = A.write_account_proof([(0, 1)] * 35, a2)
    assert m.lib.verify_account_checks(p2, pubs[0])[0] == 1
    p3 = bytearray(proofs[2]); p3[8 + 12 + 3] ^= 1
    assert m.lib.verify_account_checks(bytes(p3), pubs[2])[0] in (1 | 64, 0)          # sibling changed (or no longer canonical)
    q = bytearray(pubs[3]); q[0] ^= 1
    assert m.lib.verify_account_checks(proofs[3], bytes(q))[0] == 1 | 64
    assert m.lib.verify_account(b"", pubs[0]) is False and m.lib.verify_account(proofs[0][:-3], pubs[0]) is False
    assert m.lib.verify_account_batch(proofs + [b"x"], pubs + [pubs[0]]).tolist() == [1, 1, 1, 1, 0]
    # C4: 256 proofs in one batch (64 distinct accounts x 4), one tampered
    big_p, big_q = proofs * 64, pubs * 64
    big_q[100] = pubs[(100 + 1) % 4]
    v = m.lib.verify_account_batch(big_p, big_q)
    assert v.sum() == 255 and v[100] == 0
    # the reference's call pattern: one proof per call from many threads at once -- every caller its own verdict (calls that arrive while a
    # job runs are merged into the next one, api_verify.hip)
    import threading
    calls = [(big_p[i], big_q[i], i != 100) for i in range(90, 122)]
    got = [None] * len(calls)
    def worker(i): got[i] = m.lib.verify_account(calls[i][0], calls[i][1])
    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(calls))]
    for t in th: t.start()
    for t in th: t.join()
    assert got == [c[2] for c in calls]
    # small BATCHES from several threads share jobs the same way: thread t's batch of 12 has a tampered public input at index t, another at 11
    outs = [None] * 6
    def batch_worker(t):
        P = [proofs[(t + j) % 4] for j in range(12)]; Q = [pubs[(t + j) % 4] for j in range(12)]
        Q[t] = pubs[(t + t + 1) % 4]; Q[11] = pubs[(t + 11 + 2) % 4]
        for _ in range(3): outs[t] = m.lib.verify_account_batch(P, Q).tolist()
    th = [threading.Thread(target=batch_worker, args=(t,)) for t in range(6)]
    for t in th: t.start()
    for t in th: t.join()
    for t in range(6): assert outs[t] == [0 if j in (t, 11) else 1 for j in range(12)], (t, outs[t])


def test_c_consumer_of_the_boundary(world, srs_oracle, tmp_path):
    """plain C (gcc, no HIP headers): reads the two files and calls mina_verify_state like the operator's cgo stub would"""
    import mina_bridge_amd as m
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "consumer.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include "mina_verify.h"
static unsigned char *slurp(const char *p, size_t *n) { FILE *f = fopen(p, "rb"); if (!f) return 0; fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); rewind(f);
  unsigned char *b = malloc(*n + 1); if (fread(b, 1, *n, f) != *n) { fclose(f); return 0; } fclose(f); return b; }
int main(int argc, char **argv) {
  size_t pl = 0, ql = 0; unsigned char *p = slurp(argv[1], &pl), *q = slurp(argv[2], &ql);
  if (argc < 3 || !p || !q) return 2;
  unsigned passed = 0, ran = 0;
  if (mina_verify_state_checks(p, pl, q, ql, &passed, &ran) != MINA_OK) return 3;
  printf("%s passed=%u ran=%u\n", mina_verify_state(p, pl, q, ql) ? "true" : "false", passed, ran);
  return 0; }
''')
    exe = tmp_path / "consumer"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L", os.path.dirname(m.LIB_PATH), "-lminaverify", "-Wl,-rpath," + os.path.dirname(m.LIB_PATH)])
    wrap, states, hashes = mint_state_proof(world, srs_oracle, 3000)
    proof, pub = to_bytes(wrap, states, hashes)
    (tmp_path / "p").write_bytes(proof); (tmp_path / "q").write_bytes(pub)
    r = subprocess.run([str(exe), str(tmp_path / "p"), str(tmp_path / "q")], capture_output=True, text=True, timeout=300)
    # a fresh process has no verifier index installed: every other step passes, the verdict is false (MINA_CHECK_KIMCHI = 32 did not run)
    assert r.returncode == 0 and r.stdout.strip() == "false passed=31 ran=31", r.stdout + r.stderr


def test_c4_account_batch_concurrent_with_msm_mix(ctx_srs, oracle, srs_oracle):
    """BASELINE config C4 ("256 Proof-of-Account verifies, Poseidon-heavy Merkle-path + MSM mix"): the 256-proof account batch on
    one context while another context of the same GPU runs Proof-of-State jobs (MSM-heavy) from a second host thread; both results
    stay exact"""
    import struct
    import threading
    import mina_bridge_amd as m
    from ipa_helpers import poseidon_pp
    from oracle import mina_account_ref as A, pasta_ref as R
    from state_job_helpers import build_jobs, mint_job
    pp = poseidon_pp(0)
    rng = random.Random(77)
    accounts = [A.synth_account(rng, i % 3 == 0, i % 2 == 0, i % 5 != 0) for i in range(16)]
    proofs, pubs = [], []
    for a in accounts:
        leaf = A.account_hash(a, pp)
        path = [(rng.randrange(2), rng.randrange(R.P)) for _ in range(35)]
        enc = A.abi_encode_account(a)
        proofs.append(A.write_account_proof(path, a)); pubs.append(R.merkle_root(leaf, path, pp).to_bytes(32, "little") + struct.pack("<Q", len(enc)) + enc)
    proofs, pubs = proofs * 16, pubs * 16                        # 256
    pubs[200] = pubs[201]
    shape = dict(k=7, log2_domain=7, npub=8, n_comms=6, slot=2, n_points=2, acc_k=16)          # 2^16 Vesta accumulator MSM per proof
    jobs = [mint_job(srs_oracle[0], srs_oracle[1], 4000 + i, **shape) for i in range(4)]
    sj = build_jobs(m, jobs, 7, 7, 2, 16)
    acc_ctx = m.MinaContext(0)
    for f in (0, 1):
        acc_ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
    res = {}

    def accounts_thread():
        for it in range(3):
            res["acc%d" % it] = acc_ctx.verify_account_checks(proofs, pubs)[0]

    def msm_thread():
        for it in range(6):
            res["job%d" % it] = ctx_srs.state_job_batch(sj)

    try:
        ts = [threading.Thread(target=accounts_thread), threading.Thread(target=msm_thread)]
        for t in ts: t.start()
        for t in ts: t.join()
    finally:
        acc_ctx.close()
    OK = 1 | 64 | 128
    for it in range(3):
        v = res["acc%d" % it]
        assert (v == OK).sum() == 255 and v[200] == 1          # another account's public input: neither the ABI bytes nor the ledger hash match
    for it in range(6):
        assert res["job%d" % it].tolist() == [1, 1, 1, 1]


def test_pickles_public_input_matches_oracle(world, srs_oracle):
    """compute_deferred_values + message digests + statement packing (api_pickles.hip, sponges on the GPU) == oracle/pickles_ref.py for
    random statements: optional evaluations present, chunked evaluations, 0..3 previous accumulators, every step domain"""
    import mina_bridge_amd as m
    from ipa_helpers import poseidon_pp
    from oracle import oracle as O, pickles_ref as PK
    from wire_writers import synth_wrap_proof, wrap_proof_bytes
    rng = random.Random(123)
    ix = world["circ"].index
    comms = list(ix.sigma_comm) + list(ix.coefficients_comm) + list(ix.selector_comm)
    gctx = world["gctx"]
    for case in range(8):
        w = synth_wrap_proof(rng, k=K_LOG2)
        if case % 2:
            w["prev_evals"][5] = ([rng.randrange(PK.P), rng.randrange(PK.P)], [rng.randrange(PK.P), rng.randrange(PK.P)])        # two chunks
        nprev = case % 4
        w["step_comms"] = w["step_comms"][:1] * nprev if nprev else []
        w["step_old_chals"] = [[rng.getrandbits(128) for _ in range(16)] for _ in range(nprev)]
        app = rng.randrange(PK.P)
        want, dv, mw_, ms_ = PK.statement_public_input(w, world["step"], comms, app, poseidon_pp(0), poseidon_pp(1))
        for enc, bp in ((m.lib.ENC_BINPROT, True), (m.lib.ENC_BINCODE, False)):
            pub, der = gctx.pickles_public_input(wrap_proof_bytes(w, bp), enc, O.int_to_le(app))
            assert [O.le_to_int(x) for x in pub] == want, case
            assert [O.le_to_int(x) for x in der] == [dv[k] for k in ("combined_inner_product", "b", "zeta_to_srs_length", "zeta_to_domain_size", "perm", "xi", "r")]
    w["domain_log2"] = 9                                         # a step domain the installed index does not know
    with pytest.raises(m.MinaError):
        gctx.pickles_public_input(wrap_proof_bytes(w, True), m.lib.ENC_BINPROT, O.int_to_le(5))


def test_pickles_statements_on_the_gpu_match_oracle(world, srs_oracle):
    """the batch form (expand / digest / tick / scalar kernels of api_pickles.hip, no host arithmetic) == oracle/pickles_ref.py, for both
    sponge forms (16-lane up to 64 statements per call, 8-lane up to 1024, 3-lane above), 0..2 previous accumulators, optional + chunked evaluations; a
    malformed statement is flagged without disturbing its neighbours"""
    import mina_bridge_amd as m
    from ipa_helpers import poseidon_pp
    from kimchi_helpers import statements_soa
    from oracle import pickles_ref as PK
    from oracle import oracle as O
    from wire_writers import synth_wrap_proof
    rng = random.Random(321)
    ix = world["circ"].index
    comms = list(ix.sigma_comm) + list(ix.coefficients_comm) + list(ix.selector_comm)
    gctx = world["gctx"]
    for n_old, count in ((2, 5), (0, 3), (1, 4)):
        optional = [rng.randrange(3) == 0 for _ in range(19)]
        wraps, apps = [], []
        for i in range(count):
            w = synth_wrap_proof(rng, k=K_LOG2)
            w["prev_optional"] = [(([rng.randrange(PK.P)], [rng.randrange(PK.P)]) if o else None) for o in optional]        # one shape per call
            if i % 2:
                w["prev_evals"][7] = ([rng.randrange(PK.P), rng.randrange(PK.P)], [rng.randrange(PK.P), rng.randrange(PK.P)])
            w["step_comms"] = w["step_comms"][:n_old]
            w["step_old_chals"] = w["step_old_chals"][:n_old]
            wraps.append(w); apps.append(rng.randrange(PK.P))
        want = [PK.statement_public_input(w, world["step"], comms, a, poseidon_pp(0), poseidon_pp(1))[0] for w, a in zip(wraps, apps)]
        n_old_, n_evals, sec = statements_soa(wraps, apps)
        st = gctx.make_pickles_statements(n_old_, n_evals, sec)
        pub, ok = gctx.pickles_public_inputs_batch(st, count)
        assert ok.tolist() == [1] * count
        for b in range(count):
            assert [O.le_to_int(x) for x in pub[b]] == want[b], (n_old, b)
        # `count` statements ran in the 16-lane sponge form (up to 64 per call); the 8-lane form (up to 1024) and the 3-lane form (above):
        # the same statements tiled past either threshold
        for reps in (70 // count + 1, 1030 // count + 1):
            big = {k: np.tile(v.reshape(count, -1), (reps, 1)).reshape(-1) if v.size >= count else v for k, v in sec.items()}
            pub2, ok2 = gctx.pickles_public_inputs_batch(gctx.make_pickles_statements(n_old_, n_evals, big), count * reps)
            assert ok2.all() and (pub2.reshape(reps, count, 40, 32) == pub[None]).all()
        # malformed: a non-canonical evaluation, an unknown step domain, a bad branch byte -- each flags exactly its own statement
        bad = {k: v.copy() for k, v in sec.items()}
        bad["prev_evals"].reshape(count, -1)[1, :32] = 0xff
        bad["misc"].reshape(count, 32)[2, 0] = 9
        bad["misc"].reshape(count, 32)[0, 1] = 3
        pub3, ok3 = gctx.pickles_public_inputs_batch(gctx.make_pickles_statements(n_old_, n_evals, bad), count)
        assert ok3.tolist() == [0, 0, 0] + [1] * (count - 3)
        for b in range(3, count):
            assert (pub3[b] == pub[b]).all()


def test_c_consumer_many_threads_one_proof_per_call(world, srs_oracle, tmp_path):
    """plain C with pthreads, the operator's call pattern without an interpreter lock in the way: 16 threads x 6 calls of mina_verify_state, one
    proof per call, every fourth thread with a public input whose tip hash is wrong.  A fresh process has no verifier index, so the
    verdicts are taken with MINA_VERIFY_ALLOW_MISSING_KIMCHI (every other step runs).  Each caller must get the verdict of ITS proof
    although the calls are merged into shared jobs."""
    import mina_bridge_amd as m
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "threads.c"
    src.write_text(r'''
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "mina_verify.h"
static unsigned char *slurp(const char *p, size_t *n) { FILE *f = fopen(p, "rb"); if (!f) return 0; fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); rewind(f);
  unsigned char *b = malloc(*n + 1); if (fread(b, 1, *n, f) != *n) { fclose(f); return 0; } fclose(f); return b; }
static unsigned char *proof, *pub, *bad_pub; static size_t pl, ql; static int wrong[64];
enum { THREADS = 16, CALLS = 6 };
static void *worker(void *arg) { long t = (long)arg;
  for (int k = 0; k < CALLS; ++k) { const int bad = t % 4 == 3; bool v = mina_verify_state(proof, pl, bad ? bad_pub : pub, ql); if (v != !bad) wrong[t]++; }
  return 0; }
int main(int argc, char **argv) {
  if (argc < 3 || !(proof = slurp(argv[1], &pl)) || !(pub = slurp(argv[2], &ql))) return 2;
  bad_pub = malloc(ql); memcpy(bad_pub, pub, ql); bad_pub[1 + 7] ^= 1;             /* bridge tip state hash */
  mina_verify_configure(MINA_VERIFY_ALLOW_MISSING_KIMCHI | MINA_VERIFY_ALLOW_SURROGATE);
  if (!mina_verify_state(proof, pl, pub, ql) || mina_verify_state(proof, pl, bad_pub, ql)) return 3;     /* warm, and the lone-caller path */
  struct timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
  for (int k = 0; k < CALLS; ++k) if (!mina_verify_state(proof, pl, pub, ql)) return 4;
  clock_gettime(CLOCK_MONOTONIC, &b);
  const double one = ((b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6) / CALLS;
  pthread_t th[THREADS]; clock_gettime(CLOCK_MONOTONIC, &a);
  for (long t = 0; t < THREADS; ++t) pthread_create(&th[t], 0, worker, (void *)t);
  for (int t = 0; t < THREADS; ++t) pthread_join(th[t], 0);
  clock_gettime(CLOCK_MONOTONIC, &b);
  const double all = (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6;
  int w = 0; for (int t = 0; t < THREADS; ++t) w += wrong[t];
  printf("wrong=%d one_call_ms=%.2f threads=%d calls=%d all_ms=%.2f\n", w, one, THREADS, THREADS * CALLS, all);
  return 0; }
''')
    exe = tmp_path / "threads"
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-pthread", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L", os.path.dirname(m.LIB_PATH), "-lminaverify", "-Wl,-rpath," + os.path.dirname(m.LIB_PATH)])
    wrap, states, hashes = mint_state_proof(world, srs_oracle, 3100)
    proof, pub = to_bytes(wrap, states, hashes)
    (tmp_path / "p").write_bytes(proof); (tmp_path / "q").write_bytes(pub)
    r = subprocess.run([str(exe), str(tmp_path / "p"), str(tmp_path / "q")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    f = dict(kv.split("=") for kv in r.stdout.split())
    assert f["wrong"] == "0", r.stdout
    # 96 calls one after the other would take 96 x one_call_ms; merged they share a handful of jobs
    assert float(f["all_ms"]) < 0.5 * 96 * float(f["one_call_ms"]), r.stdout
    print(r.stdout.strip())


def test_boundary_pipeline_chunks_and_device_shards_give_the_same_verdicts(world, srs_oracle):
    """The bytes -> bools pipeline behind mina_verify_state_batch: the same 37 proofs (good ones, a tampered opening, a tampered public
    input, garbage) as ONE chunk, as chunks of 5 on one device, as chunks of 2 (19 chunks over the 16 slots of a device: slots are
    recycled), and cut into 3 contiguous shards over three contexts ($MINA_VERIFY_DEVICES=0,0,0: the multi-GPU path of SURVEY.md 8e.1
    on one GPU; ragged shards, bad proofs in every shard, chunked inside each shard) -- verdict bytes identical every time."""
    import mina_bridge_amd as m
    from kimchi_helpers import install_index
    minted = [mint_state_proof(world, srs_oracle, 4000 + i) for i in range(3)]
    good = [to_bytes(*x) for x in minted]
    w_bad = copy.deepcopy(minted[1][0]); w_bad["z2"] = (w_bad["z2"] + 1) % (1 << 254)
    bad_open = to_bytes(w_bad, minted[1][1], minted[1][2])
    bad_pub = bytearray(good[2][1]); bad_pub[77] ^= 4
    proofs, pubs, want = [], [], []
    for i in range(37):
        if i in (0, 12, 13, 36): p, q, v = bad_open[0], bad_open[1], 0
        elif i in (5, 25): p, q, v = good[2][0], bytes(bad_pub), 0
        elif i == 30: p, q, v = b"\x01\x02\x03", good[0][1], 0
        else: p, q, v = good[i % 3][0], good[i % 3][1], 1
        proofs.append(p); pubs.append(q); want.append(v)
    assert m.lib.verify_state_batch(proofs, pubs).tolist() == want
    T = m.lib.tuning
    keep_dev = os.environ.get("MINA_VERIFY_DEVICES")
    try:
        for chunk in (5, 2):
            with T(single_max=1, chunk=chunk):
                assert m.lib.verify_state_batch(proofs, pubs).tolist() == want, f"chunks of {chunk}"
        # the streamed form of a chunk (records uploaded and hashed run by run while the rest is parsed): runs of 3 -- the garbage entry at 30
        # ends the streaming, the rest goes up after the patching -- as one chunk, in chunks of 5 (runs of 2), and a call that streams to the end
        # (early_min = 1, head_min = 0: + the first run parsed whole and hashed ahead of everything else)
        for single, chunk, sub in ((8192, 8192, 3), (1, 5, 2), (8192, 8192, 1)):
            with T(early_min=1, head_min=0, single_max=single, chunk=chunk, early_sub=sub):
                assert m.lib.verify_state_batch(proofs, pubs).tolist() == want, f"streamed, runs of {sub}, chunks of {chunk}"
                assert m.lib.verify_state_batch(proofs[1:12], pubs[1:12]).tolist() == want[1:12], "streamed to the end"
                assert m.lib.verify_state_batch(proofs[14:25], pubs[14:25]).tolist() == want[14:25], "streamed to the end, all valid"
        with T(early_min=1, head_min=0, early_sub=0):
            assert m.lib.verify_state_batch(proofs, pubs).tolist() == want, "streaming off"
        # the window of chunks on the GPU at a time, and chunks parsed ahead of it (19 chunks of 2)
        for window, ahead in ((1, 0), (3, 2), (16, 4)):
            with T(early_min=1, head_min=0, single_max=1, chunk=2, early_sub=1, window=window, ahead=ahead, slots=16):
                assert m.lib.verify_state_batch(proofs, pubs).tolist() == want, f"window {window}, {ahead} ahead"
        # three logical devices on GPU 0
        m.lib.verify_shutdown()
        os.environ["MINA_VERIFY_DEVICES"] = "0,0,0"
        with T(early_min=1, head_min=0, early_sub=4, single_max=1, min_shard=1, chunk=4):
            assert m.lib.verify_device_count() == 3
            alld = m.lib.verify_all_devices()
            install_index(alld, world["circ"].index); install_step_index(alld, world["step"])
            assert m.lib.verify_state_batch(proofs, pubs).tolist() == want, "3 shards"
            assert m.lib.verify_state_batch(proofs[:2], pubs[:2]).tolist() == want[:2]
            assert m.lib.verify_state(*good[0]) is True and m.lib.verify_state(*bad_open) is False      # single calls are dealt round-robin over the devices
            assert m.lib.verify_state(*good[1]) is True and m.lib.verify_state(*good[2]) is True
    finally:
        if keep_dev is None: os.environ.pop("MINA_VERIFY_DEVICES", None)
        else: os.environ["MINA_VERIFY_DEVICES"] = keep_dev
        m.lib.verify_configure_ex(None)
        m.lib.verify_shutdown()
        gctx = m.lib.verify_global_ctx()
        install_index(gctx, world["circ"].index); install_step_index(gctx, world["step"])
        world["gctx"] = gctx


def test_proofs_of_different_evaluation_shapes_in_one_call(world, srs_oracle):
    """Step proofs may carry optional evaluations (43 + the ones present): a job has ONE shape, so well-formed proofs of another shape are
    verified in a job of their own (the `deferred` pass of run_device) instead of failing -- three shapes interleaved in one call, a tampered
    proof in two of them, and the same through merged single-proof calls."""
    import threading
    import mina_bridge_amd as m
    a = [to_bytes(*mint_state_proof(world, srs_oracle, 5000 + i)) for i in range(2)]                                    # 43 evaluations
    b = [to_bytes(*mint_state_proof(world, srs_oracle, 5100 + i, optional_slots=(2,))) for i in range(2)]                # 44
    c = [to_bytes(*mint_state_proof(world, srs_oracle, 5200, optional_slots=(2, 9, 17)))]                                # 46
    wb, sb, hb = mint_state_proof(world, srs_oracle, 5100, optional_slots=(2,))
    wb = dict(wb); wb["z1"] = (wb["z1"] + 1) % (1 << 254)
    bad_b = to_bytes(wb, sb, hb)
    bad_pub = bytearray(a[1][1]); bad_pub[100] ^= 2
    calls = [a[0], b[0], c[0], bad_b, a[1], b[1], (a[1][0], bytes(bad_pub)), c[0], a[0]]
    want = [1, 1, 1, 0, 1, 1, 0, 1, 1]
    assert m.lib.verify_state_batch([x[0] for x in calls], [x[1] for x in calls]).tolist() == want
    # the first proof decides the first job's shape: any order gives the same verdicts
    order = [2, 3, 0, 8, 6, 1, 5, 7, 4]
    assert m.lib.verify_state_batch([calls[i][0] for i in order], [calls[i][1] for i in order]).tolist() == [want[i] for i in order]
    got = [None] * len(calls)
    gate = threading.Barrier(len(calls))
    def worker(i):
        gate.wait(); got[i] = int(m.lib.verify_state(*calls[i]))
    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(calls))]
    for t in th: t.start()
    for t in th: t.join()
    assert got == want


def _feature_statement(rng, flags, extra_slots=(), joint=True, k=K_LOG2):
    """a random statement whose feature flags are `flags` and which carries exactly the optional evaluations the feature-aware program of
    kimchi_helpers.make_feature_step_index reads for them (+ `extra_slots`)"""
    from oracle import kimchi_ref as K, pickles_ref as PK
    from wire_writers import synth_wrap_proof
    w = synth_wrap_proof(rng, k=k)
    w["feature_flags"] = list(flags)
    fm = K.feature_mask(flags)
    need = set(extra_slots)
    if fm >> 6 & 1: need |= {6, 7}                  # LookupTables: lookup aggregation, lookup table
    if fm >> 0 & 1: need |= {0}                     # the range_check0 selector
    if (fm >> 13 & 1) and (fm >> 10 & 1): need |= {17}      # TableWidth(1) and LookupPattern RangeCheck: its selector
    w["prev_optional"] = [(([rng.randrange(PK.P)], [rng.randrange(PK.P)]) if j in need else None) for j in range(19)]
    w["joint_combiner"] = rng.getrandbits(128) if joint else None
    return w


def test_feature_aware_step_linearization_matches_oracle(world, srs_oracle):
    """kimchi's feature-flagged linearization on the GPU: SkipIf / SkipIfNot regions decided by every proof's OWN flags (LookupTables, a gate
    flag, a lookup pattern nested inside TableWidth), optional evaluations found through the proof's presence mask, the joint combiner, a value
    cached inside a region -- the statements' 40 public inputs == oracle/pickles_ref.py for every flag combination; a proof that switches a
    feature on without carrying the evaluation its term reads fails ALONE"""
    import mina_bridge_amd as m
    from ipa_helpers import poseidon_pp
    from kimchi_helpers import install_step_index, make_feature_step_index, statements_soa
    from oracle import oracle as O, pickles_ref as PK
    rng = random.Random(777)
    ix = world["circ"].index
    comms = list(ix.sigma_comm) + list(ix.coefficients_comm) + list(ix.selector_comm)
    gctx = world["gctx"]
    fstep = make_feature_step_index(99)
    install_step_index(gctx, fstep)
    try:
        combos = [[False] * 8, [True] + [False] * 7, [False] * 4 + [True] + [False] * 3, [False] * 6 + [True, False], [False, True, False, True, False, False, False, True],
                  [True] * 8, [False, False, True, False, False, True, False, False], [False] * 3 + [True] + [False] * 4]
        by_shape = {}
        for flags in combos:
            for extra in ((), (3, 12)):
                for joint in (True, False):
                    w = _feature_statement(rng, flags, extra, joint)
                    by_shape.setdefault(sum(e is not None for e in w["prev_optional"]), []).append(w)
        total = 0
        for n_opt, wraps in by_shape.items():
            apps = [rng.randrange(PK.P) for _ in wraps]
            want = [PK.statement_public_input(w, fstep, comms, a, poseidon_pp(0), poseidon_pp(1))[0] for w, a in zip(wraps, apps)]
            n_old, n_evals, sec = statements_soa(wraps, apps)
            assert n_evals == 43 + n_opt
            pub, ok = gctx.pickles_public_inputs_batch(gctx.make_pickles_statements(n_old, n_evals, sec), len(wraps))
            assert ok.tolist() == [1] * len(wraps)
            for b in range(len(wraps)):
                assert [O.le_to_int(x) for x in pub[b]] == want[b], (n_opt, b, wraps[b]["feature_flags"])
            total += len(wraps)
            # the single-proof host form agrees (its interpreter is polish.h's)
            from wire_writers import wrap_proof_bytes
            pub1, _ = gctx.pickles_public_input(wrap_proof_bytes(wraps[0], True), m.lib.ENC_BINPROT, O.int_to_le(apps[0]))
            assert [O.le_to_int(x) for x in pub1] == want[0]
        assert total == 32
        # a feature switched on without the evaluation its term reads: that statement alone is flagged (same count of optional evaluations, other slots)
        good = _feature_statement(rng, [False] * 6 + [True, False])                       # lookup: carries slots 6, 7
        bad = copy.deepcopy(good); bad["prev_optional"][7], bad["prev_optional"][9] = None, bad["prev_optional"][7]
        n_old, n_evals, sec = statements_soa([good, bad, good], [5, 6, 7])
        pub, ok = gctx.pickles_public_inputs_batch(gctx.make_pickles_statements(n_old, n_evals, sec), 3)
        assert ok.tolist() == [1, 0, 1]
        with pytest.raises(KeyError):
            PK.statement_public_input(bad, fstep, comms, 6, poseidon_pp(0), poseidon_pp(1))
        # a presence mask that does not match the number of evaluations carried: malformed
        sec2 = {k: v.copy() for k, v in sec.items()}; sec2["misc"].reshape(3, 32)[0, 11] ^= 1
        assert gctx.pickles_public_inputs_batch(gctx.make_pickles_statements(n_old, n_evals, sec2), 3)[1].tolist() == [0, 0, 1]
    finally:
        install_step_index(gctx, world["step"])


def test_boundary_accepts_lookup_features_only_with_a_feature_aware_step_index(world, srs_oracle):
    """mina_verify_state on proofs whose statement switches lookup features on: rejected at the kimchi step while the installed step
    linearization has no feature-dependent tokens (it could not evaluate them faithfully), verified -- and tamper-rejected -- once a
    feature-aware one is installed"""
    import mina_bridge_amd as m
    from kimchi_helpers import install_step_index, make_feature_step_index
    from oracle import kimchi_ref as K, oracle as O, pickles_ref as PK
    gctx = world["gctx"]
    fstep = make_feature_step_index(99)
    rng = random.Random(4242)
    plain_world = dict(world)
    feat_world = dict(world); feat_world["step"] = fstep

    def mint(w_, seed, flags, joint):
        # as mint_state_proof, with the statement's features chosen and the optional evaluations the feature-aware program reads
        st = _feature_statement(random.Random(seed), flags, joint=joint)
        wrap, states, hashes = mint_state_proof(w_, srs_oracle, seed, statement_overrides={k: st[k] for k in ("feature_flags", "prev_optional", "joint_combiner")})
        return to_bytes(wrap, states, hashes), wrap, states, hashes
    lookup_flags = [False, False, True, False, False, False, True, False]             # foreign_field_add + lookup
    rc_flags = [True, False, False, False, False, True, False, False]                 # range_check0 + rot
    install_step_index(gctx, fstep)
    try:
        a, wa, sa, ha = mint(feat_world, 6100, lookup_flags, True)
        b, _, _, _ = mint(feat_world, 6200, rc_flags, False)
        c, _, _, _ = mint(feat_world, 6300, [False] * 8, False)
        assert m.lib.verify_state(*a) is True and m.lib.verify_state(*b) is True and m.lib.verify_state(*c) is True
        assert m.lib.verify_state_checks(*a) == (ALL, ALL)
        wt = dict(wa); wt["joint_combiner"] = wa["joint_combiner"] ^ 1                 # the joint combiner enters ft_eval0 AND the packing
        assert m.lib.verify_state(*to_bytes(wt, sa, ha)) is False
        wt = dict(wa); po = list(wa["prev_optional"]); e = po[6]; po[6] = ([(e[0][0] + 1) % PK.P], e[1]); wt["prev_optional"] = po     # the lookup aggregation evaluation
        assert m.lib.verify_state(*to_bytes(wt, sa, ha)) is False
        assert m.lib.verify_state_batch([a[0], b[0], c[0], a[0]], [a[1], b[1], c[1], a[1]]).tolist() == [1, 1, 1, 1]      # three evaluation shapes in one call
        install_step_index(gctx, world["step"])                                        # the plain linearization: lookup features are refused, the others go on
        assert m.lib.verify_state(*a) is False and m.lib.verify_state(*b) is False
        passed, ran = m.lib.verify_state_checks(*a)
        assert ran & 32 and not passed & 32
    finally:
        install_step_index(gctx, world["step"])


def test_concurrent_batch_callers_share_the_pipeline(world, srs_oracle):
    """mina_verify_state_batch from several threads at once (a batcher's tasks): the callers share the device's slots and lanes, a call whose
    chunk fails its folded check runs its culprit search while other callers' chunks are in flight -- every caller gets exactly its verdicts"""
    import threading
    import mina_bridge_amd as m
    minted = [mint_state_proof(world, srs_oracle, 7000 + i) for i in range(3)]
    good = [to_bytes(*x) for x in minted]
    w_bad = copy.deepcopy(minted[0][0]); w_bad["z1"] = (w_bad["z1"] + 1) % (1 << 254)
    bad = to_bytes(w_bad, minted[0][1], minted[0][2])
    rng = random.Random(99)
    plans = []
    for t in range(6):
        calls = []
        for _ in range(4):
            n = rng.choice([1, 3, 9, 20])
            items = [(bad, 0) if rng.randrange(5) == 0 else (good[rng.randrange(3)], 1) for _ in range(n)]
            calls.append(items)
        plans.append(calls)
    # chunk = 4, single_max = 6: some calls in several chunks.  Twice: the callers' small batches merged into shared jobs (the default), and every call
    # through the pipeline on its own (merge = 0) -- then a chunk's culprit search really runs while other callers' chunks are in flight on the lanes it
    # uses (it drains the device first and holds its lock: api_verify.hip `fallback`)
    for merge in (1, 0):
        errors = []
        with m.lib.tuning(chunk=4, single_max=6, merge=merge):
            def worker(t):
                for items in plans[t]:
                    got = m.lib.verify_state_batch([x[0][0] for x in items], [x[0][1] for x in items]).tolist()
                    if got != [x[1] for x in items]:
                        errors.append((t, got, [x[1] for x in items]))
            th = [threading.Thread(target=worker, args=(t,)) for t in range(len(plans))]
            for t in th: t.start()
            for t in th: t.join()
        assert not errors, (merge, errors[:2])


def test_state_and_account_callers_at_once(world, srs_oracle):
    """Everything an operator process does at the same time: single state proofs, state batches (one with a tampered opening: its job runs a
    culprit search), single account proofs and account batches (tampered public inputs), from ten threads -- the host pool, the device lock,
    the two call mergers and the slots are shared; every caller gets exactly its own verdicts."""
    import struct
    import threading
    import mina_bridge_amd as m
    from ipa_helpers import poseidon_pp
    from oracle import mina_account_ref as A, pasta_ref as R
    minted = [mint_state_proof(world, srs_oracle, 7100 + i) for i in range(2)]
    good = [to_bytes(*x) for x in minted]
    w_bad = copy.deepcopy(minted[1][0]); w_bad["z1"] = (w_bad["z1"] + 3) % (1 << 254)
    bad = to_bytes(w_bad, minted[1][1], minted[1][2])
    pp = poseidon_pp(0)
    rng = random.Random(5)
    accounts = [A.synth_account(rng, zk, timed, deleg, with_vk=vk) for zk, timed, deleg, vk in [(False, False, False, True), (True, True, True, True), (True, False, True, False)]]
    aproofs, apubs = [], []
    for a in accounts:
        path = [(rng.randrange(2), rng.randrange(R.P)) for _ in range(35)]
        enc = A.abi_encode_account(a)
        aproofs.append(A.write_account_proof(path, a)); apubs.append(R.merkle_root(A.account_hash(a, pp), path, pp).to_bytes(32, "little") + struct.pack("<Q", len(enc)) + enc)
    errors = []

    def state_single(t):
        for k in range(5):
            is_bad = (t + k) % 3 == 0
            if m.lib.verify_state(*(bad if is_bad else good[k % 2])) is not (not is_bad): errors.append(("state single", t, k))

    def state_batch(t):
        for k in range(3):
            items = [(bad, 0) if (j == t + k) else (good[j % 2], 1) for j in range(9)]
            got = m.lib.verify_state_batch([x[0][0] for x in items], [x[0][1] for x in items]).tolist()
            if got != [x[1] for x in items]: errors.append(("state batch", t, k, got))

    def account_single(t):
        for k in range(8):
            i = (t + k) % 3; is_bad = k % 4 == 1
            if m.lib.verify_account(aproofs[i], apubs[(i + 1) % 3] if is_bad else apubs[i]) is not (not is_bad): errors.append(("account single", t, k))

    def account_batch(t):
        for k in range(4):
            P = [aproofs[j % 3] for j in range(10)]; Q = [apubs[j % 3] for j in range(10)]
            Q[(t + k) % 10] = apubs[((t + k) % 10 + 1) % 3]
            got = m.lib.verify_account_batch(P, Q).tolist()
            if got != [0 if j == (t + k) % 10 else 1 for j in range(10)]: errors.append(("account batch", t, k, got))

    th = [threading.Thread(target=f, args=(t,)) for t, f in enumerate([state_single, state_single, state_single, state_batch, state_batch, account_single, account_single, account_single, account_batch, account_batch])]
    for t in th: t.start()
    for t in th: t.join()
    assert not errors, errors[:3]
