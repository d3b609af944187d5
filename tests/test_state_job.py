"""The Proof-of-State job (BASELINE config C3; README.md:281-310) through the C-ABI vs the CPU oracle composite
(oracle/state_job_ref.py): every intermediate bit-exact (state hashes, body hashes, public-input commitment), the verdict,
and tamper rejection per stage with culprit isolation."""
import base64
import copy
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _le(x):
    return np.frombuffer(int(x).to_bytes(32, "little"), np.uint8)


def test_state_hash_parity_reference_state_and_random(ctx, oracle):
    """MinaHash(ProtocolState) on the GPU == oracle, for the reference's own serialized state (constants.rs:22) and random ones;
    both lane-cooperative forms (8 lanes for small batches, 4 lanes above 8192 states)"""
    import mina_bridge_amd as m
    from oracle import mina_state_ref as S, state_job_ref as J
    from state_job_helpers import pp_fp, state_records
    pp = pp_fp()
    fx = json.load(open(os.path.join(HERE, "golden", "tip_protocol_state.json")))
    raw = base64.b64decode(fx["protocol_state_base64"])
    rng = random.Random(5)
    states = [S.parse_protocol_state(raw)] + [J.synth_state(rng, rng.randrange(S.P), i) for i in range(20)]
    states[3]["body"]["consensus_state"]["sub_window_densities"] = []           # ragged: fewer packed chunks
    states[4]["body"]["consensus_state"]["sub_window_densities"] = [7] * 16
    blobs = [S.write_protocol_state(s) for s in states]
    got = ctx.protocol_state_hash_bytes(blobs)
    for s, g in zip(states, got):
        assert oracle.le_to_int(g) == S.protocol_state_hash(s, pp)
    recs, nf = state_records(states)
    h, body = ctx.protocol_state_hash_batch(recs, nf, want_body=True)
    assert (h == got).all()
    for s, b in zip(states, body):
        assert oracle.le_to_int(b) == S.protocol_state_body_hash(s["body"], pp)
    # n_body_fields = 0 and 1 (degenerate records) against the oracle's sponge
    r2 = np.zeros((2, 64 * 32), np.uint8); r2[0, :32] = _le(5); r2[1, :32] = _le(6); r2[1, 32:64] = _le(9)
    h2 = ctx.protocol_state_hash_batch(r2, np.array([0, 1], np.uint32))
    assert oracle.le_to_int(h2[0]) == S.hash_with_kimchi(S.PREFIX_PROTOCOL_STATE, [5, S.hash_with_kimchi(S.PREFIX_PROTOCOL_STATE_BODY, [], pp)], pp)
    assert oracle.le_to_int(h2[1]) == S.hash_with_kimchi(S.PREFIX_PROTOCOL_STATE, [6, S.hash_with_kimchi(S.PREFIX_PROTOCOL_STATE_BODY, [9], pp)], pp)
    # 4-lane form: 8200 records (the 21 above, tiled)
    reps = 8200 // len(states) + 1
    big = ctx.protocol_state_hash_batch(np.tile(recs, (reps, 1)), np.tile(nf, reps))
    assert (big.reshape(reps, len(states), 32) == got[None]).all()


SMALL = dict(k=7, log2_domain=7, npub=8, n_comms=6, slot=2, n_points=2, acc_k=8)


@pytest.fixture(scope="module")
def small_jobs(srs_oracle):
    from state_job_helpers import mint_job
    return [mint_job(srs_oracle[0], srs_oracle[1], 100 + 10 * i, **SMALL) for i in range(5)]


def _run(ctx, m, jobs, **kw):
    from state_job_helpers import build_jobs
    sj = build_jobs(m, jobs, SMALL["k"], SMALL["log2_domain"], SMALL["slot"], SMALL["acc_k"], **kw)
    return ctx.state_job_batch(sj)


def test_state_job_accepts_and_matches_oracle_composite(ctx_srs, oracle, srs_oracle, small_jobs):
    import mina_bridge_amd as m
    from oracle import state_job_ref as J
    from state_job_helpers import oracle_job, pp_fp
    for B in (1, 5):
        v = _run(ctx_srs, m, small_jobs[:B])
        assert v.tolist() == [1] * B
    # oracle composite on the same jobs: intermediates bit-exact
    for job in small_jobs[:2]:
        ref = J.verify_state_job(pp_fp(), srs_oracle[0], srs_oracle[1], oracle_job(job))
        assert ref["verdict"] and ref["chain_ok"] and ref["ipa_ok"] and ref["acc_ok"]
        got_h = ctx_srs.protocol_state_hash_batch(job["records"], job["nfields"])
        assert [oracle.le_to_int(x) for x in got_h] == ref["hashes"]
        pc = ctx_srs.public_input_commitment(0, SMALL["log2_domain"], oracle.ints_to_le(job["pubs"]))
        assert oracle.bytes_to_point(pc) == ref["public_comm"] == job["entry"]["comms"][SMALL["slot"]]
    # each leg alone
    assert _run(ctx_srs, m, small_jobs[:3], with_ipa=False, with_acc=False).tolist() == [1, 1, 1]
    assert _run(ctx_srs, m, small_jobs[:3], with_states=False, with_acc=False).tolist() == [1, 1, 1]
    assert _run(ctx_srs, m, small_jobs[:3], with_states=False, with_ipa=False).tolist() == [1, 1, 1]


def test_state_job_tamper_each_stage(ctx_srs, oracle, srs_oracle, small_jobs):
    """one bad proof per stage inside a batch of 5: only the culprit is rejected, and the oracle composite agrees on why"""
    import mina_bridge_amd as m
    from oracle import state_job_ref as J
    from state_job_helpers import entry_arrays, oracle_job, pp_fp, state_records
    jobs = [copy.deepcopy(j) for j in small_jobs]
    # job 0: a flipped bit in one state's body (the hash no longer matches the public input)
    jobs[0]["states"][7]["body"]["consensus_state"]["total_currency"] ^= 1
    jobs[0]["records"], jobs[0]["nfields"] = state_records(jobs[0]["states"])
    # job 1: states hash to the public inputs but do not link (state 5 re-parented, its expected hash recomputed)
    from oracle import mina_state_ref as S
    jobs[1]["states"][5]["previous_state_hash"] = 12345
    jobs[1]["records"], jobs[1]["nfields"] = state_records(jobs[1]["states"])
    jobs[1]["expected"][5] = S.protocol_state_hash(jobs[1]["states"][5], pp_fp())
    # job 2: one public input changed -> different public-input commitment -> the opening no longer verifies
    jobs[2]["pubs"][3] = (jobs[2]["pubs"][3] + 1) % (1 << 200)
    # job 3: accumulator commitment replaced by another valid point
    jobs[3]["acc_sg"] = small_jobs[4]["acc_sg"].copy()
    v = _run(ctx_srs, m, jobs)
    assert v.tolist() == [0, 0, 0, 0, 1]
    refs = [J.verify_state_job(pp_fp(), srs_oracle[0], srs_oracle[1], oracle_job(j)) for j in jobs]
    assert [r["verdict"] for r in refs] == [False, False, False, False, True]
    assert (refs[0]["chain_ok"], refs[1]["chain_ok"], refs[2]["ipa_ok"], refs[3]["acc_ok"]) == (False, False, False, False)
    assert refs[2]["chain_ok"] and refs[2]["acc_ok"] and refs[3]["ipa_ok"]
    # tampered opening scalar (z1) in the middle of the batch, malformed point (off-curve delta) at the end
    jobs = [copy.deepcopy(j) for j in small_jobs]
    jobs[2]["entry"]["opening"]["z1"] = (jobs[2]["entry"]["opening"]["z1"] + 1) % (1 << 250)
    jobs[2]["abi"] = entry_arrays(jobs[2]["entry"], jobs[2]["sponge"])
    jobs[4]["abi"] = dict(jobs[4]["abi"]); d = jobs[4]["abi"]["delta"].copy(); d[0] ^= 1; jobs[4]["abi"]["delta"] = d
    assert _run(ctx_srs, m, jobs).tolist() == [1, 1, 0, 1, 0]
    # the bridge tip hash (state 16) participates in the comparison
    jobs = [copy.deepcopy(j) for j in small_jobs[:2]]
    jobs[1]["expected"][16] ^= 1
    assert _run(ctx_srs, m, jobs).tolist() == [1, 0]


def test_state_job_device_resident_and_pipelined(ctx_srs, oracle, small_jobs):
    """`_dev` entry: inputs in HBM, several jobs in flight over pipeline lanes, verdict words + flags read back at the end"""
    import mina_bridge_amd as m
    from state_job_helpers import build_jobs
    sj = build_jobs(m, small_jobs, SMALL["k"], SMALL["log2_domain"], SMALL["slot"], SMALL["acc_k"])
    ctx_srs.state_jobs_prepare(SMALL["log2_domain"], SMALL["npub"])
    d, ptrs = ctx_srs.state_jobs_to_device(sj)
    B = len(small_jobs)
    ctx_srs.set_pipeline(4)
    try:
        outs = [ctx_srs.dev_malloc(4 * B + 16) for _ in range(6)]
        for o in outs:
            ctx_srs.state_job_batch_dev(d, o, o + 4 * B)
        ctx_srs.synchronize()
        for o in outs:
            w = ctx_srs.dev_download(o, 4 * B + 16).view(np.uint32)
            assert w[:B].tolist() == [1] * B and w[B:].tolist() == [1, 0, 1, 0]
    finally:
        ctx_srs.set_pipeline(1)
        for p in outs + ptrs:
            ctx_srs.dev_free(p)


@pytest.mark.parametrize("tune", [dict(dev_fork=0), dict(dev_fork=1), dict(dev_fork=1, dev_piece_waves=1, coop16_max=0, coop8_max=0), dict(dev_fork=1, dev_hash_lds_kb=33, coop16_max=0, coop8_max=0),
                                  dict(dev_fork=3, dev_chain_cus=96), dict(dev_fork=5), dict(dev_fork=1, dev_acc_lane=1), dict(dev_fork=1, dev_acc_lane=2)])
def test_state_job_dev_legs_forked_in_every_tuning(oracle, small_jobs, tune):
    """round 6: the three legs of a device-resident job on streams of their own (mina_verify_tuning.dev_fork: plain / CU-masked / priority streams, the hashes in pieces --
    coop*_max = 0 forces the wave-packed 3-lane form, the only one launched in pieces -- or with an LDS reservation; the accumulator leg on a stream of its own or on the
    hashes' stream, behind or ahead of them) give the verdict words of the one-stream job, lane by
    lane, with 1 and with 4 jobs in flight: all good; a state whose hash no longer matches fails ITS proof only; a changed public input fails the folded opening (flag 0);
    another proof's accumulator commitment fails the folded accumulator check (flag 2).  A context of its own per tuning: streams keep their mask / priority for life."""
    import mina_bridge_amd as m
    from state_job_helpers import build_jobs, state_records
    B = len(small_jobs)
    variants = {"good": small_jobs}
    j = [copy.deepcopy(x) for x in small_jobs]
    j[0]["states"][7]["body"]["consensus_state"]["total_currency"] ^= 1
    j[0]["records"], j[0]["nfields"] = state_records(j[0]["states"])
    variants["bad_state"] = j
    j = [copy.deepcopy(x) for x in small_jobs]; j[2]["pubs"][3] = (j[2]["pubs"][3] + 1) % (1 << 200)
    variants["bad_opening"] = j
    j = [copy.deepcopy(x) for x in small_jobs]; j[3]["acc_sg"] = small_jobs[4]["acc_sg"].copy()
    variants["bad_accumulator"] = j
    expect = {"good": ([1] * B, [1, 0, 1, 0]), "bad_state": ([0] + [1] * (B - 1), [1, 0, 1, 0]), "bad_opening": ([0] * B, [0, 0, 1, 0]), "bad_accumulator": ([0] * B, [1, 0, 0, 0])}
    with m.lib.tuning(**tune):
        c = m.MinaContext(0)
        try:
            for f in (0, 1):
                c.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
            c.srs_create(0, 1 << 10); c.srs_create(1, 1 << 10)          # as smoke(): depth >= 2^k, 2^acc_k
            c.state_jobs_prepare(SMALL["log2_domain"], SMALL["npub"])
            dev = {name: c.state_jobs_to_device(build_jobs(m, jobs, SMALL["k"], SMALL["log2_domain"], SMALL["slot"], SMALL["acc_k"])) for name, jobs in variants.items()}
            for lanes in (1, 4, 10):                                                  # 10: more lanes than a forked pipeline has (8) -- every job on one stream again
                c.set_pipeline(lanes)
                calls = [name for _ in range(3) for name in variants]                  # 12 calls back to back: with 4 lanes, 4 jobs of different kinds in flight
                outs = [c.dev_malloc(4 * B + 16) for _ in calls]
                for name, o in zip(calls, outs):
                    c.state_job_batch_dev(dev[name][0], o, o + 4 * B)
                c.synchronize()
                for name, o in zip(calls, outs):
                    w = c.dev_download(o, 4 * B + 16).view(np.uint32)
                    assert (w[:B].tolist(), w[B:].tolist()) == expect[name], (tune, lanes, name, w.tolist())
                    c.dev_free(o)
            c.set_pipeline(1)
            for d, ptrs in dev.values():
                for p in ptrs: c.dev_free(p)
        finally:
            c.close()


def test_full_size_forked_job_rejects_what_the_one_stream_job_rejects(ctx_srs):
    """the bench's own job at full size (17 hashes, statement -> 40 public inputs, kimchi, k = 15 opening with the matrix-core fold, 2^16 accumulator; 512 proofs per call: the
    wave-packed 3-lane forms) through `mina_state_job_batch_dev`, forked over 4 lanes and on one stream: the same verdict words and flags for a clean batch, for one whose proof 7 has a
    flipped bit in a protocol state (its verdict alone), one with a flipped opening scalar (folded opening flag), one with a flipped accumulator prechallenge (folded accumulator flag)."""
    torch = pytest.importorskip("torch")
    import bench
    import mina_bridge_amd as m
    ctx, B = ctx_srs, 512
    dev = torch.device("cuda", 0)
    variants = {}
    for name in ("good", "bad_state", "bad_opening", "bad_accumulator"):
        (hj, keep), kp, _, _ = bench.build_full_job(ctx, m, B, seed=4100)
        by_addr = {a.ctypes.data: a for a in keep if isinstance(a, np.ndarray)}
        if name == "bad_state": by_addr[hj.state_records].view(np.uint8).reshape(B, 17, 64, 32)[7, 9, 20, 3] ^= 4
        if name == "bad_opening": by_addr[hj.z1].view(np.uint8).reshape(B, 32)[300, 0] ^= 1
        if name == "bad_accumulator": by_addr[hj.acc_prechallenges].view(np.uint8).reshape(B, 16, 16)[41, 3, 0] ^= 1
        variants[name] = bench.device_jobs(m, hj, keep, kp, dev)
    expect = {"good": ([1] * B, [1, 0, 1, 0]), "bad_state": ([1] * 7 + [0] + [1] * (B - 8), [1, 0, 1, 0]), "bad_opening": ([0] * B, [0, 0, 1, 0]), "bad_accumulator": ([0] * B, [1, 0, 0, 0])}
    ctx.state_jobs_prepare(bench.LOG2_DOMAIN, bench.NPUB)
    try:
        for tune, lanes in ((dict(dev_fork=1), 4), (dict(dev_fork=0), 1), (dict(dev_fork=1), 1)):
            with m.lib.tuning(**tune):
                ctx.set_pipeline(lanes)
                order = [n for _ in range(2) for n in variants]
                outs = [torch.full((B + 4,), 9, dtype=torch.int32, device=dev) for _ in order]
                torch.cuda.synchronize()
                for n, o in zip(order, outs):
                    ctx.state_job_batch_dev(variants[n][0], o.data_ptr(), o.data_ptr() + 4 * B)
                ctx.synchronize()
                for n, o in zip(order, outs):
                    w = o.cpu().numpy().tolist()
                    assert (w[:B], w[B:]) == expect[n], (tune, lanes, n, [i for i, (a, e) in enumerate(zip(w[:B], expect[n][0])) if a != e][:8], w[B:])
    finally:
        ctx.synchronize(); ctx.set_pipeline(1)


def test_state_job_full_size_c3(ctx_srs, oracle, srs_oracle):
    """BASELINE config C3 at full size: 17 states, 40 public inputs over the 2^15 wrap domain, k = 15 opening with 45 commitments
    and 2 points (committed oracle-minted fixture), 2^16 Vesta accumulator; B = 16 with one tampered proof"""
    import mina_bridge_amd as m
    from oracle import state_job_ref as J
    from state_job_helpers import build_jobs, load_k15_openings, mint_job, oracle_job, pp_fp
    fx, ops = load_k15_openings()
    shape = dict(k=fx["k"], log2_domain=fx["log2_domain"], npub=fx["npub"], n_comms=fx["n_comms"], slot=fx["slot"], n_points=fx["n_points"], acc_k=16)
    jobs = [mint_job(srs_oracle[0], srs_oracle[1], 500 + i, opening=ops[i % len(ops)], **shape) for i in range(16)]
    sj = build_jobs(m, jobs, shape["k"], shape["log2_domain"], shape["slot"], 16)
    assert ctx_srs.state_job_batch(sj).tolist() == [1] * 16
    ref = J.verify_state_job(pp_fp(), srs_oracle[0], srs_oracle[1], oracle_job(jobs[3]))
    assert ref["verdict"]
    pc = ctx_srs.public_input_commitment(0, 15, oracle.ints_to_le(jobs[3]["pubs"]))
    assert oracle.bytes_to_point(pc) == ref["public_comm"]
    jobs[9] = copy.deepcopy(jobs[9]); jobs[9]["pubs"][39] ^= 1
    sj = build_jobs(m, jobs, shape["k"], shape["log2_domain"], shape["slot"], 16)
    assert ctx_srs.state_job_batch(sj).tolist() == [1] * 9 + [0] + [1] * 6
    assert not J.verify_state_job(pp_fp(), srs_oracle[0], srs_oracle[1], oracle_job(jobs[9]))["ipa_ok"]


def test_state_hash_extreme_field_values_in_every_lane_form(ctx, oracle):
    """the permutation's own arithmetic (9 limbs of 29 bits, no carries, no conditional subtractions: fp29.cuh) on the values that stress
    it -- 0, 1, p - 1, p - 2, 2^254 - 1 (all limbs full), 2^29 - 1, 2^29, single-limb boundaries -- as state fields: the 16-lane form
    (100 states), the 8-lane form (2000) and the wave-packed 3-lane form (8200 states) all equal the CPU oracle's sponge"""
    import mina_bridge_amd as m
    import mina_bridge_amd.poseidon_params as PP
    from oracle import mina_state_ref as S, pasta_ref as R
    from state_job_helpers import pp_fp
    P = R.P
    ext = [0, 1, P - 1, P - 2, (1 << 254) - 1, (1 << 29) - 1, 1 << 29, (1 << 58) - 1, 1 << 232, (1 << 232) - 1, (1 << 253) + 12345, P >> 1]
    rng = np.random.Generator(np.random.PCG64(77))
    nrec, nbody, slots = 12, 49, 64
    recs = np.zeros((nrec, slots, 32), np.uint8)
    vals = [[ext[(r * 7 + j * 5) % len(ext)] if (j + r) % 3 else int(rng.integers(0, 1 << 62)) * ext[4] % P for j in range(nbody + 1)] for r in range(nrec)]
    for r in range(nrec):
        for j in range(nbody + 1):
            recs[r, j] = oracle.int_to_le(vals[r][j])
    nf = np.full(nrec, nbody, np.uint32)
    # expected: H_"MinaProtoState"(previous, H_"MinaProtoStateBody"(body fields)) with the oracle's permutation
    pp = pp_fp(); params = PP.default_params_bytes(0)
    salts = [S.salt(S.PREFIX_PROTOCOL_STATE_BODY, pp), S.salt(S.PREFIX_PROTOCOL_STATE, pp)]
    perm = lambda st: [int.from_bytes(x.tobytes(), "little") for x in oracle.poseidon_permute(0, params, oracle.ints_to_le(st).reshape(1, 96)).reshape(3, 32)]
    want = []
    for r in range(nrec):
        st = list(salts[0])
        for blk in range(0, nbody, 2):
            for t in range(2):
                if blk + t < nbody:
                    st[t] = (st[t] + vals[r][1 + blk + t]) % P
            st = perm(st)
        st = perm([(salts[1][0] + vals[r][0]) % P, (salts[1][1] + st[0]) % P, salts[1][2]])
        want.append(st[0])
    for n in (100, 2000, 8200):
        idx = np.arange(n) % nrec
        got = ctx.protocol_state_hash_batch(recs[idx].reshape(n, -1).copy(), nf[idx].copy())
        assert [oracle.le_to_int(x) for x in got[:nrec]] == want, n
        assert (got == got[idx % nrec][: n]).all() and (got[nrec:2 * nrec] == got[:nrec]).all(), n


@pytest.mark.gpu
def test_state_hash_every_field_count_in_every_lane_form(ctx, oracle):
    """records with 0, 1, 2, 3, ... 63 body fields (ragged sponges: an empty body, one lonely field in the last block, a full record) and 256-bit words that are NOT
    canonical field elements (>= p: the absorbing product reduces them) through the 16-lane, the 8-lane and the wave-packed 3-lane form -- the last absorbs in the 29-bit
    form (a lane with nothing to absorb multiplies zero) -- all equal the CPU oracle's sponge; body hashes as well"""
    import mina_bridge_amd.poseidon_params as PP
    from oracle import mina_state_ref as S, pasta_ref as R
    from state_job_helpers import pp_fp
    P = R.P
    rng = random.Random(4242)
    counts = list(range(0, 64)) + [49, 49, 48, 1, 0]
    nrec, slots = len(counts), 64
    recs = np.zeros((nrec, slots, 32), np.uint8)
    vals = [[rng.randrange(P) for _ in range(slots)] for _ in range(nrec)]
    vals[5][3] = P + 12345; vals[9][0] = (1 << 256) - 1; vals[17][17] = P; vals[63][63] = (1 << 255) + 7      # non-canonical words
    for r in range(nrec):
        for j in range(slots):
            recs[r, j] = np.frombuffer(vals[r][j].to_bytes(32, "little"), np.uint8)
    nf = np.array(counts, np.uint32)
    pp = pp_fp(); params = PP.default_params_bytes(0)
    salts = [S.salt(S.PREFIX_PROTOCOL_STATE_BODY, pp), S.salt(S.PREFIX_PROTOCOL_STATE, pp)]
    perm = lambda st: [int.from_bytes(x.tobytes(), "little") for x in oracle.poseidon_permute(0, params, oracle.ints_to_le(st).reshape(1, 96)).reshape(3, 32)]
    want, want_body = [], []
    for r in range(nrec):
        st = list(salts[0]); nbody = min(counts[r], slots - 1)
        for blk in range(0, nbody, 2):
            if blk: st = perm(st)
            for t in range(2):
                if blk + t < nbody:
                    st[t] = (st[t] + vals[r][1 + blk + t]) % P
        st = perm(st)
        want_body.append(st[0])
        st = perm([(salts[1][0] + vals[r][0]) % P, (salts[1][1] + st[0]) % P, salts[1][2]])
        want.append(st[0])
    for n in (nrec, 2000, 8200):
        idx = np.arange(n) % nrec
        got, body = ctx.protocol_state_hash_batch(recs[idx].reshape(n, -1).copy(), nf[idx].copy(), want_body=True)
        assert [oracle.le_to_int(x) for x in got[:nrec]] == want, n
        assert [oracle.le_to_int(x) for x in body[:nrec]] == want_body, n
        assert (got == got[idx % nrec][: n]).all(), n


def test_many_distinct_wrap_proofs_in_one_job():
    """round 5 (VERDICT r04 weak #5 / next #2c): bench.py's headline batch is tiled from 256 DISTINCT complete wrap proofs (the 4 of statement_k15_encoded.json + the 252 of
    statement_k15_many.npz, all minted by the repo's CPU prover) and one distinct chain per proof.  Here the 256 go through ONE job (`mina_state_job_batch_dev`, the headline's
    entry point): every verdict ACCEPT, both folded checks pass; then a flipped bit in the z1 of proof 200 fails the folded opening check of the device job, and the
    host-buffer form (`mina_state_job_batch`: culprit search) names exactly that proof.  (The CPU oracle accepts a sample of the same proofs: tests/test_statement_fixture.py.)"""
    import ctypes
    import sys
    import torch
    import mina_bridge_amd as m
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    B = 256
    c = m.MinaContext(0)
    try:
        for f in (0, 1):
            c.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
        c.srs_create(1, 1 << 16); c.srs_create(0, 1 << 16)
        (hj, keep), kp, _, distinct = bench.build_full_job(c, m, B, seed=77)
        assert distinct == {"chains": B, "wrap_proofs": B}
        dev = torch.device("cuda", 0)
        dj, dk, tensors = bench.device_jobs(m, hj, keep, kp, dev)
        c.state_jobs_prepare(bench.LOG2_DOMAIN, bench.NPUB)
        out = torch.zeros(B + 4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        c.state_job_batch_dev(dj, out.data_ptr(), out.data_ptr() + 4 * B); c.synchronize()
        assert out.cpu().numpy().tolist() == [1] * B + [1, 0, 1, 0]
        assert c.state_job_batch((hj, keep)).tolist() == [1] * B
        by_addr = {a.ctypes.data: a for a in keep if isinstance(a, np.ndarray)}
        by_addr[hj.z1].view(np.uint8).reshape(B, 32)[200, 0] ^= 1
        dj2, dk2, tensors2 = bench.device_jobs(m, hj, keep, kp, dev)
        out.zero_(); torch.cuda.synchronize()
        c.state_job_batch_dev(dj2, out.data_ptr(), out.data_ptr() + 4 * B); c.synchronize()
        w = out.cpu().numpy()
        assert w[B] == 0 and w[B + 2] == 1 and not w[:B].any(), "a bad opening fails the job's folded opening check (every verdict 0 until the culprit search)"
        assert c.state_job_batch((hj, keep)).tolist() == [1] * 200 + [0] + [1] * 55
    finally:
        c.close()
