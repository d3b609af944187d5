"""Error behaviour at the boundary: bad arguments come back as negative return codes with a message (`MinaError` through the
ctypes wrapper), never as a crash or a silent wrong answer; the context stays usable afterwards."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_bad_arguments_are_errors_not_crashes(ctx_srs):
    import mina_bridge_amd as m
    c = ctx_srs
    lib = c._lib
    z32, z64 = np.zeros(32, np.uint8), np.zeros(64, np.uint8)
    pre = np.zeros(16 * 16, np.uint8)
    with pytest.raises(m.MinaError):
        c.msm_srs(2, z32)                                            # bad curve
    with pytest.raises(m.MinaError):
        c.msm_srs_multi(1, np.zeros((65537, 32), np.uint8), 1)       # n beyond the SRS depth
    with pytest.raises(m.MinaError):
        c.accumulator_check_batch(1, 21, np.zeros(21 * 16, np.uint8), z64)     # k > 20
    with pytest.raises(m.MinaError):
        c.accumulator_check_batch(1, 17, np.zeros(17 * 16, np.uint8), z64)     # 2^k beyond the SRS depth
    with pytest.raises(m.MinaError):
        c.public_input_commitment_batch(0, 17, np.zeros((1, 32), np.uint8), 1)  # domain beyond the SRS depth
    with pytest.raises(m.MinaError):
        c.public_input_commitment_batch(0, 3, np.zeros((9, 32), np.uint8), 1)   # more public inputs than the domain
    with pytest.raises(m.MinaError):
        c.set_pipeline(0)
    with pytest.raises(m.MinaError):
        c.set_pipeline(33)
    d = c.dev_malloc(64)
    try:
        with pytest.raises(m.MinaError):
            c.accumulator_check_multi_dev(1, 16, 0, d, d, d)          # empty group
        with pytest.raises(m.MinaError):
            c.accumulator_check_multi_dev(1, 16, 65, d, d, d)         # group larger than 64
        with pytest.raises(m.MinaError):
            c.accumulator_check_multi_dev(1, 16, 1, 0, d, d)          # null device pointer
    finally:
        c.dev_free(d)
    # raw C calls with null pointers
    assert lib.mina_msm_srs(c._h, 1, ctypes.c_size_t(4), None, None) < 0
    assert lib.mina_dev_malloc(c._h, ctypes.c_size_t(16), None) < 0
    assert lib.mina_accumulator_check_multi(c._h, 1, ctypes.c_uint32(16), ctypes.c_size_t(1), None, None, None) < 0
    assert b"null" in lib.mina_last_error()
    # and the context still computes
    one = np.zeros(32, np.uint8); one[0] = 1
    assert (c.msm_srs(1, one) == c.srs_get_g(1, 0, 1)[0]).all()
    assert c.accumulator_check_multi(1, 16, pre, z64).tolist() == [0]


@pytest.mark.gpu
def test_scalars_with_bit_255_set_are_rejected_and_small_srs_roundtrips(ctx_srs, oracle, srs_oracle):
    """the signed-digit recoding has no window for a carry out of bit 255 -> such scalars are an argument error, not a
    silently wrong point; mina_srs_load / serialize use the minimal MessagePack array header (fixarray / array16 / array32)"""
    import mina_bridge_amd as m
    g, _ = srs_oracle[1]
    sc = np.zeros((4, 32), np.uint8); sc[2, 31] = 0x80
    for call in (lambda: ctx_srs.msm_srs(1, sc), lambda: ctx_srs.msm(1, g[:4], sc), lambda: ctx_srs.msm_srs_multi(1, sc, 2), lambda: ctx_srs.msm_srs_range(1, 0, sc),
                 lambda: ctx_srs.public_input_commitment(0, 5, sc)):
        with pytest.raises(m.MinaError, match="2\\^255"):
            call()
    sc[2, 31] = 0x7f                                            # non-canonical but below 2^255: still computed (digits are exact)
    ctx_srs.msm_srs(1, sc)
    with pytest.raises(m.MinaError):                            # k is validated before anything is sized from it
        ctx_srs.accumulator_check_batch(1, 0, np.zeros(0, np.uint8), np.zeros(64, np.uint8))
    with pytest.raises(m.MinaError):
        ctx_srs.accumulator_check_multi(1, 40, np.zeros(40 * 16, np.uint8), np.zeros(64, np.uint8))
    c2 = m.MinaContext(0)
    try:
        for depth, hdr in ((8, 2), (32, 4)):
            c2.srs_create(1, depth)
            blob = c2.srs_serialize(1)
            assert len(blob) == hdr + (depth + 1) * 35 and blob[0] == 0x92 and blob[1] == ((0x90 | depth) if depth < 16 else 0xdc)
            pts = c2.srs_get_g(1, 0, depth)
            c2.srs_load(1, blob)
            assert (c2.srs_get_g(1, 0, depth) == pts).all() and (pts == g[:depth]).all()
    finally:
        c2.close()


@pytest.mark.gpu
def test_statement_and_kimchi_sections_reject_bad_shapes(oracle):
    """round-2 entry points: misuse is an error code, never a crash or a silent wrong answer -- the statement stage without a step index,
    impossible shapes, null sections, a kimchi section whose statements do not derive exactly 40 public inputs"""
    import ctypes
    import mina_bridge_amd as m
    from kimchi_helpers import install_index, install_step_index, load_k15_fixture, make_step_index, statements_soa
    from wire_writers import synth_wrap_proof
    import random
    rng = random.Random(2)
    c = m.MinaContext(0)
    try:
        for f in (0, 1):
            c.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
        c.srs_create(0, 1 << 15)
        w = synth_wrap_proof(rng, k=15); w["prev_optional"] = [None] * 19
        n_old, n_evals, sec = statements_soa([w], [5])
        st = c.make_pickles_statements(n_old, n_evals, sec)
        with pytest.raises(m.MinaError, match="index"):          # no step index / no wrap index yet
            c.pickles_public_inputs_batch(st, 1)
        ix, _, _ = load_k15_fixture()
        install_index(c, ix)
        with pytest.raises(m.MinaError, match="step index"):
            c.pickles_public_inputs_batch(st, 1)
        install_step_index(c, make_step_index(99))
        pub, ok = c.pickles_public_inputs_batch(st, 1)
        assert ok.tolist() == [1]
        for n_old_bad, n_evals_bad in ((5, n_evals), (n_old, 42), (n_old, 63)):
            with pytest.raises(m.MinaError):
                c.pickles_public_inputs_batch(c.make_pickles_statements(n_old_bad, n_evals_bad, sec), 1)
        missing = dict(sec); missing.pop("prev_evals")
        with pytest.raises(m.MinaError, match="null"):
            c.pickles_public_inputs_batch(c.make_pickles_statements(n_old, n_evals, missing), 1)
        with pytest.raises(m.MinaError):
            c.pickles_public_inputs_batch(st, 0)
        # a job whose kimchi section carries statements must ask for exactly 40 public inputs
        z = np.zeros(64 * 64, np.uint8)
        karr = {"prev_prechallenges": z, "prev_comms": z, "w_comm": z, "z_comm": z, "t_comm": z, "evals": z, "ft_eval1": z}
        kp = c.make_kimchi_proofs(1, 2, 39, karr, statements=st)
        ja = {"lr": z, "delta": z, "sg": z, "z1": z, "z2": z, "rand_base": oracle.int_to_le(7), "sg_rand_base": oracle.int_to_le(9)}
        with pytest.raises(m.MinaError):
            c.state_job_batch(c.make_state_jobs(1, ja, with_ipa=1, kimchi=kp, k=15, log2_domain=15, npub=39, n_evalpoints=2, n_comms=47))
        # a step index with more than 8 domains, or a malformed constant term, is refused
        with pytest.raises(m.MinaError):
            c.step_index_install(3, list(range(5, 14)), np.zeros(9 * 7 * 32, np.uint8), b"")
        with pytest.raises(m.MinaError):
            c.step_index_install(3, [10], np.zeros(7 * 32, np.uint8), bytes([250]))
        # and the context still works
        pub2, ok2 = c.pickles_public_inputs_batch(st, 1)
        assert (pub2 == pub).all() and ok2.tolist() == [1]
    finally:
        c.close()
