"""Sizes above BASELINE's 2^16: an SRS of depth 2^18 (the README's figure for the step circuit, README.md:485-486) in its own
context -- regeneration, window tables (256 MiB), the 2^18-base MSM, b_poly_coefficients with k = 18 and the accumulator
check at k = 18, all against the CPU oracle."""
import os

import numpy as np
import pytest

from conftest import rand_scalars

pytestmark = pytest.mark.gpu
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001


@pytest.fixture(scope="module")
def big(oracle):
    import mina_bridge_amd as m
    curve, k = 1, 18
    c = m.MinaContext(0)
    c.srs_create(curve, 1 << k)
    g, h = oracle.srs_create(curve, 1 << k, threads=os.cpu_count() or 8)
    yield c, curve, k, g, h
    c.close()


def test_srs_depth_2_18_matches_oracle(big):
    c, curve, k, g, h = big
    assert c.srs_depth(curve) == 1 << k
    for first in (0, 65536, (1 << k) - 4096):
        assert (c.srs_get_g(curve, first, 4096) == g[first:first + 4096]).all(), first
    assert (c.srs_get_h(curve) == h).all()


def test_msm_2_18_bases(big, oracle):
    c, curve, k, g, h = big
    n = 1 << k
    import mina_bridge_amd as m
    sc = rand_scalars(n, P, seed=181)
    want = oracle.msm_pippenger(curve, g, sc, threads=16)
    c.srs_split_table(curve)
    for fp29 in (1, 2, 3, 0):                                    # 4 M table points: the size that caught the one-p negation of round 4 (one table y in 2^21 reaches 2^254 - 2^233)
        with m.lib.tuning(msm_fp29=fp29):
            assert (c.msm_srs(curve, sc) == want).all(), fp29
    # a slice in the upper half of the table and the variable-base path on the same points
    assert (c.msm_srs_range(curve, 200000, sc[:50000]) == oracle.msm_pippenger(curve, g[200000:250000], sc[:50000], threads=16)).all()
    assert (c.msm(curve, g[100000:230000], sc[:130000]) == oracle.msm_pippenger(curve, g[100000:230000], sc[:130000], threads=16)).all()


def test_accumulator_check_k18(big, oracle):
    c, curve, k, g, h = big
    fs = 0
    _, endo_r = oracle.endo(curve)
    pre = rand_scalars(k, P, seed=182, bits=128)[:, :16].copy()
    chals = np.stack([oracle.challenge_to_field(fs, pre[i].copy(), endo_r) for i in range(k)])
    s = oracle.b_poly_coefficients(fs, chals)
    assert (c.b_poly_coefficients(fs, chals) == s).all()
    sg = oracle.msm_pippenger(curve, g, s, threads=16)
    import mina_bridge_amd as m
    c.srs_split_table(curve)
    for fp29 in (1, 2, 3):
        with m.lib.tuning(msm_fp29=fp29):
            assert c.accumulator_check_batch(curve, k, pre, sg).tolist() == [1], fp29
            assert c.accumulator_check_multi(curve, k, np.concatenate([pre, pre]), np.stack([sg, g[7]])).tolist() == [1, 0], fp29
