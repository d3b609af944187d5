"""Lagrange-basis commitments of the SRS (poly-commitment `SRS::add_lagrange_basis`, used for kimchi's public-input
commitment): GPU group-iFFT vs the defining sum L_i = (1/n) sum_j w^(-ij) g_j evaluated by the CPU oracle's MSM."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def expected_basis_point(oracle, R, curve, g, k, i):
    r = R.scalar_modulus(curve)
    n = 1 << k
    w = pow(R.two_adic_root_of_unity(r), 1 << (32 - k), r)
    assert pow(w, n, r) == 1 and (n == 1 or pow(w, n // 2, r) != 1)
    w_inv, n_inv = pow(w, r - 2, r), pow(n, r - 2, r)
    sc = [pow(w_inv, (i * j) % n, r) * n_inv % r for j in range(n)]
    return oracle.msm_pippenger(curve, g[:n], oracle.ints_to_le(sc), threads=8)


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("k", [0, 1, 2, 5, 8])
def test_lagrange_basis_small_full(ctx_srs, oracle, srs_oracle, curve, k):
    from oracle import pasta_ref as R
    g, _ = srs_oracle[curve]
    got = ctx_srs.srs_lagrange_basis(curve, k)
    n = 1 << k
    idx = range(n) if n <= 32 else [0, 1, 2, n // 2, n - 1]
    for i in idx:
        assert (got[i] == expected_basis_point(oracle, R, curve, g, k, i)).all(), (k, i)
    # property: sum_i L_i = g_0  (the Lagrange polynomials sum to the constant 1)
    acc = np.zeros(64, np.uint8)
    for i in range(n):
        acc = oracle.point_add(curve, acc, got[i])
    assert (acc == g[0]).all()


def test_lagrange_basis_wrap_domain(ctx_srs, oracle, srs_oracle):
    """full size of the Pickles wrap domain (2^15 on Pallas): spot-check against the defining sums + a random public-input
    commitment: MSM(lagrange[0..40], pub) == MSM(g, coefficients of the interpolant) is implied by the per-index checks"""
    from oracle import pasta_ref as R
    curve, k = 0, 15
    g, _ = srs_oracle[curve]
    got = ctx_srs.srs_lagrange_basis(curve, k)
    for i in (0, 1, 39, 12345, (1 << k) - 1):
        assert (got[i] == expected_basis_point(oracle, R, curve, g, k, i)).all(), i
    assert all(oracle.is_on_curve(curve, p) for p in got[:50])
