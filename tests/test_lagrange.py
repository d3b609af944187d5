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


def test_public_input_commitment(ctx_srs, oracle, srs_oracle):
    """kimchi: public_comm = h - sum_i pub_i * lagrange_i  == h - commit(interpolant of pub on the domain), checked through the
    oracle by expanding the Lagrange basis: sum_i pub_i L_i = MSM(g, c) with c_j = (1/n) sum_i pub_i w^(-ij)"""
    from conftest import rand_scalars
    from oracle import pasta_ref as R
    curve, k, npub = 0, 6, 40
    g, h = srs_oracle[curve]
    r = R.scalar_modulus(curve); n = 1 << k
    pub = rand_scalars(npub, r, seed=77)
    got = ctx_srs.public_input_commitment(curve, k, pub)
    w = pow(R.two_adic_root_of_unity(r), 1 << (32 - k), r); w_inv, n_inv = pow(w, r - 2, r), pow(n, r - 2, r)
    pubs = [oracle.le_to_int(x) for x in pub]
    coeffs = [n_inv * sum(pubs[i] * pow(w_inv, (i * j) % n, r) for i in range(npub)) % r for j in range(n)]
    msm = oracle.bytes_to_point(oracle.msm_pippenger(curve, g[:n], oracle.ints_to_le(coeffs), threads=4))
    m = R.base_modulus(curve)
    exp = R.add(oracle.bytes_to_point(h), R.neg(msm, m), m)
    assert oracle.bytes_to_point(got) == exp
    # no public inputs -> h itself; cached basis is reused (second call) and survives a different npub
    assert (ctx_srs.public_input_commitment(curve, k, np.zeros((0, 32), np.uint8)) == h).all()
    assert (ctx_srs.public_input_commitment(curve, k, pub) == got).all()


@pytest.mark.parametrize("curve,k,npub,batch", [(0, 6, 40, 5), (1, 5, 32, 3), (0, 15, 40, 17), (0, 7, 1, 2), (0, 8, 100, 4), (1, 7, 64, 3), (0, 7, 65, 3)])
def test_public_input_commitment_batch(ctx_srs, oracle, srs_oracle, curve, k, npub, batch):
    """batched form (<= 64 inputs: straight from the digit table; more: one fixed-base problem per proof over the Lagrange window table) == the single-proof entry point
    (variable-base MSM), which test_public_input_commitment pins to the oracle; plus a direct oracle check of row 0"""
    from conftest import rand_scalars
    from oracle import pasta_ref as R
    g, h = srs_oracle[curve]
    r = R.scalar_modulus(curve)
    pub = rand_scalars(batch * npub, r, seed=1000 + 7 * k + npub).reshape(batch, npub, 32)
    pub[1, :, :] = 0                                            # an all-zero public input -> h
    if batch > 2:
        pub[2, :, :] = 0; pub[2, npub - 1, 0] = 1               # a single unit scalar -> h - L_{npub-1}
    got = ctx_srs.public_input_commitment_batch(curve, k, pub, batch)
    assert got.shape == (batch, 64)
    for m in range(batch):
        assert (got[m] == ctx_srs.public_input_commitment(curve, k, pub[m])).all(), m
    assert (got[1] == h).all()
    basis = ctx_srs.srs_lagrange_basis(curve, k)
    mod = R.base_modulus(curve)
    a0 = oracle.bytes_to_point(oracle.msm_naive(curve, basis[:npub], pub[0]))
    assert oracle.bytes_to_point(got[0]) == R.add(oracle.bytes_to_point(h), R.neg(a0, mod), mod)
    # table is reused for a smaller npub and rebuilt for a larger one
    small = ctx_srs.public_input_commitment_batch(curve, k, pub[:, :1, :].copy(), batch)
    assert (small[0] == ctx_srs.public_input_commitment(curve, k, pub[0, :1])).all()
    # empty public input and empty batch
    assert (ctx_srs.public_input_commitment_batch(curve, k, np.zeros((0, 32), np.uint8), 3) == np.broadcast_to(h, (3, 64))).all()
    assert ctx_srs.public_input_commitment_batch(curve, k, np.zeros((0, 32), np.uint8), 0).shape == (0, 64)


def test_public_input_commitment_digit_table_edge_scalars(ctx_srs, oracle, srs_oracle):
    """commitments of <= 64 inputs come straight from the table of digit multiples d * 2^(8w) * L_i (lagrange.cuh): scalars whose bytes sit
    on the signed-digit boundaries (0x80, 0x7f, 0xff runs that carry through every window, r - 1, powers of two, zero) in the 64-lane form
    (small batch) and the 8-lane form (> 1024 proofs), against the oracle's naive sum; the tiled rows must repeat"""
    from oracle import pasta_ref as R
    curve, k, npub = 0, 6, 40
    g, h = srs_oracle[curve]
    r = R.scalar_modulus(curve); mod = R.base_modulus(curve)
    pats = [0, 1, r - 1, r - 2, (1 << 254) - 1, int.from_bytes(b"\x80" * 31 + b"\x00", "little"), int.from_bytes(b"\x7f" * 31 + b"\x3f", "little"),
            int.from_bytes(b"\xff" * 31 + b"\x1f", "little"), int.from_bytes(b"\x81\x7f" * 15 + b"\x80\x00", "little"), 1 << 248, (1 << 248) - 1, 128, 129, 255, 256, (1 << 128) - 1]
    rng = np.random.Generator(np.random.PCG64(5))
    rows = 6
    vals = [[pats[(m * 7 + i * 3) % len(pats)] if (m + i) % 4 else int(rng.integers(0, 1 << 62)) * pats[4] % r for i in range(npub)] for m in range(rows)]
    pub = np.stack([oracle.ints_to_le(v) for v in vals])
    basis = ctx_srs.srs_lagrange_basis(curve, k)
    want = []
    for m in range(rows):
        a = oracle.bytes_to_point(oracle.msm_naive(curve, basis[:npub], pub[m]))
        want.append(R.add(oracle.bytes_to_point(h), R.neg(a, mod), mod))
    got = ctx_srs.public_input_commitment_batch(curve, k, pub, rows)
    assert [oracle.bytes_to_point(x) for x in got] == want
    reps = 1100 // rows + 1
    big = np.tile(pub, (reps, 1, 1))
    got2 = ctx_srs.public_input_commitment_batch(curve, k, big, rows * reps)
    assert (got2.reshape(reps, rows, 64) == got[None]).all()


def test_combined_inner_product_matches_restatement(oracle):
    import mina_bridge_amd as m
    from conftest import rand_scalars
    from oracle import ipa_ref as I, pasta_ref as R
    for field in (0, 1):
        r = R.P if field == 0 else R.Q
        n_polys, n_points = 7, 2
        ev = rand_scalars(n_polys * n_points, r, seed=5 + field)
        xi, rs = rand_scalars(2, r, seed=9 + field)
        got = m.combined_inner_product(field, ev, xi, rs, n_polys, n_points)
        evals = [[oracle.le_to_int(ev[i * n_points + j]) for j in range(n_points)] for i in range(n_polys)]
        assert oracle.le_to_int(got) == I.combined_inner_product(evals, oracle.le_to_int(xi), oracle.le_to_int(rs), r)
