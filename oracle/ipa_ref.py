"""IPA opening proof: prover + combined batch verifier, CPU restatement (TEST INFRASTRUCTURE ONLY).

Restates poly-commitment `SRS::open` and `SRS::verify` / `combine_commitments` / `combined_inner_product`
and mina-poseidon `DefaultFqSponge` (pins core/Cargo.toml:14,16; README.md:469-475, 534-544).  The reference
tree holds neither source nor vectors for these ("parity unpinned"): what pins this file is
self-consistency -- a proof minted by `ipa_open` must verify, any tampering must not -- plus the SRS KAT for
the group map used for the point U.

Heavy lifting (MSM, group map, b_poly) goes through the C oracle; scalar-field vector arithmetic is Python ints,
so keep k <= 10 in tests.
"""
from __future__ import annotations

import random

import numpy as np

from . import oracle as O
from . import pasta_ref as R


def shift_scalar(curve: int, x: int) -> int:
    """poly-commitment `shift_scalar::<G>`"""
    n1, n2 = R.scalar_modulus(curve), R.base_modulus(curve)
    two_pow = pow(2, 255, n1)
    if n1 < n2:
        return (x - (two_pow + 1)) * R.inv(2, n1) % n1
    return (x - two_pow) % n1


class FqSponge:
    """mina-poseidon `DefaultFqSponge<P, PlonkSpongeConstantsKimchi>` over the base field of `curve`."""

    def __init__(self, curve: int, pp: R.PoseidonParams, state=None, mode="absorbed", count=0):
        self.curve = curve
        self.sp = R.Sponge(pp)
        if state is not None:
            self.sp.state = list(state)
            self.sp.mode, self.sp.count = mode, count

    def clone(self):
        c = FqSponge(self.curve, self.sp.pp, self.sp.state, self.sp.mode, self.sp.count)
        return c

    def absorb_g(self, pts):
        for p in pts:
            if p is None:
                self.sp.absorb([0, 0])
            else:
                self.sp.absorb([p[0], p[1]])

    def absorb_fq(self, xs):
        self.sp.absorb(list(xs))

    def absorb_fr(self, xs):
        fr, fq = R.scalar_modulus(self.curve), R.base_modulus(self.curve)
        for x in xs:
            if fr < fq:
                self.sp.absorb([x])
            else:
                self.sp.absorb([x >> 1])
                self.sp.absorb([x & 1])

    def challenge_fq(self) -> int:
        return self.sp.squeeze()

    def challenge(self) -> int:
        return self.sp.squeeze() & ((1 << 128) - 1)

    def raw(self):
        """(state ints, mode 0/1, count) as handed to mina_ipa_opening"""
        return list(self.sp.state), (1 if self.sp.mode == "squeezed" else 0), self.sp.count


def _msm(curve, pts, scalars):
    """points: list of affine tuples/None; scalars: ints"""
    pb = np.stack([O.point_to_bytes(p) for p in pts])
    return O.bytes_to_point(O.msm_pippenger(curve, pb, O.ints_to_le(scalars), threads=4))


def commit(curve, g_bytes, h, coeffs, blinder):
    r = R.scalar_modulus(curve)
    m = R.base_modulus(curve)
    c = O.bytes_to_point(O.msm_pippenger(curve, g_bytes[: len(coeffs)], O.ints_to_le([x % r for x in coeffs]), threads=4))
    return R.add(c, R.scalar_mul(blinder % r, h, m), m)


def eval_poly(coeffs, x, r):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % r
    return acc


def combined_inner_product(evals, xi, rscale, r):
    """evals[i][j] = f_i(pt_j); single-chunk polynomials, no degree bounds"""
    res, xi_i = 0, 1
    for row in evals:
        term = 0
        for e in reversed(row):            # eval_polynomial(row, evalscale)
            term = (term * rscale + e) % r
        res = (res + xi_i * term) % r
        xi_i = xi_i * xi % r
    return res


def ipa_open(curve, g_bytes, h, polys, blinders, evalpoints, xi, rscale, sponge: FqSponge, rng: random.Random):
    """poly-commitment `SRS::open`.  g_bytes: [n,64] uint8 (n = 2^k); polys: list of coefficient lists (len n).
    Returns the opening-proof dict (points as tuples/None, scalars as ints)."""
    r, m = R.scalar_modulus(curve), R.base_modulus(curve)
    n = g_bytes.shape[0]
    k = n.bit_length() - 1
    assert 1 << k == n
    endo = R.endo_r(curve)
    bw = R.BWParams(m)
    g = [O.bytes_to_point(b) for b in g_bytes]

    a = [0] * n
    blind = 0
    xi_i = 1
    for f, bl in zip(polys, blinders):
        for j, c in enumerate(f):
            a[j] = (a[j] + xi_i * c) % r
        blind = (blind + xi_i * bl) % r
        xi_i = xi_i * xi % r
    b = [0] * n
    scale = 1
    for pt in evalpoints:
        pw = 1
        for j in range(n):
            b[j] = (b[j] + scale * pw) % r
            pw = pw * pt % r
        scale = scale * rscale % r
    cip = sum(x * y for x, y in zip(a, b)) % r

    sponge.absorb_fr([shift_scalar(curve, cip)])
    t = sponge.challenge_fq()
    u = bw.to_group(t)

    lr, chals, chal_invs = [], [], []
    r_prime = blind
    for _ in range(k):
        half = len(g) // 2
        g_lo, g_hi, a_lo, a_hi, b_lo, b_hi = g[:half], g[half:], a[:half], a[half:], b[:half], b[half:]
        rand_l, rand_r = rng.randrange(r), rng.randrange(r)
        ip_l = sum(x * y for x, y in zip(a_hi, b_lo)) % r
        ip_r = sum(x * y for x, y in zip(a_lo, b_hi)) % r
        L = _msm(curve, g_lo + [h, u], a_hi + [rand_l, ip_l])
        Rp = _msm(curve, g_hi + [h, u], a_lo + [rand_r, ip_r])
        lr.append((L, Rp))
        sponge.absorb_g([L])
        sponge.absorb_g([Rp])
        u_pre = sponge.challenge()
        ch = R.challenge_to_field(u_pre, endo, r)
        ch_inv = R.inv(ch, r)
        chals.append(ch)
        chal_invs.append(ch_inv)
        a = [(lo + ch_inv * hi) % r for lo, hi in zip(a_lo, a_hi)]
        b = [(lo + ch * hi) % r for lo, hi in zip(b_lo, b_hi)]
        chb = O.int_to_le(ch)
        g = [O.bytes_to_point(O.point_add(curve, O.point_to_bytes(lo), O.scalar_mul(curve, O.point_to_bytes(hi), chb)))
             for lo, hi in zip(g_lo, g_hi)]
        r_prime = (r_prime + rand_l * ch_inv + rand_r * ch) % r
    a0, b0, g0 = a[0], b[0], g[0]
    # cross-check: folding g_lo + u * g_hi through all rounds yields <b_poly_coefficients(chals), g>
    s_vec = O.b_poly_coefficients(O.scalar_field_of(curve), O.ints_to_le(chals))
    assert g0 == O.bytes_to_point(O.msm_pippenger(curve, g_bytes, s_vec, threads=4))
    d, r_delta = rng.randrange(r), rng.randrange(r)
    delta = R.add(R.scalar_mul(d, R.add(g0, R.scalar_mul(b0, u, m), m), m), R.scalar_mul(r_delta, h, m), m)
    sponge.absorb_g([delta])
    c = R.challenge_to_field(sponge.challenge(), endo, r)
    z1 = (a0 * c + d) % r
    z2 = (c * r_prime + r_delta) % r
    return {"lr": lr, "delta": delta, "z1": z1, "z2": z2, "sg": g0, "combined_inner_product": cip, "chals": chals}


def ipa_open_fast(curve, g_bytes, h, polys, blinders, evalpoints, xi, rscale, sponge: FqSponge, rng: random.Random):
    """Same proof as `ipa_open` (bit-identical for the same rng), without folding the bases point by point: the folded base
    g^(j)[i] = sum_S prod_{r in S} chal_r * g[i + offset(S)] is a fixed linear combination of the ORIGINAL bases, so every
    L_j / R_j is one MSM over g with scalars a[.] * W_j[top j bits of the index] (C oracle), and sg = <b_poly_coefficients, g>.
    Makes k = 15 openings a matter of seconds (fixtures for the full-size Proof-of-State job)."""
    r, m = R.scalar_modulus(curve), R.base_modulus(curve)
    n = g_bytes.shape[0]
    k = n.bit_length() - 1
    assert 1 << k == n
    endo = R.endo_r(curve)
    bw = R.BWParams(m)
    a = [0] * n
    blind = 0
    xi_i = 1
    for f, bl in zip(polys, blinders):
        if xi_i:
            for j, c in enumerate(f):
                if c:
                    a[j] = (a[j] + xi_i * c) % r
        blind = (blind + xi_i * bl) % r
        xi_i = xi_i * xi % r
    b = [0] * n
    scale = 1
    for pt in evalpoints:
        pw = 1
        for j in range(n):
            b[j] = (b[j] + scale * pw) % r
            pw = pw * pt % r
        scale = scale * rscale % r
    cip = sum(x * y for x, y in zip(a, b)) % r
    sponge.absorb_fr([shift_scalar(curve, cip)])
    u = bw.to_group(sponge.challenge_fq())
    hu = np.stack([O.point_to_bytes(h), O.point_to_bytes(u)])
    bases = np.concatenate([g_bytes, hu])

    def msm_full(sc, extra):
        return O.bytes_to_point(O.msm_pippenger(curve, bases, O.ints_to_le(sc + extra), threads=8))

    lr, chals = [], []
    r_prime = blind
    W = [1]                                             # W_j[top]: weight of the original bases whose top j index bits are `top`
    for j in range(k):
        nj = n >> j
        half = nj >> 1
        a_lo, a_hi, b_lo, b_hi = a[:half], a[half:], b[:half], b[half:]
        rand_l, rand_r = rng.randrange(r), rng.randrange(r)
        ip_l = sum(x * y for x, y in zip(a_hi, b_lo)) % r
        ip_r = sum(x * y for x, y in zip(a_lo, b_hi)) % r
        sl, sr = [0] * n, [0] * n
        for top, w in enumerate(W):
            base = top * nj
            for i in range(half):
                sl[base + i] = a_hi[i] * w % r          # L: the low half of every block of n_j original bases
                sr[base + half + i] = a_lo[i] * w % r   # R: the high half
        L = msm_full(sl, [rand_l, ip_l])
        Rp = msm_full(sr, [rand_r, ip_r])
        lr.append((L, Rp))
        sponge.absorb_g([L])
        sponge.absorb_g([Rp])
        ch = R.challenge_to_field(sponge.challenge(), endo, r)
        ch_inv = R.inv(ch, r)
        chals.append(ch)
        a = [(lo + ch_inv * hi) % r for lo, hi in zip(a_lo, a_hi)]
        b = [(lo + ch * hi) % r for lo, hi in zip(b_lo, b_hi)]
        W = [w * f % r for w in W for f in (1, ch)]     # next index bit set -> factor chal_j
        r_prime = (r_prime + rand_l * ch_inv + rand_r * ch) % r
    a0, b0 = a[0], b[0]
    s_vec = O.b_poly_coefficients(O.scalar_field_of(curve), O.ints_to_le(chals))
    g0 = O.bytes_to_point(O.msm_pippenger(curve, g_bytes, s_vec, threads=8))
    d, r_delta = rng.randrange(r), rng.randrange(r)
    delta = R.add(R.scalar_mul(d, R.add(g0, R.scalar_mul(b0, u, m), m), m), R.scalar_mul(r_delta, h, m), m)
    sponge.absorb_g([delta])
    c = R.challenge_to_field(sponge.challenge(), endo, r)
    z1 = (a0 * c + d) % r
    z2 = (c * r_prime + r_delta) % r
    return {"lr": lr, "delta": delta, "z1": z1, "z2": z2, "sg": g0, "combined_inner_product": cip, "chals": chals}


def ipa_verify_batch(curve, g_bytes, h, batch, rand_base, sg_rand_base, threads: int = 8) -> bool:
    """poly-commitment `SRS::verify`.  `batch`: list of dicts with keys
    sponge (FqSponge, consumed), evalpoints, polyscale, evalscale, comms (list of points), opening (dict from ipa_open
    or equivalent), combined_inner_product."""
    r, m = R.scalar_modulus(curve), R.base_modulus(curve)
    n = g_bytes.shape[0]
    endo = R.endo_r(curve)
    bw = R.BWParams(m)
    fs = O.scalar_field_of(curve)
    scalars_g = [0] * n
    scalar_h = 0
    pts, scs = [], []
    rho, sigma = 1, 1
    for e in batch:
        sp, op = e["sponge"], e["opening"]
        cip = e["combined_inner_product"]
        sp.absorb_fr([shift_scalar(curve, cip)])
        u = bw.to_group(sp.challenge_fq())
        chal = []
        for (L, Rp) in op["lr"]:
            sp.absorb_g([L])
            sp.absorb_g([Rp])
            chal.append(R.challenge_to_field(sp.challenge(), endo, r))
        chal_inv = [R.inv(x, r) for x in chal]
        sp.absorb_g([op["delta"]])
        c = R.challenge_to_field(sp.challenge(), endo, r)
        b0, scale = 0, 1
        for pt in e["evalpoints"]:
            b0 = (b0 + scale * R.b_poly(chal, pt, r)) % r
            scale = scale * e["evalscale"] % r
        s = [O.le_to_int(x) for x in O.b_poly_coefficients(fs, O.ints_to_le(chal))]
        neg_rho = (-rho) % r
        pts.append(op["sg"]); scs.append((neg_rho * op["z1"] - sigma) % r)
        for j in range(len(s)):
            scalars_g[j] = (scalars_g[j] + sigma * s[j]) % r
        scalar_h = (scalar_h - rho * op["z2"]) % r
        pts.append(u); scs.append(neg_rho * op["z1"] % r * b0 % r)
        rho_c = c * rho % r
        for (L, Rp), ui, uu in zip(op["lr"], chal_inv, chal):
            pts.append(L); scs.append(rho_c * ui % r)
            pts.append(Rp); scs.append(rho_c * uu % r)
        xi_i = 1
        for cm in e["comms"]:
            pts.append(cm); scs.append(rho_c * xi_i % r)
            xi_i = xi_i * e["polyscale"] % r
        pts.append(u); scs.append(rho_c * cip % r)
        pts.append(op["delta"]); scs.append(rho)
        rho = rho * rand_base % r
        sigma = sigma * sg_rand_base % r
    all_pts = np.concatenate([O.point_to_bytes(h)[None, :], g_bytes, np.stack([O.point_to_bytes(p) for p in pts])])
    all_scs = O.ints_to_le([scalar_h] + scalars_g + scs)
    return not O.msm_pippenger(curve, all_pts, all_scs, threads=threads).any()


def make_instance(curve, g_bytes, h, pp: R.PoseidonParams, k: int, n_polys: int, n_points: int, seed: int, xi=None):
    """A valid (commitments, evaluations, opening) instance over g[0..2^k) with a fresh transcript.
    Returns (verifier_entry_without_sponge, sponge_before) so that callers can clone the sponge."""
    rng = random.Random(seed)
    r = R.scalar_modulus(curve)
    n = 1 << k
    polys = [[rng.randrange(r) for _ in range(n)] for _ in range(n_polys)]
    blinders = [rng.randrange(r) for _ in range(n_polys)]
    comms = [commit(curve, g_bytes[:n], h, f, bl) for f, bl in zip(polys, blinders)]
    evalpoints = [rng.randrange(r) for _ in range(n_points)]
    xi_rand, rscale = rng.randrange(r), rng.randrange(r)
    xi = xi_rand if xi is None else xi          # tests force polyscale = 0 to get commitments with zero weight
    evals = [[eval_poly(f, pt, r) for pt in evalpoints] for f in polys]
    sponge = FqSponge(curve, pp)
    sponge.absorb_g(comms)                      # some transcript prefix, as kimchi's oracles would leave it
    sponge.challenge()
    before = sponge.clone()
    op = ipa_open(curve, g_bytes[:n], h, polys, blinders, evalpoints, xi, rscale, sponge, rng)
    assert op["combined_inner_product"] == combined_inner_product(evals, xi, rscale, r)
    entry = {"evalpoints": evalpoints, "polyscale": xi, "evalscale": rscale, "comms": comms, "opening": op,
             "combined_inner_product": op["combined_inner_product"], "k": k}
    return entry, before
