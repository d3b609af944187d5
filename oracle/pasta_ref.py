"""Python big-int twin of the CPU oracle (TEST INFRASTRUCTURE ONLY).

This file is part of `oracle/`: it may be imported only by `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg.  It is never on
the product path (the product is `mina_bridge_amd/csrc/*.hip` behind
`include/mina_verify.h`).

PARITY STATUS: the reference tree (/root/reference) contains *no* verifier and
no tests for this path.  What this restatement is
pinned against:
  * srs/vesta.srs and srs/pallas.srs (2 x 65 537 points): BLAKE2b-512 ->
    bit packing -> BW19 group map -> ark-serialize 0.3 compressed point codec
    -> MessagePack container, reproduced byte-for-byte (sha256 match).
  * everything else (Pippenger window rule, b_poly, IPA equation, Poseidon
    sponge, endo challenges) restates the *published* algorithms of the
    un-vendored crates pinned in core/Cargo.toml:14-25 -- "parity unpinned".

Pure-Python loops: use only for small cases; the C oracle (oracle/pasta_oracle.c)
is the one used at size.
"""
from __future__ import annotations

import hashlib
import struct

# --------------------------------------------------------------------------
# Fields (mina-curves pasta; core/Cargo.toml:17).  Values verified against the
# in-tree SRS points.
# --------------------------------------------------------------------------
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001  # Pallas base / Vesta scalar ("Fp")
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001  # Vesta base / Pallas scalar ("Fq")
TWO_ADICITY = 32
GENERATOR = 5  # multiplicative generator (and quadratic non-residue) of both fields
CURVE_B = 5    # y^2 = x^3 + 5 for both curves

FIELD_FP, FIELD_FQ = 0, 1
CURVE_PALLAS, CURVE_VESTA = 0, 1  # Pallas: base Fp, scalar Fq.  Vesta: base Fq, scalar Fp.


def base_modulus(curve: int) -> int:
    return P if curve == CURVE_PALLAS else Q


def scalar_modulus(curve: int) -> int:
    return Q if curve == CURVE_PALLAS else P


def inv(a: int, m: int) -> int:
    return pow(a, m - 2, m)


def legendre_is_square(a: int, m: int) -> bool:
    a %= m
    return a == 0 or pow(a, (m - 1) // 2, m) == 1


def two_adic_root_of_unity(m: int) -> int:
    t = (m - 1) >> TWO_ADICITY
    return pow(GENERATOR, t, m)


def sqrt_ark(a: int, m: int):
    """ark-ff 0.3 `SquareRootField::sqrt` (Tonelli-Shanks, eprint 2012/685 alg. 5).

    Deterministic: the returned root depends on TWO_ADIC_ROOT_OF_UNITY = 5^t.
    Which of {y,-y} comes out is pinned by the flag bits of the SRS files
    (tests/test_srs_kat.py).
    """
    a %= m
    if a == 0:
        return 0
    if pow(a, (m - 1) // 2, m) != 1:
        return None
    t = (m - 1) >> TWO_ADICITY
    z = pow(GENERATOR, t, m)
    w = pow(a, (t - 1) // 2, m)
    x = w * a % m
    b = x * w % m
    v = TWO_ADICITY
    while b != 1:
        k = 0
        b2k = b
        while b2k != 1:
            b2k = b2k * b2k % m
            k += 1
        j = v - k - 1
        w = z
        for _ in range(j):
            w = w * w % m
        z = w * w % m
        b = b * z % m
        x = x * w % m
        v = k
    return x


# --------------------------------------------------------------------------
# Curve arithmetic (affine tuples, None = infinity).
# --------------------------------------------------------------------------
def is_on_curve(pt, m):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - CURVE_B) % m == 0


def neg(pt, m):
    return None if pt is None else (pt[0], (-pt[1]) % m)


def add(p1, p2, m):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % m == 0:
            return None
        lam = 3 * x1 * x1 * inv(2 * y1, m) % m
    else:
        lam = (y2 - y1) * inv(x2 - x1, m) % m
    x3 = (lam * lam - x1 - x2) % m
    y3 = (lam * (x1 - x3) - y1) % m
    return (x3, y3)


def scalar_mul(k: int, pt, m):
    acc = None
    while k:
        if k & 1:
            acc = add(acc, pt, m)
        pt = add(pt, pt, m)
        k >>= 1
    return acc


def msm_naive(points, scalars, m):
    acc = None
    for p_, k in zip(points, scalars):
        acc = add(acc, scalar_mul(k, p_, m), m)
    return acc


# --------------------------------------------------------------------------
# groupmap crate `BWParameters` (a14)
# --------------------------------------------------------------------------
class BWParams:
    """`BWParameters::setup` for y^2 = x^3 + 5 with u = 1 (first u with f(u) != 0)."""

    def __init__(self, m: int):
        self.m = m
        u = 1
        fu = (u * u * u + CURVE_B) % m  # 6
        self.u = u
        self.fu = fu
        three_u2 = 3 * u * u % m
        self.sqrt_neg_three_u_squared = sqrt_ark((-three_u2) % m, m)
        self.sqrt_neg_three_u_squared_minus_u_over_2 = (
            (self.sqrt_neg_three_u_squared - u) * inv(2, m) % m
        )
        self.inv_three_u_squared = inv(three_u2, m)

    def potential_xs(self, t: int):
        m = self.m
        t2 = t * t % m
        alpha_inv = (t2 + self.fu) * t2 % m
        alpha = inv(alpha_inv, m) if alpha_inv else 0
        t4 = t2 * t2 % m
        x1 = (self.sqrt_neg_three_u_squared_minus_u_over_2
              - t4 * alpha % m * self.sqrt_neg_three_u_squared) % m
        x2 = (-self.u - x1) % m
        t2_plus_fu = (t2 + self.fu) % m
        x3 = (self.u - t2_plus_fu * t2_plus_fu % m * alpha % m * t2_plus_fu % m
              * self.inv_three_u_squared) % m
        return x1, x2, x3

    def to_group(self, t: int):
        m = self.m
        for x in self.potential_xs(t):
            y = sqrt_ark((x * x * x + CURVE_B) % m, m)
            if y is not None:
                return (x, y)
        raise ArithmeticError("group map failed")


# --------------------------------------------------------------------------
# poly-commitment `SRS::create` (a6)
# --------------------------------------------------------------------------
def _field_from_digest(digest: bytes, m: int) -> int:
    # bits[8i+j] = (digest[i] >> j) & 1 for i < 31 ; value = from_bits_be(bits)
    v = 0
    for i in range(31):
        for j in range(8):
            v = (v << 1) | ((digest[i] >> j) & 1)
    return v % m


def srs_point(curve: int, i: int, bw: BWParams | None = None):
    m = base_modulus(curve)
    bw = bw or BWParams(m)
    d = hashlib.blake2b(struct.pack(">I", i), digest_size=64).digest()
    return bw.to_group(_field_from_digest(d, m))


def srs_h(curve: int, bw: BWParams | None = None):
    m = base_modulus(curve)
    bw = bw or BWParams(m)
    d = hashlib.blake2b(b"srs_misc" + struct.pack(">I", 0), digest_size=64).digest()
    return bw.to_group(_field_from_digest(d, m))


# --------------------------------------------------------------------------
# ark-serialize 0.3 compressed short-Weierstrass point (33 bytes) + msgpack SRS
# --------------------------------------------------------------------------
def point_compress(pt, m) -> bytes:
    if pt is None:
        return bytes(32) + b"\x40"
    x, y = pt
    flag = 0x80 if y > (m - y) else 0x00
    return x.to_bytes(32, "little") + bytes([flag])


def point_decompress(blob: bytes, m):
    assert len(blob) == 33
    flag = blob[32]
    if flag & 0x40:
        return None
    x = int.from_bytes(blob[:32], "little")
    y = sqrt_ark((x * x * x + CURVE_B) % m, m)
    if y is None:
        raise ValueError("x not on curve")
    neg_y = (m - y) % m
    lo, hi = (y, neg_y) if y < neg_y else (neg_y, y)
    return (x, hi if (flag & 0x80) else lo)


def srs_parse(data: bytes):
    """MessagePack fixarray(2)[ array32(n)[bin8(33)...], bin8(33) ] -> (list[blob], blob)."""
    assert data[0] == 0x92 and data[1] == 0xDD
    n = struct.unpack(">I", data[2:6])[0]
    off = 6
    g = []
    for _ in range(n):
        assert data[off] == 0xC4 and data[off + 1] == 33
        g.append(data[off + 2: off + 35])
        off += 35
    assert data[off] == 0xC4 and data[off + 1] == 33
    h = data[off + 2: off + 35]
    assert off + 35 == len(data)
    return g, h


def srs_serialize(g_blobs, h_blob) -> bytes:
    out = bytearray(b"\x92\xdd" + struct.pack(">I", len(g_blobs)))
    for b in g_blobs:
        out += b"\xc4\x21" + b
    out += b"\xc4\x21" + h_blob
    return bytes(out)


# --------------------------------------------------------------------------
# IPA challenge polynomial (a9) -- poly-commitment `b_poly`, `b_poly_coefficients`
# --------------------------------------------------------------------------
def b_poly(chals, x, m):
    k = len(chals)
    pow_twos = [x % m]
    for _ in range(1, k):
        pow_twos.append(pow_twos[-1] * pow_twos[-1] % m)
    r = 1
    for i in range(k):
        r = r * (1 + chals[i] * pow_twos[k - 1 - i]) % m
    return r


def b_poly_coefficients(chals, m):
    k = len(chals)
    s = [1] * (1 << k)
    kk, pw = 0, 1
    for i in range(1, 1 << k):
        if i == pw << 1:
            kk += 1
            pw <<= 1
        s[i] = s[i - pw] * chals[k - 1 - kk] % m
    return s


# --------------------------------------------------------------------------
# kimchi `ScalarChallenge::to_field` (a13).  endo_r values: SURVEY appendix A.
# --------------------------------------------------------------------------
def cube_root_of_unity(m: int) -> int:
    return pow(GENERATOR, (m - 1) // 3, m)


def endo_r(curve: int) -> int:
    """Scalar-field endo coefficient: Pallas -> omega_Fq^2 ; Vesta -> omega_Fp^2."""
    m = scalar_modulus(curve)
    w = cube_root_of_unity(m)
    return w * w % m


def endo_q(curve: int) -> int:
    return cube_root_of_unity(base_modulus(curve))


def challenge_to_field(chal128: int, endo: int, m: int) -> int:
    a, b = 2, 2
    for i in range(63, -1, -1):
        a = 2 * a % m
        b = 2 * b % m
        r0 = (chal128 >> (2 * i)) & 1
        r1 = (chal128 >> (2 * i + 1)) & 1
        s = 1 if r0 else -1
        if r1 == 0:
            b = (b + s) % m
        else:
            a = (a + s) % m
    return (a * endo + b) % m


# --------------------------------------------------------------------------
# ark-ec 0.3 `VariableBaseMSM::multi_scalar_mul` window rule (a7)
# --------------------------------------------------------------------------
def ark_window_bits(n: int) -> int:
    if n < 32:
        return 3
    # ln_without_floats: log2(n) * 69 / 100, then + 2
    return (n.bit_length() - 1) * 69 // 100 + 2


def msm_pippenger_ark(points, scalars, m, scalar_bits=255):
    n = min(len(points), len(scalars))
    c = ark_window_bits(n)
    window_sums = []
    for w_start in range(0, scalar_bits, c):
        res = None
        buckets = [None] * ((1 << c) - 1)
        for k, pt in zip(scalars, points):
            if k == 0:
                continue
            if k == 1:
                if w_start == 0:
                    res = add(res, pt, m)
                continue
            d = (k >> w_start) & ((1 << c) - 1)
            if d:
                buckets[d - 1] = add(buckets[d - 1], pt, m)
        running = None
        for bkt in reversed(buckets):
            running = add(running, bkt, m)
            res = add(res, running, m)
        window_sums.append(res)
    total = None
    for ws in reversed(window_sums):
        for _ in range(c):
            total = add(total, total, m)
        total = add(total, ws, m)
    return total


# --------------------------------------------------------------------------
# Poseidon (a12): width 3, rate 2, x^7, 55 full rounds, no initial ARK,
# round = sbox -> MDS -> +rc.  Constants are *parameters*.
# --------------------------------------------------------------------------
class PoseidonParams:
    def __init__(self, m, mds, rc, name):
        self.m, self.mds, self.rc, self.name = m, mds, rc, name
        assert len(mds) == 3 and all(len(r) == 3 for r in mds)
        assert len(rc) == 55 and all(len(r) == 3 for r in rc)


def poseidon_permute(state, pp: PoseidonParams):
    m = pp.m
    s = list(state)
    for r in range(55):
        s = [pow(x, 7, m) for x in s]
        s = [sum(pp.mds[i][j] * s[j] for j in range(3)) % m for i in range(3)]
        s = [(s[i] + pp.rc[r][i]) % m for i in range(3)]
    return s


class Sponge:
    """mina-poseidon `ArithmeticSponge` state machine."""
    RATE = 2

    def __init__(self, pp: PoseidonParams):
        self.pp = pp
        self.state = [0, 0, 0]
        self.mode = "absorbed"
        self.count = 0

    def absorb(self, xs):
        m = self.pp.m
        for x in xs:
            if self.mode == "absorbed":
                if self.count == self.RATE:
                    self.state = poseidon_permute(self.state, self.pp)
                    self.state[0] = (self.state[0] + x) % m
                    self.count = 1
                else:
                    self.state[self.count] = (self.state[self.count] + x) % m
                    self.count += 1
            else:
                self.state[0] = (self.state[0] + x) % m
                self.mode, self.count = "absorbed", 1

    def squeeze(self):
        if self.mode == "absorbed" or self.count == self.RATE:
            self.state = poseidon_permute(self.state, self.pp)
            self.mode, self.count = "squeezed", 1
            return self.state[0]
        r = self.state[self.count]
        self.count += 1
        return r


# --------------------------------------------------------------------------
# a16: Merkle-path fold of Proof-of-Account (core/src/proof/account_proof.rs:9-14; README.md:358-362).
# mina `hash_with_kimchi(MERKLE_PARAM[height], [left, right])`: the initial state of height h is the state after
# absorbing the prefix element ("MinaMklTree%03d" padded to 20 bytes with '*', little-endian integer) and permuting.
# [UPSTREAM-RECALL]: un-vendored (mina-tree / mina-hasher); direction convention as in mina's Merkle_path.implied_root:
# `Left h`: the running node is the LEFT input and h the right one.
# --------------------------------------------------------------------------
def merkle_prefix_field(height: int) -> int:
    s = ("MinaMklTree%03d" % height).encode()
    s = s + b"*" * (20 - len(s))
    return int.from_bytes(s, "little")


def merkle_salt(height: int, pp: PoseidonParams):
    return poseidon_permute([merkle_prefix_field(height) % pp.m, 0, 0], pp)


def merkle_root(leaf: int, path, pp: PoseidonParams):
    """path: list of (dir, sibling) with dir 0 = MerkleNode::Left(sibling), 1 = MerkleNode::Right(sibling)"""
    node = leaf
    for h, (d, sib) in enumerate(path):
        st = merkle_salt(h, pp)
        l, r = (node, sib) if d == 0 else (sib, node)
        st = [(st[0] + l) % pp.m, (st[1] + r) % pp.m, st[2]]
        node = poseidon_permute(st, pp)[0]
    return node
