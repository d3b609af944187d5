"""The LAZY 29-bit-limb Montgomery products of mina_bridge_amd/csrc/fp29.cuh (tools/gen_fe29.py: fe29_mul_lz / fe29_sqr_lz / fe29_dot3rc_lz / fe29_mulrc_lz /
fe29_dot2rc_lz), restated on Python integers column by column exactly as the generated code runs them (64-bit accumulator, quotient digit = -col mod 2^32
NOT masked to 29 bits, the tenth operand added before the reduction).  What the GPU parity tests cannot reach -- the worst case of a column -- is checked here:
  * no column ever exceeds 64 bits, for operands at the bounds the permutation's lane forms keep (analytic maximum + adversarial limb patterns);
  * the result is the same field element as (sum of products + c) / 2^261 and stays below (sum + c) / 2^261 + 8.0001 p;
  * the value bounds quoted in sponge.cuh for the 3-, 8- and 16-lane forms are fixed points of a round."""
import random

P = {0: 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001, 1: 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001}
L, W = 9, 29
M29 = (1 << W) - 1
R = 1 << (L * W)


def limbs(x):
    return [(x >> (W * i)) & M29 for i in range(L - 1)] + [x >> (W * (L - 1))]


def value(v):
    return sum(x << (W * i) for i, x in enumerate(v))


def lazy_product(p, pairs, c=None):
    """(result limbs, largest accumulator value seen): the column loop of gen_fe29.body(lazy=True)"""
    pl = limbs(p)
    assert pl[0] == 1 and pl[5] == pl[6] == pl[7] == 0 and pl[8] == 1 << 22          # the shape P29<F> asserts
    col, m, r, peak = 0, [0] * L, [0] * L, 0
    for k in range(2 * L - 1):
        for a, b in pairs:
            for i in range(L):
                if 0 <= k - i < L:
                    col += a[i] * b[k - i]
        if c is not None and k < L:
            col += c[k]
        for j in (1, 2, 3, 4, 8):
            if 0 <= k - j < L and k - j < k:
                col += m[k - j] * pl[j]
        peak = max(peak, col)
        if k < L:
            m[k] = (-col) & 0xFFFFFFFF
            col += m[k]
            peak = max(peak, col)
            assert col & 0xFFFFFFFF == 0
            col >>= W
        else:
            r[k - L] = col & M29
            col >>= W
    assert col < 1 << 32
    r[L - 1] = col
    return r, peak


def test_analytic_column_maximum_fits_64_bits():
    # a column of the 3-term dot product: 27 limb products, the quotient terms m p_1 .. m p_4 (m < 2^32, p_j < 2^29), m 2^22, m, the tenth operand, the carry
    worst = 27 * M29 * M29 + 4 * (2**32 - 1) * M29 + (2**32 - 1) * (1 << 22) + (2**32 - 1) + M29 + (1 << 35)
    assert worst < 1 << 64 and worst / 2**64 < 0.93


def test_lazy_products_are_the_same_field_element_and_columns_stay_below_2_64():
    rng = random.Random(2024)
    seen = 0
    for F, p in P.items():
        rinv = pow(R, -1, p)
        top = (26 * p) >> (W * (L - 1))                       # the 16-lane form's bound on the state (24.3 p) with margin
        extreme = [M29] * (L - 1) + [top]

        def operand():
            t = rng.random()
            if t < 0.25: return list(extreme)
            if t < 0.45: return limbs(rng.randrange(26 * p))
            if t < 0.55: return [rng.choice([0, M29]) for _ in range(L - 1)] + [rng.choice([0, top])]
            return limbs(rng.randrange(p))
        for _ in range(1500):
            ops = [operand() for _ in range(6)]
            c = limbs(rng.randrange(p)) if rng.random() < 0.7 else [M29] * (L - 1) + [p >> (W * (L - 1))]
            for pairs, cc in (([(ops[0], ops[1]), (ops[2], ops[3]), (ops[4], ops[5])], c), ([(ops[0], ops[1]), (ops[2], ops[3])], c), ([(ops[0], ops[1])], c), ([(ops[0], ops[1])], None),
                              ([(ops[0], ops[0])], None)):
                r, peak = lazy_product(p, pairs, cc)
                total = sum(value(a) * value(b) for a, b in pairs) + (value(cc) if cc else 0)
                assert peak < 1 << 64
                assert value(r) % p == total * rinv % p
                assert value(r) * R < total + 8.0001 * p * R
                assert all(x <= M29 for x in r[:-1])
                seen = max(seen, peak)
    assert seen > 1 << 62                                      # the adversarial patterns do come near the top


def test_value_bounds_of_the_lane_forms_are_fixed_points_of_a_round():
    # in units of p; a lazy product of operands below A p and B p is below A B / 128 + 8.0001 (p / 2^261 < 2^-7), the round constant inside a reduction adds 2^-7
    lz = lambda a, b, n=1, rc=0: n * a * b / 128 + rc / 128 + 8.0001
    # 3-lane: x -> x^2, x^4, x^6, x^7, row (three terms, MDS entries below p, the round constant inside)
    x = 8.3
    x2 = lz(x, x); x4 = lz(x2, x2); x7 = lz(lz(x4, x2), x)
    assert lz(1, x7, 3, 1) < x and max(x2, x4, x7) < 8.8
    # 8-lane: x = u + swap(u), u a two-term half row
    x = 16.5
    x2 = lz(x, x); y = max(lz(x2, x), lz(x2, x2)); t = lz(y, y)
    assert 2 * lz(1, t, 2, 1) < x
    # 16-lane: x = sum of three single products
    x = 24.4
    x2 = lz(x, x); y = max(lz(x2, x), lz(x2, x2)); t = lz(y, y)
    assert 3 * lz(1, t, 1, 1) < x
    # every bound leaves the top limb far below 2^29 and the strict product on the way out below 2^256
    assert 26 * P[0] < 1 << (W * (L - 1) + 27) and (24.4 / 128 + 1) * P[1] < 1 << 256


def strict_product_hi(p, a, b, h):
    """gen_fe29.body(lazy=False, hi=h): quotient digits masked to 29 bits, the 9-limb addend h entering columns 9 .. 17"""
    pl = limbs(p)
    col, m, r, peak = 0, [0] * L, [0] * L, 0
    for k in range(2 * L - 1):
        for i in range(L):
            if 0 <= k - i < L:
                col += a[i] * b[k - i]
        for j in (1, 2, 3, 4, 8):
            if 0 <= k - j < L and k - j < k:
                col += m[k - j] * pl[j]
        if k >= L:
            col += h[k - L]
        peak = max(peak, col)
        if k < L:
            m[k] = (-col) & M29
            col += m[k]
            col >>= W
        else:
            r[k - L] = col & M29
            col >>= W
    r[L - 1] = col + h[L - 1]
    assert r[L - 1] < 1 << 32
    return r, peak


def test_group_law_subtractions_inside_a_reduction():
    """ec29.cuh: u2 + 8 p - x1 and r^2 + 4 p - ppp - 2 q as ONE product / square each -- the "K p - ..." operand in limbs that never go negative,
    added to the high half of the product before the carries (fe29_mul_hi_asm / fe29_sqr_hi_asm): the same integers as fe29_sub_kp gave"""
    rng = random.Random(7)
    for F, p in P.items():
        rinv = pow(R, -1, p)
        n8, n4 = limbs(8 * p), limbs(4 * p)
        k8 = [n8[0] + (1 << 30)] + [n8[i] + (1 << 30) - 2 for i in range(1, 8)] + [n8[8] - 2]
        k4 = [n4[0] + (1 << 31)] + [n4[i] + (1 << 31) - 4 for i in range(1, 8)] + [n4[8] - 4]
        assert value(k8) == 8 * p and value(k4) == 4 * p and all(0 <= x < 1 << 32 for x in k8 + k4)
        for _ in range(3000):
            edge = rng.random() < 0.2
            x1 = 6 * p - 1 if edge else rng.randrange(6 * p)                     # accumulator x below 6 p
            qx, zz = rng.randrange(p), (3 * p - 1 if edge else rng.randrange(3 * p))
            h = [k8[i] - limbs(x1)[i] for i in range(L)]
            assert all(0 <= v < 1 << 32 for v in h)
            r, peak = strict_product_hi(p, limbs(qx), limbs(zz), h)
            assert peak < 1 << 64 and all(v <= M29 for v in r[:-1])
            got = value(r)
            assert got % p == (qx * zz * rinv - x1) % p and got < 11 * p       # u2 + 8 p - x1, u2 < 3 p
            ppp, q = (int(1.2 * p) - 1 if edge else rng.randrange(int(1.2 * p))), (int(1.1 * p) - 1 if edge else rng.randrange(int(1.1 * p)))
            h = [k4[i] - limbs(ppp)[i] - 2 * limbs(q)[i] for i in range(L)]
            assert all(0 <= v < 1 << 32 for v in h)
            rr = limbs(11 * p - 1 if edge else rng.randrange(11 * p))
            r, peak = strict_product_hi(p, rr, rr, h)
            assert peak < 1 << 64
            got = value(r)
            assert got % p == (value(rr) ** 2 * rinv - ppp - 2 * q) % p and got < 6 * p


def test_group_law_dot_product_takes_its_differences_raw():
    """ec29.cuh y3 = r (q + 8 p - x3) + (8 p - y1) ppp: both differences enter the strict two-term dot product limb by limb, NOT normalised (limbs up to 2^31).
    The worst column -- every limb at its maximum -- stays inside 64 bits, and the value is the one the normalised operands gave."""
    rng = random.Random(11)
    for F, p in P.items():
        rinv = pow(R, -1, p)
        pl = limbs(p)
        n8 = limbs(8 * p)
        k8 = [n8[0] + (1 << 30)] + [n8[i] + (1 << 30) - 2 for i in range(1, 8)] + [n8[8] - 2]
        amax = [k8[i] + M29 for i in range(8)] + [k8[8] + (int(1.1 * p) >> 232)]
        rmax, pppmax = [M29] * 8 + [(11 * p) >> 232], [M29] * 8 + [int(1.2 * p) >> 232]
        carry = worst = 0
        for col in range(2 * L - 1):
            s = sum(rmax[i] * amax[col - i] + k8[i] * pppmax[col - i] for i in range(L) if 0 <= col - i < L)
            s += sum(M29 * pl[j] for j in (1, 2, 3, 4, 8) if 0 <= col - j < L and col - j < col) + M29 + carry
            worst, carry = max(worst, s), s >> W
        assert worst < 0.7 * 2**64
        for _ in range(2000):
            r_, q, x3, y1, ppp = rng.randrange(11 * p), rng.randrange(int(1.1 * p)), rng.randrange(6 * p), rng.randrange(2 * p), rng.randrange(int(1.2 * p))
            a = [limbs(q)[i] + k8[i] - limbs(x3)[i] for i in range(L)]
            b = [k8[i] - limbs(y1)[i] for i in range(L)]
            assert all(0 <= v < 1 << 32 for v in a + b) and value(a) == q + 8 * p - x3 and value(b) == 8 * p - y1
            zero = [0] * L
            # the two-term dot product, strict: model it as two products accumulated (strict_product_hi takes one pair: add the second pair's columns through `h` is not possible -- run the columns here)
            col, m, out = 0, [0] * L, [0] * L
            for k in range(2 * L - 1):
                for i in range(L):
                    if 0 <= k - i < L:
                        col += limbs(r_)[i] * a[k - i] + b[i] * limbs(ppp)[k - i]
                for j in (1, 2, 3, 4, 8):
                    if 0 <= k - j < L and k - j < k:
                        col += m[k - j] * pl[j]
                assert col < 1 << 64
                if k < L:
                    m[k] = (-col) & M29; col += m[k]; col >>= W
                else:
                    out[k - L] = col & M29; col >>= W
            out[L - 1] = col
            assert value(out) % p == ((r_ * (q - x3) - y1 * ppp) * rinv) % p and value(out) < 2 * p


def test_negated_table_point_enters_its_product_raw():
    """ec29.cuh: a negative digit adds (x, p - y); p - y is NOT normalised (limbs below 2^30 + 2^29) where it is a product's operand: s2 + 8 p - y1 = (p - y) zzz / 2^261 + h"""
    rng = random.Random(13)
    for F, p in P.items():
        rinv = pow(R, -1, p)
        n1, n8 = limbs(2 * p), limbs(8 * p)                   # TWO p: with one p the top limb 2^22 - 2 - y_8 goes negative for y >= 2^254 - 2^233 (no carry pass lends to it)
        k1 = [n1[0] + (1 << 30)] + [n1[i] + (1 << 30) - 2 for i in range(1, 8)] + [n1[8] - 2]
        k8 = [n8[0] + (1 << 30)] + [n8[i] + (1 << 30) - 2 for i in range(1, 8)] + [n8[8] - 2]
        assert value(k1) == 2 * p and all(0 <= v < 1 << 32 for v in k1)
        edge = [p - 1, p - 2, 1 << 254, (1 << 254) - 1, (1 << 254) - (1 << 232), (1 << 254) - (1 << 233) - 1, 1]
        for it in range(2000):
            y, zzz, y1 = (edge[it] if it < len(edge) else rng.randrange(1, p)), rng.randrange(3 * p), rng.randrange(2 * p)
            qy = [k1[i] - limbs(y)[i] for i in range(L)]
            h = [k8[i] - limbs(y1)[i] for i in range(L)]
            assert all(0 <= v < (1 << 30) + (1 << 29) for v in qy)
            r, peak = strict_product_hi(p, qy, limbs(zzz), h)
            assert peak < 1 << 63 and value(r) % p == ((p - y) * zzz * rinv - y1) % p and value(r) < 11 * p and value(qy) == 2 * p - y
        worst = strict_product_hi(p, list(k1), [M29] * 8 + [(3 * p) >> 232], list(k8))[1]
        assert worst < 1 << 63
