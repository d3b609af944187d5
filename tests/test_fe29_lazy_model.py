"""The 29-bit-limb routines of mina_bridge_amd/csrc/fp29.cuh / ec29.cuh against their PROOFS (tools/fe29_bounds.py) and against Python integers.

Round 5: the bounds are no longer restated here.  tools/fe29_bounds.py carries an interval for every operand and column of every routine and
of the routines' callers (the XYZZ mixed add, the Poseidon lane forms); tools/gen_fe29.py refuses to write fp29.cuh when a rule fails and emits the constants the
proofs ran with (`struct EC29`, `struct SPONGE29`) for the C++ to use by name.  This file
  * runs the proofs, checks the committed fp29.cuh IS what the generator writes, and that ec29.cuh takes every multiple of p from `EC29::`;
  * re-introduces the round-4 bug (ONE p under a canonical y) and other broken disciplines: the prover -- and the generator -- must refuse;
  * restates the generated column loops on Python integers (64-bit accumulator; masked, unmasked or SIGNED quotient digits; the high-half addend, the tenth operand) and
    runs the group law and the lane forms' rounds on random AND adversarial concrete values (coordinates at their invariants, y = p - 1, 2^254, 2^254 - 2^233 - 1 ...):
    every column below 2^64, every limb-wise difference non-negative, every intermediate inside the interval the prover derived, every result the field element
    the textbook formula gives.
What the GPU parity tests cannot reach -- the worst case of a column, the rare table point -- is reached here, on the CPU tier."""
import importlib.util
import os
import random
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


B = _load("fe29_bounds")
L, W, M29, R, P = B.L, B.W, B.M29, B.R, B.P


def limbs(x):
    return B.limbs_of(x)


def value(v):
    return sum(x << (W * i) for i, x in enumerate(v))


# ------------------------------------------------------------------------------------------------ the generated column loop on concrete integers
def model_product_signed(p, pairs, hi=None, c=None):
    """tools/gen_fe29.py `body_sg` on a 64-bit two's-complement accumulator: the digit of column k < 8 is the column's low word read as an int32 and SUBTRACTED against the
    prime limbs; digit 8 is (col & M29) - 2^30.  Returns (result limbs, largest |true column value| seen); asserts that the wrapped register and the true value agree
    whenever the column is shifted."""
    pl = limbs(p)
    MASK = (1 << 64) - 1
    s64 = lambda x: x - (1 << 64) if x >> 63 else x
    col, true, d, r, peak = 0, 0, [0] * L, [0] * L, 0
    for k in range(2 * L - 1):
        for a, b in pairs:
            for i in range(L):
                if 0 <= k - i < L:
                    col = (col + a[i] * b[k - i]) & MASK; true += a[i] * b[k - i]
        for j in (1, 2, 3, 4, 8):
            if 0 <= k - j < L and k - j < k:
                col = (col - d[k - j] * pl[j]) & MASK; true -= d[k - j] * pl[j]
        if hi is not None and k >= L:
            col = (col + hi[k - L]) & MASK; true += hi[k - L]
        if c is not None and k < L:
            col = (col + c[k]) & MASK; true += c[k]
        if k < L:
            lo = col & 0xFFFFFFFF
            d[k] = (lo - (1 << 32) if lo >> 31 else lo) if k < L - 1 else (lo & M29) - (1 << 30)
            col = (col - d[k]) & MASK; true -= d[k]
            assert s64(col) == true and true & M29 == 0, "the signed accumulator wrapped"
            peak = max(peak, abs(true))
            true >>= W; col = (s64(col) >> W) & MASK
        else:
            assert s64(col) == true, "the signed accumulator wrapped"
            peak = max(peak, abs(true))
            r[k - L] = true & M29
            true >>= W; col = (s64(col) >> W) & MASK
    assert true >= 0, "negative top limb"
    r[L - 1] = true + (hi[L - 1] if hi is not None else 0)
    assert peak < 1 << 63 and r[L - 1] < 1 << 32
    return r, peak


def model_product(p, pairs, lazy=False, hi=None, c=None):
    """tools/gen_fe29.py `body`: (result limbs, largest accumulator value seen); lazy = "sg": the signed-digit loop"""
    if lazy == "sg":
        return model_product_signed(p, pairs, hi=hi, c=c)
    pl = limbs(p)
    col, m, r, peak = 0, [0] * L, [0] * L, 0
    for k in range(2 * L - 1):
        for a, b in pairs:
            for i in range(L):
                if 0 <= k - i < L:
                    col += a[i] * b[k - i]
        for j in (1, 2, 3, 4, 8):
            if 0 <= k - j < L and k - j < k:
                col += m[k - j] * pl[j]
        if hi is not None and k >= L:
            col += hi[k - L]
        if c is not None and k < L:
            col += c[k]
        peak = max(peak, col)
        if k < L:
            m[k] = (-col) & (0xFFFFFFFF if lazy else M29)
            col += m[k]
            peak = max(peak, col)
            assert col & M29 == 0
            col >>= W
        else:
            r[k - L] = col & M29
            col >>= W
    r[L - 1] = col + (hi[L - 1] if hi is not None else 0)
    assert peak < 1 << 64, "a column left its 64-bit accumulator"
    assert r[L - 1] < 1 << 32
    return r, peak


def kp_minus(p, mult, b, lend=30):
    k = B.kp_redundant(p, mult, lend)
    out = [k[i] - b[i] for i in range(L)]
    assert all(0 <= x < 1 << 32 for x in out), "a limb of K p - b went negative"
    return out


def inside(v, iv):
    """concrete limbs `v` lie inside the interval the prover derived"""
    return value(v) <= iv["vmax"] and v[8] <= iv["top_limb"]


# ------------------------------------------------------------------------------------------------ the proofs and the generated header
def test_the_shipped_constants_are_proven_and_the_header_is_what_the_generator_writes():
    t = B.prove_all()
    assert t["constants"]["EC29"] == B.EC29 and all(t["fields"][f]["group_law_worst_column"] < 1 and t["fields"][f]["sponge_worst_column"] < 1 for f in (0, 1))
    gen = _load("gen_fe29")
    path = os.path.join(ROOT, "mina_bridge_amd", "csrc", "fp29.cuh")
    s = open(path).read()
    assert gen.rewrite(s) == s, "run `python tools/gen_fe29.py --write` after changing the generator or tools/fe29_bounds.py"
    for k, v in B.EC29.items():
        assert f"static constexpr uint32_t {k} = {v};" in s
    ec = open(os.path.join(ROOT, "mina_bridge_amd", "csrc", "ec29.cuh")).read()
    law = ec[ec.index("template <int F, class FirstY, class Done>"):ec.index("// the bucket value in the 8 x 32 form")]
    assert not re.search(r"fe29_(?:add_)?kp_minus(?:_a_minus_2b)?<F, \d+>", law), "ec29.cuh must take every multiple of p from EC29:: (the proven constants), not from a literal"
    assert law.count("EC29::") >= 6


def test_the_prover_refuses_the_round_4_bug_and_other_broken_disciplines():
    for f in (0, 1):
        with pytest.raises(B.BoundError, match="limb 8 of 1 p - b goes negative"):
            B.prove_group_law(f, {"NEG_Y_MULT": 1})                        # -y = p - y, limb by limb: the top limb 2^22 - 2 - y_8 underflows for y >= 2^254 - 2^233
        with pytest.raises(B.BoundError, match="goes negative"):
            B.prove_group_law(f, {"SUB_X1_MULT": B.EC29["INV_X"]})         # K must exceed the subtrahend's bound by one: no carry pass lends to the top limb
        with pytest.raises(B.BoundError, match="goes negative"):
            B.prove_group_law(f, {"X3_SUB_MULT": 4})                       # round 4's 4 p under the LAZY ppp, q
        with pytest.raises(B.BoundError, match="new acc"):
            B.prove_group_law(f, {"INV_X": B.EC29["INV_X"] - 1})           # the invariants are the least fixed point
        with pytest.raises(B.BoundError):
            B.prove_group_law(f, lazy=B.EC29_LAZY + ("y3", "q", "x3"))     # every product lazy: the shipped constants do not hold (and no constants do: search_lazy_sets)
        with pytest.raises(B.BoundError):
            B.prove_sponge_rounds(f, {"LANES3_STATE_MILLI_P": 2000})       # 2.0 p is not a fixed point of a round (the signed quotient alone brings up to 2.0000001 p)


def test_the_generator_writes_nothing_when_a_proof_fails(monkeypatch):
    gen = _load("gen_fe29")
    path = os.path.join(ROOT, "mina_bridge_amd", "csrc", "fp29.cuh")
    s = open(path).read()
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fe29_bounds                                                    # the module object gen_fe29.proven_constants() imports
    monkeypatch.setitem(fe29_bounds.EC29, "NEG_Y_MULT", 1)
    with pytest.raises(fe29_bounds.BoundError):
        gen.rewrite(s)
    assert open(path).read() == s


# ------------------------------------------------------------------------------------------------ concrete values against the intervals
def test_lazy_products_are_the_same_field_element_and_columns_stay_below_2_64():
    rng = random.Random(2024)
    seen = 0
    for F, p in P.items():
        rinv = pow(R, -1, p)
        bound = B.SPONGE["LANES16_STATE_MILLI_P"] * p // 1000            # the largest state bound any lane form keeps
        top = bound >> B.TOP
        extreme = [M29] * (L - 1) + [top]

        def operand():
            t = rng.random()
            if t < 0.25: return list(extreme)
            if t < 0.45: return limbs(rng.randrange(bound))
            if t < 0.55: return [rng.choice([0, M29]) for _ in range(L - 1)] + [rng.choice([0, top])]
            return limbs(rng.randrange(p))
        for _ in range(1200):
            ops = [operand() for _ in range(6)]
            c = limbs(rng.randrange(p)) if rng.random() < 0.7 else [M29] * (L - 1) + [p >> B.TOP]
            for pairs, cc in (([(ops[0], ops[1]), (ops[2], ops[3]), (ops[4], ops[5])], c), ([(ops[0], ops[1]), (ops[2], ops[3])], c), ([(ops[0], ops[1])], c), ([(ops[0], ops[1])], None),
                              ([(ops[0], ops[0])], None)):
                r, peak = model_product(p, pairs, lazy=True, c=cc)
                total = sum(value(a) * value(b) for a, b in pairs) + (value(cc) if cc else 0)
                assert value(r) % p == total * rinv % p
                assert value(r) * R < total + 8.0001 * p * R
                assert all(x <= M29 for x in r[:-1])
                seen = max(seen, peak)
    assert seen > 1 << 62                                                 # the adversarial patterns do come near the top


def test_signed_digit_products_are_the_same_field_element_inside_one_to_two_p_and_their_columns_fit_a_signed_accumulator():
    """fe29_mul_sg / fe29_sqr_sg / fe29_mul_hi_sg / fe29_sqr_hi_sg (round 5): the digit is the column's own low word (int32), digit 8 = (col & M29) - 2^30.  The result is
    the same field element as T / R, lies in (T / R + 0.99 p, T / R + 2.0001 p) -- positive without an offset term -- with limbs 0..7 normalised, and the TRUE column
    value equals the wrapped 64-bit register at every shift.  Operands up to the largest value any caller feeds (pd < 14 p), all-ones limbs, zero, one."""
    rng = random.Random(77)
    seen = 0
    for F, p in P.items():
        rinv = pow(R, -1, p)
        bound = B.EC29["PD_MAX"] * p
        top = bound >> B.TOP
        extreme = [M29] * (L - 1) + [top]

        def operand():
            t = rng.random()
            if t < 0.2: return list(extreme)
            if t < 0.3: return limbs(rng.choice([0, 1, 2, p - 1, p, p + 1]))
            if t < 0.5: return limbs(rng.randrange(bound))
            if t < 0.6: return [rng.choice([0, M29]) for _ in range(L - 1)] + [rng.choice([0, top])]
            return limbs(rng.randrange(p))
        for _ in range(1500):
            a, b = operand(), operand()
            k = B.kp_redundant(p, B.EC29["SUB_X1_MULT"], 30)
            h = [k[i] - x for i, x in enumerate(limbs(rng.randrange(B.EC29["INV_X"] * p)))]
            for pairs, hh in (([(a, b)], None), ([(a, a)], None), ([(a, b)], h), ([(a, a)], h)):
                r, peak = model_product(p, pairs, lazy="sg", hi=hh)
                total = sum(value(x) * value(y) for x, y in pairs)
                extra = value(hh) if hh else 0
                assert (value(r) - extra) % p == total * rinv % p
                assert 0.99 * p * R < (value(r) - extra) * R - total < 2.0001 * p * R
                assert all(x <= M29 for x in r[:-1])
                seen = max(seen, peak)
    assert (1 << 60) < seen < (1 << 63)
    # what the signed form cannot carry is refused by the prover: the group law's two-term dot product on raw operands, a four-term row
    for f in (0, 1):
        with pytest.raises(B.BoundError, match="signed accumulator"):
            B.prove_group_law(f, signed=B.EC29_SIGNED + ("y3",))
        pr = B.Prover(f)
        pr.product("row", [(B.norm(pr.p), B.norm(3 * pr.p))] * 3, lazy="sg", c=B.norm(pr.p))       # the MDS row (27 limb products per column) still fits: 0.92 of the range ...
        with pytest.raises(B.BoundError, match="signed accumulator"):
            pr.product("four-term row", [(B.norm(pr.p), B.norm(3 * pr.p))] * 4, lazy="sg", c=B.norm(pr.p))                                              # ... a fourth term would not


def _law(p, acc, qx, qy, c):
    """xyzz29_add_affine on concrete limbs (ec29.cuh, statement by statement); returns every intermediate"""
    md = lambda name: B.mode_of(name, B.EC29_LAZY, B.EC29_SIGNED)
    pd, _ = model_product(p, [(qx, acc["zz"])], lazy=md("pd"), hi=kp_minus(p, c["SUB_X1_MULT"], acc["x"]))
    r, _ = model_product(p, [(qy, acc["zzz"])], lazy=md("r"), hi=kp_minus(p, c["SUB_Y1_MULT"], acc["y"]))
    pp, _ = model_product(p, [(pd, pd)], lazy=md("pp"))
    ppp, _ = model_product(p, [(pd, pp)], lazy=md("ppp"))
    q, _ = model_product(p, [(acc["x"], pp)], lazy=md("q"))
    k = B.kp_redundant(p, c["X3_SUB_MULT"], 31)
    h = [k[i] - ppp[i] - 2 * q[i] for i in range(L)]
    assert all(0 <= x < 1 << 32 for x in h), "a limb of K p - ppp - 2 q went negative"
    x3, _ = model_product(p, [(r, r)], lazy=md("x3"), hi=h)
    k3 = kp_minus(p, c["SUB_X3_MULT"], x3)
    a = [q[i] + k3[i] for i in range(L)]
    assert all(x < 1 << 32 for x in a)
    y3, peak = model_product(p, [(r, a), (kp_minus(p, c["SUB_Y1_MULT"], acc["y"]), ppp)], lazy=md("y3"))
    zz, _ = model_product(p, [(acc["zz"], pp)], lazy=md("zz"))
    zzz, _ = model_product(p, [(acc["zzz"], ppp)], lazy=md("zzz"))
    return {"pd": pd, "r": r, "pp": pp, "ppp": ppp, "q": q, "x3": x3, "y3": y3, "zz": zz, "zzz": zzz}, peak


def test_the_group_law_on_concrete_values_stays_inside_the_proven_intervals_and_is_the_textbook_formula():
    """madd-2008-s on XYZZ coordinates in the Montgomery-2^261 domain: with U2 = X2 ZZ1, S2 = Y2 ZZZ1, P = U2 - X1, R = S2 - Y1: X3 = R^2 - PPP - 2 Q, Y3 = R (Q - X3) - Y1 PPP,
    ZZ3 = ZZ1 PP, ZZZ3 = ZZZ1 PPP (Q = X1 PP) -- each product carrying one factor 1/2^261"""
    c = B.EC29
    rng = random.Random(29)
    worst = 0
    for F, p in P.items():
        table = B.prove_all()["fields"][F]["group_law"]
        rinv = pow(R, -1, p)
        edge_y = [p - 1, p - 2, 1 << 254, (1 << 254) - 1, (1 << 254) - (1 << 232), (1 << 254) - (1 << 233) - 1, 1]
        inv = {"x": c["INV_X"] * p, "y": c["INV_Y"] * p, "zz": c["INV_ZZ"] * p, "zzz": c["INV_ZZZ"] * p}
        for it in range(1500):
            at_edge = it % 5 == 0
            acc_v = {k: (b - 1 - rng.randrange(4) if at_edge else rng.randrange(b)) for k, b in inv.items()}
            acc = {k: limbs(v) for k, v in acc_v.items()}
            x2 = p - 1 - rng.randrange(3) if at_edge else rng.randrange(p)
            y = edge_y[it % len(edge_y)] if it < 4 * len(edge_y) else rng.randrange(1, p)
            neg = rng.random() < 0.5
            # the two table forms: the pre-split record holds p - y normalised; the 8-word twin negates limb by limb with NEG_Y_MULT p (raw)
            for qy in ((limbs(p - y) if neg else limbs(y)), (kp_minus(p, c["NEG_Y_MULT"], limbs(y)) if neg else limbs(y))):
                out, peak = _law(p, acc, limbs(x2), qy, c)
                worst = max(worst, peak)
                y2 = (p - y) % p if neg else y
                U2, S2 = x2 * acc_v["zz"] * rinv, y2 * acc_v["zzz"] * rinv
                Pd, Rr = (U2 - acc_v["x"]) % p, (S2 - acc_v["y"]) % p
                PP = Pd * Pd * rinv % p; PPP = Pd * PP * rinv % p; Q = acc_v["x"] * PP * rinv % p
                X3 = (Rr * Rr * rinv - PPP - 2 * Q) % p
                Y3 = (Rr * (Q - X3) * rinv - acc_v["y"] * PPP * rinv) % p
                want = {"pd": Pd, "r": Rr, "pp": PP, "ppp": PPP, "q": Q, "x3": X3, "y3": Y3, "zz": acc_v["zz"] * PP * rinv % p, "zzz": acc_v["zzz"] * PPP * rinv % p}
                for name, v in out.items():
                    assert value(v) % p == want[name], (F, it, name)
                    assert inside(v, table[name]), (F, it, name, value(v) / p, table[name]["vmax"] / p)
                    assert all(x <= M29 for x in v[:-1])
                # the new accumulator is inside the invariants again
                assert value(out["x3"]) < inv["x"] and value(out["y3"]) < inv["y"] and value(out["zz"]) < inv["zz"] and value(out["zzz"]) < inv["zzz"]
    assert worst < 1 << 64


def test_the_exact_zero_test_of_the_group_law_covers_every_multiple_of_p_below_its_bound():
    """ec29.cuh fe29_is_multiple_of_p: a normalised value below EC29::PD_MAX p is 0 mod p iff limbs 5..7 and the low 22 bits of limb 8 are zero and limbs 0..4 equal k c"""
    for F, p in P.items():
        cpart = p - (1 << 254)
        pl = limbs(p)

        def is_multiple(a):
            if a[5] | a[6] | a[7] | (a[8] & 0x3FFFFF): return False
            k = a[8] >> 22
            t, carry = [], 0
            for j in range(5):
                x = k * pl[j] + carry; t.append(x & M29); carry = x >> W
            return a[:5] == t and carry == 0
        rng = random.Random(5)
        for k in list(range(B.EC29["PD_MAX"] + 1)) + [rng.randrange(1 << 19) for _ in range(50)]:
            assert k * cpart < 1 << 145                                   # k c stays inside limbs 0..4
            assert is_multiple(limbs(k * p))
            for d in (1, 1 << 29, 1 << 145, 1 << 232):
                assert not is_multiple(limbs(k * p + d))


def test_the_lane_forms_rounds_on_concrete_values_stay_inside_their_proven_bounds():
    rng = random.Random(31)
    for F, p in P.items():
        rinv = pow(R, -1, p)
        table = B.prove_all()["fields"][F]["sponge"]
        bound = B.SPONGE["LANES3_STATE_MILLI_P"] * p // 1000
        for it in range(400):
            st = [bound - 1 - rng.randrange(3) if it % 4 == 0 else rng.randrange(bound) for _ in range(3)]
            mds = [[p - 1 - rng.randrange(2) if it % 4 == 0 else rng.randrange(p) for _ in range(3)] for _ in range(3)]
            rc = [rng.randrange(p) for _ in range(3)]
            x7 = []
            for x in st:
                xl = limbs(x)
                x2, _ = model_product(p, [(xl, xl)], lazy="sg"); x4, _ = model_product(p, [(x2, x2)], lazy="sg")
                x6, _ = model_product(p, [(x4, x2)], lazy="sg"); t, _ = model_product(p, [(x6, xl)], lazy="sg")
                assert value(t) <= table["lanes3"]["x7"]["vmax"] and value(t) % p == pow(x, 7, p) * pow(rinv, 6, p) % p
                x7.append(t)
            for row in range(3):
                out, _ = model_product(p, [(limbs(mds[row][cidx]), x7[cidx]) for cidx in range(3)], lazy="sg", c=limbs(rc[row]))
                assert value(out) < bound and value(out) <= table["lanes3"]["row"]["vmax"]
                assert value(out) % p == (sum(mds[row][cidx] * value(x7[cidx]) for cidx in range(3)) + rc[row]) * rinv % p


def test_the_general_add_on_concrete_values_stays_inside_the_proven_intervals_and_is_the_textbook_formula():
    """ec29.cuh xyzz29_add (add-2008-s on XYZZ, both operands accumulators within the invariants): U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1, P = U2 - U1, R = S2 - S1,
    X3 = R^2 - PPP - 2 Q (Q = U1 PP), Y3 = R (Q - X3) - S1 PPP, ZZ3 = ZZ1 ZZ2 PP, ZZZ3 = ZZZ1 ZZZ2 PPP -- each product carrying one factor 1 / 2^261"""
    c, e = B.EC29_GENERAL, B.EC29
    md = lambda name: B.mode_of(name, B.EC29_GENERAL_LAZY, B.EC29_GENERAL_SIGNED)
    rng = random.Random(41)
    for F, p in P.items():
        table = B.prove_all()["fields"][F]["group_add"]
        rinv = pow(R, -1, p)
        inv = {"x": e["INV_X"] * p, "y": e["INV_Y"] * p, "zz": e["INV_ZZ"] * p, "zzz": e["INV_ZZZ"] * p}
        for it in range(800):
            edge = it % 4 == 0
            av = {k: (b - 1 - rng.randrange(3) if edge else rng.randrange(b)) for k, b in inv.items()}
            bv = {k: (b - 1 - rng.randrange(3) if edge and it % 8 == 0 else rng.randrange(b)) for k, b in inv.items()}
            a = {k: limbs(v) for k, v in av.items()}; b = {k: limbs(v) for k, v in bv.items()}
            u1, _ = model_product(p, [(a["x"], b["zz"])], lazy=md("u1")); s1, _ = model_product(p, [(a["y"], b["zzz"])], lazy=md("s1"))
            pd, _ = model_product(p, [(b["x"], a["zz"])], lazy=md("pd"), hi=kp_minus(p, c["G_U1_MULT"], u1))
            r, _ = model_product(p, [(b["y"], a["zzz"])], lazy=md("r"), hi=kp_minus(p, c["G_S1_MULT"], s1))
            pp, _ = model_product(p, [(pd, pd)], lazy=md("pp")); ppp, _ = model_product(p, [(pd, pp)], lazy=md("ppp")); q, _ = model_product(p, [(u1, pp)], lazy=md("q"))
            k4 = B.kp_redundant(p, c["G_X3_SUB_MULT"], 31)
            h = [k4[i] - ppp[i] - 2 * q[i] for i in range(L)]
            assert all(0 <= x < 1 << 32 for x in h)
            x3, _ = model_product(p, [(r, r)], lazy=md("x3"), hi=h)
            k3 = kp_minus(p, c["G_SUB_X3_MULT"], x3)
            y3, _ = model_product(p, [(r, [q[i] + k3[i] for i in range(L)]), (kp_minus(p, c["G_S1_MULT"], s1), ppp)], lazy=md("y3"))
            zz12, _ = model_product(p, [(a["zz"], b["zz"])], lazy=md("zz12")); zz, _ = model_product(p, [(zz12, pp)], lazy=md("zz"))
            zzz12, _ = model_product(p, [(a["zzz"], b["zzz"])], lazy=md("zzz12")); zzz, _ = model_product(p, [(zzz12, ppp)], lazy=md("zzz"))
            U1, U2, S1, S2 = av["x"] * bv["zz"] * rinv % p, bv["x"] * av["zz"] * rinv % p, av["y"] * bv["zzz"] * rinv % p, bv["y"] * av["zzz"] * rinv % p
            Pd, Rr = (U2 - U1) % p, (S2 - S1) % p
            PP = Pd * Pd * rinv % p; PPP = Pd * PP * rinv % p; Q = U1 * PP * rinv % p
            X3 = (Rr * Rr * rinv - PPP - 2 * Q) % p; Y3 = (Rr * (Q - X3) * rinv - S1 * PPP * rinv) % p
            want = {"u1": U1, "s1": S1, "pd": Pd, "r": Rr, "pp": PP, "ppp": PPP, "q": Q, "x3": X3, "y3": Y3,
                    "zz": av["zz"] * bv["zz"] * rinv % p * PP * rinv % p, "zzz": av["zzz"] * bv["zzz"] * rinv % p * PPP * rinv % p}
            got = {"u1": u1, "s1": s1, "pd": pd, "r": r, "pp": pp, "ppp": ppp, "q": q, "x3": x3, "y3": y3, "zz": zz, "zzz": zzz}
            for name, v in got.items():
                assert value(v) % p == want[name], (F, it, name)
                assert inside(v, table[name]), (F, it, name)
            assert value(x3) < inv["x"] and value(y3) < inv["y"] and value(zz) < inv["zz"] and value(zzz) < inv["zzz"]
