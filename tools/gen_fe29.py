#!/usr/bin/env python3
"""Generates the straight-line gfx950 routines of mina_bridge_amd/csrc/fp29.cuh (9 limbs of 29 bits, Montgomery R = 2^261, no carries):
    fe29_mul_asm    a * b / R
    fe29_sqr_asm    a * a / R        cross terms once, against the doubled limbs (45 limb products instead of 81)
    fe29_dot2_asm   (a0 b0 + a1 b1) / R             one reduction (the 8-lane permutation's split MDS row)
    fe29_dot3_asm   (a0 b0 + a1 b1 + a2 b2) / R     one reduction
    fe29_mul_hi_asm a * b / R + h    the same on a product (the group law's  x2 zz1 + (8 p - x1))
    fe29_sqr_hi_asm a * a / R + h    h added limb by limb to the high half of the product (strict form; the group law's  r^2 + (4 p - ppp - 2 q))
and the LAZY forms the chip-filling 3-lane permutation runs its rounds in (fe29_sqr_lz, fe29_mul_lz, fe29_dot3rc_lz; fe29_mulrc_lz / fe29_dot2rc_lz for the 16- and 8-lane latency forms): the quotient digit
m_k = -col mod 2^32 is NOT masked to 29 bits (its three high bits add a multiple of p 2^(29 k): the value stays the same field element,
the result is < a b / R + 8.0001 p instead of < a b / R + p), the accumulator starts from the first product (no zeroing), and the dot
product takes a tenth operand c added before the reduction ((sum + c) / R: the round constant, stored times R).  Bounds: fp29.cuh.
SIGNED-digit forms (round 5, fe29_mul_sg / fe29_sqr_sg / fe29_mul_hi_sg / fe29_sqr_hi_sg): the quotient digit of column k < 8 is the column's own low word read as an
int32 -- NO instruction makes it (the lazy forms spend a v_sub per digit, the strict ones a v_sub and a v_and) -- and it is SUBTRACTED by `v_mad_i64_i32` against the
negated prime limbs; digit 8 is (col & M29) - 2^30 (one v_and_or), always negative, so the quotient is positive without an offset term: the result is
T / R + (1 p, 2 p], limbs 0..7 normalised (tools/fe29_bounds.py `product_signed`; measured: tools/probes/sg_probe.hip).
One asm statement per chunk of <= 12 multiply-accumulates of a column (inline asm takes at most 30 operands); between the columns plain
C++ (mask, shift).  The signed-digit routines: one statement per COLUMN (cancel + carry + products + reduction terms), split only where 28 inputs do not suffice.  The compiler's own schedule of the C++ form spreads a column over several accumulators and re-adds them (+116 64-bit
adds and +70 products per Poseidon round); pinned like this a round is ~1150 VALU instructions instead of ~1610.
    python tools/gen_fe29.py > /tmp/fe29_gen.inc   (pasted between the GENERATED markers of fp29.cuh by the same script with --write)"""
import os
import sys

MAXT = 12
L = 9
W = 29


def mads(terms, fresh=False):
    """asm statements accumulating `terms` (pairs of C expressions, or (expr, int literal)) into col; fresh: col is not read (the first product starts it)"""
    out = []
    for c0 in range(0, len(terms), MAXT):
        chunk = terms[c0:c0 + MAXT]
        lines, ops = [], []
        n = 2
        for x, y in chunk:
            addend = "0" if fresh and c0 == 0 and not lines else "%0"
            if isinstance(y, int):
                lines.append(f"v_mad_u64_u32 %0, %1, %{n}, {y}, {addend}"); ops.append(f'"v"({x})'); n += 1
            else:
                lines.append(f"v_mad_u64_u32 %0, %1, %{n}, %{n + 1}, {addend}"); ops += [f'"v"({x})', f'"v"({y})']; n += 2
        out.append('    asm("' + '\\n\\t'.join(lines) + '"\n        : "' + ("=&v" if fresh and c0 == 0 else "+&v") + '"(col), "=&s"(cc) : ' + ", ".join(ops) + ");")
    return out


def body(col_terms, lazy=False, hi=None, fresh=True):
    """hi: name of a 9-limb operand h (limbs below 2^32) added to the HIGH half -- columns 9 .. 17 -- before the carries: the result is (sum + m p) / 2^261 + h exactly.
    fresh (round 5: every form): the accumulator starts from the first product of column 0 instead of a zero (two v_mov less per routine)"""
    out = ["    uint64_t col, cc; fe29_t r;" if fresh else "    uint64_t col = 0, cc; fe29_t r;", "    uint32_t " + ", ".join(f"m{i}" for i in range(L)) + ";",
           "    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;"]
    for k in range(2 * L - 1):
        terms = list(col_terms(k))
        for j, pj in ((1, "p1"), (2, "p2"), (3, "p3"), (4, "p4")):
            i = k - j
            if 0 <= i < L and i < k:
                terms.append((f"m{i}", pj))
        i = k - 8
        if 0 <= i < L and i < k:
            terms.append((f"m{i}", "p8"))                        # p_8 = 2^22
        if hi is not None and k >= L:
            terms.append((f"{hi}.v[{k - L}]", 1))
        out.append(f"    // column {k}: {len(terms)} products")
        out += mads(terms, fresh=fresh and k == 0)
        if k < L:
            out.append(f"    m{k} = 0u - (uint32_t)col;" if lazy else f"    m{k} = (0u - (uint32_t)col) & M29;")
            out += mads([(f"m{k}", 1)])                          # + m_k p_0: the low limb cancels
            out.append(f"    col >>= {W};")
        else:
            out.append(f"    r.v[{k - L}] = (uint32_t)col & M29; col >>= {W};")
    out.append(f"    r.v[{L - 1}] = (uint32_t)col;" if hi is None else f"    r.v[{L - 1}] = (uint32_t)col + {hi}.v[{L - 1}];")
    out.append("    return r;")
    return out


def smads(terms):
    """signed multiply-accumulates (quotient digit x negated prime limb) into col, in place"""
    out = []
    for c0 in range(0, len(terms), MAXT):
        chunk = terms[c0:c0 + MAXT]
        lines, ops, n = [], [], 2
        for x, y in chunk:
            lines.append(f"v_mad_i64_i32 %0, %1, %{n}, %{n + 1}, %0"); ops += [f'"v"({x})', f'"s"({y})']; n += 2      # the negated prime limb rides the constant bus: no VGPR
        out.append('    asm("' + '\\n\\t'.join(lines) + '"\n        : "+&v"(col), "=&s"(cc) : ' + ", ".join(ops) + ");")
    return out


def _pack(units, first_out, var):
    """units: (instruction text with %0 = the accumulator pair, %1 = the scratch carry pair and @i = its own i-th input, [input operands]); packed greedily into asm
    statements of at most 28 inputs (+ 2 outputs = inline asm's 30).  first_out: constraint of the accumulator in the FIRST statement ("=&v": its first instruction
    writes it without reading it); later statements of the same column read and write it ("+&v")."""
    out, lines, ops = [], [], []
    def flush():
        nonlocal lines, ops
        if lines:
            out.append('    asm("' + '\\n\\t'.join(lines) + '"\n        : "' + (first_out if not out else "+&v") + f'"({var}), "=&s"(cc) : ' + ", ".join(ops) + ");")
        lines, ops = [], []
    for text, uops in units:
        if len(ops) + len(uops) > 28:
            flush()
        for i, o in enumerate(uops):
            text = text.replace(f"@{i}", f"%{2 + len(ops) + i}")
        lines.append(text); ops += list(uops)
    flush()
    return out


def body_sg(col_terms, hi=None):
    """the column loop with SIGNED quotient digits.  m_k (k < 8) is the low register of the column itself: the instruction that cancels the low limb writes the new column
    to OTHER registers (early-clobber output), so the old low word stays where it is for the five later uses of the digit -- no copy, no negation, no mask.
    ONE asm statement per column where the operand limit allows (end of round 5): cancel the previous column's low limb, carry, this column's products, its reduction
    terms.  The compiler must put an `s_nop 0` after every asm statement whose result the next instruction reads (the gfx940 forwarding-hazard rule, which it has to assume of
    an asm it cannot read); three statements per column were 43 of them per product, free at five waves per SIMD and 5 - 7 % of the chain at one."""
    out = ["    uint64_t col, nc, cc; fe29_t r;", "    uint32_t " + ", ".join(f"m{i}" for i in range(L)) + ";",
           "    const int32_t n1 = -(int32_t)P29<F>::L1, n2 = -(int32_t)P29<F>::L2, n3 = -(int32_t)P29<F>::L3, n4 = -(int32_t)P29<F>::L4, n8 = -(int32_t)P29<F>::L8;"]
    for k in range(2 * L - 1):
        terms, st = list(col_terms(k)), []
        for j, nj in ((1, "n1"), (2, "n2"), (3, "n3"), (4, "n4"), (8, "n8")):
            i = k - j
            if 0 <= i < L and i < k:
                st.append((f"m{i}", nj))
        if hi is not None and k >= L:
            terms.append((f"{hi}.v[{k - L}]", 1))
        cancel = 1 <= k <= L                                     # the previous column was a digit column: its low limb is cancelled and its carry taken HERE
        out.append(f"    // column {k}: {len(terms)} + {len(st)} products" + (f"; first: - s_{k - 1} p_0 (the low limb of column {k - 1} cancels) and its carry" if cancel else ""))
        units = []
        if cancel:
            units.append(("v_mad_i64_i32 %0, %1, @0, -1, @1", [f'"v"(m{k - 1})', '"v"(col)']))
            units.append((f"v_ashrrev_i64 %0, {W}, %0", []))
        for t, (x, y) in enumerate(terms):
            addend = "0" if k == 0 and t == 0 else "%0"
            if isinstance(y, int):
                units.append((f"v_mad_u64_u32 %0, %1, @0, {y}, {addend}", [f'"v"({x})']))
            else:
                units.append((f"v_mad_u64_u32 %0, %1, @0, @1, {addend}", [f'"v"({x})', f'"v"({y})']))
        for x, y in st:
            units.append(("v_mad_i64_i32 %0, %1, @0, @1, %0", [f'"v"({x})', f'"s"({y})']))      # the negated prime limb rides the constant bus: no VGPR
        if cancel:
            out += _pack(units, "=&v", "nc")
            out.append("    col = nc;")
        else:
            out += _pack(units, "=&v" if k == 0 else "+&v", "col")
        if k < L:
            out.append(f"    m{k} = (uint32_t)col;" if k < L - 1 else f"    m{k} = ((uint32_t)col & M29) | 0xC0000000u;        // (col & M29) - 2^30: the one digit with a fixed sign")
        else:
            out.append(f"    r.v[{k - L}] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> {W});")
    out.append(f"    r.v[{L - 1}] = (uint32_t)col;" if hi is None else f"    r.v[{L - 1}] = (uint32_t)col + {hi}.v[{L - 1}];")
    out.append("    return r;")
    return out


def _mul_terms(k):
    for i in range(L):
        j = k - i
        if 0 <= j < L:
            yield (f"a.v[{i}]", f"b.v[{j}]")


def _sqr_terms(k):
    for i in range(L):
        j = k - i
        if 0 <= j < L and i < j:
            yield (f"d{i}", f"a.v[{j}]")
    if k % 2 == 0 and k // 2 < L:
        yield (f"a.v[{k // 2}]", f"a.v[{k // 2}]")


_DBL = "    const uint32_t " + ", ".join(f"d{i} = a.v[{i}] << 1" for i in range(L - 1)) + ";"


def emit_signed():
    out = ["template <int F> __device__ __forceinline__ fe29_t fe29_mul_sg(const fe29_t &a, const fe29_t &b) {"] + body_sg(_mul_terms) + ["}"]
    out += ["template <int F> __device__ __forceinline__ fe29_t fe29_sqr_sg(const fe29_t &a) {", _DBL] + body_sg(_sqr_terms) + ["}"]
    out += ["template <int F> __device__ __forceinline__ fe29_t fe29_mul_hi_sg(const fe29_t &a, const fe29_t &b, const fe29_t &h) {"] + body_sg(_mul_terms, hi="h") + ["}"]
    out += ["template <int F> __device__ __forceinline__ fe29_t fe29_sqr_hi_sg(const fe29_t &a, const fe29_t &h) {", _DBL] + body_sg(_sqr_terms, hi="h") + ["}"]
    # the Poseidon rows: n products and the round constant (stored times 2^261, added before the reduction) in ONE signed reduction
    def dot_rc(n):
        def terms(k):
            for t in range(n):
                for i in range(L):
                    j = k - i
                    if 0 <= j < L:
                        yield (f"a{t}.v[{i}]", f"b{t}.v[{j}]") if n > 1 else (f"a.v[{i}]", f"b.v[{j}]")
            if k < L:
                yield (f"c.v[{k}]", 1)
        return terms
    out += ["template <int F> __device__ __forceinline__ fe29_t fe29_mulrc_sg(const fe29_t &a, const fe29_t &b, const fe29_t &c) {"] + body_sg(dot_rc(1)) + ["}"]
    out += ["template <int F> __device__ __forceinline__ fe29_t fe29_dot2rc_sg(const fe29_t &a0, const fe29_t &b0, const fe29_t &a1, const fe29_t &b1, const fe29_t &c) {"] + body_sg(dot_rc(2)) + ["}"]
    out += ["template <int F> __device__ __forceinline__ fe29_t fe29_dot3rc_sg(const fe29_t &a0, const fe29_t &b0, const fe29_t &a1, const fe29_t &b1, const fe29_t &a2, const fe29_t &b2, const fe29_t &c) {"] + body_sg(dot_rc(3)) + ["}"]
    return out


def emit_mul(lazy=False):
    def terms(k):
        for i in range(L):
            j = k - i
            if 0 <= j < L:
                yield (f"a.v[{i}]", f"b.v[{j}]")
    return [f"template <int F> __device__ __forceinline__ fe29_t fe29_mul_{'lz' if lazy else 'asm'}(const fe29_t &a, const fe29_t &b) {{"] + body(terms, lazy) + ["}"]


def emit_mulrc_lz():
    def terms(k):
        for i in range(L):
            j = k - i
            if 0 <= j < L:
                yield (f"a.v[{i}]", f"b.v[{j}]")
        if k < L:
            yield (f"c.v[{k}]", 1)
    return ["template <int F> __device__ __forceinline__ fe29_t fe29_mulrc_lz(const fe29_t &a, const fe29_t &b, const fe29_t &c) {"] + body(terms, True) + ["}"]


def emit_dot2rc_lz():
    def terms(k):
        for t in range(2):
            for i in range(L):
                j = k - i
                if 0 <= j < L:
                    yield (f"a{t}.v[{i}]", f"b{t}.v[{j}]")
        if k < L:
            yield (f"c.v[{k}]", 1)
    return ["template <int F> __device__ __forceinline__ fe29_t fe29_dot2rc_lz(const fe29_t &a0, const fe29_t &b0, const fe29_t &a1, const fe29_t &b1, const fe29_t &c) {"] + body(terms, True) + ["}"]


def emit_sqr(lazy=False):
    def terms(k):
        for i in range(L):
            j = k - i
            if 0 <= j < L and i < j:
                yield (f"d{i}", f"a.v[{j}]")
        if k % 2 == 0 and k // 2 < L:
            yield (f"a.v[{k // 2}]", f"a.v[{k // 2}]")
    pre = [f"template <int F> __device__ __forceinline__ fe29_t fe29_sqr_{'lz' if lazy else 'asm'}(const fe29_t &a) {{",
           "    const uint32_t " + ", ".join(f"d{i} = a.v[{i}] << 1" for i in range(L - 1)) + ";   // limbs < 2^29: the doubled ones fit 32 bits"]
    return pre + body(terms, lazy) + ["}"]


def emit_sqr_hi():
    def terms(k):
        for i in range(L):
            j = k - i
            if 0 <= j < L and i < j:
                yield (f"d{i}", f"a.v[{j}]")
        if k % 2 == 0 and k // 2 < L:
            yield (f"a.v[{k // 2}]", f"a.v[{k // 2}]")
    pre = ["template <int F> __device__ __forceinline__ fe29_t fe29_sqr_hi_asm(const fe29_t &a, const fe29_t &h) {",
           "    const uint32_t " + ", ".join(f"d{i} = a.v[{i}] << 1" for i in range(L - 1)) + ";"]
    return pre + body(terms, False, hi="h", fresh=True) + ["}"]


def emit_mul_hi(lazy=False):
    def terms(k):
        for i in range(L):
            j = k - i
            if 0 <= j < L:
                yield (f"a.v[{i}]", f"b.v[{j}]")
    return [f"template <int F> __device__ __forceinline__ fe29_t fe29_mul_hi_{'lz' if lazy else 'asm'}(const fe29_t &a, const fe29_t &b, const fe29_t &h) {{"] + body(terms, lazy, hi="h", fresh=True) + ["}"]


def emit_dot3rc_lz():
    def terms(k):
        for t in range(3):
            for i in range(L):
                j = k - i
                if 0 <= j < L:
                    yield (f"a{t}.v[{i}]", f"b{t}.v[{j}]")
        if k < L:
            yield (f"c.v[{k}]", 1)
    return ["template <int F> __device__ __forceinline__ fe29_t fe29_dot3rc_lz(const fe29_t &a0, const fe29_t &b0, const fe29_t &a1, const fe29_t &b1, const fe29_t &a2, const fe29_t &b2, const fe29_t &c) {"] + body(terms, True) + ["}"]


def emit_dot3():
    def terms(k):
        for t in range(3):
            for i in range(L):
                j = k - i
                if 0 <= j < L:
                    yield (f"a{t}.v[{i}]", f"b{t}.v[{j}]")
    return ["template <int F> __device__ __forceinline__ fe29_t fe29_dot3_asm(const fe29_t &a0, const fe29_t &b0, const fe29_t &a1, const fe29_t &b1, const fe29_t &a2, const fe29_t &b2) {"] + body(terms) + ["}"]


def emit_dot2():
    def terms(k):
        for t in range(2):
            for i in range(L):
                j = k - i
                if 0 <= j < L:
                    yield (f"a{t}.v[{i}]", f"b{t}.v[{j}]")
    return ["template <int F> __device__ __forceinline__ fe29_t fe29_dot2_asm(const fe29_t &a0, const fe29_t &b0, const fe29_t &a1, const fe29_t &b1) {"] + body(terms) + ["}"]


def generated():
    return "\n".join(["// ---- GENERATED by tools/gen_fe29.py: do not edit by hand"] + emit_mul() + emit_sqr() + emit_dot2() + emit_dot3() + emit_mul(True) + emit_sqr(True) + emit_dot3rc_lz() + emit_mulrc_lz() + emit_dot2rc_lz() + emit_sqr_hi() + emit_mul_hi() + emit_mul_hi(True) + emit_signed() + ["// ---- END GENERATED"]) + "\n"


def proven_constants():
    """tools/fe29_bounds.py: the interval proofs of every routine above AND of their callers' value discipline (the XYZZ mixed add of ec29.cuh, the Poseidon lane
    forms of sponge.cuh).  Raises fe29_bounds.BoundError -- nothing is written -- when a column can reach 2^64, a limb-wise "K p - b" can go negative in a limb,
    a top limb can outgrow its register or an invariant is not a fixed point.  The constants the proofs ran with are emitted as C++ for the callers to use BY NAME."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fe29_bounds
    return "// ---- PROVEN CONSTANTS (tools/gen_fe29.py <- tools/fe29_bounds.py): do not edit by hand\n" + fe29_bounds.emit_constants() + "\n// ---- END PROVEN CONSTANTS\n"


def rewrite(s: str) -> str:
    """fp29.cuh with both generated regions replaced"""
    text, consts = generated(), proven_constants()
    a, b = s.index("// ---- GENERATED by tools/gen_fe29.py"), s.index("// ---- END GENERATED")
    s = s[:a] + text.rstrip("\n") + s[b + len("// ---- END GENERATED"):]
    a, b = s.index("// ---- PROVEN CONSTANTS"), s.index("// ---- END PROVEN CONSTANTS")
    return s[:a] + consts.rstrip("\n") + s[b + len("// ---- END PROVEN CONSTANTS"):]


if __name__ == "__main__":
    if "--write" in sys.argv:
        p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mina_bridge_amd", "csrc", "fp29.cuh")
        new = rewrite(open(p).read())                              # a BoundError from the proofs is raised here: the file is only opened for writing afterwards
        open(p, "w").write(new)
    else:
        sys.stdout.write(proven_constants() + generated())
