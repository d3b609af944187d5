"""kimchi `verifier::{oracles, to_batch}` -- CPU restatement (TEST INFRASTRUCTURE ONLY) + a miniature prover that mints accepting
instances on a SYNTHETIC verifier index (transcript order: README.md:413-475).

[UPSTREAM-RECALL] kimchi / poly-commitment / mina-poseidon are not vendored (pins core/Cargo.toml:14-16) and the tree holds neither
a verifier index nor a proof, so nothing here is pinned by reference bytes.  What pins it: self-consistency -- the prover below
builds w/z/t polynomials that satisfy the permutation argument + a gate expressed as a PolishToken program, and the verifier
restated here (Fiat-Shamir order, `ft_eval0`, `perm_scalars`, chunked `ft_comm`, evaluation list order, combined inner product)
accepts exactly those proofs and rejects any tampered evaluation / commitment / public input.

The real blockchain-snark index (domain, shifts, sigma/coefficient/selector commitments, linearization token program) is data the
engine takes as a parameter (`mina_verifier_index`, include/mina_verify.h).
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field

import numpy as np

from . import ipa_ref as I
from . import oracle as O
from . import pasta_ref as R

COLUMNS, PERMUTS = 15, 7
# PolishToken opcodes of the engine's byte-code (include/mina_verify.h MINA_TOK_*)
(T_ALPHA, T_BETA, T_GAMMA, T_JOINT, T_ENDO, T_MDS, T_LITERAL, T_CELL, T_DUP, T_POW, T_ADD, T_MUL, T_SUB, T_VANISH_ZK, T_LAGRANGE, T_STORE, T_LOAD,
 T_SKIP_IF, T_SKIP_IF_NOT) = range(19)


def feature_mask(flags) -> int:
    """kimchi `FeatureFlag`s of a proof from the eight Pickles statement flags (range_check0, range_check1, foreign_field_add, foreign_field_mul,
    xor, rot, lookup, runtime_tables) [UPSTREAM-RECALL]: codes 0..5 the optional gates, 6 LookupTables, 7 RuntimeLookupTables, 8..11 LookupPattern
    Xor / Lookup / RangeCheck / ForeignFieldMul, 12 + w TableWidth(w), 16 + n LookupsPerRow(n) (the maxima over the patterns in use:
    Xor 3 / 4, Lookup 2 / 3, RangeCheck 1 / 4, ForeignFieldMul 2 / 2)"""
    rc0, rc1, ffadd, ffmul, x, rot, lk, rt = (bool(f) for f in flags)
    pats = {"xor": x, "lookup": lk, "rc": rc0 or rc1 or rot, "ffmul": ffmul}
    any_ = any(pats.values())
    width = max([w for p, w in (("xor", 3), ("lookup", 2), ("rc", 1), ("ffmul", 2)) if pats[p]], default=0)
    per_row = max([n for p, n in (("xor", 4), ("lookup", 3), ("rc", 4), ("ffmul", 2)) if pats[p]], default=0)
    m = sum(1 << i for i, f in enumerate((rc0, rc1, ffadd, ffmul, x, rot)) if f)
    m |= (any_ << 6) | (rt << 7) | (pats["xor"] << 8) | (pats["lookup"] << 9) | (pats["rc"] << 10) | (pats["ffmul"] << 11)
    for w in range(4):
        if width >= w and (w == 0 or any_): m |= 1 << (12 + w)
    for n in range(5):
        if per_row >= n and (n == 0 or any_): m |= 1 << (16 + n)
    return m
# evaluation columns, in the order the transcript absorbs them / the evaluation list names them
COL_Z, COL_GENERIC, COL_POSEIDON, COL_COMPLETE_ADD, COL_MUL, COL_EMUL, COL_ENDOMUL_SCALAR = range(7)
COL_W0, COL_COEFF0, COL_S0 = 7, 7 + COLUMNS, 7 + 2 * COLUMNS
N_EVAL_COLS = 7 + 2 * COLUMNS + (PERMUTS - 1)            # 43


@dataclass
class VerifierIndex:
    curve: int
    log2_domain: int
    zk_rows: int
    shifts: list
    sigma_comm: list                  # 7 points
    coefficients_comm: list           # 15
    selector_comm: list               # generic, poseidon, complete_add, mul, emul, endomul_scalar
    constant_term: list               # token program: list of tuples (op, *operands)
    perm_alpha_offset: int = 21
    mds: list = field(default_factory=list)           # 3x3 of the SCALAR field's Poseidon (Constants.mds)
    digest: int = 0

    @property
    def n(self):
        return 1 << self.log2_domain


def fr_sponge(curve, pp_scalar):
    """kimchi `DefaultFrSponge`: a Poseidon sponge over the proof's scalar field = the Fq-sponge machinery of the OTHER curve"""
    return I.FqSponge(1 - curve, pp_scalar)


def index_digest(index: VerifierIndex, pp_base) -> int:
    """`VerifierIndex::digest`: Fq-sponge over the index commitments, squeezed as a base-field element"""
    sp = I.FqSponge(index.curve, pp_base)
    sp.absorb_g(index.sigma_comm); sp.absorb_g(index.coefficients_comm); sp.absorb_g(index.selector_comm)
    return sp.challenge_fq()


# ------------------------------------------------------------------------------------------------ PolishToken interpreter
def zk_polynomial_eval(index: VerifierIndex, x: int, r: int) -> int:
    """permutation_vanishing_polynomial_m: prod over the last zk_rows rows of (x - w^i)"""
    w = O_domain_generator(r, index.log2_domain)
    acc = 1
    for i in range(index.n - index.zk_rows, index.n):
        acc = acc * (x - pow(w, i, r)) % r
    return acc


def O_domain_generator(r, log2):
    return pow(R.two_adic_root_of_unity(r), 1 << (32 - log2), r)


def polish_evaluate(tokens, index: VerifierIndex, pt: int, evals, consts: dict, r: int) -> int:
    """evals[col] = (at zeta, at zeta*omega); tokens = [(op, ...)]"""
    stack, cache = [], []
    n = index.n
    w = O_domain_generator(r, index.log2_domain)
    features, present, skip = consts.get("features", 0), consts.get("present"), 0
    for tok in tokens:
        op = tok[0]
        if skip:                                                   # inside a skipped region: nothing runs (kimchi PolishToken::evaluate: `skip_count -= 1; continue`) --
            skip -= 1                                              # a skipped STORE takes no cache slot
            continue
        if op in (T_SKIP_IF, T_SKIP_IF_NOT):                       # kimchi SkipIf / SkipIfNot: push zero and skip `count` tokens when the condition holds
            on = bool((features >> tok[1]) & 1)
            if on == (op == T_SKIP_IF):
                skip = tok[2]; stack.append(0)
            continue
        if op == T_ALPHA: stack.append(consts["alpha"])
        elif op == T_BETA: stack.append(consts["beta"])
        elif op == T_GAMMA: stack.append(consts["gamma"])
        elif op == T_JOINT: stack.append(consts.get("joint_combiner", 0))
        elif op == T_ENDO: stack.append(consts["endo"])
        elif op == T_MDS: stack.append(consts["mds"][tok[1]][tok[2]])
        elif op == T_LITERAL: stack.append(tok[1] % r)
        elif op == T_CELL:
            col = tok[1]
            if present is not None and col >= N_EVAL_COLS:        # the step side: optional SLOT col - 43, found through the proof's presence mask
                slot = col - N_EVAL_COLS
                if not (present >> slot) & 1: raise KeyError("the program names an optional evaluation the proof does not carry")
                col = N_EVAL_COLS + bin(present & ((1 << slot) - 1)).count("1")
            stack.append(evals[col][tok[2]])
        elif op == T_DUP: stack.append(stack[-1])
        elif op == T_POW: stack.append(pow(stack.pop(), tok[1], r))
        elif op == T_ADD: b = stack.pop(); a = stack.pop(); stack.append((a + b) % r)
        elif op == T_MUL: b = stack.pop(); a = stack.pop(); stack.append(a * b % r)
        elif op == T_SUB: b = stack.pop(); a = stack.pop(); stack.append((a - b) % r)
        elif op == T_VANISH_ZK: stack.append(zk_polynomial_eval(index, pt, r))
        elif op == T_LAGRANGE:                                 # unnormalized Lagrange basis of row `offset` (negative: counted from the zk rows)
            off = tok[1]
            i = off if off >= 0 else n - index.zk_rows + (0 if off == -(1 << 31) else off)     # -2^31: the first zero-knowledge row itself
            stack.append((pow(pt, n, r) - 1) * R.inv((pt - pow(w, i, r)) % r, r) % r)
        elif op == T_STORE: cache.append(stack[-1])
        elif op == T_LOAD:
            if tok[1] >= len(cache): raise KeyError("LOAD of a cache slot no executed STORE has filled")
            stack.append(cache[tok[1]])
        else: raise ValueError("unknown token %r" % (tok,))
    assert len(stack) == 1
    return stack[0]


# ------------------------------------------------------------------------------------------------ verifier: oracles + to_batch
def oracles_and_batch(index: VerifierIndex, proof: dict, public_inputs, pp_base, pp_scalar, g_bytes, h):
    """proof: {w_comm[15], z_comm, t_comm[7], evals[43] of (zeta, zeta_omega), ft_eval1, prev: [(chals, comm)], opening}.
    Returns every intermediate (`o`) and the `ipa_verify_batch` entry."""
    curve = index.curve
    r, m = R.scalar_modulus(curve), R.base_modulus(curve)
    endo_r = R.endo_r(curve)
    n, k = index.n, index.log2_domain
    w = O_domain_generator(r, k)
    o = {}
    # public-input commitment (negated public polynomial, blinder 1)
    from . import state_job_ref as J
    public_comm = J.public_input_commitment(curve, g_bytes, h, k, public_inputs)
    o["public_comm"] = public_comm
    fq = I.FqSponge(curve, pp_base)
    fq.absorb_fq([index.digest])
    for chals, comm in proof["prev"]:
        fq.absorb_g([comm])
    fq.absorb_g([public_comm])
    fq.absorb_g(proof["w_comm"])
    beta = fq.challenge(); gamma = fq.challenge()
    fq.absorb_g([proof["z_comm"]])
    alpha = R.challenge_to_field(fq.challenge(), endo_r, r)
    fq.absorb_g(proof["t_comm"])
    zeta = R.challenge_to_field(fq.challenge(), endo_r, r)
    o.update(beta=beta, gamma=gamma, alpha=alpha, zeta=zeta)
    fq_after = fq.clone()
    dg = fq.clone().challenge_fq()
    digest = dg if dg < r else 0
    fr = fr_sponge(curve, pp_scalar)
    fr.absorb_fq([digest])
    pf = fr_sponge(curve, pp_scalar)
    for chals, _ in proof["prev"]:
        pf.absorb_fq(chals)
    fr.absorb_fq([pf.challenge_fq()])
    zeta1 = pow(zeta, n, r); zetaw = zeta * w % r
    # negated public polynomial at zeta, zeta*omega
    def pub_eval(x):
        if not public_inputs:
            return 0
        acc = 0
        for i, p in enumerate(public_inputs):
            wi = pow(w, i, r)
            acc = (acc - R.inv((x - wi) % r, r) * p % r * wi) % r
        return acc * (pow(x, n, r) - 1) % r * R.inv(n, r) % r
    public_evals = (pub_eval(zeta), pub_eval(zetaw))
    o["public_evals"] = public_evals
    ev = proof["evals"]
    fr.absorb_fq([proof["ft_eval1"]])
    fr.absorb_fq([public_evals[0]]); fr.absorb_fq([public_evals[1]])
    for c in range(N_EVAL_COLS):
        fr.absorb_fq([ev[c][0]]); fr.absorb_fq([ev[c][1]])
    endo_own = R.endo_r(curve)                                  # Fr-sponge challenges expand with the proof curve's endo_r
    v = R.challenge_to_field(fr.challenge(), endo_own, r)
    u = R.challenge_to_field(fr.challenge(), endo_own, r)
    o.update(v=v, u=u)
    # ft_eval0
    a0, a1, a2 = (pow(alpha, index.perm_alpha_offset + i, r) for i in range(3))
    zkpm = zk_polynomial_eval(index, zeta, r)
    W = lambda i: ev[COL_W0 + i][0]
    S = lambda i: ev[COL_S0 + i][0]
    z0, z1 = ev[COL_Z]
    ft = (W(PERMUTS - 1) + gamma) * z1 % r * a0 % r * zkpm % r
    for i in range(PERMUTS - 1):
        ft = ft * ((beta * S(i) + W(i) + gamma) % r) % r
    ft = (ft - public_evals[0]) % r
    t2 = a0 * zkpm % r * z0 % r
    for i in range(PERMUTS):
        t2 = t2 * ((gamma + beta * zeta % r * index.shifts[i] + W(i)) % r) % r
    ft = (ft - t2) % r
    wz = pow(w, n - index.zk_rows, r)
    num = ((zeta1 - 1) * a1 % r * (zeta - wz) + (zeta1 - 1) * a2 % r * (zeta - 1)) % r * ((1 - z0) % r) % r
    den = R.inv((zeta - wz) * (zeta - 1) % r, r)
    ft = (ft + num * den) % r
    consts = {"alpha": alpha, "beta": beta, "gamma": gamma, "endo": R.endo_q(1 - curve), "mds": index.mds}
    ft = (ft - polish_evaluate(index.constant_term, index, zeta, ev, consts, r)) % r
    o["ft_eval0"] = ft
    # perm scalar, f_comm, chunked ft_comm
    ps = z1 * beta % r * a0 % r * zkpm % r
    for i in range(PERMUTS - 1):
        ps = ps * ((gamma + beta * S(i) + W(i)) % r) % r
    ps = (-ps) % r
    o["perm_scalar"] = ps
    f_comm = R.scalar_mul(ps, index.sigma_comm[PERMUTS - 1], m)
    tc = None
    for t in reversed(proof["t_comm"]):
        tc = R.add(R.scalar_mul(zeta1, tc, m) if tc is not None else None, t, m)
    ft_comm = R.add(f_comm, R.neg(R.scalar_mul((zeta1 - 1) % r, tc, m), m) if tc is not None else None, m)
    o["ft_comm"] = ft_comm
    # evaluation list
    comms, evals = [], []
    for chals, comm in proof["prev"]:
        comms.append(comm); evals.append((R.b_poly(chals, zeta, r), R.b_poly(chals, zetaw, r)))
    comms.append(public_comm); evals.append(public_evals)
    comms.append(ft_comm); evals.append((ft, proof["ft_eval1"]))
    col_comms = [proof["z_comm"]] + list(index.selector_comm) + list(proof["w_comm"]) + list(index.coefficients_comm) + list(index.sigma_comm[: PERMUTS - 1])
    for c in range(N_EVAL_COLS):
        comms.append(col_comms[c]); evals.append(ev[c])
    cip = I.combined_inner_product([list(e) for e in evals], v, u, r)
    o.update(combined_inner_product=cip, comms=comms, evaluations=evals)
    entry = {"sponge": fq_after, "evalpoints": [zeta, zetaw], "polyscale": v, "evalscale": u, "comms": comms, "opening": proof["opening"],
             "combined_inner_product": cip, "k": k}
    o["sponge_after"] = fq_after
    return o, entry


# ------------------------------------------------------------------------------------------------ miniature prover (synthetic circuit)
def _fft(a, w, r):
    """in-place-style radix-2 evaluation of the coefficient list a (len power of two) on <w>"""
    n = len(a)
    if n == 1:
        return a
    even = _fft(a[0::2], w * w % r, r); odd = _fft(a[1::2], w * w % r, r)
    out = [0] * n
    x = 1
    for i in range(n // 2):
        t = x * odd[i] % r
        out[i] = (even[i] + t) % r; out[i + n // 2] = (even[i] - t) % r
        x = x * w % r
    return out


def _ifft(vals, w, r):
    n = len(vals)
    c = _fft(vals, R.inv(w, r), r)
    ninv = R.inv(n, r)
    return [x * ninv % r for x in c]


def _peval(c, x, r):
    acc = 0
    for a in reversed(c):
        acc = (acc * x + a) % r
    return acc


@dataclass
class SyntheticCircuit:
    """the prover-side secret of a synthetic index: coefficient / selector / sigma columns and their polynomials"""
    index: VerifierIndex
    npub: int
    coef: list
    sel: list
    sigma: list
    cycles: list
    coef_p: list
    sel_p: list
    sig_p: list


def synthetic_circuit(curve, g_bytes, h, pp_base, pp_scalar, log2_domain: int, npub: int, seed: int) -> SyntheticCircuit:
    """A toy circuit over 2^log2_domain rows and its verifier index.  Rows i < npub: generic_selector = 1, c0 = 1, w0 = public[i];
    other usable rows: gate  c0 w0 + c1 w1 w2 - c2 w3 = 0  (expressed to the verifier as a PolishToken program over the evaluations);
    columns 4..6 carry copy constraints (cycles of cells forced equal); max_poly_size = domain size (t has 7 chunks)."""
    rng = random.Random(seed)
    r = R.scalar_modulus(curve)
    k = log2_domain; n = 1 << k; zk = 3
    w = O_domain_generator(r, k)
    dom = [pow(w, i, r) for i in range(n)]
    usable = n - zk
    coef = [[rng.randrange(r) for _ in range(n)] for _ in range(COLUMNS)]
    sel = [[0] * n for _ in range(6)]
    for i in range(usable):
        sel[0][i] = 1
        if i < npub:
            coef[0][i] = 1; coef[1][i] = 0; coef[2][i] = 0
        elif coef[2][i] == 0:
            coef[2][i] = 1
    for c in range(1, 6):                                        # the other selectors: random on the domain, unused by the gate
        sel[c] = [rng.randrange(r) for _ in range(n)]
    shifts = [1]
    while len(shifts) < PERMUTS:
        shifts.append(rng.randrange(2, r))
    sigma = [[dom[i] * shifts[c] % r for i in range(n)] for c in range(PERMUTS)]      # identity permutation, then cycles
    cycles = []
    for _ in range(max(2, usable // 3)):
        cells = list(dict.fromkeys((rng.randrange(4, PERMUTS), rng.randrange(usable)) for _ in range(rng.randrange(2, 4))))
        if len(cells) < 2 or any(sigma[c][i] != dom[i] * shifts[c] % r for c, i in cells):
            continue
        cycles.append(cells)
        for (c, i), (c2, i2) in zip(cells, cells[1:] + cells[:1]):
            sigma[c][i] = dom[i2] * shifts[c2] % r
    P = lambda vals: _ifft(vals, w, r)
    coef_p = [P(c) for c in coef]; sel_p = [P(c) for c in sel]; sig_p = [P(c) for c in sigma]
    commit = lambda poly: I.commit(curve, g_bytes[:n], h, poly, 0)
    index = VerifierIndex(curve=curve, log2_domain=k, zk_rows=zk, shifts=shifts,
                          sigma_comm=[commit(p) for p in sig_p], coefficients_comm=[commit(p) for p in coef_p], selector_comm=[commit(p) for p in sel_p],
                          constant_term=[(T_CELL, COL_GENERIC, 0), (T_CELL, COL_COEFF0 + 0, 0), (T_CELL, COL_W0 + 0, 0), (T_MUL,),
                                         (T_CELL, COL_COEFF0 + 1, 0), (T_CELL, COL_W0 + 1, 0), (T_MUL,), (T_CELL, COL_W0 + 2, 0), (T_MUL,), (T_ADD,),
                                         (T_CELL, COL_W0 + 3, 0), (T_CELL, COL_COEFF0 + 2, 0), (T_MUL,), (T_SUB,), (T_MUL,),
                                         # exercise the rest of the instruction set with terms that cancel:  + x - x
                                         (T_ALPHA,), (T_BETA,), (T_MUL,), (T_GAMMA,), (T_ADD,), (T_ENDO,), (T_MUL,), (T_MDS, 1, 2), (T_ADD,), (T_LITERAL, 12345), (T_ADD,),
                                         (T_VANISH_ZK,), (T_MUL,), (T_LAGRANGE, 0), (T_ADD,), (T_LAGRANGE, -1), (T_SUB,), (T_POW, 3), (T_STORE,), (T_DUP,), (T_SUB,),
                                         (T_LOAD, 0), (T_ADD,), (T_LOAD, 0), (T_SUB,), (T_ADD,)],
                          mds=[list(row) for row in pp_scalar.mds])
    index.digest = index_digest(index, pp_base)
    return SyntheticCircuit(index, npub, coef, sel, sigma, cycles, coef_p, sel_p, sig_p)


def synthetic_proof(circ: SyntheticCircuit, g_bytes, h, pp_base, pp_scalar, public_inputs, seed: int, n_prev: int = 2, prev_chals=None) -> dict:
    """An ACCEPTING kimchi-shaped proof for `circ` (miniature prover: witness, permutation accumulator, 7-chunk quotient on an 8n coset,
    evaluations, ft polynomial, recursion challenges with their b_poly commitments, the aggregated opening)."""
    rng = random.Random(seed)
    index = circ.index
    curve = index.curve
    r = R.scalar_modulus(curve)
    k = index.log2_domain; n = 1 << k; zk = index.zk_rows
    w = O_domain_generator(r, k)
    dom = [pow(w, i, r) for i in range(n)]
    npub = len(public_inputs)
    assert npub == circ.npub
    usable = n - zk
    coef, sel, sigma, shifts = circ.coef, circ.sel, circ.sigma, index.shifts
    wit = [[rng.randrange(r) for _ in range(n)] for _ in range(COLUMNS)]
    for i in range(usable):
        if i < npub:
            wit[0][i] = public_inputs[i]
        else:
            wit[3][i] = (coef[0][i] * wit[0][i] + coef[1][i] * wit[1][i] % r * wit[2][i]) % r * R.inv(coef[2][i], r) % r
    for cells in circ.cycles:
        for c, i in cells:
            wit[c][i] = wit[cells[0][0]][cells[0][1]]
    P = lambda vals: _ifft(vals, w, r)
    wit_p = [P(c) for c in wit]
    coef_p, sel_p, sig_p = circ.coef_p, circ.sel_p, circ.sig_p
    blind = lambda: rng.randrange(r)
    hp = h
    commit = lambda poly, bl: I.commit(curve, g_bytes[:n], hp, poly, bl)
    wb = [blind() for _ in range(COLUMNS)]
    w_comm = [commit(p, b) for p, b in zip(wit_p, wb)]
    prev = []
    for pi in range(n_prev):
        chals = list(prev_chals[pi]) if prev_chals is not None else [rng.randrange(r) for _ in range(k)]
        sc = [O.le_to_int(x) for x in O.b_poly_coefficients(O.scalar_field_of(curve), O.ints_to_le(chals))]
        prev.append((chals, sc, I.commit(curve, g_bytes[:n], hp, sc, 0)))
    # --- Fiat-Shamir up to beta, gamma
    from . import state_job_ref as J
    pub_p = J.public_poly_coeffs(curve, k, public_inputs)
    public_comm = I.commit(curve, g_bytes[:n], hp, pub_p, 1)
    endo_r = R.endo_r(curve)
    fq = I.FqSponge(curve, pp_base)
    fq.absorb_fq([index.digest])
    for _, _, comm in prev:
        fq.absorb_g([comm])
    fq.absorb_g([public_comm]); fq.absorb_g(w_comm)
    beta = fq.challenge(); gamma = fq.challenge()
    # --- permutation accumulator z (z[0] = 1, z[n - zk] = 1, random in the zk rows)
    z = [1] * n
    for i in range(usable):
        num = den = 1
        for c in range(PERMUTS):
            num = num * ((wit[c][i] + beta * dom[i] % r * shifts[c] + gamma) % r) % r
            den = den * ((wit[c][i] + beta * sigma[c][i] + gamma) % r) % r
        z[i + 1] = z[i] * num % r * R.inv(den, r) % r
    assert z[usable] == 1, "copy constraints not satisfied"
    for i in range(usable + 1, n):
        z[i] = rng.randrange(r)
    z_p = P(z); zb = blind()
    z_comm = commit(z_p, zb)
    fq.absorb_g([z_comm])
    alpha = R.challenge_to_field(fq.challenge(), endo_r, r)
    a0, a1, a2 = (pow(alpha, index.perm_alpha_offset + i, r) for i in range(3))
    # --- quotient on a coset of size 8n
    big = 8 * n
    wb8 = O_domain_generator(r, k + 3)
    shift = 7                                                    # coset generator (any non-residue of the 8n-th roots)
    xs = [shift * pow(wb8, i, r) % r for i in range(big)]
    def on_coset(poly):
        c = [a * pow(shift, i, r) % r for i, a in enumerate(poly)] + [0] * (big - len(poly))
        return _fft(c, wb8, r)
    Wp = [on_coset(p) for p in wit_p[:PERMUTS]]
    C8 = [on_coset(p) for p in coef_p[:3]]
    S8 = [on_coset(p) for p in sig_p]
    G8 = on_coset(sel_p[0]); Z8 = on_coset(z_p); PUB8 = on_coset(pub_p)
    wz = pow(w, n - zk, r)
    consts = {"alpha": alpha, "beta": beta, "gamma": gamma, "endo": R.endo_q(1 - curve), "mds": index.mds}
    tq = [0] * big
    stride = big // n
    for j, x in enumerate(xs):
        zw = Z8[(j + stride) % big]                               # z(x * omega)
        zkpm = 1
        for i in range(n - zk, n):
            zkpm = zkpm * (x - dom[i]) % r
        pa = Z8[j]; pb = zw
        for c in range(PERMUTS):
            pa = pa * ((gamma + beta * x % r * shifts[c] + Wp[c][j]) % r) % r
            pb = pb * ((gamma + beta * S8[c][j] + Wp[c][j]) % r) % r
        gate = G8[j] * ((C8[0][j] * Wp[0][j] + C8[1][j] * Wp[1][j] % r * Wp[2][j] - Wp[3][j] * C8[2][j]) % r) % r
        xn1 = (pow(x, n, r) - 1) % r
        bnd = xn1 * ((1 - Z8[j]) % r) % r * ((a1 * R.inv((x - 1) % r, r) + a2 * R.inv((x - wz) % r, r)) % r) % r
        num = (a0 * zkpm % r * ((pa - pb) % r) + PUB8[j] + gate - bnd) % r
        tq[j] = num * R.inv(xn1, r) % r
    t_c = _fft(tq, R.inv(wb8, r), r)
    binv = R.inv(big, r); sinv = R.inv(shift, r)
    t_p = [a * binv % r * pow(sinv, i, r) % r for i, a in enumerate(t_c)]
    assert not any(t_p[7 * n:]), "quotient does not fit 7 chunks: the synthetic witness violates a constraint"
    t_chunks = [t_p[i * n:(i + 1) * n] for i in range(7)]
    tb = [blind() for _ in range(7)]
    t_comm = [commit(p, b) for p, b in zip(t_chunks, tb)]
    fq.absorb_g(t_comm)
    zeta = R.challenge_to_field(fq.challenge(), endo_r, r)
    zetaw = zeta * w % r
    # --- evaluations
    cols_p = [z_p] + sel_p + wit_p + coef_p + sig_p[: PERMUTS - 1]
    cols_b = [zb] + [0] * 6 + wb + [0] * COLUMNS + [0] * (PERMUTS - 1)
    evals = [(_peval(p, zeta, r), _peval(p, zetaw, r)) for p in cols_p]
    # ft polynomial = perm_scalar * sigma6 - (zeta^n - 1) * sum_i zeta^(n i) t_i
    zeta1 = pow(zeta, n, r)
    zkpm = zk_polynomial_eval(index, zeta, r)
    ps = evals[COL_Z][1] * beta % r * a0 % r * zkpm % r
    for i in range(PERMUTS - 1):
        ps = ps * ((gamma + beta * evals[COL_S0 + i][0] + evals[COL_W0 + i][0]) % r) % r
    ps = (-ps) % r
    tcomb, tbl = [0] * n, 0
    for ch, bl in zip(reversed(t_chunks), reversed(tb)):
        tcomb = [(a * zeta1 + b) % r for a, b in zip(tcomb, ch)]; tbl = (tbl * zeta1 + bl) % r
    ft_p = [(ps * s - (zeta1 - 1) * t) % r for s, t in zip(sig_p[PERMUTS - 1], tcomb)]
    ft_b = (-(zeta1 - 1) * tbl) % r
    ft_eval1 = _peval(ft_p, zetaw, r)
    proof = {"w_comm": w_comm, "z_comm": z_comm, "t_comm": t_comm, "evals": evals, "ft_eval1": ft_eval1, "prev": [(c, cm) for c, _, cm in prev]}
    # --- the opening: run the verifier side up to v, u to learn polyscale / evalscale, then open everything at (zeta, zeta*omega)
    proof["opening"] = None
    o, entry = oracles_and_batch(index, proof, public_inputs, pp_base, pp_scalar, g_bytes, h)
    assert o["ft_eval0"] == _peval(ft_p, zeta, r), "verifier's ft_eval0 differs from the prover's ft(zeta)"
    polys = [sc for _, sc, _ in prev] + [pub_p, ft_p] + cols_p
    blinders = [0] * len(prev) + [1, ft_b] + cols_b
    sponge = o["sponge_after"].clone()
    op = I.ipa_open_fast(curve, g_bytes[:n], hp, polys, blinders, [zeta, zetaw], o["v"], o["u"], sponge, rng)
    assert op["combined_inner_product"] == o["combined_inner_product"]
    proof["opening"] = op
    return proof
