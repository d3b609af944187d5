"""Pickles `verify_block` glue around the kimchi step -- CPU restatement (TEST INFRASTRUCTURE ONLY) of what openmina's
`ledger::proofs::verification` does before it calls `kimchi::verifier::verify` on the wrap proof:

  * `compute_deferred_values`: expand the statement's 128-bit challenges through the endomorphism (Fp), run the Tick (Fp) sponge over
    the step proof's evaluations (`prev_evals`) to obtain xi and r, evaluate `ft_eval0` of the STEP proof (permutation + boundary
    - public - the step linearization's constant term, a PolishToken program), `derive_plonk` (perm, zeta^(2^16), zeta^n), the
    combined inner product and b = b_poly(chals, zeta) + r b_poly(chals, zeta omega)
  * the two message digests: `messages_for_next_wrap_proof` (Tock / Fq sponge over the padded old bulletproof challenges and the step
    accumulator sg) and `messages_for_next_step_proof` (Tick / Fp sponge over the wrap index commitments, the application state =
    the protocol-state hash, and each previous accumulator with its challenges)
  * the wrap circuit's public input: the statement packed into 40 Fq elements (5 shifted Fp values, 2 challenges, 3 scalar
    challenges, 3 digests, 16 bulletproof prechallenges, branch data, 8 feature flags, 2 lookup slots)

[UPSTREAM-RECALL]: none of the crates involved is vendored and the tree holds no proof, so this is written from knowledge of the
published code and pinned only by self-consistency (tests mint a wrap proof whose public input is exactly this packing, and any
change to a statement field makes the kimchi step reject).  The STEP index (domain shifts, linearization) is data, like the wrap one.
"""
from __future__ import annotations

from dataclasses import dataclass, field

from . import ipa_ref as I
from . import kimchi_ref as K
from . import pasta_ref as R

P, Q = R.P, R.Q
TICK_ROUNDS, TOCK_ROUNDS, SRS_LENGTH_LOG2, PERM_ALPHA0 = 16, 15, 16, 21
# the order in which the step proof's evaluation pairs enter the Tick sponge and the combined inner product
# (kimchi column order, as `K.COL_*`): z, 6 selectors, 15 w, 15 coefficients, 6 sigma -- then the optional ones that are present


@dataclass
class StepIndex:
    """what `compute_deferred_values` needs to know about the step circuit: data installed by the caller"""
    zk_rows: int
    shifts: dict                      # domain_log2 -> 7 shifts (Fp)
    constant_term: list               # PolishToken program over Fp (same byte-code as the wrap index)
    mds: list = field(default_factory=list)


def endo_fp():
    """endo_r of Vesta: expands the Tick-side (Fp) challenges"""
    return R.endo_r(1)


def endo_fq():
    return R.endo_r(0)


def limbs4_to_int(limbs):
    return sum(int(l) << (64 * i) for i, l in enumerate(limbs))


def shifted_type1(x: int) -> int:
    """`Shifted_value.Type1.of_field`: (x - (2^255 + 1)) / 2 in Fp"""
    return (x - (pow(2, 255, P) + 1)) * R.inv(2, P) % P


def challenge_polynomial(chals, x, m):
    return R.b_poly(chals, x, m)


def tick_sponge(pp_fp):
    return I.FqSponge(0, pp_fp)                    # a Poseidon sponge over Fp


def tock_sponge(pp_fq):
    return I.FqSponge(1, pp_fq)                    # over Fq


def prev_evals_sequence(w: dict):
    """(zeta, zeta*omega) pairs of the step proof's evaluations in kimchi column order; chunked evaluations are combined by the caller"""
    ev = w["prev_evals"]
    seq = [ev[30]] + list(ev[37:43]) + list(ev[0:15]) + list(ev[15:30]) + list(ev[31:37])      # wire order is w, coefficients, z, s, selectors
    seq += [e for e in w["prev_optional"] if e is not None]
    return seq


def combine_chunks(pair, zeta_n, zetaw_n, m):
    """`evals_of_split_evals`: sum_i pt^(n i) chunk_i"""
    def comb(chunks, ptn):
        acc = 0
        for c in reversed(chunks):
            acc = (acc * ptn + c) % m
        return acc
    return comb(pair[0], zeta_n), comb(pair[1], zetaw_n)


def compute_deferred_values(w: dict, step: StepIndex, pp_fp) -> dict:
    """w: wrap-proof dict in the layout of tests/wire_writers.py.  Returns the deferred values (plain Fp integers) and xi / r."""
    e = endo_fp()
    to_f = lambda c: R.challenge_to_field(c, e, P)
    alpha, zeta = to_f(w["alpha"]), to_f(w["zeta"])
    beta, gamma = w["beta"] % P, w["gamma"] % P
    k = w["domain_log2"]
    n = 1 << k
    omega = K.O_domain_generator(P, k)
    zetaw = zeta * omega % P
    zeta_n, zetaw_n = pow(zeta, 1 << SRS_LENGTH_LOG2, P), pow(zetaw, 1 << SRS_LENGTH_LOG2, P)
    seq = [combine_chunks(pr, zeta_n, zetaw_n, P) for pr in prev_evals_sequence(w)]
    bp = [to_f(c) for c in w["bulletproof_challenges"]]
    old = [[to_f(c) for c in row] for row in w["step_old_chals"]]
    # ---- Tick sponge: xi, r
    sp = tick_sponge(pp_fp)
    sp.absorb_fq([limbs4_to_int(w["sponge_digest"]) % P])
    ch = tick_sponge(pp_fp)
    for row in old:
        ch.absorb_fq(row)
    sp.absorb_fq([ch.challenge_fq()])
    sp.absorb_fq([w["prev_ft_eval1"]])
    sp.absorb_fq([w["prev_public_input"][0], w["prev_public_input"][1]])
    for a, b in seq:
        sp.absorb_fq([a]); sp.absorb_fq([b])
    xi_chal = sp.challenge(); r_chal = sp.challenge()
    xi, r = to_f(xi_chal), to_f(r_chal)
    # ---- ft_eval0 of the step proof, derive_plonk
    shifts = step.shifts[k]
    zkp = 1
    for i in range(n - step.zk_rows, n):
        zkp = zkp * (zeta - pow(omega, i, P)) % P
    a0, a1, a2 = (pow(alpha, PERM_ALPHA0 + i, P) for i in range(3))
    W = lambda i: seq[K.COL_W0 + i][0]
    S = lambda i: seq[K.COL_S0 + i][0]
    z0, z1 = seq[K.COL_Z]
    zeta1m1 = (pow(zeta, n, P) - 1) % P
    ft = (W(6) + gamma) * z1 % P * a0 % P * zkp % P
    for i in range(6):
        ft = ft * ((beta * S(i) + W(i) + gamma) % P) % P
    ft = (ft - w["prev_public_input"][0]) % P
    t2 = a0 * zkp % P * z0 % P
    for i in range(7):
        t2 = t2 * ((gamma + beta * zeta % P * shifts[i] + W(i)) % P) % P
    ft = (ft - t2) % P
    wz = pow(omega, n - step.zk_rows, P)
    num = (zeta1m1 * a1 % P * (zeta - wz) + zeta1m1 * a2 % P * (zeta - 1)) % P * ((1 - z0) % P) % P
    ft = (ft + num * R.inv((zeta - wz) * (zeta - 1) % P, P)) % P
    idx = K.VerifierIndex(curve=1, log2_domain=k, zk_rows=step.zk_rows, shifts=shifts, sigma_comm=[], coefficients_comm=[], selector_comm=[],
                          constant_term=step.constant_term, mds=step.mds)
    present = sum(1 << j for j, x in enumerate(w["prev_optional"]) if x is not None)
    consts = {"alpha": alpha, "beta": beta, "gamma": gamma, "endo": R.endo_q(0), "mds": step.mds, "features": K.feature_mask(w["feature_flags"]), "present": present,
              "joint_combiner": to_f(w["joint_combiner"]) if w["joint_combiner"] is not None else 0}
    ft = (ft - K.polish_evaluate(step.constant_term, idx, zeta, seq, consts, P)) % P if step.constant_term else ft
    perm = z1 * beta % P * a0 % P * zkp % P
    for i in range(6):
        perm = perm * ((gamma + beta * S(i) + W(i)) % P) % P
    perm = (-perm) % P
    # ---- combined inner product, b
    def combine(side, ftv, pt):
        v = [challenge_polynomial(c, pt, P) for c in old] + [w["prev_public_input"][side], ftv] + [pr[side] for pr in seq]
        acc = 0
        for x in reversed(v):
            acc = (acc * xi + x) % P
        return acc
    cip = (combine(0, ft, zeta) + r * combine(1, w["prev_ft_eval1"], zetaw)) % P
    b = (challenge_polynomial(bp, zeta, P) + r * challenge_polynomial(bp, zetaw, P)) % P
    return {"alpha": alpha, "zeta": zeta, "xi_chal": xi_chal, "r_chal": r_chal, "xi": xi, "r": r, "ft_eval0": ft, "perm": perm,
            "zeta_to_srs_length": pow(zeta, 1 << SRS_LENGTH_LOG2, P), "zeta_to_domain_size": pow(zeta, n, P), "combined_inner_product": cip, "b": b,
            "bulletproof_challenges": bp}


def hash_messages_for_next_wrap_proof(w: dict, pp_fq) -> int:
    """Tock sponge over the two rows of old bulletproof challenges (expanded in Fq) and the step accumulator sg"""
    e = endo_fq()
    sp = tock_sponge(pp_fq)
    for row in w["old_bulletproof_challenges"]:
        sp.absorb_fq([R.challenge_to_field(c, e, Q) for c in row])
    sg = w["challenge_polynomial_commitment"]
    sp.absorb_fq([sg[0], sg[1]])
    return sp.challenge_fq()


def hash_messages_for_next_step_proof(w: dict, wrap_index_comms, app_state: int, pp_fp) -> int:
    """Tick sponge over the wrap index commitments (sigma 7, coefficients 15, six selectors), the application state (the protocol
    state hash) and each previous wrap accumulator followed by its 16 expanded challenges"""
    e = endo_fp()
    sp = tick_sponge(pp_fp)
    for x, y in wrap_index_comms:
        sp.absorb_fq([x, y])
    sp.absorb_fq([app_state % P])
    for (x, y), row in zip(w["step_comms"], w["step_old_chals"]):
        sp.absorb_fq([x, y])
        sp.absorb_fq([R.challenge_to_field(c, e, P) for c in row])
    return sp.challenge_fq()


def wrap_public_input(w: dict, dv: dict, msg_wrap: int, msg_step: int) -> list:
    """`PreparedStatement::to_public_input(40)`"""
    out = [shifted_type1(dv[name]) for name in ("combined_inner_product", "b", "zeta_to_srs_length", "zeta_to_domain_size", "perm")]
    out += [w["beta"], w["gamma"]]
    out += [w["alpha"], w["zeta"], dv["xi_chal"]]
    out += [limbs4_to_int(w["sponge_digest"]), msg_wrap, msg_step]
    out += list(w["bulletproof_challenges"])
    mask = {0: 0, 1: 2, 2: 3}[w["proofs_verified"]]
    out += [4 * w["domain_log2"] + mask]
    out += [1 if f else 0 for f in w["feature_flags"]]
    out += [1 if w["joint_combiner"] is not None else 0, w["joint_combiner"] or 0]
    assert len(out) == 40
    return [x % Q for x in out]


def statement_public_input(w: dict, step: StepIndex, wrap_index_comms, app_state: int, pp_fp, pp_fq):
    dv = compute_deferred_values(w, step, pp_fp)
    mw = hash_messages_for_next_wrap_proof(w, pp_fq)
    ms = hash_messages_for_next_step_proof(w, wrap_index_comms, app_state, pp_fp)
    return wrap_public_input(w, dv, mw, ms), dv, mw, ms
