"""CPU composite of one Proof-of-State job (TEST INFRASTRUCTURE ONLY) -- the checker for `mina_state_job_batch`
(BASELINE config C3 / C1; README.md:281-310).

Per proof, in the order the reference's verifier works (README.md:283-310):
  1. hash the 16 candidate-chain states and the bridge tip state, compare with the public inputs, check the chain linkage
     (oracle/mina_state_ref.py)
  2. wrap-proof public-input commitment  h - sum_i pub_i L_i  (kimchi `verifier::to_batch`: `public_comm`), which is one of the
     commitments the wrap proof's opening covers
  3. the combined IPA opening check of the wrap proof (oracle/ipa_ref.py `ipa_verify_batch`)
  4. the step accumulator check  MSM(vesta.g, b_poly_coefficients(chals)) == sg  (openmina `accumulator_check`)
The instance is synthetic but shape-faithful: real Mina proofs, the blockchain-snark verifier index and the
kimchi oracles that derive (xi, r, evaluation points) from the proof are not available offline.  PARITY: unpinned (every
piece restates un-vendored code); the protocol-state READER is pinned by core/src/utils/constants.rs:22.
"""
from __future__ import annotations

import random

import numpy as np

from . import ipa_ref as I
from . import mina_state_ref as S
from . import oracle as O
from . import pasta_ref as R


# ------------------------------------------------------------------------------------------------ synthetic chain
def _rf(rng):
    return rng.randrange(R.P)


def synth_state(rng: random.Random, previous_state_hash: int, height: int, n_sub_windows: int = 11) -> dict:
    def signed():
        return {"magnitude": rng.randrange(1 << 64), "sgn": rng.randrange(2)}

    def local():
        return {"stack_frame": _rf(rng), "call_stack": _rf(rng), "transaction_commitment": _rf(rng), "full_transaction_commitment": _rf(rng),
                "excess": signed(), "supply_increase": signed(), "ledger": _rf(rng), "success": bool(rng.randrange(2)),
                "account_update_index": rng.randrange(1 << 32), "failure_status_tbl": [], "will_succeed": bool(rng.randrange(2))}

    def registers():
        return {"first_pass_ledger": _rf(rng), "second_pass_ledger": _rf(rng),
                "pending_coinbase_stack": {"data": _rf(rng), "state": {"init": _rf(rng), "curr": _rf(rng)}}, "local_state": local()}

    def epoch():
        return {"ledger": {"hash": _rf(rng), "total_currency": rng.randrange(1 << 64)}, "seed": _rf(rng), "start_checkpoint": _rf(rng),
                "lock_checkpoint": _rf(rng), "epoch_length": rng.randrange(1 << 16)}

    def pk():
        return {"x": _rf(rng), "is_odd": bool(rng.randrange(2))}

    return {
        "previous_state_hash": previous_state_hash,
        "body": {
            "genesis_state_hash": _rf(rng),
            "blockchain_state": {
                "staged_ledger_hash": {"non_snark": {"ledger_hash": _rf(rng), "aux_hash": rng.randbytes(32), "pending_coinbase_aux": rng.randbytes(32)},
                                       "pending_coinbase_hash": _rf(rng)},
                "genesis_ledger_hash": _rf(rng),
                "ledger_proof_statement": {"source": registers(), "target": registers(), "connecting_ledger_left": _rf(rng),
                                           "connecting_ledger_right": _rf(rng), "supply_increase": signed(),
                                           "fee_excess": {"fee_token_l": _rf(rng), "fee_excess_l": signed(), "fee_token_r": _rf(rng), "fee_excess_r": signed()}},
                "timestamp": rng.randrange(1 << 48), "body_reference": rng.randbytes(32)},
            "consensus_state": {
                "blockchain_length": height, "epoch_count": rng.randrange(64), "min_window_density": rng.randrange(78),
                "sub_window_densities": [rng.randrange(8) for _ in range(n_sub_windows)], "last_vrf_output": rng.randbytes(32),
                "total_currency": rng.randrange(1 << 62),
                "curr_global_slot_since_hard_fork": {"slot_number": rng.randrange(1 << 24), "slots_per_epoch": 7140},
                "global_slot_since_genesis": rng.randrange(1 << 24), "staking_epoch_data": epoch(), "next_epoch_data": epoch(),
                "has_ancestor_in_same_checkpoint_window": bool(rng.randrange(2)), "block_stake_winner": pk(), "block_creator": pk(),
                "coinbase_receiver": pk(), "supercharge_coinbase": bool(rng.randrange(2))},
            "constants": {"k": 290, "slots_per_epoch": 7140, "slots_per_sub_window": 7, "grace_period_slots": 2160, "delta": 0,
                          "genesis_state_timestamp": 1717545600000}}}


def synth_chain(rng: random.Random, pp: R.PoseidonParams, n: int = 16):
    """n linked candidate states (oldest .. tip) + an unrelated bridge tip state.  Returns (states[n+1], hashes[n+1])."""
    states, hashes = [], []
    prev = _rf(rng)
    for i in range(n):
        st = synth_state(rng, prev, 1000 + i)
        prev = S.protocol_state_hash(st, pp)
        states.append(st); hashes.append(prev)
    tip = synth_state(rng, _rf(rng), 990)
    states.append(tip); hashes.append(S.protocol_state_hash(tip, pp))
    return states, hashes


# ------------------------------------------------------------------------------------------------ public-input polynomial
def domain_generator(r: int, log2: int) -> int:
    return pow(R.two_adic_root_of_unity(r), 1 << (32 - log2), r)


def public_poly_coeffs(curve: int, log2_domain: int, pubs) -> list[int]:
    """coefficients of  -sum_i pub_i l_i(X)  over the radix-2 domain (kimchi commits the NEGATED public polynomial)"""
    r = R.scalar_modulus(curve)
    n = 1 << log2_domain
    ev = [(-p) % r for p in pubs] + [0] * (n - len(pubs))
    # inverse FFT (iterative, bit-reversal, w^-1)
    w_inv = pow(domain_generator(r, log2_domain), r - 2, r)
    a = list(ev)
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit; bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = pow(w_inv, n // length, r)
        for start in range(0, n, length):
            w = 1
            for t in range(length // 2):
                u, v = a[start + t], a[start + t + length // 2] * w % r
                a[start + t], a[start + t + length // 2] = (u + v) % r, (u - v) % r
                w = w * wl % r
        length <<= 1
    n_inv = pow(n, r - 2, r)
    return [x * n_inv % r for x in a]


def public_input_commitment(curve: int, g_bytes, h, log2_domain: int, pubs):
    """h - sum_i pub_i L_i, through the monomial basis: commit(-p(X)) + 1 * h"""
    return I.commit(curve, g_bytes[: 1 << log2_domain], h, public_poly_coeffs(curve, log2_domain, pubs), 1)


# ------------------------------------------------------------------------------------------------ the wrap opening
def make_wrap_opening(curve, g_bytes, h, pp: R.PoseidonParams, k: int, log2_domain: int, pubs, n_comms: int, slot: int, n_points: int, seed: int,
                      sparse: int | None = None):
    """A valid opening over g[0..2^k) whose commitment `slot` is the public-input commitment of `pubs` (needs log2_domain <= k).
    Other polynomials are random (`sparse`: only that many non-zero coefficients each, to keep k = 15 generation quick --
    the verifier's work does not depend on it).  Returns (entry, sponge_before)."""
    rng = random.Random(seed)
    r = R.scalar_modulus(curve)
    n = 1 << k
    polys, blinders = [], []
    for i in range(n_comms):
        if i == slot:
            c = public_poly_coeffs(curve, log2_domain, pubs)
            polys.append(c + [0] * (n - len(c))); blinders.append(1)
        else:
            if sparse is None:
                polys.append([rng.randrange(r) for _ in range(n)])
            else:
                f = [0] * n
                for _ in range(sparse):
                    f[rng.randrange(n)] = rng.randrange(r)
                polys.append(f)
            blinders.append(rng.randrange(r))
    comms = [I.commit(curve, g_bytes[:n], h, f, bl) for f, bl in zip(polys, blinders)]
    evalpoints = [rng.randrange(r) for _ in range(n_points)]
    xi, rscale = rng.randrange(r), rng.randrange(r)
    sponge = I.FqSponge(curve, pp)
    sponge.absorb_g(comms)
    sponge.challenge()
    before = sponge.clone()
    op = I.ipa_open_fast(curve, g_bytes[:n], h, polys, blinders, evalpoints, xi, rscale, sponge, rng)
    entry = {"evalpoints": evalpoints, "polyscale": xi, "evalscale": rscale, "comms": comms, "opening": op,
             "combined_inner_product": op["combined_inner_product"], "k": k}
    return entry, before


# ------------------------------------------------------------------------------------------------ the accumulator
def make_accumulator(curve, g_bytes, k: int, seed: int):
    """k 128-bit prechallenges and the matching sg = <b_poly_coefficients(to_field(prechallenges)), g>"""
    rng = random.Random(seed)
    fs = O.scalar_field_of(curve)
    pre = np.frombuffer(rng.randbytes(16 * k), dtype=np.uint8).reshape(k, 16).copy()
    _, endo_r = O.endo(curve)
    chals = np.stack([O.challenge_to_field(fs, pre[i].copy(), endo_r) for i in range(k)])
    s = O.b_poly_coefficients(fs, chals)
    sg = O.msm_pippenger(curve, g_bytes[: 1 << k], s, threads=8)
    return pre, sg


def accumulator_ok(curve, g_bytes, k: int, pre, sg, threads: int = 8) -> bool:
    fs = O.scalar_field_of(curve)
    _, endo_r = O.endo(curve)
    chals = np.stack([O.challenge_to_field(fs, np.ascontiguousarray(pre[i]), endo_r) for i in range(k)])
    s = O.b_poly_coefficients(fs, chals)
    return bool((O.msm_pippenger(curve, g_bytes[: 1 << k], s, threads=threads) == np.asarray(sg, dtype=np.uint8).reshape(64)).all())


# ------------------------------------------------------------------------------------------------ the composite verdict
def verify_state_job(pp_fp: R.PoseidonParams, srs_pallas, srs_vesta, job: dict) -> dict:
    """job: {states[17] dicts, expected_hashes[17] ints, pubs, log2_domain, slot, entry (comms[slot] is IGNORED and recomputed),
    sponge_before, acc_k, acc_pre, acc_sg}.  Returns every intermediate and the verdict."""
    out = {}
    hashes = [S.protocol_state_hash(st, pp_fp) for st in job["states"]]
    out["hashes"] = hashes
    ok = all(h == e for h, e in zip(hashes, job["expected_hashes"]))
    for i in range(1, 16):
        ok = ok and job["states"][i]["previous_state_hash"] == hashes[i - 1]
    out["chain_ok"] = ok
    g, h = srs_pallas
    hp = O.bytes_to_point(h)
    pc = public_input_commitment(0, g, hp, job["log2_domain"], job["pubs"])
    out["public_comm"] = pc
    entry = dict(job["entry"])
    entry["comms"] = list(entry["comms"])
    entry["comms"][job["slot"]] = pc
    entry["sponge"] = job["sponge_before"].clone()
    out["ipa_ok"] = I.ipa_verify_batch(0, g[: 1 << entry["k"]], hp, [entry], 7, 9)
    gv, _ = srs_vesta
    out["acc_ok"] = accumulator_ok(1, gv, job["acc_k"], job["acc_pre"], job["acc_sg"])
    out["verdict"] = bool(out["chain_ok"] and out["ipa_ok"] and out["acc_ok"])
    return out
