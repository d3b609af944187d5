"""shared by the kimchi tests: oracle index / proof -> the C-ABI's `mina_verifier_index` / `mina_kimchi_proofs` layouts"""
import struct

import numpy as np


def encode_tokens(tokens) -> bytes:
    from oracle import kimchi_ref as K
    out = bytearray()
    for t in tokens:
        op = t[0]
        out.append(op)
        if op == K.T_MDS:
            out += bytes([t[1], t[2]])
        elif op == K.T_LITERAL:
            out += int(t[1]).to_bytes(32, "little")
        elif op == K.T_CELL:
            out += bytes([t[1], t[2]])
        elif op == K.T_POW:
            out += struct.pack("<Q", t[1])
        elif op == K.T_LAGRANGE:
            out += struct.pack("<i", t[1])
        elif op == K.T_LOAD:
            out += struct.pack("<H", t[1])
        elif op in (K.T_SKIP_IF, K.T_SKIP_IF_NOT):
            out += bytes([t[1]]) + struct.pack("<H", t[2])
    return bytes(out)


def pts(points):
    from oracle import oracle as O
    return np.concatenate([O.point_to_bytes(p) for p in points])


def install_index(ctx, index):
    from oracle import oracle as O
    ctx.verifier_index_install(index.log2_domain, index.zk_rows, index.perm_alpha_offset, O.ints_to_le(index.shifts).reshape(-1), pts(index.sigma_comm),
                               pts(index.coefficients_comm), pts(index.selector_comm), encode_tokens(index.constant_term))


def kimchi_arrays(proofs, publics):
    """list of oracle proofs (same shape) -> dict of arrays for MinaContext.make_kimchi_proofs + the opening arrays (lr, delta, sg, z1, z2)"""
    from oracle import oracle as O
    cat = lambda xs: np.concatenate(xs)
    a = {
        "prev_chals": cat([O.ints_to_le(ch).reshape(-1) for p in proofs for ch, _ in p["prev"]]) if proofs[0]["prev"] else None,
        "prev_comms": cat([O.point_to_bytes(cm) for p in proofs for _, cm in p["prev"]]) if proofs[0]["prev"] else None,
        "w_comm": cat([pts(p["w_comm"]) for p in proofs]), "z_comm": cat([O.point_to_bytes(p["z_comm"]) for p in proofs]),
        "t_comm": cat([pts(p["t_comm"]) for p in proofs]),
        "evals": cat([O.ints_to_le([e for pair in p["evals"] for e in pair]).reshape(-1) for p in proofs]),
        "ft_eval1": cat([O.int_to_le(p["ft_eval1"]) for p in proofs]),
    }
    if publics and len(publics[0]):
        a["public_inputs"] = cat([O.ints_to_le(pi).reshape(-1) for pi in publics])
    op = {
        "lr": cat([cat([cat([O.point_to_bytes(L), O.point_to_bytes(R)]) for L, R in p["opening"]["lr"]]) for p in proofs]),
        "delta": cat([O.point_to_bytes(p["opening"]["delta"]) for p in proofs]), "sg": cat([O.point_to_bytes(p["opening"]["sg"]) for p in proofs]),
        "z1": cat([O.int_to_le(p["opening"]["z1"]) for p in proofs]), "z2": cat([O.int_to_le(p["opening"]["z2"]) for p in proofs]),
    }
    return a, op


def load_k15_fixture():
    """tests/golden/kimchi_k15.json -> (oracle VerifierIndex, [(pubs, proof dict)], raw json)"""
    import json, os
    from oracle import kimchi_ref as K, oracle as O
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kimchi_k15.json")))
    pt = lambda hx: O.bytes_to_point(np.frombuffer(bytes.fromhex(hx), np.uint8))
    ix = K.VerifierIndex(curve=0, log2_domain=fx["log2_domain"], zk_rows=fx["zk_rows"], shifts=[int(x) for x in fx["shifts"]],
                         sigma_comm=[pt(x) for x in fx["sigma_comm"]], coefficients_comm=[pt(x) for x in fx["coefficients_comm"]],
                         selector_comm=[pt(x) for x in fx["selector_comm"]], constant_term=[tuple(t) for t in fx["constant_term"]],
                         perm_alpha_offset=fx["perm_alpha_offset"], digest=int(fx["digest"]))
    from ipa_helpers import poseidon_pp
    ix.mds = [list(row) for row in poseidon_pp(1).mds]       # Constants.mds of the scalar field, as synthetic_circuit sets it
    proofs = []
    for p in fx["proofs"]:
        proof = {"prev": [([int(c) for c in ch], pt(cm)) for ch, cm in p["prev"]], "w_comm": [pt(x) for x in p["w_comm"]], "z_comm": pt(p["z_comm"]),
                 "t_comm": [pt(x) for x in p["t_comm"]], "evals": [(int(a), int(b)) for a, b in p["evals"]], "ft_eval1": int(p["ft_eval1"]),
                 "opening": {"lr": [(pt(l), pt(r)) for l, r in p["lr"]], "delta": pt(p["delta"]), "sg": pt(p["sg"]), "z1": int(p["z1"]), "z2": int(p["z2"])},
                 "expect": p["expect"]}
        proofs.append(([int(x) for x in p["pubs"]], proof))
    return ix, proofs, fx


def statements_soa(wraps, apps):
    """wrap-proof dicts (tests/wire_writers.py layout) + application states -> the sections of `mina_pickles_statements` (numpy uint8).
    Chunked evaluations are combined with zeta^(2^16) here, as the struct asks (the product's own container path does the same on the host)."""
    import numpy as np
    from oracle import kimchi_ref as K, pasta_ref as R, pickles_ref as PK
    le = lambda x, n: int(x).to_bytes(n, "little")
    sec = {n: bytearray() for n in ("plonk", "bulletproof_challenges", "step_old_challenges", "step_comms", "wrap_old_challenges", "wrap_sg", "sponge_digest", "prev_evals",
                                    "prev_public_input", "prev_ft_eval1", "app_state", "misc")}
    n_old = len(wraps[0]["step_old_chals"]); n_evals = None
    for w, app in zip(wraps, apps):
        assert len(w["step_old_chals"]) == n_old and len(w["step_comms"]) == n_old
        for name in ("alpha", "beta", "gamma", "zeta"):
            sec["plonk"] += le(w[name], 16)
        for c in w["bulletproof_challenges"]:
            sec["bulletproof_challenges"] += le(c, 16)
        for row in w["step_old_chals"]:
            for c in row:
                sec["step_old_challenges"] += le(c, 16)
        for x, y in w["step_comms"]:
            sec["step_comms"] += le(x, 32) + le(y, 32)
        for row in w["old_bulletproof_challenges"]:
            for c in row:
                sec["wrap_old_challenges"] += le(c, 16)
        sg = w["challenge_polynomial_commitment"]
        sec["wrap_sg"] += le(sg[0], 32) + le(sg[1], 32)
        for l in w["sponge_digest"]:
            sec["sponge_digest"] += le(l, 8)
        zeta = R.challenge_to_field(w["zeta"], PK.endo_fp(), PK.P)
        zetaw = zeta * K.O_domain_generator(PK.P, w["domain_log2"]) % PK.P
        zn, zwn = pow(zeta, 1 << 16, PK.P), pow(zetaw, 1 << 16, PK.P)
        seq = [PK.combine_chunks(pr, zn, zwn, PK.P) for pr in PK.prev_evals_sequence(w)]
        n_evals = n_evals or len(seq); assert len(seq) == n_evals
        for a, b in seq:
            sec["prev_evals"] += le(a, 32) + le(b, 32)
        sec["prev_public_input"] += le(w["prev_public_input"][0], 32) + le(w["prev_public_input"][1], 32)
        sec["prev_ft_eval1"] += le(w["prev_ft_eval1"], 32)
        sec["app_state"] += le(app, 32)
        misc = bytearray(32)
        misc[0] = w["domain_log2"]; misc[1] = w["proofs_verified"]
        for j, f in enumerate(w["feature_flags"]):
            misc[2 + j] = 1 if f else 0
        if w["joint_combiner"] is not None:
            misc[10] = 1; misc[16:32] = le(w["joint_combiner"], 16)
        present = sum(1 << j for j, e in enumerate(w["prev_optional"]) if e is not None)      # which of the 19 optional evaluations the proof carries (wire order)
        misc[11:14] = present.to_bytes(3, "little")
        sec["misc"] += misc
    return n_old, n_evals, {k: np.frombuffer(bytes(v), np.uint8).copy() if len(v) else np.zeros(1, np.uint8) for k, v in sec.items()}


STEP_DOMAINS = list(range(10, 17))


def make_step_index(seed):
    """a synthetic STEP index: random shifts for every step domain and a constant-term program over the step evaluations"""
    from ipa_helpers import poseidon_pp
    from oracle import kimchi_ref as K, pickles_ref as PK, pasta_ref as R
    import random
    rng = random.Random(seed)
    toks = [(K.T_CELL, K.COL_W0 + 2, 0), (K.T_CELL, K.COL_COEFF0 + 1, 1), (K.T_MUL,), (K.T_CELL, K.COL_GENERIC, 0), (K.T_ADD,), (K.T_ALPHA,), (K.T_MUL,),
            (K.T_ENDO,), (K.T_MDS, 2, 0), (K.T_MUL,), (K.T_ADD,), (K.T_LITERAL, 987654321), (K.T_SUB,), (K.T_VANISH_ZK,), (K.T_LAGRANGE, -2), (K.T_MUL,), (K.T_ADD,),
            (K.T_BETA,), (K.T_GAMMA,), (K.T_MUL,), (K.T_POW, 5), (K.T_STORE,), (K.T_ADD,), (K.T_LOAD, 0), (K.T_SUB,)]
    return PK.StepIndex(zk_rows=3, shifts={k: [1] + [rng.randrange(2, R.P) for _ in range(6)] for k in STEP_DOMAINS}, constant_term=toks,
                        mds=[list(r) for r in poseidon_pp(0).mds])


def make_feature_step_index(seed):
    """a FEATURE-AWARE synthetic step index: on top of make_step_index's program, regions guarded by kimchi's SkipIf / SkipIfNot that read
    optional evaluations (lookup aggregation / table, the range_check0 selector, a lookup-pattern selector), the joint combiner, a value STOREd
    and LOADed inside a region, and -- behind the regions -- an outer STORE whose cache slot depends on whether that region ran (kimchi's
    `evaluate` pushes a slot per EXECUTED Store: skipped tokens have no effect), followed by a LOAD of slot 1: the region's value when
    LookupTables is on, the outer one when it is off.  The shape kimchi's feature-flagged linearization compiles to:
        if_feature(f, e1, e2)  ->  SkipIfNot(f, |e1|) e1 SkipIf(f, |e2|) e2 Add"""
    from oracle import kimchi_ref as K
    base = make_step_index(seed)
    assert sum(t[0] == K.T_STORE for t in base.constant_term) == 1      # slot 0 is the base program's
    OPT = K.N_EVAL_COLS                                                   # 43 + slot (wire order of the optional evaluations)
    e1 = [(K.T_CELL, OPT + 6, 1), (K.T_JOINT,), (K.T_MUL,), (K.T_CELL, OPT + 7, 0), (K.T_ADD,), (K.T_STORE,), (K.T_LOAD, 1), (K.T_MUL,)]   # (aggregation(zeta w) * joint + table(zeta))^2 through the cache
    e2 = [(K.T_LITERAL, 31337), (K.T_ALPHA,), (K.T_MUL,)]
    r1 = [(K.T_SKIP_IF_NOT, 6, len(e1))] + e1 + [(K.T_SKIP_IF, 6, len(e2))] + e2 + [(K.T_ADD,)]                          # if_feature(LookupTables, e1, e2)
    e3 = [(K.T_CELL, OPT + 0, 0), (K.T_BETA,), (K.T_MUL,)]
    r2 = [(K.T_SKIP_IF_NOT, 0, len(e3))] + e3                                                                           # range_check0 selector, else zero
    inner = [(K.T_CELL, OPT + 17, 1), (K.T_GAMMA,), (K.T_ADD,)]
    e4 = [(K.T_ENDO,), (K.T_SKIP_IF_NOT, 10, len(inner))] + inner + [(K.T_MUL,)]                                        # nested: LookupPattern RangeCheck inside TableWidth(1)
    r3 = [(K.T_SKIP_IF_NOT, 13, len(e4))] + e4
    tail = [(K.T_ALPHA,), (K.T_STORE,), (K.T_LOAD, 1), (K.T_MUL,), (K.T_ADD,)]                                          # slot 1 (region skipped) or 2 (region ran)
    toks = list(base.constant_term) + r1 + [(K.T_ADD,)] + r2 + [(K.T_ADD,)] + r3 + [(K.T_SUB,)] + tail
    return type(base)(zk_rows=base.zk_rows, shifts=base.shifts, constant_term=toks, mds=base.mds)


def install_step_index(ctx, step):
    from kimchi_helpers import encode_tokens
    from oracle import oracle as O
    sh = np.concatenate([O.ints_to_le(step.shifts[k]).reshape(-1) for k in STEP_DOMAINS])
    ctx.step_index_install(step.zk_rows, STEP_DOMAINS, sh, encode_tokens(step.constant_term))


def load_statement_fixture(path=None):
    """tests/golden/statement_k15.json (or another file of tests/golden/gen_statement_fixture.py) -> [dict(wrap=statement fields as in tests/wire_writers.py, app, pubs,
    acc_pre [16,16] u8, acc_sg [64] u8, proof)] for the wrap index of kimchi_k15.json and the step index make_step_index(99)"""
    import json, os
    from oracle import oracle as O
    fx = json.load(open(path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "statement_k15.json")))
    pt = lambda hx: O.bytes_to_point(np.frombuffer(bytes.fromhex(hx), np.uint8))
    dec = lambda v: None if v is None else v if isinstance(v, bool) else int(v) if isinstance(v, str) else [dec(x) for x in v]
    out = []
    for p in fx["proofs"]:
        wrap = {k: dec(v) for k, v in p["statement"].items()}
        for key in ("step_comms",):
            wrap[key] = [tuple(x) for x in wrap[key]]
        wrap["challenge_polynomial_commitment"] = tuple(wrap["challenge_polynomial_commitment"])
        wrap["prev_public_input"] = tuple(wrap["prev_public_input"])
        wrap["prev_evals"] = [tuple(x) for x in wrap["prev_evals"]]
        proof = {"w_comm": [pt(x) for x in p["w_comm"]], "z_comm": pt(p["z_comm"]), "t_comm": [pt(x) for x in p["t_comm"]],
                 "evals": [(int(a), int(b)) for a, b in p["evals"]], "ft_eval1": int(p["ft_eval1"]),
                 "opening": {"lr": [(pt(l), pt(r)) for l, r in p["lr"]], "delta": pt(p["delta"]), "sg": pt(p["sg"]), "z1": int(p["z1"]), "z2": int(p["z2"])}}
        from oracle import pasta_ref as R
        proof["prev"] = [([R.challenge_to_field(c, R.endo_r(0), R.Q) for c in row], cm) for row, cm in zip(wrap["old_bulletproof_challenges"], wrap["step_comms"])]
        out.append({"wrap": wrap, "app": int(p["app_state"]), "chain_seed": p.get("chain_seed"), "pubs": [int(x) for x in p["pubs"]],
                    "acc_pre": np.frombuffer(bytes.fromhex(p["acc_pre"]), np.uint8).reshape(16, 16).copy(), "acc_sg": np.frombuffer(bytes.fromhex(p["acc_sg"]), np.uint8).copy(),
                    "proof": proof})
    return out, fx


def make_chain(rng, pp):
    """16 linked candidate states + a bridge tip the candidate tip beats by the short-range rule (same epoch, same staking lock
    checkpoint, longer chain)"""
    from oracle import mina_state_ref as S, state_job_ref as J
    states, hashes = [], []
    prev = rng.randrange(S.P)
    lock = rng.randrange(S.P)
    for i in range(17):
        st = J.synth_state(rng, prev if i < 16 else rng.randrange(S.P), 1000 + i if i < 16 else 990)
        if i >= 15:
            st["body"]["consensus_state"]["epoch_count"] = 7
            st["body"]["consensus_state"]["staking_epoch_data"]["lock_checkpoint"] = lock
        h = S.protocol_state_hash(st, pp)
        states.append(st); hashes.append(h); prev = h
    return states, hashes
