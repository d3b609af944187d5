"""Inputs of bench.py's PARTIAL modes (`--mode kimchi`: public inputs given; `--mode prepared`: BatchEvaluationProof rows given) and of
the dev tools that time the host-buffer entry points.  These builders go through the test helpers (oracle objects of the committed
fixtures); bench.py's default mode does not use this module."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import ACC_K, CURVE_VESTA, FIELD_FP, LOG2_DOMAIN, NCOMMS, NPTS, NPUB, PSTATE_SLOTS, SLOT, STATES_PER_PROOF, WRAP_K, le32, make_accumulators  # noqa: E402


def make_chains_serialized(ctx, m, n_chains: int, seed: int):
    """`n_chains` synthetic candidate chains (16 linked states + bridge tip), serialized with the bin_prot writer, flattened by the
    LIBRARY (mina_protocol_state_pack) and hashed by the GPU path itself, state by state, so that each state names its
    predecessor's hash (parity of that path vs the CPU oracle is what tests/ establish).  Returns (records[n,17,2048], nfields[n,17],
    hashes[n,17,32])."""
    from oracle import mina_state_ref as S, state_job_ref as J
    rng = random.Random(seed)
    recs = np.zeros((n_chains, STATES_PER_PROOF, PSTATE_SLOTS * 32), np.uint8)
    nf = np.zeros((n_chains, STATES_PER_PROOF), np.uint32)
    hashes = np.zeros((n_chains, STATES_PER_PROOF, 32), np.uint8)
    prev = [rng.randrange(S.P) for _ in range(n_chains)]
    for s in range(STATES_PER_PROOF):
        for c in range(n_chains):
            st = J.synth_state(rng, prev[c] if s < 16 else rng.randrange(S.P), 1000 + s)
            recs[c, s], nf[c, s], _, _ = m.lib.protocol_state_pack(S.write_protocol_state(st))
        hashes[:, s] = ctx.protocol_state_hash_batch(recs[:, s].copy(), nf[:, s].copy())
        prev = [int.from_bytes(hashes[c, s].tobytes(), "little") for c in range(n_chains)]
    return recs, nf, hashes


def build_batch(ctx, m, B: int, seed: int):
    """host-side `mina_state_jobs` of B jobs from 32 distinct chains, the committed full-size wrap openings
    (tests/golden/state_job_k15.json) and 32 distinct accumulators"""
    from state_job_helpers import entry_arrays, load_k15_openings
    fx, ops = load_k15_openings()
    assert (fx["k"], fx["log2_domain"], fx["npub"], fx["n_comms"], fx["n_points"], fx["slot"]) == (WRAP_K, LOG2_DOMAIN, NPUB, NCOMMS, NPTS, SLOT)
    nd = min(B, 32)
    recs, nf, hashes = make_chains_serialized(ctx, m, nd, seed)
    pre, sgs = make_accumulators(ctx, nd, seed + 1)
    abi = [entry_arrays(e, s) for (_, e, s) in ops]
    pubs = [np.concatenate([le32(x) for x in p]) for (p, _, _) in ops]
    idx = np.arange(B) % nd
    oi = np.arange(B) % len(ops)
    cat = lambda key: np.concatenate([np.asarray(abi[i][key], np.uint8).reshape(-1) for i in oi])
    rho = np.random.Generator(np.random.PCG64(seed + 2)).integers(0, 256, (B, 32), dtype=np.uint8); rho[:, 31] &= 0x3F
    arrays = dict(
        state_records=recs[idx].reshape(-1), state_nfields=nf[idx].reshape(-1), expected_hashes=hashes[idx].reshape(-1),
        public_inputs=np.concatenate([pubs[i] for i in oi]),
        sponge_state=cat("sponge_state"), cip=cat("combined_inner_product"), lr=cat("lr"), delta=cat("delta"), sg=cat("sg"), z1=cat("z1"), z2=cat("z2"),
        evalpoints=cat("evalpoints"), evalscale=cat("evalscale"), polyscale=cat("polyscale"), comms=cat("comms"),
        sponge_pos=np.array([[abi[i]["sponge_mode"], abi[i]["sponge_count"]] for i in oi], np.uint32),
        rand_base=le32(7), sg_rand_base=le32(9), acc_prechallenges=pre[idx].reshape(-1), acc_sg=sgs[idx].reshape(-1), acc_rho=rho.reshape(-1))
    scal = dict(with_states=1, with_ipa=1, with_accumulator=1, log2_domain=LOG2_DOMAIN, npub=NPUB, pub_comm_slot=SLOT, k=WRAP_K, n_evalpoints=NPTS,
                n_comms=NCOMMS, acc_k=ACC_K)
    return m.MinaContext.make_state_jobs(B, arrays, **scal), (recs[0], nf[0], hashes[0], ops[0], pre[0], sgs[0])


def build_kimchi_section(ctx, m, B: int):
    """the raw wrap proofs of the committed wrap-size fixture, tiled to B: (KimchiProofs + keep, opening arrays, public inputs)"""
    from kimchi_helpers import install_index, kimchi_arrays, load_k15_fixture
    ix, proofs, fx = load_k15_fixture()
    install_index(ctx, ix)
    idx = np.arange(B) % len(proofs)
    arrays, op = kimchi_arrays([p for _, p in proofs], [pi for pi, _ in proofs])
    per = {"prev_chals": 2 * 15 * 32, "prev_comms": 2 * 64, "w_comm": 15 * 64, "z_comm": 64, "t_comm": 7 * 64, "evals": 43 * 64, "ft_eval1": 32, "public_inputs": 40 * 32,
           "lr": 30 * 64, "delta": 64, "sg": 64, "z1": 32, "z2": 32}
    tile = lambda a, n: np.ascontiguousarray(a.reshape(len(proofs), n)[idx].reshape(-1))
    arrays = {k: tile(v, per[k]) for k, v in arrays.items() if v is not None}
    op = {k: tile(v, per[k]) for k, v in op.items()}
    return m.MinaContext.make_kimchi_proofs(B, 2, 40, arrays), op, arrays["public_inputs"]


def build_full_section(ctx, m, B: int):
    """the complete wrap proofs of tests/golden/statement_k15.json (Pickles statement + wrap proof whose public input is its packing + the
    step accumulator the statement carries), tiled to B: (KimchiProofs with statements + keep, opening arrays, accumulator arrays, sample)"""
    from kimchi_helpers import install_index, install_step_index, kimchi_arrays, load_k15_fixture, load_statement_fixture, make_step_index, statements_soa
    ix, _, _ = load_k15_fixture()
    install_index(ctx, ix)
    step = make_step_index(99)
    install_step_index(ctx, step)
    items, fx = load_statement_fixture()
    n = len(items)
    idx = np.arange(B) % n
    arrays, op = kimchi_arrays([it["proof"] for it in items], [])
    per = {"prev_chals": 2 * 15 * 32, "prev_comms": 2 * 64, "w_comm": 15 * 64, "z_comm": 64, "t_comm": 7 * 64, "evals": 43 * 64, "ft_eval1": 32,
           "lr": 30 * 64, "delta": 64, "sg": 64, "z1": 32, "z2": 32}
    tile = lambda a, w: np.ascontiguousarray(np.asarray(a, np.uint8).reshape(n, w)[idx].reshape(-1))
    arrays = {k: tile(v, per[k]) for k, v in arrays.items() if v is not None}
    op = {k: tile(v, per[k]) for k, v in op.items()}
    n_old, n_evals, sec = statements_soa([it["wrap"] for it in items], [it["app"] for it in items])
    sec = {k: tile(v, v.size // n) for k, v in sec.items()}
    st = m.MinaContext.make_pickles_statements(n_old, n_evals, sec)
    acc = {"acc_prechallenges": tile(np.stack([it["acc_pre"] for it in items]), ACC_K * 16), "acc_sg": tile(np.stack([it["acc_sg"] for it in items]), 64)}
    return m.MinaContext.make_kimchi_proofs(B, 2, 40, arrays, statements=st), op, acc, (ix, step, items[0])
