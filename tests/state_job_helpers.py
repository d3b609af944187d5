"""shared by the Proof-of-State tests / bench / smoke: oracle-side job minting and conversion to the C-ABI `mina_state_jobs` layout"""
import json
import os
import random

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pp_fp():
    from ipa_helpers import poseidon_pp
    return poseidon_pp(0)


def state_records(states):
    """17 state dicts -> (records[17, 64*32] uint8, nfields[17] uint32) through the ORACLE's flattening"""
    from oracle import mina_state_ref as S
    recs = np.zeros((len(states), 64 * 32), np.uint8); nf = np.zeros(len(states), np.uint32)
    for i, st in enumerate(states):
        f = [st["previous_state_hash"]] + S.body_to_input(st["body"]).to_fields()
        b = b"".join(x.to_bytes(32, "little") for x in f)
        recs[i, : len(b)] = np.frombuffer(b, np.uint8)
        nf[i] = len(f) - 1
    return recs, nf


def entry_arrays(entry, sponge):
    from ipa_helpers import to_abi
    return to_abi(entry, sponge)


def build_jobs(m, jobs, k, log2_domain, slot, acc_k, rand_base=7, sg_rand_base=9, rho_seed=11, with_states=True, with_ipa=True, with_acc=True):
    """jobs: list of dicts {records, nfields, expected (17 ints), pubs (ints), abi (entry_arrays), acc_pre, acc_sg} -> (StateJobs, keep)"""
    from oracle import oracle as O
    B = len(jobs)
    abi = [j["abi"] for j in jobs]
    arrays = {}
    scal = dict(with_states=int(with_states), with_ipa=int(with_ipa), with_accumulator=int(with_acc), log2_domain=log2_domain, pub_comm_slot=slot,
                k=k, acc_k=acc_k, npub=0, n_evalpoints=0, n_comms=0)
    if with_states:
        arrays["state_records"] = np.concatenate([j["records"] for j in jobs])
        arrays["state_nfields"] = np.concatenate([j["nfields"] for j in jobs]).astype(np.uint32)
        arrays["expected_hashes"] = np.concatenate([O.ints_to_le(j["expected"]) for j in jobs])
    if with_ipa:
        npub = len(jobs[0]["pubs"])
        scal.update(npub=npub, n_evalpoints=abi[0]["n_evalpoints"], n_comms=abi[0]["n_comms"])
        if npub:
            arrays["public_inputs"] = np.concatenate([O.ints_to_le(j["pubs"]) for j in jobs])
        cat = lambda key: np.concatenate([np.asarray(a[key], dtype=np.uint8).reshape(-1) for a in abi])
        arrays.update(sponge_state=cat("sponge_state"), cip=cat("combined_inner_product"), lr=cat("lr"), delta=cat("delta"), sg=cat("sg"), z1=cat("z1"),
                      z2=cat("z2"), evalpoints=cat("evalpoints"), evalscale=cat("evalscale"), polyscale=cat("polyscale"), comms=cat("comms"))
        arrays["sponge_pos"] = np.array([[a["sponge_mode"], a["sponge_count"]] for a in abi], np.uint32)
        arrays["rand_base"] = O.int_to_le(rand_base); arrays["sg_rand_base"] = O.int_to_le(sg_rand_base)
    if with_acc:
        arrays["acc_prechallenges"] = np.concatenate([np.asarray(j["acc_pre"], np.uint8).reshape(-1) for j in jobs])
        arrays["acc_sg"] = np.concatenate([np.asarray(j["acc_sg"], np.uint8).reshape(-1) for j in jobs])
        rng = np.random.Generator(np.random.PCG64(rho_seed))
        rho = rng.integers(0, 256, (B, 32), dtype=np.uint8); rho[:, 31] &= 0x3F
        arrays["acc_rho"] = rho
    return m.MinaContext.make_state_jobs(B, arrays, **scal)


def mint_job(srs_pallas, srs_vesta, seed, k, log2_domain, npub, n_comms, slot, n_points, acc_k, sparse=None, opening=None):
    """one oracle-minted, valid Proof-of-State job (dict for build_jobs + the oracle-side pieces for verify_state_job)"""
    from oracle import oracle as O, pasta_ref as R, state_job_ref as J
    rng = random.Random(seed)
    pp = pp_fp()
    states, hashes = J.synth_chain(rng, pp)
    recs, nf = state_records(states)
    g, h = srs_pallas
    if opening is None:
        pubs = [rng.randrange(R.Q) for _ in range(npub)]
        entry, sponge = J.make_wrap_opening(0, g, O.bytes_to_point(h), pp, k, log2_domain, pubs, n_comms, slot, n_points, seed + 1, sparse=sparse)
    else:
        pubs, entry, sponge = opening
    gv, _ = srs_vesta
    pre, sg = J.make_accumulator(1, gv, acc_k, seed + 2)
    return {"states": states, "expected": list(hashes), "records": recs, "nfields": nf, "pubs": pubs, "entry": entry, "sponge": sponge,
            "abi": entry_arrays(entry, sponge), "acc_pre": pre, "acc_sg": sg, "log2_domain": log2_domain, "slot": slot, "acc_k": acc_k}


def oracle_job(job):
    """the dict `state_job_ref.verify_state_job` takes, from the (possibly tampered) numpy side of a minted job"""
    from oracle import oracle as O
    return {"states": job["states"], "expected_hashes": job["expected"], "pubs": job["pubs"], "log2_domain": job["log2_domain"], "slot": job["slot"],
            "entry": job["entry"], "sponge_before": job["sponge"], "acc_k": job["acc_k"], "acc_pre": job["acc_pre"], "acc_sg": job["acc_sg"]}


# ---- committed full-size wrap openings (k = 15, 45 commitments, 2 points, 40 public inputs): tests/golden/state_job_k15.json
def load_k15_openings():
    from oracle import ipa_ref as I, oracle as O
    fx = json.load(open(os.path.join(GOLDEN, "state_job_k15.json")))
    out = []
    pt = lambda hx: O.bytes_to_point(np.frombuffer(bytes.fromhex(hx), np.uint8))
    for e in fx["openings"]:
        entry = {"evalpoints": [int(x) for x in e["evalpoints"]], "polyscale": int(e["polyscale"]), "evalscale": int(e["evalscale"]),
                 "comms": [pt(c) for c in e["comms"]], "combined_inner_product": int(e["cip"]), "k": fx["k"],
                 "opening": {"lr": [(pt(l), pt(r)) for l, r in e["lr"]], "delta": pt(e["delta"]), "sg": pt(e["sg"]), "z1": int(e["z1"]), "z2": int(e["z2"]),
                             "combined_inner_product": int(e["cip"])}}
        sponge = I.FqSponge(0, pp_fp(), [int(x) for x in e["sponge_state"]], "squeezed" if e["sponge_mode"] else "absorbed", e["sponge_count"])
        out.append(([int(x) for x in e["pubs"]], entry, sponge))
    return fx, out
