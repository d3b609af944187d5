#!/usr/bin/env python3
"""Randomised soak of the reference-shaped boundary on full-size proofs: batches of random size made of the four proofs of
tests/golden/statement_k15.json and tampered copies (wrong public hash -> CHAIN; changed opening scalar -> the folded opening check; changed
step prechallenge -> accumulator + statement; changed statement flag -> kimchi through the public input; an off-curve commitment -> the
malformed-input path of the culprit search; truncated bytes -> FORMAT), through mina_verify_state_batch and, every few rounds, as concurrent
single-proof calls from threads (merged into shared jobs).  Every verdict is compared with what the tampering implies.
usage: soak_boundary.py SECONDS"""
import copy
import json
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.setrecursionlimit(10000)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
from ipa_helpers import poseidon_pp
from kimchi_helpers import install_index, install_step_index, load_k15_fixture, load_statement_fixture, make_chain, make_step_index
from oracle import mina_state_ref as S
from wire_writers import state_proof_bytes, state_pub_bytes

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(time.time()); rng = random.Random(seed0)
ix, _, _ = load_k15_fixture()
items, _ = load_statement_fixture()
m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)
gctx = m.lib.verify_global_ctx()
install_index(gctx, ix); install_step_index(gctx, make_step_index(99))
cases = []
for it in items:
    states, hashes = make_chain(random.Random(it["chain_seed"]), poseidon_pp(0))
    p, ev = it["proof"], it["proof"]["evals"]
    wrap = dict(it["wrap"])
    wrap.update(w_comm=p["w_comm"], z_comm=p["z_comm"], t_comm=p["t_comm"], z_eval=ev[0], selector_eval=ev[1:7], w_eval=ev[7:22], coefficients_eval=ev[22:37],
                s_eval=ev[37:43], ft_eval1=p["ft_eval1"], lr=p["opening"]["lr"], z1=p["opening"]["z1"], z2=p["opening"]["z2"], delta=p["opening"]["delta"], sg=p["opening"]["sg"])
    ledger = [S.snarked_ledger_hash(s) for s in states[:16]]
    cases.append({"wrap": wrap, "states": states, "proof": state_proof_bytes(wrap, states), "pub": state_pub_bytes(True, hashes[16], hashes[:16], ledger)})

def tampered(kind, c):
    w = copy.deepcopy(c["wrap"]); pub = c["pub"]
    if kind == "pub": b = bytearray(pub); b[1 + rng.randrange(17 * 32)] ^= 1 << rng.randrange(8); return c["proof"], bytes(b)
    if kind == "flag2": w["feature_flags"][2] = not w["feature_flags"][2]      # foreign_field_add: the one flag the boundary lets through to the statement
    elif kind == "z1": w["z1"] = (w["z1"] + 1 + rng.randrange(5)) % (1 << 254)
    elif kind == "bp": w["bulletproof_challenges"][rng.randrange(16)] ^= 1 << rng.randrange(128)
    elif kind == "flag": i = rng.randrange(len(w["feature_flags"])); w["feature_flags"][i] = not w["feature_flags"][i]
    elif kind == "offcurve": x, y = w["w_comm"][rng.randrange(15)]; w["w_comm"][0] = (x, (y + 1) % S.P if hasattr(S, "P") else y + 1)
    elif kind == "trunc": return c["proof"][: rng.randrange(len(c["proof"]))], pub
    elif kind == "trunc_states": return c["proof"][: len(c["proof"]) - 1 - rng.randrange(17 * 1400)], pub        # the wrap-proof half parses, the protocol states do not
    elif kind == "trailing": return c["proof"] + bytes([rng.randrange(256)]), pub
    return state_proof_bytes(w, c["states"]), pub

pool = [(c["proof"], c["pub"], True) for c in cases]
kinds = ["pub", "z1", "bp", "flag", "flag2", "offcurve", "trunc", "trunc_states", "trailing"]
Q_MOD = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001


def cancelling_pair():
    """two INVALID proofs whose opening errors cancel in a fold with rho_1 / rho_0 = `guess` (z2 is not bound by the transcript and rides on the
    batch-shared point h): z2_a += guess * t, z2_b -= t.  The round-2 build folded with rand_base = 7 and accepted the pair for guess = 7."""
    i, j = rng.sample(range(4), 2)
    guess = rng.choice([7, 9, 49, 1, 2]) if rng.randrange(3) == 0 else rng.randrange(1, Q_MOD)
    t = rng.randrange(1, Q_MOD)
    a = copy.deepcopy(cases[i]["wrap"]); b = copy.deepcopy(cases[j]["wrap"])
    a["z2"] = (a["z2"] + guess * t) % Q_MOD; b["z2"] = (b["z2"] - t) % Q_MOD
    return (state_proof_bytes(a, cases[i]["states"]), cases[i]["pub"], False), (state_proof_bytes(b, cases[j]["states"]), cases[j]["pub"], False)
bad_pool = {k: [tampered(k, cases[i % 4]) + (False,) for i in range(6)] for k in kinds}
for k in kinds:                                             # every tampering must be rejected on its own, through the lone-call path
    for p, q, _ in bad_pool[k][:2]:
        assert m.lib.verify_state(p, q) is False, ("tampering not rejected", k)
counts = {"batches": 0, "proofs": 0, "bad": 0, "thread_bursts": 0, "cancelling_pairs": 0}
t_end = time.time() + budget; rnd = 0
while time.time() < t_end:
    rnd += 1
    n = rng.choice([1, 2, 3, 5, 8, 13, 24, 33, 40, 70, 130])
    if rnd % 25 == 0: n = rng.choice([2100, 4500, 9000, 17000])      # the streamed form at its real sizes: runs of 1024, hash pieces, several chunks
    nbad = rng.choice([0, 0, 1, 1, 2, 3, n // 4]) if n < 2000 else rng.choice([0, 1, 3, 40])
    batch = [pool[rng.randrange(4)] for _ in range(n)]
    for pos in rng.sample(range(n), min(nbad, n)):
        batch[pos] = rng.choice(bad_pool[rng.choice(kinds)])
    if n >= 2 and rng.randrange(3) == 0:                    # a cancelling pair, adjacent or apart, in either order
        pa, pb = cancelling_pair()
        i, j = rng.sample(range(n), 2)
        batch[i], batch[j] = pa, pb
        counts["cancelling_pairs"] += 1
    want = [int(b[2]) for b in batch]
    # the pipeline's shape, at random: chunks of a few entries, runs of a few entries, the window of chunks and the parsing ahead of it
    tune = {}
    if n >= 2000: pass                                     # library defaults
    elif rng.randrange(2):
        tune.update(chunk=rng.choice([2, 3, 7, 16]), single_max=1, window=rng.choice([1, 2, 4]), ahead=rng.choice([0, 1, 3]), slots=16)
    if n < 2000 and rng.randrange(2):
        tune.update(early_min=1, early_sub=rng.choice([0, 1, 2, 5]), head_min=rng.choice([0, 1000000]))
    m.lib.tune_from_string(",".join(f"{k}={v}" for k, v in tune.items()) or "chunk=8192")
    if rnd % 4 == 0 and n <= 40:
        got = [None] * n
        def worker(i): got[i] = int(m.lib.verify_state(batch[i][0], batch[i][1]))
        th = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
        for t in th: t.start()
        for t in th: t.join()
        counts["thread_bursts"] += 1
    else:
        got = m.lib.verify_state_batch([b[0] for b in batch], [b[1] for b in batch]).tolist()
    assert got == want, ("verdicts", seed0, rnd, n, got, want)
    counts["batches"] += 1; counts["proofs"] += n; counts["bad"] += n - sum(want)
print(json.dumps({"seed": seed0, "seconds": budget, **counts}))
