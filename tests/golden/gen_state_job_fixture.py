"""Generates tests/golden/state_job_k15.json: full-size wrap-proof openings for the Proof-of-State job (BASELINE config C3) --
Pallas, k = 15, 45 commitments (slot 0 = the public-input commitment of 40 public inputs over the 2^15 domain), 2 evaluation
points -- minted by the repo's OWN CPU oracle (oracle/state_job_ref.py, ipa_ref.ipa_open_fast) under the Poseidon constant set
named in the file.  There is no reference implementation to import; these are inputs + expected verdict, not
reference outputs.  Run:  python tests/golden/gen_state_job_fixture.py [count]"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O, pasta_ref as R, state_job_ref as J
from ipa_helpers import poseidon_pp
import mina_bridge_amd.poseidon_params as PP

K, LOG2, NPUB, NCOMMS, SLOT, NPTS = 15, 15, 40, 45, 0, 2
count = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g, h = O.srs_create(0, 1 << K, threads=os.cpu_count() or 4)
hp = O.bytes_to_point(h)
pp = poseidon_pp(0)
hx = lambda p: O.point_to_bytes(p).tobytes().hex()
ops = []
for i in range(count):
    rng = random.Random(0xC3 + i)
    pubs = [rng.randrange(R.Q) for _ in range(NPUB)]
    entry, sponge = J.make_wrap_opening(0, g, hp, pp, K, LOG2, pubs, NCOMMS, SLOT, NPTS, 1000 + i, sparse=64)
    from oracle import ipa_ref as I
    chk = dict(entry); chk["sponge"] = sponge.clone()
    assert I.ipa_verify_batch(0, g, hp, [chk], 7, 9), "minted opening does not verify"
    st, mode, cnt = sponge.raw()
    op = entry["opening"]
    ops.append({"pubs": [str(x) for x in pubs], "evalpoints": [str(x) for x in entry["evalpoints"]], "polyscale": str(entry["polyscale"]),
                "evalscale": str(entry["evalscale"]), "comms": [hx(c) for c in entry["comms"]], "cip": str(entry["combined_inner_product"]),
                "lr": [[hx(l), hx(r)] for l, r in op["lr"]], "delta": hx(op["delta"]), "sg": hx(op["sg"]), "z1": str(op["z1"]), "z2": str(op["z2"]),
                "sponge_state": [str(x) for x in st], "sponge_mode": mode, "sponge_count": cnt})
    print("opening", i, "ok", flush=True)
json.dump({"poseidon_constants": PP.NAME, "curve": 0, "k": K, "log2_domain": LOG2, "npub": NPUB, "n_comms": NCOMMS, "slot": SLOT, "n_points": NPTS,
           "openings": ops}, open(os.path.join(ROOT, "tests/golden/state_job_k15.json"), "w"), indent=0)
