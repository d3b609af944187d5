#!/usr/bin/env python3
"""Mints tests/golden/ipa_pallas_k15.json with the CPU oracle's IPA prover (oracle/ipa_ref.py): one valid opening
proof over the first 2^15 Pallas SRS points (the Pickles wrap-proof shape: k = 15), 4 commitments, 2 evaluation
points, under the default (UNPINNED) Poseidon constants.  Run in the build container: python tests/golden/gen_ipa_fixture.py
(IPA_FIXTURE_POLYS=45 IPA_FIXTURE_NAME=ipa_pallas_k15_c45.json mints the 45-commitment wrap-proof shape of SURVEY.md 8d C3.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from ipa_helpers import mint, to_abi  # noqa: E402
import mina_bridge_amd.poseidon_params as PP  # noqa: E402

CURVE, K = 0, 15
g, h = O.srs_create(CURVE, 1 << K, threads=os.cpu_count() or 4)
entry, sponge = mint(CURVE, g, h, K, n_polys=int(os.environ.get("IPA_FIXTURE_POLYS", "4")), n_points=2, seed=20250928)
abi = to_abi(entry, sponge)
out = {"curve": CURVE, "k": K, "poseidon_constants": PP.NAME,
       "fields": {k: (v.tobytes().hex() if hasattr(v, "tobytes") else int(v)) for k, v in abi.items()}}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("IPA_FIXTURE_NAME", "ipa_pallas_k15.json")), "w"), indent=1)
print("wrote", os.environ.get("IPA_FIXTURE_NAME", "ipa_pallas_k15.json"), {k: (len(v) // 2 if isinstance(v, str) else v) for k, v in out["fields"].items()})
