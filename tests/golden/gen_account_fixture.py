"""Generates tests/golden/account_proofs_bytes.json: BASELINE config C4's batch -- 256 DISTINCT Proof-of-Account pairs as the reference's caller sends them
(core/src/aligned.rs:31-58: bincode `MinaAccountProof` = depth-35 Merkle path + account, core/src/proof/account_proof.rs:9-35; `MinaAccountPubInputs` = ledger hash ||
u64 length || Solidity-ABI account, core/src/sol/serialization.rs:63-86), base64.  Accounts cycle through the four shapes (plain / zkapp + timing + delegate /
zkapp without verification key / timed + delegate); Merkle roots are computed by the repo's CPU oracle (oracle/mina_account_ref.py, pasta_ref.merkle_root) under
the Poseidon constant set named in the file, so every pair ACCEPTs.  bench.py's `c4_account_256` leg reads only this file (nothing under oracle/ outside its
cpu_baseline leg); tests/test_merkle.py re-derives a sample of the roots.
Run:  python tests/golden/gen_account_fixture.py [count]      (~1 min)"""
import base64
import json
import os
import random
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ipa_helpers import poseidon_pp
from oracle import mina_account_ref as A, pasta_ref as R
import mina_bridge_amd.poseidon_params as PP

count = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(False, False, False, True), (True, True, True, True), (True, False, True, False), (False, True, True, True)]
pp = poseidon_pp(0)
rng = random.Random(0xACC0)
b64 = lambda b: base64.b64encode(b).decode()
out = {"poseidon_constants": PP.NAME, "merkle_depth": 35, "seed": 0xACC0, "proofs": []}
for i in range(count):
    zk, timed, deleg, vk = SHAPES[i % 4]
    a = A.synth_account(rng, zk, timed, deleg, with_vk=vk)
    path = [(rng.randrange(2), rng.randrange(R.P)) for _ in range(35)]
    enc = A.abi_encode_account(a)
    root = R.merkle_root(A.account_hash(a, pp), path, pp)
    out["proofs"].append({"proof": b64(A.write_account_proof(path, a)), "pub": b64(root.to_bytes(32, "little") + struct.pack("<Q", len(enc)) + enc)})
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "account_proofs_bytes.json"), "w"), indent=0)
print(count, "pairs;", sum(len(p["proof"]) + len(p["pub"]) for p in out["proofs"]), "base64 characters")
