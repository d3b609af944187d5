# dev tool: BASELINE config C4 through the boundary -- mina_verify_account_batch(proof bytes, public-input bytes) over N Proof-of-Account
# pairs (4 distinct accounts, depth-35 Merkle paths): host parsing + ABI cross-check, four dependent Poseidon stages (zkapp uri /
# verification key -> zkapp -> account hash) and the Merkle fold on the GPU.  PCIe and host side included.
import json, os, random, struct, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
from ipa_helpers import poseidon_pp
from oracle import mina_account_ref as A, pasta_ref as R

pp = poseidon_pp(0)
rng = random.Random(31)
accounts = [A.synth_account(rng, zk, timed, deleg, with_vk=vk) for zk, timed, deleg, vk in
            [(False, False, False, True), (True, True, True, True), (True, False, True, False), (False, True, True, True)]]
proofs, pubs = [], []
for a in accounts:
    leaf = A.account_hash(a, pp)
    path = [(rng.randrange(2), rng.randrange(R.P)) for _ in range(35)]
    enc = A.abi_encode_account(a)
    proofs.append(A.write_account_proof(path, a)); pubs.append(R.merkle_root(leaf, path, pp).to_bytes(32, "little") + struct.pack("<Q", len(enc)) + enc)
m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)            # the compiled-in Poseidon tables are the surrogate set: the boundary answers `false` without the flag
print(json.dumps({"proof_bytes": [len(p) for p in proofs], "pub_bytes": [len(q) for q in pubs]}))
for n in (1, 256, 4096, 16384):
    P = [proofs[i % 4] for i in range(n)]; Q = [pubs[i % 4] for i in range(n)]
    assert m.lib.verify_account_batch(P, Q).all()
    t = time.perf_counter(); reps = 8
    for _ in range(reps):
        m.lib.verify_account_batch(P, Q)
    dt = (time.perf_counter() - t) / reps
    print(json.dumps({"proofs_per_call": n, "ms_per_call": round(dt * 1e3, 2), "proofs_per_s": round(n / dt, 1)}))
