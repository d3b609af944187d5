import json, os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
sys.argv = ["bench.py"]
import bench
fxb = json.load(open(os.path.join(ROOT, "tests/golden/state_proofs_k15_bytes.json")))
fx, un = bench.load_encoded_fixture()
m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)
bench.install_fixture_indexes(m.lib.verify_all_devices(), fx, un)
items = [(bytes.fromhex(it["proof"]), bytes.fromhex(it["pub"])) for it in fxb["proofs"]]
for B, pos in ((70, 33), (1024, 500), (3000, 1500), (8192, 2730)):
    P = [items[i % 4][0] for i in range(B)]; Q = [items[i % 4][1] for i in range(B)]
    out = m.lib.verify_state_batch(P, Q); print(B, "clean", int(out.sum()))
    Q[pos] = items[(pos + 1) % 4][1]      # the public input of ANOTHER proof: chain and, through the application state, the kimchi step fail
    out = m.lib.verify_state_batch(P, Q); print(B, pos, "tampered pub ->", int(out.sum()), np.flatnonzero(out == 0)[:8].tolist(), flush=True)
