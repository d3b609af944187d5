# dev tool: A/B of the boundary pipeline's knobs INSIDE one process (boxes differ by more than the knobs do): the settings are cycled call by
# call -- fields of mina_verify_tuning (include/mina_verify.h), set through mina_verify_configure_ex between calls -- and the median time per setting is reported.
# usage: python tools/boundary_ab.py SIZE ROUNDS "chunk=4096,slots=8" "hash_piece_waves=512" ...      ("-" = library defaults)
import ctypes, json, os, random, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import mina_bridge_amd as m
from mina_bridge_amd import lib as L
from kimchi_helpers import install_index, install_step_index, load_k15_fixture, make_step_index

size, rounds = int(sys.argv[1]), int(sys.argv[2])
configs = sys.argv[3:] or ["-"]
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "state_proofs_k15_bytes.json")))
ix, _, _ = load_k15_fixture()
m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)
gctx = m.lib.verify_global_ctx()
install_index(gctx, ix); install_step_index(gctx, make_step_index(99))
proofs = [bytes.fromhex(p["proof"]) for p in fx["proofs"]]; pubs = [bytes.fromhex(p["pub"]) for p in fx["proofs"]]
P = [proofs[i % len(proofs)] for i in range(size)]; Q = [pubs[i % len(pubs)] for i in range(size)]
pa, PP, PL = L._ptr_arrays(P); qa, QQ, QL = L._ptr_arrays(Q)
out = np.zeros(size, np.uint8)
lib = L.load_library()
def call():
    t = time.perf_counter()
    rc = lib.mina_verify_state_batch(ctypes.c_size_t(size), PP, PL, QQ, QL, L._p(out))
    dt = time.perf_counter() - t
    assert rc == 0 and out.all()
    return dt * 1e3
def apply(cfg):
    if cfg == "-": m.lib.verify_configure_ex(None)
    else: m.lib.tune_from_string(cfg.replace(" ", ","))
for cfg in configs: apply(cfg); call(); call()
times = {cfg: [] for cfg in configs}
for r in range(rounds):
    for cfg in configs:
        apply(cfg); times[cfg].append(call())
for cfg in configs:
    t = sorted(times[cfg])
    print(json.dumps({"config": cfg, "median_ms": round(statistics.median(t), 2), "min_ms": round(t[0], 2), "p90_ms": round(t[int(len(t) * 0.9)], 2), "proofs_per_s": round(size / statistics.median(t) * 1e3)}))
