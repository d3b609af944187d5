# dev tool: does a culprit search (32 concurrent streams) leave the process slower?  8192- and 65 536-proof calls before and after one
import ctypes, json, os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import mina_bridge_amd as m
from mina_bridge_amd import lib as L
from kimchi_helpers import install_index, install_step_index, load_k15_fixture, make_step_index
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "state_proofs_k15_bytes.json")))
ix, _, _ = load_k15_fixture()
m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)
gctx = m.lib.verify_global_ctx()
install_index(gctx, ix); install_step_index(gctx, make_step_index(99))
proofs = [bytes.fromhex(p["proof"]) for p in fx["proofs"]]; pubs = [bytes.fromhex(p["pub"]) for p in fx["proofs"]]
lib = L.load_library()
def mk(size, bad=None):
    P = [proofs[i % 4] for i in range(size)]; Q = [pubs[i % 4] for i in range(size)]
    if bad is not None: b = bytearray(Q[bad]); b[40] ^= 1; Q[bad] = bytes(b)
    return (size,) + L._ptr_arrays(P)[1:] + L._ptr_arrays(Q)[1:] + (P, Q)
def run(a, reps):
    out = np.zeros(a[0], np.uint8); ts = []
    for _ in range(reps):
        t = time.perf_counter(); rc = lib.mina_verify_state_batch(ctypes.c_size_t(a[0]), a[1], a[2], a[3], a[4], L._p(out)); ts.append((time.perf_counter() - t) * 1e3); assert rc == 0
    return round(statistics.median(ts), 2), int(out.sum())
a8 = mk(8192); a64 = mk(65536); bad = mk(8192, 2730)
# a tampered OPENING (z1 + 1): the folded check of its chunk fails -> culprit search (32 concurrent streams)
import copy, random
from ipa_helpers import poseidon_pp
from kimchi_helpers import load_statement_fixture, make_chain
from wire_writers import state_proof_bytes
it = load_statement_fixture()[0][0]
states, hashes = make_chain(random.Random(it["chain_seed"]), poseidon_pp(0))
pz, ev = it["proof"], it["proof"]["evals"]
wrap = dict(it["wrap"])
wrap.update(w_comm=pz["w_comm"], z_comm=pz["z_comm"], t_comm=pz["t_comm"], z_eval=ev[0], selector_eval=ev[1:7], w_eval=ev[7:22], coefficients_eval=ev[22:37], s_eval=ev[37:43],
            ft_eval1=pz["ft_eval1"], lr=pz["opening"]["lr"], z1=(pz["opening"]["z1"] + 1) % (1 << 254), z2=pz["opening"]["z2"], delta=pz["opening"]["delta"], sg=pz["opening"]["sg"])
bad_proof = state_proof_bytes(wrap, states)
def mk_bad_opening(size, pos):
    P = [proofs[i % 4] for i in range(size)]; Q = [pubs[i % 4] for i in range(size)]
    assert pos % 4 == 0; P[pos] = bad_proof
    return (size,) + L._ptr_arrays(P)[1:] + L._ptr_arrays(Q)[1:] + (P, Q)
bad2 = mk_bad_opening(8192, 2732)
run(a8, 3); run(a64, 1)
print("before: 8192", run(a8, 10), " 65536", run(a64, 4))
print("tampered public input (no search):", run(bad, 1))
print("tampered opening (culprit search), three calls:", run(bad2, 1), run(bad2, 1), run(bad2, 1))
print("after:  8192", run(a8, 10), " 65536", run(a64, 4))
