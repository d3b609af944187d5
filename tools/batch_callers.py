# dev tool: N threads calling mina_verify_state_batch with SIZE full-size proofs each, CALLS times (a batcher's tasks): aggregate rate
# usage: python tools/batch_callers.py THREADS SIZE CALLS
import ctypes, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import mina_bridge_amd as m
from mina_bridge_amd import lib as L
from kimchi_helpers import install_index, install_step_index, load_k15_fixture, make_step_index
nthreads, size, calls = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "state_proofs_k15_bytes.json")))
ix, _, _ = load_k15_fixture()
m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)
gctx = m.lib.verify_global_ctx()
install_index(gctx, ix); install_step_index(gctx, make_step_index(99))
proofs = [bytes.fromhex(p["proof"]) for p in fx["proofs"]]; pubs = [bytes.fromhex(p["pub"]) for p in fx["proofs"]]
P = [proofs[i % len(proofs)] for i in range(size)]; Q = [pubs[i % len(pubs)] for i in range(size)]
pa, PP, PL = L._ptr_arrays(P); qa, QQ, QL = L._ptr_arrays(Q)
lib = L.load_library()
L.tune_from_string(os.environ.get("MINA_TUNE", ""))      # e.g. MINA_TUNE=split_max=0,slots=8
def worker(k):
    out = np.zeros(size, np.uint8)
    for _ in range(k):
        rc = lib.mina_verify_state_batch(ctypes.c_size_t(size), PP, PL, QQ, QL, L._p(out)); assert rc == 0 and out.all()
# warm-up with the same number of threads: every slot the run will use allocates its page-locked staging (~480 MB each) on first use
wu=[threading.Thread(target=worker,args=(2,)) for _ in range(nthreads)]
for x in wu: x.start()
for x in wu: x.join()
th=[threading.Thread(target=worker,args=(calls,)) for _ in range(nthreads)]
t=time.perf_counter()
for x in th: x.start()
for x in th: x.join()
dt=time.perf_counter()-t
print(json.dumps({"threads":nthreads,"size":size,"proofs_per_s":round(nthreads*calls*size/dt),"ms_per_call":round(dt/calls*1e3,2), "tune": os.environ.get("MINA_TUNE", "default")}))
