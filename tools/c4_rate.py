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

# BASELINE C4 from several caller threads: batches of 256 share jobs (call merging), pthreads of a C helper
import ctypes, subprocess, tempfile
src = r"""
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
int mina_verify_account_batch(size_t n, const uint8_t *const *proofs, const size_t *pl, const uint8_t *const *pubs, const size_t *ql, uint8_t *out);
struct job { size_t n; const uint8_t *const *proofs; const size_t *pl; const uint8_t *const *pubs; const size_t *ql; int calls; long bad; };
static void *worker(void *a) { struct job *j = a; uint8_t out[4096]; for (int k = 0; k < j->calls; ++k) { if (mina_verify_account_batch(j->n, j->proofs, j->pl, j->pubs, j->ql, out)) j->bad++; for (size_t i = 0; i < j->n; ++i) if (!out[i]) j->bad++; } return 0; }
long run_callers(int nthreads, int calls, size_t n, const uint8_t *const *proofs, const size_t *pl, const uint8_t *const *pubs, const size_t *ql) {
  pthread_t th[256]; struct job jobs[256]; long bad = 0;
  for (int t = 0; t < nthreads; ++t) { jobs[t] = (struct job){n, proofs, pl, pubs, ql, calls, 0}; pthread_create(&th[t], 0, worker, &jobs[t]); }
  for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], 0); bad += jobs[t].bad; }
  return bad; }
"""
tmp = tempfile.mkdtemp(); open(os.path.join(tmp, "c.c"), "w").write(src)
libdir = os.path.dirname(m.LIB_PATH)
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", os.path.join(tmp, "c.c"), "-o", os.path.join(tmp, "libc4.so"), "-L", libdir, "-lminaverify", "-Wl,-rpath," + libdir])
h = ctypes.CDLL(os.path.join(tmp, "libc4.so")); h.run_callers.restype = ctypes.c_long
n = 256
P = [proofs[i % 4] for i in range(n)]; Q = [pubs[i % 4] for i in range(n)]
cP = (ctypes.c_char_p * n)(*P); cPL = (ctypes.c_size_t * n)(*map(len, P)); cQ = (ctypes.c_char_p * n)(*Q); cQL = (ctypes.c_size_t * n)(*map(len, Q))
for nthreads in (1, 2, 4, 8, 16):
    h.run_callers(nthreads, 2, n, cP, cPL, cQ, cQL)
    t = time.perf_counter(); calls = 24
    bad = h.run_callers(nthreads, calls, n, cP, cPL, cQ, cQL)
    dt = time.perf_counter() - t
    assert bad == 0
    print(json.dumps({"caller_threads": nthreads, "proofs_per_call": n, "proofs_per_s": round(nthreads * calls * n / dt, 1), "ms_per_call_seen_by_a_thread": round(dt / calls * 1e3, 2)}))
