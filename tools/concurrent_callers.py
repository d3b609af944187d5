#!/usr/bin/env python3
"""The reference's call pattern: `mina_verify_state(proof, pub) -> bool`, ONE proof per call, from N threads at once (SURVEY.md 8b:
goroutines of Aligned's operator).  Calls that arrive while a job runs leave together as the next job (api_verify.hip), so the rate
grows with the number of callers although every call still hands over one proof.  Full-size proofs of tests/golden/statement_k15.json.
usage: concurrent_callers.py [CALLS_PER_THREAD]     prints one JSON line per thread count."""
import json
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.setrecursionlimit(10000)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")      # before HIP initialises: the pipeline's lanes need a hardware queue each
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
from ipa_helpers import poseidon_pp
from kimchi_helpers import install_index, install_step_index, load_k15_fixture, load_statement_fixture, make_chain, make_step_index
from oracle import mina_state_ref as S
from wire_writers import state_proof_bytes, state_pub_bytes

per_thread = int(sys.argv[1]) if len(sys.argv) > 1 else 4
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
    cases.append((state_proof_bytes(wrap, states), state_pub_bytes(True, hashes[16], hashes[:16], ledger)))
assert all(m.lib.verify_state(p, q) for p, q in cases)
# the callers as pthreads of a small C helper (no interpreter lock between a verdict and the next call: what a Rust / Go operator's tasks do)
import ctypes, subprocess, tempfile
helper_src = r"""
#include <pthread.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
bool mina_verify_state(const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len);
struct job { const uint8_t *const *proofs; const size_t *pl; const uint8_t *const *pubs; const size_t *ql; int ncases, calls, t; long bad; };
static void *worker(void *a) { struct job *j = a; for (int k = 0; k < j->calls; ++k) { int c = (j->t + k) % j->ncases; if (!mina_verify_state(j->proofs[c], j->pl[c], j->pubs[c], j->ql[c])) j->bad++; } return 0; }
long run_callers(int nthreads, int calls, int ncases, const uint8_t *const *proofs, const size_t *pl, const uint8_t *const *pubs, const size_t *ql) {
  pthread_t th[1024]; struct job jobs[1024]; long bad = 0;
  for (int t = 0; t < nthreads; ++t) { jobs[t] = (struct job){proofs, pl, pubs, ql, ncases, calls, t, 0}; pthread_create(&th[t], 0, worker, &jobs[t]); }
  for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], 0); bad += jobs[t].bad; }
  return bad; }
"""
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "callers.c"), "w").write(helper_src)
libdir = os.path.dirname(m.LIB_PATH)
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", os.path.join(tmp, "callers.c"), "-o", os.path.join(tmp, "libcallers.so"), "-L", libdir, "-lminaverify", "-Wl,-rpath," + libdir])
helper = ctypes.CDLL(os.path.join(tmp, "libcallers.so"))
helper.run_callers.restype = ctypes.c_long
nc = len(cases)
c_pr = (ctypes.c_char_p * nc)(*[c[0] for c in cases]); c_pl = (ctypes.c_size_t * nc)(*[len(c[0]) for c in cases])
c_pu = (ctypes.c_char_p * nc)(*[c[1] for c in cases]); c_ql = (ctypes.c_size_t * nc)(*[len(c[1]) for c in cases])
for nthreads in (1, 4, 16, 64, 256, 1024):
    helper.run_callers(nthreads, 2, nc, c_pr, c_pl, c_pu, c_ql)
    t0 = time.perf_counter()
    bad = helper.run_callers(nthreads, per_thread, nc, c_pr, c_pl, c_pu, c_ql)
    dt = time.perf_counter() - t0
    assert bad == 0
    print(json.dumps({"callers": "pthreads", "threads": nthreads, "calls": nthreads * per_thread, "seconds": round(dt, 4), "proofs_per_s": round(nthreads * per_thread / dt, 1),
                      "ms_per_call_seen_by_a_thread": round(dt / per_thread * 1e3, 2)}))
for nthreads in (1, 4, 16, 64, 256):
    bad = [0]
    def worker(t):
        for k in range(per_thread):
            p, q = cases[(t + k) % len(cases)]
            if not m.lib.verify_state(p, q): bad[0] += 1
    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    assert bad[0] == 0
    print(json.dumps({"threads": nthreads, "calls": nthreads * per_thread, "seconds": round(dt, 4), "proofs_per_s": round(nthreads * per_thread / dt, 1),
                      "ms_per_call_seen_by_a_thread": round(dt / per_thread * 1e3, 2)}), flush=True)
