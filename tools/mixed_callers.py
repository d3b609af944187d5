import ctypes, json, os, sys, time, threading, random, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import mina_bridge_amd as m
from mina_bridge_amd import lib as L
from kimchi_helpers import install_index, install_step_index, load_k15_fixture, make_step_index
from ipa_helpers import poseidon_pp
from oracle import mina_account_ref as A, pasta_ref as R
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "state_proofs_k15_bytes.json")))
ix, _, _ = load_k15_fixture()
m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)
gctx = m.lib.verify_global_ctx()
install_index(gctx, ix); install_step_index(gctx, make_step_index(99))
proofs = [bytes.fromhex(p["proof"]) for p in fx["proofs"]]; pubs = [bytes.fromhex(p["pub"]) for p in fx["proofs"]]
n=8192
P=[proofs[i%4] for i in range(n)]; Q=[pubs[i%4] for i in range(n)]
pa,PP,PL=L._ptr_arrays(P); qa,QQ,QL=L._ptr_arrays(Q)
pp = poseidon_pp(0); rng = random.Random(31)
accounts = [A.synth_account(rng, zk, timed, deleg, with_vk=vk) for zk, timed, deleg, vk in [(False, False, False, True), (True, True, True, True), (True, False, True, False), (False, True, True, True)]]
ap, aq = [], []
for a in accounts:
    path = [(rng.randrange(2), rng.randrange(R.P)) for _ in range(35)]; enc = A.abi_encode_account(a)
    ap.append(A.write_account_proof(path, a)); aq.append(R.merkle_root(A.account_hash(a, pp), path, pp).to_bytes(32, "little") + struct.pack("<Q", len(enc)) + enc)
na=256
AP=[ap[i%4] for i in range(na)]; AQ=[aq[i%4] for i in range(na)]
_,APP,APL=L._ptr_arrays(AP); _,AQQ,AQL=L._ptr_arrays(AQ)
lib=L.load_library()
def run(ns, nacc, secs=3.0):
    stop=[False]; cs=[0]*ns; ca=[0]*nacc; lat=[[] for _ in range(nacc)]
    def sw(i):
        out=np.zeros(n,np.uint8)
        while not stop[0]:
            assert lib.mina_verify_state_batch(ctypes.c_size_t(n),PP,PL,QQ,QL,L._p(out))==0; cs[i]+=1
    def aw(i):
        out=np.zeros(na,np.uint8)
        while not stop[0]:
            t=time.perf_counter(); assert lib.mina_verify_account_batch(ctypes.c_size_t(na),APP,APL,AQQ,AQL,L._p(out))==0; lat[i].append(time.perf_counter()-t); ca[i]+=1
    th=[threading.Thread(target=sw,args=(i,)) for i in range(ns)]+[threading.Thread(target=aw,args=(i,)) for i in range(nacc)]
    t0=time.perf_counter()
    for t in th: t.start()
    time.sleep(secs); stop[0]=True
    for t in th: t.join()
    dt=time.perf_counter()-t0
    al=sorted(x for l in lat for x in l)
    print(json.dumps({"state_callers":ns,"account_callers":nacc,"state_proofs_per_s":round(sum(cs)*n/dt),"account_proofs_per_s":round(sum(ca)*na/dt),"account_ms_median":round(al[len(al)//2]*1e3,2) if al else None}))
run(2,0,1.0); run(0,2,1.0)
run(2,0); run(0,4); run(2,4); run(4,4)
