# dev tool: throughput of the reference-shaped boundary at the real sizes -- mina_verify_state_batch(proof bytes, public-input bytes) over
# N serialized MinaStateProofs (the four of tests/golden/statement_k15.json, repeated): host parsing (bincode / bin_prot), to_input
# flattening, packing, ONE upload, the whole GPU job, one download.  PCIe and the host side included: this is never bench.py's `value`.
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")      # before HIP initialises: the pipeline's lanes need a hardware queue each
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
m.lib.tune_from_string(os.environ.get("MINA_TUNE", ""))      # e.g. MINA_TUNE=chunk=4096,slots=8 (fields of mina_verify_tuning)
from ipa_helpers import poseidon_pp
from kimchi_helpers import install_index, install_step_index, load_k15_fixture, load_statement_fixture, make_chain, make_step_index
from oracle import mina_state_ref as S
from wire_writers import state_proof_bytes, state_pub_bytes

ix, _, _ = load_k15_fixture()
items, _ = load_statement_fixture()
m.lib.verify_configure(m.lib.VERIFY_ALLOW_SURROGATE)
gctx = m.lib.verify_global_ctx()
install_index(gctx, ix); install_step_index(gctx, make_step_index(99))
proofs, pubs = [], []
for it in items:
    states, hashes = make_chain(random.Random(it["chain_seed"]), poseidon_pp(0))
    p, ev = it["proof"], it["proof"]["evals"]
    wrap = dict(it["wrap"])
    wrap.update(w_comm=p["w_comm"], z_comm=p["z_comm"], t_comm=p["t_comm"], z_eval=ev[0], selector_eval=ev[1:7], w_eval=ev[7:22], coefficients_eval=ev[22:37], s_eval=ev[37:43],
                ft_eval1=p["ft_eval1"], lr=p["opening"]["lr"], z1=p["opening"]["z1"], z2=p["opening"]["z2"], delta=p["opening"]["delta"], sg=p["opening"]["sg"])
    proofs.append(state_proof_bytes(wrap, states)); pubs.append(state_pub_bytes(True, hashes[16], hashes[:16], [S.snarked_ledger_hash(s) for s in states[:16]]))
print(json.dumps({"proof_bytes": len(proofs[0]), "pub_bytes": len(pubs[0])}))
if os.environ.get("BOUNDARY_RATE_POLLUTE"):        # a process that held another context with many streams before (bench.py): the runtime's queue pool is then populated
    c0 = m.MinaContext(0); c0.set_pipeline(int(os.environ["BOUNDARY_RATE_POLLUTE"]))
    import numpy as np
    c0.poseidon_set_params(0, m.poseidon_params.default_params_bytes(0))
    for _ in range(64): c0.poseidon_permute(0, np.zeros((4, 96), np.uint8))
    if os.environ.get("BOUNDARY_RATE_POLLUTE_KEEP") is None: c0.close()
sizes = [int(x) for x in sys.argv[1:] if x.isdigit()] or [1, 64, 1024, 4096, 8192]
for n in sizes:
    P = [proofs[i % 4] for i in range(n)]; Q = [pubs[i % 4] for i in range(n)]
    assert m.lib.verify_state_batch(P, Q).all()
    t = time.perf_counter(); reps = 5
    for _ in range(reps):
        m.lib.verify_state_batch(P, Q)
    dt = (time.perf_counter() - t) / reps
    print(json.dumps({"proofs_per_call": n, "ms_per_call": round(dt * 1e3, 2), "proofs_per_s": round(n / dt, 1)}))
