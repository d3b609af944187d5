# dev tool: the C3 synthetic state-proof job of tools/c3_timing.py on T host threads, one context each, all on GPU 0
# (the reference's callers verify concurrently from tokio tasks / goroutines; a context serves one thread)
import json, os, sys, threading, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mina_bridge_amd as m
import bench
fx = json.load(open(os.path.join(ROOT, "tests/golden/ipa_pallas_k15_c45.json")))
opening = {k: (np.frombuffer(bytes.fromhex(v), dtype=np.uint8).copy() if isinstance(v, str) else v) for k, v in fx["fields"].items()}
rb = np.zeros(32, np.uint8); rb[:8] = 7
sb = np.zeros(32, np.uint8); sb[:8] = 9
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(5)
def fe(n):
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8); a[:, 31] &= 0x3f
    return a
msgs, pub, rho = fe(17 * B * 48), fe(B * 40), fe(B)
for T in (1, 2, 4, 8):
    ctxs = []
    for _ in range(T):
        c = m.MinaContext(0)
        for f in (0, 1): c.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
        c.srs_create(0, 65536); c.srs_create(1, 65536)
        ctxs.append(c)
    pre1, sg1 = bench.make_instances(ctxs[0], 1, seed=3)
    pre = np.repeat(pre1, B, axis=0); sg = np.repeat(sg1, B, axis=0)
    ops = [c.pack_ipa_openings([opening] * B) for c in ctxs]
    def job(c, o):
        c.poseidon_hash(0, msgs, 17 * B, 48)
        c.public_input_commitment_batch(0, 15, pub, B)
        assert c.ipa_batch_check(0, o, rb, sb)
        assert c.accumulator_check_batch(1, 16, pre, sg, rho).all()
    for c, o in zip(ctxs, ops): job(c, o)
    reps = 4
    def work(c, o):
        for _ in range(reps): job(c, o)
    th = [threading.Thread(target=work, args=(c, o)) for c, o in zip(ctxs, ops)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "C3 synthetic, threads", "threads": T, "jobs_per_call": B, "jobs_per_s": round(T * reps * B / dt, 1),
                      "ms_per_round": round(dt / reps * 1e3, 2)}))
    for c in ctxs: c.close()
