# dev tool: T host threads, one context each (the integration's "one mina_ctx per worker thread"), all on GPU 0, each running
# mina_ipa_batch_check on its own batch: does the serial transcript latency of one batch hide behind the others?
import json, os, sys, threading, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mina_bridge_amd as m
fx = json.load(open(os.path.join(ROOT, "tests/golden/ipa_pallas_k15_c45.json")))
a = {k: (np.frombuffer(bytes.fromhex(v), dtype=np.uint8).copy() if isinstance(v, str) else v) for k, v in fx["fields"].items()}
rb = np.zeros(32, np.uint8); rb[:8] = 7; sb = np.zeros(32, np.uint8); sb[:8] = 9
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for T in (1, 2, 4, 8):
    ctxs = []
    for _ in range(T):
        c = m.MinaContext(0)
        c.poseidon_set_params(0, m.poseidon_params.default_params_bytes(0)); c.srs_create(0, 65536)
        ctxs.append(c)
    ops = [c.pack_ipa_openings([a] * B) for c in ctxs]
    for c, o in zip(ctxs, ops): assert c.ipa_batch_check(0, o, rb, sb)
    reps = 6
    def work(c, o):
        for _ in range(reps): assert c.ipa_batch_check(0, o, rb, sb)
    th = [threading.Thread(target=work, args=(c, o)) for c, o in zip(ctxs, ops)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"threads={T} B={B}: {T*reps*B/dt:.0f} openings/s ({dt/reps*1e3:.1f} ms per round of {T} batches)")
    for c in ctxs: c.close()
