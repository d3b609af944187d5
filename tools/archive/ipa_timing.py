# dev tool: wall time of mina_ipa_batch_check (host-buffer API: packs, H2D, transcript kernel, fold, 2 MSMs, compare, sync)
# on the committed full-size opening (Pallas, k = 15) replicated B times
import json, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import mina_bridge_amd as m
fx = json.load(open('tests/golden/ipa_pallas_k15.json'))
a = {k: (np.frombuffer(bytes.fromhex(v), dtype=np.uint8).copy() if isinstance(v, str) else v) for k, v in fx['fields'].items()}
ctx = m.MinaContext(0)
for f in (0, 1): ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(0, 65536)
rb = np.zeros(32, np.uint8); rb[:8] = 7; sb = np.zeros(32, np.uint8); sb[:8] = 9
for B in (1, 16, 256, 1024):
    ops = ctx.pack_ipa_openings([a] * B)              # ctypes packing once: time the C entry point
    assert ctx.ipa_batch_check(0, ops, rb, sb)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): ok = ctx.ipa_batch_check(0, ops, rb, sb)
    dt = (time.perf_counter() - t0) / reps
    print(f"B={B}: {dt*1e3:.1f} ms per batch, {B/dt:.0f} openings/s, verdict {ok}")
