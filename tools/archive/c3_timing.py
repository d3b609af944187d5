# dev tool: BASELINE config C3 -- "full Pickles step+wrap verify, chain of 16" -- as a synthetic but shape-faithful job
# (SURVEY.md 8d): per state proof
#   * 17 protocol-state hashes (48 Fp elements each, Poseidon sponge)                     -> mina_poseidon_hash
#   * the wrap proof's public-input commitment (40 Fq scalars, Pallas domain 2^15)        -> mina_public_input_commitment_batch
#   * the wrap proof's IPA opening (Pallas, k = 15, 45 commitments x 2 evaluation points) -> mina_ipa_batch_check
#   * the step accumulator check (Vesta, 2^16 bases)                                     -> mina_accumulator_check_batch
# B jobs per call through the HOST-BUFFER entry points (H2D/D2H and Python packing included).  What is NOT in the job:
# kimchi's oracles / linearisation scalars and the binprot parsing (not built).
import json, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mina_bridge_amd as m
import bench

fx = json.load(open(os.path.join(ROOT, "tests/golden/ipa_pallas_k15_c45.json")))
opening = {k: (np.frombuffer(bytes.fromhex(v), dtype=np.uint8).copy() if isinstance(v, str) else v) for k, v in fx["fields"].items()}
ctx = m.MinaContext(0)
for f in (0, 1):
    ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(0, 65536); ctx.srs_create(1, 65536)
pre1, sg1 = bench.make_instances(ctx, 1, seed=3)
rng = np.random.default_rng(5)
rb = np.zeros(32, np.uint8); rb[:8] = 7
sb = np.zeros(32, np.uint8); sb[:8] = 9


def field_elems(n):
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8); a[:, 31] &= 0x3f
    return a


def run(B, reps):
    msgs = field_elems(17 * B * 48)
    pub = field_elems(B * 40)
    pre = np.repeat(pre1, B, axis=0); sg = np.repeat(sg1, B, axis=0)
    rho = field_elems(B)
    ops = ctx.pack_ipa_openings([opening] * B)        # ctypes packing is the caller's cost, not the library's
    stages = {"state_hashes": 0.0, "public_comm": 0.0, "wrap_ipa": 0.0, "step_accumulator": 0.0}
    for it in range(reps + 1):
        t = [time.perf_counter()]
        ctx.poseidon_hash(0, msgs, 17 * B, 48); t.append(time.perf_counter())
        ctx.public_input_commitment_batch(0, 15, pub, B); t.append(time.perf_counter())
        ok = ctx.ipa_batch_check(0, ops, rb, sb); t.append(time.perf_counter())
        v = ctx.accumulator_check_batch(1, 16, pre, sg, rho); t.append(time.perf_counter())
        assert ok and v.all()
        if it:                                      # first pass warms caches (Lagrange basis, tables, buffers)
            for k, d in zip(stages, np.diff(t)):
                stages[k] += d / reps
    total = sum(stages.values())
    print(json.dumps({"config": "C3 synthetic", "jobs_per_call": B, "ms_per_call": round(total * 1e3, 2), "jobs_per_s": round(B / total, 1),
                      "stage_ms": {k: round(v * 1e3, 2) for k, v in stages.items()}}))


for B, reps in ((1, 5), (16, 5), (256, 3), (1024, 2)):
    run(B, reps)
