# dev tool: latency of the PCIe-inclusive host-buffer entry point mina_state_job_batch (full-size C3 job) for small batches,
# for the prepared-rows job and for the full job from parsed proofs (statements -> public inputs -> kimchi -> opening -> accumulator)
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch; torch.cuda.is_available()
import mina_bridge_amd as m
import bench
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_partial_inputs as P
ctx = m.MinaContext(0)
for f in (0, 1):
    ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(1, 1 << 16); ctx.srs_create(0, 1 << 16)
for mode in ("prepared", "full"):
    for B in (1, 16, 256):
        (hj, keep), _ = P.build_batch(ctx, m, B, seed=5)
        if mode == "full":
            kp, op, acc, _ = P.build_full_section(ctx, m, B)
            for name in ("sponge_state", "sponge_pos", "cip", "evalpoints", "evalscale", "polyscale", "comms", "public_inputs"):
                setattr(hj, name, None)
            keep = list(keep) + [kp]
            for name, arr in list(op.items()) + list(acc.items()):
                arr = np.ascontiguousarray(arr); keep.append(arr); setattr(hj, name, arr.ctypes.data)
            hj.n_comms = 47
            import ctypes
            hj.kimchi = ctypes.addressof(kp[0])
        for _ in range(2):
            assert ctx.state_job_batch((hj, keep)).tolist() == [1] * B
        t = time.perf_counter(); reps = 5
        for _ in range(reps):
            ctx.state_job_batch((hj, keep))
        dt = (time.perf_counter() - t) / reps
        print(json.dumps({"mode": mode, "jobs_per_call": B, "ms_per_call": round(dt * 1e3, 2), "jobs_per_s": round(B / dt, 1)}))
