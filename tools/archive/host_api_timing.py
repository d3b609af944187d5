# dev tool: PCIe-inclusive rate of the host-buffer entry point mina_accumulator_check_batch (H2D of inputs, kernels,
# synchronous verdict read-back) for B = 1 and B = 256
import os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, '.')
import mina_bridge_amd as m, bench
ctx = m.MinaContext(0); ctx.srs_create(1, 65536)
for B in (1, 256):
    pre, sg = bench.make_instances(ctx, min(B, 4), 3)
    pre = np.tile(pre, (B // min(B, 4), 1, 1)); sg = np.tile(sg, (B // min(B, 4), 1))
    rho = np.random.default_rng(1).integers(0, 256, (B, 32), dtype=np.uint8); rho[:, 31] &= 0x3f
    assert ctx.accumulator_check_batch(1, 16, pre.reshape(-1), sg.reshape(-1), rho if B > 1 else None).all()
    t0 = time.perf_counter(); reps = 200 if B == 1 else 50
    for _ in range(reps): ctx.accumulator_check_batch(1, 16, pre.reshape(-1), sg.reshape(-1), rho if B > 1 else None)
    dt = (time.perf_counter() - t0) / reps
    print(f"B={B}: {dt*1e3:.3f} ms per call, {B/dt:.0f} proofs/s (host buffers, synchronous)")
