# dev tool: un-folded group checks (mina_accumulator_check_multi_dev): proofs/s vs proofs per kernel pipeline and lanes
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, mina_bridge_amd as m, bench
ctx = m.MinaContext(0); ctx.srs_create(1, 65536)
dev = torch.device('cuda', 0)
pre4, sg4 = bench.make_instances(ctx, 4, 1)
for lanes in (16, 8):
    ctx.set_pipeline(lanes)
    for M in (1, 2, 4, 8, 16):
        pre = np.concatenate([pre4[i % 4] for i in range(M)]); sg = np.stack([sg4[i % 4] for i in range(M)])
        d_pre = torch.from_numpy(pre.reshape(-1)).to(dev); d_sg = torch.from_numpy(sg.reshape(-1)).to(dev)
        d_v = torch.zeros(M, dtype=torch.int32, device=dev); torch.cuda.synchronize()
        def step(): ctx.accumulator_check_multi_dev(1, 16, M, d_pre.data_ptr(), d_sg.data_ptr(), d_v.data_ptr())
        for _ in range(2 * lanes): step()
        ctx.synchronize()
        assert d_v.cpu().numpy().tolist() == [1] * M
        n = max(64, 2048 // M)
        t0 = time.perf_counter()
        for _ in range(n): step()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print(f"lanes={lanes} group={M}: {1e6*dt/n/M:.1f} us/proof, {n*M/dt:.0f} proofs/s")
