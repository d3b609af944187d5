# dev tool: is the bench host-bound?  time to ISSUE n steps vs time until they are done
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, '.')
import torch, mina_bridge_amd as m, bench
ctx = m.MinaContext(0); ctx.srs_create(1, 65536); ctx.set_pipeline(int(sys.argv[1]) if len(sys.argv) > 1 else 16)
pre, sgs = bench.make_instances(ctx, 1, 1)
dev = torch.device('cuda', 0)
d_pre = torch.from_numpy(pre.reshape(-1)).to(dev); d_sg = torch.from_numpy(sgs.reshape(-1)).to(dev)
d_v = torch.zeros(1, dtype=torch.int32, device=dev); torch.cuda.synchronize()
def step(): ctx.accumulator_check_dev(1, 16, 1, d_pre.data_ptr(), d_sg.data_ptr(), 0, d_v.data_ptr())
for _ in range(64): step()
ctx.synchronize()
n = 1000
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
ctx.synchronize()
t2 = time.perf_counter()
print(f"issue {1e6*(t1-t0)/n:.1f} us/step, total {1e6*(t2-t0)/n:.1f} us/step")
