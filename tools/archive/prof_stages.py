# per-stage HIP-event breakdown of one accumulator-check step (dev tool)
import sys, json, numpy as np, torch
sys.path.insert(0,'.'); import os; os.environ.setdefault('GPU_MAX_HW_QUEUES','16')
import mina_bridge_amd as m, bench
ctx=m.MinaContext(0); ctx.srs_create(1,65536)
B=int(sys.argv[1]) if len(sys.argv)>1 else 1
pre,sgs=bench.make_instances(ctx,B,1)
dev=torch.device('cuda',0)
d_pre=torch.from_numpy(pre.reshape(-1)).to(dev); d_sg=torch.from_numpy(sgs.reshape(-1)).to(dev)
rho=np.random.default_rng(1).integers(0,256,(B,32),dtype=np.uint8); rho[:,31]&=0x3f
d_rho=torch.from_numpy(rho.reshape(-1)).to(dev); d_v=torch.zeros(1,dtype=torch.int32,device=dev)
torch.cuda.synchronize()
def step(): ctx.accumulator_check_dev(1,16,B,d_pre.data_ptr(),d_sg.data_ptr(),d_rho.data_ptr() if B>1 else 0,d_v.data_ptr())
for _ in range(5): step()
ctx.synchronize()
ctx.prof_enable(-1)
for _ in range(20): step()
p=ctx.prof_read()
print(json.dumps({k:round(v[1]/v[0]*1000,1) for k,v in p.items()}), 'us per launch; verdict',int(d_v.item()))
