# per-stage HIP-event breakdown of one un-folded group step (mina_accumulator_check_multi_dev), one lane, nothing overlapped
import sys, json, os, numpy as np
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mina_bridge_amd as m, bench
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = m.MinaContext(0); ctx.srs_create(1, 65536)
pre4, sg4 = bench.make_instances(ctx, 4, 1)
pre = np.concatenate([pre4[i % 4] for i in range(G)]); sg = np.stack([sg4[i % 4] for i in range(G)])
d_pre = ctx.dev_upload(ctx.dev_malloc(pre.size), pre); d_sg = ctx.dev_upload(ctx.dev_malloc(sg.size), sg); d_v = ctx.dev_malloc(4 * G)
def step(): ctx.accumulator_check_multi_dev(1, 16, G, d_pre, d_sg, d_v)
for _ in range(4): step()
ctx.synchronize()
ctx.prof_enable(-1)
for _ in range(10): step()
p = ctx.prof_read()
print(json.dumps({k: round(v[1] / v[0] * 1000, 1) for k, v in p.items()}), f'us per launch of {G} MSMs; verdicts',
      ctx.dev_download(d_v, 4 * G).view(np.uint32).tolist())
