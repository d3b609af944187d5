"""One rank of tests/test_sharded_state_job.py: a shard of full-size Proof-of-State jobs through `ShardedStateJob` (SURVEY.md 8e.2 for the whole job).
Launched with RANK / WORLD_SIZE / MASTER_* in the environment; every rank uses GPU 0 and the ranks rendezvous over gloo (a 1-GPU box), or its own GPU over
RCCL when the box has enough of them.  argv: B_per_rank  scenario  (ok | ok_delayed | bad_opening_on_last_rank | bad_accumulator_on_rank0 | opposite_z2_on_first_proofs).  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
import torch
import torch.distributed as dist

import bench
import mina_bridge_amd as m
from mina_bridge_amd.sharded import DeviceBackend, ShardedStateJob

B, scenario = int(sys.argv[1]), sys.argv[2]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
real = torch.cuda.device_count() >= world
dist.init_process_group("nccl" if real else "gloo")
dev_ix = rank if real else 0
torch.cuda.set_device(dev_ix)
dev = torch.device("cuda", dev_ix)
ctx = m.MinaContext(dev_ix)
for f in (0, 1):
    ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(1, 1 << 16); ctx.srs_create(0, 1 << 16)
(hj, keep), kp, _, _ = bench.build_full_job(ctx, m, B, seed=900 + rank)
by_addr = {a.ctypes.data: a for a in keep if isinstance(a, np.ndarray)}
bad_at = None
if scenario == "bad_opening_on_last_rank" and rank == world - 1:
    bad_at = B // 2; by_addr[hj.z1].view(np.uint8).reshape(B, 32)[bad_at, 0] ^= 1
if scenario == "bad_accumulator_on_rank0" and rank == 0:
    bad_at = 1; by_addr[hj.acc_prechallenges].view(np.uint8).reshape(B, 16, 16)[bad_at, 3, 0] ^= 1
if scenario == "opposite_z2_on_first_proofs" and rank < 2:
    # z2 + t on the first proof of rank 0's shard, z2 - t on the first proof of rank 1's: discrepancies -tH and +tH.  They cancelled in the
    # exchanged total while every shard's first proof carried coefficient 1 (rho_b = rand_base^b per shard); with rho_b = rand_base^(b+1) they cannot.
    Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001          # Pallas scalar field (z2 lives there)
    bad_at = 0; z2 = by_addr[hj.z2].view(np.uint8).reshape(B, 32)
    v = (int.from_bytes(z2[0].tobytes(), "little") + (0x1234567 if rank == 0 else -0x1234567)) % Q
    z2[0] = np.frombuffer(v.to_bytes(32, "little"), np.uint8)
if scenario == "opposite_acc_sg_fixed_rho" and rank < 2:
    # the accumulator fold of the exchange variant trusted the caller's acc_rho.  A caller following upstream's convention (rho_0 = 1) on every shard,
    # sg + T first in shard A and sg - T first in shard B: the discrepancies -T and +T cancelled in the exchanged total.  The library now multiplies a shard's
    # acc_rho by a scalar of its own draw (api_ipa.hip mb_accumulator_check_dev): they cannot.
    from oracle import oracle as O
    FQ = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001          # Vesta's base field
    T = ctx.srs_get_g(1, 5, 1)[0].copy()
    if rank == 1:
        y = (FQ - int.from_bytes(T[32:].tobytes(), "little")) % FQ; T[32:] = np.frombuffer(y.to_bytes(32, "little"), np.uint8)
    sg = by_addr[hj.acc_sg].view(np.uint8).reshape(B, 64)
    sg[0] = O.point_add(1, sg[0].copy(), T)
    rho = by_addr[hj.acc_rho].view(np.uint8).reshape(B, 32)
    rho[0] = 0; rho[0, 0] = 1
    bad_at = 0
dj, dk, tensors = bench.device_jobs(m, hj, keep, kp, dev)
ctx.state_jobs_prepare(bench.LOG2_DOMAIN, bench.NPUB)
be = DeviceBackend(ctx, dev)
job = ShardedStateJob(be, k=bench.WRAP_K, acc_k=bench.ACC_K)
if scenario == "ok_delayed":
    job.verify(dj, B)                                            # warm: workspaces, RCCL's first-call set-up
    job.coll_streams.clear(); be.host_syncs = 0; job.host_reads = 0
    job.test_delay_cycles = 400_000_000                          # ~0.2 s of spin on the ordering stream ahead of the shard's kernels
import time
t_call = time.perf_counter()
verdicts, ok = job.verify(dj, B)
call_s = time.perf_counter() - t_call
host_syncs, host_reads = be.host_syncs, job.host_reads          # of the exchange step alone (the fallback of a failed batch runs the ordinary job and synchronises)
# the ordinary single-GPU job on the same shard, for comparison
plain = torch.zeros(B + 4, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
ctx.state_job_batch_dev(dj, plain.data_ptr(), plain.data_ptr() + 4 * B); ctx.synchronize(); torch.cuda.synchronize()
print(json.dumps({"rank": rank, "world": world, "backend": dist.get_backend(), "batch_ok": ok, "verdicts": verdicts.cpu().numpy().tolist(),
                  "plain": plain[:B].cpu().numpy().tolist(), "plain_flags": plain[B:].cpu().numpy().tolist(), "bad_at": bad_at, "detail": job.last, "host_syncs": host_syncs, "host_reads": host_reads,
                  "coll_streams": job.coll_streams, "ctx_stream": int(ctx.stream or 0), "call_s": call_s}), flush=True)
dist.barrier()
dist.destroy_process_group()
ctx.close()
