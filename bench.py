#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X-native Kimchi/Pickles IPA hot path (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic state proofs: for each proof the 2^16-base
Vesta IPA accumulator check of BASELINE config C2 -- 16 128-bit prechallenges (already in HBM) ->
ScalarChallenge::to_field -> b_poly_coefficients (K2) -> 2^16-point MSM over the Vesta SRS (K1) -> compare
with the proof's sg -- through the C-ABI, verdicts left in HBM.  Default: the `--group` (8) proofs of a step are
verified INDEPENDENTLY (no random folding, one verdict each) by one kernel pipeline
(`mina_accumulator_check_multi_dev`: every MSM is computed in full; the group only shares the ~14 dependent
dispatches).  `--batch B` instead folds B proofs into one MSM with random weights (`mina_accumulator_check_dev`).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--group G] [--batch B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: proof-level sharding, no data-path collective (SURVEY.md 8e variant 1): every rank verifies its
own proofs against its replica of the SRS tables; weak scaling.  One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# ROCm exposes 4 hardware queues per process by default; the pipeline lanes of the context need one each
# to overlap (must be set before the HIP runtime initialises).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

CURVE_VESTA = 1
FIELD_FP = 0
K_ROUNDS = 16
N_BASES = 1 << K_ROUNDS
MSM_ALGORITHMIC_BYTES = N_BASES * (64 + 32) + 96      # 6 291 552 B for n = 2^16
HBM_PEAK_GBPS = 8000.0                                # MI355X_MICROARCH.md: 8 TB/s spec
PS_ACCUMULATE_BIT = 1 << 3      # ProfStage::PS_ACCUMULATE in csrc/ctx.h
# HBM-side bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, KiB * 1024;
# profiles/r01j_rocprof.md section 2).  14x the algorithmic bytes by design: the fixed-base window tables gather 16
# precomputed 64-B points per base and write 128-B XYZZ partials.
ACCUMULATE_TRAFFIC_BYTES = 80_470_000
# VALU side of the roofline (the path is integer-multiply bound): one Montgomery product is 88
# v_mad_u64_u32 (8 cycles per wave64 instruction, profiles/r01_microbench_valu.jsonl) -> issue floor 704 cycles.
MODMUL_ISSUE_FLOOR_CYCLES = 88 * 8
# entries E = one per non-zero signed 16-bit digit; in the throughput form of the accumulate kernel (one lane per bucket,
# used from 4 MSMs per launch or with pipelined lanes) the first entry of each of the 32768 buckets is a copy, not an add
MIXED_ADDS_PER_MSM = 16 * N_BASES * (1 - 2 ** -16) - 32768
MODMUL_PER_MIXED_ADD = 10                                # XYZZ madd-2008-s: 8M + 2S
CHIP_SIMDS, CLOCK_HZ = 1024, 2.4e9


def make_instances(ctx, count: int, seed: int):
    """`count` synthetic accumulator-check instances made consistent with the GPU path itself
    (sg = MSM(g, b_poly_coefficients(chals)) through the library; parity of that path vs the CPU oracle is
    what tests/ establish).  Returns (prechallenges[count,16,16], sg[count,64])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pre = rng.integers(0, 256, size=(count, K_ROUNDS, 16), dtype=np.uint8)
    sgs = np.empty((count, 64), np.uint8)
    for i in range(count):
        chals = ctx.challenge_to_field(FIELD_FP, pre[i])
        s = ctx.b_poly_coefficients(FIELD_FP, chals)
        sgs[i] = ctx.msm_srs(CURVE_VESTA, s)
    return pre, sgs


def cpu_baseline(pre_one: np.ndarray, sg_one: np.ndarray, budget_s: float = 12.0):
    """The CPU restatement (oracle/, kind="port") of the same step, on the host cores of this box."""
    from oracle import oracle as O
    O.lib()
    cores = min(os.cpu_count() or 1, 20)          # ark Pippenger has 20 windows at n = 2^16 -> 20 useful threads
    g, _ = O.srs_create(CURVE_VESTA, N_BASES, threads=os.cpu_count() or 1)
    _, endo_r = O.endo(CURVE_VESTA)

    def one():
        chals = np.stack([O.challenge_to_field(FIELD_FP, pre_one[i].copy(), endo_r) for i in range(K_ROUNDS)])
        s = O.b_poly_coefficients(FIELD_FP, chals)
        return O.msm_pippenger(CURVE_VESTA, g, s, threads=cores)

    ok = bool((one() == sg_one).all())
    t0 = time.perf_counter()
    reps = 0
    while True:
        one()
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 200:
            break
    return {"value": reps / el, "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": f"{reps} single-proof accumulator checks (to_field + b_poly_coefficients + ark-style Pippenger c=13, "
                      f"one thread per window) in {el:.1f}s; matches GPU sg: {ok}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--group", type=int, default=8, help="independent (un-folded) proofs verified per step by one kernel pipeline")
    ap.add_argument("--no-probes", action="store_true", help="skip the single-proof latency and folded-batch probes (profiling runs)")
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1, help="proofs folded into one MSM per step")
    ap.add_argument("--pipeline", type=int, default=16, help="internal stream lanes over which consecutive steps are issued")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import mina_bridge_amd as m

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} without a torch.distributed.run launch (WORLD_SIZE=1): measuring 1 GPU", file=sys.stderr)
        args.gpus = world                      # the number of ranks actually running is what is reported
    dist_on = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # MINA_BENCH_SHARE_GPU=1 (test hook): every rank uses GPU 0 and the ranks rendezvous over gloo -- lets the N > 1 code path
    # (barriers, MAX over ranks, aggregate value) be exercised on a 1-GPU box; never set by the driver
    share_gpu = os.environ.get("MINA_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ctx = m.MinaContext(local_rank)
    ctx.srs_create(CURVE_VESTA, N_BASES)                       # SRS regenerated on the GPU (K4) + window tables
    ctx.set_pipeline(args.pipeline)
    B = args.batch
    G = args.group if B == 1 else 1                            # un-folded proofs per step; folding and grouping are alternatives
    if not 1 <= G <= 64:
        raise SystemExit("--group must be in 1..64")
    pre, sgs = make_instances(ctx, max(B, min(G, 4)), seed=0x6D696E61 + rank)
    if G > 1:                                                  # G proofs per step from 4 distinct instances
        pre = np.stack([pre[i % len(pre)] for i in range(G)]); sgs = np.stack([sgs[i % len(sgs)] for i in range(G)])
    dev = torch.device("cuda", local_rank)
    d_pre = torch.from_numpy(pre.reshape(-1)).to(dev)
    d_sg = torch.from_numpy(sgs.reshape(-1)).to(dev)
    rho = np.random.Generator(np.random.PCG64(99 + rank)).integers(0, 256, size=(B, 32), dtype=np.uint8)
    rho[:, 31] &= 0x3F
    d_rho = torch.from_numpy(rho.reshape(-1)).to(dev)
    d_verdict = torch.zeros(G, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step():
        if B == 1:
            ctx.accumulator_check_multi_dev(CURVE_VESTA, K_ROUNDS, G, d_pre.data_ptr(), d_sg.data_ptr(), d_verdict.data_ptr())
        else:
            ctx.accumulator_check_dev(CURVE_VESTA, K_ROUNDS, B, d_pre.data_ptr(), d_sg.data_ptr(), d_rho.data_ptr(), d_verdict.data_ptr())

    def verdicts_ok():
        return d_verdict.cpu().numpy().tolist() == [1] * G

    for _ in range(max(args.warmup, args.pipeline)):          # every lane allocates its workspace during warm-up
        step()
    ctx.synchronize()
    assert verdicts_ok(), "warm-up verdicts must be ACCEPT"

    def barrier():
        if dist_on:
            dist.barrier()

    ctx.prof_enable(PS_ACCUMULATE_BIT)                         # HIP events around the dominant kernel, on the ctx stream
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize(); torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.prof_read()
    ctx.prof_enable(0)
    assert verdicts_ok(), "timed-region verdicts must be ACCEPT"
    # the same kernel with nothing else on the GPU: one lane, HIP events on that lane's stream
    ctx.set_pipeline(1)
    for _ in range(4):
        step()
    ctx.synchronize()
    ctx.prof_enable(PS_ACCUMULATE_BIT)
    for _ in range(16):
        step()
    prof_iso = ctx.prof_read()
    ctx.prof_enable(0)
    # single-stream latency of ONE proof alone (one lane, nothing overlapped, no group)
    latency_ms = None
    if not args.no_probes:
        d_v1 = torch.zeros(1, dtype=torch.int32, device=dev)
        def step1():
            ctx.accumulator_check_dev(CURVE_VESTA, K_ROUNDS, 1, d_pre.data_ptr(), d_sg.data_ptr(), 0, d_v1.data_ptr())
        for _ in range(4):
            step1()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(64):
            step1()
        ctx.synchronize()
        latency_ms = (time.perf_counter() - t1) / 64 * 1e3
        assert int(d_v1.item()) == 1
    # the design's batch mode: B proofs folded into ONE MSM per step (kimchi batch_verify's shape), same entry point
    extra = {}
    if B == 1 and rank == 0 and not args.no_probes:
        BB = 256
        pre_b, sg_b = make_instances(ctx, 4, seed=77)
        pre_b = np.tile(pre_b, (BB // 4, 1, 1)); sg_b = np.tile(sg_b, (BB // 4, 1))
        rho_b = np.random.Generator(np.random.PCG64(5)).integers(0, 256, size=(BB, 32), dtype=np.uint8); rho_b[:, 31] &= 0x3F
        dpb, dsb, drb = (torch.from_numpy(x.reshape(-1)).to(dev) for x in (pre_b, sg_b, rho_b))
        dvb = torch.zeros(1, dtype=torch.int32, device=dev)
        ctx.set_pipeline(args.pipeline)
        def step_b():
            ctx.accumulator_check_dev(CURVE_VESTA, K_ROUNDS, BB, dpb.data_ptr(), dsb.data_ptr(), drb.data_ptr(), dvb.data_ptr())
        for _ in range(args.pipeline):
            step_b()
        ctx.synchronize(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        nb_steps = 128
        for _ in range(nb_steps):
            step_b()
        ctx.synchronize()
        el_b = time.perf_counter() - t2
        assert int(dvb.item()) == 1
        extra = {"folded_batch_mode": {"proofs_per_step": BB, "value": BB * nb_steps / el_b, "unit": "proofs/s", "ms_per_step": el_b / nb_steps * 1e3,
                                       "note": "same C-ABI entry, 256 proofs folded with random rho into one 2^16 MSM + one 256-point variable-base MSM"}}

    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        launches_o, total_ms_o = prof.get("msm_accumulate", [0, 0.0])
        overlapped_us = (total_ms_o / launches_o) * 1e3 if launches_o else None
        launches, total_ms = prof_iso.get("msm_accumulate", [0, 0.0])
        kern_s = (total_ms / launches) * 1e-3 if launches else float("nan")
        msms = G                                               # MSMs one msm_accumulate launch processes
        achieved = msms * MSM_ALGORITHMIC_BYTES / kern_s / 1e9 if launches else None
        out = {
            "metric": "Mina state proofs verified/sec (batch)",
            "value": args.gpus * args.steps * B * G / elapsed,
            "unit": "proofs/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "single_stream_latency_ms": latency_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32x8-montgomery (255-bit prime field, integer)", "data": "synthetic",
            "config": {"workload": "C2: per-proof 2^16-base Vesta IPA accumulator check (to_field + b_poly_coefficients + MSM over "
                                   "vesta.srs + compare), bit-exact vs CPU oracle", "curve": "vesta", "n_bases": N_BASES,
                       "proofs_per_step": B * G, "mode": (f"{G} independent checks per kernel pipeline, no folding" if B == 1 else f"{B} proofs folded into one MSM"),
                       "pipeline_lanes": args.pipeline, "sharding": f"proof-level, {args.gpus} rank(s), no collective"},
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate_bucket_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": msms * ACCUMULATE_TRAFFIC_BYTES,
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/r01j_rocprof.md",
                         "traffic_GBps": msms * ACCUMULATE_TRAFFIC_BYTES / kern_s / 1e9 if launches else None,
                         "algorithmic_bytes_per_launch": msms * MSM_ALGORITHMIC_BYTES, "msms_per_launch": msms, "avg_launch_us": kern_s * 1e6,
                         "avg_launch_us_in_timed_region": overlapped_us,
                         "note": "avg_launch_us: HIP events on the lane stream, 16 launches with nothing else on the GPU, right after the "
                                 "timed region; in the timed region the launches of 16 lanes overlap and time-share the CUs (second figure). "
                                 "Integer-VALU-bound path (SURVEY.md 8d): HBM fraction reported as the metric demands, see roofline_valu"},
        }
        if launches:
            peak = CHIP_SIMDS * 64 * CLOCK_HZ / MODMUL_ISSUE_FLOOR_CYCLES          # modmul/s if only the 88 mads issued
            got = msms * MIXED_ADDS_PER_MSM * MODMUL_PER_MIXED_ADD / kern_s
            out["roofline_valu"] = {"bound": "int32 multiply issue (v_mad_u64_u32)", "achieved": got / 1e9, "peak": peak / 1e9,
                                    "unit": "G modmul/s", "frac": got / peak,
                                    "note": "same isolated launches as roofline"}
        out.update(extra)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pre[0], sgs[0])
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
