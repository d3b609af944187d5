# dev tool (round 6): stress of the FORKED device-resident Proof-of-State job (mina_state_job_batch_dev with the three legs of a job on streams of their own): random lane
# counts, random tunings (pieces, LDS reservation, accumulator leg on its own stream / behind / ahead of the hashes, one-stream jobs in between, the 3-lane forms forced
# or not), batches of reduced-size jobs with random tampering of every leg, many calls back to back over the lanes -- every verdict word and flag checked against what
# the tampering implies.  A join that comes too early, a leg on the wrong stream or a workspace shared between legs shows up as a wrong word.  usage: soak_fork.py SECONDS
import copy, os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mina_bridge_amd as m
from oracle import oracle as O
from state_job_helpers import build_jobs, mint_job, state_records
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(time.time()) & 0xffffffff
rng = np.random.Generator(np.random.PCG64(seed))
print("seed", seed)
SMALL = dict(k=7, log2_domain=7, npub=8, n_comms=6, slot=2, n_points=2, acc_k=8)
srs = {c: O.srs_create(c, 1 << 10, threads=8) for c in (0, 1)}
POOL = [mint_job(srs[0], srs[1], 7000 + 10 * i, **SMALL) for i in range(6)]
ctx = m.MinaContext(0)
for f in (0, 1):
    ctx.poseidon_set_params(f, m.poseidon_params.default_params_bytes(f))
ctx.srs_create(0, 1 << 10); ctx.srs_create(1, 1 << 10)
ctx.state_jobs_prepare(SMALL["log2_domain"], SMALL["npub"])


def batch():
    """a batch of 2 .. 6 jobs, each tampered with probability 1/4: (jobs, expected verdict words, expected flags)"""
    B = int(rng.integers(2, 7))
    jobs, chain_ok, ipa_ok, acc_ok = [], [], True, True
    for b in range(B):
        idx = int(rng.integers(0, len(POOL)))
        j = POOL[idx]
        ok = True
        r = rng.random()
        if r < 0.25:
            j = copy.deepcopy(j)
            kind = int(rng.integers(0, 4))
            if kind == 0:                                       # a state whose hash no longer matches: ITS proof fails
                j["states"][int(rng.integers(0, 17))]["body"]["consensus_state"]["total_currency"] ^= 1 << int(rng.integers(0, 30))
                j["records"], j["nfields"] = state_records(j["states"]); ok = False
            elif kind == 1:                                     # an expected hash changed
                j["expected"][int(rng.integers(0, 17))] ^= 1; ok = False
            elif kind == 2:                                     # a public input changed: the folded opening fails
                j["pubs"][int(rng.integers(0, SMALL["npub"]))] ^= 1; ipa_ok = False
            else:                                               # another proof's accumulator commitment: the folded accumulator check fails
                other = POOL[(idx + 1 + int(rng.integers(0, len(POOL) - 1))) % len(POOL)]
                if not (other["acc_sg"] == j["acc_sg"]).all(): j["acc_sg"] = other["acc_sg"].copy(); acc_ok = False
        jobs.append(j); chain_ok.append(ok)
    words = [1 if (c and ipa_ok and acc_ok) else 0 for c in chain_ok]
    return jobs, words, [1 if ipa_ok else 0, 0, 1 if acc_ok else 0, 0]


t0 = time.time(); rounds = calls_total = words_total = folds_total = 0
while time.time() - t0 < budget:
    rounds += 1
    lanes = int(rng.choice([1, 2, 3, 4, 6, 8, 11]))                      # 11: beyond the forked pipelines (<= 8 lanes): one stream per job whatever dev_fork says
    tune = dict(dev_fork=int(rng.choice([0, 1, 1, 1])), dev_piece_waves=int(rng.choice([0, 1, 2, 0xffffffff])), dev_hash_lds_kb=int(rng.choice([0, 33, 41, 0xffffffff])),
                dev_acc_lane=int(rng.choice([0, 1, 2])))
    if rng.random() < 0.5: tune.update(coop16_max=0, coop8_max=0)            # the wave-packed 3-lane forms (the ones launched in pieces / with the LDS reservation)
    ctx.synchronize()
    with m.lib.tuning(**tune):
        ctx.set_pipeline(lanes)
        ncalls = int(rng.integers(lanes, 3 * lanes + 1))
        pending = []
        for _ in range(ncalls):
            jobs, words, flags = batch()
            d, ptrs = ctx.state_jobs_to_device(build_jobs(m, jobs, SMALL["k"], SMALL["log2_domain"], SMALL["slot"], SMALL["acc_k"], rand_base=int(rng.integers(2, 1 << 30)),
                                                          sg_rand_base=int(rng.integers(2, 1 << 30)), rho_seed=int(rng.integers(1, 1 << 30))))
            o = ctx.dev_upload(ctx.dev_malloc(4 * len(jobs) + 16), np.full(len(jobs) + 4, 9, np.uint32).view(np.uint8))
            pending.append((d, ptrs, o, len(jobs), words, flags))
        for d, ptrs, o, B, words, flags in pending:            # nothing waits between the calls
            ctx.state_job_batch_dev(d, o, o + 4 * B)
        ctx.synchronize()
        for i, (d, ptrs, o, B, words, flags) in enumerate(pending):
            w = ctx.dev_download(o, 4 * B + 16).view(np.uint32).tolist()
            assert (w[:B], w[B:]) == (words, flags), ("fork soak", seed, rounds, lanes, tune, i, w, words, flags)
            for p in ptrs + [o]: ctx.dev_free(p)
            words_total += B
        calls_total += ncalls
    # the exchange variant's shard job (mina_state_job_fold_dev: both folds exported instead of checked) forked against one-stream, same job, same opening randomisers: the
    # per-proof verdict words, the flags, the folded Pallas scalar vector and the variable-base partial point must be the same bytes (the accumulator side carries a fresh
    # CSPRNG scalar per call and is not compared)
    if rounds % 4 == 0:
        jobs, words, flags = batch()
        chain_only = None
        outs = []
        for fork in (0, 1):
            ctx.synchronize()
            with m.lib.tuning(dev_fork=fork, **({"coop16_max": 0, "coop8_max": 0} if rng.random() < 0.5 else {})):
                ctx.set_pipeline(1)
                d, ptrs = ctx.state_jobs_to_device(build_jobs(m, jobs, SMALL["k"], SMALL["log2_domain"], SMALL["slot"], SMALL["acc_k"], rand_base=12345 + rounds, sg_rand_base=777 + rounds, rho_seed=rounds))
                B = len(jobs)
                o = ctx.dev_upload(ctx.dev_malloc(4 * B + 16), np.full(B + 4, 9, np.uint32).view(np.uint8))
                n_ipa, n_acc = (1 << SMALL["k"]) * 32, (1 << SMALL["acc_k"]) * 32
                bufs = [ctx.dev_upload(ctx.dev_malloc(n), np.zeros(n, np.uint8)) for n in (n_ipa, 68, n_acc, 68)]
                ctx.state_job_fold_dev(d, o, o + 4 * B, *bufs)
                ctx.synchronize()
                w = ctx.dev_download(o, 4 * B + 16).view(np.uint32).tolist()
                outs.append((w, ctx.dev_download(bufs[0], n_ipa).tobytes(), ctx.dev_download(bufs[1], 68).tobytes()))
                for p in ptrs + [o] + bufs: ctx.dev_free(p)
        assert outs[0] == outs[1], ("fold soak", seed, rounds, outs[0][0], outs[1][0])
        folds_total += 1
ctx.synchronize(); ctx.set_pipeline(1)
print(f"fork soak ok: {rounds} rounds, {calls_total} jobs, {words_total} verdict words, {folds_total} forked-vs-one-stream fold exports in {time.time() - t0:.0f}s")
