"""Multi-GPU sharding logic without a multi-GPU box:
  * world_size-2 `gloo` on CPU: both strategies give the single-process verdicts, with an oracle-backed double of the device
    backend (allowed: tests/ may use oracle/ as the checker; the product path on the GPU box uses DeviceBackend -> MinaContext);
    includes the C5 shape -- 4096 proofs (at a small k), ragged shards, one bad proof per shard;
  * -m gpu: the real DeviceBackend under a 1-rank RCCL group on the GPU (all-to-all / all-gather on device tensors)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, CURVE = 6, 1


class OracleBackend:
    """DeviceBackend's interface on CPU tensors, computed by oracle/ (test double)"""
    RECORD = 68

    def __init__(self):
        import torch
        from oracle import oracle as O
        self.O, self.torch = O, torch
        self.g, self.h = O.srs_create(CURVE, 1 << K, threads=2)
        _, self.endo_r = O.endo(CURVE)

    def sync(self):
        pass

    def _t(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8).reshape(-1).copy())

    def _rec(self, pt64):
        r = np.zeros(68, np.uint8); r[:64] = pt64
        if not pt64.any():
            r[64] = 1
        return self._t(r)

    def _chals(self, field, pre):
        return np.stack([self.O.challenge_to_field(field, p.copy(), self.endo_r) for p in pre.numpy().reshape(-1, 16)])

    def accumulator_verdicts(self, curve, k, pre, sg, rho):
        sgn = sg.numpy().reshape(-1, 64)
        out = []
        for b in range(sgn.shape[0]):
            chals = self._chals(0, pre[b * k * 16:(b + 1) * k * 16])
            s = self.O.b_poly_coefficients(0, chals)
            out.append(int((self.O.msm_pippenger(curve, self.g, s, threads=1) == sgn[b]).all()))
        return self.torch.tensor(out, dtype=self.torch.uint8)

    def fold_scalars(self, field, k, pre, rho):
        from oracle import pasta_ref as R
        m = R.P if field == 0 else R.Q
        w = rho.numpy().reshape(-1, 32)
        acc = [0] * (1 << k)
        for b in range(w.shape[0]):
            s = self.O.b_poly_coefficients(field, self._chals(field, pre[b * k * 16:(b + 1) * k * 16]))
            wb = self.O.le_to_int(w[b])
            for j in range(1 << k):
                acc[j] = (acc[j] + wb * self.O.le_to_int(s[j])) % m
        return self._t(self.O.ints_to_le(acc))

    def sum_rows(self, field, rows, m, stacked):
        from oracle import pasta_ref as R
        mod = R.P if field == 0 else R.Q
        a = stacked.numpy().reshape(rows, m, 32)
        return self._t(self.O.ints_to_le([sum(self.O.le_to_int(a[r, j]) for r in range(rows)) % mod for j in range(m)]))

    def msm_srs_range(self, curve, first, n, scalars):
        return self._rec(self.O.msm_pippenger(curve, self.g[first:first + n], scalars.numpy().reshape(-1, 32), threads=1))

    def msm(self, curve, n, bases, scalars):
        if n == 0:
            return self._rec(np.zeros(64, np.uint8))
        return self._rec(self.O.msm_naive(curve, bases.numpy().reshape(-1, 64), scalars.numpy().reshape(-1, 32)))

    def points_sum(self, curve, n, records):
        acc = np.zeros(64, np.uint8)
        for r in records.numpy().reshape(n, 68):
            if not r[64]:
                acc = self.O.point_add(curve, acc, r[:64].copy())
        return self._rec(acc)

    def records_equal(self, a, b):
        return bool((a == b).all())


def make_batch(O, g, B, seed=2024, tamper=()):
    rng = np.random.Generator(np.random.PCG64(seed))
    pre = rng.integers(0, 256, size=(B, K, 16), dtype=np.uint8)
    _, endo_r = O.endo(CURVE)
    sg = np.empty((B, 64), np.uint8)
    cache = {}
    for b in range(B):
        if B > 64:                                           # C5 shape: 4096 proofs from 64 distinct instances
            pre[b] = pre[b % 64]
            if b % 64 in cache and b >= 64:
                sg[b] = cache[b % 64]; continue
        chals = np.stack([O.challenge_to_field(0, pre[b, i].copy(), endo_r) for i in range(K)])
        sg[b] = O.msm_pippenger(CURVE, g, O.b_poly_coefficients(0, chals), threads=1)
        cache[b % 64] = sg[b]
    rho = rng.integers(0, 256, size=(B, 32), dtype=np.uint8); rho[:, 31] &= 0x3F
    for t in tamper:
        sg[t] = g[1]
    return pre, sg, rho


def _shard(t, pre, sg, rho, rank, world):
    from mina_bridge_amd.sharded import shard_range
    lo, hi = shard_range(sg.shape[0], rank, world)
    f = lambda a: t.from_numpy(np.ascontiguousarray(a[lo:hi]).reshape(-1).copy())
    return f(pre), f(sg), f(rho)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from mina_bridge_amd.sharded import ShardedAccumulatorCheck
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    be = OracleBackend()
    sh = ShardedAccumulatorCheck(be, CURVE, K)
    res = {}
    for name, B, tamper in (("ok", 5, ()), ("bad", 5, (3,)), ("c5", 4096, (100, 4000))):          # 5 over 2 ranks: ragged shards (2 + 3)
        pre, sg, rho = make_batch(be.O, be.g, B, tamper=tamper)
        p, s, r = _shard(torch, pre, sg, rho, rank, world)
        res[name + "_proof"] = sh.verify_proof_level(p, s, r, B).numpy().tolist()
        if B <= 64:
            res[name + "_sliced"] = sh.verify_base_sliced(p, s, r)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert got[r]["ok_proof"] == [1] * 5 and got[r]["ok_sliced"] is True
        assert got[r]["bad_proof"] == [1, 1, 1, 0, 1] and got[r]["bad_sliced"] is False
        c5 = got[r]["c5_proof"]                               # 4096 proofs, one bad proof in each rank's shard
        assert len(c5) == 4096 and sum(c5) == 4094 and c5[100] == 0 and c5[4000] == 0


def test_shard_range_covers_everything():
    from mina_bridge_amd.sharded import shard_range
    for n in (0, 1, 5, 4096, 65536):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))


@pytest.mark.gpu
def test_device_backend_under_rccl_single_rank(ctx_srs, oracle, srs_oracle):
    """the product's backend on the GPU: device tensors through a 1-rank RCCL group (all-to-all, all-gather), both strategies,
    k = 16 on the real Vesta SRS, C5-sized proof-level batch with bad proofs"""
    import torch
    import torch.distributed as dist
    from mina_bridge_amd.sharded import DeviceBackend, ShardedAccumulatorCheck
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dev = torch.device("cuda", 0)
        be = DeviceBackend(ctx_srs, dev)
        g, _ = srs_oracle[1]
        k = 16
        sh = ShardedAccumulatorCheck(be, 1, k)
        rng = np.random.Generator(np.random.PCG64(5))
        nd = 8
        pre = rng.integers(0, 256, size=(nd, k, 16), dtype=np.uint8)
        sg = np.stack([ctx_srs.msm_srs(1, ctx_srs.b_poly_coefficients(0, ctx_srs.challenge_to_field(0, pre[i]))) for i in range(nd)])
        _, endo_r = oracle.endo(1)
        chals0 = np.stack([oracle.challenge_to_field(0, pre[0, i].copy(), endo_r) for i in range(k)])
        assert (sg[0] == oracle.msm_pippenger(1, g, oracle.b_poly_coefficients(0, chals0), threads=8)).all()      # instance 0 pinned to the oracle
        for B, tamper in ((8, ()), (8, (5,)), (4096, (7, 3000))):
            idx = np.arange(B) % nd
            P, S = pre[idx].copy(), sg[idx].copy()
            for t in tamper:
                S[t] = g[3]
            rho = rng.integers(0, 256, size=(B, 32), dtype=np.uint8); rho[:, 31] &= 0x3F
            tp, ts, tr = (torch.from_numpy(a.reshape(-1)).to(dev) for a in (P, S, rho))
            v = sh.verify_proof_level(tp, ts, tr, B).cpu().numpy()
            assert v.sum() == B - len(tamper) and all(v[t] == 0 for t in tamper)
            if B == 8:
                assert sh.verify_base_sliced(tp, ts, tr) is (len(tamper) == 0)
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------- SURVEY.md 8e.2 for the WHOLE Proof-of-State job
class OracleJobBackend:
    """the backend interface of `ShardedStateJob` on CPU tensors, computed by oracle/ at FULL size (2^15 wrap domain, 2^16 accumulator): the native composite's
    fold with the fixed-base MSMs left out (composite_oracle.c oc_fold_export) stands in for mina_state_job_fold_dev, ark-style Pippenger for the MSM kernels"""
    RECORD = 68
    dev = "cpu"

    def __init__(self, threads=2):
        import json
        import torch
        import mina_bridge_amd.poseidon_params as PP
        from oracle import composite as C, oracle as O
        self.O, self.C, self.torch, self.threads = O, C, torch, threads
        self.srs = {c: O.srs_create(c, 1 << 16, threads=threads) for c in (0, 1)}
        self.fx = json.load(open(os.path.join(ROOT, "tests", "golden", "statement_k15_encoded.json")))
        C.setup(self.srs[0], self.srs[1], PP.default_params_bytes(0), PP.default_params_bytes(1), self.fx["wrap_index"], self.fx["step_index"], threads=threads)

    def sync(self): pass

    def _t(self, a): return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8).reshape(-1).copy())

    def _rec(self, pt64):
        r = np.zeros(68, np.uint8); r[:64] = pt64
        if not np.asarray(pt64).any(): r[64] = 1
        return self._t(r)

    def state_job_fold(self, job, batch, k, acc_k):
        assert (k, acc_k) == (15, 16) and len(job) == batch
        ipa_s, ipa_p, acc_s, acc_p, ok = self.C.fold_export(job, self.threads)
        return self._t(ok), self._t(np.array([1, 0, 1, 0], np.uint8)), self._t(ipa_s), self._rec(ipa_p), self._t(acc_s), self._rec(acc_p)

    def state_job_plain(self, job, batch):
        return self._t(self.C.verify_folded(job, self.threads)[1])

    def sum_rows(self, field, rows, m, stacked):
        from oracle import pasta_ref as R
        mod = R.P if field == 0 else R.Q
        a = stacked.numpy().reshape(rows, m, 32)
        return self._t(self.O.ints_to_le([sum(self.O.le_to_int(a[r, j]) for r in range(rows)) % mod for j in range(m)]))

    def msm_srs_range(self, curve, first, n, scalars):
        return self._rec(self.O.msm_pippenger(curve, self.srs[curve][0][first:first + n], scalars.numpy().reshape(-1, 32), threads=self.threads))

    def points_sum(self, curve, n, records):
        acc = np.zeros(64, np.uint8)
        for r in records.numpy().reshape(n, 68):
            if not r[64]: acc = self.O.point_add(curve, acc, r[:64].copy())
        return self._rec(acc)

    def records_equal(self, a, b): return bool((a == b).all())

    def records_equal_word(self, a, b): return self.torch.tensor([int(bool((a == b).all()))], dtype=self.torch.int32)


def _state_job_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import copy
    import random
    import torch.distributed as dist
    from ipa_helpers import poseidon_pp
    from kimchi_helpers import load_statement_fixture, make_chain
    from mina_bridge_amd.sharded import ShardedStateJob
    from oracle import mina_state_ref as S
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    be = OracleJobBackend()
    items, _ = load_statement_fixture()

    def proof(i, tamper=False, z2_delta=0):
        it = items[i % 4]
        states, hashes = make_chain(random.Random(it["chain_seed"]), poseidon_pp(0))
        recs = np.zeros((17, 64, 32), np.uint8); nf = np.zeros(17, np.uint32)
        for s_, st in enumerate(states):
            f = [st["previous_state_hash"]] + S.body_to_input(st["body"]).to_fields()
            nf[s_] = len(f) - 1
            recs[s_, : len(f)] = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in f), np.uint8).reshape(len(f), 32)
        exp = np.frombuffer(b"".join(int(h).to_bytes(32, "little") for h in hashes), np.uint8).reshape(17, 32).copy()
        enc = copy.deepcopy(be.fx["proofs"][i % 4])
        if tamper:
            b = bytearray(bytes.fromhex(enc["opening"]["z1"])); b[0] ^= 1; enc["opening"]["z1"] = bytes(b).hex()
        if z2_delta:                                   # z2 lives in the scalar field of Pallas (Fq): the opening's discrepancy becomes -z2_delta * H
            from oracle import pasta_ref as R
            z2 = (int.from_bytes(bytes.fromhex(enc["opening"]["z2"]), "little") + z2_delta) % R.Q
            enc["opening"]["z2"] = z2.to_bytes(32, "little").hex()
        return be.C.make_proof(enc, recs.reshape(17, -1), nf, exp)
    job = ShardedStateJob(be, k=15, acc_k=16)
    res = {}
    shard = [proof(3 * rank + i) for i in range(3)]
    v, ok = job.verify(shard, 3); res["ok"] = (v.numpy().tolist(), ok, dict(job.last))
    shard_bad = [proof(3 * rank + i, tamper=(rank == 1 and i == 2)) for i in range(3)]
    v, ok = job.verify(shard_bad, 3); res["bad"] = (v.numpy().tolist(), ok, dict(job.last))
    # the FIRST proof of rank 0's shard opens with z2 + t, the first proof of rank 1's with z2 - t.  While every shard's first proof carried
    # coefficient 1 (rho_b = rand_base^b per shard) the two discrepancies -tH and +tH cancelled in the exchanged total and both invalid proofs were accepted.
    t = 0x1234567
    shard_pm = [proof(3 * rank + i, z2_delta=((t if rank == 0 else -t) if i == 0 else 0)) for i in range(3)]
    v, ok = job.verify(shard_pm, 3); res["plus_minus"] = (v.numpy().tolist(), ok, dict(job.last))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_variant_of_the_whole_state_job_on_the_cpu_double():
    """`ShardedStateJob` (SURVEY.md 8e.2 for the whole Proof-of-State job: one all-to-all of folded scalar vectors, base-sliced MSMs, one all-gather of partial
    points) under a 2-rank gloo group with the oracle-backed double at FULL size -- the distributed logic of the N > 1 path on CPU.  An accepting batch passes on
    both ranks; ONE bad opening on rank 1 fails the exchanged check on both ranks and the fallback localises it to rank 1's shard.  (The GPU side of the same
    class: tests/test_sharded_state_job.py.)"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_state_job_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    got = dict(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        v, ok, detail = got[r]["ok"]
        assert ok is True and v == [1, 1, 1], (r, detail)
        v, ok, detail = got[r]["bad"]
        assert ok is False and detail["opening_fold_ok"] is False and detail["accumulator_fold_ok"] is True, (r, detail)
        assert v == ([1, 1, 1] if r == 0 else [0, 0, 0]), (r, v)
        v, ok, detail = got[r]["plus_minus"]                                   # opposite discrepancies on the two shards' first proofs do NOT cancel
        assert ok is False and detail["opening_fold_ok"] is False, (r, detail)
        assert v == [0, 0, 0], (r, v)
