"""world_size-2 gloo test of the multi-GPU sharding logic on CPU: both strategies must give the
single-process verdicts.  No GPU here, so the compute backend plugged into ShardedAccumulatorCheck is the CPU oracle
(allowed: tests/ may use oracle/ as the checker; the product path on the GPU box uses MinaContext)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, CURVE, B = 6, 1, 5


class OracleBackend:
    """the subset of MinaContext's methods sharded.py calls, computed by oracle/ (test double)"""

    def __init__(self):
        from oracle import oracle as O
        self.O = O
        self.g, self.h = O.srs_create(CURVE, 1 << K, threads=2)
        _, self.endo_r = O.endo(CURVE)

    def challenge_to_field(self, field, pre):
        return np.stack([self.O.challenge_to_field(field, p.copy(), self.endo_r) for p in pre.reshape(-1, 16)])

    def b_poly_fold(self, field, k, chals, weights):
        from oracle import pasta_ref as R
        m = R.P if field == 0 else R.Q
        chals = chals.reshape(-1, k, 32); w = weights.reshape(-1, 32)
        acc = [0] * (1 << k)
        for b in range(chals.shape[0]):
            s = self.O.b_poly_coefficients(field, chals[b])
            wb = self.O.le_to_int(w[b])
            for j in range(1 << k):
                acc[j] = (acc[j] + wb * self.O.le_to_int(s[j])) % m
        return self.O.ints_to_le(acc)

    def msm_srs_range(self, curve, first, scalars):
        n = scalars.size // 32
        return self.O.msm_pippenger(curve, self.g[first:first + n], scalars, threads=1)

    def msm(self, curve, bases, scalars):
        return self.O.msm_naive(curve, bases, scalars)

    def accumulator_check_batch(self, curve, k, pre, sg, rho=None):
        pre = pre.reshape(-1, k, 16); sg = sg.reshape(-1, 64)
        out = []
        for b in range(sg.shape[0]):
            chals = self.challenge_to_field(1 if curve == 0 else 0, pre[b])
            s = self.O.b_poly_coefficients(1 if curve == 0 else 0, chals)
            out.append(int((self.O.msm_pippenger(curve, self.g, s, threads=1) == sg[b]).all()))
        return np.array(out, np.uint8)


def make_batch(backend, tamper=None):
    rng = np.random.Generator(np.random.PCG64(2024))
    pre = rng.integers(0, 256, size=(B, K, 16), dtype=np.uint8)
    sg = np.empty((B, 64), np.uint8)
    for b in range(B):
        chals = backend.challenge_to_field(0, pre[b])
        sg[b] = backend.O.msm_pippenger(CURVE, backend.g, backend.O.b_poly_coefficients(0, chals), threads=1)
    rho = rng.integers(0, 256, size=(B, 32), dtype=np.uint8); rho[:, 31] &= 0x3F
    if tamper is not None:
        sg[tamper] = backend.g[1]
    return pre, sg, rho


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from mina_bridge_amd.sharded import ShardedAccumulatorCheck
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    be = OracleBackend()
    sh = ShardedAccumulatorCheck(be, CURVE, K)
    res = {}
    pre, sg, rho = make_batch(be)
    res["ok_proof"] = sh.verify_proof_level(pre, sg, rho).tolist()
    res["ok_sliced"] = sh.verify_base_sliced(pre, sg, rho)
    pre, sg, rho = make_batch(be, tamper=3)
    res["bad_proof"] = sh.verify_proof_level(pre, sg, rho).tolist()
    res["bad_sliced"] = sh.verify_base_sliced(pre, sg, rho)
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
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert got[r]["ok_proof"] == [1] * B
        assert got[r]["ok_sliced"] is True
        assert got[r]["bad_proof"] == [1, 1, 1, 0, 1]
        assert got[r]["bad_sliced"] is False


def test_shard_range_covers_everything():
    from mina_bridge_amd.sharded import shard_range
    for n in (0, 1, 5, 4096, 65536):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
