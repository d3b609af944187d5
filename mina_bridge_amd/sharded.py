"""Multi-GPU sharding of the batched IPA accumulator check, one process per GPU.

Two exact strategies over a `torch.distributed` process group (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests):

  * proof-level (default, zero data-path exchange): rank g verifies proofs [g*B/G, (g+1)*B/G) with its own folded
    MSM against its replica of the SRS tables (4 MiB of bases, 64 MiB of window tables per curve per GPU); the only
    traffic is the final all-gather of one verdict byte per proof.
  * base-sliced (one exchange step; pays when B/G is small and the 2^16 MSM dominates): every rank folds the whole
    batch's challenge polynomials but runs the MSM only over its slice g[r*n/G, (r+1)*n/G) of the SRS; the G partial
    points (64 B each) are all-gathered and summed with the group law locally -- RCCL moves bytes, the reduction
    operator (Pasta point addition) is ours, as RCCL has no such op.  Payload per GPU: 64 B out, 64*(G-1) B in, so the
    step is latency-bound (~10-30 us on xGMI), never bandwidth-bound.

Both produce the same verdicts as a single GPU.  The compute backend is any object with the `MinaContext` methods used
below, so the distributed logic can be exercised on CPU-only machines (tests plug the CPU oracle there).
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """contiguous, balanced [lo, hi) of `n_items` owned by `rank`"""
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


def _all_gather_bytes(arr: np.ndarray, group, device):
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1)).to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    return [o.cpu().numpy() for o in outs]


class ShardedAccumulatorCheck:
    def __init__(self, backend, curve: int, k: int, group=None, device="cpu"):
        import torch.distributed as dist
        self.b, self.curve, self.k = backend, curve, k
        self.group = group
        self.device = device
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.scalar_field = 1 if curve == 0 else 0

    # ---- strategy 1
    def verify_proof_level(self, prechallenges: np.ndarray, sg: np.ndarray, rho: np.ndarray) -> np.ndarray:
        """prechallenges [B,k,16], sg [B,64], rho [B,32] (identical on every rank) -> verdict bytes [B] on every rank"""
        B = sg.shape[0]
        lo, hi = shard_range(B, self.rank, self.world)
        mine = np.zeros(0, np.uint8)
        if hi > lo:
            mine = self.b.accumulator_check_batch(self.curve, self.k, prechallenges[lo:hi].reshape(-1), sg[lo:hi].reshape(-1),
                                                  rho[lo:hi].reshape(-1) if hi - lo > 1 else None)
        # ragged shards: pad to the largest shard for the all-gather
        cap = max(shard_range(B, r, self.world)[1] - shard_range(B, r, self.world)[0] for r in range(self.world))
        buf = np.zeros(cap, np.uint8)
        buf[: hi - lo] = mine
        parts = _all_gather_bytes(buf, self.group, self.device)
        out = np.zeros(B, np.uint8)
        for r, p in enumerate(parts):
            a, z = shard_range(B, r, self.world)
            out[a:z] = p[: z - a]
        return out

    # ---- strategy 2
    def verify_base_sliced(self, prechallenges: np.ndarray, sg: np.ndarray, rho: np.ndarray) -> bool:
        """One folded check for the whole batch, MSM sharded over the SRS bases.  Returns the batch verdict."""
        B = sg.shape[0]
        n = 1 << self.k
        chals = self.b.challenge_to_field(self.scalar_field, prechallenges.reshape(-1, 16))
        folded = self.b.b_poly_fold(self.scalar_field, self.k, chals.reshape(-1), rho.reshape(-1))     # [n,32]
        lo, hi = shard_range(n, self.rank, self.world)
        part = self.b.msm_srs_range(self.curve, lo, folded[lo:hi].reshape(-1)) if hi > lo else np.zeros(64, np.uint8)
        parts = np.stack(_all_gather_bytes(part, self.group, self.device))                              # [G,64]
        ones = np.zeros((self.world, 32), np.uint8); ones[:, 0] = 1
        lhs = self.b.msm(self.curve, parts.reshape(-1), ones.reshape(-1))                               # EC fold of the partials
        rhs = self.b.msm(self.curve, sg.reshape(-1), rho.reshape(-1))                                   # sum_b rho_b * sg_b
        return bool((lhs == rhs).all())
