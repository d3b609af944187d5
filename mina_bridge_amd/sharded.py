"""Multi-GPU sharding of the batched IPA accumulator check, one process per GPU.

Two exact strategies over a `torch.distributed` process group (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests).  Every rank holds ONLY its shard of the proofs; tensors stay on the rank's device from the first kernel to the
verdict -- RCCL moves bytes between HBMs, nothing is staged through numpy.

  * proof-level (default, zero data-path exchange): rank g verifies its proofs with its own folded MSM against its replica of the
    SRS tables (4 MiB of bases, 64 MiB of window tables per curve per GPU); the only traffic is the final all-gather of one
    verdict byte per proof.
  * base-sliced (one exchange step; pays when B/G is small and the 2^16 MSM dominates), exactly SURVEY.md 8e.2:
      1. rank g folds ITS shard's challenge polynomials into a scalar vector  S_g[j] = sum_{b in shard} rho_b s_b[j]   (K2)
      2. all-to-all over xGMI: rank r receives slice j in [r n/G, (r+1) n/G) of every S_g                  (G-1 x n/G x 32 B in)
      3. modular-add kernel folds the G slices; K1 runs over the rank's n/G bases only
      4. rank g also computes its share of the right-hand side  sum_{b in shard} rho_b sg_b  (small variable-base MSM)
      5. all-gather of the 2 partial points per rank (68-byte records), local group-law fold, compare
    RCCL has no field / curve reduction op: it only transports; the reduction operators are kernels of this library
    (`mina_field_sum_rows_dev`, `mina_points_sum_dev`).  Payloads are tiny (<= 2 MiB per curve per GPU), so the step is latency-
    bound on the 7 x ~153 GB/s links.

Both produce the verdicts of a single GPU.  The compute backend is an object with the methods of `DeviceBackend` below (device
tensors in, device tensors out), so the distributed logic runs on CPU-only machines too (tests plug an oracle-backed double).
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """contiguous, balanced [lo, hi) of `n_items` owned by `rank`"""
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


class DeviceBackend:
    """MinaContext + torch device tensors: every method queues kernels on the context and returns tensors that live in HBM.
    Two ways of ordering the library's work against torch's (allocator fills, `cat`, collectives):
      * `ordered()` (round 5; what `ShardedStateJob` uses): the context is pinned to ONE lane (`mina_ctx_pin_lane`) and that lane's stream becomes torch's current
        stream (`torch.cuda.ExternalStream`): library kernels, torch kernels and RCCL collectives are then ordered by the stream itself -- no host synchronisation.
      * `sync()` / `_ready()` outside such a scope: full synchronisation of the library's lanes and of torch's stream (the round-4 form; `ShardedAccumulatorCheck`)."""

    RECORD = 68

    def __init__(self, ctx, device):
        import torch
        self.ctx, self.dev, self.torch = ctx, device, torch
        self._ext = None
        self.host_syncs = 0                                          # full device / stream synchronisations issued through this backend (the exchange step's budget: tests read it)

    def ordered(self):
        """context manager: pin the library to its lane 0 and make that lane's stream torch's current stream.
        Needs EXCLUSIVE use of the context for the length of the scope: the pin is state of the context, and another thread's `_dev` call during the scope would be
        forced onto lane 0 as well.  Tensors allocated inside belong to the external stream's pool of torch's caching allocator: one that leaves the scope
        must be handed to the caller's stream with `record_stream` (ShardedStateJob.verify does), or the allocator may reuse its block while that stream still reads it."""
        import contextlib
        torch = self.torch
        if torch.device(self.dev).type != "cuda":
            return contextlib.nullcontext()
        be = self

        @contextlib.contextmanager
        def scope():
            be.ctx.pin_lane(0)
            if be._ext is None:
                be._ext = torch.cuda.ExternalStream(be.ctx.stream, device=be.dev)
            prev = torch.cuda.current_stream(be.dev)
            be._ext.wait_stream(prev)                                # whatever the caller queued on its stream so far (uploads of the job's sections) comes first
            try:
                with torch.cuda.stream(be._ext):
                    yield
                    prev.wait_stream(be._ext)                        # and what was queued in here is visible to the caller's stream afterwards
            finally:
                be.ctx.pin_lane(-1)
        return scope()

    def _in_order(self) -> bool:
        return self._ext is not None and self.torch.cuda.current_stream(self.dev) == self._ext

    def _buf(self, nbytes):
        return self.torch.empty(max(nbytes, 4), dtype=self.torch.uint8, device=self.dev)

    def sync(self):
        if self._in_order():
            return                                                   # one stream carries everything: nothing to wait for
        self.host_syncs += 1
        self.ctx.synchronize()
        if self.torch.device(self.dev).type == "cuda":
            self.torch.cuda.synchronize(self.dev)

    def _ready(self):
        """torch's stream and the library's streams are asynchronous to each other: whatever torch queued (a `zeros`, a `cat`, a copy) must have happened before a
        library kernel reads or writes the same memory.  (Found in round 4: `records_equal` zeroed its verdict word on torch's stream AFTER the library's
        comparison kernel had written it -- a sporadic false 'not equal'.)  Inside `ordered()` the stream is shared and orders them."""
        if self._in_order():
            return
        if self.torch.device(self.dev).type == "cuda":
            self.host_syncs += 1
            self.torch.cuda.current_stream(self.dev).synchronize()

    def accumulator_verdicts(self, curve, k, pre, sg, rho):
        """per-proof verdict bytes of this rank's shard: one folded check, per-proof checks only if it fails"""
        torch = self.torch
        b = sg.numel() // 64
        if b == 0:
            return torch.zeros(0, dtype=torch.uint8, device=self.dev)
        v = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._ready()
        self.ctx.accumulator_check_dev(curve, k, b, pre.data_ptr(), sg.data_ptr(), rho.data_ptr() if b > 1 else 0, v.data_ptr())
        self.sync()
        if int(v.item()) == 1:
            return torch.ones(b, dtype=torch.uint8, device=self.dev)
        out = torch.zeros(b, dtype=torch.int32, device=self.dev)
        self._ready()
        for lo in range(0, b, 64):
            cnt = min(64, b - lo)
            self.ctx.accumulator_check_multi_dev(curve, k, cnt, pre.data_ptr() + lo * k * 16, sg.data_ptr() + lo * 64, out.data_ptr() + 4 * lo)
        self.sync()
        return out.to(torch.uint8)

    def fold_scalars(self, field, k, pre, rho):
        """S[j] = sum_b rho_b * b_poly_coefficients(to_field(pre_b))[j]  ->  [2^k * 32] bytes"""
        b = rho.numel() // 32
        chals = self._buf(b * k * 32); out = self._buf((1 << k) * 32)
        if b == 0:
            return out.zero_()
        self._ready()
        self.ctx.challenge_to_field_dev(field, b * k, pre.data_ptr(), chals.data_ptr())
        self.sync()
        self.ctx.b_poly_fold_dev(field, k, b, chals.data_ptr(), rho.data_ptr(), out.data_ptr())
        return out

    def sum_rows(self, field, rows, m, stacked):
        out = self._buf(m * 32)
        self._ready()
        self.ctx.field_sum_rows_dev(field, rows, m, stacked.data_ptr(), out.data_ptr())
        return out

    def msm_srs_range(self, curve, first, n, scalars):
        out = self._buf(self.RECORD)
        self._ready()
        self.ctx.msm_srs_range_dev(curve, first, n, scalars.data_ptr(), out.data_ptr())
        return out[: self.RECORD]

    def msm(self, curve, n, bases, scalars):
        out = self._buf(self.RECORD)
        if n == 0:
            out.zero_(); out[64] = 1
            return out[: self.RECORD]
        self._ready()
        self.ctx.msm_dev(curve, n, bases.data_ptr(), scalars.data_ptr(), out.data_ptr())
        return out[: self.RECORD]

    def points_sum(self, curve, n, records):
        out = self._buf(self.RECORD)
        self._ready()
        self.ctx.points_sum_dev(curve, n, records.data_ptr(), out.data_ptr())
        return out[: self.RECORD]

    def state_job_fold(self, d_jobs, batch: int, k: int, acc_k: int):
        """this shard's job with the two fixed-base MSMs left out (mina_state_job_fold_dev): (per-proof verdicts uint8 [batch], flags uint8 [4] = {opening legs
        well-formed, malformed flag, accumulators well-formed, 0}, folded Pallas scalars [2^k * 32], Pallas variable-base partial [68], folded Vesta scalars
        [2^acc_k * 32], Vesta variable-base partial [68]) -- all in HBM"""
        torch, rec = self.torch, self.RECORD
        out = torch.zeros(batch + 4, dtype=torch.int32, device=self.dev)
        ipa_s, acc_s, ipa_p, acc_p = self._buf((1 << k) * 32), self._buf((1 << acc_k) * 32), self._buf(rec), self._buf(rec)
        self._ready()                                                # `out` was zeroed on torch's stream
        self.ctx.state_job_fold_dev(d_jobs, out.data_ptr(), out.data_ptr() + 4 * batch, ipa_s.data_ptr(), ipa_p.data_ptr(), acc_s.data_ptr(), acc_p.data_ptr())
        self.sync()
        return out[:batch].to(torch.uint8), out[batch: batch + 4].to(torch.uint8), ipa_s, ipa_p[:rec], acc_s, acc_p[:rec]

    def state_job_plain(self, d_jobs, batch: int):
        """the ordinary job on the same shard (mina_state_job_batch_dev): per-proof verdicts with the shard's OWN folded checks"""
        torch = self.torch
        plain = torch.zeros(batch + 4, dtype=torch.int32, device=self.dev)
        self._ready()
        self.ctx.state_job_batch_dev(d_jobs, plain.data_ptr(), plain.data_ptr() + 4 * batch)
        self.sync()
        return plain[:batch].to(torch.uint8)

    def records_equal_word(self, a, b):
        """1-element int32 tensor in HBM: 1 iff the two 68-byte point records name the same point (no host read)"""
        v = self.torch.zeros(1, dtype=self.torch.int32, device=self.dev)
        self._ready()
        self.ctx.point_records_equal_dev(a.data_ptr(), b.data_ptr(), v.data_ptr())
        return v

    def records_equal(self, a, b) -> bool:
        v = self.records_equal_word(a, b)
        self.sync()
        return bool(int(v.item()))


class ShardedAccumulatorCheck:
    def __init__(self, backend, curve: int, k: int, group=None):
        import torch.distributed as dist
        self.b, self.curve, self.k = backend, curve, k
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.scalar_field = 1 if curve == 0 else 0

    # ---- strategy 1
    def verify_proof_level(self, pre, sg, rho, total: int):
        """pre [b*k*16], sg [b*64], rho [b*32]: THIS rank's shard (uint8 tensors on its device), b = its share of `total` by
        `shard_range`.  Returns the `total` verdict bytes (same tensor on every rank)."""
        import torch
        import torch.distributed as dist
        lo, hi = shard_range(total, self.rank, self.world)
        assert sg.numel() == (hi - lo) * 64
        mine = self.b.accumulator_verdicts(self.curve, self.k, pre, sg, rho)
        cap = max(shard_range(total, r, self.world)[1] - shard_range(total, r, self.world)[0] for r in range(self.world))   # ragged shards: pad
        buf = torch.zeros(cap, dtype=torch.uint8, device=mine.device)
        buf[: hi - lo] = mine
        outs = [torch.empty_like(buf) for _ in range(self.world)]
        self.b.sync()
        dist.all_gather(outs, buf, group=self.group)
        return torch.cat([outs[r][: shard_range(total, r, self.world)[1] - shard_range(total, r, self.world)[0]] for r in range(self.world)])

    # ---- strategy 2
    def verify_base_sliced(self, pre, sg, rho) -> bool:
        """One folded check for the whole batch; every rank passes its shard.  Returns the batch verdict (same on every rank)."""
        import torch
        import torch.distributed as dist
        G, n = self.world, 1 << self.k
        assert n % G == 0, "the SRS slice per rank must be whole"
        m = n // G
        nb = sg.numel() // 64
        S = self.b.fold_scalars(self.scalar_field, self.k, pre, rho)            # 1. this shard's combined scalars, [n*32]
        recv = torch.empty_like(S)
        self.b.sync()
        dist.all_to_all_single(recv, S, group=self.group)                      # 2. slice r of every peer's vector lands here
        self.b.sync()                                                          #    (the collective ran on torch's stream, the kernels below on the library's)
        mine = self.b.sum_rows(self.scalar_field, G, m, recv)                   # 3. fold the G slices ...
        lhs = self.b.msm_srs_range(self.curve, self.rank * m, m, mine)          #    ... and commit over this rank's n/G bases
        rhs = self.b.msm(self.curve, nb, sg, rho)                               # 4. this shard's share of sum_b rho_b sg_b
        self.b.sync()                                                          # the records come off the library's stream: order them before torch touches them
        pair = torch.cat([lhs, rhs])
        outs = [torch.empty_like(pair) for _ in range(G)]
        dist.all_gather(outs, pair, group=self.group)                          # 5. 2 x 68 B per rank
        self.b.sync()
        rec = DeviceBackend.RECORD
        lrec, rrec = torch.cat([o[:rec] for o in outs]), torch.cat([o[rec:] for o in outs])
        self.b.sync()                                                          # torch's stream built the record lists; the library's stream reads them
        L = self.b.points_sum(self.curve, G, lrec)
        R = self.b.points_sum(self.curve, G, rrec)
        return self.b.records_equal(L, R)


class ShardedStateJob:
    """SURVEY.md 8e.2 for the WHOLE Proof-of-State job -- the multi-GPU variant `north_star` names ("proof batches shard across the GPUs with a single
    reduce of partial sums over xGMI"): every rank runs all per-proof stages of ITS shard (17 state hashes, Pickles statement, kimchi oracles, opening
    transcripts, both folds; `mina_state_job_fold_dev`), then ONE exchange step decides both folded checks for the whole batch:
      1. all-to-all of the shards' folded scalar vectors: rank r receives slice [r n/G, (r+1) n/G) of every shard's Pallas (2^k) and Vesta (2^acc_k) vector
      2. a modular-add kernel folds the G slices; K1 runs over the rank's n/G bases of each curve (`mina_msm_srs_range_dev`)
      3. all-gather of 4 point records per rank (its two fixed-base partials, its two variable-base partials) + one flag word
      4. every rank folds the records with the group law (`mina_points_sum_dev`): Pallas total == infinity, Vesta fixed-base total == variable-base total
    RCCL only transports; the reductions are this library's kernels.  The folding randomisers are per shard (independent CSPRNG draws) and NO proof carries a
    fixed coefficient: inside `mina_state_job_fold_dev` the opening fold uses rho_b = rand_base^(b+1), sigma_b = sg_rand_base^(b+1) (upstream's batch_verify uses
    ^b -- one coefficient-1 proof is fine in ONE combination, but here G partial sums are added and G coefficient-1 proofs could cancel each other's
    discrepancies: z2 + t first in shard A, z2 - t first in shard B), and the caller draws every `acc_rho[b]` (b = 0 included) at random.  The
    combined check is then a polynomial of degree <= B in 2G + B independent uniform variables that vanishes identically only if every proof's discrepancy is
    zero (Schwartz-Zippel: a bad batch passes with probability <= B / |F|).
    Returns (verdicts of THIS shard as a uint8 tensor, batch_ok).  When the batch fails every shard is re-run through the ordinary job: one whose OWN folded
    checks pass keeps its per-proof verdicts; a shard that fails answers 0 for all its proofs -- the per-proof culprit search is the host-form entry point's
    (`mina_state_job_batch`), which the caller runs on that shard alone."""

    REC = 68

    def __init__(self, backend, k: int = 15, acc_k: int = 16, group=None):
        """backend: DeviceBackend(ctx, device) on the GPU box; any object with its methods elsewhere (the CPU tests plug an oracle-backed double)"""
        import torch
        import torch.distributed as dist
        self.be, self.dev, self.k, self.acc_k, self.group = backend, backend.dev, k, acc_k, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.cpu_collectives = dist.get_backend(group) == "gloo"          # gloo moves host tensors (the CPU / shared-GPU tests); RCCL moves HBM to HBM
        self.torch = torch
        self.coll_streams = []                                            # raw stream handle each collective of the last calls was queued on (GPU tests)
        self.test_delay_cycles = 0                                        # test hook: a spin kernel of this many cycles on the ordering stream AHEAD of the shard's job

    def _coll(self, t):
        return t.cpu() if self.cpu_collectives else t

    def _note_stream(self):
        # which stream the collective is queued on (tests: it must be the context's pinned lane, or nothing orders it against the library's kernels)
        if self.torch.device(self.dev).type == "cuda":
            self.coll_streams.append(int(self.torch.cuda.current_stream(self.dev).cuda_stream))

    def _all_to_all(self, t):
        import torch.distributed as dist
        self._note_stream()
        src = self._coll(t).contiguous(); out = self.torch.empty_like(src)
        dist.all_to_all_single(out, src, group=self.group)
        return out.to(self.dev)

    def _all_gather(self, t):
        import torch.distributed as dist
        self._note_stream()
        src = self._coll(t).contiguous(); outs = [self.torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(outs, src, group=self.group)
        return [o.to(self.dev) for o in outs]

    def verify(self, job, batch: int):
        """job: the shard's `mina_state_jobs` with DEVICE pointers (lib.StateJobs; whatever the backend's state_job_fold takes), batch = its proof count (>= 2).
        Round 5: the whole exchange step is ordered by ONE stream (the backend's `ordered()` scope: library kernels, torch's small kernels and the RCCL collectives
        all ride the context's pinned lane), the well-formed flags and the two comparison words stay in HBM and are folded there; the host reads ONE word at the
        end -- one device-to-host copy, one wait -- and nothing else.  (Round 4: >= 8 full device synchronisations and 3 G scalar reads per call.  With gloo --
        the CPU tests, ranks sharing a GPU -- the collectives move host tensors and each costs its copies; the stream still orders everything.)"""
        torch, be, G, rec = self.torch, self.be, self.world, self.REC
        n, na = 1 << self.k, 1 << self.acc_k
        assert n % G == 0 and na % G == 0, "the SRS slices per rank must be whole"
        m, ma = n // G, na // G
        scope = be.ordered() if hasattr(be, "ordered") else None
        import contextlib
        on_gpu = torch.device(self.dev).type == "cuda"
        caller_stream = torch.cuda.current_stream(self.dev) if on_gpu else None
        with (scope if scope is not None else contextlib.nullcontext()):
            if self.test_delay_cycles and on_gpu:
                # everything below is queued at once by the host; with the stream held up here the library's kernels START late -- a collective on any other stream
                # would read the shard's vectors before they exist (tests/test_sharded_state_job.py: ordering, not luck)
                torch.cuda._sleep(int(self.test_delay_cycles))
            local, flags, ipa_s, ipa_p, acc_s, acc_p = be.state_job_fold(job, batch, self.k, self.acc_k)
            recv_p, recv_v = self._all_to_all(ipa_s), self._all_to_all(acc_s)
            mine_p, mine_v = be.sum_rows(1, G, m, recv_p), be.sum_rows(0, G, ma, recv_v)                        # Pallas scalars live in Fq, Vesta scalars in Fp
            be.sync()                                                # (no-op inside `ordered()`; a backend without it issues consecutive calls on different lanes)
            lhs_p = be.msm_srs_range(0, self.rank * m, m, mine_p)
            lhs_v = be.msm_srs_range(1, self.rank * ma, ma, mine_v)
            be.sync()
            mine = torch.cat([lhs_p[:rec], ipa_p[:rec], lhs_v[:rec], acc_p[:rec], flags.to(torch.uint8)])
            parts = torch.stack(self._all_gather(mine))              # [G, 4 rec + 4]
            col = lambda i: parts[:, i * rec: (i + 1) * rec].reshape(-1).contiguous()
            fl = parts[:, 4 * rec: 4 * rec + 4]
            wellformed = ((fl[:, 0] == 1) & (fl[:, 1] == 0) & (fl[:, 2] == 1)).all()                               # stays in HBM
            inf = torch.zeros(rec, dtype=torch.uint8, device=self.dev); inf[64] = 1
            pallas_all, vesta_l, vesta_r = torch.cat([col(0), col(1)]).contiguous(), col(2), col(3)
            pallas_total = be.points_sum(0, 2 * G, pallas_all)                                                   # fixed-base parts + variable-base parts == infinity
            L, R = be.points_sum(1, G, vesta_l), be.points_sum(1, G, vesta_r)
            be.sync()
            w_ipa, w_acc = be.records_equal_word(pallas_total, inf), be.records_equal_word(L, R)
            word = wellformed.to(torch.int32) | (w_ipa.reshape(()).to(torch.int32) << 1) | (w_acc.reshape(()).to(torch.int32) << 2)
            self.host_reads = getattr(self, "host_reads", 0) + 1
            bits = int(word.item())                                  # THE host read of the call: 1 = flags well-formed, 2 = opening fold, 4 = accumulator fold
            batch_ok = bits == 7
            self.last = {"wellformed": bool(bits & 1), "opening_fold_ok": bool(bits & 2), "accumulator_fold_ok": bool(bits & 4), "flags": fl.cpu().tolist() if not batch_ok else [[1, 0, 1, 0]] * G}
            if batch_ok:
                if on_gpu and scope is not None: local.record_stream(caller_stream)      # allocated on the ordering stream, used by the caller on its own
                return local, True
        return be.state_job_plain(job, batch), False
