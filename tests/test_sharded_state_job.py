"""SURVEY.md 8e.2 for the WHOLE Proof-of-State job (`ShardedStateJob`: per-proof stages on every rank, ONE exchange of folded scalar vectors and partial
points, the two fixed-base MSMs base-sliced over the ranks) -- two ranks on the GPU box (sharing GPU 0 over gloo when it has one GPU, RCCL when it has two),
full-size jobs (2^15 wrap domain, 2^16 accumulator, the committed statement fixture).  The exchanged verdict equals the single-GPU job's on every shard; a bad
opening on one rank or a bad accumulator on the other fails the BATCH on every rank, and the fallback localises it to its shard."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(world, per_rank, scenario):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = {k: v for k, v in os.environ.items() if k not in ("MINA_VERIFY_DEVICES",)}
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_state_job_worker.py"), str(per_rank), scenario], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, (so + se)[-3000:]
        outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith('{"rank"')][-1]))
    return sorted(outs, key=lambda o: o["rank"])


def test_exchange_variant_equals_the_single_gpu_job_on_every_shard():
    outs = run_ranks(2, 6, "ok")
    for o in outs:
        assert o["batch_ok"] is True and o["verdicts"] == [1] * 6 == o["plain"] and o["plain_flags"] == [1, 0, 1, 0], o
        # the exchange step is ordered by the context's pinned stream -- no full device synchronisation, ONE word read by the host
        assert o["host_syncs"] == 0 and o["host_reads"] == 1, (o["host_syncs"], o["host_reads"])


def test_a_bad_opening_on_one_rank_fails_the_batch_everywhere_and_is_localised_to_its_shard():
    outs = run_ranks(2, 6, "bad_opening_on_last_rank")
    assert all(o["batch_ok"] is False for o in outs), "the exchanged fold answers for the whole batch: every rank sees it fail"
    assert outs[0]["verdicts"] == [1] * 6, "the other shard's own folded checks pass: its verdicts stand"
    assert outs[1]["verdicts"] == [0] * 6 and outs[1]["plain_flags"][0] == 0, "the failing shard answers 0 until its culprit search (mina_state_job_batch) runs"


def test_a_bad_accumulator_fails_the_batch():
    outs = run_ranks(2, 4, "bad_accumulator_on_rank0")
    assert all(o["batch_ok"] is False for o in outs)
    assert outs[0]["plain_flags"][2] == 0 and outs[0]["verdicts"] == [0] * 4 and outs[1]["verdicts"] == [1] * 4


def test_opposite_discrepancies_on_the_first_proofs_of_two_shards_do_not_cancel():
    """proof 0 of EVERY shard used to carry coefficient 1 in its shard's fold (rho_b = rand_base^b), so z2 + t first in shard A and z2 - t first
    in shard B summed to the identity in the exchanged total and both invalid proofs were accepted, with no knowledge of the randomisers.  The exchange variant
    now folds with rho_b = rand_base^(b + 1) (ctx.h IpaShape::pow_first): both shards fail, on every rank."""
    outs = run_ranks(2, 4, "opposite_z2_on_first_proofs")
    assert all(o["batch_ok"] is False and o["detail"]["opening_fold_ok"] is False for o in outs), outs
    assert outs[0]["verdicts"] == [0] * 4 and outs[1]["verdicts"] == [0] * 4, "each shard's own folded check fails too: nothing is accepted"


def test_opposite_accumulator_discrepancies_under_upstreams_rho0_do_not_cancel():
    """sg + T first in shard A, sg - T first in shard B, acc_rho[0] = 1 on both (upstream's convention): until round 6 the accumulator leg of the
    exchange variant folded with the caller's coefficients and the two discrepancies cancelled in the exchanged total.  The library now scales a shard's acc_rho by
    a CSPRNG scalar of its own per call: the exchanged accumulator fold fails on every rank, and each shard's own check fails too."""
    outs = run_ranks(2, 4, "opposite_acc_sg_fixed_rho")
    assert all(o["batch_ok"] is False and o["detail"]["accumulator_fold_ok"] is False for o in outs), outs
    assert outs[0]["verdicts"] == [0] * 4 and outs[1]["verdicts"] == [0] * 4 and all(o["plain_flags"][2] == 0 for o in outs)


def test_exchange_variant_on_the_real_backend_with_one_rank():
    """a box with ONE GPU still has RCCL: a 1-rank group runs `ShardedStateJob` with device tensors through nccl collectives ON THE CONTEXT'S PINNED STREAM
    (torch ExternalStream) -- the code path of a real multi-GPU node (all_to_all_single / all_gather of HBM tensors, stream-ordered, one word read), which the
    2-rank runs on this box cannot take (they share GPU 0 over gloo).  Verdicts as the single-GPU job's; a bad opening fails the batch."""
    (o,) = run_ranks(1, 6, "ok")
    assert o["backend"] == "nccl" and o["batch_ok"] is True and o["verdicts"] == [1] * 6 == o["plain"], o
    assert o["host_syncs"] == 0 and o["host_reads"] == 1, o
    (o,) = run_ranks(1, 6, "bad_opening_on_last_rank")
    assert o["backend"] == "nccl" and o["batch_ok"] is False and o["verdicts"] == [0] * 6, o


def test_rccl_collectives_ride_the_pinned_lane_and_wait_for_a_delayed_job():
    """VERDICT r05 next #5a, as far as one GPU allows: (1) every collective of the exchange step (2 x all_to_all_single, 1 x all_gather) is queued on the context's
    pinned lane -- torch's current stream inside the scope IS `mina_ctx_stream` -- so the stream itself orders it behind the library's kernels; (2) ordering, not
    luck: with ~0.2 s of spin queued on that stream AHEAD of the shard's job (the host has enqueued the collectives long before a single kernel of the job has
    run) the exchanged partials are still the right ones: the batch verifies, and the call took at least the delay."""
    (o,) = run_ranks(1, 6, "ok_delayed")
    assert o["backend"] == "nccl" and o["batch_ok"] is True and o["verdicts"] == [1] * 6 == o["plain"], o
    assert len(o["coll_streams"]) == 3 and o["ctx_stream"] != 0 and all(s_ == o["ctx_stream"] for s_ in o["coll_streams"]), (o["coll_streams"], o["ctx_stream"])
    assert o["host_syncs"] == 0 and o["host_reads"] == 1, o
    assert o["call_s"] > 0.1, f"the spin kernel did not hold the stream up ({o['call_s']:.3f} s): the test proves nothing"
