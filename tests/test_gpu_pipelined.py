"""The context's pipelined mode (`mina_ctx_set_pipeline(lanes > 1)`, what bench.py runs): the MSM pipeline then uses the
throughput forms of its tail kernels (one-lane bucket sums and row/column sums instead of the 4-lane cooperative ones) and
the `_dev` entry points rotate over lanes with their own workspaces.  Same parity bar as the single-lane mode."""
import numpy as np
import pytest

from conftest import rand_scalars
from test_gpu_sponge_ipa import make_accumulator_instance

pytestmark = pytest.mark.gpu
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001


@pytest.fixture()
def pipelined(ctx_srs):
    ctx_srs.set_pipeline(4)
    yield ctx_srs
    ctx_srs.synchronize()
    ctx_srs.set_pipeline(1)


@pytest.mark.parametrize("curve,n", [(1, 65536), (0, 32768), (1, 1000), (0, 5)])
def test_fixed_base_msm_throughput_kernels(pipelined, oracle, srs_oracle, curve, n):
    g, _ = srs_oracle[curve]
    sc = rand_scalars(n, P if curve == 1 else Q, seed=60 + n)
    assert (pipelined.msm_srs(curve, sc) == oracle.msm_pippenger(curve, g[:n], sc, threads=8)).all()
    # skewed digits: heavy-bucket queue + one-lane bucket sums
    sc[::2] = sc[0]
    assert (pipelined.msm_srs(curve, sc) == oracle.msm_pippenger(curve, g[:n], sc, threads=8)).all()


@pytest.mark.parametrize("n", [40, 3000, 70000])
def test_variable_base_msm_throughput_kernels(pipelined, oracle, srs_oracle, n):
    curve = 0
    g, _ = srs_oracle[curve]
    base = g[np.arange(n) % 65536].copy()
    sc = rand_scalars(n, Q, seed=77 + n)
    assert (pipelined.msm(curve, base, sc) == oracle.msm_pippenger(curve, base, sc, threads=8)).all()


def test_multi_problem_msm_throughput_kernels(pipelined, oracle, srs_oracle):
    curve, n, nprob = 1, 65536, 3
    g, _ = srs_oracle[curve]
    sc = rand_scalars(nprob * n, P, seed=91).reshape(nprob, n, 32)
    got = pipelined.msm_srs_multi(curve, sc, nprob)
    for m in range(nprob):
        assert (got[m] == oracle.msm_pippenger(curve, g[:n], sc[m], threads=8)).all(), m


def test_accumulator_checks_pipelined(pipelined, oracle, srs_oracle):
    curve, k = 1, 16
    inst = [make_accumulator_instance(oracle, srs_oracle, curve, k, seed=1200 + b) for b in range(3)]
    pre = np.concatenate([i[0] for i in inst]); sg = np.stack([i[1] for i in inst])
    assert pipelined.accumulator_check_multi(curve, k, pre, sg).tolist() == [1, 1, 1]
    bad = sg.copy(); bad[1] = sg[2]
    assert pipelined.accumulator_check_multi(curve, k, pre, bad).tolist() == [1, 0, 1]
    rho = rand_scalars(3, P, seed=3)
    assert pipelined.accumulator_check_batch(curve, k, pre, sg, rho).tolist() == [1, 1, 1]
    assert pipelined.accumulator_check_batch(curve, k, pre, bad, rho).tolist() == [1, 0, 1]


def test_dev_entry_points_rotate_over_lanes(pipelined, oracle, srs_oracle):
    """the bench's path: `_dev` calls issued back to back over 4 lanes, distinct inputs per call, verdict words in HBM"""
    curve, k, calls, group = 1, 16, 12, 3
    c = pipelined
    inst = [make_accumulator_instance(oracle, srs_oracle, curve, k, seed=1300 + b) for b in range(4)]
    expect, bufs = [], []
    for cidx in range(calls):
        pre = np.concatenate([inst[(cidx + j) % 4][0] for j in range(group)])
        sg = np.stack([inst[(cidx + j) % 4][1] for j in range(group)])
        exp = [1] * group
        if cidx % 3 == 1:                                       # every third call carries one wrong commitment
            j = cidx % group
            sg[j] = inst[(cidx + j + 1) % 4][1]
            exp[j] = 0
        d_pre = c.dev_upload(c.dev_malloc(pre.size), pre)
        d_sg = c.dev_upload(c.dev_malloc(sg.size), sg)
        d_v = c.dev_upload(c.dev_malloc(4 * group), np.full(group, 7, np.uint32).view(np.uint8))
        bufs.append((d_pre, d_sg, d_v)); expect.append(exp)
    for d_pre, d_sg, d_v in bufs:                               # issued back to back: 12 calls over 4 lanes, nothing waits
        c.accumulator_check_multi_dev(curve, k, group, d_pre, d_sg, d_v)
    got = [c.dev_download(b[2], 4 * group).view(np.uint32).tolist() for b in bufs]
    for b in bufs:
        for ptr in b:
            c.dev_free(ptr)
    assert got == expect
    assert any(0 in e for e in expect) and any(0 not in e for e in expect)
