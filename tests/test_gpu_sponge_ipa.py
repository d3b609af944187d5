"""GPU parity for K2 (b_poly), K3 (Poseidon / endo challenges) and the a10 accumulator check."""
import numpy as np
import pytest

from conftest import rand_scalars

pytestmark = pytest.mark.gpu

P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
MODS = {0: P, 1: Q}


@pytest.mark.parametrize("field", [0, 1])
@pytest.mark.parametrize("k", [1, 2, 5, 9, 16])
def test_b_poly_coefficients(ctx, oracle, field, k):
    chals = rand_scalars(k, MODS[field], seed=50 + k)
    assert (ctx.b_poly_coefficients(field, chals) == oracle.b_poly_coefficients(field, chals)).all()


@pytest.mark.parametrize("field", [0, 1])
def test_b_poly_eval_and_identity(ctx, oracle, field):
    k = 12
    chals = rand_scalars(k, MODS[field], seed=3)
    xs = rand_scalars(9, MODS[field], seed=4)
    got = ctx.b_poly(field, chals, xs)
    exp = np.stack([oracle.b_poly(field, chals, x) for x in xs])
    assert (got == exp).all()
    # property: sum_j s_j x^j == b_poly(chals, x)
    s = [oracle.le_to_int(v) for v in ctx.b_poly_coefficients(field, chals)]
    x = oracle.le_to_int(xs[0]); m = MODS[field]
    acc = 0
    for c in reversed(s):
        acc = (acc * x + c) % m
    assert acc == oracle.le_to_int(got[0])


@pytest.mark.parametrize("field", [0, 1])
@pytest.mark.parametrize("k,batch", [(6, 1), (6, 7), (10, 33), (16, 5)])
def test_b_poly_fold(ctx, oracle, field, k, batch):
    m = MODS[field]
    chals = rand_scalars(batch * k, m, seed=60 + k + batch)
    w = rand_scalars(batch, m, seed=61)
    got = ctx.b_poly_fold(field, k, chals, w)
    acc = [0] * (1 << k)
    for b in range(batch):
        s = oracle.b_poly_coefficients(field, chals[b * k:(b + 1) * k])
        wb = oracle.le_to_int(w[b])
        for j in range(1 << k):
            acc[j] = (acc[j] + wb * int.from_bytes(s[j].tobytes(), "little")) % m
    assert (got == oracle.ints_to_le(acc)).all()


@pytest.mark.parametrize("field", [0, 1])
@pytest.mark.parametrize("k,batch", [(7, 300), (9, 1030), (2, 256), (12, 257)])
def test_b_poly_fold_matrix_core_path(ctx, oracle, field, k, batch):
    """batches >= 256 take the int8 MFMA field-GEMM (bpoly_mfma.cuh): balanced base-256 digit planes, 32x32x32 tiles, anti-diagonal
    sums, one reduction per output.  Bit-exact vs the CPU fold, with extreme values in the mix (0, 1, p - 1, 2^254 - 1 as challenges
    and weights) so that every digit carry and the signed columns are exercised; batch sizes off the K-tile (padding columns)."""
    m = MODS[field]
    chals = rand_scalars(batch * k, m, seed=160 + k + batch)
    w = rand_scalars(batch, m, seed=161)
    ext = [0, 1, m - 1, (1 << 254) - 1, m - 2, 255, 256, (1 << 128) - 1]
    for i in range(0, batch * k, 7):
        chals[i] = oracle.int_to_le(ext[(i // 7) % len(ext)])
    for i in range(0, batch, 5):
        w[i] = oracle.int_to_le(ext[(i // 5 + 3) % len(ext)])
    got = ctx.b_poly_fold(field, k, chals, w)
    acc = [0] * (1 << k)
    for b in range(batch):
        s = oracle.b_poly_coefficients(field, chals[b * k:(b + 1) * k])
        wb = oracle.le_to_int(w[b])
        for j in range(1 << k):
            acc[j] = (acc[j] + wb * int.from_bytes(s[j].tobytes(), "little")) % m
    assert (got == oracle.ints_to_le(acc)).all()


@pytest.mark.parametrize("field", [0, 1])
def test_poseidon_permute_and_hash(ctx, oracle, field):
    import mina_bridge_amd as m
    params = m.poseidon_params.default_params_bytes(field)
    st = np.concatenate([np.zeros((1, 96), np.uint8), rand_scalars(3 * 700, MODS[field], seed=8).reshape(700, 96)])
    assert (ctx.poseidon_permute(field, st) == oracle.poseidon_permute(field, params, st)).all()
    for length in (0, 1, 2, 3, 5, 8):
        n = 37
        inp = rand_scalars(n * max(length, 1), MODS[field], seed=90 + length)[: n * length]
        got = ctx.poseidon_hash(field, inp, n, length)
        exp = np.stack([oracle.poseidon_hash(field, params, inp[i * length:(i + 1) * length]) for i in range(n)])
        assert (got == exp).all(), length


@pytest.mark.parametrize("field", [0, 1])
def test_challenge_to_field(ctx, oracle, field):
    curve = 1 if field == 0 else 0       # the curve whose scalar field is `field`
    _, endo_r = oracle.endo(curve)
    ch = np.concatenate([np.zeros((1, 16), np.uint8), np.full((1, 16), 255, np.uint8), rand_scalars(300, MODS[field], seed=2)[:, :16]])
    got = ctx.challenge_to_field(field, ch)
    exp = np.stack([oracle.challenge_to_field(field, c.copy(), endo_r) for c in ch])
    assert (got == exp).all()


def make_accumulator_instance(oracle, srs_oracle, curve, k, seed):
    """prechallenges + the matching sg = <b_poly_coefficients(chals), g> computed by the CPU oracle"""
    fs = 1 if curve == 0 else 0
    g, _ = srs_oracle[curve]
    _, endo_r = oracle.endo(curve)
    pre = rand_scalars(k, MODS[fs], seed=seed, bits=128)[:, :16].copy()
    chals = np.stack([oracle.challenge_to_field(fs, pre[i].copy(), endo_r) for i in range(k)])
    s = oracle.b_poly_coefficients(fs, chals)
    sg = oracle.msm_pippenger(curve, g[: 1 << k], s, threads=8)
    return pre, sg


@pytest.mark.parametrize("curve,k", [(1, 16), (0, 15), (1, 8)])
def test_accumulator_check_single(ctx_srs, oracle, srs_oracle, curve, k):
    """BASELINE config C2: one state proof's 2^16 Vesta accumulator check; accept + tampered reject."""
    pre, sg = make_accumulator_instance(oracle, srs_oracle, curve, k, seed=700 + k)
    assert ctx_srs.accumulator_check_batch(curve, k, pre, sg).tolist() == [1]
    bad = pre.copy(); bad[3, 0] ^= 1
    assert ctx_srs.accumulator_check_batch(curve, k, bad, sg).tolist() == [0]
    g, _ = srs_oracle[curve]
    assert ctx_srs.accumulator_check_batch(curve, k, pre, g[5]).tolist() == [0]


def test_accumulator_check_batch_with_culprit(ctx_srs, oracle, srs_oracle):
    curve, k, batch = 1, 10, 6
    inst = [make_accumulator_instance(oracle, srs_oracle, curve, k, seed=800 + b) for b in range(batch)]
    pre = np.concatenate([i[0] for i in inst]); sg = np.stack([i[1] for i in inst])
    rho = rand_scalars(batch, P, seed=5)
    assert ctx_srs.accumulator_check_batch(curve, k, pre, sg, rho).tolist() == [1] * batch
    sg_bad = sg.copy(); sg_bad[4] = sg[0]
    assert ctx_srs.accumulator_check_batch(curve, k, pre, sg_bad, rho).tolist() == [1, 1, 1, 1, 0, 1]


@pytest.mark.parametrize("curve,k,count", [(1, 16, 5), (0, 15, 3), (1, 10, 37), (1, 8, 1)])
def test_accumulator_check_multi_unfolded(ctx_srs, oracle, srs_oracle, curve, k, count):
    """`count` independent checks per kernel pipeline (the MSMs are problems of the multi-problem pipeline): per-proof
    verdicts equal the single-proof entry point's, with tampered proofs anywhere in a group"""
    distinct = min(count, 4)
    inst = [make_accumulator_instance(oracle, srs_oracle, curve, k, seed=900 + 13 * k + b) for b in range(distinct)]
    pre = np.concatenate([inst[b % distinct][0] for b in range(count)]); sg = np.stack([inst[b % distinct][1] for b in range(count)])
    assert ctx_srs.accumulator_check_multi(curve, k, pre, sg).tolist() == [1] * count
    bad_pre, bad_sg = pre.copy(), sg.copy()
    expect = [1] * count
    bad_pre[0 * k + 2, 5] ^= 0x10; expect[0] = 0                       # a flipped prechallenge bit in proof 0
    if count > 2:
        bad_sg[count - 1] = sg[(count - 2) % count] if distinct > 1 else oracle.point_add(curve, sg[0], sg[0]); expect[count - 1] = 0
    if count > 20:
        bad_sg[17] = 0; expect[17] = 0                                  # infinity instead of the commitment, second group
    got = ctx_srs.accumulator_check_multi(curve, k, bad_pre, bad_sg).tolist()
    assert got == expect
    for b in sorted({0, count - 1}):
        assert ctx_srs.accumulator_check_batch(curve, k, bad_pre[b * k:(b + 1) * k], bad_sg[b]).tolist() == [expect[b]]


@pytest.mark.parametrize("n", [8192, 8193, 199999, 200000])
def test_poseidon_hash_every_kernel_form(ctx, oracle, n):
    """the sponge-hash entry point picks its kernel by batch size: 8 lanes per sponge (<= 8192), 4 lanes (< 200 000), one lane;
    each side of both thresholds against the oracle (sampled sponges; the inputs repeat with period 1009)"""
    import mina_bridge_amd as m
    field, length = 0, 5
    params = m.poseidon_params.default_params_bytes(field)
    base = rand_scalars(1009 * length, MODS[field], seed=4242).reshape(1009, length * 32)
    inp = base[np.arange(n) % 1009].copy()
    got = ctx.poseidon_hash(field, inp, n, length)
    exp = {i: oracle.poseidon_hash(field, params, base[i].reshape(length, 32)) for i in (0, 1, 500, 1008)}
    for i in (0, 1, 500, 1008, 1009, n - 1, n // 2):
        ref = exp.get(i % 1009)
        assert (got[i] == (ref if ref is not None else got[i % 1009])).all(), i
    assert (got[:1009] == got[1009:2018]).all()                    # same inputs, other lanes / waves


def test_accumulator_check_rejects_alias_encoding_of_sg(ctx_srs, oracle, srs_oracle):
    """(x + p, y) names the same point as (x, y); only the canonical encoding is a valid proof field"""
    curve, k = 1, 8
    pre, sg = make_accumulator_instance(oracle, srs_oracle, curve, k, seed=4400)
    assert ctx_srs.accumulator_check_batch(curve, k, pre, sg).tolist() == [1]
    x = oracle.le_to_int(sg[:32])
    alias = sg.copy(); alias[:32] = np.frombuffer((x + Q).to_bytes(32, "little"), np.uint8)     # Vesta base field = Fq
    assert ctx_srs.accumulator_check_batch(curve, k, pre, alias).tolist() == [0]
    assert ctx_srs.accumulator_check_multi(curve, k, np.concatenate([pre, pre]), np.stack([alias, sg])).tolist() == [0, 1]


def test_folded_accumulator_check_validates_sg(ctx_srs, oracle, srs_oracle):
    """folded (random-combination) path: a malformed sg -- alias encoding or off-curve -- fails the batch and is singled out"""
    curve, k = 1, 8
    inst = [make_accumulator_instance(oracle, srs_oracle, curve, k, seed=4500 + b) for b in range(3)]
    pre = np.concatenate([i[0] for i in inst]); sg = np.stack([i[1] for i in inst])
    rho = rand_scalars(3, P, seed=9)
    assert ctx_srs.accumulator_check_batch(curve, k, pre, sg, rho).tolist() == [1, 1, 1]
    x = oracle.le_to_int(sg[1][:32])
    alias = sg.copy(); alias[1, :32] = np.frombuffer((x + Q).to_bytes(32, "little"), np.uint8)
    assert ctx_srs.accumulator_check_batch(curve, k, pre, alias, rho).tolist() == [1, 0, 1]
    off = sg.copy(); off[2, 40] ^= 4
    assert ctx_srs.accumulator_check_batch(curve, k, pre, off, rho).tolist() == [1, 1, 0]
