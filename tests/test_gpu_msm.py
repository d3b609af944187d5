"""GPU parity for K1 (MSM) through the C-ABI: bit-exact affine result vs the CPU oracle."""
import numpy as np
import pytest

from conftest import rand_scalars

pytestmark = pytest.mark.gpu

P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
SCALAR_MOD = {0: Q, 1: P}


@pytest.mark.parametrize("curve", [1, 0])
@pytest.mark.parametrize("n", [1, 2, 31, 32, 257, 1000])
def test_msm_variable_base_small(ctx, oracle, srs_oracle, curve, n):
    g, _ = srs_oracle[curve]
    sc = rand_scalars(n, SCALAR_MOD[curve], seed=100 + n)
    assert (ctx.msm(curve, g[:n], sc) == oracle.msm_pippenger(curve, g[:n], sc, threads=4)).all()


@pytest.mark.parametrize("curve", [1, 0])
def test_msm_edge_cases(ctx, oracle, srs_oracle, curve):
    g, _ = srs_oracle[curve]
    r = SCALAR_MOD[curve]
    n = 64
    base = g[:n].copy()
    cases = {
        "zeros": oracle.ints_to_le([0] * n),
        "ones": oracle.ints_to_le([1] * n),
        "r_minus_1": oracle.ints_to_le([r - 1] * n),
        "one_nonzero": oracle.ints_to_le([0] * (n - 1) + [12345678901234567890]),
        "pow2": oracle.ints_to_le([1 << (i * 4 % 254) for i in range(n)]),
        "digit_boundaries": oracle.ints_to_le([(1 << 254) - 1, 0x8000, 0x7FFF, 0x8001, 0xFFFF, 0x10000, (1 << 240) | 0x8000] + [0x80008000800080008000] * (n - 7)),
    }
    for name, sc in cases.items():
        got, exp = ctx.msm(curve, base, sc), oracle.msm_naive(curve, base, sc)
        assert (got == exp).all(), name
    # empty input -> identity
    assert not ctx.msm(curve, np.zeros((0, 64), np.uint8), np.zeros((0, 32), np.uint8)).any()
    # repeated points (bucket collisions P = Q), P and -P (cancellation), infinity among the bases
    rep = np.repeat(g[:1], n, axis=0)
    sc = oracle.ints_to_le([7] * n)
    assert (ctx.msm(curve, rep, sc) == oracle.msm_naive(curve, rep, sc)).all()
    neg = g[:2].copy()
    m = P if curve == 0 else Q
    y = oracle.le_to_int(neg[0, 32:])
    neg[1, :32] = neg[0, :32]
    neg[1, 32:] = oracle.int_to_le(m - y)
    sc2 = oracle.ints_to_le([99, 99])
    assert not ctx.msm(curve, neg, sc2).any()                      # P*99 + (-P)*99 = identity
    withinf = base.copy(); withinf[3] = 0; withinf[10] = 0
    sc3 = rand_scalars(n, r, seed=77)
    assert (ctx.msm(curve, withinf, sc3) == oracle.msm_naive(curve, withinf, sc3)).all()


@pytest.mark.parametrize("curve,n", [(1, 65536), (0, 32768), (1, 1000), (0, 65536)])
def test_msm_srs_fixed_base_uniform(ctx_srs, oracle, srs_oracle, curve, n):
    """BASELINE config C2 (ii): uniform scalars on the real SRS, full size, bit-exact."""
    g, _ = srs_oracle[curve]
    sc = rand_scalars(n, SCALAR_MOD[curve], seed=4242 + n + curve)
    assert (ctx_srs.msm_srs(curve, sc) == oracle.msm_pippenger(curve, g[:n], sc, threads=8)).all()


@pytest.mark.parametrize("dist", ["bits128", "all_equal", "zeros_plus_one", "r_minus_1", "b_poly"])
def test_msm_srs_distributions(ctx_srs, oracle, srs_oracle, dist):
    """BASELINE config C2 (i),(iii),(iv) on Vesta 2^16."""
    curve, n, r = 1, 65536, P
    g, _ = srs_oracle[curve]
    if dist == "bits128":
        sc = rand_scalars(n, r, seed=9, bits=128)
    elif dist == "all_equal":
        sc = np.repeat(rand_scalars(1, r, seed=10), n, axis=0)      # every window: ONE bucket holds all 2^16 points
    elif dist == "zeros_plus_one":
        sc = np.zeros((n, 32), np.uint8); sc[n - 1] = rand_scalars(1, r, seed=12)[0]
    elif dist == "r_minus_1":
        sc = np.repeat(oracle.ints_to_le([r - 1]), n, axis=0)
    else:
        pre = rand_scalars(16, r, seed=13, bits=128)[:, :16]
        _, endo_r = oracle.endo(curve)
        chals = np.stack([oracle.challenge_to_field(0, pre[i].copy(), endo_r) for i in range(16)])
        sc = oracle.b_poly_coefficients(0, chals)
    assert (ctx_srs.msm_srs(curve, sc) == oracle.msm_pippenger(curve, g[:n], sc, threads=8)).all()


def test_msm_linearity_property(ctx_srs, oracle):
    """size-independent property at full size: MSM(a) + MSM(b) == MSM(a + b mod r)"""
    curve, n, r = 1, 65536, P
    a = rand_scalars(n, r, seed=31)
    b = rand_scalars(n, r, seed=32)
    ai = [int.from_bytes(x.tobytes(), "little") for x in a]
    bi = [int.from_bytes(x.tobytes(), "little") for x in b]
    s = oracle.ints_to_le([(x + y) % r for x, y in zip(ai, bi)])
    pa, pb, ps = ctx_srs.msm_srs(curve, a), ctx_srs.msm_srs(curve, b), ctx_srs.msm_srs(curve, s)
    assert (oracle.point_add(curve, pa, pb) == ps).all()


def test_msm_srs_range_slices_sum_to_full(ctx_srs, oracle, srs_oracle):
    """base-sliced sharding (SURVEY.md 8e variant 2): per-rank slices of the SRS, partial points added = full MSM"""
    curve, n, r = 1, 65536, P
    g, _ = srs_oracle[curve]
    sc = rand_scalars(n, r, seed=555)
    full = ctx_srs.msm_srs(curve, sc)
    for world in (2, 8):
        acc = np.zeros(64, np.uint8)
        for rank in range(world):
            lo, hi = n * rank // world, n * (rank + 1) // world
            part = ctx_srs.msm_srs_range(curve, lo, sc[lo:hi])
            if rank == 3:
                assert (part == oracle.msm_pippenger(curve, g[lo:hi], sc[lo:hi], threads=8)).all()
            acc = oracle.point_add(curve, acc, part)
        assert (acc == full).all()


@pytest.mark.parametrize("curve,n", [(1, 2048), (0, 5000), (1, 40000), (0, 65536)])
def test_msm_variable_base_large(ctx, oracle, srs_oracle, curve, n):
    """variable-base path at the window shapes n selects: c=11 (24 bucket sets of 1024) and c=14 (19 sets of 8192,
    155 648 buckets in total: exercises the multi-row scan and the 64-row 2-D reduction), bit-exact vs the CPU oracle"""
    g, _ = srs_oracle[curve]
    bases = g[:n][::-1].copy()                       # any points: here the SRS in reverse order
    sc = rand_scalars(n, SCALAR_MOD[curve], seed=31337 + n)
    sc[::7] = 0                                     # sprinkle zero scalars
    assert (ctx.msm(curve, bases, sc) == oracle.msm_pippenger(curve, bases, sc, threads=8)).all()


def test_msm_variable_base_adversarial_large(ctx, oracle, srs_oracle):
    """all scalars equal on 40 000 distinct points: one bucket per window holds everything (remainder-class overflow guard)"""
    curve, n = 1, 40000
    g, _ = srs_oracle[curve]
    sc = np.repeat(rand_scalars(1, P, seed=99), n, axis=0)
    assert (ctx.msm(curve, g[:n], sc) == oracle.msm_pippenger(curve, g[:n], sc, threads=8)).all()


@pytest.mark.parametrize("n", [1, 7, 31, 32, 33, 4096, 40000])
def test_msm_variable_base_heavily_duplicated_points(ctx, oracle, srs_oracle, n):
    """a batch of repeated commitments (the same proof tiled; adversarial duplicates) -- three distinct points and their negatives over n entries, so that
    nearly every bucket / task of the 29-bit accumulate holds equal or opposite points and goes through the 8 x 32 redo path; sizes on both sides of the 32-point threshold
    below which the variable-base MSM stays on the 8 x 32 law.  Sum == the oracle's, whatever the path; lane 0 and the pipelined forms."""
    curve = 0
    g, _ = srs_oracle[curve]
    three = g[[5, 6, 7]].copy()
    neg = three.copy()
    for i in range(3):                                                # Pallas' base field is P
        y = (P - int.from_bytes(three[i, 32:].tobytes(), "little")) % P
        neg[i, 32:] = np.frombuffer(y.to_bytes(32, "little"), np.uint8)
    six = np.concatenate([three, neg])
    rng = np.random.Generator(np.random.PCG64(1234 + n))
    base = six[rng.integers(0, 6, size=n)].copy()
    sc = rand_scalars(n, Q, seed=4321 + n)
    if n >= 8: sc[: n // 2] = sc[0]                                   # ... and repeated scalars: the same (point, digit) pairs land in one bucket
    want = oracle.msm_naive(curve, base, sc) if n <= 64 else oracle.msm_pippenger(curve, base, sc, threads=8)
    assert (ctx.msm(curve, base, sc) == want).all()
    ctx.set_pipeline(4)
    try:
        assert (ctx.msm(curve, base, sc) == want).all()
    finally:
        ctx.synchronize(); ctx.set_pipeline(1)


def test_msm_randomised_stress(ctx_srs, oracle, srs_oracle):
    """many small random instances with deliberately colliding inputs: repeated points (P == Q inside a bucket -> doubling
    branches of the mixed, full and cooperative adds), negated repeats (P == -Q -> identity mid-sum), infinity bases,
    zero / one / tiny / maximal scalars; both curves; variable-base and fixed-base entry points."""
    rng = np.random.Generator(np.random.PCG64(20250928))
    for trial in range(120):
        curve = int(rng.integers(0, 2))
        r = SCALAR_MOD[curve]; m = P if curve == 0 else Q
        g, _ = srs_oracle[curve]
        n = int(rng.choice([1, 2, 3, 7, 8, 9, 63, 64, 65, 200, 777, 2500]))
        pool = g[rng.integers(0, max(2, n // 4), size=n)].copy()                  # few distinct points -> many repeats
        neg_mask = rng.random(n) < 0.3
        for i in np.nonzero(neg_mask)[0]:
            y = oracle.le_to_int(pool[i, 32:])
            pool[i, 32:] = oracle.int_to_le(m - y)
        pool[rng.random(n) < 0.05] = 0                                             # some infinities
        kind = trial % 4
        if kind == 0:
            sc = rand_scalars(n, r, seed=7000 + trial)
        elif kind == 1:
            sc = oracle.ints_to_le([int(x) for x in rng.integers(0, 5, size=n)])    # 0..4
        elif kind == 2:
            sc = np.repeat(rand_scalars(1, r, seed=7000 + trial), n, axis=0)        # all equal: every repeat collides
        else:
            sc = oracle.ints_to_le([(r - 1 - int(x)) for x in rng.integers(0, 3, size=n)])
        got = ctx_srs.msm(curve, pool, sc)
        exp = oracle.msm_pippenger(curve, pool, sc, threads=2) if n > 64 else oracle.msm_naive(curve, pool, sc)
        assert (got == exp).all(), (trial, curve, n, kind)
    # fixed-base path with tiny and equal scalars on short prefixes of the SRS
    for trial in range(20):
        curve = trial & 1
        r = SCALAR_MOD[curve]
        g, _ = srs_oracle[curve]
        n = int(rng.choice([1, 5, 64, 1000, 4097]))
        sc = oracle.ints_to_le([int(x) for x in rng.integers(0, 3, size=n)]) if trial % 2 else np.repeat(rand_scalars(1, r, seed=trial), n, axis=0)
        assert (ctx_srs.msm_srs(curve, sc) == oracle.msm_pippenger(curve, g[:n], sc, threads=2)).all(), (trial, n)


@pytest.mark.parametrize("curve,n,nprob", [(1, 65536, 3), (0, 32768, 2), (1, 1000, 7), (0, 1, 4), (1, 65536, 1)])
def test_msm_srs_multi_matches_single(ctx_srs, oracle, srs_oracle, curve, n, nprob):
    """nprob MSMs over the SRS in one pipeline (one bucket set per problem) == nprob single fixed-base MSMs == oracle"""
    from conftest import rand_scalars
    from oracle import pasta_ref as R
    g, _ = srs_oracle[curve]
    r = R.scalar_modulus(curve)
    sc = rand_scalars(nprob * n, r, seed=4242 + n + nprob).reshape(nprob, n, 32)
    sc[0, :, :] = 0                                              # problem 0: all-zero scalars -> infinity
    if nprob > 1 and n > 8:
        sc[1, 8:, :] = 0                                         # problem 1: only 8 non-zero scalars
    got = ctx_srs.msm_srs_multi(curve, sc, nprob)
    assert got.shape == (nprob, 64)
    assert not got[0].any()
    for m in range(nprob):
        assert (got[m] == ctx_srs.msm_srs(curve, sc[m])).all(), m
    last = nprob - 1
    assert (got[last] == oracle.msm_pippenger(curve, g[:n], sc[last], threads=8)).all()


def test_msm_srs_multi_many_problems(ctx_srs, oracle, srs_oracle):
    """45 commitments of degree-2^12 polynomials (the per-proof commitment count of kimchi) in one call"""
    from conftest import rand_scalars
    from oracle import pasta_ref as R
    curve, n, nprob = 0, 4096, 45
    g, _ = srs_oracle[curve]
    sc = rand_scalars(nprob * n, R.scalar_modulus(curve), seed=99).reshape(nprob, n, 32)
    got = ctx_srs.msm_srs_multi(curve, sc, nprob)
    for m in (0, 22, 44):
        assert (got[m] == oracle.msm_pippenger(curve, g[:n], sc[m], threads=8)).all(), m
    assert len({bytes(x) for x in got}) == nprob


@pytest.mark.parametrize("n", [300, 5000, 70000, 140000])
def test_msm_heavy_buckets_mixed_with_light(ctx, oracle, srs_oracle, n):
    """a third of the scalars are equal (their buckets hold thousands of task partials -> the block-wide heavy-bucket sum),
    the rest uniform (ordinary quads), on repeated points so that heavy buckets also hit P + P; every variable-base window
    shape (c = 8 / 13 / 15) is crossed by the sizes"""
    curve = 0
    g, _ = srs_oracle[curve]
    base = g[np.arange(n) % 65536].copy()                     # n > 65536 reuses points
    sc = rand_scalars(n, Q, seed=31 + n)
    sc[::3] = sc[0]
    sc[1::7] = oracle.int_to_le(Q - 1)                        # and a second heavy family with all-ones digits
    assert (ctx.msm(curve, base, sc) == oracle.msm_pippenger(curve, base, sc, threads=8)).all()


def test_msm_srs_fixed_base_heavy_buckets(ctx_srs, oracle, srs_oracle):
    """fixed-base path, 2^16 equal scalars: each of the 16 windows puts all 65536 table points into one bucket"""
    curve, n = 1, 65536
    g, _ = srs_oracle[curve]
    sc = np.repeat(rand_scalars(1, P, seed=5), n, axis=0)
    sc[12345] = rand_scalars(1, P, seed=6)[0]
    assert (ctx_srs.msm_srs(curve, sc) == oracle.msm_pippenger(curve, g[:n], sc, threads=8)).all()


def test_msm_srs_multi_randomised_shapes(ctx_srs, oracle, srs_oracle):
    """random (n, nprob) incl. n not a multiple of the 256-scalar sort blocks, with zero / equal / tiny / maximal scalar
    rows mixed in: every problem's result == the oracle's; in both context modes (latency and throughput kernel forms)"""
    rng = np.random.Generator(np.random.PCG64(777))
    for trial in range(14):
        curve = int(rng.integers(0, 2))
        r = SCALAR_MOD[curve]
        g, _ = srs_oracle[curve]
        n = int(rng.choice([1, 2, 255, 256, 257, 700, 1023, 4097, 10000]))
        nprob = int(rng.integers(1, 13))
        sc = rand_scalars(nprob * n, r, seed=5000 + trial).reshape(nprob, n, 32)
        for m in range(nprob):
            kind = int(rng.integers(0, 6))
            if kind == 0: sc[m] = 0
            elif kind == 1: sc[m] = sc[m, 0]
            elif kind == 2: sc[m, :, 2:] = 0
            elif kind == 3: sc[m] = oracle.int_to_le(r - 1)
        ctx_srs.set_pipeline(1 if trial % 2 == 0 else 3)
        try:
            got = ctx_srs.msm_srs_multi(curve, sc, nprob)
        finally:
            ctx_srs.set_pipeline(1)
        for m in range(nprob):
            assert (got[m] == oracle.msm_pippenger(curve, g[:n], sc[m], threads=8)).all(), (trial, curve, n, nprob, m)


def _srs_blob(oracle, curve, g, h):
    """SRS{g, h} in the reference's file format: fixarray(2)[ array32(n)[bin8(33) ...], bin8(33) ]"""
    import struct
    comp = oracle.point_compress(curve, np.concatenate([g, h.reshape(1, 64)]))
    body = b"".join(b"\xc4\x21" + comp[i].tobytes() for i in range(len(g)))
    return b"\x92" + b"\xdd" + struct.pack(">I", len(g)) + body + b"\xc4\x21" + comp[len(g)].tobytes()


@pytest.mark.parametrize("curve", [1, 0])
def test_fixed_base_msm_on_29_bit_limbs_equals_the_8x32_law_and_the_oracle(ctx_srs, oracle, srs_oracle, curve):
    """round 4: SRS-table MSMs accumulate their buckets with the XYZZ mixed add on 9 x 29-bit limbs (ec29.cuh, lazy reduction, the table's 2^261-domain
    twin).  Same result, bit for bit, as the 8 x 32 law (`mina_verify_tuning.msm_fp29 = 0`) and as the CPU oracle: single MSMs (task form) and groups of
    6 (bucket-lane form), uniform / 128-bit / structured / all-equal scalars, values at the limb boundaries."""
    import mina_bridge_amd as m
    g, _ = srs_oracle[curve]
    r = SCALAR_MOD[curve]
    n = 4096
    ctx_srs.srs_split_table(curve)                                # msm_fp29 = 2 reads it (without it the setting behaves as 1)
    sets = {"uniform": rand_scalars(n, r, seed=501), "bits128": rand_scalars(n, r, seed=502, bits=128),
            "all_equal": oracle.ints_to_le([0x1234567890ABCDEF1234567890ABCDEF % r] * n),
            "limb_edges": oracle.ints_to_le([((1 << 29) - 1) << (29 * (i % 8)) | (1 << (29 * (i % 9))) % r for i in range(n)]),
            "r_minus_1": oracle.ints_to_le([r - 1 - i for i in range(n)])}
    for name, sc in sets.items():
        want = oracle.msm_pippenger(curve, g[:n], sc, threads=8)
        assert (ctx_srs.msm_srs(curve, sc) == want).all(), name
        for fp29, label in ((0, " (8 x 32)"), (1, " (29-bit limbs, 8-word twin table)"), (2, " (29-bit limbs, pre-split table)"), (3, " (29-bit limbs through the bucket reduction)")):
            with m.lib.tuning(msm_fp29=fp29):
                assert (ctx_srs.msm_srs(curve, sc) == want).all(), name + label
    multi = np.stack([rand_scalars(n, r, seed=600 + i) for i in range(6)])
    got = ctx_srs.msm_srs_multi(curve, multi, 6)
    for fp29 in (0, 1, 2, 3):
        with m.lib.tuning(msm_fp29=fp29):
            ref = ctx_srs.msm_srs_multi(curve, multi, 6)
        assert (got == ref).all(), fp29
    assert (got[3] == oracle.msm_pippenger(curve, g[:n], multi[3], threads=8)).all()


def test_29_bit_group_law_handles_equal_and_opposite_points_exactly(oracle, srs_oracle):
    """the exceptional cases of the mixed add (the accumulator equals the next point, or its negative) never occur on the real SRS; an SRS crafted to
    contain them -- g[1] = g[0], g[3] = -g[2], g[5] = g[4] = g[6] -- loaded through mina_srs_load puts equal and opposite points into ONE bucket when their
    scalars agree: the fp29 path must find P = 0 (mod p) on lazily reduced limbs and hand the bucket to the 8 x 32 law's redo queue (ec29.cuh xyzz29_add_affine returns false; msm.cuh msm_bucket_redo_kernel)"""
    import mina_bridge_amd as m
    curve, n = 1, 256
    g, h = srs_oracle[curve]
    g = g[:n].copy()
    fq = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001          # Vesta base field
    neg = lambda pt: np.concatenate([pt[:32], np.frombuffer(((fq - int.from_bytes(pt[32:].tobytes(), "little")) % fq).to_bytes(32, "little"), np.uint8)])
    g[1] = g[0]; g[3] = neg(g[2]); g[5] = g[4]; g[6] = g[4]
    c = m.MinaContext(0)
    try:
        c.srs_load(curve, _srs_blob(oracle, curve, g, h))
        c.srs_split_table(curve)
        assert (c.srs_get_g(curve, 0, n) == g).all()
        rng = np.random.Generator(np.random.PCG64(77))
        for trial in range(4):
            sc = rand_scalars(n, P, seed=700 + trial)
            k = rng.integers(0, 256, size=32, dtype=np.uint8); k[31] &= 0x3F
            for i in (0, 1, 2, 3, 4, 5, 6): sc[i] = k                                  # equal scalars: the same digit in every window -> the same buckets
            if trial == 1: sc[8:] = 0                                                   # nothing else in those buckets: acc == next point at the second entry
            if trial == 2: sc[4] = 0; sc[5] = 0; sc[6] = 0                              # only the opposite pair: the bucket goes through infinity
            want = oracle.msm_naive(curve, g, sc)
            for fp29 in (3, 2, 1, 0):
                with m.lib.tuning(msm_fp29=fp29):
                    assert (c.msm_srs(curve, sc) == want).all(), (trial, fp29)
                    assert (c.msm_srs_multi(curve, np.stack([sc] * 5), 5)[2] == want).all(), (trial, fp29, "bucket-lane form")
    finally:
        c.close()


def test_29_bit_bucket_reduction_handles_equal_and_opposite_partial_sums(oracle, srs_oracle):
    """round 5 (`msm_fp29 = 3`): in the multi-MSM form the buckets stay on 29-bit limbs and the 2-D bucket reduction adds them with the general XYZZ add on those limbs
    (ec29.cuh xyzz29_add, msm.cuh msm_segsum29_kernel).  Its exceptional case -- two partial sums equal or opposite -- cannot occur on honest data; here an SRS with
    g[1] = g[0], g[3] = -g[2], g[5] = g[4] and small scalars puts EQUAL points into neighbouring buckets of one row (one worker's serial loop), OPPOSITE points into
    neighbouring buckets of another, and equal points into two different workers' chunks of a third (the shuffle tree): every such segment must be flagged and recomputed
    with the complete law (msm_segsum29_redo_kernel), the result equal to the naive oracle; with random scalars on top of them as well."""
    import mina_bridge_amd as m
    curve, n = 1, 256
    g, h = srs_oracle[curve]
    g = g[:n].copy()
    fq = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
    neg = lambda pt: np.concatenate([pt[:32], np.frombuffer(((fq - int.from_bytes(pt[32:].tobytes(), "little")) % fq).to_bytes(32, "little"), np.uint8)])
    g[1] = g[0]; g[3] = neg(g[2]); g[5] = g[4]
    c = m.MinaContext(0)
    try:
        c.srs_load(curve, _srs_blob(oracle, curve, g, h))
        small = {0: 5, 1: 6, 2: 300, 3: 301, 4: 1025, 5: 1041}                          # window-0 digits only: bucket = digit - 1
        for trial in range(3):
            sc = rand_scalars(n, P, seed=900 + trial) if trial == 2 else np.zeros((n, 32), np.uint8)
            for i, v in small.items(): sc[i] = oracle.int_to_le(v)
            if trial == 1: sc[7] = oracle.int_to_le(P - 5)                               # a negative top and a wrap-around beside them
            want = oracle.msm_naive(curve, g, sc)
            for fp29 in (3, 1, 0):
                with m.lib.tuning(msm_fp29=fp29):
                    got = c.msm_srs_multi(curve, np.stack([sc] * 5), 5)
                    assert all((got[j] == want).all() for j in range(5)), (trial, fp29)
    finally:
        c.close()


@pytest.mark.parametrize("curve", [1, 0])
def test_29_bit_group_law_on_table_points_at_the_top_of_the_field(oracle, srs_oracle, curve):
    """A table coordinate enters the 29-bit law as y 2^261 mod p, canonical -- and one table point in 2^21 has that value at or above 2^254 - 2^233, where the
    top limb of a canonical number reaches the top limb of p itself.  The law subtracts without carry passes (ec29.cuh: raw "K p - y" operands), so the top limb of
    K p must cover it by itself: a first version negated table points as ONE p - y and failed the 2^18-point accumulator test on exactly such a point.  Here the
    SRS is crafted to hold them: Vesta points whose y 2^261 mod p is 2^254 + 7, 2^254 - 2, 2^254 - 2^232 + 5, p - 2, 2^254 - 2^233 + 1 and 2^254 - 2^233 - 3 (found
    by solving x^3 = y^2 - 5), met with negative AND positive digits of the signed recoding, in the task form and the bucket-lane form, against the naive oracle."""
    import mina_bridge_amd as m
    n = 256
    fq = Q if curve == 1 else P                                 # the curve's BASE field (Vesta: Fq, Pallas: Fp); scalars are reduced mod the other one
    pallas = [(0x3272e1925ed8260631c2cba71f0fbdb819950d0ce7acd0d44aee6cb848bf7f85, 0xcaeaa77ead33b5402930be6267ace798d45742f5f6caca2542aaa9fddd7b1e8, (1 << 254) + 3),
              (0x19bc51b3d4e4b80289dec45ad8eb2d3dd6f7aa4d841c5509999ce51878e41d2f, 0x107071d8070eec395479a6b34881bb2cdbfa500198874c3acd685eec9362c4b3, (1 << 254) - 1),
              (0xbfddbeaa272867f18119e4671db1e03ac21c3516fece33bf034ea2d464302b2, 0x3251558a152cc4abfd6cf419d985318694780a69cbefe07521059a663791d581, (1 << 254) - (1 << 232) - 3),
              (0x3b787af542ee31938aa017d5573312e07cb19b13117b7e6be01e1f3e4d7dc245, 0x2964727046953a3d4cc08300d5114fc055e7720b93690b8afbb0db3645dbaefc, P - 10),
              (0x38161fe0df7f6aa37e2e894f29604bb05ab97d2f2cd709da34351ed1d2834f49, 0xcaeaa7bead33b5402930be6267ace798d45743183d63c62e8fa3c5970aac0b8, (1 << 254) - (1 << 233) + 3)]
    special = pallas if curve == 0 else [(0x38f57ef5216d8ce7ee0d3eb9f8702dd5ad0bfab3bf2fd2b6f36a87367e5703d2, 0x1462ce285d1dfa8a3b27a9a36a11b0cc5aa9ed85d488ac0a38d491716d15c864, (1 << 254) + 7),
               (0x38b4606582d3f626b64ad4061c925abfb54602e92cd584595b16fb0a9b59a29, 0x398857622e89b86aca86f41a73faf20ee374c427d70bd4df4258545ad51e5909, (1 << 254) - 2),
               (0x181165aaea28a80fb9336acfd8efb429a51eab27376642157f8468dc204e53c6, 0xe6b258c8ba7b2f505ae9dbdde0ca2db1c1ca5e4ac47c93df6a3cf6de66b7a74, (1 << 254) - (1 << 232) + 5),
               (0x24bde7e1b84d0791a80595f601a3ea4331f752b432a5b14b007a4e84c7823340, 0x3a0857622e89b86aca86f41a73faf20ee3b95159cf1efe30fd70e231171e5909, fq - 2),
               (0x3f7afad333faf9c5165b4956f9b25d7a68245d3b4364145f725c2cca553ac76c, 0x27bd452e8bb23ca9abc85f2c60286f89f0216a149913bc5259d047a76df858c, (1 << 254) - (1 << 233) + 1),
               (0x3c0e1b40033c82fc3861cb9c0c6df9f61333e8010fb1064b9596a3445bdc64cf, 0x368c831745ce94a02fca6e27adf86b16442e2058de3a8f499437ddbba51c379d, (1 << 254) - (1 << 233) - 3)]
    g, h = srs_oracle[curve]
    g = g[:n].copy()
    for i, (x, y, y261) in enumerate(special):
        assert (y * y - x * x * x - 5) % fq == 0 and (y << 261) % fq == y261
        g[100 + i] = np.frombuffer(x.to_bytes(32, "little") + y.to_bytes(32, "little"), np.uint8)      # not first in their buckets: bases 10 + i share them (below)
    c = m.MinaContext(0)
    try:
        c.srs_load(curve, _srs_blob(oracle, curve, g, h))
        c.srs_split_table(curve)
        assert (c.srs_get_g(curve, 0, n) == g).all()
        for trial in range(4):
            sc = rand_scalars(n, SCALAR_MOD[curve], seed=900 + trial)
            for i in range(len(special)):
                low = 0xC000 + 17 * i if (trial + i) % 2 == 0 else 0x1234 + i          # window 0 digit negative (>= 2^15) or positive
                for base in (10 + i, 100 + i, 200 + i):                                 # the special point is the SECOND of three entries of its window-0 bucket:
                    s = int.from_bytes(sc[base].tobytes(), "little")                    # the first entry only initialises the accumulator (a normalised copy)
                    s = (s & ~0xFFFF) | low
                    if trial == 2: s = low                                              # nothing of these bases in the other windows
                    sc[base] = np.frombuffer((s % SCALAR_MOD[curve]).to_bytes(32, "little"), np.uint8)
            if trial == 3:
                keep = [b0 + i for i in range(len(special)) for b0 in (10, 100, 200)]
                mask = np.ones(n, bool); mask[keep] = False; sc[mask] = 0
            want = oracle.msm_naive(curve, g, sc)
            for fp29 in (3, 2, 1, 0):
                with m.lib.tuning(msm_fp29=fp29):
                    assert (c.msm_srs(curve, sc) == want).all(), (trial, fp29)
                    assert (c.msm_srs_multi(curve, np.stack([sc] * 5), 5)[3] == want).all(), (trial, fp29, "bucket-lane form")
    finally:
        c.close()
