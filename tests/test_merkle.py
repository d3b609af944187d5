"""a16 (Proof-of-Account Merkle-path fold, BASELINE config C4 shape: 256 paths of depth 35) and the lane-cooperative
Poseidon: GPU vs the Python big-int restatement; CPU leg checks the restatement's own structure."""
import numpy as np
import pytest

from conftest import rand_scalars

P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
MODS = {0: P, 1: Q}


def pp_for(field):
    from oracle import pasta_ref as R
    import mina_bridge_amd.poseidon_params as PP
    mds, rc = PP.default_params_ints(field)
    return R.PoseidonParams(MODS[field], mds, rc, PP.NAME)


def test_merkle_restatement_structure(oracle):
    from oracle import pasta_ref as R
    pp = pp_for(0)
    assert R.merkle_prefix_field(7).to_bytes(20, "little") == b"MinaMklTree007******"
    # the salted hash equals a sponge that absorbed the prefix block, then the two children
    st = R.merkle_salt(3, pp)
    l, r = 1234567, 7654321
    sp = R.Sponge(pp); sp.absorb([R.merkle_prefix_field(3), 0]); sp.squeeze()      # one full block, permuted
    assert sp.state == st
    st0 = R.merkle_salt(0, pp)                                                    # a one-step path hashes at height 0
    st2 = [(st0[0] + l) % P, (st0[1] + r) % P, st0[2]]
    assert R.merkle_root(l, [(0, r)], pp) != R.merkle_root(l, [(1, r)], pp)      # direction matters
    assert R.merkle_root(l, [(0, r)], pp) == R.poseidon_permute(st2, pp)[0]
    # the C oracle's permutation agrees with the Python one used above
    import mina_bridge_amd.poseidon_params as PP
    got = oracle.poseidon_permute(0, PP.default_params_bytes(0), oracle.ints_to_le(st2).reshape(1, 96))
    assert oracle.le_to_int(got[0][:32]) == R.poseidon_permute(st2, pp)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("field", [0, 1])
@pytest.mark.parametrize("n,depth", [(1, 1), (3, 5), (256, 35), (5, 0)])
def test_gpu_merkle_roots(ctx, oracle, field, n, depth):
    from oracle import pasta_ref as R
    m = MODS[field]
    pp = pp_for(field)
    leaves = rand_scalars(n, m, seed=11 + n + depth)
    sib = rand_scalars(max(n * depth, 1), m, seed=9000 + 12 + n)[: n * depth]
    dirs = np.random.Generator(np.random.PCG64(13 + n)).integers(0, 2, size=n * depth, dtype=np.uint8)
    got = ctx.merkle_roots(field, leaves, sib, dirs, depth)
    check = range(n) if n <= 8 else [0, 1, n // 2, n - 1]          # big-int Python loops only on a sample
    for i in check:
        path = [(int(dirs[i * depth + h]), oracle.le_to_int(sib[i * depth + h])) for h in range(depth)]
        assert oracle.le_to_int(got[i]) == R.merkle_root(oracle.le_to_int(leaves[i]), path, pp), i
    # verify-batch: accept the right roots, reject a flipped direction / tampered sibling
    v = ctx.merkle_verify_batch(field, leaves, sib, dirs, depth, got)
    assert v.tolist() == [1] * n
    if depth:
        bad_dirs = dirs.copy(); bad_dirs[0] ^= 1
        assert ctx.merkle_verify_batch(field, leaves, sib, bad_dirs, depth, got)[0] == 0
        bad_sib = sib.copy(); bad_sib[-1, 0] ^= 1
        assert ctx.merkle_verify_batch(field, leaves, bad_sib, dirs, depth, got)[n - 1] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("field", [0, 1])
def test_gpu_cooperative_and_single_lane_poseidon_agree(ctx, oracle, field):
    """mina_poseidon_hash uses the 4-lane cooperative kernel below 65 536 sponges and one lane per sponge above:
    both must reproduce the oracle"""
    import mina_bridge_amd as m
    params = m.poseidon_params.default_params_bytes(field)
    for n, length in ((5, 3), (70000, 2)):
        inp = rand_scalars(n * length, MODS[field], seed=700 + n)
        got = ctx.poseidon_hash(field, inp, n, length)
        for i in (0, 1, n // 2, n - 1):
            assert (got[i] == oracle.poseidon_hash(field, params, inp[i * length:(i + 1) * length])).all()
