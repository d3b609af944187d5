"""GPU parity (through the C-ABI): field arithmetic hooks, K4 group map, SRS generation/codec."""
import hashlib

import numpy as np
import pytest

from conftest import SRS_SHA256, rand_scalars

pytestmark = pytest.mark.gpu

MODS = {0: 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,
        1: 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001}


def edge_values(oracle, field):
    p = MODS[field]
    return oracle.ints_to_le([0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << 254, (1 << 128) - 1, 5])


@pytest.mark.parametrize("field", [0, 1])
def test_field_mul_inv_sqrt(ctx, oracle, field):
    a = np.concatenate([edge_values(oracle, field), rand_scalars(3000, MODS[field], seed=11 + field)])
    b = np.concatenate([edge_values(oracle, field)[::-1], rand_scalars(3000, MODS[field], seed=21 + field)])
    assert (ctx.field_mul(field, a, b) == oracle.field_mul(field, a, b)).all()
    nz = a[np.any(a != 0, axis=1)]
    assert (ctx.field_inv(field, nz) == oracle.field_inv(field, nz)).all()
    # squares of random values are residues and must give ark's root; raw values are ~50% residues
    sq = oracle.field_mul(field, a, a)
    for vals in (sq, a):
        r_gpu, ok_gpu = ctx.field_sqrt(field, vals)
        r_cpu, ok_cpu = oracle.field_sqrt(field, vals)
        assert (ok_gpu == ok_cpu).all() and (r_gpu == r_cpu).all()
    assert ok_gpu.sum() not in (0, len(ok_gpu))


@pytest.mark.parametrize("curve", [0, 1])
def test_to_group(ctx, oracle, curve):
    f = 0 if curve == 0 else 1
    t = np.concatenate([oracle.ints_to_le([0, 1, 2, MODS[f] - 1]), rand_scalars(512, MODS[f], seed=5 + curve)])
    got, exp = ctx.to_group(curve, t), oracle.to_group(curve, t)
    assert (got == exp).all()
    assert all(oracle.is_on_curve(curve, p) for p in got[:64])


@pytest.mark.parametrize("curve", [1, 0])
def test_srs_create_matches_reference_files(ctx_srs, curve):
    """K4 on the GPU regenerates srs/{vesta,pallas}.srs byte-for-byte (the in-tree known-answer)."""
    data = ctx_srs.srs_serialize(curve)
    assert len(data) == 2293801
    assert hashlib.sha256(data).hexdigest() == SRS_SHA256[curve]


def test_srs_load_roundtrip_and_small_depth(ctx_srs, oracle):
    import mina_bridge_amd as m
    c2 = m.MinaContext(0)
    try:
        c2.srs_create(1, 300)                         # non power-of-two depth
        blob = c2.srs_serialize(1)
        g_small = c2.srs_get_g(1, 0, 300)
        assert (g_small == ctx_srs.srs_get_g(1, 0, 300)).all()
        c3 = m.MinaContext(0)
        c3.srs_load(1, blob)                          # decompress path (sqrt + y-sign flag)
        assert c3.srs_depth(1) == 300
        assert (c3.srs_get_g(1, 0, 300) == g_small).all()
        assert (c3.srs_get_h(1) == ctx_srs.srs_get_h(1)).all()
        # malformed blobs are rejected, never crash
        with pytest.raises(m.MinaError):
            c3.srs_load(1, blob[:-1])
        bad = bytearray(blob); bad[8] ^= 0xFF
        try:
            c3.srs_load(1, bytes(bad))
        except m.MinaError:
            pass
        # encodings ark's deserialiser refuses: x + q (non-canonical alias of a valid x), stray bits in the flag byte
        Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001      # Vesta base field
        assert blob[0] == 0x92 and blob[1] == 0xdc      # 300 points: rmp's array16 header
        off = 4 + 2                                   # first point: 4-byte header, then c4 21, 33 bytes
        x = int.from_bytes(blob[off:off + 32], "little")
        alias = bytearray(blob); alias[off:off + 32] = (x + Q).to_bytes(32, "little")
        with pytest.raises(m.MinaError):
            c3.srs_load(1, bytes(alias))
        stray = bytearray(blob); stray[off + 32] |= 0x01
        with pytest.raises(m.MinaError):
            c3.srs_load(1, bytes(stray))
        c3.srs_load(1, blob)                          # and the context still works afterwards
        assert (c3.srs_get_g(1, 0, 300) == g_small).all()
        c3.close()
    finally:
        c2.close()


@pytest.mark.parametrize("curve", [0, 1])
def test_cooperative_group_law_matches_single_lane(ctx, oracle, srs_oracle, curve):
    """4-lane cooperative XYZZ add/double == single-lane routines word for word (incl. the P == Q, P == -Q and
    infinity branches), and both == the CPU oracle on the resulting points."""
    g, _ = srs_oracle[curve]
    n = 300
    p = g[:n].copy(); q = g[n:2 * n].copy()
    q[5] = p[5]                      # P == Q at the first add
    q[6] = p[6]; q[6, 32:] = oracle.int_to_le(MODS[0 if curve == 0 else 1] - oracle.le_to_int(p[6, 32:]))   # Q == -P
    p[7] = 0                         # infinity inputs
    q[8] = 0
    p[9] = 0; q[9] = 0
    a, b, same = ctx.selftest_group_law(curve, p, q)
    assert same.all()
    assert (a == b).all()
    # oracle: S = ((P+Q)*2 + Q)*2 ; result = S + (S + P)
    for i in (0, 1, 5, 6, 7, 8, 9, 123, n - 1):
        s = oracle.point_add(curve, p[i], q[i])
        s = oracle.point_add(curve, s, s)
        s = oracle.point_add(curve, s, q[i])
        s = oracle.point_add(curve, s, s)
        t = oracle.point_add(curve, s, p[i])
        s = oracle.point_add(curve, s, t)
        assert (a[i] == s).all(), i
