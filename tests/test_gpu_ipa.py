"""GPU parity for a8 (`SRS::verify`, combined IPA opening check) through the C-ABI: the library must accept exactly
what the CPU restatement accepts (oracle-minted openings) and reject what it rejects."""
import numpy as np
import pytest

from ipa_helpers import mint, to_abi

pytestmark = pytest.mark.gpu

P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001


@pytest.mark.parametrize("curve,k,batch", [(0, 4, 1), (1, 4, 3), (0, 6, 2)])
def test_ipa_batch_check_accepts_and_rejects(ctx_srs, oracle, srs_oracle, curve, k, batch):
    from oracle import ipa_ref as I, pasta_ref as R
    g, h = srs_oracle[curve]
    r = R.scalar_modulus(curve)
    made = [mint(curve, g, h, k, n_polys=3, n_points=2, seed=900 + 10 * curve + b) for b in range(batch)]
    rb, sb = oracle.int_to_le(0x1234567890ABCDEF1234567890ABCDEF % r), oracle.int_to_le(0xFEDCBA0987654321 % r)

    # the CPU restatement accepts ...
    cpu_batch = []
    for e, sp in made:
        e2 = dict(e); e2["sponge"] = sp.clone(); cpu_batch.append(e2)
    assert I.ipa_verify_batch(curve, g[: 1 << k], oracle.bytes_to_point(h), cpu_batch, oracle.le_to_int(rb), oracle.le_to_int(sb))
    # ... and so does the GPU path
    ops = [to_abi(e, sp) for e, sp in made]
    assert ctx_srs.ipa_batch_check(curve, ops, rb, sb) is True
    assert ctx_srs.ipa_batch_check(curve, ops, oracle.int_to_le(1), oracle.int_to_le(1)) is True

    def tampered(idx, key, fn):
        out = [dict(o) for o in ops]
        out[idx][key] = fn(out[idx][key].copy())
        return out

    def flip(a):
        a[0] ^= 1
        return a
    last = batch - 1
    for key in ("z1", "z2", "combined_inner_product", "polyscale", "evalscale", "sponge_state", "evalpoints"):
        assert ctx_srs.ipa_batch_check(curve, tampered(last, key, flip), rb, sb) is False, key
    # swap two points inside L/R, replace delta / sg / a commitment by another valid curve point
    def swap_lr(a):
        a = a.reshape(-1, 64); a[[0, 1]] = a[[1, 0]]; return a.reshape(-1)
    assert ctx_srs.ipa_batch_check(curve, tampered(0, "lr", swap_lr), rb, sb) is False
    for key in ("delta", "sg"):
        assert ctx_srs.ipa_batch_check(curve, tampered(0, key, lambda a: g[7].copy()), rb, sb) is False, key
    def repl_comm(a):
        a = a.reshape(-1, 64); a[1] = g[9]; return a.reshape(-1)
    assert ctx_srs.ipa_batch_check(curve, tampered(last, "comms", repl_comm), rb, sb) is False
    # wrong sponge position
    bad = [dict(o) for o in ops]; bad[0]["sponge_mode"] = 1 - bad[0]["sponge_mode"]
    assert ctx_srs.ipa_batch_check(curve, bad, rb, sb) is False


@pytest.mark.parametrize("curve", [0, 1])
def test_ipa_rejects_malformed_points_the_equation_cannot_see(ctx_srs, oracle, srs_oracle, curve):
    """With polyscale = 0 the commitments after the first enter the combined MSM with scalar 0, so the equation holds
    whatever they are.  Upstream never gets that far with a malformed point (its deserialiser checks canonical
    coordinates and the curve equation): the library must reject off-curve and non-canonical encodings itself, and must
    still accept a different VALID point in a zero-weight slot."""
    from oracle import ipa_ref as I
    k = 5
    g, h = srs_oracle[curve]
    p = P if curve == 0 else Q                                   # base field of the curve
    e, sp = mint(curve, g, h, k, n_polys=3, n_points=2, seed=4100 + curve, xi=0)
    rb, sb = oracle.int_to_le(77), oracle.int_to_le(1234567)
    cpu = dict(e); cpu["sponge"] = sp.clone()
    assert I.ipa_verify_batch(curve, g[: 1 << k], oracle.bytes_to_point(h), [cpu], 77, 1234567)
    op = to_abi(e, sp)
    assert ctx_srs.ipa_batch_check(curve, [op], rb, sb) is True

    def with_comm(idx, point_bytes):
        o = dict(op); c = o["comms"].copy().reshape(-1, 64); c[idx] = point_bytes; o["comms"] = c.reshape(-1); return o
    # a different valid point (and infinity) in the zero-weight slot: still a valid opening
    assert ctx_srs.ipa_batch_check(curve, [with_comm(1, g[11])], rb, sb) is True
    assert ctx_srs.ipa_batch_check(curve, [with_comm(2, np.zeros(64, np.uint8))], rb, sb) is True
    # off-curve point there: the equation cannot notice, the input check must
    off = g[11].copy(); off[32] ^= 1
    assert not oracle.is_on_curve(curve, off)
    assert ctx_srs.ipa_batch_check(curve, [with_comm(1, off)], rb, sb) is False
    assert ctx_srs.ipa_batch_check(curve, [op, with_comm(2, off), op], rb, sb) is False      # anywhere in a batch
    # non-canonical alias of a valid point (x + p): same group element, rejected encoding
    x = oracle.le_to_int(g[11][:32])
    alias = g[11].copy(); alias[:32] = np.frombuffer((x + p).to_bytes(32, "little"), np.uint8)
    assert ctx_srs.ipa_batch_check(curve, [with_comm(1, alias)], rb, sb) is False
    # the same checks guard L/R, delta and sg (weight non-zero there, but the verdict must come from the input check too)
    for key, idx in (("lr", 3), ("delta", 0), ("sg", 0)):
        o = dict(op); a = o[key].copy().reshape(-1, 64); a[idx, 32] ^= 1; o[key] = a.reshape(-1)
        assert ctx_srs.ipa_batch_check(curve, [o], rb, sb) is False, key


def test_ipa_rejects_non_canonical_field_elements(ctx_srs, oracle, srs_oracle):
    """x and x + modulus are the same field element; upstream's deserialiser only admits the canonical one"""
    curve, k = 0, 4
    g, h = srs_oracle[curve]
    e, sp = mint(curve, g, h, k, n_polys=2, n_points=2, seed=4300)
    rb, sb = oracle.int_to_le(5), oracle.int_to_le(6)
    op = to_abi(e, sp)
    assert ctx_srs.ipa_batch_check(curve, [op], rb, sb) is True
    for key, mod in (("z1", Q), ("z2", Q), ("combined_inner_product", Q), ("polyscale", Q), ("evalscale", Q), ("evalpoints", Q), ("sponge_state", P)):
        o = dict(op); a = o[key].copy()
        v = oracle.le_to_int(a[-32:])
        if v + mod >= 1 << 256:
            continue
        a[-32:] = np.frombuffer((v + mod).to_bytes(32, "little"), np.uint8)
        o[key] = a
        assert ctx_srs.ipa_batch_check(curve, [o], rb, sb) is False, key
