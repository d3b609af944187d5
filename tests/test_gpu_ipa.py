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
