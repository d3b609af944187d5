"""CPU: the IPA prover / combined-verifier restatement is self-consistent (the only pin available for a8,
SURVEY.md 8c): minted openings verify, every kind of tampering is rejected, batching randomisers do not matter."""
import pytest

from ipa_helpers import mint


@pytest.mark.parametrize("curve", [0, 1])
def test_ipa_open_verify_roundtrip_and_tamper(oracle, curve):
    from oracle import ipa_ref as I, pasta_ref as R
    k = 4
    g, h = oracle.srs_create(curve, 1 << k, threads=2)
    hp = oracle.bytes_to_point(h)
    r = R.scalar_modulus(curve)
    made = [mint(curve, g, h, k, n_polys=3, n_points=2, seed=40 + b) for b in range(2)]

    def batch(mut=None):
        out = []
        for i, (e, sp) in enumerate(made):
            e = dict(e); e["opening"] = dict(e["opening"]); e["sponge"] = sp.clone()
            if mut and mut[0] == i:
                mut[1](e)
            out.append(e)
        return out

    assert I.ipa_verify_batch(curve, g, hp, batch(), 111, 222)
    assert I.ipa_verify_batch(curve, g, hp, batch(), 1, 1)            # randomisers are free parameters
    assert I.ipa_verify_batch(curve, g, hp, batch()[:1], 5, 7)

    def bump(field):
        def f(e): e["opening"][field] = (e["opening"][field] + 1) % r
        return f
    for field in ("z1", "z2"):
        assert not I.ipa_verify_batch(curve, g, hp, batch((1, bump(field))), 111, 222)
    assert not I.ipa_verify_batch(curve, g, hp, batch((0, lambda e: e.__setitem__("combined_inner_product", (e["combined_inner_product"] + 1) % r))), 111, 222)
    assert not I.ipa_verify_batch(curve, g, hp, batch((0, lambda e: e.__setitem__("comms", e["comms"][::-1]))), 111, 222)
    assert not I.ipa_verify_batch(curve, g, hp, batch((1, lambda e: e["opening"].__setitem__("sg", oracle.bytes_to_point(g[3])))), 111, 222)

    def swap_lr(e):
        lr = list(e["opening"]["lr"]); lr[0] = (lr[0][1], lr[0][0]); e["opening"]["lr"] = lr
    assert not I.ipa_verify_batch(curve, g, hp, batch((0, swap_lr)), 111, 222)


@pytest.mark.parametrize("curve", [0, 1])
def test_shift_scalar_and_sponge_modes(oracle, curve):
    from oracle import ipa_ref as I, pasta_ref as R
    from ipa_helpers import poseidon_pp
    r = R.scalar_modulus(curve)
    # shift_scalar is a bijection with the documented closed forms
    x = 123456789
    s = I.shift_scalar(curve, x)
    if curve == 0:      # Pallas: scalar field larger than base field
        assert (s + pow(2, 255, r)) % r == x
    else:
        assert (2 * s + pow(2, 255, r) + 1) % r == x
    # sponge state machine: squeeze after <rate absorbs permutes once; absorbing after a squeeze restarts the rate
    sp = I.FqSponge(curve, poseidon_pp(curve))
    sp.absorb_fq([1])
    a = sp.challenge_fq()
    b = sp.challenge_fq()          # second squeeze reads state[1] without a permutation
    assert a != b and sp.sp.mode == "squeezed" and sp.sp.count == 2
    sp.absorb_fq([2])
    assert sp.sp.mode == "absorbed" and sp.sp.count == 1
