"""a8 at full size: the committed opening proofs over 2^15 Pallas bases (tests/golden/ipa_pallas_k15.json with 4
commitments and ipa_pallas_k15_c45.json with the 45 commitments of the wrap-proof shape; both minted by
tests/golden/gen_ipa_fixture.py with the oracle prover).  CPU leg: the verifier restatement accepts it / rejects a
tampered copy.  GPU leg: `mina_ipa_batch_check` does the same, alone and replicated in a batch."""
import json
import os

import numpy as np
import pytest

FIXTURES = {n: json.load(open(os.path.join(os.path.dirname(__file__), "golden", n + ".json")))
            for n in ("ipa_pallas_k15", "ipa_pallas_k15_c45")}
FX = FIXTURES["ipa_pallas_k15"]


def abi_entry(fx=None):
    out = {}
    for k, v in (fx or FX)["fields"].items():
        out[k] = np.frombuffer(bytes.fromhex(v), dtype=np.uint8).copy() if isinstance(v, str) else v
    return out


def to_oracle_entry(oracle, a):
    """C-ABI layout -> the dict oracle/ipa_ref.ipa_verify_batch takes"""
    from oracle import ipa_ref as I
    from ipa_helpers import poseidon_pp
    curve, k = FX["curve"], a["k"]
    P = lambda b: oracle.bytes_to_point(b)
    lr = a["lr"].reshape(k, 2, 64)
    st = [oracle.le_to_int(a["sponge_state"][32 * i: 32 * i + 32]) for i in range(3)]
    sp = I.FqSponge(curve, poseidon_pp(curve), st, "squeezed" if a["sponge_mode"] else "absorbed", a["sponge_count"])
    return {"sponge": sp, "evalpoints": [oracle.le_to_int(x) for x in a["evalpoints"].reshape(-1, 32)],
            "polyscale": oracle.le_to_int(a["polyscale"]), "evalscale": oracle.le_to_int(a["evalscale"]),
            "comms": [P(c) for c in a["comms"].reshape(-1, 64)], "combined_inner_product": oracle.le_to_int(a["combined_inner_product"]),
            "opening": {"lr": [(P(l), P(r)) for l, r in lr], "delta": P(a["delta"]), "sg": P(a["sg"]),
                        "z1": oracle.le_to_int(a["z1"]), "z2": oracle.le_to_int(a["z2"])}}


@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_oracle_accepts_fullsize_fixture(oracle, srs_oracle, name):
    from oracle import ipa_ref as I
    fx = FIXTURES[name]
    curve = fx["curve"]
    g, h = srs_oracle[curve]
    a = abi_entry(fx)
    assert a["n_comms"] == (45 if name.endswith("c45") else 4)
    n = 1 << a["k"]
    assert I.ipa_verify_batch(curve, g[:n], oracle.bytes_to_point(h), [to_oracle_entry(oracle, a)], 7, 11)
    bad = abi_entry(fx); bad["z2"][5] ^= 4
    assert not I.ipa_verify_batch(curve, g[:n], oracle.bytes_to_point(h), [to_oracle_entry(oracle, bad)], 7, 11)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_gpu_accepts_fullsize_fixture(ctx_srs, oracle, name):
    fx = FIXTURES[name]
    curve = fx["curve"]
    rb, sb = oracle.int_to_le(0xABCDEF0123456789ABCDEF), oracle.int_to_le(0x1234567)
    a = abi_entry(fx)
    assert ctx_srs.ipa_batch_check(curve, [a], rb, sb) is True
    assert ctx_srs.ipa_batch_check(curve, [a] * 5, rb, sb) is True            # batch of identical valid openings
    for key, idx in (("z1", 0), ("lr", 700), ("comms", 100), ("combined_inner_product", 3), ("sponge_state", 40)):
        bad = abi_entry(fx); bad[key][idx] ^= 1
        assert ctx_srs.ipa_batch_check(curve, [a, bad, a], rb, sb) is False, key


@pytest.mark.gpu
def test_gpu_mixed_shapes_in_one_batch(ctx_srs, oracle):
    """openings with different commitment counts (4 and 45) combine into one check"""
    rb, sb = oracle.int_to_le(77), oracle.int_to_le(99)
    a4, a45 = abi_entry(FIXTURES["ipa_pallas_k15"]), abi_entry(FIXTURES["ipa_pallas_k15_c45"])
    assert ctx_srs.ipa_batch_check(0, [a4, a45, a45, a4], rb, sb) is True
    bad = abi_entry(FIXTURES["ipa_pallas_k15_c45"]); bad["comms"][64 * 44 + 7] ^= 2      # last commitment of the 45
    assert ctx_srs.ipa_batch_check(0, [a4, bad], rb, sb) is False
