"""Pins the oracle against the ONLY golden data the reference tree holds for this path: the two SRS files.
BLAKE2b-512 -> bit packing -> BW group map -> ark Tonelli-Shanks root -> compressed codec -> MessagePack must
reproduce srs/vesta.srs and srs/pallas.srs byte-for-byte (sha256)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import SRS_SHA256

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("curve", [1, 0])
def test_srs_regenerates_byte_for_byte(oracle, srs_oracle, curve):
    from oracle import pasta_ref as R
    g, h = srs_oracle[curve]
    blobs = oracle.point_compress(curve, g)
    hb = oracle.point_compress(curve, h)
    data = R.srs_serialize([b.tobytes() for b in blobs], hb[0].tobytes())
    assert len(data) == 2293801
    assert hashlib.sha256(data).hexdigest() == SRS_SHA256[curve]


@pytest.mark.parametrize("curve", [1, 0])
def test_srs_head_tail_fixture(oracle, srs_oracle, curve):
    """first/last 8 points + h as committed under tests/golden (copied from the reference files' bytes)"""
    fx = json.load(open(os.path.join(GOLD, "srs_head_tail.json")))[str(curve)]
    g, h = srs_oracle[curve]
    blobs = oracle.point_compress(curve, np.concatenate([g[:8], g[-8:]]))
    assert [b.tobytes().hex() for b in blobs] == fx["g_head"] + fx["g_tail"]
    assert oracle.point_compress(curve, h)[0].tobytes().hex() == fx["h"]
    # decompression is the inverse
    dec = oracle.point_decompress(curve, blobs)
    assert (dec == np.concatenate([g[:8], g[-8:]])).all()


@pytest.mark.parametrize("curve", [1, 0])
def test_python_twin_agrees_on_srs_points(oracle, srs_oracle, curve):
    from oracle import pasta_ref as R
    g, h = srs_oracle[curve]
    bw = R.BWParams(R.base_modulus(curve))
    for i in (0, 1, 2, 65535):
        assert oracle.bytes_to_point(g[i]) == R.srs_point(curve, i, bw)
    assert oracle.bytes_to_point(h) == R.srs_h(curve, bw)
    m, r = R.base_modulus(curve), R.scalar_modulus(curve)
    p0 = oracle.bytes_to_point(g[0])
    assert R.is_on_curve(p0, m) and R.scalar_mul(r, p0, m) is None   # on the curve, of prime order
