"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY -- see pasta_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
Byte formats: field element = 32-byte LE canonical; affine point = x||y (64 B), infinity = zeros.
numpy uint8 arrays in / out.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

FIELD_FP, FIELD_FQ = 0, 1
CURVE_PALLAS, CURVE_VESTA = 0, 1


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("pasta_oracle.c", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_init()
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _u8(x) -> np.ndarray:
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), dtype=np.uint8).copy()
    return np.ascontiguousarray(x, dtype=np.uint8)


def scalar_field_of(curve: int) -> int:
    return FIELD_FQ if curve == CURVE_PALLAS else FIELD_FP


def base_field_of(curve: int) -> int:
    return FIELD_FP if curve == CURVE_PALLAS else FIELD_FQ


def field_mul(field, a, b):
    a, b = _u8(a), _u8(b)
    n = a.size // 32
    out = np.empty(n * 32, np.uint8)
    lib().oracle_field_mul(field, ctypes.c_size_t(n), _p(a), _p(b), _p(out))
    return out.reshape(n, 32)


def field_inv(field, a):
    a = _u8(a)
    n = a.size // 32
    out = np.empty(n * 32, np.uint8)
    lib().oracle_field_inv(field, ctypes.c_size_t(n), _p(a), _p(out))
    return out.reshape(n, 32)


def field_sqrt(field, a):
    a = _u8(a)
    n = a.size // 32
    out = np.empty(n * 32, np.uint8)
    ok = np.empty(n, np.uint8)
    lib().oracle_field_sqrt(field, ctypes.c_size_t(n), _p(a), _p(out), _p(ok))
    return out.reshape(n, 32), ok


def point_add(curve, p, q):
    p, q = _u8(p), _u8(q)
    out = np.empty(64, np.uint8)
    lib().oracle_point_add(curve, _p(p), _p(q), _p(out))
    return out


def scalar_mul(curve, p, k):
    p, k = _u8(p), _u8(k)
    out = np.empty(64, np.uint8)
    lib().oracle_scalar_mul(curve, _p(p), _p(k), _p(out))
    return out


def is_on_curve(curve, p) -> bool:
    return bool(lib().oracle_is_on_curve(curve, _p(_u8(p))))


def msm_naive(curve, bases, scalars):
    bases, scalars = _u8(bases), _u8(scalars)
    n = scalars.size // 32
    out = np.empty(64, np.uint8)
    lib().oracle_msm_naive(curve, ctypes.c_size_t(n), _p(bases), _p(scalars), _p(out))
    return out


def msm_pippenger(curve, bases, scalars, threads: int = 1):
    bases, scalars = _u8(bases), _u8(scalars)
    n = scalars.size // 32
    out = np.empty(64, np.uint8)
    lib().oracle_msm_pippenger(curve, ctypes.c_size_t(n), _p(bases), _p(scalars), _p(out), int(threads))
    return out


def b_poly(field, chals, x):
    chals, x = _u8(chals), _u8(x)
    k = chals.size // 32
    out = np.empty(32, np.uint8)
    lib().oracle_b_poly(field, k, _p(chals), _p(x), _p(out))
    return out


def b_poly_coefficients(field, chals):
    chals = _u8(chals)
    k = chals.size // 32
    out = np.empty((1 << k) * 32, np.uint8)
    lib().oracle_b_poly_coefficients(field, k, _p(chals), _p(out))
    return out.reshape(1 << k, 32)


def challenge_to_field(field, chal16, endo):
    chal16, endo = _u8(chal16), _u8(endo)
    out = np.empty(32, np.uint8)
    lib().oracle_challenge_to_field(field, _p(chal16), _p(endo), _p(out))
    return out


def endo(curve):
    q = np.empty(32, np.uint8)
    r = np.empty(32, np.uint8)
    lib().oracle_endo(curve, _p(q), _p(r))
    return q, r


def blake2b512(msg: bytes) -> bytes:
    m = _u8(msg) if len(msg) else np.zeros(1, np.uint8)
    out = np.empty(64, np.uint8)
    rc = lib().oracle_blake2b512(_p(m), ctypes.c_size_t(len(msg)), _p(out))
    assert rc == 0
    return out.tobytes()


def to_group(curve, t):
    t = _u8(t)
    n = t.size // 32
    out = np.empty(n * 64, np.uint8)
    lib().oracle_to_group(curve, ctypes.c_size_t(n), _p(t), _p(out))
    return out.reshape(n, 64)


def srs_create(curve, depth: int, threads: int = 8):
    g = np.empty(depth * 64, np.uint8)
    h = np.empty(64, np.uint8)
    lib().oracle_srs_create(curve, ctypes.c_uint32(depth), _p(g), _p(h), int(threads))
    return g.reshape(depth, 64), h


def point_compress(curve, pts):
    pts = _u8(pts)
    n = pts.size // 64
    out = np.empty(n * 33, np.uint8)
    lib().oracle_point_compress(curve, ctypes.c_size_t(n), _p(pts), _p(out))
    return out.reshape(n, 33)


def point_decompress(curve, blobs):
    blobs = _u8(blobs)
    n = blobs.size // 33
    out = np.empty(n * 64, np.uint8)
    rc = lib().oracle_point_decompress(curve, ctypes.c_size_t(n), _p(blobs), _p(out))
    if rc != 0:
        raise ValueError("point not on curve")
    return out.reshape(n, 64)


def poseidon_permute(field, params, states):
    params, states = _u8(params), _u8(states).copy()
    n = states.size // 96
    lib().oracle_poseidon_permute(field, _p(params), ctypes.c_size_t(n), _p(states))
    return states.reshape(n, 96)


def poseidon_hash(field, params, inputs):
    params, inputs = _u8(params), _u8(inputs)
    n = inputs.size // 32
    if n == 0:
        inputs = np.zeros(32, np.uint8)
    out = np.empty(32, np.uint8)
    lib().oracle_poseidon_hash(field, _p(params), ctypes.c_size_t(n), _p(inputs), _p(out))
    return out


# ---- helpers shared by tests ----
def int_to_le(x: int) -> np.ndarray:
    return np.frombuffer(int(x).to_bytes(32, "little"), dtype=np.uint8).copy()


def le_to_int(b) -> int:
    return int.from_bytes(bytes(bytearray(_u8(b).tolist())), "little")


def ints_to_le(xs) -> np.ndarray:
    return np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in xs), dtype=np.uint8).copy().reshape(-1, 32)


def point_to_bytes(pt) -> np.ndarray:
    if pt is None:
        return np.zeros(64, np.uint8)
    return np.concatenate([int_to_le(pt[0]), int_to_le(pt[1])])


def bytes_to_point(b):
    b = _u8(b)
    if not b.any():
        return None
    return (le_to_int(b[:32]), le_to_int(b[32:]))
