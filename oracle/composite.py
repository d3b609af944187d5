"""ctypes binding of oracle/libcomposite.so -- the NATIVE CPU composite of one Proof-of-State verification (composite_oracle.c).
TEST INFRASTRUCTURE ONLY: tests/ and bench.py's cpu_baseline leg.  Inputs are the C-ABI's per-proof byte layouts, i.e. what
tests/golden/statement_k15_encoded.json holds per proof, plus flattened protocol-state records."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcomposite.so")
_lib = None


class OcProof(ctypes.Structure):
    _fields_ = [("records", ctypes.c_void_p), ("nfields", ctypes.c_void_p), ("expected", ctypes.c_void_p), ("n_old", ctypes.c_uint32), ("n_evals", ctypes.c_uint32)] + \
               [(n, ctypes.c_void_p) for n in ("plonk", "bp", "old_chals", "step_comms", "wrap_old", "wrap_sg", "sponge_digest", "prev_evals", "prev_pi", "prev_ft1", "app_state", "misc",
                                               "prev_comms", "w_comm", "z_comm", "t_comm", "evals", "ft_eval1", "lr", "delta", "sg", "z1", "z2", "acc_pre", "acc_sg")]


class OcResult(ctypes.Structure):
    _fields_ = [("hashes", ctypes.c_uint8 * (17 * 32)), ("pubs", ctypes.c_uint8 * (40 * 32)), ("ft_eval0", ctypes.c_uint8 * 32), ("cip", ctypes.c_uint8 * 32), ("v", ctypes.c_uint8 * 32),
                ("u", ctypes.c_uint8 * 32), ("step_cip", ctypes.c_uint8 * 32), ("step_b", ctypes.c_uint8 * 32), ("public_comm", ctypes.c_uint8 * 64),
                ("chain_ok", ctypes.c_int), ("statement_ok", ctypes.c_int), ("ipa_ok", ctypes.c_int), ("acc_ok", ctypes.c_int), ("verdict", ctypes.c_int)]


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_HERE, f) for f in ("composite_oracle.c", "pasta_oracle.c", "Makefile")]
        if not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libcomposite.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oc_sizeof_proof.restype = ctypes.c_size_t; _lib.oc_sizeof_result.restype = ctypes.c_size_t
        assert _lib.oc_sizeof_proof() == ctypes.sizeof(OcProof) and _lib.oc_sizeof_result() == ctypes.sizeof(OcResult), "struct layout out of sync with composite_oracle.c"
    return _lib


_keep = []


def _u8(x):
    return np.ascontiguousarray(np.frombuffer(x, np.uint8) if isinstance(x, (bytes, bytearray)) else x, dtype=np.uint8)


def setup(srs_pallas, srs_vesta, params_fp: bytes, params_fq: bytes, wrap_index: dict, step_index: dict, n_lagrange: int = 40, threads: int = 8):
    """srs_*: (g [n,64] uint8, h [64] uint8) from oracle.srs_create; wrap_index / step_index: the dicts of statement_k15_encoded.json
    (hex strings) -- the same data mina_verifier_index_install / mina_step_index_install take."""
    un = lambda h: _u8(bytes.fromhex(h))
    w, s = wrap_index, step_index
    arrs = [_u8(srs_pallas[0]), _u8(srs_pallas[1]), _u8(srs_vesta[0]), _u8(params_fp), _u8(params_fq), un(w["shifts"]), un(w["sigma_comm"]), un(w["coefficients_comm"]), un(w["selector_comm"]),
            un(w["constant_term"]), np.ascontiguousarray(s["domains"], dtype=np.uint32), un(s["shifts"]), un(s["constant_term"])]
    _keep.append(arrs)                                   # the library keeps pointers into the SRS arrays
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib().oc_setup(P(arrs[0]), P(arrs[1]), P(arrs[2]), P(arrs[3]), P(arrs[4]), int(w["log2_domain"]), int(w["zk_rows"]), int(w["perm_alpha_offset"]), P(arrs[5]), P(arrs[6]), P(arrs[7]), P(arrs[8]),
                        P(arrs[9]), ctypes.c_size_t(arrs[9].size), int(s["zk_rows"]), len(s["domains"]), P(arrs[10]), P(arrs[11]), P(arrs[12]), ctypes.c_size_t(arrs[12].size), int(n_lagrange), int(threads))
    if rc != 0:
        raise RuntimeError("oc_setup failed")


def make_proof(item: dict, records, nfields, expected):
    """item: one entry of statement_k15_encoded.json["proofs"]; records [17, 2048] uint8, nfields [17] uint32, expected [17, 32] uint8"""
    un = lambda h: _u8(bytes.fromhex(h)) if h else np.zeros(1, np.uint8)
    p = OcProof(); keep = []

    def put(name, arr):
        arr = np.ascontiguousarray(arr); keep.append(arr); setattr(p, name, arr.ctypes.data)
    put("records", _u8(records).reshape(-1)); put("nfields", np.ascontiguousarray(nfields, dtype=np.uint32)); put("expected", _u8(expected).reshape(-1))
    p.n_old, p.n_evals = int(item["n_old"]), int(item["n_evals"])
    st = item["statement"]
    for cname, key in (("plonk", "plonk"), ("bp", "bulletproof_challenges"), ("old_chals", "step_old_challenges"), ("step_comms", "step_comms"), ("wrap_old", "wrap_old_challenges"),
                       ("wrap_sg", "wrap_sg"), ("sponge_digest", "sponge_digest"), ("prev_evals", "prev_evals"), ("prev_pi", "prev_public_input"), ("prev_ft1", "prev_ft_eval1"),
                       ("app_state", "app_state"), ("misc", "misc")):
        put(cname, un(st[key]))
    for cname, key in (("prev_comms", "prev_comms"), ("w_comm", "w_comm"), ("z_comm", "z_comm"), ("t_comm", "t_comm"), ("evals", "evals"), ("ft_eval1", "ft_eval1")):
        put(cname, un(item["kimchi"][key]))
    for key in ("lr", "delta", "sg", "z1", "z2"):
        put(key, un(item["opening"][key]))
    put("acc_pre", un(item["acc_prechallenges"])); put("acc_sg", un(item["acc_sg"]))
    return p, keep


def verify_one(proof):
    p, keep = proof
    r = OcResult()
    lib().oc_verify_one(ctypes.byref(p), ctypes.byref(r))
    return r


def verify_many(proofs, threads: int) -> np.ndarray:
    arr = (OcProof * len(proofs))(*[p for p, _ in proofs])
    out = np.zeros(len(proofs), np.uint8)
    lib().oc_verify_many(arr, ctypes.c_size_t(len(proofs)), int(threads), out.ctypes.data_as(ctypes.c_void_p))
    return out


def verify_folded(proofs, threads: int, rand: bytes = None):
    """the batch folded as the GPU job folds it (composite_oracle.c oc_verify_folded): per-proof transcripts on `threads` pthreads, then ONE MSM per
    curve over the whole batch under randomisers derived from `rand` (3 x 32 bytes; default: the OS CSPRNG).  Returns (batch_ok, verdicts[n])."""
    import os
    rand = os.urandom(96) if rand is None else rand
    assert len(rand) == 96
    arr = (OcProof * len(proofs))(*[p for p, _ in proofs])
    out = np.zeros(len(proofs), np.uint8)
    lib().oc_verify_folded.restype = ctypes.c_int
    ok = lib().oc_verify_folded(arr, ctypes.c_size_t(len(proofs)), int(threads), ctypes.c_char_p(rand), out.ctypes.data_as(ctypes.c_void_p))
    return bool(ok), out


def fold_export(proofs, threads: int, rand: bytes = None):
    """one SHARD of the multi-GPU exchange variant on the CPU (composite_oracle.c oc_fold_export): (ipa_scalars [2^k,32], ipa_point [64], acc_scalars [2^16,32],
    acc_point [64], ok [n]) -- the fold of `verify_folded` with the two fixed-base MSMs left to the caller.  k = the installed wrap index's domain."""
    import os
    rand = os.urandom(96) if rand is None else rand
    arr = (OcProof * len(proofs))(*[p for p, _ in proofs])
    k = 15
    ipa_s, acc_s = np.zeros(((1 << k), 32), np.uint8), np.zeros((1 << 16, 32), np.uint8)
    ipa_p, acc_p, ok = np.zeros(64, np.uint8), np.zeros(64, np.uint8), np.zeros(len(proofs), np.uint8)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib().oc_fold_export(arr, ctypes.c_size_t(len(proofs)), int(threads), ctypes.c_char_p(rand), P(ipa_s), P(ipa_p), P(acc_s), P(acc_p), P(ok))
    return ipa_s, ipa_p, acc_s, acc_p, ok
