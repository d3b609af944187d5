"""ctypes binding of libminaverify.so -- the host-side mirror of include/mina_verify.h.

The library is HIP-only (gfx950).  There is NO CPU fallback: if the shared object is missing, or no
GPU is visible, loading / context creation raises.  Nothing here imports `oracle/`.

If PyTorch-ROCm shares the process (device tensors for the `_dev` entry points, torch.distributed), import torch and touch
`torch.cuda` BEFORE creating the first MinaContext: the wheel bundles its own HIP runtime and must initialise first.

Byte conventions (same as the C-ABI): field element = 32-byte LE canonical; affine point = x||y
(64 B), infinity = zeros; numpy uint8 arrays in and out.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libminaverify.so")

FIELD_FP, FIELD_FQ = 0, 1
CURVE_PALLAS, CURVE_VESTA = 0, 1

# every symbol include/mina_verify.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "mina_ctx_create", "mina_ctx_destroy", "mina_last_error", "mina_ctx_synchronize", "mina_ctx_stream", "mina_ctx_pin_lane", "mina_ctx_set_pipeline", "mina_prof_enable", "mina_prof_read",
    "mina_dev_malloc", "mina_dev_free", "mina_dev_upload", "mina_dev_download",
    "mina_srs_create", "mina_srs_split_table", "mina_srs_load", "mina_srs_depth", "mina_srs_get_g", "mina_srs_get_h", "mina_srs_lagrange_basis", "mina_public_input_commitment", "mina_public_input_commitment_batch", "mina_combined_inner_product", "mina_srs_serialize",
    "mina_msm", "mina_msm_srs", "mina_msm_srs_range", "mina_msm_srs_multi", "mina_msm_srs_dev",
    "mina_b_poly", "mina_b_poly_coefficients", "mina_b_poly_fold", "mina_b_poly_fold_dev",
    "mina_poseidon_set_params", "mina_poseidon_params_parse", "mina_poseidon_load_params", "mina_poseidon_load_params_file", "mina_poseidon_permute", "mina_poseidon_permute_dev", "mina_poseidon_hash",
    "mina_challenge_to_field", "mina_fq_sponge_run", "mina_to_group", "mina_merkle_roots", "mina_merkle_verify_batch",
    "mina_field_mul", "mina_field_inv", "mina_field_sqrt", "mina_selftest_group_law",
    "mina_accumulator_check_batch", "mina_accumulator_check_dev", "mina_accumulator_check_multi_dev", "mina_accumulator_check_multi", "mina_ipa_batch_check",
    "mina_consensus_project_window", "mina_consensus_relative_min_window_density", "mina_consensus_is_short_range",
    "mina_protocol_state_pack", "mina_protocol_state_hash_batch", "mina_protocol_state_hash_bytes",
    "mina_state_jobs_prepare", "mina_state_job_batch_dev", "mina_state_job_batch", "mina_state_job_fold_dev",
    "mina_challenge_to_field_dev", "mina_field_sum_rows_dev", "mina_msm_srs_range_dev", "mina_msm_dev", "mina_points_sum_dev", "mina_point_records_equal_dev",
    "mina_step_index_install", "mina_step_index_load_json", "mina_polish_tokens_from_json", "mina_verifier_index_load_json", "mina_pickles_public_input",
    "mina_verifier_index_install", "mina_verifier_index_digest", "mina_kimchi_to_batch", "mina_pickles_public_inputs_batch", "mina_wrap_proof_flatten", "mina_state_proof_split",
    "mina_verify_state", "mina_verify_state_batch", "mina_verify_state_checks", "mina_verify_state_files", "mina_verify_account", "mina_verify_account_batch",
    "mina_verify_account_files", "mina_verify_account_checks", "mina_verify_account_ctx", "mina_account_hash_batch", "mina_account_abi_encode", "mina_verify_configure", "mina_verify_shutdown", "mina_verify_global_ctx", "mina_poseidon_params_name",
    "mina_verify_device_count", "mina_verify_device_ctx", "mina_verify_set_network", "mina_verify_install_verifier_index", "mina_verify_install_step_index", "mina_verify_set_poseidon_params",
    "mina_poseidon_install_default_params",
    "verify_mina_state_ffi", "verify_account_inclusion_ffi", "verify_mina_state_ffi_u32", "verify_account_inclusion_ffi_u32",
    "mina_verify_tuning_default", "mina_verify_tuning_get", "mina_verify_configure_ex", "mina_verify_retired_env",
    "mina_consensus_select_secure_chain", "mina_parse_state_pub_inputs", "mina_parse_account_pub_inputs", "mina_parse_merkle_path", "mina_verify_account_inclusion",
]


class MinaError(RuntimeError):
    pass


class StatePubInputs(ctypes.Structure):
    _fields_ = [("is_state_proof_from_devnet", ctypes.c_uint8), ("bridge_tip_state_hash", ctypes.c_uint8 * 32),
                ("candidate_chain_state_hashes", (ctypes.c_uint8 * 32) * 16), ("candidate_chain_ledger_hashes", (ctypes.c_uint8 * 32) * 16)]


def combined_inner_product(field: int, evals, polyscale, evalscale, n_polys: int, n_points: int) -> np.ndarray:
    lib = load_library()
    ev = _u8(evals) if n_polys * n_points else np.zeros(32, np.uint8)
    out = np.empty(32, np.uint8)
    rc = lib.mina_combined_inner_product(field, ctypes.c_size_t(n_polys), ctypes.c_size_t(n_points), _p(ev), _p(_u8(polyscale)), _p(_u8(evalscale)), _p(out))
    if rc != 0:
        raise MinaError(f"mina_combined_inner_product failed ({rc}): {lib.mina_last_error().decode()}")
    return out


def parse_state_pub_inputs(data: bytes) -> dict:
    """MinaStatePubInputs bytes (1057) -> dict; raises MinaError on malformed input.  No GPU needed."""
    lib = load_library()
    out = StatePubInputs()
    b = _u8(data) if len(data) else np.zeros(1, np.uint8)
    rc = lib.mina_parse_state_pub_inputs(_p(b), ctypes.c_size_t(len(data)), ctypes.byref(out))
    if rc != 0:
        raise MinaError(f"mina_parse_state_pub_inputs failed ({rc}): {lib.mina_last_error().decode()}")
    return {"is_state_proof_from_devnet": bool(out.is_state_proof_from_devnet), "bridge_tip_state_hash": bytes(out.bridge_tip_state_hash),
            "candidate_chain_state_hashes": [bytes(h) for h in out.candidate_chain_state_hashes],
            "candidate_chain_ledger_hashes": [bytes(h) for h in out.candidate_chain_ledger_hashes]}


def parse_account_pub_inputs(data: bytes):
    lib = load_library()
    b = _u8(data) if len(data) else np.zeros(1, np.uint8)
    lh = np.empty(32, np.uint8); off = ctypes.c_size_t(0); ln = ctypes.c_size_t(0)
    rc = lib.mina_parse_account_pub_inputs(_p(b), ctypes.c_size_t(len(data)), _p(lh), ctypes.byref(off), ctypes.byref(ln))
    if rc != 0:
        raise MinaError(f"mina_parse_account_pub_inputs failed ({rc}): {lib.mina_last_error().decode()}")
    return lh.tobytes(), data[off.value: off.value + ln.value]


def parse_merkle_path(proof: bytes, max_depth: int = 64):
    lib = load_library()
    b = _u8(proof) if len(proof) else np.zeros(1, np.uint8)
    sib = np.empty(max_depth * 32, np.uint8); dirs = np.empty(max_depth, np.uint8)
    depth = ctypes.c_uint32(0); off = ctypes.c_size_t(0)
    rc = lib.mina_parse_merkle_path(_p(b), ctypes.c_size_t(len(proof)), ctypes.c_uint32(max_depth), _p(sib), _p(dirs), ctypes.byref(depth), ctypes.byref(off))
    if rc != 0:
        raise MinaError(f"mina_parse_merkle_path failed ({rc}): {lib.mina_last_error().decode()}")
    d = depth.value
    return sib[: d * 32].reshape(d, 32).copy(), dirs[:d].copy(), off.value


class ConsensusParams(ctypes.Structure):
    _fields_ = [("slots_per_sub_window", ctypes.c_uint32), ("sub_windows_per_window", ctypes.c_uint32)]


class ConsensusState(ctypes.Structure):
    _fields_ = [("blockchain_length", ctypes.c_uint32), ("epoch_count", ctypes.c_uint32), ("curr_global_slot", ctypes.c_uint32),
                ("min_window_density", ctypes.c_uint32), ("sub_window_densities", ctypes.c_uint32 * 16),
                ("staking_lock_checkpoint", ctypes.c_uint8 * 32), ("next_lock_checkpoint", ctypes.c_uint8 * 32),
                ("last_vrf_output_hash", ctypes.c_uint8 * 32), ("state_hash", ctypes.c_uint8 * 32)]

    @classmethod
    def make(cls, length, epoch, slot, min_density, window, staking_cp=b"", next_cp=b"", vrf=b"", state_hash=b""):
        s = cls()
        s.blockchain_length, s.epoch_count, s.curr_global_slot, s.min_window_density = length, epoch, slot, min_density
        for i, d in enumerate(window):
            s.sub_window_densities[i] = d
        for name, val in (("staking_lock_checkpoint", staking_cp), ("next_lock_checkpoint", next_cp), ("last_vrf_output_hash", vrf), ("state_hash", state_hash)):
            buf = bytes(val).ljust(32, b"\0")[:32]
            ctypes.memmove(getattr(s, name), buf, 32)
        return s


MAINNET_CONSENSUS = ConsensusParams(7, 11)


def consensus_project_window(state, next_slot, params=MAINNET_CONSENSUS):
    lib = load_library()
    out = (ctypes.c_uint32 * 16)()
    rc = lib.mina_consensus_project_window(ctypes.byref(params), ctypes.byref(state), ctypes.c_uint32(next_slot), out)
    if rc != 0:
        raise MinaError(f"mina_consensus_project_window failed ({rc}): {lib.mina_last_error().decode()}")
    return list(out)[: params.sub_windows_per_window]


def consensus_relative_min_window_density(a, b, params=MAINNET_CONSENSUS) -> int:
    lib = load_library()
    out = ctypes.c_uint32(0)
    rc = lib.mina_consensus_relative_min_window_density(ctypes.byref(params), ctypes.byref(a), ctypes.byref(b), ctypes.byref(out))
    if rc != 0:
        raise MinaError(f"mina_consensus_relative_min_window_density failed ({rc}): {lib.mina_last_error().decode()}")
    return out.value


def consensus_is_short_range(a, b) -> bool:
    return bool(load_library().mina_consensus_is_short_range(ctypes.byref(a), ctypes.byref(b)))


def consensus_select_secure_chain(tip, candidate, params=MAINNET_CONSENSUS) -> bool:
    """True if the candidate replaces the tip (img/consensus07.png)"""
    lib = load_library()
    out = ctypes.c_int(0)
    rc = lib.mina_consensus_select_secure_chain(ctypes.byref(params), ctypes.byref(tip), ctypes.byref(candidate), ctypes.byref(out))
    if rc != 0:
        raise MinaError(f"mina_consensus_select_secure_chain failed ({rc}): {lib.mina_last_error().decode()}")
    return bool(out.value)


class IpaOpening(ctypes.Structure):
    _fields_ = [
        ("k", ctypes.c_uint32),
        ("lr", ctypes.c_void_p), ("delta", ctypes.c_void_p), ("sg", ctypes.c_void_p),
        ("z1", ctypes.c_void_p), ("z2", ctypes.c_void_p),
        ("n_evalpoints", ctypes.c_uint32), ("evalpoints", ctypes.c_void_p),
        ("n_comms", ctypes.c_uint32), ("comms", ctypes.c_void_p),
        ("combined_inner_product", ctypes.c_void_p), ("polyscale", ctypes.c_void_p), ("evalscale", ctypes.c_void_p),
        ("sponge_state", ctypes.c_void_p), ("sponge_mode", ctypes.c_uint32), ("sponge_count", ctypes.c_uint32),
    ]


PSTATE_SLOTS, STATES_PER_PROOF = 64, 17
ENC_BINPROT, ENC_BINCODE = 0, 1


class ProtocolStateInfo(ctypes.Structure):
    _fields_ = [("previous_state_hash", ctypes.c_uint8 * 32), ("genesis_state_hash", ctypes.c_uint8 * 32), ("snarked_ledger_hash", ctypes.c_uint8 * 32),
                ("n_body_fields", ctypes.c_uint32), ("k", ctypes.c_uint32), ("slots_per_epoch", ctypes.c_uint32), ("slots_per_sub_window", ctypes.c_uint32),
                ("sub_windows_per_window", ctypes.c_uint32), ("grace_period_slots", ctypes.c_uint32), ("delta", ctypes.c_uint32),
                ("consensus", ConsensusState)]


class StateJobs(ctypes.Structure):
    """mirror of `mina_state_jobs` (include/mina_verify.h); every pointer is a host or a device address depending on the entry point"""
    _fields_ = [
        ("batch", ctypes.c_size_t),
        ("with_states", ctypes.c_int), ("state_records", ctypes.c_void_p), ("state_nfields", ctypes.c_void_p), ("expected_hashes", ctypes.c_void_p),
        ("precheck", ctypes.c_void_p),
        ("log2_domain", ctypes.c_uint32), ("npub", ctypes.c_uint32), ("pub_comm_slot", ctypes.c_uint32), ("public_inputs", ctypes.c_void_p),
        ("with_ipa", ctypes.c_int), ("k", ctypes.c_uint32), ("n_evalpoints", ctypes.c_uint32), ("n_comms", ctypes.c_uint32),
        ("sponge_state", ctypes.c_void_p), ("sponge_pos", ctypes.c_void_p), ("cip", ctypes.c_void_p), ("lr", ctypes.c_void_p), ("delta", ctypes.c_void_p),
        ("sg", ctypes.c_void_p), ("z1", ctypes.c_void_p), ("z2", ctypes.c_void_p), ("evalpoints", ctypes.c_void_p), ("evalscale", ctypes.c_void_p),
        ("polyscale", ctypes.c_void_p), ("comms", ctypes.c_void_p), ("rand_base", ctypes.c_void_p), ("sg_rand_base", ctypes.c_void_p),
        ("kimchi", ctypes.c_void_p),
        ("with_accumulator", ctypes.c_int), ("acc_k", ctypes.c_uint32), ("acc_prechallenges", ctypes.c_void_p), ("acc_sg", ctypes.c_void_p),
        ("acc_rho", ctypes.c_void_p),
    ]

    POINTER_FIELDS = ("state_records", "state_nfields", "expected_hashes", "precheck", "public_inputs", "sponge_state", "sponge_pos", "cip", "lr",
                      "delta", "sg", "z1", "z2", "evalpoints", "evalscale", "polyscale", "comms", "rand_base", "sg_rand_base", "acc_prechallenges",
                      "acc_sg", "acc_rho")


class VerifierIndex(ctypes.Structure):
    _fields_ = [("log2_domain", ctypes.c_uint32), ("zk_rows", ctypes.c_uint32), ("perm_alpha_offset", ctypes.c_uint32), ("shifts", ctypes.c_void_p),
                ("sigma_comm", ctypes.c_void_p), ("coefficients_comm", ctypes.c_void_p), ("selector_comm", ctypes.c_void_p), ("constant_term", ctypes.c_void_p),
                ("constant_term_len", ctypes.c_size_t)]


class KimchiProofs(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_size_t), ("n_prev", ctypes.c_uint32), ("npub", ctypes.c_uint32), ("public_inputs", ctypes.c_void_p), ("prev_chals", ctypes.c_void_p),
                ("prev_comms", ctypes.c_void_p), ("w_comm", ctypes.c_void_p), ("z_comm", ctypes.c_void_p), ("t_comm", ctypes.c_void_p), ("evals", ctypes.c_void_p),
                ("ft_eval1", ctypes.c_void_p), ("statements", ctypes.c_void_p), ("prev_prechallenges", ctypes.c_void_p)]
    POINTER_FIELDS = ("public_inputs", "prev_chals", "prev_comms", "w_comm", "z_comm", "t_comm", "evals", "ft_eval1", "prev_prechallenges")


class PicklesStatements(ctypes.Structure):
    """mina_pickles_statements: structure-of-arrays over the batch (include/mina_verify.h)"""
    POINTER_FIELDS = ("plonk", "bulletproof_challenges", "step_old_challenges", "step_comms", "wrap_old_challenges", "wrap_sg", "sponge_digest", "prev_evals",
                      "prev_public_input", "prev_ft_eval1", "app_state", "misc")
    _fields_ = [("n_old", ctypes.c_uint32), ("n_evals", ctypes.c_uint32)] + [(n, ctypes.c_void_p) for n in POINTER_FIELDS]

    def strides(self):
        return dict(zip(self.POINTER_FIELDS, (64, 256, self.n_old * 256, self.n_old * 64, 480, 64, 32, self.n_evals * 64, 64, 32, 32, 32)))


class KimchiBatchOut(ctypes.Structure):
    _fields_ = [("sponge_state", ctypes.c_void_p), ("sponge_pos", ctypes.c_void_p), ("cip", ctypes.c_void_p), ("evalpoints", ctypes.c_void_p), ("polyscale", ctypes.c_void_p),
                ("evalscale", ctypes.c_void_p), ("comms", ctypes.c_void_p), ("ft_eval0", ctypes.c_void_p), ("malformed", ctypes.c_void_p)]


CHECK_FORMAT, CHECK_LEDGER, CHECK_CHAIN, CHECK_CONSENSUS, CHECK_ACCUMULATOR, CHECK_KIMCHI, CHECK_ACCOUNT_ABI, CHECK_MERKLE = 1, 2, 4, 8, 16, 32, 64, 128
VERIFY_ALLOW_MISSING_KIMCHI, VERIFY_ALLOW_UNBOUND_STATEMENT, VERIFY_ALLOW_SURROGATE = 1, 2, 4


def _bytes_arg(b):
    a = _u8(b) if len(b) else np.zeros(1, np.uint8)
    return a, ctypes.c_size_t(len(b))


def verify_state(proof: bytes, pub: bytes) -> bool:
    """the reference-shaped entry point (Aligned's verify_mina_state_ffi): bincode MinaStateProof + MinaStatePubInputs -> bool"""
    lib = load_library()
    lib.mina_verify_state.restype = ctypes.c_bool
    p, pl = _bytes_arg(proof); q, ql = _bytes_arg(pub)
    return bool(lib.mina_verify_state(_p(p), pl, _p(q), ql))


def verify_state_checks(proof: bytes, pub: bytes):
    lib = load_library()
    p, pl = _bytes_arg(proof); q, ql = _bytes_arg(pub)
    passed, ran = ctypes.c_uint32(0), ctypes.c_uint32(0)
    rc = lib.mina_verify_state_checks(_p(p), pl, _p(q), ql, ctypes.byref(passed), ctypes.byref(ran))
    if rc != 0:
        raise MinaError(f"mina_verify_state_checks failed ({rc}): {lib.mina_last_error().decode()}")
    return passed.value, ran.value


def _ptr_arrays(items):
    n = len(items)
    if all(type(x) is bytes for x in items):               # zero-copy: the pointers are the bytes objects' own buffers (kept alive by `items`)
        return items, (ctypes.c_char_p * n)(*items), (ctypes.c_size_t * n)(*map(len, items))
    arrs = [_u8(x) if len(x) else np.zeros(1, np.uint8) for x in items]
    return arrs, (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs]), (ctypes.c_size_t * n)(*[len(x) for x in items])


def verify_state_batch(proofs: list, pubs: list) -> np.ndarray:
    lib = load_library()
    n = len(proofs)
    pa, PP, PL = _ptr_arrays(proofs); qa, QQ, QL = _ptr_arrays(pubs)
    out = np.zeros(max(n, 1), np.uint8)
    rc = lib.mina_verify_state_batch(ctypes.c_size_t(n), PP, PL, QQ, QL, _p(out))
    if rc != 0:
        raise MinaError(f"mina_verify_state_batch failed ({rc}): {lib.mina_last_error().decode()}")
    return out[:n]


def verify_state_files(proof_path: str, pub_path: str) -> bool:
    lib = load_library()
    lib.mina_verify_state_files.restype = ctypes.c_bool
    return bool(lib.mina_verify_state_files(proof_path.encode(), pub_path.encode()))


def verify_account(proof: bytes, pub: bytes) -> bool:
    lib = load_library()
    lib.mina_verify_account.restype = ctypes.c_bool
    p, pl = _bytes_arg(proof); q, ql = _bytes_arg(pub)
    return bool(lib.mina_verify_account(_p(p), pl, _p(q), ql))


def verify_account_batch(proofs: list, pubs: list) -> np.ndarray:
    lib = load_library()
    n = len(proofs)
    pa, PP, PL = _ptr_arrays(proofs); qa, QQ, QL = _ptr_arrays(pubs)
    out = np.zeros(max(n, 1), np.uint8)
    rc = lib.mina_verify_account_batch(ctypes.c_size_t(n), PP, PL, QQ, QL, _p(out))
    if rc != 0:
        raise MinaError(f"mina_verify_account_batch failed ({rc}): {lib.mina_last_error().decode()}")
    return out[:n]


def verify_account_checks(proof: bytes, pub: bytes):
    lib = load_library()
    p, pl = _bytes_arg(proof); q, ql = _bytes_arg(pub)
    passed, ran = ctypes.c_uint32(0), ctypes.c_uint32(0)
    rc = lib.mina_verify_account_checks(_p(p), pl, _p(q), ql, ctypes.byref(passed), ctypes.byref(ran))
    if rc != 0:
        raise MinaError(f"mina_verify_account_checks failed ({rc}): {lib.mina_last_error().decode()}")
    return passed.value, ran.value


def verify_account_files(proof_path: str, pub_path: str) -> bool:
    lib = load_library()
    lib.mina_verify_account_files.restype = ctypes.c_bool
    return bool(lib.mina_verify_account_files(proof_path.encode(), pub_path.encode()))


def account_abi_encode(account: bytes, encoding: int) -> bytes:
    lib = load_library()
    b, bl = _bytes_arg(account)
    n = ctypes.c_size_t(0)
    rc = lib.mina_account_abi_encode(_p(b), bl, int(encoding), None, ctypes.c_size_t(0), ctypes.byref(n))
    if rc != 0:
        raise MinaError(f"mina_account_abi_encode failed ({rc}): {lib.mina_last_error().decode()}")
    out = np.zeros(n.value, np.uint8)
    rc = lib.mina_account_abi_encode(_p(b), bl, int(encoding), _p(out), ctypes.c_size_t(out.size), ctypes.byref(n))
    if rc != 0:
        raise MinaError(f"mina_account_abi_encode failed ({rc}): {lib.mina_last_error().decode()}")
    return out.tobytes()


def poseidon_params_parse(field: int, text: str) -> np.ndarray:
    """upstream's text form of a Poseidon table (o1js JSON / mina-poseidon Rust source) -> the (9 + 165) x 32-byte layout; host-side"""
    lib = load_library()
    t = text.encode()
    out = np.zeros((9 + 165) * 32, np.uint8)
    rc = lib.mina_poseidon_params_parse(int(field), ctypes.c_char_p(t), ctypes.c_size_t(len(t)), _p(out))
    if rc != 0:
        raise MinaError(f"mina_poseidon_params_parse failed ({rc}): {lib.mina_last_error().decode()}")
    return out


def polish_tokens_from_json(field: int, text: str, enabled_features: int = 0, optional_present: int = 0) -> bytes:
    """serde_json of kimchi's Vec<PolishToken> -> the library's byte-code; host-side"""
    lib = load_library()
    t = text.encode(); n = ctypes.c_size_t(0)
    rc = lib.mina_polish_tokens_from_json(int(field), ctypes.c_char_p(t), ctypes.c_size_t(len(t)), ctypes.c_uint32(enabled_features), ctypes.c_uint32(optional_present), None, ctypes.c_size_t(0), ctypes.byref(n))
    if rc != 0:
        raise MinaError(f"mina_polish_tokens_from_json failed ({rc}): {lib.mina_last_error().decode()}")
    out = np.zeros(max(n.value, 1), np.uint8)
    rc = lib.mina_polish_tokens_from_json(int(field), ctypes.c_char_p(t), ctypes.c_size_t(len(t)), ctypes.c_uint32(enabled_features), ctypes.c_uint32(optional_present), _p(out), ctypes.c_size_t(out.size), ctypes.byref(n))
    if rc != 0:
        raise MinaError(f"mina_polish_tokens_from_json failed ({rc}): {lib.mina_last_error().decode()}")
    return out[: n.value].tobytes()


def verify_configure(flags: int):
    load_library().mina_verify_configure(ctypes.c_uint32(flags))


class Tuning(ctypes.Structure):
    """`mina_verify_tuning` (include/mina_verify.h): every knob of the library, process-wide"""
    FIELDS = ("struct_size", "chunk", "single_max", "slots", "window", "ahead", "early_min", "early_sub", "head_min", "split_max", "chain_cus", "cu_period", "acc_mask",
              "hash_piece_waves", "up_stream", "min_shard", "pace_us", "merge", "merge_batch_max", "linger_us", "max_jobs", "coop16_max", "coop8_max", "coop8_per_call",
              "transcript_coop8_max", "ipa_coop8_max", "kimchi_coop8_max", "bpoly_mfma", "pubcomm_direct", "ipa_shared_points", "kimchi_shared_digest", "ipa_side_stream",
              "search_fan", "search_full", "msm_fp29", "search_ctx", "dev_fork", "dev_chain_cus", "dev_piece_waves", "dev_hash_lds_kb", "dev_acc_lane")
    _fields_ = [(n, ctypes.c_uint32) for n in FIELDS]


def verify_tuning_default() -> Tuning:
    t = Tuning()
    load_library().mina_verify_tuning_default(ctypes.byref(t))
    return t


def verify_tuning_get() -> Tuning:
    t = Tuning()
    load_library().mina_verify_tuning_get(ctypes.byref(t))
    return t


def verify_configure_ex(t: "Tuning | None"):
    lib = load_library()
    rc = lib.mina_verify_configure_ex(ctypes.byref(t) if t is not None else None)
    if rc != 0:
        raise MinaError(f"mina_verify_configure_ex failed ({rc}): {lib.mina_last_error().decode()}")


class tuning:
    """`with m.lib.tuning(chunk=4, single_max=1): ...` -- the library's defaults with the named fields changed for the block (tests force pipeline /
    lane shapes this way: the verdicts must not depend on them).  Nested blocks stack; the previous tuning is restored on exit."""
    def __init__(self, **fields):
        self.fields = fields

    def __enter__(self):
        self.prev = verify_tuning_get()
        t = verify_tuning_get()
        for k, v in self.fields.items():
            if k not in Tuning.FIELDS or k == "struct_size":
                raise KeyError(k)
            setattr(t, k, int(v))
        verify_configure_ex(t)
        return t

    def __exit__(self, *exc):
        verify_configure_ex(self.prev)
        return False


def tune_from_string(spec: str):
    """"chunk=4,slots=16" -> mina_verify_configure_ex with those fields over the defaults (tools/ pass their --tune / $MINA_TUNE strings here; the
    library itself reads no tuning from the environment)"""
    if not spec:
        return
    t = verify_tuning_default()
    for item in spec.split(","):
        k, v = item.split("=")
        if k.strip() not in Tuning.FIELDS:
            raise KeyError(k)
        setattr(t, k.strip(), int(v))
    verify_configure_ex(t)


def verify_set_network(devnet: int):
    rc = load_library().mina_verify_set_network(int(devnet))
    if rc != 0:
        raise MinaError(f"mina_verify_set_network failed ({rc})")


def verify_shutdown():
    load_library().mina_verify_shutdown()


def _borrowed_ctx(lib, h):
    c = MinaContext.__new__(MinaContext)
    c._lib, c._h, c.device, c._borrowed = lib, ctypes.c_void_p(h), 0, True
    return c


def verify_global_ctx():
    """the process-wide context of mina_verify_* (the first device's) as a non-owning MinaContext (e.g. to install a verifier index)"""
    lib = load_library()
    lib.mina_verify_global_ctx.restype = ctypes.c_void_p
    h = lib.mina_verify_global_ctx()
    if not h:
        raise MinaError("no process-wide context: " + lib.mina_last_error().decode())
    return _borrowed_ctx(lib, h)


def verify_device_count() -> int:
    return int(load_library().mina_verify_device_count())


def verify_device_ctx(i: int):
    lib = load_library()
    lib.mina_verify_device_ctx.restype = ctypes.c_void_p
    h = lib.mina_verify_device_ctx(int(i))
    if not h:
        raise MinaError(f"no device context {i}")
    return _borrowed_ctx(lib, h)


class AllDevices:
    """the installers that put the same data on EVERY device of the process (mina_verify_install_*): same method names as MinaContext,
    so the helpers that install an index on a context work on this object too"""
    def __init__(self):
        self._lib = load_library()

    def _ck(self, rc, what):
        if rc != 0:
            raise MinaError(f"{what} failed ({rc}): {self._lib.mina_last_error().decode()}")

    def verifier_index_install(self, *a):
        vi, keep = MinaContext._verifier_index_struct(*a)
        self._ck(self._lib.mina_verify_install_verifier_index(ctypes.byref(vi)), "mina_verify_install_verifier_index")

    def step_index_install(self, *a):
        si, keep = MinaContext._step_index_struct(*a)
        self._ck(self._lib.mina_verify_install_step_index(ctypes.byref(si)), "mina_verify_install_step_index")

    def poseidon_set_params(self, field: int, params):
        a = _u8(params)
        self._ck(self._lib.mina_verify_set_poseidon_params(int(field), _p(a)), "mina_verify_set_poseidon_params")


def verify_all_devices() -> AllDevices:
    return AllDevices()


def poseidon_params_name() -> str:
    lib = load_library()
    lib.mina_poseidon_params_name.restype = ctypes.c_char_p
    return lib.mina_poseidon_params_name().decode()


def wrap_proof_flatten(data: bytes, encoding: int, exact: bool = True):
    lib = load_library()
    b, bl = _bytes_arg(data)
    n = ctypes.c_size_t(0); used = ctypes.c_size_t(0)
    rc = lib.mina_wrap_proof_flatten(_p(b), bl, int(encoding), None, ctypes.c_size_t(0), ctypes.byref(n), None if exact else ctypes.byref(used))
    if rc != 0:
        raise MinaError(f"mina_wrap_proof_flatten failed ({rc}): {lib.mina_last_error().decode()}")
    out = np.zeros(n.value, np.uint8)
    rc = lib.mina_wrap_proof_flatten(_p(b), bl, int(encoding), _p(out), ctypes.c_size_t(out.size), ctypes.byref(n), None if exact else ctypes.byref(used))
    if rc != 0:
        raise MinaError(f"mina_wrap_proof_flatten failed ({rc}): {lib.mina_last_error().decode()}")
    return out.tobytes(), (len(data) if exact else used.value)


def state_proof_split(data: bytes):
    lib = load_library()
    b, bl = _bytes_arg(data)
    pl = ctypes.c_size_t(0); offs = (ctypes.c_size_t * 17)(); lens = (ctypes.c_size_t * 17)()
    rc = lib.mina_state_proof_split(_p(b), bl, ctypes.byref(pl), offs, lens)
    if rc != 0:
        raise MinaError(f"mina_state_proof_split failed ({rc}): {lib.mina_last_error().decode()}")
    return pl.value, list(offs), list(lens)


def protocol_state_pack(data: bytes, encoding: int = ENC_BINPROT, exact: bool = True):
    """serialized protocol state -> (record[64*32] uint8, n_body_fields, info dict, consumed).  No GPU needed."""
    lib = load_library()
    b = _u8(data) if len(data) else np.zeros(1, np.uint8)
    rec = np.zeros(PSTATE_SLOTS * 32, np.uint8); nf = ctypes.c_uint32(0); info = ProtocolStateInfo(); used = ctypes.c_size_t(0)
    rc = lib.mina_protocol_state_pack(_p(b), ctypes.c_size_t(len(data)), int(encoding), _p(rec), ctypes.byref(nf), ctypes.byref(info),
                                      None if exact else ctypes.byref(used))
    if rc != 0:
        raise MinaError(f"mina_protocol_state_pack failed ({rc}): {lib.mina_last_error().decode()}")
    return rec, nf.value, info, (len(data) if exact else used.value)


_lib = None


def load_library():
    """dlopen libminaverify.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MinaError(
            f"{LIB_PATH} is missing: build it with `python -m mina_bridge_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.mina_last_error.restype = ctypes.c_char_p
    lib.mina_ctx_stream.restype = ctypes.c_void_p
    lib.mina_ctx_stream.argtypes = [ctypes.c_void_p]
    lib.mina_srs_depth.restype = ctypes.c_uint32
    lib.mina_ctx_destroy.restype = None
    _lib = lib
    return lib


def _u8(x) -> np.ndarray:
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), dtype=np.uint8).copy()
    return np.ascontiguousarray(x, dtype=np.uint8)


def _p(a):
    if a is None:
        return None
    if isinstance(a, int):
        return ctypes.c_void_p(a)
    return a.ctypes.data_as(ctypes.c_void_p)


def scalar_field_of(curve: int) -> int:
    return FIELD_FQ if curve == CURVE_PALLAS else FIELD_FP


def base_field_of(curve: int) -> int:
    return FIELD_FP if curve == CURVE_PALLAS else FIELD_FQ


class MinaContext:
    """One GPU: stream + SRS tables + workspace.  Mirrors `mina_ctx`."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        h = ctypes.c_void_p()
        rc = self._lib.mina_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise MinaError(f"mina_ctx_create failed ({rc}): {self._lib.mina_last_error().decode()}")
        self._h = h
        self.device = device

    # -- plumbing
    def _ck(self, rc: int, what: str):
        if rc != 0:
            raise MinaError(f"{what} failed ({rc}): {self._lib.mina_last_error().decode()}")

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._lib.mina_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._ck(self._lib.mina_ctx_synchronize(self._h), "mina_ctx_synchronize")

    @property
    def stream(self) -> int:
        return int(self._lib.mina_ctx_stream(self._h) or 0)

    def srs_split_table(self, curve: int, on: bool = True):
        """build / drop the pre-split window table the accumulate kernels read under `tuning(msm_fp29=2)`"""
        self._ck(self._lib.mina_srs_split_table(self._h, int(curve), 1 if on else 0), "mina_srs_split_table")

    def pin_lane(self, lane: int):
        """every `_dev` entry point on ONE lane (negative: round-robin again): the library's work is then ordered on `self.stream`"""
        self._ck(self._lib.mina_ctx_pin_lane(self._h, int(lane)), "mina_ctx_pin_lane")

    def set_pipeline(self, lanes: int):
        self._ck(self._lib.mina_ctx_set_pipeline(self._h, int(lanes)), "mina_ctx_set_pipeline")

    # -- device memory for the `_dev` entry points (no torch needed)
    def dev_malloc(self, nbytes: int) -> int:
        p = ctypes.c_void_p()
        self._ck(self._lib.mina_dev_malloc(self._h, ctypes.c_size_t(nbytes), ctypes.byref(p)), "mina_dev_malloc")
        return p.value

    def dev_free(self, ptr: int):
        self._ck(self._lib.mina_dev_free(self._h, ctypes.c_void_p(ptr)), "mina_dev_free")

    def dev_upload(self, ptr: int, data) -> int:
        a = _u8(data)
        self._ck(self._lib.mina_dev_upload(self._h, ctypes.c_void_p(ptr), _p(a), ctypes.c_size_t(a.size)), "mina_dev_upload")
        return ptr

    def dev_download(self, ptr: int, nbytes: int) -> np.ndarray:
        out = np.empty(nbytes, np.uint8)
        self._ck(self._lib.mina_dev_download(self._h, _p(out), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes)), "mina_dev_download")
        return out

    def prof_enable(self, stage_mask: int = -1):
        self._ck(self._lib.mina_prof_enable(self._h, int(stage_mask)), "mina_prof_enable")

    def prof_read(self) -> dict:
        import json
        buf = ctypes.create_string_buffer(4096)
        self._ck(self._lib.mina_prof_read(self._h, buf, ctypes.c_size_t(4096)), "mina_prof_read")
        return json.loads(buf.value.decode())

    # -- SRS
    def srs_create(self, curve: int, depth: int):
        self._ck(self._lib.mina_srs_create(self._h, curve, ctypes.c_uint32(depth)), "mina_srs_create")

    def srs_load(self, curve: int, blob: bytes):
        b = _u8(blob)
        self._ck(self._lib.mina_srs_load(self._h, curve, _p(b), ctypes.c_size_t(b.size)), "mina_srs_load")

    def srs_depth(self, curve: int) -> int:
        return int(self._lib.mina_srs_depth(self._h, curve))

    def srs_get_g(self, curve: int, first: int, count: int) -> np.ndarray:
        out = np.empty((count, 64), np.uint8)
        self._ck(self._lib.mina_srs_get_g(self._h, curve, ctypes.c_uint32(first), ctypes.c_uint32(count), _p(out)), "mina_srs_get_g")
        return out

    def srs_get_h(self, curve: int) -> np.ndarray:
        out = np.empty(64, np.uint8)
        self._ck(self._lib.mina_srs_get_h(self._h, curve, _p(out)), "mina_srs_get_h")
        return out

    def srs_lagrange_basis(self, curve: int, log2_domain: int) -> np.ndarray:
        n = 1 << log2_domain
        out = np.empty((n, 64), np.uint8)
        self._ck(self._lib.mina_srs_lagrange_basis(self._h, curve, ctypes.c_uint32(log2_domain), _p(out)), "mina_srs_lagrange_basis")
        return out

    def public_input_commitment(self, curve: int, log2_domain: int, public_inputs) -> np.ndarray:
        pub = _u8(public_inputs) if len(public_inputs) else np.zeros(32, np.uint8)
        npub = 0 if not len(public_inputs) else pub.size // 32
        out = np.empty(64, np.uint8)
        self._ck(self._lib.mina_public_input_commitment(self._h, curve, ctypes.c_uint32(log2_domain), ctypes.c_size_t(npub), _p(pub), _p(out)),
                 "mina_public_input_commitment")
        return out

    def public_input_commitment_batch(self, curve: int, log2_domain: int, public_inputs, batch: int) -> np.ndarray:
        """public_inputs: batch x npub x 32 bytes -> batch x 64 (one commitment per proof)"""
        pub = _u8(public_inputs) if batch and len(public_inputs) else np.zeros(32, np.uint8)
        npub = (pub.size // 32) // batch if batch and len(public_inputs) else 0
        out = np.empty((max(batch, 1), 64), np.uint8)
        self._ck(self._lib.mina_public_input_commitment_batch(self._h, curve, ctypes.c_uint32(log2_domain), ctypes.c_size_t(npub), ctypes.c_size_t(batch),
                                                              _p(pub), _p(out)), "mina_public_input_commitment_batch")
        return out[:batch]

    def srs_serialize(self, curve: int) -> bytes:
        depth = self.srs_depth(curve)
        cap = 6 + (depth + 1) * 35
        out = np.empty(cap, np.uint8)
        ln = ctypes.c_size_t(0)
        self._ck(self._lib.mina_srs_serialize(self._h, curve, _p(out), ctypes.c_size_t(cap), ctypes.byref(ln)), "mina_srs_serialize")
        return out[: ln.value].tobytes()

    # -- K1
    def msm(self, curve: int, bases, scalars) -> np.ndarray:
        bases, scalars = _u8(bases), _u8(scalars)
        n = scalars.size // 32
        assert bases.size == n * 64
        out = np.empty(64, np.uint8)
        self._ck(self._lib.mina_msm(self._h, curve, ctypes.c_size_t(n), _p(bases), _p(scalars), _p(out)), "mina_msm")
        return out

    def msm_srs(self, curve: int, scalars) -> np.ndarray:
        scalars = _u8(scalars)
        n = scalars.size // 32
        out = np.empty(64, np.uint8)
        self._ck(self._lib.mina_msm_srs(self._h, curve, ctypes.c_size_t(n), _p(scalars), _p(out)), "mina_msm_srs")
        return out

    def msm_srs_range(self, curve: int, first: int, scalars) -> np.ndarray:
        scalars = _u8(scalars)
        n = scalars.size // 32
        out = np.empty(64, np.uint8)
        self._ck(self._lib.mina_msm_srs_range(self._h, curve, ctypes.c_uint32(first), ctypes.c_size_t(n), _p(scalars), _p(out)), "mina_msm_srs_range")
        return out

    def msm_srs_multi(self, curve: int, scalars, nprob: int) -> np.ndarray:
        """scalars: nprob x n x 32 bytes -> nprob x 64: out[m] = sum_i scalars[m][i] * g[i] in one kernel pipeline"""
        scalars = _u8(scalars)
        n = (scalars.size // 32) // nprob if nprob else 0
        assert nprob == 0 or scalars.size == nprob * n * 32
        out = np.empty((max(nprob, 1), 64), np.uint8)
        self._ck(self._lib.mina_msm_srs_multi(self._h, curve, ctypes.c_size_t(n), ctypes.c_size_t(nprob), _p(scalars), _p(out)), "mina_msm_srs_multi")
        return out[:nprob]

    def msm_srs_dev(self, curve: int, n: int, d_scalars: int, d_out: int):
        self._ck(self._lib.mina_msm_srs_dev(self._h, curve, ctypes.c_size_t(n), ctypes.c_void_p(d_scalars), ctypes.c_void_p(d_out)), "mina_msm_srs_dev")

    # -- K2
    def b_poly(self, field: int, chals, xs) -> np.ndarray:
        chals, xs = _u8(chals), _u8(xs)
        k, npts = chals.size // 32, xs.size // 32
        out = np.empty((npts, 32), np.uint8)
        self._ck(self._lib.mina_b_poly(self._h, field, ctypes.c_uint32(k), _p(chals), ctypes.c_size_t(npts), _p(xs), _p(out)), "mina_b_poly")
        return out

    def b_poly_coefficients(self, field: int, chals) -> np.ndarray:
        chals = _u8(chals)
        k = chals.size // 32
        out = np.empty((1 << k, 32), np.uint8)
        self._ck(self._lib.mina_b_poly_coefficients(self._h, field, ctypes.c_uint32(k), _p(chals), _p(out)), "mina_b_poly_coefficients")
        return out

    def b_poly_fold(self, field: int, k: int, chals, weights=None) -> np.ndarray:
        chals = _u8(chals)
        batch = chals.size // (32 * k)
        w = _u8(weights) if weights is not None else None
        out = np.empty((1 << k, 32), np.uint8)
        self._ck(self._lib.mina_b_poly_fold(self._h, field, ctypes.c_uint32(k), ctypes.c_size_t(batch), _p(chals), _p(w), _p(out)), "mina_b_poly_fold")
        return out

    def b_poly_fold_dev(self, field: int, k: int, batch: int, d_chals: int, d_weights: int, d_out: int):
        self._ck(self._lib.mina_b_poly_fold_dev(self._h, field, ctypes.c_uint32(k), ctypes.c_size_t(batch), ctypes.c_void_p(d_chals),
                                                 ctypes.c_void_p(d_weights) if d_weights else None, ctypes.c_void_p(d_out)), "mina_b_poly_fold_dev")

    # -- K3
    def poseidon_set_params(self, field: int, params):
        params = _u8(params)
        assert params.size == (9 + 165) * 32
        self._ck(self._lib.mina_poseidon_set_params(self._h, field, _p(params)), "mina_poseidon_set_params")

    def poseidon_permute(self, field: int, states) -> np.ndarray:
        st = _u8(states).copy()
        n = st.size // 96
        self._ck(self._lib.mina_poseidon_permute(self._h, field, ctypes.c_size_t(n), _p(st)), "mina_poseidon_permute")
        return st.reshape(n, 96)

    def poseidon_permute_dev(self, field: int, n: int, d_states: int):
        self._ck(self._lib.mina_poseidon_permute_dev(self._h, field, ctypes.c_size_t(n), ctypes.c_void_p(d_states)), "mina_poseidon_permute_dev")

    def poseidon_hash(self, field: int, inputs, n: int, length: int) -> np.ndarray:
        inputs = _u8(inputs) if length else np.zeros(32, np.uint8)
        out = np.empty((n, 32), np.uint8)
        self._ck(self._lib.mina_poseidon_hash(self._h, field, ctypes.c_size_t(n), ctypes.c_size_t(length), _p(inputs), _p(out)), "mina_poseidon_hash")
        return out

    def challenge_to_field(self, field: int, chal128) -> np.ndarray:
        ch = _u8(chal128)
        n = ch.size // 16
        out = np.empty((n, 32), np.uint8)
        self._ck(self._lib.mina_challenge_to_field(self._h, field, ctypes.c_size_t(n), _p(ch), _p(out)), "mina_challenge_to_field")
        return out

    def fq_sponge_run(self, curve: int, batch: int, tape: bytes, inputs, init=None, want_final=False):
        """init: None or (states[batch,96] uint8, pos[batch,2] uint32).  Returns outputs[batch, n_squeeze, 32] (and final state/pos)."""
        tp = _u8(tape)
        n_out = sum(1 for t in tape if t >= 3)
        inp = _u8(inputs) if len(inputs) else np.zeros(32, np.uint8)
        out = np.zeros((batch, max(n_out, 1), 32), np.uint8)
        st = pos = None
        if init is not None:
            st, pos = _u8(init[0]), np.ascontiguousarray(init[1], dtype=np.uint32)
        fs = np.zeros((batch, 96), np.uint8) if want_final else None
        fp = np.zeros((batch, 2), np.uint32) if want_final else None
        self._ck(self._lib.mina_fq_sponge_run(self._h, curve, ctypes.c_size_t(batch), _p(tp), ctypes.c_size_t(len(tape)), _p(st),
                                              pos.ctypes.data_as(ctypes.c_void_p) if pos is not None else None, _p(inp), _p(out), _p(fs),
                                              fp.ctypes.data_as(ctypes.c_void_p) if fp is not None else None), "mina_fq_sponge_run")
        out = out[:, :n_out]
        return (out, fs, fp) if want_final else out

    # -- a16
    def merkle_roots(self, field: int, leaves, siblings, dirs, depth: int) -> np.ndarray:
        leaves, siblings, dirs = _u8(leaves), _u8(siblings), _u8(dirs)
        n = leaves.size // 32
        assert siblings.size == n * depth * 32 and dirs.size == n * depth
        out = np.empty((n, 32), np.uint8)
        self._ck(self._lib.mina_merkle_roots(self._h, field, ctypes.c_size_t(n), ctypes.c_uint32(depth), _p(leaves), _p(siblings), _p(dirs), _p(out)),
                 "mina_merkle_roots")
        return out

    def merkle_verify_batch(self, field: int, leaves, siblings, dirs, depth: int, expected_roots) -> np.ndarray:
        leaves, siblings, dirs, exp = _u8(leaves), _u8(siblings), _u8(dirs), _u8(expected_roots)
        n = leaves.size // 32
        out = np.empty(n, np.uint8)
        self._ck(self._lib.mina_merkle_verify_batch(self._h, field, ctypes.c_size_t(n), ctypes.c_uint32(depth), _p(leaves), _p(siblings), _p(dirs),
                                                    _p(exp), _p(out)), "mina_merkle_verify_batch")
        return out

    def verify_account_inclusion(self, proofs: list, pub_inputs: list, leaf_hashes) -> np.ndarray:
        n = len(proofs)
        pa = [_u8(p) if len(p) else np.zeros(1, np.uint8) for p in proofs]
        qa = [_u8(p) if len(p) else np.zeros(1, np.uint8) for p in pub_inputs]
        PP = (ctypes.c_void_p * n)(*[a.ctypes.data for a in pa]); QQ = (ctypes.c_void_p * n)(*[a.ctypes.data for a in qa])
        PL = (ctypes.c_size_t * n)(*[len(p) for p in proofs]); QL = (ctypes.c_size_t * n)(*[len(p) for p in pub_inputs])
        lh = _u8(leaf_hashes); out = np.empty(n, np.uint8)
        self._ck(self._lib.mina_verify_account_inclusion(self._h, ctypes.c_size_t(n), PP, PL, QQ, QL, _p(lh), _p(out)), "mina_verify_account_inclusion")
        return out

    # -- K4
    def to_group(self, curve: int, t) -> np.ndarray:
        t = _u8(t)
        n = t.size // 32
        out = np.empty((n, 64), np.uint8)
        self._ck(self._lib.mina_to_group(self._h, curve, ctypes.c_size_t(n), _p(t), _p(out)), "mina_to_group")
        return out

    # -- field hooks
    def field_mul(self, field: int, a, b) -> np.ndarray:
        a, b = _u8(a), _u8(b)
        n = a.size // 32
        out = np.empty((n, 32), np.uint8)
        self._ck(self._lib.mina_field_mul(self._h, field, ctypes.c_size_t(n), _p(a), _p(b), _p(out)), "mina_field_mul")
        return out

    def field_inv(self, field: int, a) -> np.ndarray:
        a = _u8(a)
        n = a.size // 32
        out = np.empty((n, 32), np.uint8)
        self._ck(self._lib.mina_field_inv(self._h, field, ctypes.c_size_t(n), _p(a), _p(out)), "mina_field_inv")
        return out

    def field_sqrt(self, field: int, a):
        a = _u8(a)
        n = a.size // 32
        out = np.empty((n, 32), np.uint8)
        ok = np.empty(n, np.uint8)
        self._ck(self._lib.mina_field_sqrt(self._h, field, ctypes.c_size_t(n), _p(a), _p(out), _p(ok)), "mina_field_sqrt")
        return out, ok

    def selftest_group_law(self, curve: int, p, q):
        p, q = _u8(p), _u8(q)
        n = p.size // 64
        a, b, same = np.empty((n, 64), np.uint8), np.empty((n, 64), np.uint8), np.empty(n, np.uint8)
        self._ck(self._lib.mina_selftest_group_law(self._h, curve, ctypes.c_size_t(n), _p(p), _p(q), _p(a), _p(b), _p(same)), "mina_selftest_group_law")
        return a, b, same

    # -- a10 / a8
    def accumulator_check_batch(self, curve: int, k: int, prechallenges, sg, rho=None) -> np.ndarray:
        pre, sg = _u8(prechallenges), _u8(sg)
        batch = sg.size // 64
        assert pre.size == batch * k * 16
        r = _u8(rho) if rho is not None else None
        out = np.empty(batch, np.uint8)
        self._ck(self._lib.mina_accumulator_check_batch(self._h, curve, ctypes.c_uint32(k), ctypes.c_size_t(batch), _p(pre), _p(sg), _p(r), _p(out)),
                 "mina_accumulator_check_batch")
        return out

    def accumulator_check_dev(self, curve: int, k: int, batch: int, d_pre: int, d_sg: int, d_rho: int, d_verdict: int):
        self._ck(self._lib.mina_accumulator_check_dev(self._h, curve, ctypes.c_uint32(k), ctypes.c_size_t(batch), ctypes.c_void_p(d_pre),
                                                       ctypes.c_void_p(d_sg), ctypes.c_void_p(d_rho) if d_rho else None, ctypes.c_void_p(d_verdict)),
                 "mina_accumulator_check_dev")

    def accumulator_check_multi(self, curve: int, k: int, prechallenges, sg) -> np.ndarray:
        """un-folded: one deterministic verdict per proof (groups of 16 checks per kernel pipeline)"""
        pre, sg = _u8(prechallenges), _u8(sg)
        count = sg.size // 64
        assert pre.size == count * k * 16
        out = np.zeros(count, np.uint8)
        self._ck(self._lib.mina_accumulator_check_multi(self._h, curve, ctypes.c_uint32(k), ctypes.c_size_t(count), _p(pre), _p(sg), _p(out)),
                 "mina_accumulator_check_multi")
        return out

    def accumulator_check_multi_dev(self, curve: int, k: int, count: int, d_pre: int, d_sg: int, d_verdicts: int):
        """`count` independent checks in one kernel pipeline; d_verdicts: count u32 words"""
        self._ck(self._lib.mina_accumulator_check_multi_dev(self._h, curve, ctypes.c_uint32(k), ctypes.c_size_t(count), ctypes.c_void_p(d_pre),
                                                            ctypes.c_void_p(d_sg), ctypes.c_void_p(d_verdicts)), "mina_accumulator_check_multi_dev")

    @staticmethod
    def pack_ipa_openings(openings: list):
        """list of dicts with the fields of `mina_ipa_opening` (numpy uint8 arrays) -> (ctypes array, keep-alive list).
        Packing is pure host work; callers that verify the same batch repeatedly (timing tools) do it once."""
        keep = []
        arr = (IpaOpening * len(openings))()

        def ptr(x):
            a = _u8(x)
            keep.append(a)
            return a.ctypes.data

        for i, o in enumerate(openings):
            e = arr[i]
            e.k = int(o["k"])
            e.lr, e.delta, e.sg = ptr(o["lr"]), ptr(o["delta"]), ptr(o["sg"])
            e.z1, e.z2 = ptr(o["z1"]), ptr(o["z2"])
            e.n_evalpoints = int(o["n_evalpoints"])
            e.evalpoints = ptr(o["evalpoints"])
            e.n_comms = int(o["n_comms"])
            e.comms = ptr(o["comms"])
            e.combined_inner_product = ptr(o["combined_inner_product"])
            e.polyscale, e.evalscale = ptr(o["polyscale"]), ptr(o["evalscale"])
            e.sponge_state = ptr(o["sponge_state"])
            e.sponge_mode, e.sponge_count = int(o["sponge_mode"]), int(o["sponge_count"])
        return arr, keep

    def ipa_batch_check(self, curve: int, openings, rand_base, sg_rand_base) -> bool:
        """`openings`: list of dicts with the fields of `mina_ipa_opening` (numpy uint8 arrays), or the result of
        `pack_ipa_openings`."""
        arr, keep = openings if isinstance(openings, tuple) else self.pack_ipa_openings(openings)
        rb, sb = _u8(rand_base), _u8(sg_rand_base)
        v = np.zeros(1, np.uint8)
        self._ck(self._lib.mina_ipa_batch_check(self._h, curve, ctypes.c_size_t(len(arr)), arr, _p(rb), _p(sb), _p(v)), "mina_ipa_batch_check")
        return bool(v[0])

    # -- protocol states / the Proof-of-State job
    def protocol_state_hash_batch(self, records, nfields, want_body: bool = False):
        rec = _u8(records); nf = np.ascontiguousarray(nfields, dtype=np.uint32)
        n = nf.size
        assert rec.size == n * PSTATE_SLOTS * 32
        out = np.empty((n, 32), np.uint8); body = np.empty((n, 32), np.uint8) if want_body else None
        self._ck(self._lib.mina_protocol_state_hash_batch(self._h, ctypes.c_size_t(n), _p(rec), nf.ctypes.data_as(ctypes.c_void_p), _p(out), _p(body)),
                 "mina_protocol_state_hash_batch")
        return (out, body) if want_body else out

    def protocol_state_hash_bytes(self, states: list, encoding: int = ENC_BINPROT) -> np.ndarray:
        n = len(states)
        arrs = [_u8(s) for s in states]
        PP = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs]); PL = (ctypes.c_size_t * n)(*[len(s) for s in states])
        out = np.empty((n, 32), np.uint8)
        self._ck(self._lib.mina_protocol_state_hash_bytes(self._h, int(encoding), ctypes.c_size_t(n), PP, PL, _p(out)), "mina_protocol_state_hash_bytes")
        return out

    def account_hash_batch(self, accounts: list, encoding: int = ENC_BINPROT) -> np.ndarray:
        n = len(accounts)
        arrs, PP, PL = _ptr_arrays(accounts)
        out = np.empty((n, 32), np.uint8)
        self._ck(self._lib.mina_account_hash_batch(self._h, int(encoding), ctypes.c_size_t(n), PP, PL, _p(out)), "mina_account_hash_batch")
        return out

    def verify_account_checks(self, proofs: list, pubs: list):
        n = len(proofs)
        pa, PP, PL = _ptr_arrays(proofs); qa, QQ, QL = _ptr_arrays(pubs)
        passed = np.zeros(n, np.uint32); ran = np.zeros(n, np.uint32)
        self._ck(self._lib.mina_verify_account_ctx(self._h, ctypes.c_size_t(n), PP, PL, QQ, QL, passed.ctypes.data_as(ctypes.c_void_p), ran.ctypes.data_as(ctypes.c_void_p)),
                 "mina_verify_account_ctx")
        return passed, ran

    # -- multi-GPU building blocks (device pointers)
    def challenge_to_field_dev(self, field: int, n: int, d_chal: int, d_out: int):
        self._ck(self._lib.mina_challenge_to_field_dev(self._h, field, ctypes.c_size_t(n), ctypes.c_void_p(d_chal), ctypes.c_void_p(d_out)), "mina_challenge_to_field_dev")

    def field_sum_rows_dev(self, field: int, rows: int, m: int, d_in: int, d_out: int):
        self._ck(self._lib.mina_field_sum_rows_dev(self._h, field, ctypes.c_size_t(rows), ctypes.c_size_t(m), ctypes.c_void_p(d_in), ctypes.c_void_p(d_out)), "mina_field_sum_rows_dev")

    def msm_srs_range_dev(self, curve: int, first: int, n: int, d_scalars: int, d_out: int):
        self._ck(self._lib.mina_msm_srs_range_dev(self._h, curve, ctypes.c_uint32(first), ctypes.c_size_t(n), ctypes.c_void_p(d_scalars), ctypes.c_void_p(d_out)), "mina_msm_srs_range_dev")

    def msm_dev(self, curve: int, n: int, d_bases: int, d_scalars: int, d_out: int):
        self._ck(self._lib.mina_msm_dev(self._h, curve, ctypes.c_size_t(n), ctypes.c_void_p(d_bases), ctypes.c_void_p(d_scalars), ctypes.c_void_p(d_out)), "mina_msm_dev")

    def points_sum_dev(self, curve: int, n: int, d_records: int, d_out: int):
        self._ck(self._lib.mina_points_sum_dev(self._h, curve, ctypes.c_size_t(n), ctypes.c_void_p(d_records), ctypes.c_void_p(d_out)), "mina_points_sum_dev")

    def point_records_equal_dev(self, d_a: int, d_b: int, d_verdict: int):
        self._ck(self._lib.mina_point_records_equal_dev(self._h, ctypes.c_void_p(d_a), ctypes.c_void_p(d_b), ctypes.c_void_p(d_verdict)), "mina_point_records_equal_dev")

    def state_jobs_prepare(self, log2_domain: int, npub: int):
        self._ck(self._lib.mina_state_jobs_prepare(self._h, ctypes.c_uint32(log2_domain), ctypes.c_uint32(npub)), "mina_state_jobs_prepare")

    @staticmethod
    def make_state_jobs(batch: int, arrays: dict, **scalars):
        """host-side `mina_state_jobs`: arrays maps pointer-field name -> numpy array (kept alive by the returned list)"""
        j = StateJobs(); keep = []
        j.batch = batch
        kimchi = scalars.pop("kimchi", None)
        if kimchi is not None:                      # (KimchiProofs, keep-alive list) from make_kimchi_proofs
            j.kimchi = ctypes.addressof(kimchi[0]); keep.append(kimchi)
        for name, val in scalars.items():
            setattr(j, name, val)
        for name in StateJobs.POINTER_FIELDS:
            a = arrays.get(name)
            if a is None:
                continue
            a = np.ascontiguousarray(a); keep.append(a)
            setattr(j, name, a.ctypes.data)
        return j, keep

    @staticmethod
    def make_kimchi_proofs(batch: int, n_prev: int, npub: int, arrays: dict, statements=None):
        """statements: (PicklesStatements, keep) from make_pickles_statements -- the public inputs are then derived on the GPU"""
        kp = KimchiProofs(); keep = []
        kp.batch, kp.n_prev, kp.npub = batch, n_prev, npub
        for name in KimchiProofs.POINTER_FIELDS:
            a = arrays.get(name)
            if a is None:
                continue
            a = np.ascontiguousarray(a); keep.append(a)
            setattr(kp, name, a.ctypes.data)
        if statements is not None:
            st, skeep = statements
            kp.statements = ctypes.addressof(st); keep.append((st, skeep))
        return kp, keep

    @staticmethod
    def make_pickles_statements(n_old: int, n_evals: int, arrays: dict):
        """arrays: section name -> host uint8 array or an int device address"""
        st = PicklesStatements(); keep = []
        st.n_old, st.n_evals = n_old, n_evals
        for name in PicklesStatements.POINTER_FIELDS:
            a = arrays.get(name)
            if a is None:
                continue
            if isinstance(a, int):
                setattr(st, name, a); continue
            a = np.ascontiguousarray(a); keep.append(a)
            setattr(st, name, a.ctypes.data)
        return st, keep

    def pickles_public_inputs_batch(self, statements, batch: int):
        """statements -> (public inputs [batch, 40, 32], ok [batch]) through the GPU kernels (mina_pickles_public_inputs_batch)"""
        st, _keep = statements
        pub = np.zeros((batch, 40, 32), np.uint8); ok = np.zeros(batch, np.uint8)
        self._ck(self._lib.mina_pickles_public_inputs_batch(self._h, ctypes.byref(st), ctypes.c_size_t(batch), _p(pub), _p(ok)), "mina_pickles_public_inputs_batch")
        return pub, ok

    @staticmethod
    def _verifier_index_struct(log2_domain: int, zk_rows: int, perm_alpha_offset: int, shifts, sigma_comm, coefficients_comm, selector_comm, constant_term: bytes):
        vi = VerifierIndex(); arrs = [_u8(shifts), _u8(sigma_comm), _u8(coefficients_comm), _u8(selector_comm), _u8(constant_term) if len(constant_term) else np.zeros(1, np.uint8)]
        vi.log2_domain, vi.zk_rows, vi.perm_alpha_offset = log2_domain, zk_rows, perm_alpha_offset
        vi.shifts, vi.sigma_comm, vi.coefficients_comm, vi.selector_comm, vi.constant_term = (a.ctypes.data for a in arrs)
        vi.constant_term_len = len(constant_term)
        return vi, arrs

    def verifier_index_install(self, *a):
        vi, keep = self._verifier_index_struct(*a)
        self._ck(self._lib.mina_verifier_index_install(self._h, ctypes.byref(vi)), "mina_verifier_index_install")

    @staticmethod
    def _step_index_struct(zk_rows: int, domains: list, shifts, constant_term: bytes):
        """domains: list of log2 sizes; shifts: len(domains) x 7 x 32 bytes (Fp)"""
        class StepIndex(ctypes.Structure):
            _fields_ = [("zk_rows", ctypes.c_uint32), ("n_domains", ctypes.c_uint32), ("domain_log2", ctypes.c_void_p), ("shifts", ctypes.c_void_p),
                        ("constant_term", ctypes.c_void_p), ("constant_term_len", ctypes.c_size_t)]
        d = np.ascontiguousarray(domains, dtype=np.uint32); sh = _u8(shifts); ct = _u8(constant_term) if len(constant_term) else np.zeros(1, np.uint8)
        return StepIndex(zk_rows, len(domains), d.ctypes.data, sh.ctypes.data, ct.ctypes.data, len(constant_term)), (d, sh, ct)

    def step_index_install(self, *a):
        si, keep = self._step_index_struct(*a)
        self._ck(self._lib.mina_step_index_install(self._h, ctypes.byref(si)), "mina_step_index_install")

    def verifier_index_load_json(self, index_json: str, constant_term_json: str, perm_alpha_offset: int = 21):
        a, b = index_json.encode(), constant_term_json.encode()
        self._ck(self._lib.mina_verifier_index_load_json(self._h, ctypes.c_char_p(a), ctypes.c_size_t(len(a)), ctypes.c_char_p(b), ctypes.c_size_t(len(b)), ctypes.c_uint32(perm_alpha_offset)),
                 "mina_verifier_index_load_json")

    def step_index_load_json(self, index_jsons: list, constant_term_json: str, enabled_features: int = 0, optional_present: int = 0):
        enc = [j.encode() for j in index_jsons]; ct = constant_term_json.encode()
        arr = (ctypes.c_char_p * len(enc))(*enc); lens = (ctypes.c_size_t * len(enc))(*map(len, enc))
        self._ck(self._lib.mina_step_index_load_json(self._h, ctypes.c_size_t(len(enc)), arr, lens, ctypes.c_char_p(ct), ctypes.c_size_t(len(ct)), ctypes.c_uint32(enabled_features),
                                                     ctypes.c_uint32(optional_present)), "mina_step_index_load_json")

    def poseidon_load_params(self, field: int, text: str):
        t = text.encode()
        self._ck(self._lib.mina_poseidon_load_params(self._h, int(field), ctypes.c_char_p(t), ctypes.c_size_t(len(t))), "mina_poseidon_load_params")

    def pickles_public_input(self, wrap_proof: bytes, encoding: int, app_state):
        b = _u8(wrap_proof); a = _u8(app_state)
        pub = np.zeros((40, 32), np.uint8); der = np.zeros((7, 32), np.uint8)
        self._ck(self._lib.mina_pickles_public_input(self._h, _p(b), ctypes.c_size_t(len(wrap_proof)), int(encoding), _p(a), _p(pub), _p(der)), "mina_pickles_public_input")
        return pub, der

    def verifier_index_digest(self) -> np.ndarray:
        out = np.empty(32, np.uint8)
        self._ck(self._lib.mina_verifier_index_digest(self._h, _p(out)), "mina_verifier_index_digest")
        return out

    def kimchi_to_batch(self, kimchi, k: int) -> dict:
        kp, keep = kimchi
        B, nc = kp.batch, kp.n_prev + 45
        o = {"sponge_state": np.zeros((B, 96), np.uint8), "sponge_pos": np.zeros((B, 2), np.uint32), "cip": np.zeros((B, 32), np.uint8),
             "evalpoints": np.zeros((B, 64), np.uint8), "polyscale": np.zeros((B, 32), np.uint8), "evalscale": np.zeros((B, 32), np.uint8),
             "comms": np.zeros((B, nc, 64), np.uint8), "ft_eval0": np.zeros((B, 32), np.uint8), "malformed": np.zeros(1, np.uint8)}
        out = KimchiBatchOut()
        for name in o:
            setattr(out, name, o[name].ctypes.data)
        self._ck(self._lib.mina_kimchi_to_batch(self._h, ctypes.byref(kp), ctypes.byref(out)), "mina_kimchi_to_batch")
        return o

    def state_job_batch(self, jobs) -> np.ndarray:
        j, keep = jobs
        out = np.zeros(j.batch, np.uint8)
        self._ck(self._lib.mina_state_job_batch(self._h, ctypes.byref(j), _p(out)), "mina_state_job_batch")
        return out

    def state_jobs_to_device(self, jobs):
        """upload every section of a host-side StateJobs; returns (device StateJobs, [device pointers to free])"""
        j, keep = jobs
        d = StateJobs(); ptrs = []
        ctypes.memmove(ctypes.byref(d), ctypes.byref(j), ctypes.sizeof(StateJobs))
        flat = []
        for x in keep:
            flat.extend(x[1] if isinstance(x, tuple) else [x])
        by_addr = {a.ctypes.data: a for a in flat}

        def up(addr):
            a = by_addr[addr]
            p = self.dev_malloc(max(a.nbytes, 4)); ptrs.append(p)
            self.dev_upload(p, a.view(np.uint8).reshape(-1))
            return p

        for name in StateJobs.POINTER_FIELDS:
            addr = getattr(j, name)
            if addr:
                setattr(d, name, up(addr))
        if j.kimchi:
            src = ctypes.cast(j.kimchi, ctypes.POINTER(KimchiProofs)).contents
            dk = KimchiProofs()
            ctypes.memmove(ctypes.byref(dk), ctypes.byref(src), ctypes.sizeof(KimchiProofs))
            for name in KimchiProofs.POINTER_FIELDS:
                addr = getattr(src, name)
                if addr:
                    setattr(dk, name, d.public_inputs if name == "public_inputs" and j.public_inputs == addr else up(addr))
            d.kimchi = ctypes.addressof(dk)
            self._keep_dk = dk
        return d, ptrs

    def state_job_fold_dev(self, d_jobs, d_verdicts: int, d_flags: int, d_ipa_scalars: int, d_ipa_point: int, d_acc_scalars: int, d_acc_point: int):
        """SURVEY.md 8e.2 for the whole job: every stage but the two fixed-base MSMs; the shard's folded scalar vectors and variable-base partials go to the caller"""
        self._ck(self._lib.mina_state_job_fold_dev(self._h, ctypes.byref(d_jobs), ctypes.c_void_p(d_verdicts), ctypes.c_void_p(d_flags) if d_flags else None,
                                                   ctypes.c_void_p(d_ipa_scalars), ctypes.c_void_p(d_ipa_point), ctypes.c_void_p(d_acc_scalars), ctypes.c_void_p(d_acc_point)),
                 "mina_state_job_fold_dev")

    def state_job_batch_dev(self, d_jobs, d_verdicts: int, d_flags: int = 0):
        self._ck(self._lib.mina_state_job_batch_dev(self._h, ctypes.byref(d_jobs), ctypes.c_void_p(d_verdicts), ctypes.c_void_p(d_flags) if d_flags else None),
                 "mina_state_job_batch_dev")
