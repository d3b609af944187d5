"""Container readers (8f-2, 8f-3) -- CPU tests: wrap proof (bin_prot + bincode) -> kernel-ready flat form,
MinaStateProof split, account -> Solidity ABI bytes; round trips against independent Python writers and truncation fuzz."""
import random

import numpy as np
import pytest


def test_wrap_proof_roundtrip_both_encodings():
    import mina_bridge_amd as m
    from wire_writers import synth_wrap_proof, wrap_proof_bytes, wrap_proof_flat
    rng = random.Random(4)
    for i in range(6):
        d = synth_wrap_proof(rng, k=rng.choice([15, 15, 6, 1]))
        want = wrap_proof_flat(d)
        for enc, bp in ((m.lib.ENC_BINPROT, True), (m.lib.ENC_BINCODE, False)):
            raw = wrap_proof_bytes(d, bp)
            got, used = m.lib.wrap_proof_flatten(raw, enc)
            assert used == len(raw) and got == want, (i, enc)
            got2, used2 = m.lib.wrap_proof_flatten(raw + b"tail", enc, exact=False)
            assert used2 == len(raw) and got2 == want
            with pytest.raises(m.MinaError):
                m.lib.wrap_proof_flatten(raw + b"\0", enc)
    raw = wrap_proof_bytes(d, True)
    assert 9000 < len(raw) < 16000                              # the proof is ~ 11 KB of a ~ 37 KB container
    for cut in list(range(0, len(raw), 131)) + [len(raw) - 1]:
        with pytest.raises(m.MinaError):
            m.lib.wrap_proof_flatten(raw[:cut], m.lib.ENC_BINPROT)
    for _ in range(300):                                        # single-bit corruption: parses or is rejected, never crashes
        b = bytearray(raw); b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        try:
            m.lib.wrap_proof_flatten(bytes(b), m.lib.ENC_BINPROT)
        except m.MinaError:
            pass


def test_state_proof_container_split():
    import mina_bridge_amd as m
    from oracle import mina_state_ref as S, state_job_ref as J
    from test_protocol_state import bincode_state
    from wire_writers import state_proof_bytes, synth_wrap_proof, wrap_proof_bytes
    rng = random.Random(8)
    wrap = synth_wrap_proof(rng)
    states = [J.synth_state(rng, rng.randrange(S.P), i) for i in range(17)]
    raw = state_proof_bytes(wrap, states)
    assert 30000 < len(raw) < 48 * 1024                         # within the FFI's 48 KiB proof buffer
    plen, offs, lens = m.lib.state_proof_split(raw)
    assert plen == len(wrap_proof_bytes(wrap, False)) and offs[0] == plen
    for i in range(17):
        blob = raw[offs[i]: offs[i] + lens[i]]
        assert blob == bincode_state(states[i])
        rec, nf, info, _ = m.lib.protocol_state_pack(blob, m.lib.ENC_BINCODE)
        assert bytes(info.previous_state_hash) == states[i]["previous_state_hash"].to_bytes(32, "little")
    for cut in (0, 100, plen - 1, plen + 5, len(raw) - 1):
        with pytest.raises(m.MinaError):
            m.lib.state_proof_split(raw[:cut])
    with pytest.raises(m.MinaError):
        m.lib.state_proof_split(raw + b"\0")


def test_account_abi_encoding_matches_reference_rules():
    """encoded_account re-derived by the library == the encoding rules of core/src/sol/account.rs:25-314 applied independently"""
    import mina_bridge_amd as m
    from oracle import mina_account_ref as A
    rng = random.Random(12)
    for zk, timed, deleg, vk in [(False, False, False, True), (True, True, True, True), (True, False, True, False), (False, True, False, True), (True, True, False, True)]:
        for _ in range(3):
            a = A.synth_account(rng, zk, timed, deleg, with_vk=vk)
            want = A.abi_encode_account(a)
            for enc, bp in ((m.lib.ENC_BINPROT, True), (m.lib.ENC_BINCODE, False)):
                assert m.lib.account_abi_encode(A.write_account(a, bp), enc) == want
    # layout facts the Solidity side relies on (MinaAccountValidationExample.sol:70, SudokuValidity.sol:47,67)
    a = A.synth_account(rng, True, True, True)
    enc = A.abi_encode_account(a)
    assert int.from_bytes(enc[:32], "big") == 32 and enc[32:64] == a["public_key"]["x"].to_bytes(32, "little")
    assert int.from_bytes(enc[32 + 3 * 32: 32 + 4 * 32], "big") == 30 * 32                 # tokenSymbol offset = size of the static head
    sym_off = 32 + 30 * 32
    assert int.from_bytes(enc[sym_off: sym_off + 32], "big") == len(a["token_symbol"])
    raw = A.write_account(a, False)
    for cut in range(0, len(raw), 53):
        with pytest.raises(m.MinaError):
            m.lib.account_abi_encode(raw[:cut], m.lib.ENC_BINCODE)
    bad = dict(a); bad["token_symbol"] = b"TOOLONG"
    with pytest.raises(m.MinaError):
        m.lib.account_abi_encode(A.write_account(bad, False), m.lib.ENC_BINCODE)
