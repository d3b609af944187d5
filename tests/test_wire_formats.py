"""Wire formats of the reference and the Proof-of-Account inclusion check (a16).
The byte layouts come from the serde derives in core/src/proof/{state_proof,account_proof}.rs + sol/serialization.rs
(bincode 1.3) and are double-pinned by the Solidity offsets (MinaStateSettlementExample.sol:92,100,130-135,
MinaAccountValidationExample.sol:70).  The writers below are an independent Python rendering of those derives."""
import struct

import numpy as np
import pytest

from conftest import rand_scalars

P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001


def fp_bytes(seed, n=1):
    return [r.tobytes() for r in rand_scalars(n, P, seed=seed)]


def ser_state_pub_inputs(devnet, tip, state_hashes, ledger_hashes):
    # bool, then SolSerialize'd [u8; 32] values: bincode writes fixed-size arrays without a length prefix
    return bytes([1 if devnet else 0]) + tip + b"".join(state_hashes) + b"".join(ledger_hashes)


def ser_account_pub_inputs(ledger_hash, encoded_account):
    return ledger_hash + struct.pack("<Q", len(encoded_account)) + encoded_account


def ser_account_proof(path, account_tail=b"\x07account-binprot-bytes"):
    # Vec<MerkleNode>: u64 count; enum variant as u32; SerdeAs<Fp> = byte string (u64 length + 32 bytes); then the account
    out = struct.pack("<Q", len(path))
    for d, h in path:
        out += struct.pack("<I", d) + struct.pack("<Q", 32) + h
    return out + account_tail


def test_state_pub_inputs_layout():
    import mina_bridge_amd as m
    hs = fp_bytes(1, 33)
    data = ser_state_pub_inputs(True, hs[0], hs[1:17], hs[17:33])
    assert len(data) == 1057
    got = m.parse_state_pub_inputs(data)
    assert got["is_state_proof_from_devnet"] is True and got["bridge_tip_state_hash"] == hs[0]
    assert got["candidate_chain_state_hashes"] == hs[1:17] and got["candidate_chain_ledger_hashes"] == hs[17:33]
    # Solidity-side offsets: tip hash at byte 1, state hashes from byte 33, ledger hashes from byte 545
    assert data[1:33] == hs[0] and data[33:65] == hs[1] and data[545:577] == hs[17]
    for bad in (data[:-1], data + b"\x00", b"\x02" + data[1:], data[:1] + b"\xff" * 32 + data[33:]):
        with pytest.raises(m.MinaError):
            m.parse_state_pub_inputs(bad)


def test_account_pub_inputs_and_merkle_path_layout():
    import mina_bridge_amd as m
    lh = fp_bytes(2)[0]
    enc = bytes(range(200))
    data = ser_account_pub_inputs(lh, enc)
    got_lh, got_enc = m.parse_account_pub_inputs(data)
    assert got_lh == lh and got_enc == enc and data[40:] == enc   # `pubInput[32+8:]` on the Solidity side
    for bad in (data[:-1], data[:39], lh + struct.pack("<Q", 5) + enc):
        with pytest.raises(m.MinaError):
            m.parse_account_pub_inputs(bad)
    sibs = fp_bytes(3, 35)
    path = [(i & 1, sibs[i]) for i in range(35)]
    proof = ser_account_proof(path)
    s, d, off = m.parse_merkle_path(proof)
    assert [x.tobytes() for x in s] == sibs and d.tolist() == [i & 1 for i in range(35)]
    assert off == 8 + 35 * 44 and proof[off:] == b"\x07account-binprot-bytes"
    assert m.parse_merkle_path(ser_account_proof([]))[2] == 8
    bad_tag = bytearray(proof); bad_tag[8] = 2
    bad_len = bytearray(proof); bad_len[12] = 31
    noncanon = bytearray(proof); noncanon[20:52] = b"\xff" * 32
    for bad in (proof[:100], bytes(bad_tag), bytes(bad_len), bytes(noncanon), struct.pack("<Q", 1000) + proof[8:], b""):
        with pytest.raises(m.MinaError):
            m.parse_merkle_path(bad)


@pytest.mark.gpu
def test_gpu_account_inclusion_batch(ctx, oracle):
    """end to end on the byte contract: (proof bytes, pub-input bytes, leaf hash) -> verdicts, with the Merkle fold on the GPU"""
    from oracle import pasta_ref as R
    from test_merkle import pp_for
    pp = pp_for(0)
    n, depth = 24, 35
    leaves = rand_scalars(n, P, seed=41)
    proofs, pubs, roots = [], [], []
    for i in range(n):
        sibs = rand_scalars(depth, P, seed=500 + i)
        dirs = [(i * 7 + h * 3) & 1 for h in range(depth)]
        root = R.merkle_root(oracle.le_to_int(leaves[i]), [(dirs[h], oracle.le_to_int(sibs[h])) for h in range(depth)], pp)
        proofs.append(ser_account_proof([(dirs[h], sibs[h].tobytes()) for h in range(depth)]))
        pubs.append(ser_account_pub_inputs(int(root).to_bytes(32, "little"), b"abi-encoded-account" * 3))
        roots.append(root)
    assert ctx.verify_account_inclusion(proofs, pubs, leaves).tolist() == [1] * n
    # one wrong ledger hash, one flipped direction, one truncated proof, one non-canonical leaf, one shorter (depth 3) valid path
    pubs2, proofs2, leaves2 = list(pubs), list(proofs), leaves.copy()
    pubs2[1] = ser_account_pub_inputs(int((roots[1] + 1) % P).to_bytes(32, "little"), b"x")
    p = bytearray(proofs2[2]); p[8] ^= 1; proofs2[2] = bytes(p)
    proofs2[3] = proofs2[3][:500]
    leaves2[4] = 0xFF
    sibs = rand_scalars(3, P, seed=999)
    root5 = R.merkle_root(oracle.le_to_int(leaves[5]), [(0, oracle.le_to_int(sibs[h])) for h in range(3)], pp)
    proofs2[5] = ser_account_proof([(0, sibs[h].tobytes()) for h in range(3)])
    pubs2[5] = ser_account_pub_inputs(int(root5).to_bytes(32, "little"), b"")
    exp = [1] * n
    for k in (1, 2, 3, 4):
        exp[k] = 0
    assert ctx.verify_account_inclusion(proofs2, pubs2, leaves2).tolist() == exp
