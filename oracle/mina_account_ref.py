"""CPU restatement (TEST INFRASTRUCTURE ONLY) of the Proof-of-Account leg: account container, `Account` ABI encoding, account hash
(8f-3).

* ABI side -- PINNED BY THE REFERENCE: `impl TryFrom<&MinaAccount> for Account` (core/src/sol/account.rs:25-314) + the Solidity structs
  (contract/src/MinaAccountValidationExample.sol:76-164) + standard Solidity ABI rules for `abi.decode(pubInput[40:], (Account))`
  (MinaAccountValidationExample.sol:70).  `abi_encode_account` below follows those two files field by field.
* container + hash side -- [UPSTREAM-RECALL]: `MinaBaseAccountBinableArgStableV2` (mina-p2p-messages, pin core/Cargo.toml:24) in its
  bin_prot form (core/src/mina.rs:307-313 reads it so) and in the serde/bincode form inside `MinaAccountProof`
  (core/src/proof/account_proof.rs:30-35); `Account::hash` = hash_with_kimchi("MinaAccount", to_inputs) with the zkapp / verification
  key / zkapp-uri sub-hashes (openmina ledger `account.rs`).  No serialized account or account hash exists in the tree: unpinned.
"""
from __future__ import annotations

import struct

from . import mina_state_ref as S
from . import pasta_ref as R

P = R.P
AUTH = ["None", "Either", "Proof", "Signature", "Impossible"]
PERM_FIELDS = ["edit_state", "access", "send", "receive", "set_delegate", "set_permissions", "set_verification_key", "set_zkapp_uri",
               "edit_action_state", "set_token_symbol", "increment_nonce", "set_voting_for", "set_timing"]


# ------------------------------------------------------------------------------------------------ writers (bincode / bin_prot)
class W:
    def __init__(self, binprot: bool):
        self.bp, self.out = binprot, bytearray()

    def big(self, v): self.out += int(v).to_bytes(32, "little")
    def boolean(self, b): self.out.append(1 if b else 0)
    def u32(self, v): self.out += S.w_nat(v) if self.bp else struct.pack("<I", v)
    def u64(self, v): self.out += S.w_nat(v) if self.bp else struct.pack("<Q", v)
    def variant(self, t): self.out += bytes([t]) if self.bp else struct.pack("<I", t)
    def option(self, some): self.out.append(1 if some else 0)
    def string(self, b): self.out += (S.w_nat(len(b)) if self.bp else struct.pack("<Q", len(b))) + b
    def padded_end(self):
        if self.bp:
            self.out.append(0)


def write_account(a: dict, binprot: bool) -> bytes:
    w = W(binprot)
    w.big(a["public_key"]["x"]); w.boolean(a["public_key"]["is_odd"])
    w.big(a["token_id"]); w.string(a["token_symbol"]); w.u64(a["balance"]); w.u32(a["nonce"]); w.big(a["receipt_chain_hash"])
    w.option(a["delegate"] is not None)
    if a["delegate"] is not None:
        w.big(a["delegate"]["x"]); w.boolean(a["delegate"]["is_odd"])
    w.big(a["voting_for"])
    t = a["timing"]
    if t is None:
        w.variant(0)
    else:
        w.variant(1); w.u64(t["initial_minimum_balance"]); w.variant(0); w.u32(t["cliff_time"]); w.u64(t["cliff_amount"]); w.variant(0); w.u32(t["vesting_period"])
        w.u64(t["vesting_increment"])
    p = a["permissions"]
    for f in PERM_FIELDS:
        if f == "set_verification_key":
            w.variant(p[f][0]); w.u32(p[f][1])
        else:
            w.variant(p[f])
    z = a["zkapp"]
    w.option(z is not None)
    if z is not None:
        for x in z["app_state"]:
            w.big(x)
        w.padded_end()
        vk = z["verification_key"]
        w.option(vk is not None)
        if vk is not None:
            w.variant(vk["max_proofs_verified"]); w.variant(vk["actual_wrap_domain_size"])
            for x, y in vk["sigma_comm"]:
                w.big(x); w.big(y)
            w.padded_end()
            for x, y in vk["coefficients_comm"]:
                w.big(x); w.big(y)
            w.padded_end()
            for x, y in vk["other_comm"]:                    # generic, psm, complete_add, mul, emul, endomul_scalar
                w.big(x); w.big(y)
        w.u32(z["zkapp_version"])
        for x in z["action_state"]:
            w.big(x)
        w.padded_end()
        w.variant(0); w.u32(z["last_action_slot"]); w.boolean(z["proved_state"]); w.string(z["zkapp_uri"])
    return bytes(w.out)


def write_account_proof(merkle_path, account: dict) -> bytes:
    """bincode `MinaAccountProof` (account_proof.rs:9-14,30-35): Vec<MerkleNode> then the account"""
    out = struct.pack("<Q", len(merkle_path))
    for d, sib in merkle_path:
        out += struct.pack("<I", d) + struct.pack("<Q", 32) + int(sib).to_bytes(32, "little")
    return out + write_account(account, binprot=False)


# ------------------------------------------------------------------------------------------------ Solidity ABI (pinned by the reference)
def _word_uint(v): return int(v).to_bytes(32, "big")
def _word_b32(v): return int(v).to_bytes(32, "little")       # FixedBytes::try_from(bigint.as_ref()): the 32 stored (little-endian) bytes, as they are
def _dyn_bytes(b): return _word_uint(len(b)) + b + b"\0" * ((-len(b)) % 32)


def abi_encode_account(a: dict) -> bytes:
    """`Account::abi_encode()` (alloy, as a single dynamic value: offset word, then the tuple) of sol/account.rs:25-314"""
    z = a["zkapp"]
    zero_c = (0, 0)
    if z is None:
        zk = {"app_state": [0] * 8, "verification_key": None, "zkapp_version": 0, "action_state": [0] * 5, "last_action_slot": 0, "proved_state": False, "zkapp_uri": b""}
    else:
        zk = z
    vk = zk["verification_key"]
    if vk is None:
        vk = {"max_proofs_verified": 0, "actual_wrap_domain_size": 0, "sigma_comm": [zero_c] * 7, "coefficients_comm": [zero_c] * 15, "other_comm": [zero_c] * 6}
    zk_head = b"".join(_word_b32(x) for x in zk["app_state"])
    zk_head += _word_uint(vk["max_proofs_verified"]) + _word_uint(vk["actual_wrap_domain_size"])
    for x, y in list(vk["sigma_comm"]) + list(vk["coefficients_comm"]) + list(vk["other_comm"]):
        zk_head += _word_b32(x) + _word_b32(y)
    zk_head += _word_uint(zk["zkapp_version"]) + b"".join(_word_b32(x) for x in zk["action_state"]) + _word_uint(zk["last_action_slot"]) + _word_uint(1 if zk["proved_state"] else 0)
    zk_head += _word_uint(len(zk_head) + 32)                   # offset of zkappUri inside the ZkappAccount tuple
    zk_enc = zk_head + _dyn_bytes(zk["zkapp_uri"])
    d = a["delegate"] if a["delegate"] is not None else {"x": 0, "is_odd": True}
    t = a["timing"] or {"initial_minimum_balance": 0, "cliff_time": 0, "cliff_amount": 0, "vesting_period": 0, "vesting_increment": 0}
    p = a["permissions"]
    sym = _dyn_bytes(a["token_symbol"])
    head_words = 30
    head = _word_b32(a["public_key"]["x"]) + _word_uint(1 if a["public_key"]["is_odd"] else 0) + _word_b32(a["token_id"])
    head += _word_uint(head_words * 32)                        # tokenSymbol offset
    head += _word_uint(a["balance"]) + _word_uint(a["nonce"]) + _word_b32(a["receipt_chain_hash"]) + _word_b32(d["x"]) + _word_uint(1 if d["is_odd"] else 0) + _word_b32(a["voting_for"])
    head += _word_uint(t["initial_minimum_balance"]) + _word_uint(t["cliff_time"]) + _word_uint(t["cliff_amount"]) + _word_uint(t["vesting_period"]) + _word_uint(t["vesting_increment"])
    for f in PERM_FIELDS:
        if f == "set_verification_key":
            head += _word_uint(p[f][0]) + _word_uint(p[f][1])
        else:
            head += _word_uint(p[f])
    head += _word_uint(head_words * 32 + len(sym))             # zkapp offset
    assert len(head) == head_words * 32
    return _word_uint(32) + head + sym + zk_enc


# ------------------------------------------------------------------------------------------------ account hash  [UPSTREAM-RECALL]
def auth_bits(tag: int):
    """AuthRequired -> (constant, signature_necessary, signature_sufficient)"""
    return {0: (1, 0, 1), 1: (0, 0, 1), 2: (0, 0, 0), 3: (0, 1, 1), 4: (1, 1, 0)}[tag]


def zkapp_uri_hash(uri: bytes, pp) -> int:
    inp = S.Inputs()
    inp.bytes_lsb_first(uri)
    inp.boolean(True)
    return S.hash_with_kimchi("MinaZkappUri", inp.to_fields(), pp)


DUMMY_VK = {"max_proofs_verified": 2, "actual_wrap_domain_size": 2, "sigma_comm": [(1, 2)] * 7, "coefficients_comm": [(1, 2)] * 15, "other_comm": [(1, 2)] * 6}


def vk_hash(vk, pp) -> int:
    vk = vk or DUMMY_VK
    inp = S.Inputs()
    for tag in (vk["max_proofs_verified"], vk["actual_wrap_domain_size"]):
        for i in range(3):
            inp.boolean(i == tag)
    for x, y in list(vk["sigma_comm"]) + list(vk["coefficients_comm"]) + list(vk["other_comm"]):
        inp.field(x); inp.field(y)
    return S.hash_with_kimchi("MinaSideLoadedVk", inp.to_fields(), pp)


def zkapp_hash(z, pp) -> int:
    z = z or {"app_state": [0] * 8, "verification_key": None, "zkapp_version": 0, "action_state": [0] * 5, "last_action_slot": 0, "proved_state": False, "zkapp_uri": b""}
    inp = S.Inputs()
    inp.field(zkapp_uri_hash(z["zkapp_uri"], pp))
    inp.boolean(z["proved_state"]); inp.u32(z["last_action_slot"])
    for x in z["action_state"]:
        inp.field(x)
    inp.u32(z["zkapp_version"])
    inp.field(vk_hash(z["verification_key"], pp))
    for x in z["app_state"]:
        inp.field(x)
    return S.hash_with_kimchi("MinaZkappAccount", inp.to_fields(), pp)


def account_to_input(a: dict, pp) -> S.Inputs:
    inp = S.Inputs()
    inp.field(zkapp_hash(a["zkapp"], pp))
    p = a["permissions"]
    for f in PERM_FIELDS:
        tag = p[f][0] if f == "set_verification_key" else p[f]
        for b in auth_bits(tag):
            inp.boolean(b)
        if f == "set_verification_key":
            inp.u32(p[f][1])
    t = a["timing"]
    if t is None:
        inp.boolean(False); inp.u64(0); inp.u32(0); inp.u64(0); inp.u32(1); inp.u64(0)
    else:
        inp.boolean(True); inp.u64(t["initial_minimum_balance"]); inp.u32(t["cliff_time"]); inp.u64(t["cliff_amount"]); inp.u32(t["vesting_period"]); inp.u64(t["vesting_increment"])
    inp.field(a["voting_for"])
    d = a["delegate"] or {"x": 0, "is_odd": False}
    inp.field(d["x"]); inp.boolean(d["is_odd"])
    inp.field(a["receipt_chain_hash"])
    inp.u32(a["nonce"]); inp.u64(a["balance"])
    inp.packed(int.from_bytes(a["token_symbol"].ljust(6, b"\0")[:6], "little"), 48)
    inp.field(a["token_id"])
    inp.field(a["public_key"]["x"]); inp.boolean(a["public_key"]["is_odd"])
    return inp


def account_hash(a: dict, pp) -> int:
    return S.hash_with_kimchi(S.PREFIX_ACCOUNT, account_to_input(a, pp).to_fields(), pp)


# ------------------------------------------------------------------------------------------------ synthetic accounts
def synth_account(rng, zkapp: bool, timed: bool, delegate: bool, with_vk: bool = True) -> dict:
    rf = lambda: rng.randrange(P)
    a = {"public_key": {"x": rf(), "is_odd": bool(rng.randrange(2))}, "token_id": rf(), "token_symbol": bytes(rng.randrange(65, 91) for _ in range(rng.randrange(0, 7))),
         "balance": rng.randrange(1 << 64), "nonce": rng.randrange(1 << 32), "receipt_chain_hash": rf(),
         "delegate": {"x": rf(), "is_odd": bool(rng.randrange(2))} if delegate else None, "voting_for": rf(),
         "timing": {"initial_minimum_balance": rng.randrange(1 << 64), "cliff_time": rng.randrange(1 << 32), "cliff_amount": rng.randrange(1 << 64),
                    "vesting_period": rng.randrange(1, 1 << 32), "vesting_increment": rng.randrange(1 << 64)} if timed else None,
         "permissions": {f: ((rng.randrange(5), rng.randrange(1 << 16)) if f == "set_verification_key" else rng.randrange(5)) for f in PERM_FIELDS},
         "zkapp": None}
    if zkapp:
        pt = lambda: (rf(), rf())
        a["zkapp"] = {"app_state": [rf() for _ in range(8)],
                      "verification_key": {"max_proofs_verified": rng.randrange(3), "actual_wrap_domain_size": rng.randrange(3), "sigma_comm": [pt() for _ in range(7)],
                                           "coefficients_comm": [pt() for _ in range(15)], "other_comm": [pt() for _ in range(6)]} if with_vk else None,
                      "zkapp_version": rng.randrange(1 << 16), "action_state": [rf() for _ in range(5)], "last_action_slot": rng.randrange(1 << 32),
                      "proved_state": bool(rng.randrange(2)), "zkapp_uri": bytes(rng.randrange(32, 127) for _ in range(rng.randrange(0, 40)))}
    return a
