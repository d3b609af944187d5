"""CPU restatement (test infrastructure) of the protocol-state leg of `verify_mina_state` -- SURVEY.md 8f-1 / 8a rows a1, a15.

What it restates (all un-vendored; pins core/Cargo.toml:23-24, core/Cargo.lock:3750-3752,3812-3814):

* `MinaStateProtocolStateValueStableV2::binprot_read` -- the reference reads every protocol state of the candidate chain
  with it (core/src/mina.rs:143-168) and the only serialized state in the tree is the base64 blob at
  core/src/utils/constants.rs:22 (1542 bytes).  The reader below consumes that blob EXACTLY (tests/test_protocol_state.py),
  which pins the record layout, the integer encoding and the Berkeley `Since_hard_fork` / `Since_genesis` wrapper tags.
* mina `Protocol_state.Body.to_input` / openmina `mina-p2p-messages` `hash_input.rs` + `v2/hashing.rs` (`ToInput`, `Inputs::to_fields`):
  the field/packed-chunk flattening of a protocol state body, and `hash_with_kimchi` with the 20-byte '*'-padded prefixes.
  [UPSTREAM-RECALL]: the order of fields inside `to_input` is written from knowledge of the published code and is only
  checkable end-to-end through the (protocol state, state hash) pair of constants.rs:22-24 -- which also needs the real
  `fp_kimchi` Poseidon tables.  Those tables are not in the tree; until they are installed the known-answer test is an
  expected failure ("parity unpinned" for the hash, pinned for the parser).

Nothing here is imported by the product (`mina_bridge_amd/`).
"""
from __future__ import annotations

import hashlib

from . import pasta_ref as R

P = R.P if hasattr(R, "P") else 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001


# ------------------------------------------------------------------------------------------------ bin_prot primitives
class BinprotError(ValueError):
    pass


class Reader:
    def __init__(self, data: bytes):
        self.b, self.p = bytes(data), 0

    def u8(self) -> int:
        if self.p >= len(self.b):
            raise BinprotError("truncated")
        v = self.b[self.p]
        self.p += 1
        return v

    def take(self, n: int) -> bytes:
        if self.p + n > len(self.b):
            raise BinprotError("truncated")
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def nat(self) -> int:
        """bin_prot Nat0 / non-negative int: < 0x80 one byte; 0xfe + u16; 0xfd + u32; 0xfc + u64 (little-endian)"""
        c = self.u8()
        if c < 0x80:
            return c
        if c == 0xFE:
            return int.from_bytes(self.take(2), "little")
        if c == 0xFD:
            return int.from_bytes(self.take(4), "little")
        if c == 0xFC:
            return int.from_bytes(self.take(8), "little")
        raise BinprotError("bad integer prefix 0x%02x" % c)

    def big(self) -> int:
        """mina-p2p-messages `BigInt`: 32 raw little-endian bytes"""
        return int.from_bytes(self.take(32), "little")

    def string(self) -> bytes:
        return self.take(self.nat())

    def boolean(self) -> bool:
        v = self.u8()
        if v > 1:
            raise BinprotError("bad bool")
        return bool(v)

    def tagged_u32(self) -> int:
        """`Since_hard_fork of u32` / `Since_genesis of u32`: constructor tag 0 then the number"""
        if self.u8() != 0:
            raise BinprotError("bad variant tag")
        return self.nat()


def w_nat(v: int) -> bytes:
    if v < 0x80:
        return bytes([v])
    if v < 0x10000:
        return b"\xfe" + v.to_bytes(2, "little")
    if v < 0x100000000:
        return b"\xfd" + v.to_bytes(4, "little")
    return b"\xfc" + v.to_bytes(8, "little")


def w_big(v: int) -> bytes:
    return v.to_bytes(32, "little")


def w_string(s: bytes) -> bytes:
    return w_nat(len(s)) + s


# ------------------------------------------------------------------------------------------------ the record
def _signed(r):
    return {"magnitude": r.nat(), "sgn": _sgn(r)}


def _sgn(r):
    v = r.u8()
    if v > 1:
        raise BinprotError("bad sgn")
    return v          # 0 = Pos, 1 = Neg (constructor order of `Sgn.t`)


def _local_state(r):
    d = {"stack_frame": r.big(), "call_stack": r.big(), "transaction_commitment": r.big(), "full_transaction_commitment": r.big(),
         "excess": _signed(r), "supply_increase": _signed(r), "ledger": r.big(), "success": r.boolean(), "account_update_index": r.nat()}
    tbl = []
    for _ in range(r.nat()):
        tbl.append([r.u8() for _ in range(r.nat())])      # failure tags are single-byte constructors (empty on every block seen)
    d["failure_status_tbl"] = tbl
    d["will_succeed"] = r.boolean()
    return d


def _registers(r):
    return {"first_pass_ledger": r.big(), "second_pass_ledger": r.big(),
            "pending_coinbase_stack": {"data": r.big(), "state": {"init": r.big(), "curr": r.big()}},
            "local_state": _local_state(r)}


def _epoch_data(r):
    return {"ledger": {"hash": r.big(), "total_currency": r.nat()}, "seed": r.big(), "start_checkpoint": r.big(),
            "lock_checkpoint": r.big(), "epoch_length": r.nat()}


def _pk(r):
    return {"x": r.big(), "is_odd": r.boolean()}


def read_protocol_state(r: Reader) -> dict:
    d = {"previous_state_hash": r.big()}
    b = {"genesis_state_hash": r.big()}
    bs = {"staged_ledger_hash": {"non_snark": {"ledger_hash": r.big(), "aux_hash": r.string(), "pending_coinbase_aux": r.string()},
                                 "pending_coinbase_hash": r.big()},
          "genesis_ledger_hash": r.big()}
    st = {"source": _registers(r), "target": _registers(r), "connecting_ledger_left": r.big(), "connecting_ledger_right": r.big(),
          "supply_increase": _signed(r)}
    st["fee_excess"] = {"fee_token_l": r.big(), "fee_excess_l": _signed(r), "fee_token_r": r.big(), "fee_excess_r": _signed(r)}
    if r.u8() != 0:
        raise BinprotError("sok_digest must be unit")
    bs["ledger_proof_statement"] = st
    bs["timestamp"] = r.nat()
    bs["body_reference"] = r.string()
    b["blockchain_state"] = bs
    cs = {"blockchain_length": r.nat(), "epoch_count": r.nat(), "min_window_density": r.nat()}
    cs["sub_window_densities"] = [r.nat() for _ in range(r.nat())]
    cs["last_vrf_output"] = r.string()
    cs["total_currency"] = r.nat()
    cs["curr_global_slot_since_hard_fork"] = {"slot_number": r.tagged_u32(), "slots_per_epoch": r.nat()}
    cs["global_slot_since_genesis"] = r.tagged_u32()
    cs["staking_epoch_data"] = _epoch_data(r)
    cs["next_epoch_data"] = _epoch_data(r)
    cs["has_ancestor_in_same_checkpoint_window"] = r.boolean()
    cs["block_stake_winner"] = _pk(r)
    cs["block_creator"] = _pk(r)
    cs["coinbase_receiver"] = _pk(r)
    cs["supercharge_coinbase"] = r.boolean()
    b["consensus_state"] = cs
    b["constants"] = {"k": r.nat(), "slots_per_epoch": r.nat(), "slots_per_sub_window": r.nat(), "grace_period_slots": r.nat(),
                      "delta": r.nat(), "genesis_state_timestamp": r.nat()}
    d["body"] = b
    return d


def parse_protocol_state(data: bytes) -> dict:
    r = Reader(data)
    d = read_protocol_state(r)
    if r.p != len(data):
        raise BinprotError("trailing bytes")
    return d


# writer (mints synthetic chains for the tests; the inverse of the reader above)
def _w_signed(a):
    return w_nat(a["magnitude"]) + bytes([a["sgn"]])


def _w_local(l):
    out = w_big(l["stack_frame"]) + w_big(l["call_stack"]) + w_big(l["transaction_commitment"]) + w_big(l["full_transaction_commitment"])
    out += _w_signed(l["excess"]) + _w_signed(l["supply_increase"]) + w_big(l["ledger"]) + bytes([l["success"]]) + w_nat(l["account_update_index"])
    out += w_nat(len(l["failure_status_tbl"]))
    for row in l["failure_status_tbl"]:
        out += w_nat(len(row)) + bytes(row)
    return out + bytes([l["will_succeed"]])


def _w_registers(g):
    pc = g["pending_coinbase_stack"]
    return (w_big(g["first_pass_ledger"]) + w_big(g["second_pass_ledger"]) + w_big(pc["data"]) + w_big(pc["state"]["init"]) +
            w_big(pc["state"]["curr"]) + _w_local(g["local_state"]))


def _w_epoch(e):
    return (w_big(e["ledger"]["hash"]) + w_nat(e["ledger"]["total_currency"]) + w_big(e["seed"]) + w_big(e["start_checkpoint"]) +
            w_big(e["lock_checkpoint"]) + w_nat(e["epoch_length"]))


def _w_pk(k):
    return w_big(k["x"]) + bytes([k["is_odd"]])


def write_protocol_state(d: dict) -> bytes:
    b = d["body"]
    bs, cs, k = b["blockchain_state"], b["consensus_state"], b["constants"]
    ns, st = bs["staged_ledger_hash"]["non_snark"], bs["ledger_proof_statement"]
    fe = st["fee_excess"]
    out = w_big(d["previous_state_hash"]) + w_big(b["genesis_state_hash"])
    out += w_big(ns["ledger_hash"]) + w_string(ns["aux_hash"]) + w_string(ns["pending_coinbase_aux"]) + w_big(bs["staged_ledger_hash"]["pending_coinbase_hash"])
    out += w_big(bs["genesis_ledger_hash"])
    out += _w_registers(st["source"]) + _w_registers(st["target"]) + w_big(st["connecting_ledger_left"]) + w_big(st["connecting_ledger_right"])
    out += _w_signed(st["supply_increase"]) + w_big(fe["fee_token_l"]) + _w_signed(fe["fee_excess_l"]) + w_big(fe["fee_token_r"]) + _w_signed(fe["fee_excess_r"])
    out += b"\x00" + w_nat(bs["timestamp"]) + w_string(bs["body_reference"])
    out += w_nat(cs["blockchain_length"]) + w_nat(cs["epoch_count"]) + w_nat(cs["min_window_density"])
    out += w_nat(len(cs["sub_window_densities"])) + b"".join(w_nat(x) for x in cs["sub_window_densities"])
    out += w_string(cs["last_vrf_output"]) + w_nat(cs["total_currency"])
    out += b"\x00" + w_nat(cs["curr_global_slot_since_hard_fork"]["slot_number"]) + w_nat(cs["curr_global_slot_since_hard_fork"]["slots_per_epoch"])
    out += b"\x00" + w_nat(cs["global_slot_since_genesis"])
    out += _w_epoch(cs["staking_epoch_data"]) + _w_epoch(cs["next_epoch_data"]) + bytes([cs["has_ancestor_in_same_checkpoint_window"]])
    out += _w_pk(cs["block_stake_winner"]) + _w_pk(cs["block_creator"]) + _w_pk(cs["coinbase_receiver"]) + bytes([cs["supercharge_coinbase"]])
    out += w_nat(k["k"]) + w_nat(k["slots_per_epoch"]) + w_nat(k["slots_per_sub_window"]) + w_nat(k["grace_period_slots"]) + w_nat(k["delta"])
    out += w_nat(k["genesis_state_timestamp"])
    return out


# ------------------------------------------------------------------------------------------------ to_input
class Inputs:
    """openmina `Inputs` (mina `Random_oracle_input.Chunked`): whole field elements, then (value, bits) chunks packed greedily
    into field elements of < 255 bits, in order."""

    def __init__(self):
        self.fields: list[int] = []
        self.packeds: list[tuple[int, int]] = []

    def field(self, x: int):
        self.fields.append(x % P)

    def packed(self, x: int, bits: int):
        assert 0 <= x < (1 << bits)
        self.packeds.append((x, bits))

    def boolean(self, b):
        self.packed(1 if b else 0, 1)

    def u32(self, x):
        self.packed(x, 32)

    def u64(self, x):
        self.packed(x, 64)

    def bytes_lsb_first(self, data: bytes, nbits: int | None = None):
        """one 1-bit chunk per bit, least-significant bit of each byte first (openmina `append_bytes`)"""
        n = 0
        for byte in data:
            for i in range(8):
                if nbits is not None and n == nbits:
                    return
                self.boolean((byte >> i) & 1)
                n += 1

    def to_fields(self) -> list[int]:
        out = list(self.fields)
        nbits, cur = 0, 0
        for x, b in self.packeds:
            nbits += b
            if nbits < 255:
                cur = (cur << b) + x
            else:
                out.append(cur % P)
                cur, nbits = x, b
        if nbits > 0:
            out.append(cur % P)
        return out


def _in_signed(inp, a):
    inp.u64(a["magnitude"])
    inp.boolean(a["sgn"] == 0)                       # Sgn.Pos -> 1, Neg -> 0


def _in_local(inp, l):
    inp.field(l["stack_frame"]); inp.field(l["call_stack"]); inp.field(l["transaction_commitment"]); inp.field(l["full_transaction_commitment"])
    _in_signed(inp, l["excess"]); _in_signed(inp, l["supply_increase"])
    inp.field(l["ledger"]); inp.u32(l["account_update_index"]); inp.boolean(l["success"]); inp.boolean(l["will_succeed"])


def _in_registers(inp, g):
    inp.field(g["first_pass_ledger"]); inp.field(g["second_pass_ledger"])
    pc = g["pending_coinbase_stack"]
    inp.field(pc["data"]); inp.field(pc["state"]["init"]); inp.field(pc["state"]["curr"])
    _in_local(inp, g["local_state"])


def _in_epoch(inp, e):
    inp.field(e["seed"]); inp.field(e["start_checkpoint"]); inp.u32(e["epoch_length"])
    inp.field(e["ledger"]["hash"]); inp.u64(e["ledger"]["total_currency"])
    inp.field(e["lock_checkpoint"])


def _in_pk(inp, k):
    inp.field(k["x"]); inp.boolean(k["is_odd"])


def staged_ledger_non_snark_digest(ns: dict) -> bytes:
    """SHA-256(ledger_hash as 32 big-endian bytes || aux_hash || pending_coinbase_aux)"""
    return hashlib.sha256(ns["ledger_hash"].to_bytes(32, "big") + ns["aux_hash"] + ns["pending_coinbase_aux"]).digest()


def body_to_input(b: dict) -> Inputs:
    inp = Inputs()
    inp.field(b["genesis_state_hash"])
    bs = b["blockchain_state"]
    inp.bytes_lsb_first(staged_ledger_non_snark_digest(bs["staged_ledger_hash"]["non_snark"]))
    inp.field(bs["staged_ledger_hash"]["pending_coinbase_hash"])
    inp.field(bs["genesis_ledger_hash"])
    st = bs["ledger_proof_statement"]
    _in_registers(inp, st["source"]); _in_registers(inp, st["target"])
    inp.field(st["connecting_ledger_left"]); inp.field(st["connecting_ledger_right"])
    _in_signed(inp, st["supply_increase"])
    fe = st["fee_excess"]
    inp.field(fe["fee_token_l"]); _in_signed(inp, fe["fee_excess_l"]); inp.field(fe["fee_token_r"]); _in_signed(inp, fe["fee_excess_r"])
    inp.u64(bs["timestamp"])
    inp.bytes_lsb_first(bs["body_reference"])
    cs = b["consensus_state"]
    inp.u32(cs["blockchain_length"]); inp.u32(cs["epoch_count"]); inp.u32(cs["min_window_density"])
    for x in cs["sub_window_densities"]:
        inp.u32(x)
    inp.bytes_lsb_first(cs["last_vrf_output"], 253)          # truncated VRF output: 31 bytes + the low 5 bits of the last one
    inp.u64(cs["total_currency"])
    inp.u32(cs["curr_global_slot_since_hard_fork"]["slot_number"]); inp.u32(cs["curr_global_slot_since_hard_fork"]["slots_per_epoch"])
    inp.u32(cs["global_slot_since_genesis"])
    inp.boolean(cs["has_ancestor_in_same_checkpoint_window"]); inp.boolean(cs["supercharge_coinbase"])
    _in_epoch(inp, cs["staking_epoch_data"]); _in_epoch(inp, cs["next_epoch_data"])
    _in_pk(inp, cs["block_stake_winner"]); _in_pk(inp, cs["block_creator"]); _in_pk(inp, cs["coinbase_receiver"])
    k = b["constants"]
    inp.u32(k["k"]); inp.u32(k["delta"]); inp.u32(k["slots_per_epoch"]); inp.u32(k["slots_per_sub_window"]); inp.u32(k["grace_period_slots"])
    inp.u64(k["genesis_state_timestamp"])
    return inp


# ------------------------------------------------------------------------------------------------ hashing
def prefix_field(s: str) -> int:
    b = s.encode()
    assert len(b) <= 20
    return int.from_bytes(b + b"*" * (20 - len(b)), "little")


PREFIX_PROTOCOL_STATE = "MinaProtoState"
PREFIX_PROTOCOL_STATE_BODY = "MinaProtoStateBody"
PREFIX_ACCOUNT = "MinaAccount"


def salt(prefix: str, pp: R.PoseidonParams):
    return R.poseidon_permute([prefix_field(prefix) % pp.m, 0, 0], pp)


def hash_with_init(init, xs, pp: R.PoseidonParams) -> int:
    """mina `Random_oracle.hash ~init xs` = sponge started from `init` (Absorbed 0), absorb xs, squeeze"""
    s = list(init)
    cnt = 0
    for x in xs:
        if cnt == 2:
            s = R.poseidon_permute(s, pp)
            cnt = 0
        s[cnt] = (s[cnt] + x) % pp.m
        cnt += 1
    return R.poseidon_permute(s, pp)[0]


def hash_with_kimchi(prefix: str, xs, pp: R.PoseidonParams) -> int:
    return hash_with_init(salt(prefix, pp), xs, pp)


def protocol_state_body_hash(body: dict, pp: R.PoseidonParams) -> int:
    return hash_with_kimchi(PREFIX_PROTOCOL_STATE_BODY, body_to_input(body).to_fields(), pp)


def protocol_state_hash(state: dict, pp: R.PoseidonParams) -> int:
    return hash_with_kimchi(PREFIX_PROTOCOL_STATE, [state["previous_state_hash"] % pp.m, protocol_state_body_hash(state["body"], pp)], pp)


def snarked_ledger_hash(state: dict) -> int:
    """`Blockchain_state.snarked_ledger_hash` = ledger_proof_statement.target.first_pass_ledger -- the ledger hashes of
    MinaStatePubInputs (core/src/mina.rs:203-213 takes them from the node's `snarkedLedgerHash`)"""
    return state["body"]["blockchain_state"]["ledger_proof_statement"]["target"]["first_pass_ledger"]
