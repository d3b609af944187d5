"""Protocol-state wire format and `to_input` flattening -- CPU tests, no GPU.

Pinned by the reference: the reader consumes the 1542-byte state of core/src/utils/constants.rs:22 EXACTLY (record layout,
bin_prot integers, Berkeley slot wrappers).  The state-hash known answer (constants.rs:23-24) needs mina-poseidon's real
`fp_kimchi` tables, which the tree does not hold: it is an expected failure under the surrogate constant set and turns
green the moment the real tables are installed in mina_bridge_amd/poseidon_params.py."""
import base64
import json
import os
import random
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def tip():
    fx = json.load(open(os.path.join(HERE, "golden", "tip_protocol_state.json")))
    return base64.b64decode(fx["protocol_state_base64"]), int(fx["state_hash_field"])


def test_oracle_reader_consumes_reference_state_exactly(tip):
    from oracle import mina_state_ref as S
    raw, _ = tip
    assert len(raw) == 1542
    st = S.parse_protocol_state(raw)
    cs = st["body"]["consensus_state"]
    assert cs["blockchain_length"] == 373218 and cs["curr_global_slot_since_hard_fork"]["slots_per_epoch"] == 7140
    assert len(cs["sub_window_densities"]) == 11 and st["body"]["constants"]["k"] == 290
    assert st["previous_state_hash"] == cs["next_epoch_data"]["lock_checkpoint"]          # the blob's own internal consistency
    assert S.write_protocol_state(st) == raw                                              # writer is the exact inverse
    inp = S.body_to_input(st["body"])
    assert len(inp.fields) == 38 and sum(b for _, b in inp.packeds) == 2381 and len(inp.to_fields()) == 49
    with pytest.raises(S.BinprotError):
        S.parse_protocol_state(raw + b"\0")


def test_library_pack_matches_oracle_on_reference_state(tip):
    import mina_bridge_amd as m
    from oracle import mina_state_ref as S
    raw, _ = tip
    rec, nf, info, used = m.lib.protocol_state_pack(raw)
    st = S.parse_protocol_state(raw)
    exp = b"".join(x.to_bytes(32, "little") for x in [st["previous_state_hash"]] + S.body_to_input(st["body"]).to_fields())
    assert nf == 49 and used == 1542 and rec[: len(exp)].tobytes() == exp and not rec[len(exp):].any()
    assert bytes(info.snarked_ledger_hash) == S.snarked_ledger_hash(st).to_bytes(32, "little")
    assert info.consensus.blockchain_length == 373218 and info.consensus.curr_global_slot == 23894 and info.consensus.min_window_density == 29
    assert list(info.consensus.sub_window_densities)[:11] == [1, 4, 2, 3, 5, 3, 4, 2, 3, 3, 2]
    assert bytes(info.consensus.staking_lock_checkpoint) == st["body"]["consensus_state"]["staking_epoch_data"]["lock_checkpoint"].to_bytes(32, "little")
    assert (info.k, info.slots_per_sub_window, info.sub_windows_per_window, info.grace_period_slots) == (290, 7, 11, 2160)


def test_library_rejects_every_truncation_and_trailing_bytes(tip):
    import mina_bridge_amd as m
    raw, _ = tip
    for cut in list(range(0, 1542, 7)) + [1541]:
        with pytest.raises(m.MinaError):
            m.lib.protocol_state_pack(raw[:cut])
    with pytest.raises(m.MinaError):
        m.lib.protocol_state_pack(raw + b"\0")
    # a non-canonical field element (previous_state_hash = p) and a bad bool
    bad = bytearray(raw); bad[0:32] = (0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001).to_bytes(32, "little")
    with pytest.raises(m.MinaError):
        m.lib.protocol_state_pack(bytes(bad))
    rng = random.Random(3)
    for _ in range(300):                                   # random single-byte corruption never crashes: parses or is rejected
        b = bytearray(raw); b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        try:
            m.lib.protocol_state_pack(bytes(b))
        except m.MinaError:
            pass


# ---- bincode-of-serde form of the same record (inside MinaStateProof, state_proof.rs:28-41): independent writer
def bincode_state(st) -> bytes:
    big = lambda v: v.to_bytes(32, "little")
    u32 = lambda v: struct.pack("<I", v)
    u64 = lambda v: struct.pack("<Q", v)
    byts = lambda b: u64(len(b)) + b
    signed = lambda a: u64(a["magnitude"]) + u32(a["sgn"])

    def local(l):
        return (big(l["stack_frame"]) + big(l["call_stack"]) + big(l["transaction_commitment"]) + big(l["full_transaction_commitment"]) + signed(l["excess"]) +
                signed(l["supply_increase"]) + big(l["ledger"]) + bytes([l["success"]]) + u32(l["account_update_index"]) + u64(0) + bytes([l["will_succeed"]]))

    def regs(g):
        pc = g["pending_coinbase_stack"]
        return big(g["first_pass_ledger"]) + big(g["second_pass_ledger"]) + big(pc["data"]) + big(pc["state"]["init"]) + big(pc["state"]["curr"]) + local(g["local_state"])

    def epoch(e):
        return big(e["ledger"]["hash"]) + u64(e["ledger"]["total_currency"]) + big(e["seed"]) + big(e["start_checkpoint"]) + big(e["lock_checkpoint"]) + u32(e["epoch_length"])

    pk = lambda k: big(k["x"]) + bytes([k["is_odd"]])
    b = st["body"]; bs, cs, kk = b["blockchain_state"], b["consensus_state"], b["constants"]
    ns, ps, fe = bs["staged_ledger_hash"]["non_snark"], bs["ledger_proof_statement"], bs["ledger_proof_statement"]["fee_excess"]
    out = big(st["previous_state_hash"]) + big(b["genesis_state_hash"]) + big(ns["ledger_hash"]) + byts(ns["aux_hash"]) + byts(ns["pending_coinbase_aux"])
    out += big(bs["staged_ledger_hash"]["pending_coinbase_hash"]) + big(bs["genesis_ledger_hash"]) + regs(ps["source"]) + regs(ps["target"])
    out += big(ps["connecting_ledger_left"]) + big(ps["connecting_ledger_right"]) + signed(ps["supply_increase"])
    out += big(fe["fee_token_l"]) + signed(fe["fee_excess_l"]) + big(fe["fee_token_r"]) + signed(fe["fee_excess_r"])
    out += u64(bs["timestamp"]) + byts(bs["body_reference"])
    out += u32(cs["blockchain_length"]) + u32(cs["epoch_count"]) + u32(cs["min_window_density"]) + u64(len(cs["sub_window_densities"]))
    out += b"".join(u32(x) for x in cs["sub_window_densities"]) + byts(cs["last_vrf_output"]) + u64(cs["total_currency"])
    out += u32(0) + u32(cs["curr_global_slot_since_hard_fork"]["slot_number"]) + u32(cs["curr_global_slot_since_hard_fork"]["slots_per_epoch"])
    out += u32(0) + u32(cs["global_slot_since_genesis"]) + epoch(cs["staking_epoch_data"]) + epoch(cs["next_epoch_data"])
    out += bytes([cs["has_ancestor_in_same_checkpoint_window"]]) + pk(cs["block_stake_winner"]) + pk(cs["block_creator"]) + pk(cs["coinbase_receiver"])
    out += bytes([cs["supercharge_coinbase"]])
    out += u32(kk["k"]) + u32(kk["slots_per_epoch"]) + u32(kk["slots_per_sub_window"]) + u32(kk["grace_period_slots"]) + u32(kk["delta"]) + u64(kk["genesis_state_timestamp"])
    return out


def test_bincode_and_binprot_forms_flatten_identically(tip):
    import mina_bridge_amd as m
    from oracle import mina_state_ref as S, state_job_ref as J
    raw, _ = tip
    rng = random.Random(11)
    cases = [S.parse_protocol_state(raw)] + [J.synth_state(rng, rng.randrange(S.P), 5 + i) for i in range(6)]
    for st in cases:
        a, nfa, _, _ = m.lib.protocol_state_pack(S.write_protocol_state(st), m.lib.ENC_BINPROT)
        bc = bincode_state(st)
        b, nfb, _, used = m.lib.protocol_state_pack(bc + b"tail", m.lib.ENC_BINCODE, exact=False)
        assert used == len(bc) and nfa == nfb and (a == b).all()
        exp = b"".join(x.to_bytes(32, "little") for x in [st["previous_state_hash"]] + S.body_to_input(st["body"]).to_fields())
        assert a[: len(exp)].tobytes() == exp
    for cut in range(0, len(bc), 97):
        with pytest.raises(m.MinaError):
            m.lib.protocol_state_pack(bc[:cut], m.lib.ENC_BINCODE)


def test_packing_edge_cases():
    """greedy packing: a chunk that would bring the running width to 255 bits starts a new element (openmina `Inputs::to_fields`)"""
    from oracle import mina_state_ref as S
    inp = S.Inputs()
    for _ in range(3):
        inp.u64((1 << 64) - 1)
    inp.packed((1 << 62) - 1, 62)            # 254 bits so far: fits
    inp.boolean(True)                        # 255: new element
    f = inp.to_fields()
    assert f == [(1 << 254) - 1, 1]
    empty = S.Inputs()
    assert empty.to_fields() == []


def test_state_hash_known_answer(tip):
    """constants.rs:22-24: MinaHash(MINA_TIP_PROTOCOL_STATE) == MINA_TIP_STATE_HASH_FIELD -- the only Poseidon known answer the
    reference holds.  Needs the real fp_kimchi round constants (absent offline)."""
    import mina_bridge_amd.poseidon_params as PP
    from oracle import mina_state_ref as S, pasta_ref as R
    raw, want = tip
    mds, rc = PP.default_params_ints(0); name = PP.NAME
    import os
    if os.environ.get("MINA_POSEIDON_PARAMS_FP"):            # a file drop of the real table (o1js JSON / mina-poseidon Rust source): no code change needed
        import mina_bridge_amd as m
        raw_p = m.lib.poseidon_params_parse(0, open(os.environ["MINA_POSEIDON_PARAMS_FP"]).read()).reshape(174, 32)
        vals = [int.from_bytes(r.tobytes(), "little") for r in raw_p]
        mds, rc, name = [vals[3 * i: 3 * i + 3] for i in range(3)], [vals[9 + 3 * i: 12 + 3 * i] for i in range(55)], "file:" + os.environ["MINA_POSEIDON_PARAMS_FP"]
    got = S.protocol_state_hash(S.parse_protocol_state(raw), R.PoseidonParams(R.P, mds, rc, name))
    if got != want and "UNPINNED" in name:
        pytest.xfail("surrogate Poseidon constants installed (mina_bridge_amd/poseidon_params.py): the real fp_kimchi tables are not in the reference tree")
    assert got == want
