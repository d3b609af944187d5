"""Independent Python writers of the reference's containers (tests only): the Pickles wrap proof `MinaBaseProofStableV2` in
bin_prot (core/src/mina.rs:235-248) and serde/bincode form, bincode `MinaStateProof` (core/src/proof/state_proof.rs:28-41) and
`MinaStatePubInputs` (state_proof.rs:10-25).  Written from the type definitions, not from the library's reader."""
import random
import struct


class W:
    def __init__(self, binprot: bool):
        self.bp, self.o = binprot, bytearray()

    def big(self, v): self.o += int(v).to_bytes(32, "little")
    def boolean(self, b): self.o.append(1 if b else 0)
    def variant(self, t): self.o += bytes([t]) if self.bp else struct.pack("<I", t)
    def option(self, some): self.o.append(1 if some else 0)
    def length(self, n): self.o += self._nat(n) if self.bp else struct.pack("<Q", n)
    def unit(self):
        if self.bp:
            self.o.append(0)
    padded_end = unit

    @staticmethod
    def _nat(v):
        if v < 0x80: return bytes([v])
        if v < 0x10000: return b"\xfe" + v.to_bytes(2, "little")
        if v < 0x100000000: return b"\xfd" + v.to_bytes(4, "little")
        return b"\xfc" + v.to_bytes(8, "little")

    def i64(self, u):
        """a 64-bit limb, signed on the wire"""
        v = u - (1 << 64) if u >= (1 << 63) else u
        if not self.bp:
            self.o += struct.pack("<q", v); return
        if 0 <= v < 0x80: self.o.append(v)
        elif -0x80 <= v < 0: self.o += b"\xff" + struct.pack("<b", v)
        elif -0x8000 <= v < 0x8000: self.o += b"\xfe" + struct.pack("<h", v)
        elif -0x80000000 <= v < 0x80000000: self.o += b"\xfd" + struct.pack("<i", v)
        else: self.o += b"\xfc" + struct.pack("<q", v)

    def chal(self, c):                                           # 128-bit value as two limbs, low first, PaddedSeq<_, 2>
        self.i64(c & ((1 << 64) - 1)); self.i64(c >> 64); self.padded_end()

    def pt(self, p): self.big(p[0]); self.big(p[1])
    def chr(self, v): self.o.append(v)


def synth_wrap_proof(rng: random.Random, k: int = 15, lookups: bool = True) -> dict:
    """random field content in the shape of a wrap proof (no cryptographic meaning).  lookups=False: the feature flags of gates that read
    lookup tables stay off and there is no joint combiner -- the shape of Mina's blockchain proofs, and the only one the boundary accepts
    (mina_verify_state rejects lookup features: the linearization carries no lookup terms); foreign_field_add stays random."""
    d = _synth_wrap_proof(rng, k)
    if not lookups:
        d["joint_combiner"] = None
        d["feature_flags"] = [False, False, d["feature_flags"][2], False, False, False, False, False]
    return d


def _synth_wrap_proof(rng: random.Random, k: int = 15) -> dict:
    P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
    rf = lambda: rng.randrange(P)
    c128 = lambda: rng.getrandbits(128) if rng.randrange(4) else rng.choice([0, 1, (1 << 128) - 1, 1 << 63, (1 << 64) - 1, 0x7f, 0x80, 0x7fff, 0x8000, (1 << 64) - 0x80])
    pt = lambda: (rf(), rf())
    pair = lambda: ([rf()], [rf()])
    return {"alpha": c128(), "beta": c128(), "gamma": c128(), "zeta": c128(), "joint_combiner": c128() if rng.randrange(2) else None,
            "feature_flags": [bool(rng.randrange(2)) for _ in range(8)], "bulletproof_challenges": [c128() for _ in range(16)],
            "proofs_verified": rng.randrange(3), "domain_log2": rng.randrange(10, 17), "sponge_digest": [rng.getrandbits(64) for _ in range(4)],
            "challenge_polynomial_commitment": pt(), "old_bulletproof_challenges": [[c128() for _ in range(15)] for _ in range(2)],
            "step_comms": [pt() for _ in range(2)], "step_old_chals": [[c128() for _ in range(16)] for _ in range(2)],
            "prev_public_input": (rf(), rf()), "prev_evals": [pair() for _ in range(43)], "prev_optional": [pair() if rng.randrange(3) == 0 else None for _ in range(19)],
            "prev_ft_eval1": rf(), "w_comm": [pt() for _ in range(15)], "z_comm": pt(), "t_comm": [pt() for _ in range(7)],
            "w_eval": [(rf(), rf()) for _ in range(15)], "coefficients_eval": [(rf(), rf()) for _ in range(15)], "z_eval": (rf(), rf()),
            "s_eval": [(rf(), rf()) for _ in range(6)], "selector_eval": [(rf(), rf()) for _ in range(6)], "ft_eval1": rf(),
            "lr": [(pt(), pt()) for _ in range(k)], "z1": rf(), "z2": rf(), "delta": pt(), "sg": pt()}


def wrap_proof_bytes(d: dict, binprot: bool) -> bytes:
    w = W(binprot)
    for name in ("alpha", "beta", "gamma", "zeta"):
        w.chal(d[name])
    w.option(d["joint_combiner"] is not None)
    if d["joint_combiner"] is not None:
        w.chal(d["joint_combiner"])
    for f in d["feature_flags"]:
        w.boolean(f)
    for c in d["bulletproof_challenges"]:
        w.chal(c)
    w.padded_end()
    w.variant(d["proofs_verified"]); w.chr(d["domain_log2"])
    for l in d["sponge_digest"]:
        w.i64(l)
    w.padded_end()
    w.pt(d["challenge_polynomial_commitment"])
    for row in d["old_bulletproof_challenges"]:
        for c in row:
            w.chal(c)
        w.padded_end()
    w.padded_end()
    w.unit()                                                     # app_state
    w.length(len(d["step_comms"]))
    for p in d["step_comms"]:
        w.pt(p)
    w.length(len(d["step_old_chals"]))
    for row in d["step_old_chals"]:
        for c in row:
            w.chal(c)
        w.padded_end()
    w.big(d["prev_public_input"][0]); w.big(d["prev_public_input"][1])

    def vecpair(e):
        for side in e:
            w.length(len(side))
            for x in side:
                w.big(x)
    ev = d["prev_evals"]
    for e in ev[0:15]: vecpair(e)
    w.padded_end()
    for e in ev[15:30]: vecpair(e)
    w.padded_end()
    vecpair(ev[30])
    for e in ev[31:37]: vecpair(e)
    w.padded_end()
    for e in ev[37:43]: vecpair(e)
    opt = d["prev_optional"]

    def optional(e):
        w.option(e is not None)
        if e is not None:
            vecpair(e)
    for e in opt[0:6]: optional(e)
    optional(opt[6]); optional(opt[7])
    for e in opt[8:13]: optional(e)
    w.padded_end()
    for e in opt[13:19]: optional(e)
    w.big(d["prev_ft_eval1"])
    for p in d["w_comm"]: w.pt(p)
    w.padded_end()
    w.pt(d["z_comm"])
    for p in d["t_comm"]: w.pt(p)
    w.padded_end()
    for a, b in d["w_eval"]: w.big(a); w.big(b)
    w.padded_end()
    for a, b in d["coefficients_eval"]: w.big(a); w.big(b)
    w.padded_end()
    w.big(d["z_eval"][0]); w.big(d["z_eval"][1])
    for a, b in d["s_eval"]: w.big(a); w.big(b)
    w.padded_end()
    for a, b in d["selector_eval"]: w.big(a); w.big(b)
    w.big(d["ft_eval1"])
    w.length(len(d["lr"]))
    for l, r in d["lr"]:
        w.pt(l); w.pt(r)
    w.big(d["z1"]); w.big(d["z2"]); w.pt(d["delta"]); w.pt(d["sg"])
    return bytes(w.o)


def wrap_proof_flat(d: dict) -> bytes:
    """the flat layout mina_wrap_proof_flatten documents (include/mina_verify.h)"""
    o = bytearray()
    c16 = lambda c: int(c).to_bytes(16, "little")
    b32 = lambda x: int(x).to_bytes(32, "little")
    pt = lambda p: b32(p[0]) + b32(p[1])
    u32 = lambda v: struct.pack("<I", v)
    for name in ("alpha", "beta", "gamma", "zeta"):
        o += c16(d[name])
    o += bytes([d["joint_combiner"] is not None]) + c16(d["joint_combiner"] or 0)
    o += bytes(d["feature_flags"])
    for c in d["bulletproof_challenges"]: o += c16(c)
    o += bytes([d["proofs_verified"], d["domain_log2"]])
    for l in d["sponge_digest"]: o += struct.pack("<Q", l)
    o += pt(d["challenge_polynomial_commitment"])
    for row in d["old_bulletproof_challenges"]:
        for c in row: o += c16(c)
    o += u32(len(d["step_comms"]))
    for p in d["step_comms"]: o += pt(p)
    o += u32(len(d["step_old_chals"]))
    for row in d["step_old_chals"]:
        for c in row: o += c16(c)

    def ev(e):
        r = bytearray()
        for side in e:
            r += u32(len(side))
            for x in side: r += b32(x)
        return r
    o += ev(([d["prev_public_input"][0]], [d["prev_public_input"][1]]))
    present = [e for e in d["prev_optional"] if e is not None]
    # optional evaluations are appended in wire order, interleaved after the fixed 43 -- same order as the reader keeps them
    o += u32(43 + len(present))
    for e in d["prev_evals"]: o += ev(e)
    for e in present: o += ev(e)
    o += bytes([e is not None for e in d["prev_optional"]])
    o += b32(d["prev_ft_eval1"])
    for p in d["w_comm"]: o += pt(p)
    o += pt(d["z_comm"])
    for p in d["t_comm"]: o += pt(p)
    for name in ("w_eval", "coefficients_eval"):
        for a, b in d[name]: o += b32(a) + b32(b)
    o += b32(d["z_eval"][0]) + b32(d["z_eval"][1])
    for name in ("s_eval", "selector_eval"):
        for a, b in d[name]: o += b32(a) + b32(b)
    o += b32(d["ft_eval1"])
    o += u32(len(d["lr"]))
    for l, r in d["lr"]: o += pt(l) + pt(r)
    o += b32(d["z1"]) + b32(d["z2"]) + pt(d["delta"]) + pt(d["sg"])
    return bytes(o)


def state_proof_bytes(wrap: dict, states: list) -> bytes:
    """bincode MinaStateProof: tip proof, [ProtocolState; 16], bridge tip state (17 state dicts of oracle/mina_state_ref.py)"""
    from test_protocol_state import bincode_state
    assert len(states) == 17
    return wrap_proof_bytes(wrap, binprot=False) + b"".join(bincode_state(s) for s in states)


def state_pub_bytes(devnet: bool, bridge_tip_hash: int, state_hashes: list, ledger_hashes: list) -> bytes:
    b32 = lambda x: int(x).to_bytes(32, "little")
    return bytes([1 if devnet else 0]) + b32(bridge_tip_hash) + b"".join(b32(h) for h in state_hashes) + b"".join(b32(h) for h in ledger_hashes)
