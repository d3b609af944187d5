// wire_account.h -- host-side reader of the Mina account record, its Solidity-ABI re-encoding and its `to_input` flattening
// (8f-3).
//
// Replaces, for the verifier behind core/src/aligned.rs:46-58 (`ProvingSystemId::MinaAccount`):
//   * bincode `deserialize::<MinaAccountProof>` (core/src/proof/account_proof.rs:30-35): merkle path + `MinaBaseAccountBinableArgStableV2`
//     (also readable in its bin_prot form, as core/src/mina.rs:307-313 receives it from the node)                  [UPSTREAM-RECALL]
//   * the cross-check README.md:349-352 asks of the verifier: re-derive `encoded_account` from `account` and compare bytes --
//     `impl TryFrom<&MinaAccount> for Account` (core/src/sol/account.rs:25-314) + `abi_encode`, decoded on-chain by
//     `abi.decode(pubInput[40:], (Account))` (contract/src/MinaAccountValidationExample.sol:70,76-164)             [PINNED BY THE REFERENCE]
//   * openmina `Account::hash` inputs: account / zkapp / verification-key / zkapp-uri `to_inputs`                  [UPSTREAM-RECALL]
#pragma once
#include "wire_state.h"

namespace mw {

struct CompressedPk { B32 x{}; bool is_odd = false; };
struct Timing { bool timed = false; uint64_t initial_minimum_balance = 0, cliff_amount = 0, vesting_increment = 0; uint32_t cliff_time = 0, vesting_period = 0; };
struct Commitment { B32 x{}, y{}; };
struct VerificationKey { uint8_t max_proofs_verified = 0, actual_wrap_domain_size = 0; Commitment sigma[7], coefficients[15], other[6]; };
struct ZkappAccount { B32 app_state[8]{}; bool has_vk = false; VerificationKey vk; uint32_t zkapp_version = 0; B32 action_state[5]{}; uint32_t last_action_slot = 0;
                      bool proved_state = false; std::vector<uint8_t> zkapp_uri; };
struct Account {
    CompressedPk public_key; B32 token_id{}; std::vector<uint8_t> token_symbol; uint64_t balance = 0; uint32_t nonce = 0; B32 receipt_chain_hash{};
    bool has_delegate = false; CompressedPk delegate; B32 voting_for{}; Timing timing;
    uint8_t perm[13]{}; uint32_t set_vk_txn_version = 0;       // order: edit_state, access, send, receive, set_delegate, set_permissions, set_verification_key,
                                                               // set_zkapp_uri, edit_action_state, set_token_symbol, increment_nonce, set_voting_for, set_timing
    bool has_zkapp = false; ZkappAccount zkapp;
};

template <class C> static uint8_t rd_tag(C &c, uint32_t max) { const uint32_t v = c.variant(); if (v > max) c.fail(); return (uint8_t)v; }
template <class C> static bool read_account(C &c, Account &a) {
    a.public_key.x = c.big(); a.public_key.is_odd = c.boolean();
    a.token_id = c.big(); a.token_symbol = c.string(); a.balance = c.u64(); a.nonce = c.u32(); a.receipt_chain_hash = c.big();
    a.has_delegate = c.option(); if (a.has_delegate) { a.delegate.x = c.big(); a.delegate.is_odd = c.boolean(); }
    a.voting_for = c.big();
    a.timing.timed = rd_tag(c, 1) == 1;
    if (a.timing.timed) {
        a.timing.initial_minimum_balance = c.u64(); if (c.variant() != 0) c.fail(); a.timing.cliff_time = c.u32(); a.timing.cliff_amount = c.u64();
        if (c.variant() != 0) c.fail(); a.timing.vesting_period = c.u32(); a.timing.vesting_increment = c.u64();
    }
    for (int i = 0; i < 13; ++i) { a.perm[i] = rd_tag(c, 4); if (i == 6) a.set_vk_txn_version = c.u32(); }
    a.has_zkapp = c.option();
    if (a.has_zkapp) {
        ZkappAccount &z = a.zkapp;
        for (int i = 0; i < 8; ++i) z.app_state[i] = c.big();
        c.padded_end();
        z.has_vk = c.option();
        if (z.has_vk) {
            z.vk.max_proofs_verified = rd_tag(c, 2); z.vk.actual_wrap_domain_size = rd_tag(c, 2);
            for (int i = 0; i < 7; ++i) { z.vk.sigma[i].x = c.big(); z.vk.sigma[i].y = c.big(); }
            c.padded_end();
            for (int i = 0; i < 15; ++i) { z.vk.coefficients[i].x = c.big(); z.vk.coefficients[i].y = c.big(); }
            c.padded_end();
            for (int i = 0; i < 6; ++i) { z.vk.other[i].x = c.big(); z.vk.other[i].y = c.big(); }
        }
        z.zkapp_version = c.u32();
        for (int i = 0; i < 5; ++i) z.action_state[i] = c.big();
        c.padded_end();
        if (c.variant() != 0) c.fail(); z.last_action_slot = c.u32(); z.proved_state = c.boolean(); z.zkapp_uri = c.string();
    }
    if (!c.ok || a.token_symbol.size() > 6 || a.zkapp.zkapp_uri.size() > 255) return false;     // Token_symbol.max_length = 6, Zkapp_uri max 255
    const B32 *fes[] = {&a.public_key.x, &a.token_id, &a.receipt_chain_hash, &a.voting_for};
    for (const B32 *f : fes) if (!fp_canonical(f->b)) return false;
    if (a.has_delegate && !fp_canonical(a.delegate.x.b)) return false;
    if (a.has_zkapp) {
        for (int i = 0; i < 8; ++i) if (!fp_canonical(a.zkapp.app_state[i].b)) return false;
        for (int i = 0; i < 5; ++i) if (!fp_canonical(a.zkapp.action_state[i].b)) return false;
        if (a.zkapp.has_vk) {
            const VerificationKey &v = a.zkapp.vk;
            for (int i = 0; i < 7; ++i) if (!fp_canonical(v.sigma[i].x.b) || !fp_canonical(v.sigma[i].y.b)) return false;
            for (int i = 0; i < 15; ++i) if (!fp_canonical(v.coefficients[i].x.b) || !fp_canonical(v.coefficients[i].y.b)) return false;
            for (int i = 0; i < 6; ++i) if (!fp_canonical(v.other[i].x.b) || !fp_canonical(v.other[i].y.b)) return false;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------- Solidity ABI
// `Account::abi_encode()` as sol/account.rs builds it: bytes32 = the stored 32 bytes as they are; uintN / bool / enum = one
// big-endian word; dynamic members (string tokenSymbol, ZkappAccount with its bytes zkappUri) = offset word + tail.
struct AbiWriter {
    std::vector<uint8_t> o;
    void word_uint(uint64_t v) { uint8_t w[32] = {0}; for (int i = 0; i < 8; ++i) w[31 - i] = (uint8_t)(v >> (8 * i)); o.insert(o.end(), w, w + 32); }
    void word_b32(const B32 &x) { o.insert(o.end(), x.b, x.b + 32); }
    void dyn_bytes(const std::vector<uint8_t> &b) { word_uint(b.size()); o.insert(o.end(), b.begin(), b.end()); o.insert(o.end(), (32 - b.size() % 32) % 32, 0); }
};
static inline void abi_encode_account(const Account &a, std::vector<uint8_t> &out) {
    static const ZkappAccount EMPTY_ZKAPP;                       // sol/account.rs:270-299: zeros, empty uri
    const ZkappAccount &z = a.has_zkapp ? a.zkapp : EMPTY_ZKAPP;
    static const VerificationKey EMPTY_VK;                       // sol/account.rs:213-240
    const VerificationKey &vk = z.has_vk ? z.vk : EMPTY_VK;
    AbiWriter zk;
    for (int i = 0; i < 8; ++i) zk.word_b32(z.app_state[i]);
    zk.word_uint(vk.max_proofs_verified); zk.word_uint(vk.actual_wrap_domain_size);
    for (int i = 0; i < 7; ++i) { zk.word_b32(vk.sigma[i].x); zk.word_b32(vk.sigma[i].y); }
    for (int i = 0; i < 15; ++i) { zk.word_b32(vk.coefficients[i].x); zk.word_b32(vk.coefficients[i].y); }
    for (int i = 0; i < 6; ++i) { zk.word_b32(vk.other[i].x); zk.word_b32(vk.other[i].y); }
    zk.word_uint(z.zkapp_version);
    for (int i = 0; i < 5; ++i) zk.word_b32(z.action_state[i]);
    zk.word_uint(z.last_action_slot); zk.word_uint(z.proved_state ? 1 : 0);
    zk.word_uint(zk.o.size() + 32);                              // offset of zkappUri inside the ZkappAccount tuple
    zk.dyn_bytes(z.zkapp_uri);
    AbiWriter sym; sym.dyn_bytes(a.token_symbol);
    AbiWriter w;
    w.word_uint(32);                                             // the struct is dynamic: one offset word in front
    constexpr size_t HEAD_WORDS = 30;
    w.word_b32(a.public_key.x); w.word_uint(a.public_key.is_odd ? 1 : 0); w.word_b32(a.token_id);
    w.word_uint(HEAD_WORDS * 32);
    w.word_uint(a.balance); w.word_uint(a.nonce); w.word_b32(a.receipt_chain_hash);
    if (a.has_delegate) { w.word_b32(a.delegate.x); w.word_uint(a.delegate.is_odd ? 1 : 0); }
    else { w.word_b32(B32{}); w.word_uint(1); }                  // sol/account.rs:64-69: absent delegate = (zero, isOdd = true)
    w.word_b32(a.voting_for);
    w.word_uint(a.timing.timed ? a.timing.initial_minimum_balance : 0); w.word_uint(a.timing.timed ? a.timing.cliff_time : 0);
    w.word_uint(a.timing.timed ? a.timing.cliff_amount : 0); w.word_uint(a.timing.timed ? a.timing.vesting_period : 0); w.word_uint(a.timing.timed ? a.timing.vesting_increment : 0);
    for (int i = 0; i < 13; ++i) { w.word_uint(a.perm[i]); if (i == 6) w.word_uint(a.set_vk_txn_version); }
    w.word_uint(HEAD_WORDS * 32 + sym.o.size());
    out = w.o;
    out.insert(out.end(), sym.o.begin(), sym.o.end());
    out.insert(out.end(), zk.o.begin(), zk.o.end());
}

// ---------------------------------------------------------------------------------------------- to_input of the four hashes
static inline void auth_bits(Inputs &in, uint8_t tag) {          // (constant, signature_necessary, signature_sufficient)
    static const uint8_t T[5][3] = {{1, 0, 1}, {0, 0, 1}, {0, 0, 0}, {0, 1, 1}, {1, 1, 0}};
    for (int i = 0; i < 3; ++i) in.boolean(T[tag][i]);
}
static inline void zkapp_uri_fields(const std::vector<uint8_t> &uri, std::vector<B32> &out) {
    Inputs in; in.bytes_lsb_first(uri.data(), uri.size(), uri.size() * 8); in.boolean(true); in.to_fields(out);
}
static inline const VerificationKey &dummy_vk() {                // [UPSTREAM-RECALL] stand-in for `VerificationKey::dummy()`
    static VerificationKey v = [] { VerificationKey d; d.max_proofs_verified = 2; d.actual_wrap_domain_size = 2; Commitment g; g.x.b[0] = 1; g.y.b[0] = 2;
                                    for (auto &c : d.sigma) c = g; for (auto &c : d.coefficients) c = g; for (auto &c : d.other) c = g; return d; }();
    return v;
}
static inline void vk_fields(const VerificationKey &v, std::vector<B32> &out) {
    Inputs in;
    for (int i = 0; i < 3; ++i) in.boolean(i == v.max_proofs_verified);
    for (int i = 0; i < 3; ++i) in.boolean(i == v.actual_wrap_domain_size);
    for (int i = 0; i < 7; ++i) { in.field(v.sigma[i].x); in.field(v.sigma[i].y); }
    for (int i = 0; i < 15; ++i) { in.field(v.coefficients[i].x); in.field(v.coefficients[i].y); }
    for (int i = 0; i < 6; ++i) { in.field(v.other[i].x); in.field(v.other[i].y); }
    in.to_fields(out);
}
// slot 0 = zkapp-uri hash, slot 6 = verification-key hash: patched in on the GPU
static constexpr int ZK_SLOT_URI = 0, ZK_SLOT_VK = 6;
static inline void zkapp_fields(const ZkappAccount &z, std::vector<B32> &out) {
    Inputs in;
    in.field(B32{}); in.boolean(z.proved_state); in.u32(z.last_action_slot);
    for (int i = 0; i < 5; ++i) in.field(z.action_state[i]);
    in.u32(z.zkapp_version); in.field(B32{});
    for (int i = 0; i < 8; ++i) in.field(z.app_state[i]);
    in.to_fields(out);
}
// slot 0 = zkapp hash: patched in on the GPU
static inline void account_fields(const Account &a, std::vector<B32> &out) {
    Inputs in;
    in.field(B32{});
    for (int i = 0; i < 13; ++i) { auth_bits(in, a.perm[i]); if (i == 6) in.u32(a.set_vk_txn_version); }
    const Timing &t = a.timing;
    if (!t.timed) { in.boolean(false); in.u64(0); in.u32(0); in.u64(0); in.u32(1); in.u64(0); }
    else { in.boolean(true); in.u64(t.initial_minimum_balance); in.u32(t.cliff_time); in.u64(t.cliff_amount); in.u32(t.vesting_period); in.u64(t.vesting_increment); }
    in.field(a.voting_for);
    if (a.has_delegate) { in.field(a.delegate.x); in.boolean(a.delegate.is_odd); } else { in.field(B32{}); in.boolean(false); }
    in.field(a.receipt_chain_hash); in.u32(a.nonce); in.u64(a.balance);
    { uint64_t s = 0; for (size_t i = 0; i < a.token_symbol.size() && i < 6; ++i) s |= (uint64_t)a.token_symbol[i] << (8 * i); in.packed(s, 48); }
    in.field(a.token_id); in.field(a.public_key.x); in.boolean(a.public_key.is_odd);
    in.to_fields(out);
}

}  // namespace mw
