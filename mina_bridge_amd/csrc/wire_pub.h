// wire_pub.h -- readers of the fixed layouts either side of the hot path: plain C++17, no HIP (also built
// standalone under sanitizers by tests/fuzz/).  Layouts (bincode 1.3 of the serde derives in the reference):
//   MinaStatePubInputs   core/src/proof/state_proof.rs:10-25 + sol/serialization.rs:13-61  -- exactly 1057 bytes
//   MinaAccountPubInputs core/src/proof/account_proof.rs:18-25 + sol/serialization.rs:63-86 -- ledger hash || u64-LE length || ABI bytes
//   MinaAccountProof     core/src/proof/account_proof.rs:9-14,30-35 -- merkle path (u64 count; per node u32 variant + u64 32 + 32 B), then the account
// Every function returns MINA_OK or MINA_ERR_FORMAT with a static reason in `*why`.
#pragma once
#include <cstdint>
#include <cstring>

#include "../../include/mina_verify.h"
#include "wire_state.h"

namespace mw {
static inline uint64_t pub_rd_u64(const uint8_t *p) { uint64_t v = 0; for (int i = 7; i >= 0; --i) v = (v << 8) | p[i]; return v; }
static inline uint32_t pub_rd_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

static inline int parse_state_pub_inputs(const uint8_t *bytes, size_t len, mina_state_pub_inputs *out, const char **why) {
    if (len != 1057) { *why = "MinaStatePubInputs must be exactly 1057 bytes"; return MINA_ERR_FORMAT; }
    if (bytes[0] > 1) { *why = "bool byte must be 0 or 1"; return MINA_ERR_FORMAT; }
    for (int i = 0; i < 33; ++i)
        if (!fp_canonical(bytes + 1 + 32 * i)) { *why = "hash is not a canonical field element"; return MINA_ERR_FORMAT; }
    out->is_state_proof_from_devnet = bytes[0];
    memcpy(out->bridge_tip_state_hash, bytes + 1, 32);
    memcpy(out->candidate_chain_state_hashes, bytes + 33, 512);
    memcpy(out->candidate_chain_ledger_hashes, bytes + 545, 512);
    return MINA_OK;
}

static inline int parse_account_pub_inputs(const uint8_t *bytes, size_t len, uint8_t *ledger_hash, size_t *encoded_offset, size_t *encoded_len, const char **why) {
    if (len < 40) { *why = "MinaAccountPubInputs shorter than 40 bytes"; return MINA_ERR_FORMAT; }
    if (!fp_canonical(bytes)) { *why = "ledger hash is not a canonical field element"; return MINA_ERR_FORMAT; }
    const uint64_t n = pub_rd_u64(bytes + 32);
    if (n != len - 40) { *why = "encoded_account length prefix does not match the buffer"; return MINA_ERR_FORMAT; }
    memcpy(ledger_hash, bytes, 32);
    *encoded_offset = 40; *encoded_len = (size_t)n;
    return MINA_OK;
}

static inline int parse_merkle_path(const uint8_t *proof, size_t len, uint32_t max_depth, uint8_t *siblings, uint8_t *dirs, uint32_t *depth, size_t *account_offset, const char **why) {
    if (len < 8) { *why = "MinaAccountProof shorter than its length prefix"; return MINA_ERR_FORMAT; }
    const uint64_t n = pub_rd_u64(proof);
    if (n > max_depth) { *why = "merkle path longer than max_depth"; return MINA_ERR_FORMAT; }
    size_t off = 8;
    for (uint64_t i = 0; i < n; ++i) {
        if (len - off < 4 + 8 + 32) { *why = "truncated merkle node"; return MINA_ERR_FORMAT; }
        const uint32_t tag = pub_rd_u32(proof + off);
        if (tag > 1) { *why = "MerkleNode variant must be 0 (Left) or 1 (Right)"; return MINA_ERR_FORMAT; }
        if (pub_rd_u64(proof + off + 4) != 32) { *why = "MerkleNode field element must be 32 bytes"; return MINA_ERR_FORMAT; }
        if (!fp_canonical(proof + off + 12)) { *why = "merkle node is not a canonical field element"; return MINA_ERR_FORMAT; }
        dirs[i] = (uint8_t)tag;
        memcpy(siblings + 32 * i, proof + off + 12, 32);
        off += 44;
    }
    *depth = (uint32_t)n; *account_offset = off;
    return MINA_OK;
}
}  // namespace mw
