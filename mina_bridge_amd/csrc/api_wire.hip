// api_wire.hip -- the in-tree wire formats either side of the hot path and the
// Proof-of-Account inclusion check built on the GPU Merkle fold (a16, inclusion part).
//
// Byte layouts (bincode 1.3 of the serde derives in the reference):
//   MinaStatePubInputs   core/src/proof/state_proof.rs:10-25 + sol/serialization.rs:13-61  -- exactly 1057 bytes:
//       [0] bool, [1..33) bridge tip state hash, [33..545) 16 state hashes, [545..1057) 16 ledger hashes
//       (offsets double-pinned by MinaStateSettlementExample.sol:92,100,130-135)
//   MinaAccountPubInputs core/src/proof/account_proof.rs:18-25 + sol/serialization.rs:63-86
//       ledger_hash (32 B LE Fp) || u64-LE length || ABI-encoded account   (pinned by MinaAccountValidationExample.sol:70)
//   MinaAccountProof     core/src/proof/account_proof.rs:9-14,30-35
//       merkle_path: u64-LE count, then per node u32-LE variant (0 = Left, 1 = Right) + SerdeAs<Fp> = u64-LE 32 + 32 B
//       ([UPSTREAM-RECALL] for the SerdeAs framing), then the binprot-derived account (not parsed here).
// Field elements must be canonical (< p), as ark's `CanonicalDeserialize` enforces.
#include "ctx.h"
#include "wire_pub.h"
#include "wire_state.h"
#include <type_traits>

static bool fp_is_canonical(const uint8_t *b) { return mw::fp_canonical(b); }

extern "C" int mina_parse_state_pub_inputs(const uint8_t *bytes, size_t len, mina_state_pub_inputs *out) {
    if (!bytes || !out) return fail(MINA_ERR_ARG, "null argument");
    const char *why = ""; const int rc = mw::parse_state_pub_inputs(bytes, len, out, &why);
    return rc ? fail(rc, why) : MINA_OK;
}

extern "C" int mina_parse_account_pub_inputs(const uint8_t *bytes, size_t len, uint8_t *ledger_hash, size_t *encoded_offset, size_t *encoded_len) {
    if (!bytes || !ledger_hash || !encoded_offset || !encoded_len) return fail(MINA_ERR_ARG, "null argument");
    const char *why = ""; const int rc = mw::parse_account_pub_inputs(bytes, len, ledger_hash, encoded_offset, encoded_len, &why);
    return rc ? fail(rc, why) : MINA_OK;
}

extern "C" int mina_parse_merkle_path(const uint8_t *proof, size_t len, uint32_t max_depth, uint8_t *siblings, uint8_t *dirs,
                                      uint32_t *depth, size_t *account_offset) {
    if (!proof || !siblings || !dirs || !depth || !account_offset) return fail(MINA_ERR_ARG, "null argument");
    const char *why = ""; const int rc = mw::parse_merkle_path(proof, len, max_depth, siblings, dirs, depth, account_offset, &why);
    return rc ? fail(rc, why) : MINA_OK;
}

extern "C" int mina_verify_account_inclusion(mina_ctx *c, size_t n, const uint8_t *const *proofs, const size_t *proof_lens,
                                             const uint8_t *const *pubs, const size_t *pub_lens, const uint8_t *leaf_hashes,
                                             uint8_t *verdicts) {
    if (!c || (n && (!proofs || !proof_lens || !pubs || !pub_lens || !leaf_hashes || !verdicts))) return fail(MINA_ERR_ARG, "null argument");
    if (n == 0) return MINA_OK;
    // parse everything on the host; malformed entries get verdict 0 and are left out of the GPU batch.
    // Paths of one depth go to the GPU together (ledger depth is fixed per network, so normally one group).
    constexpr uint32_t MAXD = 64;
    std::vector<uint8_t> sib(n * MAXD * 32), dir(n * MAXD), ledger(n * 32);
    std::vector<uint32_t> depth(n, 0); std::vector<uint8_t> ok(n, 0);
    for (size_t i = 0; i < n; ++i) {
        verdicts[i] = 0;
        size_t acc_off = 0, eo = 0, el = 0;
        if (!proofs[i] || !pubs[i]) continue;
        if (mina_parse_merkle_path(proofs[i], proof_lens[i], MAXD, &sib[i * MAXD * 32], &dir[i * MAXD], &depth[i], &acc_off) != MINA_OK) continue;
        if (mina_parse_account_pub_inputs(pubs[i], pub_lens[i], &ledger[i * 32], &eo, &el) != MINA_OK) continue;
        if (!fp_is_canonical(leaf_hashes + 32 * i)) continue;
        ok[i] = 1;
    }
    for (uint32_t d = 0; d <= MAXD; ++d) {
        std::vector<size_t> idx;
        for (size_t i = 0; i < n; ++i) if (ok[i] && depth[i] == d) idx.push_back(i);
        if (idx.empty()) continue;
        const size_t m = idx.size();
        std::vector<uint8_t> l(m * 32), s((size_t)m * d * 32 + 1), dd((size_t)m * d + 1), r(m * 32), v(m);
        for (size_t j = 0; j < m; ++j) {
            memcpy(&l[j * 32], leaf_hashes + 32 * idx[j], 32);
            if (d) { memcpy(&s[j * d * 32], &sib[idx[j] * MAXD * 32], (size_t)d * 32); memcpy(&dd[j * d], &dir[idx[j] * MAXD], d); }
            memcpy(&r[j * 32], &ledger[idx[j] * 32], 32);
        }
        int rc = mina_merkle_verify_batch(c, MINA_FIELD_FP, m, d, l.data(), s.data(), dd.data(), r.data(), v.data());
        if (rc) return rc;
        for (size_t j = 0; j < m; ++j) verdicts[idx[j]] = v[j];
    }
    return MINA_OK;
}

// poly-commitment `combined_inner_product` for single-chunk polynomials without degree bounds (a11):
//   cip = sum_i xi^i * sum_j r^j * evals[i][j]      (evals[i][j] = f_i(point_j))
// O(n_polys * n_points) field operations: done on the host with the host build of fp.cuh.
#include "groupmap.cuh"
extern "C" int mina_combined_inner_product(int field, size_t n_polys, size_t n_points, const uint8_t *evals, const uint8_t *polyscale,
                                           const uint8_t *evalscale, uint8_t *out) {
    if ((n_polys && n_points && !evals) || !polyscale || !evalscale || !out) return fail(MINA_ERR_ARG, "null argument");
    if (field != 0 && field != 1) return fail(MINA_ERR_ARG, "bad field");
    auto run = [&](auto tag) {
        constexpr int F = decltype(tag)::value;
        fe_t one = fe_zero(); one.v[0] = 1;
        fe_t r2 = one;                                               // R^2 mod p by doubling (no context needed here)
        for (int i = 0; i < 512; ++i) r2 = fe_add<F>(r2, r2);
        auto ld = [&](const uint8_t *b) { fe_t a; memcpy(a.v, b, 32); return fe_to_mont<F>(a, r2); };
        const fe_t xi = ld(polyscale), r = ld(evalscale);
        fe_t res = fe_zero(), xi_i = fe_to_mont<F>(one, r2);
        for (size_t i = 0; i < n_polys; ++i) {
            fe_t term = fe_zero();
            for (size_t j = n_points; j-- > 0;) term = fe_add<F>(fe_mul<F>(term, r), ld(evals + (i * n_points + j) * 32));   // Horner in r
            res = fe_add<F>(res, fe_mul<F>(xi_i, term));
            xi_i = fe_mul<F>(xi_i, xi);
        }
        res = fe_from_mont<F>(res);
        memcpy(out, res.v, 32);
    };
    if (field == FIELD_FP) run(std::integral_constant<int, FIELD_FP>{}); else run(std::integral_constant<int, FIELD_FQ>{});
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------ host: pack one protocol state (moved here from api_state.hip in round 5: parser-side, no kernel)
static void info_from_state(const mw::ProtocolState &s, mina_protocol_state_info *info, uint32_t nf) {
    memset(info, 0, sizeof *info);
    memcpy(info->previous_state_hash, s.previous_state_hash.b, 32);
    memcpy(info->genesis_state_hash, s.genesis_state_hash.b, 32);
    memcpy(info->snarked_ledger_hash, s.snarked_ledger_hash().b, 32);
    info->n_body_fields = nf;
    info->k = s.k; info->slots_per_epoch = s.c_slots_per_epoch; info->slots_per_sub_window = s.slots_per_sub_window;
    info->sub_windows_per_window = (uint32_t)s.sub_window_densities.size(); info->grace_period_slots = s.grace_period_slots; info->delta = s.delta;
    mina_consensus_state &c = info->consensus;
    c.blockchain_length = s.blockchain_length; c.epoch_count = s.epoch_count; c.curr_global_slot = s.slot_number; c.min_window_density = s.min_window_density;
    for (size_t i = 0; i < s.sub_window_densities.size() && i < MINA_MAX_SUB_WINDOWS; ++i) c.sub_window_densities[i] = s.sub_window_densities[i];
    memcpy(c.staking_lock_checkpoint, s.staking.lock_checkpoint.b, 32);
    memcpy(c.next_lock_checkpoint, s.next.lock_checkpoint.b, 32);
    memcpy(c.last_vrf_output_hash, s.last_vrf_output.data(), 32);          // the truncated VRF output string itself (compared lexicographically)
}

int mb_pack_protocol_state(const mw::ProtocolState &s, uint8_t *record, uint32_t *n_body_fields, mina_protocol_state_info *info) {
    mw::Inputs in;
    mw::protocol_state_body_inputs(s, in);
    const size_t nf = in.count();
    if (in.overflow || nf > MINA_PSTATE_SLOTS - 1) return fail(MINA_ERR_FORMAT, "protocol state body flattens to more field elements than a record holds");
    memcpy(record, s.previous_state_hash.b, 32);
    in.write(record + 32);
    memset(record + 32 * (1 + nf), 0, (size_t)(MINA_PSTATE_SLOTS - 1 - nf) * 32);
    *n_body_fields = (uint32_t)nf;
    if (info) info_from_state(s, info, (uint32_t)nf);
    return MINA_OK;
}

extern "C" int mina_protocol_state_pack(const uint8_t *bytes, size_t len, int encoding, uint8_t *record, uint32_t *n_body_fields,
                                        mina_protocol_state_info *info, size_t *consumed) {
    if (!bytes || !record || !n_body_fields) return fail(MINA_ERR_ARG, "null argument");
    mw::ProtocolState s; bool ok; size_t used;
    if (encoding == MINA_ENC_BINPROT) { mw::Binprot c(bytes, len); ok = mw::read_protocol_state(c, s); used = c.pos; }
    else if (encoding == MINA_ENC_BINCODE) { mw::Bincode c(bytes, len); ok = mw::read_protocol_state(c, s); used = c.pos; }
    else return fail(MINA_ERR_ARG, "encoding must be MINA_ENC_BINPROT or MINA_ENC_BINCODE");
    if (!ok) return fail(MINA_ERR_FORMAT, "malformed protocol state");
    if (consumed) *consumed = used; else if (used != len) return fail(MINA_ERR_FORMAT, "trailing bytes after the protocol state");
    return mb_pack_protocol_state(s, record, n_body_fields, info);
}

