// wire_proof.h -- host-side readers of the Pickles wrap proof container and of `MinaStateProof`.
//
// Replaces, for the verifier behind core/src/aligned.rs:31-58:
//   * `MinaBaseProofStableV2::binprot_read` (core/src/mina.rs:235-248: the tip's `protocolStateProof` from the node)
//   * bincode `deserialize::<MinaStateProof>` (core/src/proof/state_proof.rs:28-41): the same proof in its serde form, then
//     16 + 1 protocol states (wire_state.h)
// Record layout: mina-p2p-messages v2 `PicklesProofProofsVerified2ReprStableV2` (pin core/Cargo.toml:24, core/Cargo.lock:3812-3814).
// [UPSTREAM-RECALL] -- the crate is not vendored and the tree holds no serialized proof (`mina_state.proof` is git-ignored,
// .gitignore:3-6), so nothing here is pinned by reference bytes; sizes agree with SURVEY.md 8a a1 (proof ~ 11 KB, container ~ 37 KB).
//   bin_prot: records/tuples = fields in order; `PaddedSeq<T, N>` (OCaml `Vector.t`) = N elements then a unit byte 0;
//             lists = Nat0 length + elements; options = 0 / 1 + value; 64-bit limbs = variable-length signed ints;
//             plain variants = one tag byte.
//   bincode : records/tuples/arrays = fields in order, no terminators; Vec = u64 length; Option = u8; i64 limbs = 8 bytes;
//             enums = u32 variant index; unit = nothing.
#pragma once
#include "wire_state.h"

namespace mw {

struct Pt { B32 x, y; };                                  // affine point, coordinates as the container holds them (32-byte LE)
struct Chal128 { uint64_t lo, hi; };                      // two 64-bit limbs, least-significant first
struct EvalPair { SmallVec<B32, 16> zeta, zeta_omega; };   // chunked evaluations at zeta / zeta*omega (one chunk each in practice)

struct WrapProof {
    // statement.proof_state.deferred_values
    Chal128 alpha, beta, gamma, zeta; bool has_joint_combiner = false; Chal128 joint_combiner{};
    bool feature_flags[8] = {};                           // range_check0, range_check1, foreign_field_add, foreign_field_mul, xor, rot, lookup, runtime_tables
    Chal128 bulletproof_challenges[16];                   // step IPA prechallenges (accumulator check, a10)
    uint8_t proofs_verified = 0, domain_log2 = 0;         // branch_data
    uint64_t sponge_digest_before_evaluations[4];
    // statement.proof_state.messages_for_next_wrap_proof
    Pt challenge_polynomial_commitment;                   // the step accumulator `sg` (Vesta point: coordinates in Fq)
    Chal128 old_bulletproof_challenges[2][15];
    // statement.messages_for_next_step_proof
    SmallVec<Pt, 8> step_challenge_polynomial_commitments;           // Pallas points
    SmallVec<std::array<Chal128, 16>, 8> step_old_bulletproof_challenges;
    // prev_evals (evaluations of the step proof: Fp)
    EvalPair prev_public_input; SmallVec<EvalPair, 62> prev_evals;   // fixed order, see read_all_evals
    SmallVec<uint8_t, 32> prev_evals_present;                        // 1 per optional slot
    B32 prev_ft_eval1;
    // proof (wire form of the wrap ProverProof: commitments are Pallas points, evaluations in Fq)
    Pt w_comm[15], z_comm, t_comm[7];
    B32 w_eval[15][2], coefficients_eval[15][2], z_eval[2], s_eval[6][2], selector_eval[6][2];   // [zeta, zeta*omega]
    B32 ft_eval1;
    SmallVec<std::pair<Pt, Pt>, 32> lr; B32 z1, z2; Pt delta, sg;
};

template <class C> static Chal128 rd_chal(C &c) {          // PaddedSeq<Hex64, 2>
    Chal128 r; r.lo = (uint64_t)c.i64(); r.hi = (uint64_t)c.i64(); c.padded_end(); return r;
}
template <class C> static Pt rd_pt(C &c) { Pt p; p.x = c.big(); p.y = c.big(); return p; }
template <class C> static void rd_eval_vecs(C &c, EvalPair &e) {    // (Vec<BigInt>, Vec<BigInt>); at most 16 chunks a side (Mina's step proofs have one)
    e.zeta.clear(); e.zeta_omega.clear();
    for (int side = 0; side < 2; ++side) {
        const size_t m = c.length(); if (m > 16) { c.fail(); return; }
        SmallVec<B32, 16> &v = side ? e.zeta_omega : e.zeta;
        for (size_t i = 0; i < m && c.ok; ++i) v.push_back(c.big());
    }
}
static constexpr int N_PREV_FIXED = 15 + 15 + 1 + 6 + 6;     // w, coefficients, z, s, the six always-present selectors
static constexpr int N_PREV_OPTIONAL = 6 + 2 + 5 + 6;        // 6 optional gate selectors, lookup_aggregation/table, 5 lookup_sorted, 6 lookup selectors / runtime tables
template <class C> static void rd_all_evals(C &c, WrapProof &p) {
    p.prev_evals.clear(); p.prev_evals_present.clear();
    for (int i = 0; i < 15; ++i) rd_eval_vecs(c, p.prev_evals.emplace_back());
    c.padded_end();                                                              // w
    for (int i = 0; i < 15; ++i) rd_eval_vecs(c, p.prev_evals.emplace_back());
    c.padded_end();                                                              // coefficients
    rd_eval_vecs(c, p.prev_evals.emplace_back());                                // z
    for (int i = 0; i < 6; ++i) rd_eval_vecs(c, p.prev_evals.emplace_back());
    c.padded_end();                                                              // s
    for (int i = 0; i < 6; ++i) rd_eval_vecs(c, p.prev_evals.emplace_back());    // generic, poseidon, complete_add, mul, emul, endomul_scalar
    auto opt = [&]() { const bool some = c.option(); p.prev_evals_present.push_back(some ? 1 : 0); if (some) rd_eval_vecs(c, p.prev_evals.emplace_back()); };
    for (int i = 0; i < 6; ++i) opt();                                           // range_check0/1, foreign_field_add/mul, xor, rot selectors
    opt(); opt();                                                                // lookup_aggregation, lookup_table
    for (int i = 0; i < 5; ++i) opt();
    c.padded_end();                                                              // lookup_sorted
    for (int i = 0; i < 6; ++i) opt();                                           // runtime_lookup_table, runtime_lookup_table_selector, xor / lookup_gate / range_check / foreign_field_mul lookup selectors
}

template <class C> static bool read_wrap_proof(C &c, WrapProof &p) {
    // ---- statement.proof_state.deferred_values.plonk
    p.alpha = rd_chal(c); p.beta = rd_chal(c); p.gamma = rd_chal(c); p.zeta = rd_chal(c);
    p.has_joint_combiner = c.option(); if (p.has_joint_combiner) p.joint_combiner = rd_chal(c);
    for (int i = 0; i < 8; ++i) p.feature_flags[i] = c.boolean();
    for (int i = 0; i < 16; ++i) p.bulletproof_challenges[i] = rd_chal(c);
    c.padded_end();
    { const uint32_t v = c.variant(); if (v > 2) c.fail(); p.proofs_verified = (uint8_t)v; p.domain_log2 = c.chr(); }
    for (int i = 0; i < 4; ++i) p.sponge_digest_before_evaluations[i] = (uint64_t)c.i64();
    c.padded_end();
    p.challenge_polynomial_commitment = rd_pt(c);
    for (int a = 0; a < 2; ++a) { for (int i = 0; i < 15; ++i) p.old_bulletproof_challenges[a][i] = rd_chal(c); c.padded_end(); }
    c.padded_end();
    // ---- statement.messages_for_next_step_proof
    c.unit();                                                                    // app_state
    p.step_challenge_polynomial_commitments.clear(); p.step_old_bulletproof_challenges.clear(); p.lr.clear();
    { const size_t m = c.length(); if (m > 8) c.fail(); for (size_t i = 0; i < m && c.ok; ++i) p.step_challenge_polynomial_commitments.push_back(rd_pt(c)); }
    { const size_t m = c.length(); if (m > 8) c.fail();
      for (size_t i = 0; i < m && c.ok; ++i) { std::array<Chal128, 16> &a = p.step_old_bulletproof_challenges.emplace_back(); for (int j = 0; j < 16; ++j) a[j] = rd_chal(c); c.padded_end(); } }
    // ---- prev_evals
    { EvalPair &e = p.prev_public_input; e.zeta.clear(); e.zeta_omega.clear(); e.zeta.push_back(c.big()); e.zeta_omega.push_back(c.big()); }
    rd_all_evals(c, p);
    p.prev_ft_eval1 = c.big();
    // ---- proof.commitments
    for (int i = 0; i < 15; ++i) p.w_comm[i] = rd_pt(c);
    c.padded_end();
    p.z_comm = rd_pt(c);
    for (int i = 0; i < 7; ++i) p.t_comm[i] = rd_pt(c);
    c.padded_end();
    // ---- proof.evaluations
    auto pair2 = [&](B32 (&dst)[2]) { dst[0] = c.big(); dst[1] = c.big(); };
    for (int i = 0; i < 15; ++i) pair2(p.w_eval[i]);
    c.padded_end();
    for (int i = 0; i < 15; ++i) pair2(p.coefficients_eval[i]);
    c.padded_end();
    pair2(p.z_eval);
    for (int i = 0; i < 6; ++i) pair2(p.s_eval[i]);
    c.padded_end();
    for (int i = 0; i < 6; ++i) pair2(p.selector_eval[i]);
    p.ft_eval1 = c.big();
    // ---- proof.bulletproof
    { const size_t m = c.length(); if (m > 32) c.fail(); for (size_t i = 0; i < m && c.ok; ++i) { std::pair<Pt, Pt> &q = p.lr.emplace_back(); q.first = rd_pt(c); q.second = rd_pt(c); } }
    p.z1 = c.big(); p.z2 = c.big(); p.delta = rd_pt(c); p.sg = rd_pt(c);
    return c.ok && !p.prev_evals.overflow && !p.prev_evals_present.overflow;
}

// MinaStateProof (state_proof.rs:28-41), bincode: the proof, then [ProtocolState; 16], then the bridge tip state
struct StateProofContainer { WrapProof tip_proof; ProtocolState states[17]; };   // [0..16) candidate chain oldest..tip, [16] bridge tip
static inline bool read_state_proof(const uint8_t *bytes, size_t len, StateProofContainer &out) {
    Bincode c(bytes, len);
    if (!read_wrap_proof(c, out.tip_proof)) return false;
    for (int i = 0; i < 17; ++i) if (!read_protocol_state(c, out.states[i])) return false;
    return c.ok && c.pos == len;
}

}  // namespace mw
