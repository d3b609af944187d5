// api_pickles.hip -- the Pickles glue of `verify_block` around the kimchi step: what openmina's
// `ledger::proofs::verification` does with the statement of a `MinaBaseProofStableV2` before `kimchi::verifier::verify`:
//   compute_deferred_values   endo-expanded challenges, the Tick (Fp) sponge over the step proof's evaluations -> xi, r;
//                             ft_eval0 of the STEP proof (permutation + boundary - public - the step linearization's constant
//                             term: a PolishToken program), derive_plonk (perm, zeta^(2^16), zeta^n), combined inner product, b
//   message digests           messages_for_next_wrap_proof (Tock / Fq sponge), messages_for_next_step_proof (Tick / Fp sponge over
//                             the wrap index commitments, the application state = protocol-state hash, the previous accumulators)
//   public input              the statement packed into the 40 scalars the wrap circuit exposes
// Sponges run on the GPU as batched opcode tapes (`sponge_tape_kernel`, one tape shape per call); the scalar work between them is
// a few hundred field operations per proof on the host (host build of fp.cuh).  The result feeds the kimchi step of the
// Proof-of-State job as its public input, which binds every statement field to the wrap proof.
// [UPSTREAM-RECALL]; oracle restatement: oracle/pickles_ref.py.  The STEP index (zk rows, domain shifts, linearization) is data.
#include <mutex>

#include "ctx.h"
#include "sponge.cuh"
#include "kimchi_dev.cuh"
#include "wire_proof.h"

namespace {

constexpr uint32_t SRS_LENGTH_LOG2 = 16, PERM_ALPHA0 = 21, N_STEP_COLS = 43 + 19;

template <int F> fe_t to_mont_bytes(const uint8_t *b, const FieldK &k) { fe_t a; memcpy(a.v, b, 32); return fe_to_mont<F>(a, k.r2); }
template <int F> fe_t from_u128(const mw::Chal128 &c, const FieldK &k) {
    fe_t a = fe_zero(); a.v[0] = (uint32_t)c.lo; a.v[1] = (uint32_t)(c.lo >> 32); a.v[2] = (uint32_t)c.hi; a.v[3] = (uint32_t)(c.hi >> 32); return fe_to_mont<F>(a, k.r2);
}
template <int F> void put_plain(std::vector<uint8_t> &v, const fe_t &mont) { const fe_t p = fe_from_mont<F>(mont); const uint8_t *b = (const uint8_t *)p.v; v.insert(v.end(), b, b + 32); }
template <int F> fe_t bpoly_host(const std::vector<fe_t> &chals, fe_t x, const FieldK &k) {
    fe_t acc = k.one;
    for (size_t i = chals.size(); i-- > 0;) { acc = fe_mul<F>(acc, fe_add<F>(k.one, fe_mul<F>(chals[i], x))); x = fe_sqr<F>(x); }
    return acc;
}
// reduce a 256-bit little-endian integer modulo the field (digests are 4 x 64-bit limbs)
template <int F> fe_t reduce_256(const uint8_t *b, const FieldK &k) {
    fe_t lo = fe_zero(), hi = fe_zero();
    memcpy(lo.v, b, 16); memcpy(hi.v, b + 16, 16);                          // x = lo + 2^128 hi, both < 2^128 < p
    fe_t two128 = fe_zero(); two128.v[4] = 1;
    return fe_add<F>(fe_to_mont<F>(lo, k.r2), fe_mul<F>(fe_to_mont<F>(hi, k.r2), fe_to_mont<F>(two128, k.r2)));
}

struct StepIndexHost {
    bool installed = false; uint32_t zk_rows = 3;
    std::vector<uint32_t> domains; std::vector<std::array<fe_t, 7>> shifts;      // per domain, Montgomery (Fp)
    std::vector<mb::KimchiToken> toks; std::vector<fe_t> lits; fe_t mds[9]; fe_t endo_coeff;
    uint32_t max_col = 0;                                                        // highest evaluation column the program names
    bool feature_aware = false;                                                  // the program looks at the proof's features: SkipIf / SkipIfNot, the joint combiner or an optional column
};
StepIndexHost &step_of(mina_ctx *c) {           // host half of the step index: owned by the context, freed with it (mina_ctx_destroy)
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!c->step_host) { c->step_host = new StepIndexHost(); c->step_host_free = [](void *p) { delete (StepIndexHost *)p; }; }
    return *(StepIndexHost *)c->step_host;
}

struct Derived { fe_t cip, b, zeta_srs, zeta_dom, perm, xi, r; mw::Chal128 xi_chal; };

}  // namespace

int mb_step_index_feature_aware(mina_ctx *c) { return (c->step_host && step_of(c).installed && step_of(c).feature_aware) ? 1 : 0; }
int mb_step_index_installed(mina_ctx *c) { return (c->step_host && c->have_pickles_dev && step_of(c).installed) ? 1 : 0; }
static int upload_step_index(mina_ctx *c, const StepIndexHost &st);

extern "C" int mina_step_index_install(mina_ctx *c, const mina_step_index *si) {
    if (!c || !si || !si->domain_log2 || !si->shifts || (si->constant_term_len && !si->constant_term)) return fail(MINA_ERR_ARG, "null argument");
    if (si->n_domains == 0 || si->n_domains > 8 || si->zk_rows < 1 || si->zk_rows > 8) return fail(MINA_ERR_ARG, "bad n_domains / zk_rows");
    if (!c->have_pparams[FIELD_FP]) return fail(MINA_ERR_STATE, "Poseidon constants not installed");
    StepIndexHost st;
    std::vector<std::array<uint8_t, 32>> lits;
    if (!mb::decode_tokens(si->constant_term, si->constant_term_len, FIELD_FP, N_STEP_COLS, st.toks, lits)) return fail(MINA_ERR_FORMAT, "malformed PolishToken program");
    const FieldK &k = c->fk[FIELD_FP];
    for (uint32_t d = 0; d < si->n_domains; ++d) {
        if (si->domain_log2[d] < 1 || si->domain_log2[d] > 24) return fail(MINA_ERR_ARG, "bad domain");
        std::array<fe_t, 7> s;
        for (int i = 0; i < 7; ++i) { const uint8_t *b = si->shifts + ((size_t)d * 7 + i) * 32; if (!mw::fp_canonical(b)) return fail(MINA_ERR_FORMAT, "shift is not canonical"); s[i] = to_mont_bytes<FIELD_FP>(b, k); }
        st.domains.push_back(si->domain_log2[d]); st.shifts.push_back(s);
    }
    for (auto &l : lits) st.lits.push_back(to_mont_bytes<FIELD_FP>(l.data(), k));
    HIPC(hipSetDevice(c->device));
    { PoseidonParams pp; HIPC(hipMemcpy(&pp, c->pparams[FIELD_FP].p, sizeof pp, hipMemcpyDeviceToHost)); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) st.mds[3 * i + j] = pp.mds[i][j]; }
    st.endo_coeff = fe_sqr<FIELD_FP>(k.endo);                      // endo_q of Pallas = cube root of unity in Fp: (w^2)^2 = w
    st.zk_rows = si->zk_rows; st.installed = true;
    for (auto &t : st.toks) if (t.op == MINA_TOK_CELL && t.a > st.max_col) st.max_col = t.a;
    for (auto &t : st.toks) if (t.op == MINA_TOK_SKIP_IF || t.op == MINA_TOK_SKIP_IF_NOT || t.op == MINA_TOK_JOINT_COMBINER || (t.op == MINA_TOK_CELL && t.a >= mb::KC_COLS)) st.feature_aware = true;
    int rc = upload_step_index(c, st);
    if (rc) return rc;
    step_of(c) = st;
    return MINA_OK;
}

// run one tape over `n` sponges (all with the same tape) on the GPU; inputs / outputs are canonical 32-byte elements
static int run_tape(mina_ctx *c, int curve, const std::vector<uint8_t> &tape, size_t n, const std::vector<uint8_t> &inputs, std::vector<uint8_t> &outputs, size_t n_out) {
    outputs.assign(n * n_out * 32, 0);
    return mina_fq_sponge_run(c, curve, n, tape.data(), tape.size(), nullptr, nullptr, inputs.data(), outputs.data(), nullptr, nullptr);
}

// statement + prev_evals + app state -> the 40 public inputs (plain Fq bytes) and the derived values.  Proofs whose evaluation
// shape (number of optional evaluations, chunks) differs are handled one tape shape at a time.
int mb_pickles_public_inputs(mina_ctx *c, const mw::WrapProof *const *proofs, const uint8_t *const *app_states, size_t n, uint8_t *pub_out /* n*40*32 */,
                             uint8_t *derived_out /* n*7*32 or null */, uint8_t *ok_out /* n */) {
    StepIndexHost &st = step_of(c);
    if (!st.installed) return fail(MINA_ERR_STATE, "no step index installed");
    if (!c->have_kimchi) return fail(MINA_ERR_STATE, "no wrap verifier index installed");
    const FieldK &kp = c->fk[FIELD_FP], &kq = c->fk[FIELD_FQ];
    int rc;
    for (size_t i = 0; i < n; ++i) ok_out[i] = 1;
    // ---- per proof: challenges, combined evaluations, the element lists of the four sponges
    struct Work { std::vector<std::array<fe_t, 2>> seq; std::vector<fe_t> bp; std::vector<std::vector<fe_t>> old; fe_t alpha, zeta, beta, gamma, zetaw, omega; uint32_t k; int dom; };
    std::vector<Work> W(n);
    std::vector<uint8_t> in_chd, in_mw, in_ms, out_chd, out_mw, out_ms;
    size_t n_old = SIZE_MAX, n_seq = SIZE_MAX;
    for (size_t i = 0; i < n; ++i) {
        const mw::WrapProof &w = *proofs[i]; Work &x = W[i];
        x.k = w.domain_log2; x.dom = -1;
        for (size_t d = 0; d < st.domains.size(); ++d) if (st.domains[d] == x.k) x.dom = (int)d;
        if (x.dom < 0 || w.step_old_bulletproof_challenges.size() != w.step_challenge_polynomial_commitments.size()) { ok_out[i] = 0; x.dom = 0; x.k = st.domains[0]; }
        x.alpha = challenge_to_field<FIELD_FP>(w.alpha.lo, w.alpha.hi, kp); x.zeta = challenge_to_field<FIELD_FP>(w.zeta.lo, w.zeta.hi, kp);
        x.beta = from_u128<FIELD_FP>(w.beta, kp); x.gamma = from_u128<FIELD_FP>(w.gamma, kp);
        fe_t om = kp.root; for (uint32_t j = 0; j < 32 - x.k; ++j) om = fe_sqr<FIELD_FP>(om);
        x.omega = om; x.zetaw = fe_mul<FIELD_FP>(x.zeta, om);
        fe_t zn = x.zeta, zwn = x.zetaw; for (uint32_t j = 0; j < SRS_LENGTH_LOG2; ++j) { zn = fe_sqr<FIELD_FP>(zn); zwn = fe_sqr<FIELD_FP>(zwn); }
        // wire order: w 15, coefficients 15, z, s 6, selectors 6, then the present optional ones -> kimchi column order
        const size_t order[43] = {30, 37, 38, 39, 40, 41, 42, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 31, 32, 33, 34, 35, 36};
        auto comb = [&](const mw::SmallVec<mw::B32, 16> &chunks, const fe_t &ptn, bool &good) { fe_t acc = fe_zero(); for (size_t j = chunks.size(); j-- > 0;) { good = good && mw::fp_canonical(chunks[j].b); acc = fe_add<FIELD_FP>(fe_mul<FIELD_FP>(acc, ptn), to_mont_bytes<FIELD_FP>(chunks[j].b, kp)); } return acc; };
        bool good = w.prev_evals.size() >= 43;
        for (size_t j = 0; j < w.prev_evals.size() && good; ++j) {
            const mw::EvalPair &e = w.prev_evals[j < 43 ? order[j] : j];
            x.seq.push_back({comb(e.zeta, zn, good), comb(e.zeta_omega, zwn, good)});
        }
        if (!good) { ok_out[i] = 0; x.seq.assign(43, {fe_zero(), fe_zero()}); }
        for (int j = 0; j < 16; ++j) x.bp.push_back(challenge_to_field<FIELD_FP>(w.bulletproof_challenges[j].lo, w.bulletproof_challenges[j].hi, kp));
        for (auto &row : w.step_old_bulletproof_challenges) { std::vector<fe_t> r; for (int j = 0; j < 16; ++j) r.push_back(challenge_to_field<FIELD_FP>(row[j].lo, row[j].hi, kp)); x.old.push_back(r); }
        if (n_old == SIZE_MAX) { n_old = x.old.size(); n_seq = x.seq.size(); }
        if (x.old.size() != n_old || x.seq.size() != n_seq) return fail(MINA_ERR_ARG, "proofs of one call must share the evaluation / recursion shape");
        for (auto &r : x.old) for (auto &e : r) put_plain<FIELD_FP>(in_chd, e);
        for (int a = 0; a < 2; ++a) for (int j = 0; j < 15; ++j) put_plain<FIELD_FQ>(in_mw, challenge_to_field<FIELD_FQ>(w.old_bulletproof_challenges[a][j].lo, w.old_bulletproof_challenges[a][j].hi, kq));
        in_mw.insert(in_mw.end(), w.challenge_polynomial_commitment.x.b, w.challenge_polynomial_commitment.x.b + 32);
        in_mw.insert(in_mw.end(), w.challenge_polynomial_commitment.y.b, w.challenge_polynomial_commitment.y.b + 32);
        in_ms.insert(in_ms.end(), c->kimchi_comms_host, c->kimchi_comms_host + 28 * 64);
        { fe_t a = to_mont_bytes<FIELD_FP>(app_states[i], kp); if (!mw::fp_canonical(app_states[i])) ok_out[i] = 0; put_plain<FIELD_FP>(in_ms, a); }
        for (size_t a = 0; a < x.old.size(); ++a) {
            const mw::Pt &cm = w.step_challenge_polynomial_commitments[a];
            in_ms.insert(in_ms.end(), cm.x.b, cm.x.b + 32); in_ms.insert(in_ms.end(), cm.y.b, cm.y.b + 32);
            for (auto &e : x.old[a]) put_plain<FIELD_FP>(in_ms, e);
        }
    }
    // ---- sponges, batch 1: digest of the old challenges (Fp), messages_for_next_wrap_proof (Fq), messages_for_next_step_proof (Fp)
    {
        std::vector<uint8_t> t1(n_old * 16, MINA_TAPE_ABSORB_FQ); t1.push_back(MINA_TAPE_CHALLENGE_FQ);
        if ((rc = run_tape(c, CURVE_PALLAS, t1, n, in_chd, out_chd, 1))) return rc;
        std::vector<uint8_t> t2(32, MINA_TAPE_ABSORB_FQ); t2.push_back(MINA_TAPE_CHALLENGE_FQ);
        if ((rc = run_tape(c, CURVE_VESTA, t2, n, in_mw, out_mw, 1))) return rc;
        std::vector<uint8_t> t3(56 + 1 + n_old * 18, MINA_TAPE_ABSORB_FQ); t3.push_back(MINA_TAPE_CHALLENGE_FQ);
        if ((rc = run_tape(c, CURVE_PALLAS, t3, n, in_ms, out_ms, 1))) return rc;
    }
    // ---- sponge batch 2: the Tick sponge over the evaluations -> xi, r
    std::vector<uint8_t> in_main, out_main;
    for (size_t i = 0; i < n; ++i) {
        const mw::WrapProof &w = *proofs[i];
        uint8_t dg[32]; memcpy(dg, w.sponge_digest_before_evaluations, 32);
        put_plain<FIELD_FP>(in_main, reduce_256<FIELD_FP>(dg, kp));
        in_main.insert(in_main.end(), out_chd.begin() + i * 32, out_chd.begin() + (i + 1) * 32);
        in_main.insert(in_main.end(), w.prev_ft_eval1.b, w.prev_ft_eval1.b + 32);
        const mw::B32 &p0 = w.prev_public_input.zeta[0], &p1 = w.prev_public_input.zeta_omega[0];
        if (!mw::fp_canonical(p0.b) || !mw::fp_canonical(p1.b) || !mw::fp_canonical(w.prev_ft_eval1.b)) ok_out[i] = 0;
        in_main.insert(in_main.end(), p0.b, p0.b + 32); in_main.insert(in_main.end(), p1.b, p1.b + 32);
        for (auto &pr : W[i].seq) { put_plain<FIELD_FP>(in_main, pr[0]); put_plain<FIELD_FP>(in_main, pr[1]); }
    }
    {
        std::vector<uint8_t> t(5 + 2 * n_seq, MINA_TAPE_ABSORB_FQ); t.push_back(MINA_TAPE_CHALLENGE); t.push_back(MINA_TAPE_CHALLENGE);
        if ((rc = run_tape(c, CURVE_PALLAS, t, n, in_main, out_main, 2))) return rc;
    }
    // ---- scalar work + packing
    for (size_t i = 0; i < n; ++i) {
        const mw::WrapProof &w = *proofs[i]; Work &x = W[i];
        mw::Chal128 xc, rcl; memcpy(&xc.lo, &out_main[(i * 2) * 32], 8); memcpy(&xc.hi, &out_main[(i * 2) * 32 + 8], 8); memcpy(&rcl.lo, &out_main[(i * 2 + 1) * 32], 8); memcpy(&rcl.hi, &out_main[(i * 2 + 1) * 32 + 8], 8);
        const fe_t xi = challenge_to_field<FIELD_FP>(xc.lo, xc.hi, kp), r = challenge_to_field<FIELD_FP>(rcl.lo, rcl.hi, kp);
        const uint64_t nn = (uint64_t)1 << x.k;
        const std::array<fe_t, 7> &sh = st.shifts[x.dom];
        fe_t zkp = kp.one;
        for (uint32_t j = 0; j < st.zk_rows; ++j) zkp = fe_mul<FIELD_FP>(zkp, fe_sub<FIELD_FP>(x.zeta, mb::host_pow_u64<FIELD_FP>(x.omega, nn - st.zk_rows + j, kp.one)));
        const fe_t a0 = mb::host_pow_u64<FIELD_FP>(x.alpha, PERM_ALPHA0, kp.one), a1 = fe_mul<FIELD_FP>(a0, x.alpha), a2 = fe_mul<FIELD_FP>(a1, x.alpha);
        auto Wc = [&](int j) { return x.seq[7 + j][0]; }; auto Sc = [&](int j) { return x.seq[37 + j][0]; };
        const fe_t z0 = x.seq[0][0], z1 = x.seq[0][1];
        fe_t zeta_dom = x.zeta; for (uint32_t j = 0; j < x.k; ++j) zeta_dom = fe_sqr<FIELD_FP>(zeta_dom);
        const fe_t zeta1m1 = fe_sub<FIELD_FP>(zeta_dom, kp.one);
        fe_t prod6 = kp.one;
        for (int j = 0; j < 6; ++j) prod6 = fe_mul<FIELD_FP>(prod6, fe_add<FIELD_FP>(fe_add<FIELD_FP>(fe_mul<FIELD_FP>(x.beta, Sc(j)), Wc(j)), x.gamma));
        const fe_t common = fe_mul<FIELD_FP>(fe_mul<FIELD_FP>(a0, zkp), prod6);
        fe_t ft = fe_mul<FIELD_FP>(fe_mul<FIELD_FP>(fe_add<FIELD_FP>(Wc(6), x.gamma), z1), common);
        const fe_t p0 = to_mont_bytes<FIELD_FP>(w.prev_public_input.zeta[0].b, kp), p1 = to_mont_bytes<FIELD_FP>(w.prev_public_input.zeta_omega[0].b, kp);
        ft = fe_sub<FIELD_FP>(ft, p0);
        fe_t t2 = fe_mul<FIELD_FP>(fe_mul<FIELD_FP>(a0, zkp), z0);
        const fe_t bz = fe_mul<FIELD_FP>(x.beta, x.zeta);
        for (int j = 0; j < 7; ++j) t2 = fe_mul<FIELD_FP>(t2, fe_add<FIELD_FP>(fe_add<FIELD_FP>(x.gamma, fe_mul<FIELD_FP>(bz, sh[j])), Wc(j)));
        ft = fe_sub<FIELD_FP>(ft, t2);
        const fe_t wz = mb::host_pow_u64<FIELD_FP>(x.omega, nn - st.zk_rows, kp.one), dw = fe_sub<FIELD_FP>(x.zeta, wz), d1 = fe_sub<FIELD_FP>(x.zeta, kp.one);
        const fe_t num = fe_mul<FIELD_FP>(fe_add<FIELD_FP>(fe_mul<FIELD_FP>(fe_mul<FIELD_FP>(zeta1m1, a1), dw), fe_mul<FIELD_FP>(fe_mul<FIELD_FP>(zeta1m1, a2), d1)), fe_sub<FIELD_FP>(kp.one, z0));
        ft = fe_add<FIELD_FP>(ft, fe_mul<FIELD_FP>(num, fe_inv<FIELD_FP>(fe_mul<FIELD_FP>(dw, d1), kp)));
        if (!st.toks.empty()) {
            mb::PolishEnv<FIELD_FP> env{x.alpha, x.beta, x.gamma, st.endo_coeff, zkp, x.zeta, zeta_dom, x.omega, st.mds, x.k, st.zk_rows, &x.seq};
            { uint8_t ff[8]; for (int j = 0; j < 8; ++j) ff[j] = w.feature_flags[j] ? 1 : 0; env.features = mb::feature_mask_of_flags(ff); }
            if (w.has_joint_combiner) env.joint_combiner = challenge_to_field<FIELD_FP>(w.joint_combiner.lo, w.joint_combiner.hi, kp);
            env.slots = true; env.present = 0;
            for (size_t j = 0; j < w.prev_evals_present.size() && j < 19; ++j) if (w.prev_evals_present[j]) env.present |= 1u << j;
            fe_t ct;
            if (!mb::polish_eval_host<FIELD_FP>(st.toks, st.lits, env, kp, ct)) { ok_out[i] = 0; ct = fe_zero(); }
            ft = fe_sub<FIELD_FP>(ft, ct);
        }
        Derived d;
        d.perm = fe_neg<FIELD_FP>(fe_mul<FIELD_FP>(fe_mul<FIELD_FP>(z1, x.beta), common));
        d.zeta_dom = zeta_dom; d.zeta_srs = x.zeta; for (uint32_t j = 0; j < SRS_LENGTH_LOG2; ++j) d.zeta_srs = fe_sqr<FIELD_FP>(d.zeta_srs);
        auto combine = [&](int side, const fe_t &ftv, const fe_t &pt) {
            std::vector<fe_t> v;
            for (auto &row : x.old) v.push_back(bpoly_host<FIELD_FP>(row, pt, kp));
            v.push_back(side ? p1 : p0); v.push_back(ftv);
            for (auto &pr : x.seq) v.push_back(pr[side]);
            fe_t acc = fe_zero();
            for (size_t j = v.size(); j-- > 0;) acc = fe_add<FIELD_FP>(fe_mul<FIELD_FP>(acc, xi), v[j]);
            return acc;
        };
        d.cip = fe_add<FIELD_FP>(combine(0, ft, x.zeta), fe_mul<FIELD_FP>(r, combine(1, to_mont_bytes<FIELD_FP>(w.prev_ft_eval1.b, kp), x.zetaw)));
        d.b = fe_add<FIELD_FP>(bpoly_host<FIELD_FP>(x.bp, x.zeta, kp), fe_mul<FIELD_FP>(r, bpoly_host<FIELD_FP>(x.bp, x.zetaw, kp)));
        d.xi = xi; d.r = r;
        if (derived_out) {
            std::vector<uint8_t> o; put_plain<FIELD_FP>(o, d.cip); put_plain<FIELD_FP>(o, d.b); put_plain<FIELD_FP>(o, d.zeta_srs); put_plain<FIELD_FP>(o, d.zeta_dom);
            put_plain<FIELD_FP>(o, d.perm); put_plain<FIELD_FP>(o, d.xi); put_plain<FIELD_FP>(o, d.r);
            memcpy(derived_out + i * 7 * 32, o.data(), 7 * 32);
        }
        // ---- PreparedStatement::to_public_input(40): plain integers, reduced into Fq
        uint8_t *po = pub_out + i * 40 * 32; size_t slot = 0;
        auto put_fp_as_fq = [&](const fe_t &mont_fp) { const fe_t pl = fe_from_mont<FIELD_FP>(mont_fp); memcpy(po + 32 * slot++, pl.v, 32); };       // p < q: the integer is canonical in Fq
        auto put_u128 = [&](const mw::Chal128 &cc) { uint8_t e[32] = {0}; memcpy(e, &cc.lo, 8); memcpy(e + 8, &cc.hi, 8); memcpy(po + 32 * slot++, e, 32); };
        auto put_bytes_mod_q = [&](const uint8_t *b) { const fe_t v = fe_from_mont<FIELD_FQ>(reduce_256<FIELD_FQ>(b, kq)); memcpy(po + 32 * slot++, v.v, 32); };
        auto put_small = [&](uint64_t v) { uint8_t e[32] = {0}; memcpy(e, &v, 8); memcpy(po + 32 * slot++, e, 32); };
        const fe_t shift = fe_add<FIELD_FP>(kp.two255, kp.one);
        auto shifted = [&](const fe_t &v) { return fe_mul<FIELD_FP>(fe_sub<FIELD_FP>(v, shift), kp.inv2); };                    // Shifted_value.Type1.of_field
        put_fp_as_fq(shifted(d.cip)); put_fp_as_fq(shifted(d.b)); put_fp_as_fq(shifted(d.zeta_srs)); put_fp_as_fq(shifted(d.zeta_dom)); put_fp_as_fq(shifted(d.perm));
        put_u128(w.beta); put_u128(w.gamma);
        put_u128(w.alpha); put_u128(w.zeta); put_u128(xc);
        { uint8_t dg[32]; memcpy(dg, w.sponge_digest_before_evaluations, 32); put_bytes_mod_q(dg); }
        memcpy(po + 32 * slot++, &out_mw[i * 32], 32);                                        // Fq digest
        memcpy(po + 32 * slot++, &out_ms[i * 32], 32);                                        // Fp digest: its integer value, < p < q
        for (int j = 0; j < 16; ++j) put_u128(w.bulletproof_challenges[j]);
        { const uint64_t mask = w.proofs_verified == 0 ? 0 : (w.proofs_verified == 1 ? 2 : 3); put_small(4ull * w.domain_log2 + mask); }
        for (int j = 0; j < 8; ++j) put_small(w.feature_flags[j] ? 1 : 0);
        put_small(w.has_joint_combiner ? 1 : 0); put_u128(w.has_joint_combiner ? w.joint_combiner : mw::Chal128{0, 0});
    }
    return MINA_OK;
}

// diagnostic / test hook: one serialized wrap proof -> its 40 public inputs + the derived values
extern "C" int mina_pickles_public_input(mina_ctx *c, const uint8_t *proof, size_t len, int encoding, const uint8_t *app_state, uint8_t *public_input_out, uint8_t *derived_out) {
    if (!c || !proof || !app_state || !public_input_out) return fail(MINA_ERR_ARG, "null argument");
    mw::WrapProof w; bool ok;
    if (encoding == MINA_ENC_BINPROT) { mw::Binprot r(proof, len); ok = mw::read_wrap_proof(r, w) && r.pos == len; }
    else if (encoding == MINA_ENC_BINCODE) { mw::Bincode r(proof, len); ok = mw::read_wrap_proof(r, w) && r.pos == len; }
    else return fail(MINA_ERR_ARG, "bad encoding");
    if (!ok) return fail(MINA_ERR_FORMAT, "malformed wrap proof");
    HIPC(hipSetDevice(c->device));
    const mw::WrapProof *pw = &w; uint8_t good = 0;
    int rc = mb_pickles_public_inputs(c, &pw, &app_state, 1, public_input_out, derived_out, &good);
    if (rc) return rc;
    return good ? MINA_OK : fail(MINA_ERR_FORMAT, "statement does not fit the installed step index (domain, evaluation shape or a non-canonical element)");
}

// ------------------------------------------------------------------------------------------------ the same on the GPU
// Batch form for the Proof-of-State job: statements (structure-of-arrays in HBM) -> 40 public inputs per proof, no host work.
//   expand   one thread per challenge: endo-expansion (Fp for the step side, Fq for the wrap side), beta / gamma into the field
//   digest   lane-cooperative sponges, three roles side by side (each in its own waves): digest of the step-side old challenges (Tick), messages_for_next_
//            wrap_proof (Tock), messages_for_next_step_proof (Tick, resumed after the 28 wrap index commitments -- absorbed once at
//            set-up, the state cached in the index)
//   tick     lane-cooperative: the Tick sponge over the step proof's evaluations -> xi, r
//   scalar   one lane per proof: ft_eval0 of the step proof (kimchi_dev.cuh), derive_plonk, combined inner product, b, packing
// Values pass between the stages through `xe` (PX_* elements per proof, Montgomery in their field).
namespace mb {

static constexpr uint32_t PK_MAX_DOMAINS = 8, PK_PUB = 40;
struct PicklesIndexDev {
    uint32_t zk_rows, n_tokens, n_domains, max_col;
    uint32_t domain_log2[PK_MAX_DOMAINS];
    fe_t shifts[PK_MAX_DOMAINS][7], omega[PK_MAX_DOMAINS], omega_zk[PK_MAX_DOMAINS], zk_roots[PK_MAX_DOMAINS][KC_MAX_ZK];   // Fp, Montgomery
    fe_t mds[9], endo_coeff;
    fe_t ms_state[3]; uint32_t ms_squeezed, ms_count;             // Tick sponge after the wrap index commitments (Montgomery)
};
enum { PX_ALPHA = 0, PX_ZETA, PX_BETA, PX_GAMMA, PX_CHD, PX_MW, PX_MS, PX_XIC, PX_XI, PX_R, PX_BP = 10, PX_OLD = 26 };
__host__ __device__ static inline uint32_t px_wold(uint32_t n_old) { return PX_OLD + 16 * n_old; }
__host__ __device__ static inline uint32_t px_kdig(uint32_t n_old) { return PX_OLD + 16 * n_old + 30; }      // kimchi's digest of the wrap-side old challenges (Fq)
__host__ __device__ static inline uint32_t px_stride(uint32_t n_old) { return PX_OLD + 16 * n_old + 31; }
struct PicklesIn {
    uint32_t n_old, n_evals;
    const uint32_t *plonk, *bp, *old_chals, *step_comms, *wold, *wrap_sg, *digest, *evals, *pub_in, *ft_eval1, *app_state; const uint8_t *misc;
};

__device__ __forceinline__ void chal_words(const uint32_t *p, uint64_t &lo, uint64_t &hi) { lo = (uint64_t)p[0] | ((uint64_t)p[1] << 32); hi = (uint64_t)p[2] | ((uint64_t)p[3] << 32); }
__device__ __forceinline__ fe_t u128_fe(const uint32_t *p) { fe_t a = fe_zero(); a.v[0] = p[0]; a.v[1] = p[1]; a.v[2] = p[2]; a.v[3] = p[3]; return a; }

__global__ void __launch_bounds__(256)
pickles_expand_kernel(uint32_t batch, FieldK kp, FieldK kq, PicklesIn in, fe_t *__restrict__ xe) { mb_wave_prio();
    const uint32_t per = 4 + 16 + 16 * in.n_old + 30;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)batch * per) return;
    const uint32_t b = (uint32_t)(gid / per), it = (uint32_t)(gid % per);
    fe_t *x = xe + (size_t)b * px_stride(in.n_old);
    uint64_t lo, hi;
    if (it < 4) {                                           // alpha, beta, gamma, zeta in wire order
        const uint32_t *p = in.plonk + ((size_t)b * 4 + it) * 4;
        if (it == 0 || it == 3) { chal_words(p, lo, hi); x[it == 0 ? PX_ALPHA : PX_ZETA] = challenge_to_field<FIELD_FP>(lo, hi, kp); }
        else x[it == 1 ? PX_BETA : PX_GAMMA] = fe_to_mont<FIELD_FP>(u128_fe(p), kp.r2);
    } else if (it < 20) {
        chal_words(in.bp + ((size_t)b * 16 + (it - 4)) * 4, lo, hi); x[PX_BP + (it - 4)] = challenge_to_field<FIELD_FP>(lo, hi, kp);
    } else if (it < 20 + 16 * in.n_old) {
        const uint32_t j = it - 20;
        chal_words(in.old_chals + ((size_t)b * 16 * in.n_old + j) * 4, lo, hi); x[PX_OLD + j] = challenge_to_field<FIELD_FP>(lo, hi, kp);
    } else {
        const uint32_t j = it - 20 - 16 * in.n_old;
        chal_words(in.wold + ((size_t)b * 30 + j) * 4, lo, hi); x[px_wold(in.n_old) + j] = challenge_to_field<FIELD_FQ>(lo, hi, kq);
    }
}

template <int LANES>
__global__ void __launch_bounds__(64)
pickles_digest_kernel(uint32_t batch, FieldK kp, FieldK kq, const PoseidonParams *__restrict__ pp_p, const PoseidonParams *__restrict__ pp_q,
                      const PicklesIndexDev *__restrict__ ix, PicklesIn in, fe_t *__restrict__ xe, uint32_t *__restrict__ ok_out, uint32_t nblk) { mb_wave_prio();
    bool writer; const uint32_t role = blockIdx.x / nblk, b = coop_role_item<LANES>(blockIdx.x % nblk, writer);       // one role per wave
    if (b >= batch) return;
    fe_t *x = xe + (size_t)b * px_stride(in.n_old);
    bool ok = true;
    if (role == 0) {                                        // digest of the step-side old challenges
        DevSponge<FIELD_FP, LANES> sp; sponge_init(sp, pp_p);
#pragma unroll 1
        for (uint32_t i = 0; i < 16 * in.n_old; ++i) sp.absorb(x[PX_OLD + i]);
        const fe_t d = sp.squeeze();
        if (writer) x[PX_CHD] = d;
    } else if (role == 1) {                                 // messages_for_next_wrap_proof (Tock)
        DevSponge<FIELD_FQ, LANES> sp; sponge_init(sp, pp_q);
#pragma unroll 1
        for (uint32_t i = 0; i < 30; ++i) sp.absorb(x[px_wold(in.n_old) + i]);
        // The wrap proof's kimchi verification digests exactly these 30 challenges with the same sponge (`challenges_digest`: absorb all,
        // squeeze): its squeeze is the permutation the next absorb here would trigger.  Run it once, hand state[0] to the kimchi stage
        // (15 permutations per proof it no longer repeats) and go on as if the absorb had triggered it.
        poseidon_permute_coop<FIELD_FQ, LANES>(sp.s, sp.pp);
        { const fe_t kd = sp.get(0); if (writer) x[px_kdig(in.n_old)] = kd; }
        sp.count = 0;
        sp.absorb(ld_checked<FIELD_FQ>(in.wrap_sg + (size_t)b * 16, kq, ok)); sp.absorb(ld_checked<FIELD_FQ>(in.wrap_sg + (size_t)b * 16 + 8, kq, ok));
        const fe_t d = sp.squeeze();
        if (writer) x[PX_MW] = d;
    } else {                                                // messages_for_next_step_proof (Tick), after the index commitments
        DevSponge<FIELD_FP, LANES> sp; sp.pp = pp_p; sp.s = ix->ms_state[coop_elem<LANES>()]; sp.squeezed = (int)ix->ms_squeezed; sp.count = (int)ix->ms_count;
        sp.absorb(ld_checked<FIELD_FP>(in.app_state + (size_t)b * 8, kp, ok));
#pragma unroll 1
        for (uint32_t a = 0; a < in.n_old; ++a) {
            const uint32_t *cm = in.step_comms + ((size_t)b * in.n_old + a) * 16;
            sp.absorb(ld_checked<FIELD_FP>(cm, kp, ok)); sp.absorb(ld_checked<FIELD_FP>(cm + 8, kp, ok));
#pragma unroll 1
            for (uint32_t i = 0; i < 16; ++i) sp.absorb(x[PX_OLD + 16 * a + i]);
        }
        const fe_t d = sp.squeeze();
        if (writer) x[PX_MS] = d;
    }
    if (!ok) ok_out[b] = 0u;
}

template <int LANES>
__global__ void __launch_bounds__(64)
pickles_tick_kernel(uint32_t batch, FieldK kp, const PoseidonParams *__restrict__ pp_p, PicklesIn in, fe_t *__restrict__ xe, uint32_t *__restrict__ ok_out) { mb_wave_prio();
    constexpr int F = FIELD_FP;
    bool writer; const uint32_t b = coop_sponge_index<LANES>(writer);
    if (b >= batch) return;
    fe_t *x = xe + (size_t)b * px_stride(in.n_old);
    bool ok = true;
    DevSponge<F, LANES> sp; sponge_init(sp, pp_p);
    sp.absorb(fe_to_mont<F>(load_fe<F>(in.digest + (size_t)b * 8), kp.r2));        // any 256-bit value: the Montgomery product reduces it
    sp.absorb(x[PX_CHD]);
    sp.absorb(ld_checked<F>(in.ft_eval1 + (size_t)b * 8, kp, ok));
    sp.absorb(ld_checked<F>(in.pub_in + (size_t)b * 16, kp, ok)); sp.absorb(ld_checked<F>(in.pub_in + (size_t)b * 16 + 8, kp, ok));
    const uint32_t *ev = in.evals + (size_t)b * in.n_evals * 16;
#pragma unroll 1
    for (uint32_t c = 0; c < 2 * in.n_evals; ++c) sp.absorb(ld_checked<F>(ev + (size_t)c * 8, kp, ok));
    const fe_t xi_sq = fe_from_mont<F>(sp.squeeze()), r_sq = fe_from_mont<F>(sp.squeeze());
    if (writer) {
        fe_t xc = fe_zero(); xc.v[0] = xi_sq.v[0]; xc.v[1] = xi_sq.v[1]; xc.v[2] = xi_sq.v[2]; xc.v[3] = xi_sq.v[3];
        x[PX_XIC] = xc; x[PX_XI] = chal_endo<F>(xi_sq, kp); x[PX_R] = chal_endo<F>(r_sq, kp);
    }
    if (!ok) ok_out[b] = 0u;
}

__global__ void __launch_bounds__(64)
pickles_scalar_kernel(uint32_t batch, FieldK kp, FieldK kq, const PicklesIndexDev *__restrict__ ix, const KimchiToken *__restrict__ toks, const fe_t *__restrict__ lits,
                      PicklesIn in, const fe_t *__restrict__ xe, uint32_t *__restrict__ pub_out, uint32_t *__restrict__ ok_out) { mb_wave_prio();
    constexpr int F = FIELD_FP;
    __shared__ uint32_t lds[KC_SLOTS * 8 * 64];
    const uint32_t b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    LdsStack st{lds + threadIdx.x};
    const fe_t *x = xe + (size_t)b * px_stride(in.n_old);
    const uint8_t *misc = in.misc + (size_t)b * 32;
    bool ok = true;
    uint32_t dom = 0; { bool found = false; for (uint32_t d = 0; d < ix->n_domains; ++d) if (ix->domain_log2[d] == misc[0]) { dom = d; found = true; } ok = ok && found; }
    const uint32_t k = ix->domain_log2[dom];
    const uint32_t *ev = in.evals + (size_t)b * in.n_evals * 16;
    // optional evaluations: the program names SLOTS (43 + slot, wire order), the proof carries the ones its presence mask marks
    const uint32_t present = (uint32_t)misc[11] | ((uint32_t)misc[12] << 8) | ((uint32_t)misc[13] << 16);
    ok = ok && (present >> 19) == 0 && (uint32_t)__popc(present) == in.n_evals - KC_COLS;
    auto EV = [&](uint32_t col, uint32_t row) {
        if (col >= KC_COLS) { const uint32_t slot = col - KC_COLS; if (slot >= 19 || !((present >> slot) & 1u)) { ok = false; return fe_zero(); } col = KC_COLS + (uint32_t)__popc(present & ((1u << slot) - 1)); }
        return fe_to_mont<F>(load_fe<F>(ev + ((size_t)col * 2 + row) * 8), kp.r2);
    };
    auto EVI = [&](uint32_t idx, uint32_t row) { return fe_to_mont<F>(load_fe<F>(ev + ((size_t)idx * 2 + row) * 8), kp.r2); };     // by position (the combined inner product runs over all of them)
    const fe_t zeta = x[PX_ZETA], zetaw = fe_mul<F>(zeta, ix->omega[dom]);
    const fe_t zeta_dom = fe_pow2k<F>(zeta, k), zeta_srs = fe_pow2k<F>(zeta, 16);
    const fe_t p0 = fe_to_mont<F>(load_fe<F>(in.pub_in + (size_t)b * 16), kp.r2), p1 = fe_to_mont<F>(load_fe<F>(in.pub_in + (size_t)b * 16 + 8), kp.r2);
    fe_t perm;
    FtEnv env{x[PX_ALPHA], x[PX_BETA], x[PX_GAMMA], zeta, zeta_dom, ix->omega[dom], ix->omega_zk[dom], ix->endo_coeff, ix->zk_roots[dom], ix->shifts[dom], ix->mds, lits, toks,
              k, ix->zk_rows, 21u, ix->n_tokens};
    env.features = feature_mask_of_flags(misc + 2);
    if (misc[10]) { const uint32_t *jc = (const uint32_t *)(misc + 16); env.joint = challenge_to_field<F>((uint64_t)jc[0] | ((uint64_t)jc[1] << 32), (uint64_t)jc[2] | ((uint64_t)jc[3] << 32), kp); }
    const fe_t ft = ft_eval0_dev<F>(env, kp, p0, EV, st, ok, perm);
    const fe_t xi = x[PX_XI], r = x[PX_R];
    // combined inner product: Horner in xi over [b_poly(old_a, pt)..., public, ft, the evaluations], both points, second scaled by r
    fe_t cip;
    {
        fe_t acc0 = fe_zero(), acc1 = fe_zero();
#pragma unroll 1
        for (int c = (int)in.n_evals - 1; c >= 0; --c) { acc0 = fe_add<F>(fe_mul<F>(acc0, xi), EVI((uint32_t)c, 0)); acc1 = fe_add<F>(fe_mul<F>(acc1, xi), EVI((uint32_t)c, 1)); }
        acc0 = fe_add<F>(fe_mul<F>(acc0, xi), ft); acc1 = fe_add<F>(fe_mul<F>(acc1, xi), fe_to_mont<F>(load_fe<F>(in.ft_eval1 + (size_t)b * 8), kp.r2));
        acc0 = fe_add<F>(fe_mul<F>(acc0, xi), p0); acc1 = fe_add<F>(fe_mul<F>(acc1, xi), p1);
#pragma unroll 1
        for (int a = (int)in.n_old - 1; a >= 0; --a) {
            fe_t pw0 = zeta, pw1 = zetaw, e0 = kp.one, e1 = kp.one;
#pragma unroll 1
            for (int j = 15; j >= 0; --j) {
                const fe_t ch = x[PX_OLD + 16 * a + j];
                e0 = fe_mul<F>(e0, fe_add<F>(kp.one, fe_mul<F>(ch, pw0))); pw0 = fe_sqr<F>(pw0);
                e1 = fe_mul<F>(e1, fe_add<F>(kp.one, fe_mul<F>(ch, pw1))); pw1 = fe_sqr<F>(pw1);
            }
            acc0 = fe_add<F>(fe_mul<F>(acc0, xi), e0); acc1 = fe_add<F>(fe_mul<F>(acc1, xi), e1);
        }
        cip = fe_add<F>(acc0, fe_mul<F>(r, acc1));
    }
    fe_t bval;
    {
        fe_t pw0 = zeta, pw1 = zetaw, e0 = kp.one, e1 = kp.one;
#pragma unroll 1
        for (int j = 15; j >= 0; --j) {
            const fe_t ch = x[PX_BP + j];
            e0 = fe_mul<F>(e0, fe_add<F>(kp.one, fe_mul<F>(ch, pw0))); pw0 = fe_sqr<F>(pw0);
            e1 = fe_mul<F>(e1, fe_add<F>(kp.one, fe_mul<F>(ch, pw1))); pw1 = fe_sqr<F>(pw1);
        }
        bval = fe_add<F>(e0, fe_mul<F>(r, e1));
    }
    // ---- PreparedStatement::to_public_input(40): plain integers (every Fp value is < p < q: canonical in Fq as it stands)
    uint32_t *po = pub_out + (size_t)b * PK_PUB * 8;
    auto put = [&](uint32_t slot, const fe_t &plain) { for (int i = 0; i < 8; ++i) po[slot * 8 + i] = plain.v[i]; };
    auto put128 = [&](uint32_t slot, const uint32_t *p) { for (int i = 0; i < 4; ++i) { po[slot * 8 + i] = p[i]; po[slot * 8 + 4 + i] = 0; } };
    auto put_small = [&](uint32_t slot, uint32_t v) { po[slot * 8] = v; for (int i = 1; i < 8; ++i) po[slot * 8 + i] = 0; };
    const fe_t shift = fe_add<F>(kp.two255, kp.one);
    auto shifted = [&](const fe_t &v) { return fe_from_mont<F>(fe_mul<F>(fe_sub<F>(v, shift), kp.inv2)); };           // Shifted_value.Type1.of_field
    put(0, shifted(cip)); put(1, shifted(bval)); put(2, shifted(zeta_srs)); put(3, shifted(zeta_dom)); put(4, shifted(perm));
    const uint32_t *pl = in.plonk + (size_t)b * 16;
    put128(5, pl + 4); put128(6, pl + 8);                   // beta, gamma
    put128(7, pl); put128(8, pl + 12);                      // alpha, zeta
    put(9, x[PX_XIC]);
    put(10, fe_from_mont<FIELD_FQ>(fe_to_mont<FIELD_FQ>(load_fe<FIELD_FQ>(in.digest + (size_t)b * 8), kq.r2)));             // the digest mod q
    put(11, fe_from_mont<FIELD_FQ>(x[PX_MW]));
    put(12, fe_from_mont<F>(x[PX_MS]));
    for (uint32_t j = 0; j < 16; ++j) put128(13 + j, in.bp + ((size_t)b * 16 + j) * 4);
    { const uint32_t pv = misc[1]; ok = ok && pv <= 2; put_small(29, 4u * misc[0] + (pv == 0 ? 0u : (pv == 1 ? 2u : 3u))); }
    for (uint32_t j = 0; j < 8; ++j) put_small(30 + j, misc[2 + j] ? 1u : 0u);
    put_small(38, misc[10] ? 1u : 0u);
    if (misc[10]) put128(39, (const uint32_t *)(misc + 16)); else put_small(39, 0u);
    if (!ok) ok_out[b] = 0u;
}

}  // namespace mb

// queue the four stages on the current lane: d_pub (b*40*8 words) and d_ok (b u32, 1 = well-formed) are device buffers
int mb_pickles_dev(mina_ctx *c, size_t batch, const mb::PicklesIn &in, uint32_t *d_pub, uint32_t *d_ok) {
    if (!c->have_pickles_dev) return fail(MINA_ERR_STATE, "no step index installed");
    if (!c->have_kimchi) return fail(MINA_ERR_STATE, "no wrap verifier index installed");
    if (in.n_old > 4 || in.n_evals < mb::KC_COLS || in.n_evals > N_STEP_COLS) return fail(MINA_ERR_ARG, "bad n_old / n_evals");
    Lane &L = *c->L;
    int rc;
    if (!c->pickles_ms_valid) {     // the Tick sponge after the 28 wrap index commitments: once per (wrap index, step index) pair
        Lane *keep = c->L;
        std::vector<uint8_t> tape(28, MINA_TAPE_ABSORB_G), stt(96); uint32_t pos[2];
        std::vector<uint8_t> dummy(32);
        // the commitments are Pallas points (coordinates in Fp): absorbed by the Fp sponge = the CURVE_PALLAS tape
        if ((rc = mina_fq_sponge_run(c, CURVE_PALLAS, 1, tape.data(), tape.size(), nullptr, nullptr, c->kimchi_comms_host, dummy.data(), stt.data(), pos))) return rc;
        HIPC(hipStreamSynchronize(c->L->stream));
        mb::PicklesIndexDev *ixd = c->pickles_index.as<mb::PicklesIndexDev>();
        fe_t ms[3]; for (int e = 0; e < 3; ++e) ms[e] = to_mont_bytes<FIELD_FP>(stt.data() + 32 * e, c->fk[FIELD_FP]);
        HIPC(hipMemcpy(&ixd->ms_state, ms, sizeof ms, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(&ixd->ms_squeezed, pos, 8, hipMemcpyHostToDevice));
        c->pickles_ms_valid = true;
        c->L = keep;
    }
    const mb::PicklesIndexDev *ix = c->pickles_index.as<mb::PicklesIndexDev>();
    if ((rc = L.pk_xe.ensure(batch * mb::px_stride(in.n_old) * sizeof(fe_t)))) return rc;
    fe_t *xe = L.pk_xe.as<fe_t>();
    const FieldK &kp = c->fk[FIELD_FP], &kq = c->fk[FIELD_FQ];
    const PoseidonParams *ppp = c->pparams[FIELD_FP].as<PoseidonParams>(), *ppq = c->pparams[FIELD_FQ].as<PoseidonParams>();
    const uint32_t B = (uint32_t)batch;
    ProfScope ps_(c, PS_PICKLES);
    HIPC(hipMemsetD32Async((hipDeviceptr_t)d_ok, 1, batch, L.stream));   // a program that names an optional evaluation a proof lacks (outside a skipped region) fails THAT proof (pickles_scalar_kernel)
    mb::pickles_expand_kernel<<<cdiv(batch * (50 + 16 * in.n_old), 256), 256, 0, L.stream>>>(B, kp, kq, in, xe);
    if (use_coop16(c, batch)) {
        mb::pickles_digest_kernel<16><<<3 * coop_role_blocks<16>(batch), 64, 0, L.stream>>>(B, kp, kq, ppp, ppq, ix, in, xe, d_ok, coop_role_blocks<16>(batch));
        mb::pickles_tick_kernel<16><<<cdiv(coop_threads<16>(batch), 64), 64, 0, L.stream>>>(B, kp, ppp, in, xe, d_ok);
    } else if (use_coop8_transcripts(c, batch, 1024)) {
        mb::pickles_digest_kernel<8><<<3 * coop_role_blocks<8>(batch), 64, 0, L.stream>>>(B, kp, kq, ppp, ppq, ix, in, xe, d_ok, coop_role_blocks<8>(batch));
        mb::pickles_tick_kernel<8><<<cdiv(coop_threads<8>(batch), 64), 64, 0, L.stream>>>(B, kp, ppp, in, xe, d_ok);
    } else {
        mb::pickles_digest_kernel<3><<<3 * coop_role_blocks<3>(batch), 64, 0, L.stream>>>(B, kp, kq, ppp, ppq, ix, in, xe, d_ok, coop_role_blocks<3>(batch));
        mb::pickles_tick_kernel<3><<<cdiv(coop_threads<3>(batch), 64), 64, 0, L.stream>>>(B, kp, ppp, in, xe, d_ok);
    }
    mb::pickles_scalar_kernel<<<cdiv(batch, 64), 64, 0, L.stream>>>(B, kp, kq, ix, c->pickles_tokens.as<mb::KimchiToken>(), c->pickles_literals.as<fe_t>(), in, xe, d_pub, d_ok);
    HIPC(hipGetLastError());
    return MINA_OK;
}

static mb::PicklesIn pickles_in_of(const mina_pickles_statements &s) {
    auto W = [](const void *p) { return (const uint32_t *)p; };
    return mb::PicklesIn{s.n_old, s.n_evals, W(s.plonk), W(s.bulletproof_challenges), W(s.step_old_challenges), W(s.step_comms), W(s.wrap_old_challenges), W(s.wrap_sg),
                         W(s.sponge_digest), W(s.prev_evals), W(s.prev_public_input), W(s.prev_ft_eval1), W(s.app_state), (const uint8_t *)s.misc};
}
int mb_pickles_check(mina_ctx *c, const mina_pickles_statements *s) {
    if (!s || !s->plonk || !s->bulletproof_challenges || (s->n_old && (!s->step_old_challenges || !s->step_comms)) || !s->wrap_old_challenges || !s->wrap_sg || !s->sponge_digest ||
        !s->prev_evals || !s->prev_public_input || !s->prev_ft_eval1 || !s->app_state || !s->misc) return fail(MINA_ERR_ARG, "null statement section");
    if (s->n_old > 4 || s->n_evals < mb::KC_COLS || s->n_evals > N_STEP_COLS) return fail(MINA_ERR_ARG, "bad n_old / n_evals");
    if (!c->have_pickles_dev) return fail(MINA_ERR_STATE, "no step index installed");
    if (!c->have_kimchi) return fail(MINA_ERR_STATE, "no wrap verifier index installed");
    return MINA_OK;
}
int mb_pickles_statements_dev(mina_ctx *c, size_t batch, const mina_pickles_statements *s, uint32_t *d_pub, uint32_t *d_ok) { return mb_pickles_dev(c, batch, pickles_in_of(*s), d_pub, d_ok); }
// where the statements kernel of the current lane left kimchi's digest of the wrap-side old challenges: first element and stride (fe_t units)
void mb_pickles_kimchi_digest(mina_ctx *c, const mina_pickles_statements *s, const void **first, uint32_t *stride) {
    *first = c->L->pk_xe.as<fe_t>() + mb::px_kdig(s->n_old); *stride = mb::px_stride(s->n_old);
}
// per-proof byte strides of the statement sections, in the order of `mb_pickles_sections`
size_t mb_pickles_sections(const mina_pickles_statements *s, const void ***slots /* 12 */, size_t *strides /* 12 */, mina_pickles_statements *copy) {
    *copy = *s;
    const void **sl[12] = {&copy->plonk, &copy->bulletproof_challenges, &copy->step_old_challenges, &copy->step_comms, &copy->wrap_old_challenges, &copy->wrap_sg, &copy->sponge_digest,
                           &copy->prev_evals, &copy->prev_public_input, &copy->prev_ft_eval1, &copy->app_state, &copy->misc};
    const size_t st[12] = {64, 256, (size_t)s->n_old * 256, (size_t)s->n_old * 64, 480, 64, 32, (size_t)s->n_evals * 64, 64, 32, 32, 32};
    for (int i = 0; i < 12; ++i) { slots[i] = sl[i]; strides[i] = st[i]; }
    return 12;
}

extern "C" int mina_pickles_public_inputs_batch(mina_ctx *c, const mina_pickles_statements *s, size_t batch, uint8_t *pub_out, uint8_t *ok_out) {
    if (!c || !pub_out || !ok_out) return fail(MINA_ERR_ARG, "null argument");
    int rc = mb_pickles_check(c, s);
    if (rc) return rc;
    if (batch == 0 || batch > 65536) return fail(MINA_ERR_ARG, "bad batch");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    Lane &L = *c->L;
    mina_pickles_statements d; const void **slots[12]; size_t strides[12];
    mb_pickles_sections(s, slots, strides, &d);
    size_t total = 0, offs[12];
    for (int i = 0; i < 12; ++i) { offs[i] = total; total += (batch * strides[i] + 255) & ~(size_t)255; }
    if ((rc = L.host_stage.ensure(total + 16))) return rc;
    for (int i = 0; i < 12; ++i) if (strides[i]) memcpy((uint8_t *)L.host_stage.p + offs[i], *slots[i], batch * strides[i]);
    if ((rc = L.st_in.ensure(total + 16))) return rc;
    HIPC(hipMemcpyAsync(L.st_in.p, L.host_stage.p, total, hipMemcpyHostToDevice, L.stream));
    for (int i = 0; i < 12; ++i) *slots[i] = L.st_in.as<uint8_t>() + offs[i];
    if ((rc = L.pk_pub.ensure(batch * mb::PK_PUB * 32)) || (rc = L.pk_ok.ensure(batch * 4))) return rc;
    if ((rc = mb_pickles_dev(c, batch, pickles_in_of(d), L.pk_pub.as<uint32_t>(), L.pk_ok.as<uint32_t>()))) return rc;
    std::vector<uint32_t> okw(batch);
    HIPC(hipMemcpyAsync(pub_out, L.pk_pub.p, batch * mb::PK_PUB * 32, hipMemcpyDeviceToHost, L.stream));
    HIPC(hipMemcpyAsync(okw.data(), L.pk_ok.p, batch * 4, hipMemcpyDeviceToHost, L.stream));
    HIPC(hipStreamSynchronize(L.stream));
    for (size_t i = 0; i < batch; ++i) ok_out[i] = okw[i] ? 1 : 0;
    return MINA_OK;
}

static int upload_step_index(mina_ctx *c, const StepIndexHost &st) {
    const FieldK &k = c->fk[FIELD_FP];
    std::unique_ptr<mb::PicklesIndexDev> ix(new mb::PicklesIndexDev());
    memset(ix.get(), 0, sizeof *ix);
    ix->zk_rows = st.zk_rows; ix->n_tokens = (uint32_t)st.toks.size(); ix->n_domains = (uint32_t)st.domains.size(); ix->max_col = st.max_col;
    for (size_t d = 0; d < st.domains.size(); ++d) {
        const uint32_t lg = st.domains[d]; const uint64_t n = (uint64_t)1 << lg;
        ix->domain_log2[d] = lg;
        for (int i = 0; i < 7; ++i) ix->shifts[d][i] = st.shifts[d][i];
        fe_t w = k.root; for (uint32_t i = 0; i < 32 - lg; ++i) w = fe_sqr<FIELD_FP>(w);
        ix->omega[d] = w; ix->omega_zk[d] = mb::host_pow_u64<FIELD_FP>(w, n - st.zk_rows, k.one);
        for (uint32_t i = 0; i < st.zk_rows; ++i) ix->zk_roots[d][i] = mb::host_pow_u64<FIELD_FP>(w, n - st.zk_rows + i, k.one);
    }
    for (int i = 0; i < 9; ++i) ix->mds[i] = st.mds[i];
    ix->endo_coeff = st.endo_coeff;
    int rc;
    if ((rc = c->pickles_index.ensure(sizeof *ix)) || (rc = c->pickles_tokens.ensure((st.toks.size() ? st.toks.size() : 1) * sizeof(mb::KimchiToken))) ||
        (rc = c->pickles_literals.ensure((st.lits.size() ? st.lits.size() : 1) * sizeof(fe_t)))) return rc;
    HIPC(hipMemcpy(c->pickles_index.p, ix.get(), sizeof *ix, hipMemcpyHostToDevice));
    if (!st.toks.empty()) HIPC(hipMemcpy(c->pickles_tokens.p, st.toks.data(), st.toks.size() * sizeof(mb::KimchiToken), hipMemcpyHostToDevice));
    if (!st.lits.empty()) HIPC(hipMemcpy(c->pickles_literals.p, st.lits.data(), st.lits.size() * sizeof(fe_t), hipMemcpyHostToDevice));
    c->have_pickles_dev = true; c->pickles_ms_valid = false;
    return MINA_OK;
}
