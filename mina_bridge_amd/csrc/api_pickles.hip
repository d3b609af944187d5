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
#include "polish.h"
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
};
StepIndexHost &step_of(mina_ctx *c) {           // one per context, kept beside it (host-only data)
    static std::mutex mu; static std::vector<std::pair<mina_ctx *, StepIndexHost *>> all;
    std::lock_guard<std::mutex> lk(mu);
    for (auto &p : all) if (p.first == c) return *p.second;
    all.emplace_back(c, new StepIndexHost());
    return *all.back().second;
}

struct Derived { fe_t cip, b, zeta_srs, zeta_dom, perm, xi, r; mw::Chal128 xi_chal; };

}  // namespace

int mb_step_index_installed(mina_ctx *c) { return step_of(c).installed ? 1 : 0; }

extern "C" int mina_step_index_install(mina_ctx *c, const mina_step_index *si) {
    if (!c || !si || !si->domain_log2 || !si->shifts || (si->constant_term_len && !si->constant_term)) return fail(MINA_ERR_ARG, "null argument");
    if (si->n_domains == 0 || si->n_domains > 32 || si->zk_rows < 1 || si->zk_rows > 8) return fail(MINA_ERR_ARG, "bad n_domains / zk_rows");
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
        auto comb = [&](const std::vector<mw::B32> &chunks, const fe_t &ptn, bool &good) { fe_t acc = fe_zero(); for (size_t j = chunks.size(); j-- > 0;) { good = good && mw::fp_canonical(chunks[j].b); acc = fe_add<FIELD_FP>(fe_mul<FIELD_FP>(acc, ptn), to_mont_bytes<FIELD_FP>(chunks[j].b, kp)); } return acc; };
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
