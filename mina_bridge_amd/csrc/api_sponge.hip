// api_sponge.hip -- K2 (b_poly) and K3 (Poseidon, endo challenges) entry points.
#include "ctx.h"
#include "sponge.cuh"
#include "bpoly_mfma.cuh"

// ------------------------------------------------------------------------------------------------
// // K2
static BpolyShape bp_shape(uint32_t k, size_t batch) { BpolyShape s; s.k = k; s.lb = k / 2; s.hb = k - s.lb; s.batch = (uint32_t)batch; return s; }

template <int F>
static int run_bpoly_fold(mina_ctx *c, uint32_t k, size_t batch, const uint32_t *d_chals, const uint32_t *d_weights, uint32_t *d_out) {
    BpolyShape sh = bp_shape(k, batch);
    const uint32_t nl = 1u << sh.lb, nh = 1u << sh.hb, n = 1u << k;
    const uint32_t lo_blocks = cdiv(nl, 256), hi_tiles = cdiv(nh, BP_HT);
    uint32_t slices = 1;
    // enough blocks to fill 256 CUs x 4 when the batch is large
    while (slices * 2 <= batch && (size_t)lo_blocks * hi_tiles * slices < 2048 && slices < 64) slices *= 2;
    int rc;
    if (batch == 1 && !d_weights) {                              // b_poly_coefficients of one proof: one launch
        ProfScope ps_(c, PS_BPOLY_FOLD);
        bpoly_single_kernel<F><<<cdiv(nl, 256) * cdiv(nh, BP1_HT), nl < 256 ? (nl < 64 ? 64 : nl) : 256, 0, c->L->stream>>>(sh, c->fk[F], d_chals, nullptr, d_out);
        HIPC(hipGetLastError());
        return MINA_OK;
    }
    // large batches: the fold is a dense contraction over the batch -> int8 MFMA field-GEMM (bpoly_mfma.cuh).  mina_verify_tuning.bpoly_mfma = 0 keeps the VALU kernel
    const bool mfma_on = mb_tune().bpoly_mfma != 0;
    if (mfma_on && batch >= 256 && batch < (1u << 17) && sh.lb >= 1) {
        Lane &L = *c->L;
        const uint32_t kpad = (uint32_t)(cdiv(batch, BPM_KALIGN) * BPM_KALIGN);
        if ((rc = L.bp_ldig.ensure((size_t)nl * BPM_DIGITS * kpad)) || (rc = L.bp_hdig.ensure((size_t)nh * BPM_DIGITS * kpad)) ||
            (rc = L.bp_colsum.ensure((size_t)nh * nl * BPM_COLS * 8))) return rc;
        if (kpad != batch) { HIPC(hipMemsetAsync(L.bp_ldig.p, 0, (size_t)nl * BPM_DIGITS * kpad, L.stream)); HIPC(hipMemsetAsync(L.bp_hdig.p, 0, (size_t)nh * BPM_DIGITS * kpad, L.stream)); }
        { ProfScope ps_(c, PS_BPOLY_TABLES);
          if (sh.lb >= 3 && sh.hb >= 3) bpoly_tables_digits8_kernel<F><<<cdiv(batch * ((nl + nh) / 8), 256), 256, 0, L.stream>>>(sh, kpad, c->fk[F], d_chals, d_weights, L.bp_ldig.as<int8_t>(), L.bp_hdig.as<int8_t>());
          else bpoly_tables_digits_kernel<F><<<cdiv(batch * (nl + nh), 256), 256, 0, L.stream>>>(sh, kpad, c->fk[F], d_chals, d_weights, L.bp_ldig.as<int8_t>(), L.bp_hdig.as<int8_t>()); }
        { ProfScope ps_(c, PS_BPOLY_FOLD); bpoly_field_gemm_kernel<<<cdiv(nh, 4) * cdiv(nl, 4), 256, 0, L.stream>>>(nh, nl, kpad, L.bp_hdig.as<int8_t>(), L.bp_ldig.as<int8_t>(), L.bp_colsum.as<unsigned long long>()); }
        { ProfScope ps_(c, PS_BPOLY_FINISH); bpoly_colsum_reduce_kernel<F><<<cdiv(n, 256), 256, 0, L.stream>>>(nh, nl, sh.lb, c->fk[F], L.bp_colsum.as<unsigned long long>(), d_out); }
        HIPC(hipGetLastError());
        return MINA_OK;
    }
    if ((rc = c->L->bp_ltab.ensure(batch * nl * sizeof(fe_t)))) return rc;
    if ((rc = c->L->bp_htab.ensure(batch * nh * sizeof(fe_t)))) return rc;
    if ((rc = c->L->bp_partial.ensure((size_t)slices * n * sizeof(fe_t)))) return rc;
    { ProfScope ps_(c, PS_BPOLY_TABLES); bpoly_tables_kernel<F><<<cdiv(batch * (nl + nh), 256), 256, 0, c->L->stream>>>(sh, c->fk[F], d_chals, d_weights, c->L->bp_ltab.as<fe_t>(), c->L->bp_htab.as<fe_t>()); }
    { ProfScope ps_(c, PS_BPOLY_FOLD); bpoly_fold_kernel<F><<<lo_blocks * hi_tiles * slices, 256, 0, c->L->stream>>>(sh, slices, c->L->bp_ltab.as<fe_t>(), c->L->bp_htab.as<fe_t>(), c->L->bp_partial.as<fe_t>()); }
    { ProfScope ps_(c, PS_BPOLY_FINISH); bpoly_finish_kernel<F><<<cdiv(n, 256), 256, 0, c->L->stream>>>(n, slices, c->L->bp_partial.as<fe_t>(), d_out); }
    HIPC(hipGetLastError());
    return MINA_OK;
}

// b_poly_coefficients of `count` proofs (grid.y), each straight from its k 128-bit prechallenges -> count * 2^k scalars
int mb_bpoly_single_from_prechallenges(mina_ctx *c, int field, uint32_t k, const uint32_t *d_prechal, uint32_t *d_out, uint32_t count) {
    if (bad_field(field) || k < 1 || k > 20) return fail(MINA_ERR_ARG, "bad field or k");
    DISPATCH_FIELD(field, {
        BpolyShape sh = bp_shape(k, 1);
        const uint32_t nl = 1u << sh.lb, nh = 1u << sh.hb;
        ProfScope ps_(c, PS_BPOLY_FOLD);
        bpoly_single_kernel<F_><<<dim3(cdiv(nl, 256) * cdiv(nh, BP1_HT), count), nl < 256 ? (nl < 64 ? 64 : nl) : 256, 0, c->L->stream>>>(sh, c->fk[F_], nullptr, d_prechal, d_out);
    });
    HIPC(hipGetLastError());
    return MINA_OK;
}

int mb_bpoly_fold(mina_ctx *c, int field, uint32_t k, size_t batch, const uint32_t *d_chals, const uint32_t *d_weights, uint32_t *d_out) {
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    if (k < 1 || k > 20 || batch == 0 || batch > (1u << 24)) return fail(MINA_ERR_ARG, "bad k or batch");
    int rc = MINA_OK;
    DISPATCH_FIELD(field, { rc = run_bpoly_fold<F_>(c, k, batch, d_chals, d_weights, d_out); });
    return rc;
}

extern "C" int mina_b_poly_fold_dev(mina_ctx *c, int field, uint32_t k, size_t batch, const void *d_chals, const void *d_weights, void *d_out) {
    if (!c || !d_chals || !d_out) return fail(MINA_ERR_ARG, "null argument");
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    return mb_bpoly_fold(c, field, k, batch, (const uint32_t *)d_chals, (const uint32_t *)d_weights, (uint32_t *)d_out);
}

extern "C" int mina_b_poly_fold(mina_ctx *c, int field, uint32_t k, size_t batch, const uint8_t *chals, const uint8_t *weights, uint8_t *out) {
    if (!c || !chals || !out) return fail(MINA_ERR_ARG, "null argument");
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    if (k < 1 || k > 20 || batch == 0) return fail(MINA_ERR_ARG, "bad k or batch");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = h2d(c, c->L->tmp_a, chals, batch * k * 32))) return rc;
    if (weights && (rc = h2d(c, c->L->tmp_b, weights, batch * 32))) return rc;
    if ((rc = c->L->tmp_c.ensure(((size_t)1 << k) * 32))) return rc;
    if ((rc = mb_bpoly_fold(c, field, k, batch, c->L->tmp_a.as<uint32_t>(), weights ? c->L->tmp_b.as<uint32_t>() : nullptr, c->L->tmp_c.as<uint32_t>()))) return rc;
    return d2h_sync(c, out, c->L->tmp_c, ((size_t)1 << k) * 32);
}

extern "C" int mina_b_poly_coefficients(mina_ctx *c, int field, uint32_t k, const uint8_t *chals, uint8_t *out) {
    return mina_b_poly_fold(c, field, k, 1, chals, nullptr, out);
}

extern "C" int mina_b_poly(mina_ctx *c, int field, uint32_t k, const uint8_t *chals, size_t npoints, const uint8_t *xs, uint8_t *out) {
    if (!c || !chals || (npoints && (!xs || !out))) return fail(MINA_ERR_ARG, "null argument");
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    if (k < 1 || k > 32) return fail(MINA_ERR_ARG, "bad k");
    if (npoints == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = h2d(c, c->L->tmp_a, chals, (size_t)k * 32))) return rc;
    if ((rc = h2d(c, c->L->tmp_b, xs, npoints * 32))) return rc;
    if ((rc = c->L->tmp_c.ensure(npoints * 32))) return rc;
    DISPATCH_FIELD(field, { bpoly_eval_kernel<F_><<<cdiv(npoints, 64), 64, 0, c->L->stream>>>(k, (uint32_t)npoints, c->fk[F_], c->L->tmp_a.as<uint32_t>(), c->L->tmp_b.as<uint32_t>(), c->L->tmp_c.as<uint32_t>()); });
    return d2h_sync(c, out, c->L->tmp_c, npoints * 32);
}

// ------------------------------------------------------------------------------------------------
// // K3
template <int F> static void params_to_mont(const uint8_t *params, const FieldK &fk, PoseidonParams &pp) {
    auto load = [&](size_t idx) { fe_t w; memcpy(w.v, params + idx * 32, 32); return fe_to_mont<F>(w, fk.r2); };
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) pp.mds[i][j] = load(3 * i + j);
    for (int r = 0; r < 55; ++r) for (int j = 0; j < 3; ++j) pp.rc[r][j] = load(9 + 3 * r + j);
}

bool mb_params_are_surrogate(int field, const uint8_t *params);   // api_verify.hip (the compiled-in tables)
extern "C" int mina_poseidon_set_params(mina_ctx *c, int field, const uint8_t *params) {
    if (!c || !params) return fail(MINA_ERR_ARG, "null argument");
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    struct { PoseidonParams pp; PoseidonParams29 q; } both;
    PoseidonParams &pp = both.pp;
    static_assert(sizeof(both) == sizeof(PoseidonParams) + sizeof(PoseidonParams29), "PoseidonParams29 sits right behind PoseidonParams");
    memset(&both, 0, sizeof both);
    DISPATCH_FIELD(field, {
        params_to_mont<F_>(params, c->fk[F_], pp);
        // the 29-bit-limb copy (fp29.cuh): x 2^256 (what pp holds) times 2^5 = x 2^261, as an integer below p
        const FieldK &k = c->fk[F_];
        fe_t two5 = fe_zero(); two5.v[0] = 1u << 5; two5 = fe_to_mont<F_>(two5, k.r2);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) both.q.mds[i][j] = fe29_from_words(fe_mul<F_>(pp.mds[i][j], two5));
        for (int r = 0; r < 55; ++r) for (int i = 0; i < 3; ++i) both.q.rc[r][i] = fe29_from_words(fe_mul<F_>(pp.rc[r][i], two5));
        both.q.enter = fe29_from_words(fe_mul<F_>(two5, two5));   // Mont(2^10) = 2^10 2^256 = 2^266 mod p: (x 2^256)(2^266) / 2^261 = x 2^261
        const fe_t two522 = fe_to_mont<F_>(fe_mul<F_>(two5, two5), k.r2);   // (2^266)(2^512) / 2^256 = 2^522 mod p
        for (int r = 0; r < 55; ++r) for (int i = 0; i < 3; ++i) both.q.rc2[r][i] = fe29_from_words(fe_mul<F_>(pp.rc[r][i], two522));   // (rc 2^256)(2^522) / 2^256 = rc 2^522
        both.q.absorb = fe29_from_words(two522);                 // the integer 2^522 mod p: (canonical words)(2^522) / 2^261 = x 2^261
        both.q.leave = fe29_from_words(k.one);                   // Mont(1) = 2^256 mod p:           (x 2^261)(2^256) / 2^261 = x 2^256
    });
    int rc;
    if ((rc = c->pparams[field].ensure(sizeof both))) return rc;
    HIPC(hipMemcpyAsync(c->pparams[field].p, &both, sizeof both, hipMemcpyHostToDevice, c->L->stream));
    HIPC(hipStreamSynchronize(c->L->stream));
    c->have_pparams[field] = true;
    c->pparams_surrogate[field] = mb_params_are_surrogate(field, params);
    c->merkle_depth[field] = 0;
    if (field == FIELD_FP) c->have_state_salts = false;
    return MINA_OK;
}

static int poseidon_permute_on_lane(mina_ctx *c, int field, size_t n, void *d_states);
extern "C" int mina_poseidon_permute_dev(mina_ctx *c, int field, size_t n, void *d_states) {
    if (!c || !d_states) return fail(MINA_ERR_ARG, "null argument");
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    return poseidon_permute_on_lane(c, field, n, d_states);
}
static int poseidon_permute_on_lane(mina_ctx *c, int field, size_t n, void *d_states) {
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    if (!c->have_pparams[field]) return fail(MINA_ERR_STATE, "Poseidon constants not installed for this field");
    if (n == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    DISPATCH_FIELD(field, { poseidon_permute_kernel<F_><<<cdiv(n, 256), 256, 0, c->L->stream>>>((uint32_t)n, c->fk[F_], c->pparams[field].as<PoseidonParams>(), (uint32_t *)d_states); });
    HIPC(hipGetLastError());
    return MINA_OK;
}

extern "C" int mina_poseidon_permute(mina_ctx *c, int field, size_t n, uint8_t *states) {
    if (!c || (n && !states)) return fail(MINA_ERR_ARG, "null argument");
    if (n == 0) return MINA_OK;
    int rc;
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    if ((rc = h2d(c, c->L->tmp_a, states, n * 96))) return rc;
    if ((rc = poseidon_permute_on_lane(c, field, n, c->L->tmp_a.p))) return rc;
    return d2h_sync(c, states, c->L->tmp_a, n * 96);
}

extern "C" int mina_poseidon_hash(mina_ctx *c, int field, size_t n, size_t len, const uint8_t *inputs, uint8_t *out) {
    if (!c || (n && !out) || (n && len && !inputs)) return fail(MINA_ERR_ARG, "null argument");
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    if (!c->have_pparams[field]) return fail(MINA_ERR_STATE, "Poseidon constants not installed for this field");
    if (n == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = h2d(c, c->L->tmp_a, inputs, n * len * 32))) return rc;
    if ((rc = c->L->tmp_c.ensure(n * 32))) return rc;
    // up to 8192 sponges: 8 lanes per sponge (shortest dependent chain); above: wave-packed triples (21 sponges per wave) -- both run their
    // rounds on the 29-bit limbs (fp29.cuh), which beats one lane per sponge on the saturated 8 x 32 form at every size (19 k against 25 k
    // cycles per sponge-round), so the single-lane and 4-lane kernels are no longer dispatched here
    if (n <= COOP8_MAX_GROUPS) {
        DISPATCH_FIELD(field, { poseidon_hash_coop_kernel<F_, 8><<<cdiv(n * 8, 256), 256, 0, c->L->stream>>>((uint32_t)n, (uint32_t)len, c->fk[F_], c->pparams[field].as<PoseidonParams>(), c->L->tmp_a.as<uint32_t>(), c->L->tmp_c.as<uint32_t>()); });
    } else {
        DISPATCH_FIELD(field, { poseidon_hash_coop_kernel<F_, 3><<<cdiv(coop_threads<3>(n), 256), 256, 0, c->L->stream>>>((uint32_t)n, (uint32_t)len, c->fk[F_], c->pparams[field].as<PoseidonParams>(), c->L->tmp_a.as<uint32_t>(), c->L->tmp_c.as<uint32_t>()); });
    }
    return d2h_sync(c, out, c->L->tmp_c, n * 32);
}

extern "C" int mina_challenge_to_field(mina_ctx *c, int field, size_t n, const uint8_t *chal128, uint8_t *out) {
    if (!c || (n && (!chal128 || !out))) return fail(MINA_ERR_ARG, "null argument");
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    if (n == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = h2d(c, c->L->tmp_a, chal128, n * 16))) return rc;
    if ((rc = c->L->tmp_c.ensure(n * 32))) return rc;
    DISPATCH_FIELD(field, { challenge_to_field_kernel<F_><<<cdiv(n, 64), 64, 0, c->L->stream>>>((uint32_t)n, c->fk[F_], c->L->tmp_a.as<uint32_t>(), c->L->tmp_c.as<uint32_t>()); });
    return d2h_sync(c, out, c->L->tmp_c, n * 32);
}

// ------------------------------------------------------------------------------------------------
// a16: Merkle-path fold (Proof-of-Account)
static int merkle_prepare_salts(mina_ctx *c, int field, uint32_t depth) {
    if (c->merkle_depth[field] >= depth) return MINA_OK;
    // prefix element of height h: the 20 bytes "MinaMklTree%03d" padded with '*', read as a little-endian integer
    std::vector<uint8_t> pre((size_t)depth * 32, 0);
    for (uint32_t h = 0; h < depth; ++h) {
        char buf[32]; snprintf(buf, sizeof buf, "MinaMklTree%03u", h);
        size_t len = strlen(buf);
        for (size_t i = 0; i < 20; ++i) pre[(size_t)h * 32 + i] = (uint8_t)(i < len ? buf[i] : '*');
    }
    int rc;
    if ((rc = c->merkle_salts[field].ensure((size_t)depth * 3 * sizeof(fe_t)))) return rc;
    if ((rc = h2d(c, c->L->tmp_d, pre.data(), pre.size()))) return rc;
    DISPATCH_FIELD(field, { merkle_salt_kernel<F_><<<cdiv(depth, 64), 64, 0, c->L->stream>>>(depth, c->fk[F_], c->pparams[field].as<PoseidonParams>(), c->L->tmp_d.as<uint32_t>(), c->merkle_salts[field].as<fe_t>()); });
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(c->L->stream));
    c->merkle_depth[field] = depth;
    return MINA_OK;
}

// Merkle fold of n paths with everything in HBM, queued on the current lane (salts must have been prepared: merkle_prepare_salts)
int mb_merkle_fold_dev(mina_ctx *c, int field, size_t n, uint32_t depth, const uint32_t *d_leaves, const uint32_t *d_sib, const uint8_t *d_dirs, uint32_t *d_roots) {
    if (use_coop16(c, n)) {                                      // a few paths: 16 lanes each (shortest chain)
        DISPATCH_FIELD(field, {
            merkle_fold_coop_kernel<F_, 16><<<cdiv(n * 16, 256), 256, 0, c->L->stream>>>((uint32_t)n, depth, c->fk[F_], c->pparams[field].as<PoseidonParams>(),
                c->merkle_salts[field].as<fe_t>(), d_leaves, d_sib, d_dirs, d_roots);
        });
    } else if (n <= COOP8_MAX_GROUPS) {                          // latency-bound batch: 8 lanes per path
        DISPATCH_FIELD(field, {
            merkle_fold_coop_kernel<F_, 8><<<cdiv(n * 8, 256), 256, 0, c->L->stream>>>((uint32_t)n, depth, c->fk[F_], c->pparams[field].as<PoseidonParams>(),
                c->merkle_salts[field].as<fe_t>(), d_leaves, d_sib, d_dirs, d_roots);
        });
    } else {
        DISPATCH_FIELD(field, {
            merkle_fold_coop_kernel<F_, 3><<<cdiv(coop_threads<3>(n), 256), 256, 0, c->L->stream>>>((uint32_t)n, depth, c->fk[F_], c->pparams[field].as<PoseidonParams>(),
                c->merkle_salts[field].as<fe_t>(), d_leaves, d_sib, d_dirs, d_roots);
        });
    }
    HIPC(hipGetLastError());
    return MINA_OK;
}
int mb_merkle_prepare_salts(mina_ctx *c, int field, uint32_t depth) { return merkle_prepare_salts(c, field, depth); }

extern "C" int mina_merkle_roots(mina_ctx *c, int field, size_t n, uint32_t depth, const uint8_t *leaves, const uint8_t *siblings,
                                 const uint8_t *dirs, uint8_t *roots_out) {
    if (!c || (n && (!leaves || !roots_out)) || (n && depth && (!siblings || !dirs))) return fail(MINA_ERR_ARG, "null argument");
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    if (depth > 64) return fail(MINA_ERR_ARG, "depth must be <= 64");
    if (n > (1u << 24)) return fail(MINA_ERR_ARG, "n too large");
    if (!c->have_pparams[field]) return fail(MINA_ERR_STATE, "Poseidon constants not installed for this field");
    if (n == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if (depth && (rc = merkle_prepare_salts(c, field, depth))) return rc;
    if ((rc = h2d(c, c->L->tmp_a, leaves, n * 32))) return rc;
    if ((rc = h2d(c, c->L->tmp_b, siblings, n * depth * 32))) return rc;
    if ((rc = h2d(c, c->L->tmp_d, dirs, n * depth))) return rc;
    if ((rc = c->L->tmp_c.ensure(n * 32))) return rc;
    if ((rc = mb_merkle_fold_dev(c, field, n, depth, c->L->tmp_a.as<uint32_t>(), c->L->tmp_b.as<uint32_t>(), c->L->tmp_d.as<uint8_t>(), c->L->tmp_c.as<uint32_t>()))) return rc;
    return d2h_sync(c, roots_out, c->L->tmp_c, n * 32);
}

extern "C" int mina_merkle_verify_batch(mina_ctx *c, int field, size_t n, uint32_t depth, const uint8_t *leaves, const uint8_t *siblings,
                                        const uint8_t *dirs, const uint8_t *expected_roots, uint8_t *verdicts) {
    if (!expected_roots || !verdicts) return fail(MINA_ERR_ARG, "null argument");
    std::vector<uint8_t> roots(n * 32);
    int rc = mina_merkle_roots(c, field, n, depth, leaves, siblings, dirs, roots.data());
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) verdicts[i] = memcmp(&roots[i * 32], expected_roots + i * 32, 32) == 0 ? 1 : 0;
    return MINA_OK;
}
