/*
 * oracle/composite_oracle.c -- NATIVE CPU composite of one Proof-of-State verification (BASELINE config C1),
 * threaded ACROSS PROOFS with pthreads: what `cpu_baseline` times beside the GPU path.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as pasta_oracle.c): only tests/ and bench.py's cpu_baseline leg load this file.
 *
 * It is a C restatement of the same published algorithms the Python composite restates -- and is cross-checked against it
 * value by value on the committed fixtures (tests/test_native_composite.py):
 *   oracle/mina_state_ref.py   protocol_state_hash            -> state_hashes()            (17 x MinaHash + linkage; README.md:283-288)
 *   oracle/pickles_ref.py      statement_public_input         -> pickles_public_input()    (openmina verify_block / compute_deferred_values)
 *   oracle/kimchi_ref.py       oracles_and_batch              -> kimchi_oracles()          (kimchi verifier::{oracles, to_batch}; README.md:413-475)
 *   oracle/ipa_ref.py          ipa_verify_batch               -> ipa_verify_one()          (poly-commitment SRS::verify; README.md:469-475)
 *   oracle/state_job_ref.py    accumulator_ok                 -> accumulator_ok()          (openmina accumulator_check; README.md:534-544)
 * [UPSTREAM-RECALL] like those files: the crates are un-vendored (core/Cargo.toml:14-25), "parity unpinned" except the SRS KAT.
 * Inputs are the C-ABI's own per-proof byte layouts (include/mina_verify.h: mina_pickles_statements, mina_kimchi_proofs, the opening and
 * accumulator sections of mina_state_jobs, protocol-state records) so that the same bytes feed the GPU job and this checker; the
 * linearizations are PolishToken byte-code (MINA_TOK_*).  One difference from the Python driver, as in kimchi itself: the public-input
 * commitment is a 40-term MSM over CACHED Lagrange-basis commitments (computed once per index by oc_setup), not an inverse FFT per proof.
 *
 * The arithmetic is pasta_oracle.c's (4 x u64 Montgomery), included below so that its static routines are usable here.
 */
#include "pasta_oracle.c"

#include <stdio.h>

/* ---------------------------------------------------------------------------------------------- small helpers */
static int mw_canonical(const uint8_t *b, const fctx *f) { fe x; fe_from_bytes(&x, b); return !ge256(&x, &f->p); }
static void f_from_le128(fe *r, const uint8_t *b16, const fctx *f) { fe t = {{0, 0, 0, 0}}; memcpy(&t.v[0], b16, 8); memcpy(&t.v[1], b16 + 8, 8); f_to_mont(r, &t, f); }
static void f_from_le256_reduce(fe *r, const uint8_t *b32, const fctx *f) {     /* any 256-bit integer mod p: lo + 2^128 hi */
    fe lo, hi, t128 = {{0, 0, 1, 0}}, t;
    f_from_le128(&lo, b32, f); f_from_le128(&hi, b32 + 16, f); f_to_mont(&t, &t128, f);
    f_mul(&hi, &hi, &t, f); f_add(r, &lo, &hi, f);
}
static void f_pow_u64(fe *r, const fe *a, uint64_t e, const fctx *f) { fe b = *a, acc = f->one; for (; e; e >>= 1) { if (e & 1) f_mul(&acc, &acc, &b, f); f_sqr(&b, &b, f); } *r = acc; }
static void domain_generator(fe *w, int log2, const fctx *f) { *w = f->root; for (int i = 0; i < 32 - log2; ++i) f_sqr(w, w, f); }
static void endo_of(fe *e, const fctx *f, int squared) {                        /* cube root of unity from g = 5 (squared: endo_r) */
    fe one = {{1, 0, 0, 0}}, pm1, ex; sub256(&pm1, &f->p, &one);
    u128 rem = 0; for (int l = 3; l >= 0; --l) { u128 cur = (rem << 64) | pm1.v[l]; ex.v[l] = (uint64_t)(cur / 3); rem = cur % 3; }
    f_pow(e, &f->five, &ex, f); if (squared) f_sqr(e, e, f);
}
/* kimchi ScalarChallenge::to_field on a 128-bit challenge (16 bytes LE) */
static void chal_to_field(fe *out, const uint8_t *c16, const fe *endo, const fctx *f) {
    fe a, b, neg1; f_from_u64(&a, 2, f); b = a; f_neg(&neg1, &f->one, f);
    for (int i = 63; i >= 0; --i) {
        f_dbl(&a, &a, f); f_dbl(&b, &b, f);
        const int r0 = (c16[(2 * i) >> 3] >> ((2 * i) & 7)) & 1, r1 = (c16[(2 * i + 1) >> 3] >> ((2 * i + 1) & 7)) & 1;
        const fe *s = r0 ? &f->one : &neg1;
        if (r1 == 0) f_add(&b, &b, s, f); else f_add(&a, &a, s, f);
    }
    f_mul(&a, &a, endo, f); f_add(out, &a, &b, f);
}
static void b_poly_eval(fe *out, const fe *chals, int k, const fe *x, const fctx *f) {
    fe pw[32], r = f->one; pw[0] = *x; for (int i = 1; i < k; ++i) f_sqr(&pw[i], &pw[i - 1], f);
    for (int i = 0; i < k; ++i) { fe t; f_mul(&t, &chals[i], &pw[k - 1 - i], f); f_add(&t, &t, &f->one, f); f_mul(&r, &r, &t, f); }
    *out = r;
}

/* ---------------------------------------------------------------------------------------------- the sponge (pasta_ref.Sponge / ipa_ref.FqSponge) */
typedef struct { fe s[3]; int squeezed, count; const pparams *pp; const fctx *f; } sponge;
static void sp_init(sponge *sp, const pparams *pp, const fctx *f) { memset(sp->s, 0, sizeof sp->s); sp->squeezed = 0; sp->count = 0; sp->pp = pp; sp->f = f; }
static void sp_absorb(sponge *sp, const fe *x) {
    if (!sp->squeezed) {
        if (sp->count == 2) { poseidon_perm(sp->s, sp->pp, sp->f); f_add(&sp->s[0], &sp->s[0], x, sp->f); sp->count = 1; }
        else { f_add(&sp->s[sp->count], &sp->s[sp->count], x, sp->f); sp->count++; }
    } else { f_add(&sp->s[0], &sp->s[0], x, sp->f); sp->squeezed = 0; sp->count = 1; }
}
static void sp_squeeze(sponge *sp, fe *out) {
    if (!sp->squeezed || sp->count == 2) { poseidon_perm(sp->s, sp->pp, sp->f); sp->squeezed = 1; sp->count = 1; *out = sp->s[0]; return; }
    *out = sp->s[sp->count]; sp->count++;
}
static void sp_absorb_pt(sponge *sp, const uint8_t *pt64) { fe x, y; f_load(&x, pt64, sp->f); f_load(&y, pt64 + 32, sp->f); sp_absorb(sp, &x); sp_absorb(sp, &y); }   /* infinity = (0, 0) */
static void sp_challenge128(sponge *sp, uint8_t out16[16]) { fe x, p; sp_squeeze(sp, &x); f_from_mont(&p, &x, sp->f); memcpy(out16, &p.v[0], 8); memcpy(out16 + 8, &p.v[1], 8); }
/* absorb a scalar-field element `x` (Montgomery in fs) into a sponge over the base field fb of the same curve */
static void sp_absorb_fr(sponge *sp, const fe *x, const fctx *fs) {
    const fctx *fb = sp->f; fe plain; f_from_mont(&plain, x, fs);
    if (!ge256(&fs->p, &fb->p)) { fe t; f_to_mont(&t, &plain, fb); sp_absorb(sp, &t); return; }          /* scalar modulus < base modulus: whole */
    fe hi = plain, lo = {{plain.v[0] & 1, 0, 0, 0}}, t; shr1(&hi);
    f_to_mont(&t, &hi, fb); sp_absorb(sp, &t); f_to_mont(&t, &lo, fb); sp_absorb(sp, &t);
}

/* ---------------------------------------------------------------------------------------------- PolishToken byte-code (include/mina_verify.h MINA_TOK_*) */
enum { TOK_ALPHA = 0, TOK_BETA, TOK_GAMMA, TOK_JOINT, TOK_ENDO, TOK_MDS, TOK_LITERAL, TOK_CELL, TOK_DUP, TOK_POW, TOK_ADD, TOK_MUL, TOK_SUB, TOK_VANISH_ZK, TOK_LAGRANGE, TOK_STORE, TOK_LOAD };
typedef struct { fe alpha, beta, gamma, endo, zeta, zeta_n_minus_1, omega, zkpm; const fe *mds; const fe (*evals)[2]; int n_evals, log2_domain, zk_rows; } polish_env;
static int polish_eval(fe *out, const uint8_t *code, size_t len, const polish_env *e, const fctx *f) {
    fe stack[64], cache[32]; int sp = 0, nc = 0; size_t p = 0;
    while (p < len) {
        const uint8_t op = code[p++];
        if (sp >= 62) return -1;
        switch (op) {
            case TOK_ALPHA: stack[sp++] = e->alpha; break;
            case TOK_BETA: stack[sp++] = e->beta; break;
            case TOK_GAMMA: stack[sp++] = e->gamma; break;
            case TOK_JOINT: memset(&stack[sp++], 0, sizeof(fe)); break;
            case TOK_ENDO: stack[sp++] = e->endo; break;
            case TOK_MDS: if (len - p < 2 || code[p] > 2 || code[p + 1] > 2) return -1; stack[sp++] = e->mds[3 * code[p] + code[p + 1]]; p += 2; break;
            case TOK_LITERAL: if (len - p < 32) return -1; f_load(&stack[sp++], code + p, f); p += 32; break;
            case TOK_CELL: if (len - p < 2 || code[p] >= e->n_evals || code[p + 1] > 1) return -1; stack[sp++] = e->evals[code[p]][code[p + 1]]; p += 2; break;
            case TOK_DUP: if (sp < 1) return -1; stack[sp] = stack[sp - 1]; ++sp; break;
            case TOK_POW: { if (len - p < 8 || sp < 1) return -1; uint64_t ex; memcpy(&ex, code + p, 8); p += 8; f_pow_u64(&stack[sp - 1], &stack[sp - 1], ex, f); break; }
            case TOK_ADD: if (sp < 2) return -1; f_add(&stack[sp - 2], &stack[sp - 2], &stack[sp - 1], f); --sp; break;
            case TOK_MUL: if (sp < 2) return -1; f_mul(&stack[sp - 2], &stack[sp - 2], &stack[sp - 1], f); --sp; break;
            case TOK_SUB: if (sp < 2) return -1; f_sub(&stack[sp - 2], &stack[sp - 2], &stack[sp - 1], f); --sp; break;
            case TOK_VANISH_ZK: stack[sp++] = e->zkpm; break;
            case TOK_LAGRANGE: {
                if (len - p < 4) return -1; int32_t off; memcpy(&off, code + p, 4); p += 4;
                const uint64_t n = (uint64_t)1 << e->log2_domain;
                const uint64_t row = off >= 0 ? (uint64_t)off : n - (uint64_t)e->zk_rows - (off == INT32_MIN ? 0 : (uint64_t)(-(int64_t)off));
                fe wr, d; f_pow_u64(&wr, &e->omega, row, f); f_sub(&d, &e->zeta, &wr, f); f_inv(&d, &d, f); f_mul(&stack[sp++], &e->zeta_n_minus_1, &d, f); break; }
            case TOK_STORE: if (sp < 1 || nc >= 32) return -1; cache[nc++] = stack[sp - 1]; break;
            case TOK_LOAD: { if (len - p < 2) return -1; const int ix = code[p] | (code[p + 1] << 8); p += 2; if (ix >= nc) return -1; stack[sp++] = cache[ix]; break; }
            default: return -1;
        }
    }
    if (len == 0) { memset(out, 0, sizeof *out); return 0; }
    if (sp != 1) return -1;
    *out = stack[0]; return 0;
}

/* ---------------------------------------------------------------------------------------------- set-up data (one per process) */
typedef struct {
    /* SRS: canonical affine bytes */
    const uint8_t *g_pallas; const uint8_t *h_pallas; const uint8_t *g_vesta;       /* 2^k_wrap.., 64, 2^16 x 64 */
    pparams pp[2];                                                                   /* Poseidon tables: [0] Fp, [1] Fq */
    /* wrap index (kimchi VerifierIndex<Pallas>) */
    int log2_domain, zk_rows, perm_alpha_offset;
    fe shifts[7];                                                                    /* Fq */
    uint8_t sigma_comm[7 * 64], coeff_comm[15 * 64], sel_comm[6 * 64];
    const uint8_t *ct; size_t ct_len;                                                /* constant term byte-code (Fq literals) */
    fe index_digest;                                                                 /* Fp */
    uint8_t lagrange[64 * 64];                                                       /* commitments of the first 40 (<= 64) Lagrange basis polynomials */
    int n_lagrange;
    /* step index */
    int step_zk_rows, n_step_domains; int step_domain_log2[8]; fe step_shifts[8][7]; /* Fp */
    const uint8_t *step_ct; size_t step_ct_len;
    fe tick_after_index[3]; int tick_after_squeezed, tick_after_count;              /* Tick sponge after the 28 wrap index commitments */
} oc_setup_t;

static oc_setup_t G;

/* inverse FFT over the radix-2 domain (in place, Montgomery values) */
static void ifft(fe *a, int log2, const fctx *f) {
    const size_t n = (size_t)1 << log2;
    for (size_t i = 1, j = 0; i < n; ++i) { size_t bit = n >> 1; for (; j & bit; bit >>= 1) j ^= bit; j |= bit; if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; } }
    fe w, winv; domain_generator(&w, log2, f); f_inv(&winv, &w, f);
    for (size_t len = 2; len <= n; len <<= 1) {
        fe wl; f_pow_u64(&wl, &winv, n / len, f);
        for (size_t st = 0; st < n; st += len) {
            fe x = f->one;
            for (size_t t = 0; t < len / 2; ++t) { fe u = a[st + t], v; f_mul(&v, &a[st + t + len / 2], &x, f); f_add(&a[st + t], &u, &v, f); f_sub(&a[st + t + len / 2], &u, &v, f); f_mul(&x, &x, &wl, f); }
        }
    }
    fe nn, ninv; f_from_u64(&nn, n, f); f_inv(&ninv, &nn, f);
    for (size_t i = 0; i < n; ++i) f_mul(&a[i], &a[i], &ninv, f);
}

/* one-off: tables, index digest, Lagrange commitments (threads: for the 40 set-up MSMs) */
int oc_setup(const uint8_t *g_pallas, const uint8_t *h_pallas, const uint8_t *g_vesta, const uint8_t *params_fp, const uint8_t *params_fq,
             int log2_domain, int zk_rows, int perm_alpha_offset, const uint8_t *shifts /* 7*32 Fq */, const uint8_t *sigma_comm, const uint8_t *coeff_comm,
             const uint8_t *sel_comm, const uint8_t *ct, size_t ct_len,
             int step_zk_rows, int n_step_domains, const uint32_t *step_domain_log2, const uint8_t *step_shifts /* n*7*32 Fp */, const uint8_t *step_ct, size_t step_ct_len,
             int n_lagrange, int threads) {
    oracle_init();
    if (n_step_domains > 8 || n_lagrange > 64 || log2_domain > 20) return -1;
    const fctx *fp = &F[0], *fq = &F[1];
    G.g_pallas = g_pallas; G.h_pallas = h_pallas; G.g_vesta = g_vesta;
    load_pparams(&G.pp[0], params_fp, fp); load_pparams(&G.pp[1], params_fq, fq);
    G.log2_domain = log2_domain; G.zk_rows = zk_rows; G.perm_alpha_offset = perm_alpha_offset;
    for (int i = 0; i < 7; ++i) f_load(&G.shifts[i], shifts + 32 * i, fq);
    memcpy(G.sigma_comm, sigma_comm, sizeof G.sigma_comm); memcpy(G.coeff_comm, coeff_comm, sizeof G.coeff_comm); memcpy(G.sel_comm, sel_comm, sizeof G.sel_comm);
    uint8_t *c1 = (uint8_t *)malloc(ct_len + 1), *c2 = (uint8_t *)malloc(step_ct_len + 1);
    memcpy(c1, ct, ct_len); memcpy(c2, step_ct, step_ct_len);
    G.ct = c1; G.ct_len = ct_len; G.step_ct = c2; G.step_ct_len = step_ct_len;
    G.step_zk_rows = step_zk_rows; G.n_step_domains = n_step_domains;
    for (int d = 0; d < n_step_domains; ++d) { G.step_domain_log2[d] = (int)step_domain_log2[d]; for (int i = 0; i < 7; ++i) f_load(&G.step_shifts[d][i], step_shifts + ((size_t)d * 7 + i) * 32, fp); }
    /* VerifierIndex::digest: Fq-sponge (over Fp) of sigma, coefficients, selectors; the same prefix opens messages_for_next_step_proof */
    sponge sp; sp_init(&sp, &G.pp[0], fp);
    for (int i = 0; i < 7; ++i) sp_absorb_pt(&sp, G.sigma_comm + 64 * i);
    for (int i = 0; i < 15; ++i) sp_absorb_pt(&sp, G.coeff_comm + 64 * i);
    for (int i = 0; i < 6; ++i) sp_absorb_pt(&sp, G.sel_comm + 64 * i);
    memcpy(G.tick_after_index, sp.s, sizeof sp.s); G.tick_after_squeezed = sp.squeezed; G.tick_after_count = sp.count;
    sponge dg = sp; sp_squeeze(&dg, &G.index_digest);
    /* Lagrange-basis commitments L_i = commit(ifft(e_i)) for i < n_lagrange (kimchi caches them with the SRS: `add_lagrange_basis`) */
    G.n_lagrange = n_lagrange;
    const size_t n = (size_t)1 << log2_domain;
    fe *a = (fe *)malloc(n * sizeof(fe)); uint8_t *sc = (uint8_t *)malloc(n * 32);
    for (int i = 0; i < n_lagrange; ++i) {
        memset(a, 0, n * sizeof(fe)); a[i] = fq->one; ifft(a, log2_domain, fq);
        for (size_t j = 0; j < n; ++j) f_store(sc + 32 * j, &a[j], fq);
        oracle_msm_pippenger(0, n, g_pallas, sc, G.lagrange + 64 * i, threads);
    }
    free(a); free(sc);
    return 0;
}

/* ---------------------------------------------------------------------------------------------- per-proof input / output */
typedef struct {
    /* protocol states: 17 records of 64 x 32 bytes, field counts, expected hashes */
    const uint8_t *records; const uint32_t *nfields; const uint8_t *expected;
    /* Pickles statement (mina_pickles_statements, one proof) */
    uint32_t n_old, n_evals;
    const uint8_t *plonk, *bp, *old_chals, *step_comms, *wrap_old, *wrap_sg, *sponge_digest, *prev_evals, *prev_pi, *prev_ft1, *app_state, *misc;
    /* kimchi proof (mina_kimchi_proofs) + opening */
    const uint8_t *prev_comms, *w_comm, *z_comm, *t_comm, *evals, *ft_eval1, *lr, *delta, *sg, *z1, *z2;
    /* step accumulator */
    const uint8_t *acc_pre, *acc_sg;
} oc_proof;
typedef struct {
    uint8_t hashes[17 * 32], pubs[40 * 32], ft_eval0[32], cip[32], v[32], u[32], step_cip[32], step_b[32], public_comm[64];
    int chain_ok, statement_ok, ipa_ok, acc_ok, verdict;
} oc_result;

/* ---- 1. MinaHash of the 17 states (oracle/mina_state_ref.py protocol_state_hash) + linkage */
static void prefix_salt(fe s[3], const char *name, const fctx *f) {
    uint8_t b[32]; memset(b, 0, 32); size_t n = strlen(name); for (size_t i = 0; i < 20; ++i) b[i] = (uint8_t)(i < n ? name[i] : '*');
    memset(s, 0, 3 * sizeof(fe)); f_load(&s[0], b, f); poseidon_perm(s, &G.pp[0], f);
}
static int state_hashes(const oc_proof *p, oc_result *r) {
    const fctx *f = &F[0];
    static fe salt_body[3], salt_state[3]; static int have = 0;
    if (!have) { prefix_salt(salt_body, "MinaProtoStateBody", f); prefix_salt(salt_state, "MinaProtoState", f); have = 1; }
    int ok = 1;
    for (int s = 0; s < 17; ++s) {
        const uint8_t *rec = p->records + (size_t)s * 64 * 32; const uint32_t nf = p->nfields[s];
        if (nf > 63) return 0;
        fe st[3]; memcpy(st, salt_body, sizeof st); int count = 0;
        for (uint32_t e = 0; e < nf; ++e) { if (count == 2) { poseidon_perm(st, &G.pp[0], f); count = 0; } fe x; f_load(&x, rec + 32 * (1 + e), f); f_add(&st[count], &st[count], &x, f); ++count; }
        poseidon_perm(st, &G.pp[0], f);
        fe body = st[0], prev; memcpy(st, salt_state, sizeof st); f_load(&prev, rec, f);
        f_add(&st[0], &st[0], &prev, f); f_add(&st[1], &st[1], &body, f); poseidon_perm(st, &G.pp[0], f);
        f_store(r->hashes + 32 * s, &st[0], f);
        ok = ok && memcmp(r->hashes + 32 * s, p->expected + 32 * s, 32) == 0;
    }
    for (int s = 1; s < 16; ++s) ok = ok && memcmp(p->records + (size_t)s * 64 * 32, r->hashes + 32 * (s - 1), 32) == 0;
    return ok;
}

/* ---- 2. Pickles statement -> the wrap circuit's 40 public inputs (oracle/pickles_ref.py) */
static int pickles_public_input(const oc_proof *p, oc_result *r, fe pubs[40]) {
    const fctx *fp = &F[0], *fq = &F[1];
    if (p->n_old > 4 || p->n_evals < 43 || p->n_evals > 62) return 0;
    fe endo_fp, endo_fq; endo_of(&endo_fp, fp, 1); endo_of(&endo_fq, fq, 1);        /* endo_r of Vesta (in Fp) / of Pallas (in Fq) */
    fe alpha, zeta, beta, gamma;
    chal_to_field(&alpha, p->plonk, &endo_fp, fp); chal_to_field(&zeta, p->plonk + 48, &endo_fp, fp);
    f_from_le128(&beta, p->plonk + 16, fp); f_from_le128(&gamma, p->plonk + 32, fp);
    const int k = p->misc[0]; int dom = -1;
    for (int d = 0; d < G.n_step_domains; ++d) if (G.step_domain_log2[d] == k) dom = d;
    if (dom < 0 || p->misc[1] > 2) return 0;
    const uint64_t n = (uint64_t)1 << k;
    fe omega, zetaw; domain_generator(&omega, k, fp); f_mul(&zetaw, &zeta, &omega, fp);
    fe ev[62][2];
    for (uint32_t j = 0; j < p->n_evals; ++j) for (int sd = 0; sd < 2; ++sd) { if (!mw_canonical(p->prev_evals + (j * 2 + sd) * 32, fp)) return 0; f_load(&ev[j][sd], p->prev_evals + (j * 2 + sd) * 32, fp); }
    fe bp[16], old[4][16];
    for (int i = 0; i < 16; ++i) chal_to_field(&bp[i], p->bp + 16 * i, &endo_fp, fp);
    for (uint32_t a = 0; a < p->n_old; ++a) for (int i = 0; i < 16; ++i) chal_to_field(&old[a][i], p->old_chals + ((size_t)a * 16 + i) * 16, &endo_fp, fp);
    fe pi0, pi1, ft1; f_load(&pi0, p->prev_pi, fp); f_load(&pi1, p->prev_pi + 32, fp); f_load(&ft1, p->prev_ft1, fp);
    /* Tick sponge: xi, r */
    sponge sp, ch; sp_init(&sp, &G.pp[0], fp); sp_init(&ch, &G.pp[0], fp);
    fe dgs; f_from_le256_reduce(&dgs, p->sponge_digest, fp); sp_absorb(&sp, &dgs);
    for (uint32_t a = 0; a < p->n_old; ++a) for (int i = 0; i < 16; ++i) sp_absorb(&ch, &old[a][i]);
    fe chd; sp_squeeze(&ch, &chd); sp_absorb(&sp, &chd);
    sp_absorb(&sp, &ft1); sp_absorb(&sp, &pi0); sp_absorb(&sp, &pi1);
    for (uint32_t j = 0; j < p->n_evals; ++j) { sp_absorb(&sp, &ev[j][0]); sp_absorb(&sp, &ev[j][1]); }
    uint8_t xi_c[16], r_c[16]; sp_challenge128(&sp, xi_c); sp_challenge128(&sp, r_c);
    fe xi, rr; chal_to_field(&xi, xi_c, &endo_fp, fp); chal_to_field(&rr, r_c, &endo_fp, fp);
    /* ft_eval0 of the step proof, derive_plonk */
    const fe *shifts = G.step_shifts[dom];
    fe zkp = fp->one;
    for (uint64_t i = n - (uint64_t)G.step_zk_rows; i < n; ++i) { fe wi, d; f_pow_u64(&wi, &omega, i, fp); f_sub(&d, &zeta, &wi, fp); f_mul(&zkp, &zkp, &d, fp); }
    fe a0, a1, a2; f_pow_u64(&a0, &alpha, 21, fp); f_mul(&a1, &a0, &alpha, fp); f_mul(&a2, &a1, &alpha, fp);
#define W_(i) ev[7 + (i)][0]
#define S_(i) ev[37 + (i)][0]
    const fe z0 = ev[0][0], z1 = ev[0][1];
    fe zeta_n, zeta1m1; f_pow_u64(&zeta_n, &zeta, n, fp); f_sub(&zeta1m1, &zeta_n, &fp->one, fp);
    fe ft, t; f_add(&ft, &W_(6), &gamma, fp); f_mul(&ft, &ft, &z1, fp); f_mul(&ft, &ft, &a0, fp); f_mul(&ft, &ft, &zkp, fp);
    for (int i = 0; i < 6; ++i) { f_mul(&t, &beta, &S_(i), fp); f_add(&t, &t, &W_(i), fp); f_add(&t, &t, &gamma, fp); f_mul(&ft, &ft, &t, fp); }
    f_sub(&ft, &ft, &pi0, fp);
    fe t2; f_mul(&t2, &a0, &zkp, fp); f_mul(&t2, &t2, &z0, fp);
    for (int i = 0; i < 7; ++i) { f_mul(&t, &beta, &zeta, fp); f_mul(&t, &t, &shifts[i], fp); f_add(&t, &t, &gamma, fp); f_add(&t, &t, &W_(i), fp); f_mul(&t2, &t2, &t, fp); }
    f_sub(&ft, &ft, &t2, fp);
    fe wz, zmwz, zm1, num, den, omz0;
    f_pow_u64(&wz, &omega, n - (uint64_t)G.step_zk_rows, fp); f_sub(&zmwz, &zeta, &wz, fp); f_sub(&zm1, &zeta, &fp->one, fp);
    f_mul(&num, &zeta1m1, &a1, fp); f_mul(&num, &num, &zmwz, fp); f_mul(&t, &zeta1m1, &a2, fp); f_mul(&t, &t, &zm1, fp); f_add(&num, &num, &t, fp);
    f_sub(&omz0, &fp->one, &z0, fp); f_mul(&num, &num, &omz0, fp);
    f_mul(&den, &zmwz, &zm1, fp); f_inv(&den, &den, fp); f_mul(&num, &num, &den, fp); f_add(&ft, &ft, &num, fp);
    if (G.step_ct_len) {
        polish_env e; e.alpha = alpha; e.beta = beta; e.gamma = gamma; endo_of(&e.endo, fp, 0); e.zeta = zeta; e.zeta_n_minus_1 = zeta1m1; e.omega = omega; e.zkpm = zkp;
        e.mds = &G.pp[0].mds[0][0]; e.evals = (const fe (*)[2])ev; e.n_evals = (int)p->n_evals; e.log2_domain = k; e.zk_rows = G.step_zk_rows;
        fe ctv; if (polish_eval(&ctv, G.step_ct, G.step_ct_len, &e, fp)) return 0;
        f_sub(&ft, &ft, &ctv, fp);
    }
    fe perm; f_mul(&perm, &z1, &beta, fp); f_mul(&perm, &perm, &a0, fp); f_mul(&perm, &perm, &zkp, fp);
    for (int i = 0; i < 6; ++i) { f_mul(&t, &beta, &S_(i), fp); f_add(&t, &t, &gamma, fp); f_add(&t, &t, &W_(i), fp); f_mul(&perm, &perm, &t, fp); }
    f_neg(&perm, &perm, fp);
#undef W_
#undef S_
    /* combined inner product (Horner in xi over: old challenge polynomials, public input, ft, the evaluations), b */
    fe cip[2];
    for (int side = 0; side < 2; ++side) {
        const fe *pt = side ? &zetaw : &zeta; fe acc; memset(&acc, 0, sizeof acc);
        for (int j = (int)p->n_evals - 1; j >= 0; --j) { f_mul(&acc, &acc, &xi, fp); f_add(&acc, &acc, &ev[j][side], fp); }
        f_mul(&acc, &acc, &xi, fp); f_add(&acc, &acc, side ? &ft1 : &ft, fp);
        f_mul(&acc, &acc, &xi, fp); f_add(&acc, &acc, side ? &pi1 : &pi0, fp);
        for (int a = (int)p->n_old - 1; a >= 0; --a) { fe bv; b_poly_eval(&bv, old[a], 16, pt, fp); f_mul(&acc, &acc, &xi, fp); f_add(&acc, &acc, &bv, fp); }
        cip[side] = acc;
    }
    fe cipv, bv0, bv1, bb; f_mul(&cipv, &rr, &cip[1], fp); f_add(&cipv, &cipv, &cip[0], fp);
    b_poly_eval(&bv0, bp, 16, &zeta, fp); b_poly_eval(&bv1, bp, 16, &zetaw, fp); f_mul(&bb, &rr, &bv1, fp); f_add(&bb, &bb, &bv0, fp);
    f_store(r->step_cip, &cipv, fp); f_store(r->step_b, &bb, fp);
    fe zeta_srs; f_pow_u64(&zeta_srs, &zeta, (uint64_t)1 << 16, fp);
    /* message digests */
    sponge mw; sp_init(&mw, &G.pp[1], fq);
    for (int a = 0; a < 2; ++a) for (int i = 0; i < 15; ++i) { fe c; chal_to_field(&c, p->wrap_old + ((size_t)a * 15 + i) * 16, &endo_fq, fq); sp_absorb(&mw, &c); }
    sp_absorb_pt(&mw, p->wrap_sg);
    fe msg_wrap; sp_squeeze(&mw, &msg_wrap);
    sponge ms; sp_init(&ms, &G.pp[0], fp); memcpy(ms.s, G.tick_after_index, sizeof ms.s); ms.squeezed = G.tick_after_squeezed; ms.count = G.tick_after_count;
    if (!mw_canonical(p->app_state, fp)) return 0;
    fe app; f_load(&app, p->app_state, fp); sp_absorb(&ms, &app);
    for (uint32_t a = 0; a < p->n_old; ++a) { sp_absorb_pt(&ms, p->step_comms + 64 * a); for (int i = 0; i < 16; ++i) sp_absorb(&ms, &old[a][i]); }
    fe msg_step; sp_squeeze(&ms, &msg_step);
    /* packing: Fp values enter Fq as integers */
    fe half, two, c255; f_from_u64(&two, 2, fp); f_inv(&half, &two, fp); f_pow_u64(&c255, &two, 255, fp); f_add(&c255, &c255, &fp->one, fp);
    const fe *five[5] = {&cipv, &bb, &zeta_srs, &zeta_n, &perm};
    int q = 0;
    for (int i = 0; i < 5; ++i) { fe s, pl; f_sub(&s, five[i], &c255, fp); f_mul(&s, &s, &half, fp); f_from_mont(&pl, &s, fp); f_to_mont(&pubs[q++], &pl, fq); }     /* Shifted_value.Type1 */
    f_from_le128(&pubs[q++], p->plonk + 16, fq); f_from_le128(&pubs[q++], p->plonk + 32, fq);
    f_from_le128(&pubs[q++], p->plonk, fq); f_from_le128(&pubs[q++], p->plonk + 48, fq); f_from_le128(&pubs[q++], xi_c, fq);
    f_from_le256_reduce(&pubs[q++], p->sponge_digest, fq);
    pubs[q++] = msg_wrap;
    { fe pl; f_from_mont(&pl, &msg_step, fp); f_to_mont(&pubs[q++], &pl, fq); }
    for (int i = 0; i < 16; ++i) f_from_le128(&pubs[q++], p->bp + 16 * i, fq);
    static const int mask[3] = {0, 2, 3};
    f_from_u64(&pubs[q++], (uint64_t)(4 * k + mask[p->misc[1]]), fq);
    for (int i = 0; i < 8; ++i) f_from_u64(&pubs[q++], p->misc[2 + i] ? 1 : 0, fq);
    f_from_u64(&pubs[q++], p->misc[10] ? 1 : 0, fq);
    if (p->misc[10]) f_from_le128(&pubs[q++], p->misc + 16, fq); else memset(&pubs[q++], 0, sizeof(fe));
    for (int i = 0; i < 40; ++i) f_store(r->pubs + 32 * i, &pubs[i], fq);
    return q == 40;
}

/* ---- 3. kimchi oracles + to_batch, then the combined opening check of that one proof (oracle/kimchi_ref.py, oracle/ipa_ref.py) */
/* what the transcript of one opening leaves for the MSM: the (point, scalar) list is a function of it and of the folding randomisers (rho, sigma) */
typedef struct { int k, ncomms; fe chal[16], chal_inv[16], c, b0, cip, v, zz1, zz2; uint8_t U[64], comms[47 * 64]; const oc_proof *p; } opening_prep;
static int kimchi_and_opening_prepare(const oc_proof *p, oc_result *r, const fe pubs[40], opening_prep *o) {
    const fctx *fp = &F[0], *fq = &F[1];
    const int k = G.log2_domain; const uint64_t n = (uint64_t)1 << k;
    fe endo_r; endo_of(&endo_r, fq, 1);                                   /* Pallas endo_r (in Fq) */
    fe w; domain_generator(&w, k, fq);
    /* public-input commitment: h - sum pub_i L_i */
    uint8_t pc[64];
    { uint8_t sc[64 * 32], t[64]; fe neg; for (int i = 0; i < 40; ++i) { f_neg(&neg, &pubs[i], fq); f_store(sc + 32 * i, &neg, fq); }
      oracle_msm_pippenger(0, 40, G.lagrange, sc, t, 1); oracle_point_add(0, t, G.h_pallas, pc); memcpy(r->public_comm, pc, 64); }
    /* recursion challenges of the wrap proof: the statement's wrap_old_challenges expanded (2 x 15) */
    fe prev[2][15];
    for (int a = 0; a < 2; ++a) for (int i = 0; i < 15; ++i) chal_to_field(&prev[a][i], p->wrap_old + ((size_t)a * 15 + i) * 16, &endo_r, fq);
    sponge sq; sp_init(&sq, &G.pp[0], fp);
    sp_absorb(&sq, &G.index_digest);
    for (int a = 0; a < 2; ++a) sp_absorb_pt(&sq, p->prev_comms + 64 * a);
    sp_absorb_pt(&sq, pc);
    for (int i = 0; i < 15; ++i) sp_absorb_pt(&sq, p->w_comm + 64 * i);
    uint8_t c16[16]; fe beta, gamma, alpha, zeta;
    sp_challenge128(&sq, c16); f_from_le128(&beta, c16, fq); sp_challenge128(&sq, c16); f_from_le128(&gamma, c16, fq);
    sp_absorb_pt(&sq, p->z_comm); sp_challenge128(&sq, c16); chal_to_field(&alpha, c16, &endo_r, fq);
    for (int i = 0; i < 7; ++i) sp_absorb_pt(&sq, p->t_comm + 64 * i);
    sp_challenge128(&sq, c16); chal_to_field(&zeta, c16, &endo_r, fq);
    sponge fq_after = sq;
    fe dgp, digest; { sponge c = sq; sp_squeeze(&c, &dgp); fe pl; f_from_mont(&pl, &dgp, fp); f_to_mont(&digest, &pl, fq); }      /* Fp value < q: always fits */
    sponge fr, pf; sp_init(&fr, &G.pp[1], fq); sp_init(&pf, &G.pp[1], fq);
    sp_absorb(&fr, &digest);
    for (int a = 0; a < 2; ++a) for (int i = 0; i < 15; ++i) sp_absorb(&pf, &prev[a][i]);
    fe pfd; sp_squeeze(&pf, &pfd); sp_absorb(&fr, &pfd);
    fe zeta1, zetaw; f_pow_u64(&zeta1, &zeta, n, fq); f_mul(&zetaw, &zeta, &w, fq);
    /* negated public polynomial at zeta, zeta * omega */
    fe pe[2];
    for (int side = 0; side < 2; ++side) {
        const fe *x = side ? &zetaw : &zeta; fe acc, wi = fq->one; memset(&acc, 0, sizeof acc);
        for (int i = 0; i < 40; ++i) { fe d, t; f_sub(&d, x, &wi, fq); f_inv(&d, &d, fq); f_mul(&t, &d, &pubs[i], fq); f_mul(&t, &t, &wi, fq); f_sub(&acc, &acc, &t, fq); f_mul(&wi, &wi, &w, fq); }
        fe xn, nn, ninv; f_pow_u64(&xn, x, n, fq); f_sub(&xn, &xn, &fq->one, fq); f_from_u64(&nn, n, fq); f_inv(&ninv, &nn, fq);
        f_mul(&acc, &acc, &xn, fq); f_mul(&pe[side], &acc, &ninv, fq);
    }
    fe ev[43][2];
    for (int c = 0; c < 43; ++c) for (int sd = 0; sd < 2; ++sd) { if (!mw_canonical(p->evals + (c * 2 + sd) * 32, fq)) return 0; f_load(&ev[c][sd], p->evals + (c * 2 + sd) * 32, fq); }
    if (!mw_canonical(p->ft_eval1, fq)) return 0;
    fe ft1; f_load(&ft1, p->ft_eval1, fq);
    sp_absorb(&fr, &ft1); sp_absorb(&fr, &pe[0]); sp_absorb(&fr, &pe[1]);
    for (int c = 0; c < 43; ++c) { sp_absorb(&fr, &ev[c][0]); sp_absorb(&fr, &ev[c][1]); }
    fe v, u; sp_challenge128(&fr, c16); chal_to_field(&v, c16, &endo_r, fq); sp_challenge128(&fr, c16); chal_to_field(&u, c16, &endo_r, fq);
    f_store(r->v, &v, fq); f_store(r->u, &u, fq);
    /* ft_eval0 */
    fe a0, a1, a2; f_pow_u64(&a0, &alpha, (uint64_t)G.perm_alpha_offset, fq); f_mul(&a1, &a0, &alpha, fq); f_mul(&a2, &a1, &alpha, fq);
    fe zkpm = fq->one;
    for (uint64_t i = n - (uint64_t)G.zk_rows; i < n; ++i) { fe wi, d; f_pow_u64(&wi, &w, i, fq); f_sub(&d, &zeta, &wi, fq); f_mul(&zkpm, &zkpm, &d, fq); }
#define W_(i) ev[7 + (i)][0]
#define S_(i) ev[37 + (i)][0]
    const fe z0 = ev[0][0], z1 = ev[0][1];
    fe ft, t, t2; f_add(&ft, &W_(6), &gamma, fq); f_mul(&ft, &ft, &z1, fq); f_mul(&ft, &ft, &a0, fq); f_mul(&ft, &ft, &zkpm, fq);
    for (int i = 0; i < 6; ++i) { f_mul(&t, &beta, &S_(i), fq); f_add(&t, &t, &W_(i), fq); f_add(&t, &t, &gamma, fq); f_mul(&ft, &ft, &t, fq); }
    f_sub(&ft, &ft, &pe[0], fq);
    f_mul(&t2, &a0, &zkpm, fq); f_mul(&t2, &t2, &z0, fq);
    for (int i = 0; i < 7; ++i) { f_mul(&t, &beta, &zeta, fq); f_mul(&t, &t, &G.shifts[i], fq); f_add(&t, &t, &gamma, fq); f_add(&t, &t, &W_(i), fq); f_mul(&t2, &t2, &t, fq); }
    f_sub(&ft, &ft, &t2, fq);
    fe wz, zmwz, zm1, num, den, omz0, z1m1;
    f_pow_u64(&wz, &w, n - (uint64_t)G.zk_rows, fq); f_sub(&zmwz, &zeta, &wz, fq); f_sub(&zm1, &zeta, &fq->one, fq); f_sub(&z1m1, &zeta1, &fq->one, fq);
    f_mul(&num, &z1m1, &a1, fq); f_mul(&num, &num, &zmwz, fq); f_mul(&t, &z1m1, &a2, fq); f_mul(&t, &t, &zm1, fq); f_add(&num, &num, &t, fq);
    f_sub(&omz0, &fq->one, &z0, fq); f_mul(&num, &num, &omz0, fq);
    f_mul(&den, &zmwz, &zm1, fq); f_inv(&den, &den, fq); f_mul(&num, &num, &den, fq); f_add(&ft, &ft, &num, fq);
    { polish_env e; e.alpha = alpha; e.beta = beta; e.gamma = gamma; endo_of(&e.endo, fq, 0); e.zeta = zeta; e.zeta_n_minus_1 = z1m1; e.omega = w; e.zkpm = zkpm;
      e.mds = &G.pp[1].mds[0][0]; e.evals = (const fe (*)[2])ev; e.n_evals = 43; e.log2_domain = k; e.zk_rows = G.zk_rows;
      fe ctv; if (polish_eval(&ctv, G.ct, G.ct_len, &e, fq)) return 0; f_sub(&ft, &ft, &ctv, fq); }
    f_store(r->ft_eval0, &ft, fq);
    /* perm scalar; ft_comm = perm_scalar * sigma_6 - (zeta^n - 1) * sum_i zeta^(n i) t_i */
    fe ps; f_mul(&ps, &z1, &beta, fq); f_mul(&ps, &ps, &a0, fq); f_mul(&ps, &ps, &zkpm, fq);
    for (int i = 0; i < 6; ++i) { f_mul(&t, &beta, &S_(i), fq); f_add(&t, &t, &gamma, fq); f_add(&t, &t, &W_(i), fq); f_mul(&ps, &ps, &t, fq); }
    f_neg(&ps, &ps, fq);
#undef W_
#undef S_
    uint8_t ft_comm[64];
    { uint8_t pts[8 * 64], sc[8 * 32]; fe coef, zp = fq->one, negz; f_neg(&negz, &z1m1, fq);
      memcpy(pts, G.sigma_comm + 6 * 64, 64); f_store(sc, &ps, fq);
      for (int i = 0; i < 7; ++i) { memcpy(pts + 64 * (1 + i), p->t_comm + 64 * i, 64); f_mul(&coef, &negz, &zp, fq); f_store(sc + 32 * (1 + i), &coef, fq); f_mul(&zp, &zp, &zeta1, fq); }
      oracle_msm_naive(0, 8, pts, sc, ft_comm); }
    /* evaluation list: 2 recursion polynomials, public, ft, then the 43 columns */
    const int ncomms = 2 + 2 + 43;
    uint8_t comms[47 * 64]; fe evl[47][2];
    for (int a = 0; a < 2; ++a) { memcpy(comms + 64 * a, p->prev_comms + 64 * a, 64); b_poly_eval(&evl[a][0], prev[a], 15, &zeta, fq); b_poly_eval(&evl[a][1], prev[a], 15, &zetaw, fq); }
    memcpy(comms + 64 * 2, pc, 64); evl[2][0] = pe[0]; evl[2][1] = pe[1];
    memcpy(comms + 64 * 3, ft_comm, 64); evl[3][0] = ft; evl[3][1] = ft1;
    memcpy(comms + 64 * 4, p->z_comm, 64); memcpy(comms + 64 * 5, G.sel_comm, 6 * 64); memcpy(comms + 64 * 11, p->w_comm, 15 * 64);
    memcpy(comms + 64 * 26, G.coeff_comm, 15 * 64); memcpy(comms + 64 * 41, G.sigma_comm, 6 * 64);
    for (int c = 0; c < 43; ++c) { evl[4 + c][0] = ev[c][0]; evl[4 + c][1] = ev[c][1]; }
    fe cip, xi_i = fq->one; memset(&cip, 0, sizeof cip);
    for (int i = 0; i < ncomms; ++i) { fe term; f_mul(&term, &evl[i][1], &u, fq); f_add(&term, &term, &evl[i][0], fq); f_mul(&term, &term, &xi_i, fq); f_add(&cip, &cip, &term, fq); f_mul(&xi_i, &xi_i, &v, fq); }
    f_store(r->cip, &cip, fq);
    /* ---- SRS::verify of this one opening (rho = sigma = 1) */
    sponge sp = fq_after;
    { fe sh, two255, two; f_from_u64(&two, 2, fq); f_pow_u64(&two255, &two, 255, fq);                       /* shift_scalar: scalar modulus (q) > base modulus (p) */
      f_sub(&sh, &cip, &two255, fq); sp_absorb_fr(&sp, &sh, fq); }
    fe tch; sp_squeeze(&sp, &tch);
    aff U; bw_to_group(&U, &tch, fp);
    uint8_t Ub[64]; aff_store(Ub, &U, fp);
    fe chal[32], chal_inv[32];
    for (int j = 0; j < k; ++j) { sp_absorb_pt(&sp, p->lr + (size_t)j * 128); sp_absorb_pt(&sp, p->lr + (size_t)j * 128 + 64); sp_challenge128(&sp, c16); chal_to_field(&chal[j], c16, &endo_r, fq); f_inv(&chal_inv[j], &chal[j], fq); }
    sp_absorb_pt(&sp, p->delta);
    fe c; sp_challenge128(&sp, c16); chal_to_field(&c, c16, &endo_r, fq);
    fe b0, bz, bzw; b_poly_eval(&bz, chal, k, &zeta, fq); b_poly_eval(&bzw, chal, k, &zetaw, fq); f_mul(&b0, &u, &bzw, fq); f_add(&b0, &b0, &bz, fq);
    if (!mw_canonical(p->z1, fq) || !mw_canonical(p->z2, fq)) return 0;
    fe zz1, zz2; f_load(&zz1, p->z1, fq); f_load(&zz2, p->z2, fq);
    o->k = k; o->ncomms = ncomms; o->c = c; o->b0 = b0; o->cip = cip; o->v = v; o->zz1 = zz1; o->zz2 = zz2; o->p = p;
    memcpy(o->chal, chal, sizeof(fe) * (size_t)k); memcpy(o->chal_inv, chal_inv, sizeof(fe) * (size_t)k); memcpy(o->U, Ub, 64); memcpy(o->comms, comms, (size_t)ncomms * 64);
    return 1;
}
/* the 2k + ncomms + 4 per-proof entries of SRS::verify's MSM under the randomisers (rho, sigma); returns their count.
 * Not in the list: h (scalar -rho z2) and the SRS part g[j] (scalar sigma s[j]): the caller folds those over the batch. */
static size_t opening_small_list(const opening_prep *o, const fe *rho, const fe *sigma, uint8_t *pts, uint8_t *scs) {
    const fctx *fq = &F[1]; const oc_proof *p = o->p; const int k = o->k;
    size_t q = 0; fe s, rc, xi_i;
#define PUT(ptr, sc) do { memcpy(pts + 64 * q, (ptr), 64); f_store(scs + 32 * q, (sc), fq); ++q; } while (0)
    f_mul(&s, rho, &o->zz1, fq); f_neg(&s, &s, fq); f_sub(&s, &s, sigma, fq); PUT(p->sg, &s);                 /* sg: -rho z1 - sigma */
    f_mul(&s, rho, &o->zz1, fq); f_neg(&s, &s, fq); f_mul(&s, &s, &o->b0, fq); PUT(o->U, &s);                  /* U: -rho z1 b0 */
    f_mul(&rc, rho, &o->c, fq);
    for (int j = 0; j < k; ++j) { f_mul(&s, &rc, &o->chal_inv[j], fq); PUT(p->lr + (size_t)j * 128, &s); f_mul(&s, &rc, &o->chal[j], fq); PUT(p->lr + (size_t)j * 128 + 64, &s); }
    xi_i = fq->one;
    for (int i = 0; i < o->ncomms; ++i) { f_mul(&s, &rc, &xi_i, fq); PUT(o->comms + 64 * i, &s); f_mul(&xi_i, &xi_i, &o->v, fq); }
    f_mul(&s, &rc, &o->cip, fq); PUT(o->U, &s);
    PUT(p->delta, rho);
#undef PUT
    return q;
}
static int points_on_curve_or_zero(int curve, const uint8_t *pts, size_t q) {
    for (size_t i = 0; i < q; ++i) if (!oracle_is_on_curve(curve, pts + 64 * i)) { int z = 1; for (int b = 0; b < 64; ++b) if (pts[64 * i + b]) z = 0; if (!z) return 0; }
    return 1;
}
static void bpoly_coeffs_mont(fe *s, const fe *c, int k, const fctx *f) {      /* oracle_b_poly_coefficients, Montgomery in and out */
    const size_t n = (size_t)1 << k; s[0] = f->one; int kk = 0; size_t pw = 1;
    for (size_t i = 1; i < n; ++i) { if (i == (pw << 1)) { ++kk; pw <<= 1; } f_mul(&s[i], &s[i - pw], &c[k - 1 - kk], f); }
}
static int kimchi_and_opening(const oc_proof *p, oc_result *r, const fe pubs[40]) {
    const fctx *fq = &F[1];
    opening_prep o;
    if (!kimchi_and_opening_prepare(p, r, pubs, &o)) return 0;
    const int k = o.k; const uint64_t n = (uint64_t)1 << k; const int ncomms = o.ncomms;
    const fe *chal = o.chal; const fe zz2 = o.zz2;
    const size_t np = 1 + n + (size_t)(2 * k + ncomms + 4);
    uint8_t *pts = (uint8_t *)malloc(np * 64), *scs = (uint8_t *)malloc(np * 32);
    memcpy(pts, G.h_pallas, 64); { fe s; f_neg(&s, &zz2, fq); f_store(scs, &s, fq); }
    memcpy(pts + 64, G.g_pallas, n * 64);
    { uint8_t chb[32 * 32]; for (int j = 0; j < k; ++j) f_store(chb + 32 * j, &chal[j], fq); oracle_b_poly_coefficients(1, k, chb, scs + 32); }      /* sigma = 1 */
    size_t q = 1 + n;
    q += opening_small_list(&o, &fq->one, &fq->one, pts + 64 * q, scs + 32 * q);                         /* rho = sigma = 1 */
    uint8_t out[64]; const int bad = !points_on_curve_or_zero(0, pts, q);
    oracle_msm_pippenger(0, q, pts, scs, out, 1);
    free(pts); free(scs);
    if (bad) return 0;
    for (int i = 0; i < 64; ++i) if (out[i]) return 0;
    return 1;
}

/* ---- 4. accumulator check: MSM(vesta.g, b_poly_coefficients(to_field(prechallenges))) == sg */
static int accumulator_ok(const oc_proof *p) {
    const fctx *fp = &F[0];
    fe endo; endo_of(&endo, fp, 1);
    uint8_t chb[16 * 32];
    for (int i = 0; i < 16; ++i) { fe c; chal_to_field(&c, p->acc_pre + 16 * i, &endo, fp); f_store(chb + 32 * i, &c, fp); }
    uint8_t *s = (uint8_t *)malloc(((size_t)1 << 16) * 32), out[64];
    oracle_b_poly_coefficients(0, 16, chb, s);
    oracle_msm_pippenger(1, (size_t)1 << 16, G.g_vesta, s, out, 1);
    free(s);
    return memcmp(out, p->acc_sg, 64) == 0;
}

int oc_verify_one(const oc_proof *p, oc_result *r) {
    memset(r, 0, sizeof *r);
    r->chain_ok = state_hashes(p, r);
    fe pubs[40];
    r->statement_ok = pickles_public_input(p, r, pubs);
    r->ipa_ok = r->statement_ok ? kimchi_and_opening(p, r, pubs) : 0;
    r->acc_ok = accumulator_ok(p);
    r->verdict = r->chain_ok && r->statement_ok && r->ipa_ok && r->acc_ok;
    return r->verdict;
}

/* n proofs over `threads` pthreads, proof i on thread i % threads; verdicts[i] = 0 / 1 */
typedef struct { const oc_proof *p; size_t n; int t, nt; uint8_t *v; } oc_job;
static void *oc_worker(void *a) { oc_job *j = (oc_job *)a; oc_result r; for (size_t i = (size_t)j->t; i < j->n; i += (size_t)j->nt) j->v[i] = (uint8_t)oc_verify_one(&j->p[i], &r); return NULL; }
int oc_verify_many(const oc_proof *proofs, size_t n, int threads, uint8_t *verdicts) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = (int)n;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads); oc_job *jobs = (oc_job *)malloc(sizeof(oc_job) * (size_t)threads);
    { oc_result r; if (n) state_hashes(&proofs[0], &r); }           /* warm the function-static salts before the threads start */
    for (int t = 0; t < threads; ++t) { jobs[t] = (oc_job){proofs, n, t, threads, verdicts}; pthread_create(&th[t], NULL, oc_worker, &jobs[t]); }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
    return 0;
}
/* ---------------------------------------------------------------------------------------------- the batch FOLDED as the GPU job folds it
 * kimchi `batch_verify` / poly-commitment `SRS::verify` on n proofs at once: the per-proof transcripts run on `threads`
 * pthreads (a proof per thread at a time), then ONE MSM per curve for the whole batch --
 *   Pallas:  sum_b [ sigma_b s_b(G) - rho_b z2_b H + (per-proof entries under rho_b, sigma_b) ]  == 0,   rho_b = r^b, sigma_b = t^b
 *   Vesta:   sum_b rho'_b [ s'_b(G) - sg_b ]                                                      == 0    (the step accumulators, openmina accumulator_check)
 * with r, t, rho' drawn by the caller (`rand32`: 3 x 32 bytes, e.g. from the OS CSPRNG).  The folded check answers for the whole batch; per-proof
 * verdicts = the per-proof legs (chain, statement, well-formedness) AND the fold -- a batch with a bad opening comes back all-zero here (the product's
 * culprit search has no counterpart in this baseline: it is timed on accepting batches).  cpu_baseline_folded of bench.py: the like-for-like algorithm. */
typedef struct { const oc_proof *p; size_t n; int t, nt; uint8_t *ok; opening_prep *prep; fe *Sg, *Tg; fe *h_acc; const fe *rho, *sigma, *rho_acc; uint8_t *pts, *scs; size_t per; } fold_job;
static void *fold_worker(void *a) {
    fold_job *j = (fold_job *)a; const fctx *fp = &F[0], *fq = &F[1];
    const int k = G.log2_domain; const size_t n = (size_t)1 << k, nacc = (size_t)1 << 16;
    fe *s = (fe *)malloc(sizeof(fe) * nacc);
    fe endo_p; endo_of(&endo_p, fp, 1);
    memset(j->Sg, 0, sizeof(fe) * n); memset(j->Tg, 0, sizeof(fe) * nacc); memset(j->h_acc, 0, sizeof(fe));
    for (size_t b = (size_t)j->t; b < j->n; b += (size_t)j->nt) {
        oc_result r; fe pubs[40]; memset(&r, 0, sizeof r);
        const oc_proof *p = &j->p[b];
        int ok = state_hashes(p, &r);
        ok = pickles_public_input(p, &r, pubs) && ok;
        opening_prep *o = &j->prep[b];
        const int have = kimchi_and_opening_prepare(p, &r, pubs, o);
        uint8_t *pts = j->pts + b * j->per * 64, *scs = j->scs + b * j->per * 32;
        memset(pts, 0, j->per * 64); memset(scs, 0, j->per * 32);
        if (have) {
            const size_t q = opening_small_list(o, &j->rho[b], &j->sigma[b], pts, scs);
            if (!points_on_curve_or_zero(0, pts, q)) { ok = 0; memset(scs, 0, j->per * 32); }
            else {
                bpoly_coeffs_mont(s, o->chal, k, fq);
                for (size_t i = 0; i < n; ++i) { fe t; f_mul(&t, &s[i], &j->sigma[b], fq); f_add(&j->Sg[i], &j->Sg[i], &t, fq); }
                fe t; f_mul(&t, &j->rho[b], &o->zz2, fq); f_sub(j->h_acc, j->h_acc, &t, fq);
            }
        } else ok = 0;
        /* the step accumulator: rho'_b (s'_b(G) - sg_b); its point goes behind the opening's entries */
        { fe c[16]; for (int i = 0; i < 16; ++i) chal_to_field(&c[i], p->acc_pre + 16 * i, &endo_p, fp);
          bpoly_coeffs_mont(s, c, 16, fp);
          for (size_t i = 0; i < nacc; ++i) { fe t; f_mul(&t, &s[i], &j->rho_acc[b], fp); f_add(&j->Tg[i], &j->Tg[i], &t, fp); }
          if (!oracle_is_on_curve(1, p->acc_sg)) ok = 0; }
        j->ok[b] = (uint8_t)ok;
    }
    free(s);
    return NULL;
}
/* phase 1 + the reductions over threads: what the fixed-base MSMs of the batch would consume.  Sg / Tg: Montgomery, n / 2^16 entries; small lists per proof. */
typedef struct { fe *Sg, *Tg, h_acc, *rho_acc; uint8_t *ok, *pts, *scs; size_t per, n; opening_prep *prep; fe *rho, *sigma; } fold_state;
static void fold_free(fold_state *st) { free(st->Sg); free(st->Tg); free(st->rho_acc); free(st->ok); free(st->pts); free(st->scs); free(st->prep); free(st->rho); free(st->sigma); }
static void fold_collect(const oc_proof *proofs, size_t nproofs, int threads, const uint8_t *rand32, int pow_first, fold_state *st) {
    const fctx *fp = &F[0], *fq = &F[1];
    if (threads < 1) threads = 1;
    if ((size_t)threads > nproofs) threads = (int)nproofs;
    const int k = G.log2_domain; const size_t n = (size_t)1 << k, nacc = (size_t)1 << 16, per = (size_t)(2 * k + 47 + 4);
    { oc_result r; state_hashes(&proofs[0], &r); }                   /* warm the function-static salts before the threads start */
    /* rho_b = r^(b + pow_first), sigma_b = t^(b + pow_first), rho'_b = u^(b + pow_first).  pow_first = 0: upstream's shape (proof 0 carries coefficient 1);
     * pow_first = 1: one SHARD of the exchange variant -- partial sums of several shards are added, so no proof of any shard may carry a fixed coefficient */
    fe base[3], *rho = (fe *)malloc(sizeof(fe) * nproofs), *sigma = (fe *)malloc(sizeof(fe) * nproofs), *rho_acc = (fe *)malloc(sizeof(fe) * nproofs);
    f_from_le256_reduce(&base[0], rand32, fq); f_from_le256_reduce(&base[1], rand32 + 32, fq); f_from_le256_reduce(&base[2], rand32 + 64, fp);
    if (pow_first) { rho[0] = base[0]; sigma[0] = base[1]; rho_acc[0] = base[2]; } else { rho[0] = fq->one; sigma[0] = fq->one; rho_acc[0] = fp->one; }
    for (size_t b = 1; b < nproofs; ++b) { f_mul(&rho[b], &rho[b - 1], &base[0], fq); f_mul(&sigma[b], &sigma[b - 1], &base[1], fq); f_mul(&rho_acc[b], &rho_acc[b - 1], &base[2], fp); }
    opening_prep *prep = (opening_prep *)malloc(sizeof(opening_prep) * nproofs);
    uint8_t *ok = (uint8_t *)calloc(nproofs, 1), *pts = (uint8_t *)malloc(nproofs * per * 64), *scs = (uint8_t *)malloc(nproofs * per * 32);
    fe *Sg = (fe *)malloc(sizeof(fe) * n * (size_t)threads), *Tg = (fe *)malloc(sizeof(fe) * nacc * (size_t)threads), *h_acc = (fe *)malloc(sizeof(fe) * (size_t)threads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads); fold_job *jobs = (fold_job *)malloc(sizeof(fold_job) * (size_t)threads);
    for (int t = 0; t < threads; ++t) {
        jobs[t] = (fold_job){proofs, nproofs, t, threads, ok, prep, Sg + n * (size_t)t, Tg + nacc * (size_t)t, h_acc + t, rho, sigma, rho_acc, pts, scs, per};
        pthread_create(&th[t], NULL, fold_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    for (int t = 1; t < threads; ++t) {
        for (size_t i = 0; i < n; ++i) f_add(&Sg[i], &Sg[i], &Sg[n * (size_t)t + i], fq);
        for (size_t i = 0; i < nacc; ++i) f_add(&Tg[i], &Tg[i], &Tg[nacc * (size_t)t + i], fp);
        f_add(&h_acc[0], &h_acc[0], &h_acc[t], fq);
    }
    st->Sg = Sg; st->Tg = Tg; st->h_acc = h_acc[0]; st->rho_acc = rho_acc; st->ok = ok; st->pts = pts; st->scs = scs; st->per = per; st->n = n; st->prep = prep; st->rho = rho; st->sigma = sigma;
    free(h_acc); free(th); free(jobs);
}
int oc_verify_folded(const oc_proof *proofs, size_t nproofs, int threads, const uint8_t *rand32 /* 3 x 32 */, uint8_t *verdicts) {
    const fctx *fp = &F[0], *fq = &F[1];
    if (nproofs == 0) return 0;
    fold_state st; fold_collect(proofs, nproofs, threads, rand32, 0, &st);
    const size_t n = st.n, nacc = (size_t)1 << 16, per = st.per;
    /* Pallas: g[0..n) with the folded scalars, h, then every proof's entries (zero scalars where a proof was malformed) */
    const size_t qp = n + 1 + nproofs * per;
    uint8_t *P = (uint8_t *)malloc(qp * 64), *S = (uint8_t *)malloc(qp * 32), out[64];
    memcpy(P, G.g_pallas, n * 64); for (size_t i = 0; i < n; ++i) f_store(S + 32 * i, &st.Sg[i], fq);
    memcpy(P + n * 64, G.h_pallas, 64); f_store(S + 32 * n, &st.h_acc, fq);
    memcpy(P + (n + 1) * 64, st.pts, nproofs * per * 64); memcpy(S + (n + 1) * 32, st.scs, nproofs * per * 32);
    oracle_msm_pippenger(0, qp, P, S, out, threads);
    int ipa_ok = 1; for (int i = 0; i < 64; ++i) if (out[i]) ipa_ok = 0;
    free(P); free(S);
    /* Vesta: g[0..2^16) with the folded scalars, then -rho'_b sg_b */
    const size_t qv = nacc + nproofs;
    P = (uint8_t *)malloc(qv * 64); S = (uint8_t *)malloc(qv * 32);
    memcpy(P, G.g_vesta, nacc * 64); for (size_t i = 0; i < nacc; ++i) f_store(S + 32 * i, &st.Tg[i], fp);
    for (size_t b = 0; b < nproofs; ++b) { fe t; f_neg(&t, &st.rho_acc[b], fp); memcpy(P + (nacc + b) * 64, proofs[b].acc_sg, 64); f_store(S + 32 * (nacc + b), &t, fp); }
    oracle_msm_pippenger(1, qv, P, S, out, threads);
    int acc_ok = 1; for (int i = 0; i < 64; ++i) if (out[i]) acc_ok = 0;
    free(P); free(S);
    for (size_t b = 0; b < nproofs; ++b) verdicts[b] = (uint8_t)(st.ok[b] && ipa_ok && acc_ok);
    fold_free(&st);
    return ipa_ok && acc_ok;
}
/* The same fold with the fixed-base MSMs LEFT OUT: what one shard of the multi-GPU exchange variant (mina_state_job_fold_dev on the GPU) hands to
 * the exchange step -- the folded scalar vectors (canonical bytes) and the variable-base partial sums (affine bytes; infinity = 64 zero bytes):
 *   ipa_point = -rho-weighted h term + the per-proof entries  (to be ADDED to sum_j ipa_scalars[j] G_j: the total must be infinity)
 *   acc_point = sum_b rho'_b sg_b                               (must EQUAL sum_j acc_scalars[j] G_j)
 * ok[b] = the per-proof checks.  The CPU double of tests/test_sharded_gloo.py's whole-job test. */
int oc_fold_export(const oc_proof *proofs, size_t nproofs, int threads, const uint8_t *rand32, uint8_t *ipa_scalars, uint8_t *ipa_point, uint8_t *acc_scalars, uint8_t *acc_point, uint8_t *ok) {
    const fctx *fp = &F[0], *fq = &F[1];
    if (nproofs == 0) return 0;
    fold_state st; fold_collect(proofs, nproofs, threads, rand32, 1, &st);
    const size_t n = st.n, nacc = (size_t)1 << 16, per = st.per;
    for (size_t i = 0; i < n; ++i) f_store(ipa_scalars + 32 * i, &st.Sg[i], fq);
    for (size_t i = 0; i < nacc; ++i) f_store(acc_scalars + 32 * i, &st.Tg[i], fp);
    const size_t qp = 1 + nproofs * per;
    uint8_t *P = (uint8_t *)malloc(qp * 64), *S = (uint8_t *)malloc(qp * 32);
    memcpy(P, G.h_pallas, 64); f_store(S, &st.h_acc, fq);
    memcpy(P + 64, st.pts, nproofs * per * 64); memcpy(S + 32, st.scs, nproofs * per * 32);
    oracle_msm_pippenger(0, qp, P, S, ipa_point, threads);
    free(P); free(S);
    P = (uint8_t *)malloc(nproofs * 64); S = (uint8_t *)malloc(nproofs * 32);
    for (size_t b = 0; b < nproofs; ++b) { memcpy(P + b * 64, proofs[b].acc_sg, 64); f_store(S + 32 * b, &st.rho_acc[b], fp); }
    oracle_msm_pippenger(1, nproofs, P, S, acc_point, threads);
    free(P); free(S);
    memcpy(ok, st.ok, nproofs);
    fold_free(&st);
    return 0;
}
size_t oc_sizeof_proof(void) { return sizeof(oc_proof); }
size_t oc_sizeof_result(void) { return sizeof(oc_result); }
