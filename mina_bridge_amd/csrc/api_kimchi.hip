// api_kimchi.hip -- kimchi `verifier::{oracles, to_batch}` on the GPU (transcript order README.md:413-475).
//
// Replaces kimchi `ProverProof::oracles` + `to_batch` (pin core/Cargo.toml:14) for proofs over Pallas (the Pickles wrap proof):
// one lane group (4 or 8 lanes) per proof runs
//   * the Fq-sponge: index digest, recursion commitments, public-input commitment, w -> beta, gamma -> z -> alpha -> t -> zeta
//   * the Fr-sponge: digest of the Fq-sponge, digest of the recursion challenges, ft_eval1, public evaluations, the 43 x 2 column
//     evaluations -> v, u
//   * the scalar-field work: negated public polynomial at zeta / zeta*omega, ft_eval0 (permutation part, boundary part, the
//     linearization's constant term through a PolishToken interpreter), perm scalar, b_poly of the recursion challenges, the
//     combined inner product
//   * the chunked ft commitment  perm_scalar * sigma_6 - (zeta^n - 1) * sum_i zeta^(n i) t_i   (8 scalar multiplications over the lanes)
// and emits one `BatchEvaluationProof` row per proof in the layout `mb_ipa_batch_check_dev` consumes.
// The verifier index (domain, shifts, commitments, token program) is DATA installed by the caller (`mina_verifier_index`): the
// reference tree does not hold the blockchain-snark index.  [UPSTREAM-RECALL] throughout; checked against oracle/kimchi_ref.py,
// whose miniature prover mints proofs this code must accept and whose tampered variants it must reject.
#include "ctx.h"
#include "msm.cuh"
#include "sponge.cuh"
#include "wire_proof.h"
#include "polish.h"

namespace mb {

static constexpr uint32_t KC_COLS = 43, KC_W0 = 7, KC_C0 = 22, KC_S0 = 37, KC_MAX_ZK = 8;
struct KimchiIndexDev {
    uint32_t log2_domain, zk_rows, perm_alpha_offset, n_tokens;
    fe_t shifts[7];                                    // scalar field, Montgomery
    fe_t omega, omega_zk, n_inv, zk_roots[KC_MAX_ZK];  // w, w^(n - zk_rows), 1/n, w^(n - zk_rows + i)
    fe_t mds[9], endo_coeff;                           // scalar-field Poseidon MDS (Constants.mds), index.endo
    fe_t digest;                                       // base field, Montgomery
    affine_t sigma6;                                   // base field, Montgomery
    uint32_t col_comm_words[(6 + 15 + 6) * 16];        // selectors, coefficients, sigma[0..6): canonical words, copied into every comms row
};
struct KimchiIn { const uint32_t *pub, *prev_chals, *prev_comms, *w_comm, *z_comm, *t_comm, *evals, *ft_eval1, *pubcomm; };
struct KimchiOut { uint32_t *sponge_state, *sponge_pos, *cip, *evalpoints, *polyscale, *evalscale, *comms, *ft_eval0; };

template <int F> __device__ __forceinline__ fe_t ld_checked(const uint32_t *p, const FieldK &k, bool &ok) {
    const fe_t w = load_fe<F>(p); ok = ok && fe_words_canonical<F>(w); return fe_to_mont<F>(w, k.r2);
}
template <int F> __device__ __forceinline__ fe_t fe_pow2k(fe_t a, uint32_t k) { for (uint32_t i = 0; i < k; ++i) a = fe_sqr<F>(a); return a; }
template <int F> __device__ fe_t fe_pow_u64(const fe_t &a, uint64_t e, const fe_t &one) {
    fe_t r = one, b = a;
    for (; e; e >>= 1) { if (e & 1) r = fe_mul<F>(r, b); b = fe_sqr<F>(b); }
    return r;
}
// 128-bit squeeze as a plain scalar-field element (kimchi `fq_sponge.challenge()`: beta, gamma)
template <int FS> __device__ __forceinline__ fe_t chal128_plain(const fe_t &sq_plain, const FieldK &ks) {
    fe_t o = fe_zero(); o.v[0] = sq_plain.v[0]; o.v[1] = sq_plain.v[1]; o.v[2] = sq_plain.v[2]; o.v[3] = sq_plain.v[3];
    return fe_to_mont<FS>(o, ks.r2);
}
template <int FS> __device__ __forceinline__ fe_t chal_endo(const fe_t &sq_plain, const FieldK &ks) {
    const uint64_t lo = (uint64_t)sq_plain.v[0] | ((uint64_t)sq_plain.v[1] << 32), hi = (uint64_t)sq_plain.v[2] | ((uint64_t)sq_plain.v[3] << 32);
    return challenge_to_field<FS>(lo, hi, ks);
}
__device__ __forceinline__ xyzz_t xyzz_shfl_xor(const xyzz_t &a, int mask) {
    xyzz_t r;
    for (int i = 0; i < 8; ++i) { r.x.v[i] = (uint32_t)__shfl_xor((int)a.x.v[i], mask, 64); r.y.v[i] = (uint32_t)__shfl_xor((int)a.y.v[i], mask, 64);
                                  r.zz.v[i] = (uint32_t)__shfl_xor((int)a.zz.v[i], mask, 64); r.zzz.v[i] = (uint32_t)__shfl_xor((int)a.zzz.v[i], mask, 64); }
    return r;
}
// s * P by double-and-add (s plain scalar words, P affine Montgomery)
template <int FB> __device__ xyzz_t scalar_mul_affine(const fe_t &s_plain, const affine_t &P, const FieldK &kb) {
    xyzz_t acc = xyzz_inf();
    if (aff_is_inf(P)) return acc;
    for (int bit = 254; bit >= 0; --bit) {
        acc = xyzz_dbl<FB>(acc);
        if ((s_plain.v[bit >> 5] >> (bit & 31)) & 1u) xyzz_add_affine<FB>(acc, P.x, P.y, kb.one);
    }
    return acc;
}

template <int LANES>
__global__ void __launch_bounds__(64)
kimchi_to_batch_kernel(uint32_t batch, uint32_t n_prev, uint32_t npub, FieldK kb, FieldK ks, const PoseidonParams *__restrict__ pp_b,
                       const PoseidonParams *__restrict__ pp_s, const KimchiIndexDev *__restrict__ ix, const KimchiToken *__restrict__ toks,
                       const fe_t *__restrict__ lits, KimchiIn in, KimchiOut out, uint32_t *__restrict__ bad_input) {
    constexpr int FB = FIELD_FP, FS = FIELD_FQ;                 // Pallas: base Fp, scalar Fq
    const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) / LANES, ln = threadIdx.x & (LANES - 1);
    if (b >= batch) return;
    const uint32_t k = ix->log2_domain, ncomms = n_prev + 2 + KC_COLS;
    bool ok = true;
    const uint32_t *ev = in.evals + (size_t)b * KC_COLS * 16;           // [col][zeta | zeta_omega][8]
    uint32_t *comms_out = out.comms + (size_t)b * ncomms * 16;
    auto copy_pt = [&](uint32_t *dst, const uint32_t *src) { if (ln == 0) for (int i = 0; i < 16; ++i) dst[i] = src[i]; };

    // ---- Fq-sponge
    DevSponge<FB, LANES> fq; fq.pp = pp_b; fq.squeezed = 0; fq.count = 0; fq.s = fe_zero();
    fq.absorb(ix->digest);
    auto absorb_pt = [&](const uint32_t *p) { const affine_t P = load_point_checked<FB>(p, kb, ok); fq.absorb(P.x); fq.absorb(P.y); };
    for (uint32_t i = 0; i < n_prev; ++i) { const uint32_t *p = in.prev_comms + ((size_t)b * n_prev + i) * 16; absorb_pt(p); copy_pt(comms_out + i * 16, p); }
    absorb_pt(in.pubcomm + (size_t)b * 16); copy_pt(comms_out + n_prev * 16, in.pubcomm + (size_t)b * 16);
    for (uint32_t i = 0; i < 15; ++i) absorb_pt(in.w_comm + ((size_t)b * 15 + i) * 16);
    const fe_t beta = chal128_plain<FS>(fe_from_mont<FB>(fq.squeeze()), ks);
    const fe_t gamma = chal128_plain<FS>(fe_from_mont<FB>(fq.squeeze()), ks);
    absorb_pt(in.z_comm + (size_t)b * 16);
    const fe_t alpha = chal_endo<FS>(fe_from_mont<FB>(fq.squeeze()), ks);
    for (uint32_t i = 0; i < 7; ++i) absorb_pt(in.t_comm + ((size_t)b * 7 + i) * 16);
    const fe_t zeta = chal_endo<FS>(fe_from_mont<FB>(fq.squeeze()), ks);
    {   // the sponge handed to the opening check
        const bool owner = LANES == 8 ? (ln < 6 && !(ln & 1u)) : (ln < 3);
        if (owner) { const fe_t w = fe_from_mont<FB>(fq.s); for (int i = 0; i < 8; ++i) out.sponge_state[(size_t)b * 24 + coop_elem<LANES>() * 8 + i] = w.v[i]; }
        if (ln == 0) { out.sponge_pos[2 * b] = (uint32_t)fq.squeezed; out.sponge_pos[2 * b + 1] = (uint32_t)fq.count; }
    }
    fe_t digest;
    { DevSponge<FB, LANES> cl = fq; digest = fe_to_mont<FS>(fe_from_mont<FB>(cl.squeeze()), ks.r2); }     // p < q: always fits

    // ---- Fr-sponge (Poseidon over the scalar field)
    DevSponge<FS, LANES> fr; fr.pp = pp_s; fr.squeezed = 0; fr.count = 0; fr.s = fe_zero();
    fr.absorb(digest);
    {
        DevSponge<FS, LANES> pf; pf.pp = pp_s; pf.squeezed = 0; pf.count = 0; pf.s = fe_zero();
        for (uint32_t i = 0; i < n_prev * k; ++i) pf.absorb(ld_checked<FS>(in.prev_chals + ((size_t)b * n_prev * k + i) * 8, ks, ok));
        fr.absorb(pf.squeeze());
    }
    const fe_t zeta1 = fe_pow2k<FS>(zeta, k), zetaw = fe_mul<FS>(zeta, ix->omega);
    const fe_t zetaw1 = fe_pow2k<FS>(zetaw, k);
    // negated public polynomial at zeta and zeta*omega:  -(x^n - 1)/n * sum_i p_i w^i / (x - w^i); denominators inverted 8 at a time
    fe_t pub_e[2];
    for (int side = 0; side < 2; ++side) {
        const fe_t x = side ? zetaw : zeta;
        fe_t acc = fe_zero(), wi = ks.one;
        for (uint32_t base = 0; base < npub; base += 8) {
            fe_t den[8], pre[8], wpow[8], run = ks.one;
            const uint32_t cnt = npub - base < 8 ? npub - base : 8;
            for (uint32_t j = 0; j < cnt; ++j) { wpow[j] = wi; den[j] = fe_sub<FS>(x, wi); pre[j] = run; run = fe_mul<FS>(run, den[j]); wi = fe_mul<FS>(wi, ix->omega); }
            fe_t inv = fe_inv<FS>(run, ks);
            for (int j = (int)cnt - 1; j >= 0; --j) {
                const fe_t dinv = fe_mul<FS>(inv, pre[j]); inv = fe_mul<FS>(inv, den[j]);
                const fe_t p = ld_checked<FS>(in.pub + ((size_t)b * npub + base + j) * 8, ks, ok);
                acc = fe_sub<FS>(acc, fe_mul<FS>(fe_mul<FS>(dinv, p), wpow[j]));
            }
        }
        pub_e[side] = fe_mul<FS>(fe_mul<FS>(acc, fe_sub<FS>(side ? zetaw1 : zeta1, ks.one)), ix->n_inv);
    }
    fr.absorb(ld_checked<FS>(in.ft_eval1 + (size_t)b * 8, ks, ok));
    fr.absorb(pub_e[0]); fr.absorb(pub_e[1]);
    for (uint32_t c = 0; c < KC_COLS * 2; ++c) fr.absorb(ld_checked<FS>(ev + (size_t)c * 8, ks, ok));
    const fe_t v = chal_endo<FS>(fe_from_mont<FS>(fr.squeeze()), ks);
    const fe_t u = chal_endo<FS>(fe_from_mont<FS>(fr.squeeze()), ks);

    // ---- scalar-field work (every lane redundantly; lane 0 writes)
    auto EV = [&](uint32_t col, uint32_t row) { return fe_to_mont<FS>(load_fe<FS>(ev + ((size_t)col * 2 + row) * 8), ks.r2); };
    fe_t a0 = fe_pow_u64<FS>(alpha, ix->perm_alpha_offset, ks.one);
    const fe_t a1 = fe_mul<FS>(a0, alpha), a2 = fe_mul<FS>(a1, alpha);
    fe_t zkpm = ks.one;
    for (uint32_t i = 0; i < ix->zk_rows; ++i) zkpm = fe_mul<FS>(zkpm, fe_sub<FS>(zeta, ix->zk_roots[i]));
    const fe_t z0 = EV(0, 0), z1 = EV(0, 1);
    fe_t prod6 = ks.one;                                               // prod_{i<6} (gamma + beta s_i + w_i)
    for (uint32_t i = 0; i < 6; ++i) prod6 = fe_mul<FS>(prod6, fe_add<FS>(fe_add<FS>(fe_mul<FS>(beta, EV(KC_S0 + i, 0)), EV(KC_W0 + i, 0)), gamma));
    const fe_t common = fe_mul<FS>(fe_mul<FS>(a0, zkpm), prod6);
    fe_t ft = fe_mul<FS>(fe_mul<FS>(fe_add<FS>(EV(KC_W0 + 6, 0), gamma), z1), common);
    ft = fe_sub<FS>(ft, pub_e[0]);
    {
        fe_t t2 = fe_mul<FS>(fe_mul<FS>(a0, zkpm), z0);
        const fe_t bz = fe_mul<FS>(beta, zeta);
        for (uint32_t i = 0; i < 7; ++i) t2 = fe_mul<FS>(t2, fe_add<FS>(fe_add<FS>(gamma, fe_mul<FS>(bz, ix->shifts[i])), EV(KC_W0 + i, 0)));
        ft = fe_sub<FS>(ft, t2);
        const fe_t zm1 = fe_sub<FS>(zeta1, ks.one), dw = fe_sub<FS>(zeta, ix->omega_zk), d1 = fe_sub<FS>(zeta, ks.one);
        const fe_t num = fe_mul<FS>(fe_add<FS>(fe_mul<FS>(fe_mul<FS>(zm1, a1), dw), fe_mul<FS>(fe_mul<FS>(zm1, a2), d1)), fe_sub<FS>(ks.one, z0));
        ft = fe_add<FS>(ft, fe_mul<FS>(num, fe_inv<FS>(fe_mul<FS>(dw, d1), ks)));
    }
    {   // linearization constant term: PolishToken stack machine
        fe_t stack[KC_STACK], cache[KC_CACHE]; int sp = 0, nc = 0; bool prog_ok = true;
        for (uint32_t t = 0; t < ix->n_tokens; ++t) {
            const KimchiToken tk = toks[t];
            switch (tk.op) {
                case MINA_TOK_ALPHA: stack[sp++] = alpha; break;
                case MINA_TOK_BETA: stack[sp++] = beta; break;
                case MINA_TOK_GAMMA: stack[sp++] = gamma; break;
                case MINA_TOK_JOINT_COMBINER: stack[sp++] = fe_zero(); break;
                case MINA_TOK_ENDO_COEFFICIENT: stack[sp++] = ix->endo_coeff; break;
                case MINA_TOK_MDS: stack[sp++] = ix->mds[tk.a * 3 + tk.b]; break;
                case MINA_TOK_LITERAL: stack[sp++] = lits[tk.a]; break;
                case MINA_TOK_CELL: stack[sp++] = EV(tk.a, tk.b); break;
                case MINA_TOK_DUP: stack[sp] = stack[sp - 1]; ++sp; break;
                case MINA_TOK_POW: stack[sp - 1] = fe_pow_u64<FS>(stack[sp - 1], (uint64_t)tk.a | ((uint64_t)tk.b << 32), ks.one); break;
                case MINA_TOK_ADD: stack[sp - 2] = fe_add<FS>(stack[sp - 2], stack[sp - 1]); --sp; break;
                case MINA_TOK_MUL: stack[sp - 2] = fe_mul<FS>(stack[sp - 2], stack[sp - 1]); --sp; break;
                case MINA_TOK_SUB: stack[sp - 2] = fe_sub<FS>(stack[sp - 2], stack[sp - 1]); --sp; break;
                case MINA_TOK_VANISHES_ON_ZK_ROWS: stack[sp++] = zkpm; break;
                case MINA_TOK_UNNORMALIZED_LAGRANGE: {
                    const int32_t off = (int32_t)tk.a;
                    const uint32_t row = off >= 0 ? (uint32_t)off : (1u << k) - ix->zk_rows - (uint32_t)(-off);
                    const fe_t wr = fe_pow_u64<FS>(ix->omega, row, ks.one);
                    stack[sp++] = fe_mul<FS>(fe_sub<FS>(zeta1, ks.one), fe_inv<FS>(fe_sub<FS>(zeta, wr), ks)); break; }
                case MINA_TOK_STORE: cache[nc++] = stack[sp - 1]; break;
                case MINA_TOK_LOAD: stack[sp++] = cache[tk.a]; break;
                default: prog_ok = false;
            }
        }
        if (ix->n_tokens) { if (sp != 1 || !prog_ok) ok = false; else ft = fe_sub<FS>(ft, stack[0]); }   // the host validated stack depth: defensive
    }
    const fe_t perm_scalar = fe_neg<FS>(fe_mul<FS>(fe_mul<FS>(z1, beta), common));
    // combined inner product over the evaluation list: recursion, public, ft, then the 43 columns
    fe_t cip = fe_zero(), vi = ks.one;
    auto term = [&](const fe_t &e0, const fe_t &e1) { cip = fe_add<FS>(cip, fe_mul<FS>(vi, fe_add<FS>(e0, fe_mul<FS>(u, e1)))); vi = fe_mul<FS>(vi, v); };
    for (uint32_t i = 0; i < n_prev; ++i) {
        fe_t e[2];
        for (int side = 0; side < 2; ++side) {
            fe_t pw = side ? zetaw : zeta, acc = ks.one;
            for (int j = (int)k - 1; j >= 0; --j) {
                const fe_t ch = fe_to_mont<FS>(load_fe<FS>(in.prev_chals + (((size_t)b * n_prev + i) * k + j) * 8), ks.r2);
                acc = fe_mul<FS>(acc, fe_add<FS>(ks.one, fe_mul<FS>(ch, pw))); pw = fe_sqr<FS>(pw);
            }
            e[side] = acc;
        }
        term(e[0], e[1]);
    }
    term(pub_e[0], pub_e[1]);
    term(ft, fe_to_mont<FS>(load_fe<FS>(in.ft_eval1 + (size_t)b * 8), ks.r2));
    for (uint32_t c = 0; c < KC_COLS; ++c) term(EV(c, 0), EV(c, 1));
    store_fe<LANES>(out.cip + (size_t)b * 8, fe_from_mont<FS>(cip));
    store_fe<LANES>(out.evalpoints + (size_t)b * 16, fe_from_mont<FS>(zeta)); store_fe<LANES>(out.evalpoints + (size_t)b * 16 + 8, fe_from_mont<FS>(zetaw));
    store_fe<LANES>(out.polyscale + (size_t)b * 8, fe_from_mont<FS>(v)); store_fe<LANES>(out.evalscale + (size_t)b * 8, fe_from_mont<FS>(u));
    if (out.ft_eval0) store_fe<LANES>(out.ft_eval0 + (size_t)b * 8, fe_from_mont<FS>(ft));

    // ---- ft_comm = perm_scalar * sigma_6 - (zeta^n - 1) * sum_i zeta^(n i) t_i : term j on lane j mod LANES, then a shuffle tree
    xyzz_t part = xyzz_inf();
    {
        fe_t zpow = ks.one;                                            // zeta^(n i)
        const fe_t neg_zm1 = fe_neg<FS>(fe_sub<FS>(zeta1, ks.one));
        for (uint32_t j = 0; j < 8; ++j) {
            fe_t sc; affine_t P;
            if (j == 0) { sc = perm_scalar; P = ix->sigma6; }
            else { sc = fe_mul<FS>(neg_zm1, zpow); zpow = fe_mul<FS>(zpow, zeta1); bool dummy = true; P = load_point_checked<FB>(in.t_comm + ((size_t)b * 7 + (j - 1)) * 16, kb, dummy); }
            if ((j & (LANES - 1)) == ln) { xyzz_t r = scalar_mul_affine<FB>(fe_from_mont<FS>(sc), P, kb); xyzz_add<FB>(part, r); }
        }
        for (int m = 1; m < LANES; m <<= 1) { const xyzz_t o2 = xyzz_shfl_xor(part, m); xyzz_add<FB>(part, o2); }
    }
    if (ln == 0) {
        uint32_t *o = comms_out + (n_prev + 1) * 16;
        if (xyzz_is_inf(part)) { for (int i = 0; i < 16; ++i) o[i] = 0; }
        else {
            const fe_t zi = fe_inv<FB>(fe_mul<FB>(part.zz, part.zzz), kb);
            const fe_t x = fe_from_mont<FB>(fe_mul<FB>(part.x, fe_mul<FB>(zi, part.zzz))), y = fe_from_mont<FB>(fe_mul<FB>(part.y, fe_mul<FB>(zi, part.zz)));
            for (int i = 0; i < 8; ++i) { o[i] = x.v[i]; o[8 + i] = y.v[i]; }
        }
        // the 43 column commitments: z, 6 selectors (index), 15 w, 15 coefficients (index), 6 sigma (index)
        uint32_t *cc = comms_out + (n_prev + 2) * 16;
        for (int i = 0; i < 16; ++i) cc[i] = in.z_comm[(size_t)b * 16 + i];
        for (int i = 0; i < 6 * 16; ++i) cc[16 + i] = ix->col_comm_words[i];
        for (int i = 0; i < 15 * 16; ++i) cc[7 * 16 + i] = in.w_comm[(size_t)b * 15 * 16 + i];
        for (int i = 0; i < 21 * 16; ++i) cc[22 * 16 + i] = ix->col_comm_words[6 * 16 + i];
        if (!ok) *bad_input = 1u;
    }
}

}  // namespace mb

// ------------------------------------------------------------------------------------------------ the verifier index
namespace {
template <int F> fe_t host_mont(const uint8_t *b, const FieldK &k) { fe_t a; memcpy(a.v, b, 32); return fe_to_mont<F>(a, k.r2); }
}  // namespace

int mb_kimchi_available(mina_ctx *c) { return c && c->have_kimchi ? 1 : 0; }

extern "C" int mina_verifier_index_install(mina_ctx *c, const mina_verifier_index *vi) {
    if (!c || !vi || !vi->shifts || !vi->sigma_comm || !vi->coefficients_comm || !vi->selector_comm || (vi->constant_term_len && !vi->constant_term)) return fail(MINA_ERR_ARG, "null argument");
    if (vi->log2_domain < 1 || vi->log2_domain > 20 || vi->zk_rows < 1 || vi->zk_rows > mb::KC_MAX_ZK || vi->perm_alpha_offset > 1024) return fail(MINA_ERR_ARG, "bad domain / zk_rows / alpha offset");
    if (((uint64_t)1 << vi->log2_domain) > c->srs[CURVE_PALLAS].depth) return fail(c->srs[CURVE_PALLAS].depth ? MINA_ERR_ARG : MINA_ERR_STATE, "Pallas SRS missing or smaller than the domain");
    if (!c->have_pparams[FIELD_FP] || !c->have_pparams[FIELD_FQ]) return fail(MINA_ERR_STATE, "Poseidon constants not installed");
    std::vector<mb::KimchiToken> toks; std::vector<std::array<uint8_t, 32>> lits;
    if (!mb::decode_tokens(vi->constant_term, vi->constant_term_len, FIELD_FQ, mb::KC_COLS, toks, lits)) return fail(MINA_ERR_FORMAT, "malformed PolishToken program");
    for (int i = 0; i < 7; ++i) if (!mw::fq_canonical(vi->shifts + 32 * i)) return fail(MINA_ERR_FORMAT, "shift is not a canonical scalar");
    const uint8_t *groups[3] = {vi->selector_comm, vi->coefficients_comm, vi->sigma_comm}; const int counts[3] = {6, 15, 7};
    for (int g = 0; g < 3; ++g) for (int i = 0; i < counts[g] * 2; ++i) if (!mw::fp_canonical(groups[g] + 32 * i)) return fail(MINA_ERR_FORMAT, "index commitment coordinate is not canonical");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    const FieldK &ks = c->fk[FIELD_FQ], &kb = c->fk[FIELD_FP];
    auto *ix = new mb::KimchiIndexDev(); std::unique_ptr<mb::KimchiIndexDev> hold(ix);
    memset(ix, 0, sizeof *ix);
    ix->log2_domain = vi->log2_domain; ix->zk_rows = vi->zk_rows; ix->perm_alpha_offset = vi->perm_alpha_offset; ix->n_tokens = (uint32_t)toks.size();
    for (int i = 0; i < 7; ++i) ix->shifts[i] = host_mont<FIELD_FQ>(vi->shifts + 32 * i, ks);
    fe_t w = ks.root; for (uint32_t i = 0; i < 32 - vi->log2_domain; ++i) w = fe_sqr<FIELD_FQ>(w);
    ix->omega = w;
    const uint64_t n = (uint64_t)1 << vi->log2_domain;
    auto powu = [&](fe_t base, uint64_t e) { fe_t r = ks.one; for (; e; e >>= 1) { if (e & 1) r = fe_mul<FIELD_FQ>(r, base); base = fe_sqr<FIELD_FQ>(base); } return r; };
    ix->omega_zk = powu(w, n - vi->zk_rows);
    for (uint32_t i = 0; i < vi->zk_rows; ++i) ix->zk_roots[i] = powu(w, n - vi->zk_rows + i);
    { fe_t nn = fe_zero(); nn.v[0] = (uint32_t)n; nn.v[1] = (uint32_t)(n >> 32); ix->n_inv = fe_inv<FIELD_FQ>(fe_to_mont<FIELD_FQ>(nn, ks.r2), ks); }
    ix->endo_coeff = fe_sqr<FIELD_FQ>(ks.endo);                          // ks.endo = (cube root)^2, and (w^2)^2 = w
    {   // scalar-field Poseidon MDS from the installed parameter block
        PoseidonParams pp; HIPC(hipMemcpy(&pp, c->pparams[FIELD_FQ].p, sizeof pp, hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) ix->mds[3 * i + j] = pp.mds[i][j];
    }
    ix->sigma6.x = host_mont<FIELD_FP>(vi->sigma_comm + 6 * 64, kb); ix->sigma6.y = host_mont<FIELD_FP>(vi->sigma_comm + 6 * 64 + 32, kb);
    memcpy(ix->col_comm_words, vi->selector_comm, 6 * 64);
    memcpy(ix->col_comm_words + 6 * 16, vi->coefficients_comm, 15 * 64);
    memcpy(ix->col_comm_words + 21 * 16, vi->sigma_comm, 6 * 64);
    // digest: Fq-sponge over sigma, coefficients, selectors, squeezed as a base-field element (tape machine on the GPU)
    {
        std::vector<uint8_t> tape(28, MINA_TAPE_ABSORB_G); tape.push_back(MINA_TAPE_CHALLENGE_FQ);
        std::vector<uint8_t> inp(28 * 64);
        memcpy(inp.data(), vi->sigma_comm, 7 * 64); memcpy(inp.data() + 7 * 64, vi->coefficients_comm, 15 * 64); memcpy(inp.data() + 22 * 64, vi->selector_comm, 6 * 64);
        uint8_t dg[32];
        int rc = mina_fq_sponge_run(c, CURVE_PALLAS, 1, tape.data(), tape.size(), nullptr, nullptr, inp.data(), dg, nullptr, nullptr);
        if (rc) return rc;
        memcpy(c->kimchi_digest, dg, 32);
        ix->digest = host_mont<FIELD_FP>(dg, kb);
    }
    std::vector<fe_t> lm(lits.size() ? lits.size() : 1);
    for (size_t i = 0; i < lits.size(); ++i) lm[i] = host_mont<FIELD_FQ>(lits[i].data(), ks);
    int rc;
    if ((rc = c->kimchi_index.ensure(sizeof *ix))) return rc;
    if ((rc = c->kimchi_tokens.ensure((toks.size() ? toks.size() : 1) * sizeof(mb::KimchiToken)))) return rc;
    if ((rc = c->kimchi_literals.ensure(lm.size() * sizeof(fe_t)))) return rc;
    HIPC(hipMemcpy(c->kimchi_index.p, ix, sizeof *ix, hipMemcpyHostToDevice));
    if (!toks.empty()) HIPC(hipMemcpy(c->kimchi_tokens.p, toks.data(), toks.size() * sizeof(mb::KimchiToken), hipMemcpyHostToDevice));
    HIPC(hipMemcpy(c->kimchi_literals.p, lm.data(), lm.size() * sizeof(fe_t), hipMemcpyHostToDevice));
    memcpy(c->kimchi_comms_host, vi->sigma_comm, 7 * 64); memcpy(c->kimchi_comms_host + 7 * 64, vi->coefficients_comm, 15 * 64); memcpy(c->kimchi_comms_host + 22 * 64, vi->selector_comm, 6 * 64);
    c->kimchi_log2 = vi->log2_domain; c->have_kimchi = true;
    return MINA_OK;
}

extern "C" int mina_verifier_index_digest(mina_ctx *c, uint8_t *out32) {
    if (!c || !out32) return fail(MINA_ERR_ARG, "null argument");
    if (!c->have_kimchi) return fail(MINA_ERR_STATE, "no verifier index installed");
    memcpy(out32, c->kimchi_digest, 32);
    return MINA_OK;
}

// queue oracles + to_batch for `batch` proofs on the current lane; every pointer is a device pointer
int mb_kimchi_to_batch_dev(mina_ctx *c, size_t batch, uint32_t n_prev, uint32_t npub, const mb::KimchiIn &in, const mb::KimchiOut &out, uint32_t *d_bad) {
    if (!c->have_kimchi) return fail(MINA_ERR_STATE, "no verifier index installed");
    const PoseidonParams *ppb = c->pparams[FIELD_FP].as<PoseidonParams>(), *pps = c->pparams[FIELD_FQ].as<PoseidonParams>();
    ProfScope ps_(c, PS_KIMCHI);
    if (use_coop8(c, batch))
        mb::kimchi_to_batch_kernel<8><<<cdiv(batch * 8, 64), 64, 0, c->L->stream>>>((uint32_t)batch, n_prev, npub, c->fk[FIELD_FP], c->fk[FIELD_FQ], ppb, pps,
            c->kimchi_index.as<mb::KimchiIndexDev>(), c->kimchi_tokens.as<mb::KimchiToken>(), c->kimchi_literals.as<fe_t>(), in, out, d_bad);
    else
        mb::kimchi_to_batch_kernel<4><<<cdiv(batch * 4, 64), 64, 0, c->L->stream>>>((uint32_t)batch, n_prev, npub, c->fk[FIELD_FP], c->fk[FIELD_FQ], ppb, pps,
            c->kimchi_index.as<mb::KimchiIndexDev>(), c->kimchi_tokens.as<mb::KimchiToken>(), c->kimchi_literals.as<fe_t>(), in, out, d_bad);
    HIPC(hipGetLastError());
    return MINA_OK;
}

int mb_ensure_lagrange_table(mina_ctx *c, int curve, uint32_t log2_domain, uint32_t npub);   // api_srs.hip
int mb_pubcomm_dev(mina_ctx *c, size_t batch, uint32_t log2_domain, uint32_t npub, const uint32_t *d_pub, uint32_t *d_out16);   // api_state.hip

static int check_kimchi_in(mina_ctx *c, const mina_kimchi_proofs *p) {
    if (!c || !p) return fail(MINA_ERR_ARG, "null argument");
    if (!c->have_kimchi) return fail(MINA_ERR_STATE, "no verifier index installed");
    if (p->batch == 0 || p->batch > 65536 || p->n_prev > 8 || p->npub > 4096 || p->npub > (1u << c->kimchi_log2)) return fail(MINA_ERR_ARG, "bad batch / n_prev / npub");
    if ((p->npub && !p->public_inputs) || (p->n_prev && (!p->prev_chals || !p->prev_comms)) || !p->w_comm || !p->z_comm || !p->t_comm || !p->evals || !p->ft_eval1) return fail(MINA_ERR_ARG, "null section");
    return MINA_OK;
}

// host-buffer form (tests / tooling): the BatchEvaluationProof rows back on the host
extern "C" int mina_kimchi_to_batch(mina_ctx *c, const mina_kimchi_proofs *p, mina_kimchi_batch_out *o) {
    int rc = check_kimchi_in(c, p);
    if (rc) return rc;
    if (!o || !o->sponge_state || !o->sponge_pos || !o->cip || !o->evalpoints || !o->polyscale || !o->evalscale || !o->comms) return fail(MINA_ERR_ARG, "null output");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    const size_t B = p->batch, k = c->kimchi_log2, ncomms = p->n_prev + 2 + mb::KC_COLS;
    if (p->npub && (rc = mb_ensure_lagrange_table(c, CURVE_PALLAS, (uint32_t)k, p->npub))) return rc;
    c->use_lane0();
    Lane &L = *c->L;
    struct Sec { const void *src; size_t bytes; size_t off; };
    std::vector<Sec> secs = {{p->public_inputs, B * p->npub * 32, 0}, {p->prev_chals, B * p->n_prev * k * 32, 0}, {p->prev_comms, B * p->n_prev * 64, 0}, {p->w_comm, B * 15 * 64, 0},
                             {p->z_comm, B * 64, 0}, {p->t_comm, B * 7 * 64, 0}, {p->evals, B * mb::KC_COLS * 64, 0}, {p->ft_eval1, B * 32, 0}};
    size_t total = 0; for (auto &s : secs) { s.off = total; total += (s.bytes + 255) & ~(size_t)255; }
    const size_t o_state = total, o_pos = o_state + B * 96, o_cip = o_pos + ((B * 8 + 255) & ~(size_t)255), o_pts = o_cip + B * 32, o_v = o_pts + B * 64, o_u = o_v + B * 32,
                 o_comms = o_u + B * 32, o_ft = o_comms + B * ncomms * 64, o_pc = o_ft + B * 32, o_bad = o_pc + B * 64, all = o_bad + 256;
    if ((rc = L.host_stage.ensure(total + 16))) return rc;
    for (auto &s : secs) if (s.bytes) memcpy((uint8_t *)L.host_stage.p + s.off, s.src, s.bytes);
    if ((rc = L.st_in.ensure(all))) return rc;
    uint8_t *d = L.st_in.as<uint8_t>();
    HIPC(hipMemcpyAsync(d, L.host_stage.p, total, hipMemcpyHostToDevice, L.stream));
    HIPC(hipMemsetAsync(d + o_bad, 0, 4, L.stream));
    auto W = [&](size_t off) { return (uint32_t *)(d + off); };
    if (p->npub) { if ((rc = mb_pubcomm_dev(c, B, (uint32_t)k, p->npub, W(secs[0].off), W(o_pc)))) return rc; }
    else { std::vector<uint8_t> hb(B * 64); uint8_t h1[64]; if ((rc = mina_srs_get_h(c, CURVE_PALLAS, h1))) return rc; c->use_lane0(); for (size_t i = 0; i < B; ++i) memcpy(&hb[i * 64], h1, 64); HIPC(hipMemcpyAsync(d + o_pc, hb.data(), B * 64, hipMemcpyHostToDevice, L.stream)); HIPC(hipStreamSynchronize(L.stream)); }
    mb::KimchiIn in{W(secs[0].off), W(secs[1].off), W(secs[2].off), W(secs[3].off), W(secs[4].off), W(secs[5].off), W(secs[6].off), W(secs[7].off), W(o_pc)};
    mb::KimchiOut out{W(o_state), W(o_pos), W(o_cip), W(o_pts), W(o_v), W(o_u), W(o_comms), W(o_ft)};
    if ((rc = mb_kimchi_to_batch_dev(c, B, p->n_prev, p->npub, in, out, W(o_bad)))) return rc;
    std::vector<uint8_t> back(all - o_state);
    HIPC(hipMemcpyAsync(back.data(), d + o_state, back.size(), hipMemcpyDeviceToHost, L.stream));
    HIPC(hipStreamSynchronize(L.stream));
    auto R = [&](size_t off) { return back.data() + (off - o_state); };
    memcpy(o->sponge_state, R(o_state), B * 96); memcpy(o->sponge_pos, R(o_pos), B * 8); memcpy(o->cip, R(o_cip), B * 32); memcpy(o->evalpoints, R(o_pts), B * 64);
    memcpy(o->polyscale, R(o_v), B * 32); memcpy(o->evalscale, R(o_u), B * 32); memcpy(o->comms, R(o_comms), B * ncomms * 64);
    if (o->ft_eval0) memcpy(o->ft_eval0, R(o_ft), B * 32);
    if (o->malformed) { uint32_t f; memcpy(&f, R(o_bad), 4); *o->malformed = f ? 1 : 0; }
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------ WrapProof -> kimchi section (verify_mina_state)
// Gathers the wrap proofs' kimchi inputs into host arrays for mina_state_job_batch.  The wrap circuit's PUBLIC INPUT is the
// Pickles statement packed into scalars (`tock_unpadded_public_input_of_statement`): that packing needs the step circuit's
// deferred values (combined inner product, b, zeta powers, perm), which in turn need the STEP linearization -- data this tree does
// not hold.  With a step index installed (mina_step_index_install) the 40 public inputs are derived from the statement
// (api_pickles.hip); without one the wrap proof is verified with an EMPTY public input (npub = 0) [flagged in DESIGN.md].
int mb_step_index_installed(mina_ctx *c);                                                                                          // api_pickles.hip
int mb_pickles_public_inputs(mina_ctx *c, const mw::WrapProof *const *proofs, const uint8_t *const *app_states, size_t n, uint8_t *pub_out, uint8_t *derived_out, uint8_t *ok_out);

int mb_kimchi_fill_jobs(mina_ctx *c, const mw::WrapProof *const *proofs, const uint8_t *const *tip_hashes, size_t n, mina_state_jobs *jobs,
                        std::vector<std::vector<uint8_t>> &storage, std::vector<uint8_t> &statement_ok) {
    const uint32_t k = c->kimchi_log2;
    storage.assign(16, {});
    statement_ok.assign(n, 1);
    auto &pub = storage[0], &pch = storage[1], &pcm = storage[2], &wc = storage[3], &zc = storage[4], &tc = storage[5], &ev = storage[6], &ft1 = storage[7],
         &lr = storage[8], &dl = storage[9], &sg = storage[10], &z1 = storage[11], &z2 = storage[12], &rb = storage[13], &sb = storage[14], &kp = storage[15];
    const uint32_t n_prev = 2;
    uint32_t npub = 0;
    if (mb_step_index_installed(c)) {           // the wrap circuit's public input = the Pickles statement, deferred values recomputed (api_pickles.hip)
        npub = 40; pub.resize(n * 40 * 32);
        int prc = mb_pickles_public_inputs(c, proofs, tip_hashes, n, pub.data(), nullptr, statement_ok.data());
        if (prc) return prc;
    }
    auto put_pt = [](std::vector<uint8_t> &v, const mw::Pt &p) { v.insert(v.end(), p.x.b, p.x.b + 32); v.insert(v.end(), p.y.b, p.y.b + 32); };
    auto put32 = [](std::vector<uint8_t> &v, const mw::B32 &x) { v.insert(v.end(), x.b, x.b + 32); };
    for (size_t b = 0; b < n; ++b) {
        const mw::WrapProof &w = *proofs[b];
        if (w.lr.size() != k || w.step_challenge_polynomial_commitments.size() != n_prev) return fail(MINA_ERR_FORMAT, "wrap proof shape does not match the installed index");
        // recursion challenges of the wrap proof: messages_for_next_wrap_proof.old_bulletproof_challenges, expanded with the Pallas endo_r
        for (uint32_t a = 0; a < n_prev; ++a) for (uint32_t j = 0; j < k; ++j) {
            const mw::Chal128 &ch = w.old_bulletproof_challenges[a][j < 15 ? j : 14];
            const fe_t e = fe_from_mont<FIELD_FQ>(challenge_to_field<FIELD_FQ>(ch.lo, ch.hi, c->fk[FIELD_FQ]));
            const uint8_t *eb = (const uint8_t *)e.v;
            pch.insert(pch.end(), eb, eb + 32);
        }
        for (uint32_t a = 0; a < n_prev; ++a) put_pt(pcm, w.step_challenge_polynomial_commitments[a]);
        for (int i = 0; i < 15; ++i) put_pt(wc, w.w_comm[i]);
        put_pt(zc, w.z_comm);
        for (int i = 0; i < 7; ++i) put_pt(tc, w.t_comm[i]);
        put32(ev, w.z_eval[0]); put32(ev, w.z_eval[1]);
        for (int i = 0; i < 6; ++i) { put32(ev, w.selector_eval[i][0]); put32(ev, w.selector_eval[i][1]); }
        for (int i = 0; i < 15; ++i) { put32(ev, w.w_eval[i][0]); put32(ev, w.w_eval[i][1]); }
        for (int i = 0; i < 15; ++i) { put32(ev, w.coefficients_eval[i][0]); put32(ev, w.coefficients_eval[i][1]); }
        for (int i = 0; i < 6; ++i) { put32(ev, w.s_eval[i][0]); put32(ev, w.s_eval[i][1]); }
        put32(ft1, w.ft_eval1);
        for (auto &q : w.lr) { put_pt(lr, q.first); put_pt(lr, q.second); }
        put_pt(dl, w.delta); put_pt(sg, w.sg); put32(z1, w.z1); put32(z2, w.z2);
    }
    rb.assign(32, 0); rb[0] = 7; sb.assign(32, 0); sb[0] = 9;
    kp.resize(sizeof(mina_kimchi_proofs));
    mina_kimchi_proofs kk{}; kk.batch = n; kk.n_prev = n_prev; kk.npub = npub; kk.prev_chals = pch.data(); kk.prev_comms = pcm.data(); kk.w_comm = wc.data();
    kk.z_comm = zc.data(); kk.t_comm = tc.data(); kk.evals = ev.data(); kk.ft_eval1 = ft1.data();
    memcpy(kp.data(), &kk, sizeof kk);
    jobs->batch = n; jobs->with_ipa = 1; jobs->kimchi = (const mina_kimchi_proofs *)kp.data();
    jobs->k = k; jobs->n_evalpoints = 2; jobs->n_comms = n_prev + 2 + mb::KC_COLS; jobs->log2_domain = k; jobs->npub = npub;
    if (npub) { jobs->public_inputs = pub.data(); kk.public_inputs = pub.data(); memcpy(kp.data(), &kk, sizeof kk); }
    jobs->lr = lr.data(); jobs->delta = dl.data(); jobs->sg = sg.data(); jobs->z1 = z1.data(); jobs->z2 = z2.data(); jobs->rand_base = rb.data(); jobs->sg_rand_base = sb.data();
    return MINA_OK;
}
