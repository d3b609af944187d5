// kimchi_dev.cuh -- device pieces shared by the two places that evaluate a kimchi proof's `ft_eval0`:
//   api_kimchi.hip   the WRAP proof being verified (scalar field Fq), inside `oracles`
//   api_pickles.hip  the STEP proof whose deferred values the statement carries (Fp), inside `compute_deferred_values`
// Same formula, same PolishToken interpreter, different field / index data.  [UPSTREAM-RECALL] kimchi `ft_eval0`:
//   ft_eval0 = (w_6 + gamma) z(zeta w) a0 zkpm prod_{i<6}(beta s_i + w_i + gamma)  -  p(zeta)
//            - a0 zkpm z(zeta) prod_{i<7}(gamma + beta zeta shift_i + w_i)
//            + (zeta^n - 1)(a1 (zeta - w^(n-zk)) + a2 (zeta - 1))(1 - z(zeta)) / ((zeta - w^(n-zk))(zeta - 1))
//            - constant_term(linearization)
// with a_i = alpha^(perm_alpha_offset + i), zkpm = prod_{i<zk} (zeta - w^(n-zk+i)).  CPU restatement: oracle/kimchi_ref.py.
#pragma once
#include "ctx.h"
#include "sponge.cuh"
#include "polish.h"

namespace mb {

static constexpr uint32_t KC_COLS = 43, KC_W0 = 7, KC_C0 = 22, KC_S0 = 37, KC_MAX_ZK = 8;

template <int F> __device__ __forceinline__ fe_t ld_checked(const uint32_t *p, const FieldK &k, bool &ok) {
    const fe_t w = load_fe<F>(p); ok = ok && fe_words_canonical<F>(w); return fe_to_mont<F>(w, k.r2);
}
template <int F> __device__ __forceinline__ fe_t fe_pow2k(fe_t a, uint32_t k) { for (uint32_t i = 0; i < k; ++i) a = fe_sqr<F>(a); return a; }
template <int F> __device__ fe_t fe_pow_u64(const fe_t &a, uint64_t e, const fe_t &one) {
    fe_t r = one, b = a;
    for (; e; e >>= 1) { if (e & 1) r = fe_mul<F>(r, b); b = fe_sqr<F>(b); }
    return r;
}
// 128-bit squeeze as a plain element of field FS (kimchi `fq_sponge.challenge()`: beta, gamma)
template <int FS> __device__ __forceinline__ fe_t chal128_plain(const fe_t &sq_plain, const FieldK &ks) {
    fe_t o = fe_zero(); o.v[0] = sq_plain.v[0]; o.v[1] = sq_plain.v[1]; o.v[2] = sq_plain.v[2]; o.v[3] = sq_plain.v[3];
    return fe_to_mont<FS>(o, ks.r2);
}
template <int FS> __device__ __forceinline__ fe_t chal_endo(const fe_t &sq_plain, const FieldK &ks) {
    const uint64_t lo = (uint64_t)sq_plain.v[0] | ((uint64_t)sq_plain.v[1] << 32), hi = (uint64_t)sq_plain.v[2] | ((uint64_t)sq_plain.v[3] << 32);
    return challenge_to_field<FS>(lo, hi, ks);
}
template <int F, int LANES> __device__ __forceinline__ void sponge_init(DevSponge<F, LANES> &s, const PoseidonParams *pp) { s.pp = pp; s.squeezed = 0; s.count = 0; s.s = fe_zero(); }
// interpreter stack + cache of one lane in LDS, word-major: the 64 lanes of a slot hit 64 banks.  KC_SLOTS * 8 * 64 words per block of 64
static constexpr int KC_SLOTS = KC_STACK + KC_CACHE;
struct LdsStack {
    uint32_t *base;
    __device__ __forceinline__ void put(int slot, const fe_t &a) { for (int i = 0; i < 8; ++i) base[(slot * 8 + i) * 64] = a.v[i]; }
    __device__ __forceinline__ fe_t get(int slot) const { fe_t a; for (int i = 0; i < 8; ++i) a.v[i] = base[(slot * 8 + i) * 64]; return a; }
};

// what ft_eval0 needs to know about the circuit and the proof's oracles (field F, Montgomery)
struct FtEnv {
    fe_t alpha, beta, gamma, zeta, zeta_n;                        // zeta_n = zeta^(2^k)
    fe_t omega, omega_zk, endo; const fe_t *zk_roots, *shifts, *mds, *lits; const KimchiToken *toks;
    uint32_t k, zk_rows, alpha0, n_tokens;
    fe_t joint = fe_zero(); uint32_t features = 0;                // the proof's joint combiner (0 without one) and feature mask (polish.h feature_mask_of_flags)
};
// returns ft_eval0; `perm_scalar` = -z(zeta w) beta a0 zkpm prod_{i<6}(gamma + beta s_i + w_i); ok = false on a broken program.
// EV(col, row): the proof's evaluation of column `col` at zeta (row 0) / zeta*omega (row 1), Montgomery.
template <int F, class EvFn>
__device__ fe_t ft_eval0_dev(const FtEnv &e, const FieldK &fk, const fe_t &pub0, EvFn EV, LdsStack st, bool &ok, fe_t &perm_scalar) {
    const fe_t zm1 = fe_sub<F>(e.zeta_n, fk.one);
    const fe_t a0 = fe_pow_u64<F>(e.alpha, e.alpha0, fk.one);
    fe_t zkpm = fk.one;
    for (uint32_t i = 0; i < e.zk_rows; ++i) zkpm = fe_mul<F>(zkpm, fe_sub<F>(e.zeta, e.zk_roots[i]));
    const fe_t z0 = EV(0, 0), z1 = EV(0, 1);
    fe_t prod6 = fk.one;
#pragma unroll 1
    for (uint32_t i = 0; i < 6; ++i) prod6 = fe_mul<F>(prod6, fe_add<F>(fe_add<F>(fe_mul<F>(e.beta, EV(KC_S0 + i, 0)), EV(KC_W0 + i, 0)), e.gamma));
    const fe_t common = fe_mul<F>(fe_mul<F>(a0, zkpm), prod6);
    fe_t ft = fe_mul<F>(fe_mul<F>(fe_add<F>(EV(KC_W0 + 6, 0), e.gamma), z1), common);
    ft = fe_sub<F>(ft, pub0);
    {
        fe_t t2 = fe_mul<F>(fe_mul<F>(a0, zkpm), z0);
        const fe_t bz = fe_mul<F>(e.beta, e.zeta);
#pragma unroll 1
        for (uint32_t i = 0; i < 7; ++i) t2 = fe_mul<F>(t2, fe_add<F>(fe_add<F>(e.gamma, fe_mul<F>(bz, e.shifts[i])), EV(KC_W0 + i, 0)));
        ft = fe_sub<F>(ft, t2);
        const fe_t a1 = fe_mul<F>(a0, e.alpha), a2 = fe_mul<F>(a1, e.alpha);
        const fe_t dw = fe_sub<F>(e.zeta, e.omega_zk), d1 = fe_sub<F>(e.zeta, fk.one);
        const fe_t num = fe_mul<F>(fe_add<F>(fe_mul<F>(fe_mul<F>(zm1, a1), dw), fe_mul<F>(fe_mul<F>(zm1, a2), d1)), fe_sub<F>(fk.one, z0));
        ft = fe_add<F>(ft, fe_mul<F>(num, fe_inv<F>(fe_mul<F>(dw, d1), fk)));
    }
    if (e.n_tokens) {   // linearization constant term: PolishToken stack machine (the program is uniform over the lanes: no divergence)
        int sp = 0, nc = 0; bool prog_ok = true; uint32_t skip = 0;      // `skip` is per lane (the proofs' feature flags differ); the token loop itself stays uniform
#pragma unroll 1
        for (uint32_t t = 0; t < e.n_tokens; ++t) {
            const KimchiToken tk = e.toks[t];
            if (skip) { --skip; continue; }                              // a skipped token has no effect (polish.h: upstream's convention)
            switch (tk.op) {
                case MINA_TOK_SKIP_IF: case MINA_TOK_SKIP_IF_NOT: {
                    const bool on = (e.features >> tk.a) & 1u;
                    if (on == (tk.op == MINA_TOK_SKIP_IF)) { skip = tk.b; st.put(sp++, fe_zero()); }
                    break; }
                case MINA_TOK_ALPHA: st.put(sp++, e.alpha); break;
                case MINA_TOK_BETA: st.put(sp++, e.beta); break;
                case MINA_TOK_GAMMA: st.put(sp++, e.gamma); break;
                case MINA_TOK_JOINT_COMBINER: st.put(sp++, e.joint); break;
                case MINA_TOK_ENDO_COEFFICIENT: st.put(sp++, e.endo); break;
                case MINA_TOK_MDS: st.put(sp++, e.mds[tk.a * 3 + tk.b]); break;
                case MINA_TOK_LITERAL: st.put(sp++, e.lits[tk.a]); break;
                case MINA_TOK_CELL: st.put(sp++, EV(tk.a, tk.b)); break;
                case MINA_TOK_DUP: st.put(sp, st.get(sp - 1)); ++sp; break;
                case MINA_TOK_POW: st.put(sp - 1, fe_pow_u64<F>(st.get(sp - 1), (uint64_t)tk.a | ((uint64_t)tk.b << 32), fk.one)); break;
                case MINA_TOK_ADD: st.put(sp - 2, fe_add<F>(st.get(sp - 2), st.get(sp - 1))); --sp; break;
                case MINA_TOK_MUL: st.put(sp - 2, fe_mul<F>(st.get(sp - 2), st.get(sp - 1))); --sp; break;
                case MINA_TOK_SUB: st.put(sp - 2, fe_sub<F>(st.get(sp - 2), st.get(sp - 1))); --sp; break;
                case MINA_TOK_VANISHES_ON_ZK_ROWS: st.put(sp++, zkpm); break;
                case MINA_TOK_UNNORMALIZED_LAGRANGE: {
                    const int32_t off = (int32_t)tk.a;
                    const uint64_t row = off >= 0 ? (uint64_t)off : ((uint64_t)1 << e.k) - e.zk_rows - (off == INT32_MIN ? 0 : (uint64_t)(-(int64_t)off));   // INT32_MIN: the first zero-knowledge row itself
                    const fe_t wr = fe_pow_u64<F>(e.omega, row, fk.one);
                    st.put(sp++, fe_mul<F>(zm1, fe_inv<F>(fe_sub<F>(e.zeta, wr), fk))); break; }
                case MINA_TOK_STORE: st.put(KC_STACK + nc++, st.get(sp - 1)); break;
                case MINA_TOK_LOAD: if ((int)tk.a >= nc) prog_ok = false; st.put(sp++, st.get(KC_STACK + ((int)tk.a < nc ? (int)tk.a : 0))); break;   // a slot no executed STORE filled: this proof fails
                default: prog_ok = false;
            }
        }
        if (sp != 1 || !prog_ok) ok = false; else ft = fe_sub<F>(ft, st.get(0));   // the host validated stack depth: defensive
    }
    perm_scalar = fe_neg<F>(fe_mul<F>(fe_mul<F>(z1, e.beta), common));
    return ft;
}

}  // namespace mb
