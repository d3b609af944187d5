// api_kimchi.hip -- kimchi `verifier::{oracles, to_batch}` on the GPU (transcript order README.md:413-475).
//
// Replaces kimchi `ProverProof::oracles` + `to_batch` (pin core/Cargo.toml:14) for proofs over Pallas (the Pickles wrap proof).
// Per proof, in six small kernels (see "the stages" below):
//   * the Fq-sponge: index digest, recursion commitments, public-input commitment, w -> beta, gamma -> z -> alpha -> t -> zeta
//   * the Fr-sponge: digest of the Fq-sponge, digest of the recursion challenges, ft_eval1, public evaluations, the 43 x 2 column
//     evaluations -> v, u
//   * the scalar-field work: negated public polynomial at zeta / zeta*omega, ft_eval0 (permutation part, boundary part, the
//     linearization's constant term through a PolishToken interpreter), perm scalar, b_poly of the recursion challenges, the
//     combined inner product
//   * the chunked ft commitment  perm_scalar * sigma_6 - (zeta^n - 1) * sum_i zeta^(n i) t_i   (8 scalar multiplications over 8 lanes)
// and emits one `BatchEvaluationProof` row per proof in the layout `mb_ipa_batch_check_dev` consumes.
// The verifier index (domain, shifts, commitments, token program) is DATA installed by the caller (`mina_verifier_index`): the
// reference tree does not hold the blockchain-snark index.  [UPSTREAM-RECALL] throughout; checked against oracle/kimchi_ref.py,
// whose miniature prover mints proofs this code must accept and whose tampered variants it must reject.
#include "ctx.h"
#include "msm.cuh"
#include "sponge.cuh"
#include "wire_proof.h"
#include "kimchi_dev.cuh"

namespace mb {

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

// ---- the stages.  One monolithic kernel (round 2 first cut) held the Fq state, the Fr state, 86 evaluations' worth of scalar
// temporaries and a 24-deep interpreter stack live at once: 255 VGPRs + 2.4 KB of scratch per lane, non-inlined calls, and the
// sponges -- 9/10 of the dependent work -- ran 15x slower per permutation than `pstate_hash_kernel`.  Split by what each stage needs:
//   fq      lane-cooperative (8 or 3 lanes), two roles in separate waves of one launch: proof b's Fq-sponge, and beside it the digest of
//           its recursion challenges (a scalar-field sponge that depends on nothing else)
//   pub     8 lanes per proof: the negated public polynomial at zeta, zeta*omega (8-term chunks spread over the lanes)
//   fr      lane-cooperative: Fr-sponge -> v, u
//   scalar  one lane per proof: ft_eval0, the PolishToken program (stack in LDS), perm scalar, b_poly evaluations, cip, ft scalars
//   ftcomm  8 lanes per proof: the 8 scalar multiplications of the chunked ft commitment + shuffle tree
//   rows    one thread per word: the commitment rows that are copies
// Values pass between stages through `xf` (KC_XF scalar-field Montgomery elements per proof).
enum { XF_BETA = 0, XF_GAMMA, XF_ALPHA, XF_ZETA, XF_DIGEST, XF_PFDIGEST, XF_PUB0, XF_PUB1, XF_V, XF_U, XF_FTSC = 10, KC_XF = 18 };

__device__ __forceinline__ fe_t fe_shfl_xor(const fe_t &a, int mask) {
    fe_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__shfl_xor((int)a.v[i], mask, 64);
    return r;
}
// sum of `x` over the group's lanes, on every lane
template <int F, int LANES> __device__ __forceinline__ fe_t coop_sum(const fe_t &x) {
    if (LANES == 3) { const uint32_t base = tri_pos().base; return fe_add<F>(fe_add<F>(tri_bcast(x, base), tri_bcast(x, base + 1)), tri_bcast(x, base + 2)); }
    fe_t a = x;
    for (int m = 1; m < LANES; m <<= 1) a = fe_add<F>(a, fe_shfl_xor(a, m));
    return a;
}

template <int LANES>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LANES == 3 ? 4 : 1, 8)))   // 3-lane form: the four waves per SIMD it had before the signed-digit forms (+ 6 VGPRs)
kimchi_fq_kernel(uint32_t batch, uint32_t n_prev, FieldK kb, FieldK ks, const PoseidonParams *__restrict__ pp_b, const PoseidonParams *__restrict__ pp_s,
                 const KimchiIndexDev *__restrict__ ix, KimchiIn in, KimchiOut out, fe_t *__restrict__ xf, uint32_t *__restrict__ bad_input, uint32_t nblk,
                 const fe_t *__restrict__ pf_digest /* or null: role 1 computes it */, uint32_t pf_stride) { mb_wave_prio();
    constexpr int FB = FIELD_FP, FS = FIELD_FQ;                 // Pallas: base Fp, scalar Fq
    bool writer; const uint32_t role = blockIdx.x / nblk, b = coop_role_item<LANES>(blockIdx.x % nblk, writer);       // one role per wave
    if (b >= batch) return;
    bool ok = true;
    if (role) {                                                 // digest of the recursion challenges
        const uint32_t cnt = n_prev * ix->log2_domain;
        const uint32_t *p = in.prev_chals + (size_t)b * cnt * 8;
        DevSponge<FS, LANES> pf; sponge_init(pf, pp_s);
#pragma unroll 1
        for (uint32_t i = 0; i < cnt; ++i) pf.absorb(ld_checked<FS>(p + (size_t)i * 8, ks, ok));
        const fe_t d = pf.squeeze();
        if (writer) { xf[(size_t)b * KC_XF + XF_PFDIGEST] = d; if (!ok) *bad_input = 1u; }
        return;
    }
    if (pf_digest && writer) xf[(size_t)b * KC_XF + XF_PFDIGEST] = pf_digest[(size_t)b * pf_stride];     // the statement stage ran that sponge already (api_pickles.hip)
    DevSponge<FB, LANES> fq; sponge_init(fq, pp_b);
    fq.absorb(ix->digest);
    auto absorb_pts = [&](const uint32_t *p, uint32_t n) {
#pragma unroll 1
        for (uint32_t i = 0; i < n; ++i) { const affine_t P = load_point_checked<FB>(p + (size_t)i * 16, kb, ok); fq.absorb(P.x); fq.absorb(P.y); }
    };
    absorb_pts(in.prev_comms + (size_t)b * n_prev * 16, n_prev);
    absorb_pts(in.pubcomm + (size_t)b * 16, 1);
    absorb_pts(in.w_comm + (size_t)b * 15 * 16, 15);
    fe_t *x = xf + (size_t)b * KC_XF;
    { const fe_t beta = chal128_plain<FS>(fe_from_mont<FB>(fq.squeeze()), ks); if (writer) x[XF_BETA] = beta; }
    { const fe_t gamma = chal128_plain<FS>(fe_from_mont<FB>(fq.squeeze()), ks); if (writer) x[XF_GAMMA] = gamma; }
    absorb_pts(in.z_comm + (size_t)b * 16, 1);
    { const fe_t alpha = chal_endo<FS>(fe_from_mont<FB>(fq.squeeze()), ks); if (writer) x[XF_ALPHA] = alpha; }
    absorb_pts(in.t_comm + (size_t)b * 7 * 16, 7);
    { const fe_t zeta = chal_endo<FS>(fe_from_mont<FB>(fq.squeeze()), ks); if (writer) x[XF_ZETA] = zeta; }
    {   // the sponge handed to the opening check: element e of the state from the lane that owns it
        if (coop_state_owner<LANES>()) { const fe_t w = fe_from_mont<FB>(fq.s); for (int i = 0; i < 8; ++i) out.sponge_state[(size_t)b * 24 + coop_elem<LANES>() * 8 + i] = w.v[i]; }
        if (writer) { out.sponge_pos[2 * b] = (uint32_t)fq.squeezed; out.sponge_pos[2 * b + 1] = (uint32_t)fq.count; }
    }
    const fe_t digest = fe_to_mont<FS>(fe_from_mont<FB>(fq.squeeze()), ks.r2);      // on a copy upstream; p < q: always fits.  fq is dead after this
    if (writer) { x[XF_DIGEST] = digest; if (!ok) *bad_input = 1u; }
}

// negated public polynomial at zeta and zeta*omega:  -(x^n - 1)/n * sum_i p_i w^i / (x - w^i).  8 lanes per proof; work item = (side,
// chunk of CH terms), round-robin over the lanes, each inverts its CH denominators with one field inversion; shuffle-tree sum.
// CH = 10 makes the 40 public inputs of a wrap proof exactly 8 items -- one pass of the lanes; with chunks of 8 they are 10 items and the
// wave runs a second pass (inversion included) for two of its eight lanes.
template <int CH>
__global__ void __launch_bounds__(64)
kimchi_pub_kernel(uint32_t batch, uint32_t npub, FieldK ks, const KimchiIndexDev *__restrict__ ix, KimchiIn in, fe_t *__restrict__ xf, uint32_t *__restrict__ bad_input) { mb_wave_prio();
    constexpr int FS = FIELD_FQ;
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x, b = gid >> 3, ln = gid & 7u;
    if (b >= batch) return;
    bool ok = true;
    fe_t *x = xf + (size_t)b * KC_XF;
    const uint32_t k = ix->log2_domain;
    const fe_t zeta = x[XF_ZETA], zetaw = fe_mul<FS>(zeta, ix->omega);
    const fe_t omega_ch = fe_pow_u64<FS>(ix->omega, (uint64_t)CH, ks.one);
    fe_t acc[2] = {fe_zero(), fe_zero()};
    const uint32_t nchunks = (npub + CH - 1) / CH;
#pragma unroll 1
    for (uint32_t it = ln; it < 2 * nchunks; it += 8) {
        const uint32_t side = it & 1u, base = (it >> 1) * CH, cnt = npub - base < (uint32_t)CH ? npub - base : (uint32_t)CH;
        const fe_t pt = side ? zetaw : zeta;
        fe_t wi = fe_pow_u64<FS>(omega_ch, it >> 1, ks.one);
        fe_t den[CH], pre[CH], run = ks.one;
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)CH; ++j) if (j < cnt) { den[j] = fe_sub<FS>(pt, wi); pre[j] = run; run = fe_mul<FS>(run, den[j]); wi = fe_mul<FS>(wi, ix->omega); }
        fe_t inv = fe_inv<FS>(run, ks), part = fe_zero();
#pragma unroll
        for (int j = CH - 1; j >= 0; --j) if ((uint32_t)j < cnt) {
            const fe_t dinv = fe_mul<FS>(inv, pre[j]); inv = fe_mul<FS>(inv, den[j]);
            const fe_t p = ld_checked<FS>(in.pub + ((size_t)b * npub + base + j) * 8, ks, ok);
            part = fe_add<FS>(part, fe_mul<FS>(fe_mul<FS>(dinv, p), fe_sub<FS>(pt, den[j])));       // w^i = pt - (pt - w^i)
        }
        if (side) acc[1] = fe_sub<FS>(acc[1], part); else acc[0] = fe_sub<FS>(acc[0], part);
    }
    for (int m = 1; m < 8; m <<= 1) { acc[0] = fe_add<FS>(acc[0], fe_shfl_xor(acc[0], m)); acc[1] = fe_add<FS>(acc[1], fe_shfl_xor(acc[1], m)); }
    if (ln < 2) {
        const fe_t ptn = fe_pow2k<FS>(ln ? zetaw : zeta, k);
        x[XF_PUB0 + ln] = fe_mul<FS>(fe_mul<FS>(ln ? acc[1] : acc[0], fe_sub<FS>(ptn, ks.one)), ix->n_inv);
    }
    if (!ok) *bad_input = 1u;
}

template <int LANES>
__global__ void __launch_bounds__(64)
kimchi_fr_kernel(uint32_t batch, FieldK ks, const PoseidonParams *__restrict__ pp_s, KimchiIn in, fe_t *__restrict__ xf, uint32_t *__restrict__ bad_input) { mb_wave_prio();
    constexpr int FS = FIELD_FQ;
    bool writer; const uint32_t b = coop_sponge_index<LANES>(writer);
    if (b >= batch) return;
    bool ok = true;
    fe_t *x = xf + (size_t)b * KC_XF;
    DevSponge<FS, LANES> fr; sponge_init(fr, pp_s);
    fr.absorb(x[XF_DIGEST]);
    fr.absorb(x[XF_PFDIGEST]);
    fr.absorb(ld_checked<FS>(in.ft_eval1 + (size_t)b * 8, ks, ok));
    fr.absorb(x[XF_PUB0]); fr.absorb(x[XF_PUB1]);
    const uint32_t *ev = in.evals + (size_t)b * KC_COLS * 16;           // [col][zeta | zeta_omega][8]
#pragma unroll 1
    for (uint32_t c = 0; c < KC_COLS * 2; ++c) fr.absorb(ld_checked<FS>(ev + (size_t)c * 8, ks, ok));
    const fe_t v = chal_endo<FS>(fe_from_mont<FS>(fr.squeeze()), ks);
    const fe_t u = chal_endo<FS>(fe_from_mont<FS>(fr.squeeze()), ks);
    if (writer) { x[XF_V] = v; x[XF_U] = u; if (!ok) *bad_input = 1u; }
}

// one lane per proof; the interpreter's stack and cache live in LDS (kimchi_dev.cuh)
__global__ void __launch_bounds__(64)
kimchi_scalar_kernel(uint32_t batch, uint32_t n_prev, FieldK ks, const KimchiIndexDev *__restrict__ ix, const KimchiToken *__restrict__ toks,
                     const fe_t *__restrict__ lits, KimchiIn in, KimchiOut out, fe_t *__restrict__ xf, uint32_t *__restrict__ bad_input) { mb_wave_prio();
    constexpr int FS = FIELD_FQ;
    __shared__ uint32_t lds[KC_SLOTS * 8 * 64];
    const uint32_t b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    LdsStack st{lds + threadIdx.x};
    fe_t *x = xf + (size_t)b * KC_XF;
    const uint32_t k = ix->log2_domain;
    const uint32_t *ev = in.evals + (size_t)b * KC_COLS * 16;
    auto EV = [&](uint32_t col, uint32_t row) { return fe_to_mont<FS>(load_fe<FS>(ev + ((size_t)col * 2 + row) * 8), ks.r2); };
    const fe_t zeta = x[XF_ZETA];
    const fe_t zeta1 = fe_pow2k<FS>(zeta, k), zetaw = fe_mul<FS>(zeta, ix->omega);
    const fe_t zm1 = fe_sub<FS>(zeta1, ks.one);
    bool ok = true;
    fe_t perm_scalar;
    FtEnv env{x[XF_ALPHA], x[XF_BETA], x[XF_GAMMA], zeta, zeta1, ix->omega, ix->omega_zk, ix->endo_coeff, ix->zk_roots, ix->shifts, ix->mds, lits, toks,
              k, ix->zk_rows, ix->perm_alpha_offset, ix->n_tokens};
    const fe_t ft = ft_eval0_dev<FS>(env, ks, x[XF_PUB0], EV, st, ok, perm_scalar);
    const fe_t v = x[XF_V], u = x[XF_U];
    // combined inner product over the evaluation list: recursion, public, ft, then the 43 columns
    fe_t cip = fe_zero(), vi = ks.one;
    auto term = [&](const fe_t &e0, const fe_t &e1) { cip = fe_add<FS>(cip, fe_mul<FS>(vi, fe_add<FS>(e0, fe_mul<FS>(u, e1)))); vi = fe_mul<FS>(vi, v); };
#pragma unroll 1
    for (uint32_t i = 0; i < n_prev; ++i) {
        fe_t pw0 = zeta, pw1 = zetaw, acc0 = ks.one, acc1 = ks.one;    // b_poly(chals, zeta), b_poly(chals, zeta*omega): two chains interleaved
#pragma unroll 1
        for (int j = (int)k - 1; j >= 0; --j) {
            const fe_t ch = fe_to_mont<FS>(load_fe<FS>(in.prev_chals + (((size_t)b * n_prev + i) * k + j) * 8), ks.r2);
            acc0 = fe_mul<FS>(acc0, fe_add<FS>(ks.one, fe_mul<FS>(ch, pw0))); pw0 = fe_sqr<FS>(pw0);
            acc1 = fe_mul<FS>(acc1, fe_add<FS>(ks.one, fe_mul<FS>(ch, pw1))); pw1 = fe_sqr<FS>(pw1);
        }
        term(acc0, acc1);
    }
    term(x[XF_PUB0], x[XF_PUB1]);
    term(ft, fe_to_mont<FS>(load_fe<FS>(in.ft_eval1 + (size_t)b * 8), ks.r2));
#pragma unroll 1
    for (uint32_t c = 0; c < KC_COLS; ++c) term(EV(c, 0), EV(c, 1));
    auto put = [](uint32_t *p, const fe_t &a) { for (int i = 0; i < 8; ++i) p[i] = a.v[i]; };
    put(out.cip + (size_t)b * 8, fe_from_mont<FS>(cip));
    put(out.evalpoints + (size_t)b * 16, fe_from_mont<FS>(zeta)); put(out.evalpoints + (size_t)b * 16 + 8, fe_from_mont<FS>(zetaw));
    put(out.polyscale + (size_t)b * 8, fe_from_mont<FS>(v)); put(out.evalscale + (size_t)b * 8, fe_from_mont<FS>(u));
    if (out.ft_eval0) put(out.ft_eval0 + (size_t)b * 8, fe_from_mont<FS>(ft));
    // scalars of ft_comm = perm_scalar * sigma_6 - (zeta^n - 1) * sum_i zeta^(n i) t_i
    x[XF_FTSC] = perm_scalar;
    fe_t sc = fe_neg<FS>(zm1);
    for (uint32_t j = 1; j < 8; ++j) { x[XF_FTSC + j] = sc; sc = fe_mul<FS>(sc, zeta1); }
    if (!ok) *bad_input = 1u;
}

// 8 lanes per proof: term j on lane j, then a shuffle tree; lane 0 normalises and writes row n_prev + 1.  The t commitments were
// checked by the fq stage (a malformed one already failed the batch).  Only the host-buffer form (`mina_kimchi_to_batch`, which
// returns the rows) runs this: the verifier hands the 8 (point, scalar) pairs to the opening check's MSM instead (IpaExpand)
__global__ void __launch_bounds__(64)
kimchi_ftcomm_kernel(uint32_t batch, uint32_t n_prev, FieldK kb, const KimchiIndexDev *__restrict__ ix, KimchiIn in, KimchiOut out, const fe_t *__restrict__ xf) { mb_wave_prio();
    constexpr int FB = FIELD_FP;
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x, b = gid >> 3, j = gid & 7u;
    if (b >= batch) return;
    const fe_t sc = fe_from_mont<FIELD_FQ>(xf[(size_t)b * KC_XF + XF_FTSC + j]);      // plain words: the bits drive double-and-add
    const affine_t P = j == 0 ? ix->sigma6 : load_point_mont<FB>(in.t_comm + ((size_t)b * 7 + (j - 1)) * 16, kb);
    xyzz_t part = scalar_mul_affine<FB>(sc, P, kb);
    for (int m = 1; m < 8; m <<= 1) { const xyzz_t o2 = xyzz_shfl_xor(part, m); xyzz_add<FB>(part, o2); }
    if (j == 0) {
        uint32_t *o = out.comms + ((size_t)b * (n_prev + 2 + KC_COLS) + n_prev + 1) * 16;
        if (xyzz_is_inf(part)) { for (int i = 0; i < 16; ++i) o[i] = 0; }
        else {
            const fe_t zi = fe_inv<FB>(fe_mul<FB>(part.zz, part.zzz), kb);
            const fe_t x = fe_from_mont<FB>(fe_mul<FB>(part.x, fe_mul<FB>(zi, part.zzz))), y = fe_from_mont<FB>(fe_mul<FB>(part.y, fe_mul<FB>(zi, part.zz)));
            for (int i = 0; i < 8; ++i) { o[i] = x.v[i]; o[8 + i] = y.v[i]; }
        }
    }
}

// the rows of the commitment list that are copies: recursion accumulators, the public-input commitment, then the 43 columns
// z, 6 selectors (index), 15 w, 15 coefficients (index), 6 sigma (index).  Row n_prev + 1 (ft) belongs to the ftcomm stage.
__global__ void __launch_bounds__(256)
kimchi_rows_kernel(uint32_t batch, uint32_t n_prev, const KimchiIndexDev *__restrict__ ix, KimchiIn in, uint32_t *__restrict__ comms) { mb_wave_prio();
    const uint32_t ncomms = n_prev + 2 + KC_COLS;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)batch * ncomms * 16) return;
    const uint32_t w = (uint32_t)(gid & 15u), row = (uint32_t)((gid >> 4) % ncomms); const size_t b = (gid >> 4) / ncomms;
    uint32_t val;
    if (row < n_prev) val = in.prev_comms[(b * n_prev + row) * 16 + w];
    else if (row == n_prev) val = in.pubcomm[b * 16 + w];
    else if (row == n_prev + 1) return;
    else {
        const uint32_t c = row - (n_prev + 2);
        if (c == 0) val = in.z_comm[b * 16 + w];
        else if (c < 7) val = ix->col_comm_words[(c - 1) * 16 + w];
        else if (c < 22) val = in.w_comm[(b * 15 + (c - 7)) * 16 + w];
        else val = ix->col_comm_words[(6 + (c - 22)) * 16 + w];
    }
    comms[gid] = val;
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
    c->kimchi_log2 = vi->log2_domain; c->have_kimchi = true; c->pickles_ms_valid = false;
    return MINA_OK;
}

extern "C" int mina_verifier_index_digest(mina_ctx *c, uint8_t *out32) {
    if (!c || !out32) return fail(MINA_ERR_ARG, "null argument");
    if (!c->have_kimchi) return fail(MINA_ERR_STATE, "no verifier index installed");
    memcpy(out32, c->kimchi_digest, 32);
    return MINA_OK;
}

// queue oracles + to_batch for `batch` proofs on the current lane; every pointer is a device pointer
// `expand` non-null: row n_prev + 1 of the commitment list is NOT computed; *expand describes it as 8 (point, scalar) pairs for
// mb_ipa_batch_check_dev (IpaShape::expand_slot = n_prev + 1)
int mb_kimchi_to_batch_dev(mina_ctx *c, size_t batch, uint32_t n_prev, uint32_t npub, const mb::KimchiIn &in, const mb::KimchiOut &out, uint32_t *d_bad, mb::IpaExpand *expand,
                           const void *pf_digest, uint32_t pf_stride) {
    if (!c->have_kimchi) return fail(MINA_ERR_STATE, "no verifier index installed");
    const PoseidonParams *ppb = c->pparams[FIELD_FP].as<PoseidonParams>(), *pps = c->pparams[FIELD_FQ].as<PoseidonParams>();
    ProfScope ps_(c, PS_KIMCHI);
    Lane &L = *c->L;
    int rc;
    if ((rc = L.kc_xfer.ensure(batch * mb::KC_XF * sizeof(fe_t)))) return rc;
    fe_t *xf = L.kc_xfer.as<fe_t>();
    const mb::KimchiIndexDev *ix = c->kimchi_index.as<mb::KimchiIndexDev>();
    const uint32_t B = (uint32_t)batch, ncomms = n_prev + 2 + mb::KC_COLS;
    const FieldK &kb = c->fk[FIELD_FP], &ks = c->fk[FIELD_FQ];
    mb::kimchi_rows_kernel<<<cdiv(batch * ncomms * 16, 256), 256, 0, L.stream>>>(B, n_prev, ix, in, out.comms);
    // 8-lane sponges up to 1024 proofs per call (shortest dependent chain), 3-lane above: measured on bench.py --kimchi, 8192 proofs
    // per step -- 16 x 512: 165 k/s (8-lane) vs 162 k/s; 4 x 2048: 137 k/s (8-lane) vs 150 k/s (3-lane).  mina_verify_tuning.kimchi_coop8_max overrides
    const size_t coop8_max = (size_t)mb_tune().kimchi_coop8_max;
    const uint32_t fq_roles = pf_digest ? 1u : 2u;              // role 1 = the digest of the recursion challenges, unless the statement stage supplies it
    if (use_coop16(c, batch)) {
        mb::kimchi_fq_kernel<16><<<fq_roles * coop_role_blocks<16>(batch), 64, 0, L.stream>>>(B, n_prev, kb, ks, ppb, pps, ix, in, out, xf, d_bad, coop_role_blocks<16>(batch), (const fe_t *)pf_digest, pf_stride);
        if (npub > 32 && npub <= 40) mb::kimchi_pub_kernel<10><<<cdiv(batch * 8, 64), 64, 0, L.stream>>>(B, npub, ks, ix, in, xf, d_bad);
        else mb::kimchi_pub_kernel<8><<<cdiv(batch * 8, 64), 64, 0, L.stream>>>(B, npub, ks, ix, in, xf, d_bad);
        mb::kimchi_fr_kernel<16><<<cdiv(coop_threads<16>(batch), 64), 64, 0, L.stream>>>(B, ks, pps, in, xf, d_bad);
    } else if (use_coop8_transcripts(c, batch, coop8_max)) {
        mb::kimchi_fq_kernel<8><<<fq_roles * coop_role_blocks<8>(batch), 64, 0, L.stream>>>(B, n_prev, kb, ks, ppb, pps, ix, in, out, xf, d_bad, coop_role_blocks<8>(batch), (const fe_t *)pf_digest, pf_stride);
        if (npub > 32 && npub <= 40) mb::kimchi_pub_kernel<10><<<cdiv(batch * 8, 64), 64, 0, L.stream>>>(B, npub, ks, ix, in, xf, d_bad);
        else mb::kimchi_pub_kernel<8><<<cdiv(batch * 8, 64), 64, 0, L.stream>>>(B, npub, ks, ix, in, xf, d_bad);
        mb::kimchi_fr_kernel<8><<<cdiv(coop_threads<8>(batch), 64), 64, 0, L.stream>>>(B, ks, pps, in, xf, d_bad);
    } else {                        // chip-filling batch: 21 sponges per wave
        mb::kimchi_fq_kernel<3><<<fq_roles * coop_role_blocks<3>(batch), 64, 0, L.stream>>>(B, n_prev, kb, ks, ppb, pps, ix, in, out, xf, d_bad, coop_role_blocks<3>(batch), (const fe_t *)pf_digest, pf_stride);
        if (npub > 32 && npub <= 40) mb::kimchi_pub_kernel<10><<<cdiv(batch * 8, 64), 64, 0, L.stream>>>(B, npub, ks, ix, in, xf, d_bad);
        else mb::kimchi_pub_kernel<8><<<cdiv(batch * 8, 64), 64, 0, L.stream>>>(B, npub, ks, ix, in, xf, d_bad);
        mb::kimchi_fr_kernel<3><<<cdiv(coop_threads<3>(batch), 64), 64, 0, L.stream>>>(B, ks, pps, in, xf, d_bad);
    }
    mb::kimchi_scalar_kernel<<<cdiv(batch, 64), 64, 0, L.stream>>>(B, n_prev, ks, ix, c->kimchi_tokens.as<mb::KimchiToken>(), c->kimchi_literals.as<fe_t>(), in, out, xf, d_bad);
    if (expand) { expand->p0 = &ix->sigma6; expand->pts = in.t_comm; expand->sc = xf + mb::XF_FTSC; expand->stride = mb::KC_XF; }
    else mb::kimchi_ftcomm_kernel<<<cdiv(batch * 8, 64), 64, 0, L.stream>>>(B, n_prev, kb, ix, in, out, xf);
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
    if ((rc = mb_kimchi_to_batch_dev(c, B, p->n_prev, p->npub, in, out, W(o_bad), nullptr, nullptr, 0))) return rc;
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
