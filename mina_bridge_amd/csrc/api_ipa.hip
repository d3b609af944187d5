// api_ipa.hip -- the combined IPA checks on top of K1..K4.
//
// Replaces poly-commitment `SRS::verify` (batched opening check) and openmina `accumulator_check`
// (pins core/Cargo.toml:16,23; README.md:469-475, 534-544).  Per proof b (rho = rand_base^b,
// sigma = sg_rand_base^b) upstream pushes onto ONE multi-scalar multiplication
//     g[j] += sigma * s_b[j]          h -= rho z2          sg: -rho z1 - sigma        U: -rho z1 b0 + rho c cip
//     L_j: rho c / chal_j             R_j: rho c chal_j    comm_i: rho c xi^i         delta: rho
// and asserts the result is the identity.  Here the g-part is K2 (fold) + K1 (fixed-base tables) and the
// per-proof points go through the variable-base K1 path; the two partial results are compared on the GPU.
// The Fiat-Shamir transcript of each proof runs as one lane of `ipa_prepare_kernel` (throughput layout:
// one proof per lane; a wave-cooperative sponge is the planned latency optimisation).
#include "ctx.h"
#include "msm.cuh"
#include "sponge.cuh"

namespace mb {

// ---------------------------------------------------------------- generic Fq-sponge transcript ("tape")
// mina-poseidon `DefaultFqSponge` over the base field of CURVE (pins core/Cargo.toml:14; README.md:413-475 lists the order
// kimchi absorbs/squeezes in).  Every proof of a batch runs the same tape of opcodes over its own input stream:
enum : uint8_t {
    TAPE_ABSORB_FQ = 0,        // one base-field element                                   (32 B in)
    TAPE_ABSORB_G = 1,         // one affine point: x then y; infinity absorbs (0, 0)      (64 B in)
    TAPE_ABSORB_FR = 2,        // one scalar-field element: whole if r < q, else (x >> 1, x & 1)   (32 B in)
    TAPE_CHALLENGE = 3,        // squeeze, low 128 bits                                    (32 B out, upper 16 zero)
    TAPE_CHALLENGE_FQ = 4,     // squeeze, full base-field element                         (32 B out)
    TAPE_CHALLENGE_ENDO = 5,   // squeeze 128 bits -> ScalarChallenge::to_field (scalar field)      (32 B out)
    TAPE_DIGEST = 6,           // squeeze; as a scalar-field element if it fits, else 0    (32 B out)
    TAPE_CHALLENGE_ENDO_OWN = 7, // squeeze 128 bits -> to_field in the sponge's OWN field (kimchi `DefaultFrSponge::challenge`:
                               // run the tape on the curve whose BASE field is the proof's scalar field)      (32 B out)
};
template <int CURVE, int LANES>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LANES == 3 ? 4 : 1, 8)))   // 3-lane form: the four waves per SIMD it had before the signed-digit forms (+ 6 VGPRs)
sponge_tape_kernel(uint32_t batch, uint32_t tape_len, uint32_t in_stride_words, uint32_t out_stride_words, FieldK kb, FieldK ks,
                   const PoseidonParams *__restrict__ pp, const uint8_t *__restrict__ tape, const uint32_t *__restrict__ init_state /* b*24 or null */,
                   const uint32_t *__restrict__ init_pos /* b*2 or null */, const uint32_t *__restrict__ inputs, uint32_t *__restrict__ outputs,
                   uint32_t *__restrict__ final_state /* b*24 or null */, uint32_t *__restrict__ final_pos /* b*2 or null */) {
    constexpr int FB = (CURVE == CURVE_PALLAS) ? FIELD_FP : FIELD_FQ;
    constexpr int FS = (CURVE == CURVE_PALLAS) ? FIELD_FQ : FIELD_FP;
    bool writer_; const uint32_t b = coop_sponge_index<LANES>(writer_), qq = coop_elem<LANES>();
    if (b >= batch) return;                                   // whole lane groups leave together
    DevSponge<FB, LANES> sp; sp.pp = pp; sp.squeezed = 0; sp.count = 0; sp.s = fe_zero();
    if (init_state) { sp.s = fe_to_mont<FB>(load_fe<FB>(init_state + (size_t)b * 24 + qq * 8), kb.r2); sp.squeezed = (int)init_pos[2 * b]; sp.count = (int)init_pos[2 * b + 1]; }
    const uint32_t *in = inputs + (size_t)b * in_stride_words;
    uint32_t *out = outputs + (size_t)b * out_stride_words;
    for (uint32_t t = 0; t < tape_len; ++t) {
        const uint8_t op = tape[t];
        if (op == TAPE_ABSORB_FQ) { sp.absorb(fe_to_mont<FB>(load_fe<FB>(in), kb.r2)); in += 8; }
        else if (op == TAPE_ABSORB_G) { sp.absorb(fe_to_mont<FB>(load_fe<FB>(in), kb.r2)); sp.absorb(fe_to_mont<FB>(load_fe<FB>(in + 8), kb.r2)); in += 16; }
        else if (op == TAPE_ABSORB_FR) {
            fe_t x = load_fe<FS>(in); in += 8;
            if (CURVE == CURVE_PALLAS) {                       // scalar modulus (q) > base modulus (p): split off the low bit
                fe_t lowbit = fe_zero(); lowbit.v[0] = x.v[0] & 1u;
                fe_t hi = x; for (int i = 0; i < 7; ++i) hi.v[i] = (hi.v[i] >> 1) | (hi.v[i + 1] << 31); hi.v[7] >>= 1;
                sp.absorb(fe_to_mont<FB>(hi, kb.r2)); sp.absorb(fe_to_mont<FB>(lowbit, kb.r2));
            } else sp.absorb(fe_to_mont<FB>(x, kb.r2));
        } else {
            fe_t sq = fe_from_mont<FB>(sp.squeeze()), o = fe_zero();
            if (op == TAPE_CHALLENGE) { o.v[0] = sq.v[0]; o.v[1] = sq.v[1]; o.v[2] = sq.v[2]; o.v[3] = sq.v[3]; }
            else if (op == TAPE_CHALLENGE_FQ) o = sq;
            else if (op == TAPE_CHALLENGE_ENDO) {
                uint64_t lo = (uint64_t)sq.v[0] | ((uint64_t)sq.v[1] << 32), hi = (uint64_t)sq.v[2] | ((uint64_t)sq.v[3] << 32);
                o = fe_from_mont<FS>(challenge_to_field<FS>(lo, hi, ks));
            } else if (op == TAPE_CHALLENGE_ENDO_OWN) {
                uint64_t lo = (uint64_t)sq.v[0] | ((uint64_t)sq.v[1] << 32), hi = (uint64_t)sq.v[2] | ((uint64_t)sq.v[3] << 32);
                o = fe_from_mont<FB>(challenge_to_field<FB>(lo, hi, kb));
            } else {                                          // TAPE_DIGEST: fits the scalar field?  (compare with its modulus)
                bool fits = false;
                for (int i = 7; i >= 0; --i) { uint32_t m = modulus_limb<FS>(i); if (sq.v[i] != m) { fits = sq.v[i] < m; break; } }
                if (fits) o = sq;
            }
            store_fe<LANES>(out, o); out += 8;
        }
    }
    if (final_state && coop_state_owner<LANES>()) { fe_t w = fe_from_mont<FB>(sp.s); for (int i = 0; i < 8; ++i) final_state[(size_t)b * 24 + qq * 8 + i] = w.v[i]; }
    if (final_pos && writer_) { final_pos[2 * b] = (uint32_t)sp.squeezed; final_pos[2 * b + 1] = (uint32_t)sp.count; }
}


// One lane group (8 lanes, or a wave-packed triple above 1024 proofs per call) per proof: the Fq-sponge runs lane-cooperatively (6.3 / 4.65 dependent product latencies per
// Poseidon round instead of 21), all other (scalar-field) work is computed redundantly by the lanes, lane 0 writes.
// CURVE fixes (FB, FS).
// PHASE 0: the whole transcript.  PHASE 1 / 2: split at the first squeeze -- `U = to_group(t)` (an inversion and 1-3 square
// roots, ~0.75 ms of dependent products) is not fed back into the sponge, so it runs as its own kernel on a second stream
// while PHASE 2 continues the transcript; the sponge is handed over through `xfer` (40 words per proof: state in
// Montgomery form, position, t).
static constexpr uint32_t IPA_XFER_WORDS = 40;
template <int CURVE, int LANES, int PHASE>
__global__ void __launch_bounds__(64)
ipa_prepare_kernel(IpaShape sh, FieldK kb, FieldK ks, const PoseidonParams *__restrict__ pp,
                   const uint32_t *__restrict__ sponge_state /* b*24 */, const uint32_t *__restrict__ sponge_pos /* b*2 */,
                   const uint32_t *__restrict__ cip /* b*8 */,
                   const uint32_t *__restrict__ lr /* b*2k*16 */, const uint32_t *__restrict__ delta /* b*16 */,
                   const uint32_t *__restrict__ sg /* b*16 */, const uint32_t *__restrict__ z1, const uint32_t *__restrict__ z2,
                   const uint32_t *__restrict__ evalpoints /* b*npts*8 */, const uint32_t *__restrict__ evalscale,
                   const uint32_t *__restrict__ polyscale, const uint32_t *__restrict__ comms /* b*ncomms*16 */,
                   const uint32_t *__restrict__ comm_override /* b*16 or null */, IpaExpand ex,
                   const uint32_t *__restrict__ rand_base, const uint32_t *__restrict__ sg_rand_base,
                   const affine_t *__restrict__ srs_h,
                   affine_t *__restrict__ out_points /* b*per */, uint32_t *__restrict__ out_scalars /* b*per*8 canonical */,
                   uint32_t *__restrict__ out_chals /* b*k*8 canonical */, uint32_t *__restrict__ out_sigma /* b*8 canonical */,
                   uint32_t *__restrict__ bad_input /* zeroed by the host; set to 1 on a malformed point */,
                   uint32_t *__restrict__ xfer /* PHASE 1 writes, PHASE 2 reads: b * IPA_XFER_WORDS */,
                   uint32_t *__restrict__ shared_sc /* b * nshared * 8 canonical, or null */, uint32_t *__restrict__ shared_off /* nshared list offsets */) { mb_wave_prio();
    constexpr int FB = (CURVE == CURVE_PALLAS) ? FIELD_FP : FIELD_FQ;
    constexpr int FS = (CURVE == CURVE_PALLAS) ? FIELD_FQ : FIELD_FP;
    bool writer_; const uint32_t b = coop_sponge_index<LANES>(writer_);
    if (b >= sh.batch) return;                                // whole lane groups leave together
    const uint32_t k = sh.k;
    bool pts_ok = true;                                       // every input well-formed (canonical field elements, points on the curve)
    uint32_t *xf = xfer ? xfer + (size_t)b * IPA_XFER_WORDS : nullptr;

    // ---- Fq-sponge transcript (base field)
    DevSponge<FB, LANES> sp; sp.pp = pp; sp.squeezed = 0; sp.count = 0;
    if (PHASE == 2) {                                         // resume after the first squeeze
        sp.s = load_fe<FB>(xf + coop_elem<LANES>() * 8);
        sp.squeezed = (int)xf[24]; sp.count = (int)xf[25];
    } else {
        const fe_t w = load_fe<FB>(sponge_state + (size_t)b * 24 + coop_elem<LANES>() * 8);
        bool okw = true;                                      // all three state elements, checked redundantly by every lane
        for (int e = 0; e < 3; ++e) okw = okw && fe_words_canonical<FB>(load_fe<FB>(sponge_state + (size_t)b * 24 + e * 8));
        pts_ok = pts_ok && okw;
        sp.s = fe_to_mont<FB>(w, kb.r2);
    }
    if (PHASE != 2) { sp.squeezed = (int)sponge_pos[2 * b]; sp.count = (int)sponge_pos[2 * b + 1]; }
    auto load_scalar_checked = [&](const uint32_t *p) {           // scalar-field element: canonical or the batch is rejected
        const fe_t w = load_fe<FS>(p);
        pts_ok = pts_ok && fe_words_canonical<FS>(w);
        return fe_to_mont<FS>(w, ks.r2);
    };
    const fe_t cip_m = PHASE == 2 ? fe_to_mont<FS>(load_fe<FS>(cip + (size_t)b * 8), ks.r2) : load_scalar_checked(cip + (size_t)b * 8);
    affine_t U; U.x = fe_zero(); U.y = fe_zero();
    if (PHASE != 2) {   // absorb_fr(shift_scalar(cip))
        const fe_t two255 = ks.two255;
        if (CURVE == CURVE_PALLAS) {
            // scalar modulus > base modulus: x = cip - 2^255 ; absorb (x >> 1), then (x & 1)
            fe_t x = fe_from_mont<FS>(fe_sub<FS>(cip_m, two255));
            fe_t lowbit = fe_zero(); lowbit.v[0] = x.v[0] & 1u;
            fe_t hi = x; for (int i = 0; i < 7; ++i) hi.v[i] = (hi.v[i] >> 1) | (hi.v[i + 1] << 31); hi.v[7] >>= 1;
            sp.absorb(fe_to_mont<FB>(hi, kb.r2));
            sp.absorb(fe_to_mont<FB>(lowbit, kb.r2));
        } else {
            // scalar modulus < base modulus: x = (cip - (2^255 + 1)) / 2, absorbed as one base-field element
            fe_t x = fe_from_mont<FS>(fe_mul<FS>(fe_sub<FS>(cip_m, fe_add<FS>(two255, ks.one)), ks.inv2));
            sp.absorb(fe_to_mont<FB>(x, kb.r2));
        }
        const fe_t t = sp.squeeze();                          // challenge_fq
        if (PHASE == 1) {                                     // hand over: state (owner lanes), position, t; flag; done
            if (coop_state_owner<LANES>()) for (int i = 0; i < 8; ++i) xf[coop_elem<LANES>() * 8 + i] = sp.s.v[i];
            if (writer_) { xf[24] = (uint32_t)sp.squeezed; xf[25] = (uint32_t)sp.count; for (int i = 0; i < 8; ++i) xf[26 + i] = t.v[i]; }
            if (!pts_ok && writer_) *bad_input = 1u;
            return;
        }
        U = bw_to_group<FB>(t, kb);
    }

    affine_t *pts = out_points + (size_t)b * sh.per;
    uint32_t *scs = out_scalars + (size_t)b * sh.per * 8;
    // entries whose point is shared by the whole batch: zero scalar in the list, the real one in the side matrix, the point once at the tail
    uint32_t shq = 0;
    auto put = [&](uint32_t o, const affine_t &P, const fe_t &sc_canon, bool shared) {
        store_pt<LANES>(&pts[o], P);
        if (shared && sh.nshared) {
            store_fe<LANES>(scs + (size_t)o * 8, fe_zero());
            store_fe<LANES>(shared_sc + ((size_t)b * sh.nshared + shq) * 8, sc_canon);
            if (b == 0) { store_pt<LANES>(out_points + (size_t)sh.batch * sh.per + shq, P); if (coop_writer<LANES>()) shared_off[shq] = o; }
            ++shq;
        } else store_fe<LANES>(scs + (size_t)o * 8, sc_canon);
    };

    // rho = rand_base^(b + pow_first), sigma = sg_rand_base^(b + pow_first)   (pow_first = 0: upstream's rho_b = rand_base^b)
    fe_t rho = ks.one, sigma = ks.one;
    {
        fe_t rb = fe_to_mont<FS>(load_fe<FS>(rand_base), ks.r2), sb = fe_to_mont<FS>(load_fe<FS>(sg_rand_base), ks.r2);
        for (uint32_t e = b + sh.pow_first; e; e >>= 1) {
            if (e & 1u) { rho = fe_mul<FS>(rho, rb); sigma = fe_mul<FS>(sigma, sb); }
            rb = fe_sqr<FS>(rb); sb = fe_sqr<FS>(sb);
        }
    }

    fe_t chal_m[20];                                          // k <= 20
    // challenges: absorb L_j, R_j ; squeeze 128 bits ; endo-expand.  Points go straight to the output list.
    // layout of the per-proof list: [h, sg, U, delta, L_0, R_0, ..., L_{k-1}, R_{k-1}, comm_0 .. comm_{m-1}]
    for (uint32_t j = 0; j < k; ++j) {
        const uint32_t *lp = lr + ((size_t)b * 2 * k + 2 * j) * 16;
        affine_t L = load_point_checked<FB>(lp, kb, pts_ok), R = load_point_checked<FB>(lp + 16, kb, pts_ok);
        sp.absorb(L.x); sp.absorb(L.y);                     // infinity is (0,0): absorbs two zeros, as upstream
        sp.absorb(R.x); sp.absorb(R.y);
        fe_t sq = fe_from_mont<FB>(sp.squeeze());
        uint64_t lo = (uint64_t)sq.v[0] | ((uint64_t)sq.v[1] << 32), hi = (uint64_t)sq.v[2] | ((uint64_t)sq.v[3] << 32);
        fe_t chal = challenge_to_field<FS>(lo, hi, ks);
        chal_m[j] = chal;                                     // Montgomery copy for the scalar work below
        store_fe<LANES>(out_chals + ((size_t)b * k + j) * 8, fe_from_mont<FS>(chal));
        store_pt<LANES>(&pts[4 + 2 * j], L); store_pt<LANES>(&pts[5 + 2 * j], R);
    }
    const affine_t D = load_point_checked<FB>(delta + (size_t)b * 16, kb, pts_ok);
    sp.absorb(D.x); sp.absorb(D.y);
    fe_t c;
    {
        fe_t sq = fe_from_mont<FB>(sp.squeeze());
        uint64_t lo = (uint64_t)sq.v[0] | ((uint64_t)sq.v[1] << 32), hi = (uint64_t)sq.v[2] | ((uint64_t)sq.v[3] << 32);
        c = challenge_to_field<FS>(lo, hi, ks);
    }

    // b0 = sum_p r^p * b_poly(chal, pt_p)
    const fe_t r = load_scalar_checked(evalscale + (size_t)b * 8);
    fe_t b0 = fe_zero(), scale = ks.one;
    for (uint32_t p = 0; p < sh.npts; ++p) {
        fe_t pw = load_scalar_checked(evalpoints + ((size_t)b * sh.npts + p) * 8), acc = ks.one;
        for (int i = (int)k - 1; i >= 0; --i) {
            acc = fe_mul<FS>(acc, fe_add<FS>(ks.one, fe_mul<FS>(chal_m[i], pw)));
            pw = fe_sqr<FS>(pw);
        }
        b0 = fe_add<FS>(b0, fe_mul<FS>(scale, acc));
        scale = fe_mul<FS>(scale, r);
    }

    const fe_t z1m = load_scalar_checked(z1 + (size_t)b * 8);
    const fe_t z2m = load_scalar_checked(z2 + (size_t)b * 8);
    const fe_t neg_rho = fe_neg<FS>(rho);
    const fe_t rho_c = fe_mul<FS>(rho, c);

    put(0, *srs_h, fe_from_mont<FS>(fe_mul<FS>(neg_rho, z2m)), sh.shared_h != 0);
    store_pt<LANES>(&pts[1], load_point_checked<FB>(sg + (size_t)b * 16, kb, pts_ok));
    store_fe<LANES>(scs + 1 * 8, fe_from_mont<FS>(fe_sub<FS>(fe_mul<FS>(neg_rho, z1m), sigma)));
    if (PHASE == 0) store_pt<LANES>(&pts[2], U);                // PHASE 2: written by ipa_to_group_kernel
    store_fe<LANES>(scs + 2 * 8, fe_from_mont<FS>(fe_add<FS>(fe_mul<FS>(fe_mul<FS>(neg_rho, z1m), b0), fe_mul<FS>(rho_c, cip_m))));
    store_pt<LANES>(&pts[3], D);                    store_fe<LANES>(scs + 3 * 8, fe_from_mont<FS>(rho));
    {   // chal^-1 for all rounds with ONE inversion (Montgomery's trick, as upstream's ark_ff::batch_inversion)
        fe_t pre[20]; fe_t run = ks.one;
        for (uint32_t j = 0; j < k; ++j) { pre[j] = run; run = fe_mul<FS>(run, chal_m[j]); }
        fe_t inv_run = fe_inv<FS>(run, ks);
        for (int j = (int)k - 1; j >= 0; --j) {
            const fe_t ch_inv = fe_mul<FS>(inv_run, pre[j]);
            inv_run = fe_mul<FS>(inv_run, chal_m[j]);
            store_fe<LANES>(scs + (size_t)(4 + 2 * j) * 8, fe_from_mont<FS>(fe_mul<FS>(rho_c, ch_inv)));
            store_fe<LANES>(scs + (size_t)(5 + 2 * j) * 8, fe_from_mont<FS>(fe_mul<FS>(rho_c, chal_m[j])));
        }
    }
    const fe_t xi = load_scalar_checked(polyscale + (size_t)b * 8);
    fe_t xi_i = ks.one;
    uint32_t o = 4 + 2 * k;
    for (uint32_t i = 0; i < sh.ncomms; ++i) {
        const fe_t wgt = fe_mul<FS>(rho_c, xi_i);
        if (i == sh.expand_slot) {                            // a linear combination in place of the point: the MSM evaluates it
            for (uint32_t j = 0; j < IPA_EXPAND; ++j, ++o) {
                put(o, j == 0 ? *ex.p0 : load_point_checked<FB>(ex.pts + ((size_t)b * (IPA_EXPAND - 1) + (j - 1)) * 16, kb, pts_ok),
                    fe_from_mont<FS>(fe_mul<FS>(wgt, ex.sc[(size_t)b * ex.stride + j])), j == 0 && sh.shared_expand0 != 0);
            }
        } else {
            const uint32_t *cp = (i == sh.override_slot && comm_override) ? comm_override + (size_t)b * 16 : comms + ((size_t)b * sh.ncomms + i) * 16;
            put(o, load_point_checked<FB>(cp, kb, pts_ok), fe_from_mont<FS>(wgt), i < 64 && (((i < 32 ? sh.shared_lo >> i : sh.shared_hi >> (i - 32)) & 1u) != 0));
            ++o;
        }
        xi_i = fe_mul<FS>(xi_i, xi);
    }
    store_fe<LANES>(out_sigma + (size_t)b * 8, fe_from_mont<FS>(sigma));
    if (!pts_ok && coop_writer<LANES>()) *bad_input = 1u;         // any malformed point in any proof: the batch verdict is 0
}

// Scalars of the batch-shared points (IpaShape::shared_*): their sum over the batch -> the `nsh` tail entries of the list (canonical words; a sum
// mod r does not care about the representation).  One block per shared entry.
template <int FS>
__global__ void __launch_bounds__(256)
ipa_shared_tail_kernel(uint32_t batch, uint32_t nsh, uint32_t per, const uint32_t *__restrict__ shared_sc, uint32_t *__restrict__ scalars) { mb_wave_prio();
    __shared__ fe_t red[256];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    fe_t acc = fe_zero();
    for (uint32_t b = tid; b < batch; b += 256) acc = fe_add<FS>(acc, load_fe<FS>(shared_sc + ((size_t)b * nsh + q) * 8));
    red[tid] = acc;
    __syncthreads();
    for (uint32_t s = 128; s; s >>= 1) { if (tid < s) red[tid] = fe_add<FS>(red[tid], red[tid + s]); __syncthreads(); }
    if (tid == 0) for (int i = 0; i < 8; ++i) scalars[((size_t)batch * per + q) * 8 + i] = red[0].v[i];
}
// the culprit search re-checks slices of the rows: give the proofs [lo, lo + cnt) their own scalars of the shared points back
__global__ void ipa_shared_restore_kernel(uint32_t lo, uint32_t cnt, uint32_t nsh, uint32_t per, const uint32_t *__restrict__ shared_off,
                                          const uint32_t *__restrict__ shared_sc, uint32_t *__restrict__ scalars) { mb_wave_prio();
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)cnt * nsh * 8) return;
    const uint32_t w = (uint32_t)(gid & 7u), q = (uint32_t)((gid >> 3) % nsh); const size_t b = lo + (gid >> 3) / nsh;
    scalars[(b * per + shared_off[q]) * 8 + w] = shared_sc[(b * nsh + q) * 8 + w];
}

// U_b = to_group(t_b) for the split transcript: one lane per proof, t in Montgomery form from the hand-over buffer
template <int FB>
__global__ void __launch_bounds__(64)
ipa_to_group_kernel(uint32_t batch, uint32_t per, FieldK kb, const uint32_t *__restrict__ xfer, affine_t *__restrict__ out_points) { mb_wave_prio();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const fe_t t = load_fe<FB>(xfer + (size_t)b * IPA_XFER_WORDS + 26);
    out_points[(size_t)b * per + 2] = bw_to_group<FB>(t, kb);
}

// verdict[0] = 1 iff  A + sign * B == identity  (sign = +1: A == -B ; sign = -1: A == B)
template <int F>
__global__ void xyzz_compare_kernel(const xyzz_t *__restrict__ a, const xyzz_t *__restrict__ b, int negate_b, uint32_t *__restrict__ verdict,
                                    const uint32_t *__restrict__ bad_input = nullptr) { mb_wave_prio();
    if (threadIdx.x || blockIdx.x) return;
    if (bad_input && *bad_input) { *verdict = 0u; return; }
    xyzz_t A = *a, B = *b;
    if (negate_b) B.y = fe_neg<F>(B.y);
    bool ai = xyzz_is_inf(A), bi = xyzz_is_inf(B), eq;
    if (ai || bi) eq = ai && bi;
    else eq = fe_eq(fe_mul<F>(A.x, B.zz), fe_mul<F>(B.x, A.zz)) && fe_eq(fe_mul<F>(A.y, B.zzz), fe_mul<F>(B.y, A.zzz));
    *verdict = eq ? 1u : 0u;
}

// sg points of a folded batch: canonical words -> Montgomery, with the deserialiser's checks (canonical, on the curve);
// any malformed point raises the batch's flag (the folded verdict is then 0 and the caller falls back to per-proof checks)
template <int F>
__global__ void points_to_mont_checked_kernel(uint32_t n, const uint32_t *__restrict__ in_words, FieldK kb, affine_t *__restrict__ out,
                                              uint32_t *__restrict__ bad_input) { mb_wave_prio();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool ok = true;
    out[i] = load_point_checked<F>(in_words + (size_t)i * 16, kb, ok);
    if (!ok) *bad_input = 1u;
}

// A (xyzz) == affine point q (canonical words; zeros = infinity) ?
template <int F>
__global__ void xyzz_eq_affine_kernel(const xyzz_t *__restrict__ a, const uint32_t *__restrict__ q_words, fe_t r2, uint32_t *__restrict__ verdict) {
    if (threadIdx.x) return;                                  // one block per comparison: a[b] vs q_words[16 b ..] -> verdict[b]
    a += blockIdx.x; q_words += (size_t)blockIdx.x * 16; verdict += blockIdx.x;
    xyzz_t A = *a;
    const fe_t qxw = load_fe<F>(q_words), qyw = load_fe<F>(q_words + 8);
    if (!fe_words_canonical<F>(qxw) || !fe_words_canonical<F>(qyw)) { *verdict = 0u; return; }   // alias encodings are rejected, as upstream's deserialiser does
    fe_t qx = fe_to_mont<F>(qxw, r2), qy = fe_to_mont<F>(qyw, r2);
    bool qi = fe_is_zero(qx) && fe_is_zero(qy), ai = xyzz_is_inf(A), eq;
    if (ai || qi) eq = ai && qi;
    else eq = fe_eq(fe_mul<F>(qx, A.zz), A.x) && fe_eq(fe_mul<F>(qy, A.zzz), A.y);
    *verdict = eq ? 1u : 0u;
}

}  // namespace mb

// ------------------------------------------------------------------------------------------------
// a10: accumulator check
// exchange variant: the shard's folded check is decided by the caller; here only "this shard held no malformed point"
__global__ void fold_export_flag_kernel(const uint32_t *__restrict__ malformed, uint32_t *__restrict__ verdict) { if (threadIdx.x == 0) verdict[0] = malformed[0] ? 0u : 1u; }

// out[i] = in[i] * lambda (canonical words in and out; lambda by value: no upload, no synchronisation)
struct Words8 { uint32_t v[8]; };
template <int F>
__global__ void scale_words_kernel(uint32_t n, FieldK fk, const uint32_t *__restrict__ in, Words8 lambda, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_t a, l;
    for (int j = 0; j < 8; ++j) { a.v[j] = in[(size_t)i * 8 + j]; l.v[j] = lambda.v[j]; }
    const fe_t r = fe_from_mont<F>(fe_mul<F>(fe_to_mont<F>(a, fk.r2), fe_to_mont<F>(l, fk.r2)));
    for (int j = 0; j < 8; ++j) out[(size_t)i * 8 + j] = r.v[j];
}

int mb_accumulator_check_dev(mina_ctx *c, int curve, uint32_t k, size_t batch, const uint32_t *d_prechal,
                                 const uint32_t *d_sg_words, const uint32_t *d_rho, uint32_t *d_verdict) {
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded for this curve");
    if (k < 1 || k > 20 || ((size_t)1 << k) > s.depth) return fail(MINA_ERR_ARG, "2^k exceeds the SRS depth");
    const int FS = scalar_field_of(curve), FB = base_field_of(curve);
    const uint32_t n = 1u << k;
    int rc;
    if (c->fold_export && batch > 1) {
        // The exchange variant ADDS the shards' partial sums, so no proof of any shard may carry a coefficient an adversary can predict: the opening leg draws its
        // own powers (pow_first), and until round 5 this leg trusted the caller's acc_rho (a caller following upstream's rho_0 = 1, or fixed test randomisers,
        // re-opened the +t / -t cancellation between shards).  Now every acc_rho[b] is multiplied by ONE scalar this call draws from the OS CSPRNG:
        // both sides of the shard's check scale by it, so a good shard stays good, and the cross-shard sum is a combination with G unknown coefficients.
        Words8 lam; uint8_t raw[32];
        if (!mb_secure_random(raw, 32)) return fail(MINA_ERR_STATE, "no entropy for the shard's accumulator randomiser");
        raw[31] &= 0x3f; raw[0] |= 1;                                  // < 2^254 < p, and not zero
        memcpy(lam.v, raw, 32);
        if ((rc = c->L->acc_rho_scaled.ensure(batch * 32))) return rc;
        DISPATCH_FIELD(FS, { scale_words_kernel<F_><<<cdiv(batch, 64), 64, 0, c->L->stream>>>((uint32_t)batch, c->fk[F_], d_rho, lam, c->L->acc_rho_scaled.as<uint32_t>()); });
        HIPC(hipGetLastError());
        d_rho = c->L->acc_rho_scaled.as<uint32_t>();
    }
    if ((rc = c->L->ipa_chals.ensure(batch * k * 32))) return rc;
    if ((rc = c->L->ipa_folded.ensure((size_t)n * 32))) return rc;
    if ((rc = c->L->ipa_xyzz_a.ensure(sizeof(xyzz_t)))) return rc;
    if ((rc = c->L->ipa_xyzz_b.ensure(sizeof(xyzz_t)))) return rc;
    if (batch == 1) {                                            // prechallenges -> coefficients in one launch
        if ((rc = mb_bpoly_single_from_prechallenges(c, FS, k, d_prechal, c->L->ipa_folded.as<uint32_t>(), 1))) return rc;
    } else {
        DISPATCH_FIELD(FS, { challenge_to_field_kernel<F_><<<cdiv(batch * k, 64), 64, 0, c->L->stream>>>((uint32_t)(batch * k), c->fk[F_], d_prechal, c->L->ipa_chals.as<uint32_t>()); });
        if ((rc = mb_bpoly_fold(c, FS, k, batch, c->L->ipa_chals.as<uint32_t>(), d_rho, c->L->ipa_folded.as<uint32_t>()))) return rc;
    }
    if (c->fold_export && batch > 1) {       // the exchange variant over several GPUs: the caller folds the shards' vectors and runs the MSM over its slice of the bases
        HIPC(hipMemcpyAsync(c->fold_export->acc_scalars, c->L->ipa_folded.p, (size_t)n * 32, hipMemcpyDeviceToDevice, c->L->stream));
        if ((rc = c->L->ipa_points.ensure(batch * sizeof(affine_t)))) return rc;
        if ((rc = c->L->ipa_sigma.ensure(4))) return rc;
        HIPC(hipMemsetAsync(c->L->ipa_sigma.p, 0, 4, c->L->stream));
        DISPATCH_FIELD(FB, { points_to_mont_checked_kernel<F_><<<cdiv(batch, 256), 256, 0, c->L->stream>>>((uint32_t)batch, d_sg_words, c->fk[F_], c->L->ipa_points.as<affine_t>(), c->L->ipa_sigma.as<uint32_t>()); });
        if ((rc = mb_msm_variable(c, curve, (uint32_t)batch, d_rho, c->L->ipa_points.p, c->fold_export->acc_point, nullptr))) return rc;
        fold_export_flag_kernel<<<1, 64, 0, c->L->stream>>>(c->L->ipa_sigma.as<uint32_t>(), d_verdict);      // verdict word = "no malformed commitment in this shard"
        HIPC(hipGetLastError());
        return MINA_OK;
    }
    if ((rc = mb_msm_fixed(c, curve, n, c->L->ipa_folded.as<uint32_t>(), nullptr, c->L->ipa_xyzz_a.p))) return rc;
    if (batch == 1) {
        DISPATCH_FIELD(FB, { xyzz_eq_affine_kernel<F_><<<1, 64, 0, c->L->stream>>>(c->L->ipa_xyzz_a.as<xyzz_t>(), d_sg_words, c->fk[F_].r2, d_verdict); });
    } else {
        if ((rc = c->L->ipa_points.ensure(batch * sizeof(affine_t)))) return rc;
        if ((rc = c->L->ipa_sigma.ensure(4))) return rc;            // malformed-input flag of this folded batch
        HIPC(hipMemsetAsync(c->L->ipa_sigma.p, 0, 4, c->L->stream));
        DISPATCH_FIELD(FB, { points_to_mont_checked_kernel<F_><<<cdiv(batch, 256), 256, 0, c->L->stream>>>((uint32_t)batch, d_sg_words, c->fk[F_], c->L->ipa_points.as<affine_t>(), c->L->ipa_sigma.as<uint32_t>()); });
        if ((rc = mb_msm_variable(c, curve, (uint32_t)batch, d_rho, c->L->ipa_points.p, nullptr, c->L->ipa_xyzz_b.p))) return rc;
        DISPATCH_FIELD(FB, { xyzz_compare_kernel<F_><<<1, 64, 0, c->L->stream>>>(c->L->ipa_xyzz_a.as<xyzz_t>(), c->L->ipa_xyzz_b.as<xyzz_t>(), 0, d_verdict, c->L->ipa_sigma.as<uint32_t>()); });
    }
    HIPC(hipGetLastError());
    return MINA_OK;
}

// `count` INDEPENDENT checks (no random folding: one verdict word each) through ONE kernel pipeline: the count MSMs are
// problems of the multi-problem pipeline, so the ~14 dependent dispatches of a check are paid once per group.
static int accumulator_check_multi_dev(mina_ctx *c, int curve, uint32_t k, size_t count, const uint32_t *d_prechal,
                                       const uint32_t *d_sg_words, uint32_t *d_verdicts) {
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded for this curve");
    if (k < 1 || k > 20 || ((size_t)1 << k) > s.depth) return fail(MINA_ERR_ARG, "2^k exceeds the SRS depth");
    const int FS = scalar_field_of(curve), FB = base_field_of(curve);
    const uint32_t n = 1u << k;
    int rc;
    if ((rc = c->L->ipa_folded.ensure(count * n * 32))) return rc;
    if ((rc = c->L->ipa_xyzz_a.ensure(count * sizeof(xyzz_t)))) return rc;
    if ((rc = mb_bpoly_single_from_prechallenges(c, FS, k, d_prechal, c->L->ipa_folded.as<uint32_t>(), (uint32_t)count))) return rc;
    if ((rc = mb_msm_table(c, curve, s.table.p, s.depth, s.c, s.W, 0, n, (uint32_t)count, c->L->ipa_folded.as<uint32_t>(), nullptr, c->L->ipa_xyzz_a.p))) return rc;
    DISPATCH_FIELD(FB, { xyzz_eq_affine_kernel<F_><<<(uint32_t)count, 64, 0, c->L->stream>>>(c->L->ipa_xyzz_a.as<xyzz_t>(), d_sg_words, c->fk[F_].r2, d_verdicts); });
    HIPC(hipGetLastError());
    return MINA_OK;
}

extern "C" int mina_accumulator_check_multi_dev(mina_ctx *c, int curve, uint32_t k, size_t count, const void *d_prechallenges,
                                                const void *d_sg, void *d_verdicts) {
    if (!c || !d_prechallenges || !d_sg || !d_verdicts) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (count == 0 || count > 64) return fail(MINA_ERR_ARG, "count must be in 1..64");
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    return accumulator_check_multi_dev(c, curve, k, count, (const uint32_t *)d_prechallenges, (const uint32_t *)d_sg, (uint32_t *)d_verdicts);
}

extern "C" int mina_accumulator_check_dev(mina_ctx *c, int curve, uint32_t k, size_t batch, const void *d_prechallenges,
                                          const void *d_sg, const void *d_rho, void *d_verdict) {
    if (!c || !d_prechallenges || !d_sg || !d_verdict || (batch > 1 && !d_rho)) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (batch == 0 || batch > (1u << 20)) return fail(MINA_ERR_ARG, "bad batch");
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    return mb_accumulator_check_dev(c, curve, k, batch, (const uint32_t *)d_prechallenges, (const uint32_t *)d_sg, (const uint32_t *)d_rho, (uint32_t *)d_verdict);
}

// per-proof verdicts for the `count` proofs already staged in ipa_in_a (prechallenges) / ipa_in_b (sg) of the current lane
static int accumulator_check_each(mina_ctx *c, int curve, uint32_t k, size_t count, uint8_t *verdicts) {
    constexpr size_t GROUP = 16;
    int rc;
    if ((rc = c->L->ipa_verdict.ensure(GROUP * 4))) return rc;
    for (size_t b = 0; b < count; b += GROUP) {
        const size_t g = count - b < GROUP ? count - b : GROUP;
        if ((rc = accumulator_check_multi_dev(c, curve, k, g, c->L->ipa_in_a.as<uint32_t>() + b * k * 4, c->L->ipa_in_b.as<uint32_t>() + b * 16,
                                              c->L->ipa_verdict.as<uint32_t>()))) return rc;
        uint32_t v[GROUP];
        if ((rc = d2h_sync(c, v, c->L->ipa_verdict, g * 4))) return rc;
        for (size_t i = 0; i < g; ++i) verdicts[b + i] = v[i] ? 1 : 0;
    }
    return MINA_OK;
}

extern "C" int mina_accumulator_check_batch(mina_ctx *c, int curve, uint32_t k, size_t batch, const uint8_t *prechallenges,
                                            const uint8_t *sg, const uint8_t *rho, uint8_t *verdicts) {
    if (!c || !prechallenges || !sg || !verdicts || (batch > 1 && !rho)) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (batch == 0 || batch > (1u << 20)) return fail(MINA_ERR_ARG, "bad batch");
    if (c->srs[curve].depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded for this curve");
    if (k < 1 || k > 20 || ((size_t)1 << k) > c->srs[curve].depth) return fail(MINA_ERR_ARG, "k must be in 1..20 with 2^k <= SRS depth");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = h2d(c, c->L->ipa_in_a, prechallenges, batch * k * 16))) return rc;
    if ((rc = h2d(c, c->L->ipa_in_b, sg, batch * 64))) return rc;
    if (batch > 1 && (rc = h2d(c, c->L->ipa_in_c, rho, batch * 32))) return rc;
    if ((rc = c->L->ipa_verdict.ensure(64))) return rc;
    if ((rc = mb_accumulator_check_dev(c, curve, k, batch, c->L->ipa_in_a.as<uint32_t>(), c->L->ipa_in_b.as<uint32_t>(),
                                    batch > 1 ? c->L->ipa_in_c.as<uint32_t>() : nullptr, c->L->ipa_verdict.as<uint32_t>()))) return rc;
    uint32_t v = 0;
    if ((rc = d2h_sync(c, &v, c->L->ipa_verdict, 4))) return rc;
    if (v || batch == 1) { memset(verdicts, v ? 1 : 0, batch); return MINA_OK; }
    // the folded check failed: every proof on its own (rare path), groups of independent checks per pipeline
    return accumulator_check_each(c, curve, k, batch, verdicts);
}

// host-buffer form of the un-folded group check: per-proof verdicts, no randomness involved
extern "C" int mina_accumulator_check_multi(mina_ctx *c, int curve, uint32_t k, size_t count, const uint8_t *prechallenges,
                                            const uint8_t *sg, uint8_t *verdicts) {
    if (!c || !prechallenges || !sg || !verdicts) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (count == 0 || count > (1u << 20)) return fail(MINA_ERR_ARG, "bad count");
    if (c->srs[curve].depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded for this curve");
    if (k < 1 || k > 20 || ((size_t)1 << k) > c->srs[curve].depth) return fail(MINA_ERR_ARG, "k must be in 1..20 with 2^k <= SRS depth");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = h2d(c, c->L->ipa_in_a, prechallenges, count * k * 16))) return rc;
    if ((rc = h2d(c, c->L->ipa_in_b, sg, count * 64))) return rc;
    return accumulator_check_each(c, curve, k, count, verdicts);
}

// ------------------------------------------------------------------------------------------------
// a8: combined opening check, inputs resident in HBM (structure-of-arrays, see IpaDevIn).  Queued on the current lane, no host
// synchronisation: d_verdict[0] = 1 iff the folded check holds, d_verdict[1] = malformed-input flag.
int mb_ipa_batch_check_dev(mina_ctx *c, int curve, mb::IpaShape sh, const mb::IpaDevIn &in, uint32_t *d_verdict) {
    SrsState &s = c->srs[curve];
    const int FB = base_field_of(curve), FS = scalar_field_of(curve);
    const size_t batch = sh.batch; const uint32_t k = sh.k;
    int rc;
    if (c->fold_export) sh.pow_first = 1;          // partial sums that the caller adds to other shards': no coefficient-1 proof (ctx.h IpaShape::pow_first)
    if (sh.nshared && (sh.ncomms > 64 || sh.nshared > 64)) return fail(MINA_ERR_ARG, "shared entries need <= 64 commitments");
    const size_t npoints = batch * sh.per + sh.nshared;           // per-proof lists, then the batch-shared points once
    if ((rc = c->L->ipa_points.ensure(npoints * sizeof(affine_t)))) return rc;
    if ((rc = c->L->ipa_scalars.ensure(npoints * 32))) return rc;
    if ((rc = c->L->ipa_chals.ensure(batch * k * 32))) return rc;
    if ((rc = c->L->ipa_sigma.ensure(batch * 32))) return rc;
    if ((rc = c->L->ipa_folded.ensure(((size_t)1 << k) * 32))) return rc;
    if ((rc = c->L->ipa_xyzz_a.ensure(sizeof(xyzz_t)))) return rc;
    if ((rc = c->L->ipa_xyzz_b.ensure(sizeof(xyzz_t)))) return rc;
    if (sh.nshared) { if ((rc = c->L->ipa_shared.ensure(batch * sh.nshared * 32))) return rc; if ((rc = c->L->ipa_shared_off.ensure(sh.nshared * 4))) return rc; }
    HIPC(hipMemsetAsync(d_verdict, 0, 8, c->L->stream));
    const PoseidonParams *pp = c->pparams[FB].as<PoseidonParams>();
#define IPA_PREP(CV, LN, PH, STREAM)                                                                                          \
    mb::ipa_prepare_kernel<CV, LN, PH><<<cdiv(coop_threads<LN>(batch), 64), 64, 0, STREAM>>>(                                                      \
        sh, c->fk[FB], c->fk[FS], pp, in.state, in.pos, in.cip, in.lr, in.delta, in.sg, in.z1, in.z2, in.pts, in.r, \
        in.xi, in.comms, in.comm_override, in.expand, in.rb, in.sb, s.h.as<affine_t>(), c->L->ipa_points.as<affine_t>(), c->L->ipa_scalars.as<uint32_t>(), \
        c->L->ipa_chals.as<uint32_t>(), c->L->ipa_sigma.as<uint32_t>(), d_verdict + 1, c->L->ipa_xfer.as<uint32_t>(),          \
        c->L->ipa_shared.as<uint32_t>(), c->L->ipa_shared_off.as<uint32_t>())
    { ProfScope ps_(c, PS_IPA_TRANSCRIPT);
    {
        // the transcript splits at its first squeeze: to_group (one lane per proof) runs on a second stream beside the rest.  8 lanes per
        // transcript up to 1024 proofs per call (shortest dependent chain); above that the 3-lane form: 21 transcripts per wave, 3/8 of the
        // issue slots -- measured with 16 calls of 8192 proofs in flight (bench.py), where the VALU port is what saturates
        const mina_verify_tuning tune = mb_tune();
        const bool oct = use_coop8_transcripts(c, batch, (size_t)tune.ipa_coop8_max);
        Lane &L = *c->L;
        if ((rc = L.ipa_xfer.ensure(batch * mb::IPA_XFER_WORDS * 4))) return rc;
        // second stream only for a context that runs ONE call at a time: with pipeline lanes in flight the other lanes fill the chip, and a
        // side stream per lane would make 2 x lanes streams share the 16 hardware queues (lanes then serialise behind each other's kernels)
        const bool side = c->nlanes == 1 && tune.ipa_side_stream != 0 && !c->is_view;      // (a view context creates no streams: ctx.h)
        hipStream_t tg = L.stream;
        if (side) {
            if (!L.aux) { HIPC(hipStreamCreateWithFlags(&L.aux, hipStreamNonBlocking)); HIPC(hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming)); HIPC(hipEventCreateWithFlags(&L.ev_join, hipEventDisableTiming)); }
            tg = L.aux;
        }
        const bool hex = use_coop16(c, batch);
        if (hex) { if (curve == CURVE_PALLAS) IPA_PREP(CURVE_PALLAS, 16, 1, L.stream); else IPA_PREP(CURVE_VESTA, 16, 1, L.stream); }
        else if (oct) { if (curve == CURVE_PALLAS) IPA_PREP(CURVE_PALLAS, 8, 1, L.stream); else IPA_PREP(CURVE_VESTA, 8, 1, L.stream); }
        else { if (curve == CURVE_PALLAS) IPA_PREP(CURVE_PALLAS, 3, 1, L.stream); else IPA_PREP(CURVE_VESTA, 3, 1, L.stream); }
        if (side) { HIPC(hipEventRecord(L.ev_fork, L.stream)); HIPC(hipStreamWaitEvent(L.aux, L.ev_fork, 0)); }
        DISPATCH_FIELD(FB, { mb::ipa_to_group_kernel<F_><<<cdiv(batch, 64), 64, 0, tg>>>((uint32_t)batch, sh.per, c->fk[F_], L.ipa_xfer.as<uint32_t>(), L.ipa_points.as<affine_t>()); });
        if (side) HIPC(hipEventRecord(L.ev_join, L.aux));
        if (hex) { if (curve == CURVE_PALLAS) IPA_PREP(CURVE_PALLAS, 16, 2, L.stream); else IPA_PREP(CURVE_VESTA, 16, 2, L.stream); }
        else if (oct) { if (curve == CURVE_PALLAS) IPA_PREP(CURVE_PALLAS, 8, 2, L.stream); else IPA_PREP(CURVE_VESTA, 8, 2, L.stream); }
        else { if (curve == CURVE_PALLAS) IPA_PREP(CURVE_PALLAS, 3, 2, L.stream); else IPA_PREP(CURVE_VESTA, 3, 2, L.stream); }
        if (side) HIPC(hipStreamWaitEvent(L.stream, L.ev_join, 0));
    }
    }
#undef IPA_PREP
    HIPC(hipGetLastError());
    if (sh.nshared) { DISPATCH_FIELD(FS, { mb::ipa_shared_tail_kernel<F_><<<sh.nshared, 256, 0, c->L->stream>>>(sh.batch, sh.nshared, sh.per, c->L->ipa_shared.as<uint32_t>(), c->L->ipa_scalars.as<uint32_t>()); }); }
    if ((rc = mb_bpoly_fold(c, FS, k, batch, c->L->ipa_chals.as<uint32_t>(), c->L->ipa_sigma.as<uint32_t>(), c->L->ipa_folded.as<uint32_t>()))) return rc;
    if (c->fold_export) {                        // the exchange variant over several GPUs: scalars and the variable-base partial go to the caller
        HIPC(hipMemcpyAsync(c->fold_export->ipa_scalars, c->L->ipa_folded.p, ((size_t)1 << k) * 32, hipMemcpyDeviceToDevice, c->L->stream));
        if ((rc = mb_msm_variable(c, curve, (uint32_t)npoints, c->L->ipa_scalars.as<uint32_t>(), c->L->ipa_points.p, c->fold_export->ipa_point, nullptr))) return rc;
        fold_export_flag_kernel<<<1, 64, 0, c->L->stream>>>(d_verdict + 1, d_verdict);     // [1] = malformed flag of the transcripts -> [0] = 1 unless malformed
        HIPC(hipGetLastError());
        return MINA_OK;
    }
    if ((rc = mb_msm_fixed(c, curve, 1u << k, c->L->ipa_folded.as<uint32_t>(), nullptr, c->L->ipa_xyzz_a.p))) return rc;
    if ((rc = mb_msm_variable(c, curve, (uint32_t)npoints, c->L->ipa_scalars.as<uint32_t>(), c->L->ipa_points.p, nullptr, c->L->ipa_xyzz_b.p))) return rc;
    DISPATCH_FIELD(FB, { xyzz_compare_kernel<F_><<<1, 64, 0, c->L->stream>>>(c->L->ipa_xyzz_a.as<xyzz_t>(), c->L->ipa_xyzz_b.as<xyzz_t>(), 1, d_verdict, d_verdict + 1); });
    HIPC(hipGetLastError());
    c->ipa_rows = c->L; c->ipa_rows_batch = sh.batch; c->ipa_rows_k = k; c->ipa_rows_per = sh.per; c->ipa_rows_nshared = sh.nshared; c->ipa_rows_curve = curve;
    return MINA_OK;
}

// The rows ipa_prepare_kernel left on a lane (per proof: `per` (point, scalar) pairs, the k challenges and the fold weight) already carry the
// batch's randomisers rho^b, sigma^b, so the rows of any subset of proofs ARE a folded check of that subset: fold + the two MSMs + the
// comparison, no transcript work.  mina_state_job_batch's search for the culprits of a failed batch runs this on slices, one lane each.
// The caller has synchronised the lane that holds the rows.
int mb_ipa_recheck_rows(mina_ctx *c, size_t lo, size_t cnt, uint32_t *d_verdict) {
    Lane *src = c->ipa_rows;
    if (!src || cnt == 0 || lo + cnt > c->ipa_rows_batch) return fail(MINA_ERR_STATE, "no prepared rows for that range");
    const int curve = c->ipa_rows_curve, FB = base_field_of(curve), FS = scalar_field_of(curve);
    const uint32_t k = c->ipa_rows_k, per = c->ipa_rows_per;
    Lane &L = *c->L;
    int rc;
    if ((rc = L.ipa_folded.ensure(((size_t)1 << k) * 32))) return rc;
    if ((rc = L.ipa_xyzz_a.ensure(sizeof(xyzz_t)))) return rc;
    if ((rc = L.ipa_xyzz_b.ensure(sizeof(xyzz_t)))) return rc;
    HIPC(hipMemsetAsync(d_verdict, 0, 8, L.stream));
    if (const uint32_t nsh = c->ipa_rows_nshared)              // the slice's proofs get their own scalars of the batch-shared points back (idempotent)
        mb::ipa_shared_restore_kernel<<<cdiv(cnt * nsh * 8, 256), 256, 0, L.stream>>>((uint32_t)lo, (uint32_t)cnt, nsh, per, src->ipa_shared_off.as<uint32_t>(), src->ipa_shared.as<uint32_t>(), src->ipa_scalars.as<uint32_t>());
    if ((rc = mb_bpoly_fold(c, FS, k, cnt, src->ipa_chals.as<uint32_t>() + lo * k * 8, src->ipa_sigma.as<uint32_t>() + lo * 8, L.ipa_folded.as<uint32_t>()))) return rc;
    if ((rc = mb_msm_fixed(c, curve, 1u << k, L.ipa_folded.as<uint32_t>(), nullptr, L.ipa_xyzz_a.p))) return rc;
    if ((rc = mb_msm_variable(c, curve, (uint32_t)(cnt * per), src->ipa_scalars.as<uint32_t>() + lo * per * 8, src->ipa_points.as<affine_t>() + lo * per, nullptr, L.ipa_xyzz_b.p))) return rc;
    DISPATCH_FIELD(FB, { xyzz_compare_kernel<F_><<<1, 64, 0, L.stream>>>(L.ipa_xyzz_a.as<xyzz_t>(), L.ipa_xyzz_b.as<xyzz_t>(), 1, d_verdict, d_verdict + 1); });
    HIPC(hipGetLastError());
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------
// a8: combined opening check
extern "C" int mina_ipa_batch_check(mina_ctx *c, int curve, size_t batch, const mina_ipa_opening *op, const uint8_t *rand_base,
                                    const uint8_t *sg_rand_base, uint8_t *verdict) {
    if (!c || !op || !rand_base || !sg_rand_base || !verdict) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (batch == 0 || batch > (1u << 16)) return fail(MINA_ERR_ARG, "bad batch");
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded for this curve");
    const int FB = base_field_of(curve), FS = scalar_field_of(curve);
    if (!c->have_pparams[FB]) return fail(MINA_ERR_STATE, "Poseidon constants not installed for the base field");
    const uint32_t k = op[0].k, npts = op[0].n_evalpoints;
    uint32_t m = 0;                                             // commitments per opening in the device layout = the maximum;
    for (size_t b = 0; b < batch; ++b) if (op[b].n_comms > m) m = op[b].n_comms;   // shorter lists are padded with infinity
    if (m > 4096) return fail(MINA_ERR_ARG, "too many commitments in one opening");
    if (k < 1 || k > 20 || ((size_t)1 << k) > s.depth) return fail(MINA_ERR_ARG, "2^k exceeds the SRS depth");
    for (size_t b = 0; b < batch; ++b) {
        const mina_ipa_opening &o = op[b];
        if (o.k != k || o.n_evalpoints != npts) return fail(MINA_ERR_ARG, "openings of one batch must share k and n_evalpoints");
        if (!o.lr || !o.delta || !o.sg || !o.z1 || !o.z2 || !o.combined_inner_product || !o.polyscale || !o.evalscale || !o.sponge_state ||
            (npts && !o.evalpoints) || (o.n_comms && !o.comms)) return fail(MINA_ERR_ARG, "null field in opening");
    }
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    mb::IpaShape sh; sh.batch = (uint32_t)batch; sh.k = k; sh.npts = npts; sh.ncomms = m; sh.per = 2 * k + m + 4;

    // pack (host) -> one staging blob -> HBM
    const size_t o_state = 0, o_cip = o_state + batch * 96, o_lr = o_cip + batch * 32, o_delta = o_lr + batch * 2 * k * 64,
                 o_sg = o_delta + batch * 64, o_z1 = o_sg + batch * 64, o_z2 = o_z1 + batch * 32, o_pts = o_z2 + batch * 32,
                 o_r = o_pts + batch * npts * 32, o_xi = o_r + batch * 32, o_comms = o_xi + batch * 32, o_rb = o_comms + batch * m * 64,
                 o_sb = o_rb + 32, o_pos = o_sb + 32, total = o_pos + batch * 8;
    int rc;
    if ((rc = c->L->host_stage.ensure(total))) return rc;       // packed straight into page-locked memory
    uint8_t *blob = (uint8_t *)c->L->host_stage.p;
    memset(blob + o_comms, 0, batch * m * 64);                   // shorter commitment lists are padded with infinity = (0, 0)
    for (size_t b = 0; b < batch; ++b) {
        const mina_ipa_opening &o = op[b];
        memcpy(&blob[o_state + b * 96], o.sponge_state, 96);
        memcpy(&blob[o_cip + b * 32], o.combined_inner_product, 32);
        { uint32_t pos[2] = {o.sponge_mode, o.sponge_count}; memcpy(&blob[o_pos + b * 8], pos, 8); }
        if (o.sponge_mode > 1 || o.sponge_count > 2) return fail(MINA_ERR_ARG, "bad sponge position");
        memcpy(&blob[o_lr + b * 2 * k * 64], o.lr, (size_t)2 * k * 64);
        memcpy(&blob[o_delta + b * 64], o.delta, 64);
        memcpy(&blob[o_sg + b * 64], o.sg, 64);
        memcpy(&blob[o_z1 + b * 32], o.z1, 32);
        memcpy(&blob[o_z2 + b * 32], o.z2, 32);
        if (npts) memcpy(&blob[o_pts + b * npts * 32], o.evalpoints, (size_t)npts * 32);
        memcpy(&blob[o_r + b * 32], o.evalscale, 32);
        memcpy(&blob[o_xi + b * 32], o.polyscale, 32);
        if (o.n_comms) memcpy(&blob[o_comms + b * m * 64], o.comms, (size_t)o.n_comms * 64);   // rest stays (0, 0) = infinity: contributes nothing
    }
    memcpy(&blob[o_rb], rand_base, 32);
    memcpy(&blob[o_sb], sg_rand_base, 32);
    if ((rc = h2d(c, c->L->ipa_in_a, blob, total))) return rc;
    const uint8_t *d = c->L->ipa_in_a.as<uint8_t>();
    auto W = [&](size_t off) { return reinterpret_cast<const uint32_t *>(d + off); };
    mb::IpaDevIn in{W(o_state), W(o_pos), W(o_cip), W(o_lr), W(o_delta), W(o_sg), W(o_z1), W(o_z2), W(o_pts), W(o_r), W(o_xi), W(o_comms), nullptr, W(o_rb), W(o_sb)};
    if ((rc = c->L->ipa_verdict.ensure(8))) return rc;          // [0] verdict, [1] malformed-input flag
    if ((rc = mb_ipa_batch_check_dev(c, curve, sh, in, c->L->ipa_verdict.as<uint32_t>()))) return rc;
    uint32_t v = 0;
    if ((rc = d2h_sync(c, &v, c->L->ipa_verdict, 4))) return rc;
    *verdict = v ? 1 : 0;
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------
// a12: batched Fq-sponge transcripts
extern "C" int mina_fq_sponge_run(mina_ctx *c, int curve, size_t batch, const uint8_t *tape, size_t tape_len, const uint8_t *init_state,
                                  const uint32_t *init_pos, const uint8_t *inputs, uint8_t *outputs, uint8_t *final_state, uint32_t *final_pos) {
    if (!c || !tape || (batch && tape_len && !outputs && !final_state)) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if ((init_state == nullptr) != (init_pos == nullptr)) return fail(MINA_ERR_ARG, "init_state and init_pos go together");
    if (batch == 0) return MINA_OK;
    if (batch > (1u << 22) || tape_len > (1u << 16)) return fail(MINA_ERR_ARG, "batch or tape too large");
    const int FB = base_field_of(curve);
    if (!c->have_pparams[FB]) return fail(MINA_ERR_STATE, "Poseidon constants not installed for the base field");
    size_t in_words = 0, out_words = 0;
    for (size_t t = 0; t < tape_len; ++t) {
        switch (tape[t]) {
            case TAPE_ABSORB_FQ: case TAPE_ABSORB_FR: in_words += 8; break;
            case TAPE_ABSORB_G: in_words += 16; break;
            case TAPE_CHALLENGE: case TAPE_CHALLENGE_FQ: case TAPE_CHALLENGE_ENDO: case TAPE_DIGEST: case TAPE_CHALLENGE_ENDO_OWN: out_words += 8; break;
            default: return fail(MINA_ERR_ARG, "unknown tape opcode");
        }
    }
    if ((in_words && !inputs) || (out_words && !outputs)) return fail(MINA_ERR_ARG, "null argument");
    if (init_pos) for (size_t b = 0; b < batch; ++b) if (init_pos[2 * b] > 1 || init_pos[2 * b + 1] > 2) return fail(MINA_ERR_ARG, "bad sponge position");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    Lane &L = *c->L;
    if ((rc = h2d(c, L.ipa_in_a, tape, tape_len))) return rc;
    if ((rc = h2d(c, L.ipa_in_b, inputs, batch * in_words * 4))) return rc;
    if (init_state) { if ((rc = h2d(c, L.ipa_in_c, init_state, batch * 96))) return rc; if ((rc = h2d(c, L.ipa_sigma, init_pos, batch * 8))) return rc; }
    if ((rc = L.ipa_scalars.ensure(batch * (out_words ? out_words : 8) * 4))) return rc;
    if ((rc = L.ipa_points.ensure(batch * 96))) return rc;
    if ((rc = L.ipa_chals.ensure(batch * 8))) return rc;
    const PoseidonParams *pp = c->pparams[FB].as<PoseidonParams>();
    const int FS = scalar_field_of(curve);
#define RUN_TAPE(CV, LN)                                                                                                                   \
    mb::sponge_tape_kernel<CV, LN><<<cdiv(coop_threads<LN>(batch), 64), 64, 0, L.stream>>>((uint32_t)batch, (uint32_t)tape_len, (uint32_t)in_words, (uint32_t)out_words, \
        c->fk[FB], c->fk[FS], pp, L.ipa_in_a.as<uint8_t>(), init_state ? L.ipa_in_c.as<uint32_t>() : nullptr, init_state ? L.ipa_sigma.as<uint32_t>() : nullptr, \
        L.ipa_in_b.as<uint32_t>(), L.ipa_scalars.as<uint32_t>(), L.ipa_points.as<uint32_t>(), L.ipa_chals.as<uint32_t>())
    if (batch <= COOP8_MAX_GROUPS) { if (curve == CURVE_PALLAS) RUN_TAPE(CURVE_PALLAS, 8); else RUN_TAPE(CURVE_VESTA, 8); }
    else { if (curve == CURVE_PALLAS) RUN_TAPE(CURVE_PALLAS, 3); else RUN_TAPE(CURVE_VESTA, 3); }
#undef RUN_TAPE
    HIPC(hipGetLastError());
    if (final_state) HIPC(hipMemcpyAsync(final_state, L.ipa_points.p, batch * 96, hipMemcpyDeviceToHost, L.stream));
    if (final_pos) HIPC(hipMemcpyAsync(final_pos, L.ipa_chals.p, batch * 8, hipMemcpyDeviceToHost, L.stream));
    return d2h_sync(c, outputs, L.ipa_scalars, batch * out_words * 4);
}
