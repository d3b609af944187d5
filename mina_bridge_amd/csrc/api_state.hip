// api_state.hip -- the Proof-of-State job behind `verify_mina_state` (8f-1; BASELINE config C3).
//
// What the reference's verifier does per state proof (README.md:281-310; Aligned `verify_mina_state_ffi`, un-vendored):
//   1. hash the 16 candidate-chain protocol states and the bridge tip state (`MinaHash`: Poseidon over `to_input`, twice),
//      compare with the public inputs, check that the states form a chain                       -> pstate_hash_kernel + chain check
//   2. Pickles wrap verification of the tip proof:
//        public-input commitment (Lagrange MSM, Pallas)                                         -> mb_msm_table + pubcomm finish
//        Fiat-Shamir + combined IPA opening (k = 15)                                            -> mb_ipa_batch_check_dev
//        step accumulator check (Vesta, 2^16 bases)                                             -> mb_accumulator_check_dev
// All of it is queued on ONE lane with no host synchronisation between the stages; one verdict word per proof.
#include <chrono>
#include "ctx.h"
#include "msm.cuh"
#include "sponge.cuh"
#include "lagrange.cuh"
#include "wire_state.h"

namespace mb {

template <int F> __device__ __forceinline__ fe_t ld_fe(const uint32_t *p) { fe_t r; for (int i = 0; i < 8; ++i) r.v[i] = p[i]; return r; }

// `MinaHash(ProtocolState)`: body = H_{"MinaProtoStateBody"}(fields[1 .. 1+nf)); hash = H_{"MinaProtoState"}(fields[0], body).
// One lane group (8 lanes, or a wave-packed triple for chip-filling batches) per state; record = MINA_PSTATE_SLOTS field elements, canonical words.
template <int F, int LANES>
// LANES == 3: five waves per SIMD (96 VGPRs) as before the signed-digit forms, whose digits pin the low halves of register pairs (100 VGPRs unasked): the three values
// the allocator parks in scratch are touched outside the round loops only (pinned from the code object: tests/test_code_object.py::test_dominant_kernel_round_loops)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LANES == 3 ? 5 : 1, LANES == 3 ? 5 : 8)))
pstate_hash_kernel(uint32_t n, FieldK fk, const PoseidonParams *__restrict__ pp, const fe_t *__restrict__ salts /* [0..3) body, [3..6) state */,
                   const uint32_t *__restrict__ records, const uint32_t *__restrict__ nfields, uint32_t *__restrict__ out_hash /* n*8 */,
                   uint32_t *__restrict__ out_body /* n*8 or null */) {
    bool writer;
    const uint32_t sp = coop_sponge_index<LANES>(writer), e = coop_elem<LANES>();
    const bool live = sp < n;
    const uint32_t idx = live ? sp : 0;                            // dead groups shadow state 0 (whole waves run the cross-lane moves)
    const uint32_t *rec = records + (size_t)idx * MINA_PSTATE_SLOTS * 8;
    uint32_t nf = nfields[idx]; if (nf > MINA_PSTATE_SLOTS - 1) nf = MINA_PSTATE_SLOTS - 1;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (LANES == 3) {
        // The chip-filling form keeps the state in the 29-bit form (x 2^261, lazily reduced) from the first absorb to the last squeeze: a field enters by ONE signed-digit
        // product with 2^522 mod p -- both fields of a block at once, each on the lane that owns its state element; a lane with nothing to absorb multiplies zero, which
        // gives a multiple of p -- instead of a Montgomery conversion per field (two divergent ones per permutation) and a product into and out of the 29-bit form around
        // every permutation: ~ 600 of the ~ 51 500 instructions of a permutation.  Bounds: tools/fe29_bounds.py prove_sponge_rounds (row + absorbed field < 4.1 p).
        const TriPos tp = tri_pos();
        const PoseidonParams29 *__restrict__ q = pparams29_of(pp);
        auto field29 = [&](const uint32_t *w, bool take) {           // (words)(2^522) / 2^261; nothing to take: a multiple of p
            fe_t v = fe_zero();
            if (take) v = ld_fe<F>(w);
            return fe29_mul_sg<F>(fe29_from_words(v), q->absorb);
        };
        fe29_t x = fe29_mul_asm<F>(fe29_from_words(salts[e]), q->enter);
        const uint32_t nblk = (nf + 1) / 2;
#pragma unroll 1
        for (uint32_t k = 0; k < nblk; ++k) {
            if (k) poseidon_rounds_tri<F>(x, q, tp);
            const uint32_t el = 2 * k + e;
            x = fe29_add(x, field29(rec + (size_t)(1 + el) * 8, e < 2 && el < nf));
        }
        poseidon_rounds_tri<F>(x, q, tp);
        const fe29_t body29 = tri_bcast29(x, tp.base);               // state element 0, still x 2^261, below 2.07 p
        fe29_t y = fe29_mul_asm<F>(fe29_from_words(salts[3 + e]), q->enter);
        fe29_t add = field29(rec, e == 0);
        if (e == 1) add = body29;
        y = fe29_add(y, add);
        poseidon_rounds_tri<F>(y, q, tp);
        if (live) {                                                  // out of the 29-bit form once per state (element 0 lives on the writer lane)
            if (writer) {
                const fe_t w = fe_from_mont<F>(fe_cond_sub_p<F>(fe29_to_words(fe29_mul_asm<F>(y, q->leave)))); for (int i = 0; i < 8; ++i) out_hash[(size_t)sp * 8 + i] = w.v[i];
                if (out_body) { const fe_t bw = fe_from_mont<F>(fe_cond_sub_p<F>(fe29_to_words(fe29_mul_asm<F>(x, q->leave)))); for (int i = 0; i < 8; ++i) out_body[(size_t)sp * 8 + i] = bw.v[i]; }
            }
        }
        return;
    }
#endif
    fe_t s = salts[e];
    uint32_t count = 0;
    for (uint32_t el = 0; el < nf; ++el) {
        if (count == 2) { poseidon_permute_coop<F, LANES>(s, pp); count = 0; }
        if (e == count) s = fe_add<F>(s, fe_to_mont<F>(ld_fe<F>(rec + (size_t)(1 + el) * 8), fk.r2));
        ++count;
    }
    poseidon_permute_coop<F, LANES>(s, pp);
    const fe_t body = coop_get<LANES>(s, 0);
    s = salts[3 + e];
    if (e == 0) s = fe_add<F>(s, fe_to_mont<F>(ld_fe<F>(rec), fk.r2));
    if (e == 1) s = fe_add<F>(s, body);
    poseidon_permute_coop<F, LANES>(s, pp);
    s = coop_get<LANES>(s, 0);
    if (live && writer) {
        const fe_t w = fe_from_mont<F>(s); for (int i = 0; i < 8; ++i) out_hash[(size_t)sp * 8 + i] = w.v[i];
        if (out_body) { const fe_t bw = fe_from_mont<F>(body); for (int i = 0; i < 8; ++i) out_body[(size_t)sp * 8 + i] = bw.v[i]; }
    }
}

// salts of the hash prefixes: state after absorbing the prefix element into the zero state and permuting (3 elements each)
template <int F>
__global__ void prefix_salt_kernel(uint32_t n, FieldK fk, const PoseidonParams *__restrict__ pp, const uint32_t *__restrict__ prefixes, fe_t *__restrict__ salts) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n) return;
    fe_t s[3] = {fe_to_mont<F>(ld_fe<F>(prefixes + (size_t)h * 8), fk.r2), fe_zero(), fe_zero()};
    poseidon_permute<F>(s, pp);
    for (int j = 0; j < 3; ++j) salts[(size_t)h * 3 + j] = s[j];
}

// README.md:283-288 per proof: hashes of the 16 chain states and of the bridge tip state equal the public inputs, and
// state i+1 names state i as its predecessor.  One lane per proof (pure comparisons).
__global__ void pstate_chain_check_kernel(uint32_t batch, const uint32_t *__restrict__ hashes /* b*17*8 */, const uint32_t *__restrict__ expected /* b*17*8 */,
                                          const uint32_t *__restrict__ records, const uint8_t *__restrict__ precheck /* b or null */, uint32_t *__restrict__ ok) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    bool good = precheck ? precheck[b] != 0 : true;
    const uint32_t *h = hashes + (size_t)b * MINA_STATES_PER_PROOF * 8, *x = expected + (size_t)b * MINA_STATES_PER_PROOF * 8;
    for (uint32_t i = 0; i < MINA_STATES_PER_PROOF * 8; ++i) good = good && (h[i] == x[i]);
    for (uint32_t s = 1; s < MINA_STATES_PER_PROOF - 1; ++s) {          // candidate chain: states 0..15 (oldest .. tip); 16 = bridge tip (not linked)
        const uint32_t *prev = records + ((size_t)b * MINA_STATES_PER_PROOF + s) * MINA_PSTATE_SLOTS * 8;   // slot 0 = previous_state_hash
        for (int i = 0; i < 8; ++i) good = good && (prev[i] == h[(s - 1) * 8 + i]);
    }
    ok[b] = good ? 1u : 0u;
}

// public-input commitment h - A as canonical affine words (16 per proof, zeros = infinity): feeds ipa_prepare_kernel's override slot
template <int FB>
__global__ void __launch_bounds__(64)
pubcomm_finish16_kernel(uint32_t batch, FieldK kb, const affine_t *__restrict__ h, const xyzz_t *__restrict__ a, uint32_t *__restrict__ out_words) { mb_wave_prio();
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= batch) return;
    xyzz_t t = a[m];
    t.y = fe_neg<FB>(t.y);
    const affine_t H = *h;
    xyzz_add_affine<FB>(t, H.x, H.y, kb.one);
    uint32_t *o = out_words + (size_t)m * 16;
    if (xyzz_is_inf(t)) { for (int i = 0; i < 16; ++i) o[i] = 0; return; }
    const fe_t zi = fe_inv<FB>(fe_mul<FB>(t.zz, t.zzz), kb);
    const fe_t x = fe_from_mont<FB>(fe_mul<FB>(t.x, fe_mul<FB>(zi, t.zzz))), y = fe_from_mont<FB>(fe_mul<FB>(t.y, fe_mul<FB>(zi, t.zz)));
    for (int i = 0; i < 8; ++i) { o[i] = x.v[i]; o[8 + i] = y.v[i]; }
}

__global__ void fill_u32_kernel(uint32_t n, uint32_t v, uint32_t *__restrict__ out) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = v; }

// verdict[b] = chain_ok[b] AND folded IPA verdict AND folded accumulator verdict; flags = {ipa, ipa malformed, acc, 0}
__global__ void state_job_verdict_kernel(uint32_t batch, const uint32_t *__restrict__ chain_ok, const uint32_t *__restrict__ ipa_v /* [2] or null */,
                                         const uint32_t *__restrict__ acc_v /* [1] or null */, const uint32_t *__restrict__ kimchi_bad /* [1] or null */,
                                         const uint32_t *__restrict__ stmt_ok /* [batch] or null: Pickles statement well-formed */,
                                         uint32_t *__restrict__ verdicts, uint32_t *__restrict__ flags, uint32_t *__restrict__ stmt_out /* [batch] or null: copy of stmt_ok */) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t kb = kimchi_bad ? kimchi_bad[0] : 0u;
    const uint32_t iv = (ipa_v ? ipa_v[0] : 1u) && !kb, av = acc_v ? acc_v[0] : 1u;
    if (b == 0 && flags) { flags[0] = iv; flags[1] = (ipa_v ? ipa_v[1] : 0u) | kb; flags[2] = av; flags[3] = 0u; }
    if (b < batch) { const uint32_t so = stmt_ok ? stmt_ok[b] : 1u; verdicts[b] = (chain_ok[b] && iv && av && so) ? 1u : 0u; if (stmt_out) stmt_out[b] = so; }
}

}  // namespace mb

// ------------------------------------------------------------------------------------------------ salts
static int ensure_state_salts(mina_ctx *c) {
    if (c->have_state_salts) return MINA_OK;
    if (!c->have_pparams[FIELD_FP]) return fail(MINA_ERR_STATE, "Poseidon constants not installed for Fp");
    const char *names[MB_N_PREFIX_SALTS] = {"MinaProtoStateBody", "MinaProtoState", "MinaAccount", "MinaZkappAccount", "MinaZkappUri", "MinaSideLoadedVk"};
    uint8_t pre[MB_N_PREFIX_SALTS * 32];
    for (int i = 0; i < MB_N_PREFIX_SALTS; ++i) { const mw::B32 f = mw::prefix_field(names[i]); memcpy(pre + 32 * i, f.b, 32); }
    int rc;
    if ((rc = c->state_salts.ensure(MB_N_PREFIX_SALTS * 3 * sizeof(fe_t)))) return rc;
    DevBuf tmp;
    if ((rc = tmp.ensure(sizeof pre))) return rc;
    HIPC(hipMemcpyAsync(tmp.p, pre, sizeof pre, hipMemcpyHostToDevice, c->L->stream));
    mb::prefix_salt_kernel<FIELD_FP><<<1, 64, 0, c->L->stream>>>(MB_N_PREFIX_SALTS, c->fk[FIELD_FP], c->pparams[FIELD_FP].as<PoseidonParams>(), tmp.as<uint32_t>(), c->state_salts.as<fe_t>());
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(c->L->stream));
    tmp.release();
    c->have_state_salts = true;
    return MINA_OK;
}

int mb_ensure_state_salts(mina_ctx *c) { return ensure_state_salts(c); }

static int pstate_hash_dev(mina_ctx *c, size_t n, const uint32_t *d_records, const uint32_t *d_nfields, uint32_t *d_hashes, uint32_t *d_bodies) {
    const PoseidonParams *pp = c->pparams[FIELD_FP].as<PoseidonParams>();
    const fe_t *salts = c->state_salts.as<fe_t>();
    ProfScope ps_(c, PS_STATE_HASH);
    // below ~8 k states the chip is latency-bound: 8 lanes per state (shortest chain); above, wave-packed triples (63 of 64 lanes busy)
    if (use_coop16(c, (n + MINA_STATES_PER_PROOF - 1) / MINA_STATES_PER_PROOF))
        mb::pstate_hash_kernel<FIELD_FP, 16><<<cdiv(n * 16, 256), 256, 0, c->L->stream>>>((uint32_t)n, c->fk[FIELD_FP], pp, salts, d_records, d_nfields, d_hashes, d_bodies);
    else if (use_coop8(c, n))
        mb::pstate_hash_kernel<FIELD_FP, 8><<<cdiv(n * 8, 256), 256, 0, c->L->stream>>>((uint32_t)n, c->fk[FIELD_FP], pp, salts, d_records, d_nfields, d_hashes, d_bodies);
    else if (c->hash_piece_waves && (n + 20) / 21 > c->hash_piece_waves) {
        // The whole launch would hold every wave slot its 88 VGPRs allow (5 per SIMD) for most of its 20 ms, and the kernels of the other legs /
        // chunks of a call (a few hundred waves each, one behind the other) would wait for slots.  In pieces of `hash_piece_waves` waves
        // (~2 per SIMD: the multiplier is still saturated) the rest of the register file stays free for them.
        const size_t per = (size_t)c->hash_piece_waves * 21;
        for (size_t lo = 0; lo < n; lo += per) {
            const size_t cnt = std::min(per, n - lo);
            mb::pstate_hash_kernel<FIELD_FP, 3><<<cdiv(coop_threads<3>(cnt), 256), 256, c->hash_lds_bytes, c->L->stream>>>((uint32_t)cnt, c->fk[FIELD_FP], pp, salts, d_records + lo * MINA_PSTATE_SLOTS * 8,
                                                                                                             d_nfields + lo, d_hashes + lo * 8, d_bodies ? d_bodies + lo * 8 : nullptr);
        }
    } else
        mb::pstate_hash_kernel<FIELD_FP, 3><<<cdiv(coop_threads<3>(n), 256), 256, c->hash_lds_bytes, c->L->stream>>>((uint32_t)n, c->fk[FIELD_FP], pp, salts, d_records, d_nfields, d_hashes, d_bodies);
    HIPC(hipGetLastError());
    return MINA_OK;
}

extern "C" int mina_protocol_state_hash_batch(mina_ctx *c, size_t n, const uint8_t *records, const uint32_t *n_body_fields, uint8_t *hashes_out,
                                              uint8_t *body_hashes_out) {
    if (!c || (n && (!records || !n_body_fields || !hashes_out))) return fail(MINA_ERR_ARG, "null argument");
    if (n == 0) return MINA_OK;
    if (n > (1u << 22)) return fail(MINA_ERR_ARG, "n too large");
    for (size_t i = 0; i < n; ++i) if (n_body_fields[i] > MINA_PSTATE_SLOTS - 1) return fail(MINA_ERR_ARG, "n_body_fields exceeds the record");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = ensure_state_salts(c))) return rc;
    Lane &L = *c->L;
    if ((rc = h2d(c, L.tmp_a, records, n * MINA_PSTATE_SLOTS * 32))) return rc;
    if ((rc = h2d(c, L.tmp_b, n_body_fields, n * 4))) return rc;
    if ((rc = L.tmp_c.ensure(n * 32))) return rc;
    if ((rc = L.tmp_d.ensure(n * 32))) return rc;
    if ((rc = pstate_hash_dev(c, n, L.tmp_a.as<uint32_t>(), L.tmp_b.as<uint32_t>(), L.tmp_c.as<uint32_t>(), body_hashes_out ? L.tmp_d.as<uint32_t>() : nullptr))) return rc;
    if (body_hashes_out) HIPC(hipMemcpyAsync(body_hashes_out, L.tmp_d.p, n * 32, hipMemcpyDeviceToHost, L.stream));
    return d2h_sync(c, hashes_out, L.tmp_c, n * 32);
}

// convenience: serialized states in, state hashes out (parse on the host, hash on the GPU); malformed state -> MINA_ERR_FORMAT
extern "C" int mina_protocol_state_hash_bytes(mina_ctx *c, int encoding, size_t n, const uint8_t *const *states, const size_t *lens, uint8_t *hashes_out) {
    if (!c || (n && (!states || !lens || !hashes_out))) return fail(MINA_ERR_ARG, "null argument");
    std::vector<uint8_t> rec(n * MINA_PSTATE_SLOTS * 32); std::vector<uint32_t> nf(n);
    for (size_t i = 0; i < n; ++i) {
        if (!states[i]) return fail(MINA_ERR_ARG, "null state");
        int rc = mina_protocol_state_pack(states[i], lens[i], encoding, &rec[i * MINA_PSTATE_SLOTS * 32], &nf[i], nullptr, nullptr);
        if (rc) return rc;
    }
    return mina_protocol_state_hash_batch(c, n, rec.data(), nf.data(), hashes_out, nullptr);
}

// ------------------------------------------------------------------------------------------------ the composite job
int mb_pickles_check(mina_ctx *c, const mina_pickles_statements *s);                                                    // api_pickles.hip
int mb_pickles_statements_dev(mina_ctx *c, size_t batch, const mina_pickles_statements *s, uint32_t *d_pub, uint32_t *d_ok);
size_t mb_pickles_sections(const mina_pickles_statements *s, const void ***slots, size_t *strides, mina_pickles_statements *copy);
static int check_jobs(mina_ctx *c, const mina_state_jobs *j) {
    if (!c || !j) return fail(MINA_ERR_ARG, "null argument");
    if (j->batch == 0 || j->batch > 65536) return fail(MINA_ERR_ARG, "batch must be in 1..65536");
    if (j->with_states && (!j->state_records || !j->state_nfields || !j->expected_hashes)) return fail(MINA_ERR_ARG, "null protocol-state section");
    if (j->with_ipa) {
        if (!j->lr || !j->delta || !j->sg || !j->z1 || !j->z2 || !j->rand_base || !j->sg_rand_base) return fail(MINA_ERR_ARG, "null IPA section");
        if (j->kimchi) {
            const mina_kimchi_proofs &kp = *j->kimchi;
            if (!c->have_kimchi) return fail(MINA_ERR_STATE, "no verifier index installed");
            if (kp.batch != j->batch || j->k != c->kimchi_log2 || j->log2_domain != c->kimchi_log2 || j->n_evalpoints != 2 || j->n_comms != kp.n_prev + 45 || kp.npub != j->npub || kp.n_prev > 8)
                return fail(MINA_ERR_ARG, "kimchi section does not match the installed index / the job's shape");
            if ((kp.n_prev && ((!kp.prev_chals && !kp.prev_prechallenges && !(kp.statements && kp.n_prev == 2 && j->k == 15)) || !kp.prev_comms)) || !kp.w_comm || !kp.z_comm || !kp.t_comm || !kp.evals || !kp.ft_eval1 || (kp.npub && !kp.public_inputs && !kp.statements)) return fail(MINA_ERR_ARG, "null kimchi section");
            if (kp.statements) { if (kp.npub != 40) return fail(MINA_ERR_ARG, "statements derive exactly 40 public inputs"); int prc = mb_pickles_check(c, kp.statements); if (prc) return prc; }
        } else if (!j->sponge_state || !j->sponge_pos || !j->cip || !j->evalscale || !j->polyscale || (j->n_evalpoints && !j->evalpoints) || (j->n_comms && !j->comms))
            return fail(MINA_ERR_ARG, "null IPA section");
        if (j->k < 1 || j->k > 20 || ((size_t)1 << j->k) > c->srs[CURVE_PALLAS].depth) return fail(c->srs[CURVE_PALLAS].depth ? MINA_ERR_ARG : MINA_ERR_STATE, "Pallas SRS missing or 2^k exceeds its depth");
        if (j->n_comms > 4096 || j->n_evalpoints > 64) return fail(MINA_ERR_ARG, "too many commitments / evaluation points");
        if (!c->have_pparams[FIELD_FP]) return fail(MINA_ERR_STATE, "Poseidon constants not installed for Fp");
    }
    if (j->npub) {
        if (!j->public_inputs && !(j->kimchi && j->kimchi->statements)) return fail(MINA_ERR_ARG, "null public inputs");
        if (!j->with_ipa || (!j->kimchi && j->pub_comm_slot >= j->n_comms)) return fail(MINA_ERR_ARG, "public-input commitment needs an IPA section and a valid commitment slot");
        if (j->log2_domain > 20 || ((uint64_t)1 << j->log2_domain) > c->srs[CURVE_PALLAS].depth || j->npub > 4096 || j->npub > ((uint64_t)1 << j->log2_domain)) return fail(MINA_ERR_ARG, "bad domain / npub");
    }
    if (j->with_accumulator) {
        if (!j->acc_prechallenges || !j->acc_sg || (j->batch > 1 && !j->acc_rho)) return fail(MINA_ERR_ARG, "null accumulator section");
        if (j->acc_k < 1 || j->acc_k > 20 || ((size_t)1 << j->acc_k) > c->srs[CURVE_VESTA].depth) return fail(c->srs[CURVE_VESTA].depth ? MINA_ERR_ARG : MINA_ERR_STATE, "Vesta SRS missing or 2^k exceeds its depth");
    }
    return MINA_OK;
}

int mb_ensure_lagrange_table(mina_ctx *c, int curve, uint32_t log2_domain, uint32_t npub);   // api_srs.hip
int mb_lagrange_sums_dev(mina_ctx *c, int curve, uint32_t npub, size_t batch, const uint32_t *d_scalars, void *d_out_xyzz);
namespace mb {   // api_kimchi.hip
struct KimchiIn { const uint32_t *pub, *prev_chals, *prev_comms, *w_comm, *z_comm, *t_comm, *evals, *ft_eval1, *pubcomm; };
struct KimchiOut { uint32_t *sponge_state, *sponge_pos, *cip, *evalpoints, *polyscale, *evalscale, *comms, *ft_eval0; };
}
int mb_kimchi_to_batch_dev(mina_ctx *c, size_t batch, uint32_t n_prev, uint32_t npub, const mb::KimchiIn &in, const mb::KimchiOut &out, uint32_t *d_bad, mb::IpaExpand *expand,
                           const void *pf_digest, uint32_t pf_stride);
void mb_pickles_kimchi_digest(mina_ctx *c, const mina_pickles_statements *s, const void **first, uint32_t *stride);   // api_pickles.hip

// public-input commitments h - sum_i pub_i L_i of `batch` proofs as canonical affine words (16 per proof) on the current lane
int mb_pubcomm_dev(mina_ctx *c, size_t batch, uint32_t log2_domain, uint32_t npub, const uint32_t *d_pub, uint32_t *d_out16) {
    Lane &L = *c->L;
    SrsState &s = c->srs[CURVE_PALLAS];
    int rc;
    if ((rc = L.st_pub_xyzz.ensure(batch * sizeof(xyzz_t)))) return rc;
    if (npub == 0) {                                                // empty public input: A = infinity, commitment = h
        HIPC(hipMemsetAsync(L.st_pub_xyzz.p, 0, batch * sizeof(xyzz_t), L.stream));
    } else {
        if (s.lagrange_table_log2 != (int)log2_domain || s.lagrange_table_n < npub) return fail(MINA_ERR_STATE, "call mina_state_jobs_prepare(log2_domain, npub) first");
        if ((rc = mb_lagrange_sums_dev(c, CURVE_PALLAS, npub, batch, d_pub, L.st_pub_xyzz.p))) return rc;
    }
    mb::pubcomm_finish16_kernel<FIELD_FP><<<cdiv(batch, 64), 64, 0, L.stream>>>((uint32_t)batch, c->fk[FIELD_FP], s.h.as<affine_t>(), L.st_pub_xyzz.as<xyzz_t>(), d_out16);
    HIPC(hipGetLastError());
    return MINA_OK;
}

// everything that needs a host synchronisation (salts, Lagrange basis + its window table): done once, before the first job
extern "C" int mina_state_jobs_prepare(mina_ctx *c, uint32_t log2_domain, uint32_t npub) {
    if (!c) return fail(MINA_ERR_ARG, "null argument");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = ensure_state_salts(c))) return rc;
    if (npub && (rc = mb_ensure_lagrange_table(c, CURVE_PALLAS, log2_domain, npub))) return rc;
    return MINA_OK;
}

// all pointers of `j` are device pointers; queued on the current lane
// `split`: the three independent legs (protocol states / wrap opening / accumulator) go to three lanes of the context -- lane 0 plus
// two helper lanes -- and are joined by events before the verdict kernel: for small, latency-bound batches the serial chain of
// dependent Poseidon permutations of one leg hides behind the other legs (host-buffer entry point only).
static int leg_fork(mina_ctx *c, Lane &from, Lane &to) {
    if (!to.stream) HIPC(hipStreamCreateWithFlags(&to.stream, hipStreamNonBlocking));
    if (!from.ev_leg) HIPC(hipEventCreateWithFlags(&from.ev_leg, hipEventDisableTiming));
    HIPC(hipEventRecord(from.ev_leg, from.stream));
    HIPC(hipStreamWaitEvent(to.stream, from.ev_leg, 0));
    return MINA_OK;
}
static int leg_join(Lane &leg, Lane &into) {
    if (!leg.ev_leg) HIPC(hipEventCreateWithFlags(&leg.ev_leg, hipEventDisableTiming));
    HIPC(hipEventRecord(leg.ev_leg, leg.stream));
    HIPC(hipStreamWaitEvent(into.stream, leg.ev_leg, 0));
    return MINA_OK;
}
// Early start of a job's protocol-state leg: the boundary (api_verify.hip) streams the records of a chunk to the GPU while the rest of the
// chunk is still being parsed, and queues the hashes of states [lo, lo + cnt) -- of a job of `ns_total` states whose state leg will run on
// lane LS -- behind `after` (the upload of those records).  mb_state_jobs_on_lane then takes `c->state_hashes_early` states as hashed.
int mb_state_hashes_early(mina_ctx *c, Lane *LS, size_t ns_total, size_t lo, size_t cnt, const uint32_t *d_records, const uint32_t *d_nfields, hipEvent_t after) {
    if (!LS || lo + cnt > ns_total) return fail(MINA_ERR_ARG, "bad early state range");
    if (!c->have_state_salts) return fail(MINA_ERR_STATE, "call mina_state_jobs_prepare first");
    if (!LS->stream) HIPC(hipStreamCreateWithFlags(&LS->stream, hipStreamNonBlocking));
    int rc;
    if ((rc = LS->st_hashes.ensure(ns_total * 32))) return rc;
    if (after) HIPC(hipStreamWaitEvent(LS->stream, after, 0));
    Lane *const L0 = c->L;
    c->L = LS;
    rc = pstate_hash_dev(c, cnt, d_records + lo * MINA_PSTATE_SLOTS * 8, d_nfields + lo, LS->st_hashes.as<uint32_t>() + lo * 8, nullptr);
    c->L = L0;
    return rc;
}

// LI / LA: helper lanes of the wrap-proof leg and the accumulator leg (nullptr = everything on the current lane, in order); LS: a lane of
// its own for the protocol-state leg as well (the boundary gives the chain and the hashes streams with disjoint CU masks, api_verify.hip)
// `phase`: MB_JOB_ALL queues the whole job.  The boundary queues a job in two steps, because the wrap-proof half of its input is parsed (and
// uploaded) before the protocol states are: MB_JOB_LEGS = the accumulator and wrap-proof legs (their verdict pointers are kept in `carry`),
// later MB_JOB_FINISH = the protocol-state leg (minus what mb_state_hashes_early queued already), the joins and the verdict kernel.
int mb_state_jobs_on_lane(mina_ctx *c, const mina_state_jobs *j, uint32_t *d_verdicts, uint32_t *d_flags, Lane *LI_, Lane *LA_, uint32_t *d_stmt_out, Lane *LS_, uint32_t phase, StateJobCarry *carry) {
    if (phase != MB_JOB_ALL && !carry) return fail(MINA_ERR_ARG, "a split job needs a carry");
    Lane &L = *c->L;
    Lane *const L0 = c->L;
    Lane *LI = L0, *LA = L0, *LS = L0;
    if (LI_ && LA_ && LI_ != L0 && LA_ != L0 && LI_ != LA_) {
        LI = LI_; LA = LA_;
        if (LS_ && LS_ != L0 && LS_ != LI) LS = LS_;             // LS == LA: the accumulator leg shares the hashes' stream (behind them, or ahead: mina_ctx::acc_first)
        c->legs_forked = true;
        int frc;
        if ((phase & MB_JOB_LEGS) && ((frc = leg_fork(c, *L0, *LI)) || (frc = leg_fork(c, *L0, *LA)))) return frc;
        if ((phase & MB_JOB_FINISH) && LS != L0 && (frc = leg_fork(c, *L0, *LS))) return frc;
    }
    const size_t B = j->batch;
    int rc;
    struct Unfork { mina_ctx *c; Lane *l0; ~Unfork() { c->legs_forked = false; c->L = l0; } } unfork{c, L0};
    const uint32_t *comm_override = nullptr;
    uint32_t *ipa_v = nullptr, *acc_v = nullptr;
    uint32_t *kimchi_bad = nullptr; const uint32_t *stmt_ok = nullptr;
    // ---- accumulator leg.  It shares scratch buffers with the opening check (ipa_chals / ipa_sigma / ipa_points of its lane), and the
    // culprit search re-checks slices of a failed batch from the rows the opening check LEFT in those buffers (mb_ipa_recheck_rows): on the
    // wrap leg's own lane the accumulator therefore runs FIRST (run after it, as it did until round 3, it overwrote the rows: every
    // part of a search then failed and a batch of more than 1024 proofs with one bad opening was rejected whole).
    auto accumulator_leg = [&]() -> int {
        c->L = LA;
        if (j->with_accumulator) {
            int r;
            if ((r = LA->st_flags.ensure(16 * 4))) return r;
            acc_v = LA->st_flags.as<uint32_t>() + 8;
            if ((r = mb_accumulator_check_dev(c, CURVE_VESTA, j->acc_k, B, (const uint32_t *)j->acc_prechallenges, (const uint32_t *)j->acc_sg,
                                              B > 1 ? (const uint32_t *)j->acc_rho : nullptr, acc_v))) return r;
        }
        return MINA_OK;
    };
    const bool acc_ahead = c->acc_first && LA == LS && LS != L0 && phase == MB_JOB_ALL;      // (device-resident jobs only: the boundary queues a job in two phases)
    if (acc_ahead && (rc = accumulator_leg())) { c->L = L0; return rc; }
    Lane &S = *LS;                                              // ---- protocol-state leg
    c->L = LS;
    if ((rc = S.st_ok.ensure(B * 4))) return rc;
    if (!(phase & MB_JOB_FINISH)) {}
    else if (j->with_states) {
        const size_t ns = B * MINA_STATES_PER_PROOF;
        if ((rc = S.st_hashes.ensure(ns * 32))) return rc;
        const size_t early = std::min(c->state_hashes_early, ns);      // already queued on this lane by mb_state_hashes_early
        c->state_hashes_early = 0;
        if (early < ns && (rc = pstate_hash_dev(c, ns - early, (const uint32_t *)j->state_records + early * MINA_PSTATE_SLOTS * 8, (const uint32_t *)j->state_nfields + early,
                                                S.st_hashes.as<uint32_t>() + early * 8, nullptr))) return rc;
        mb::pstate_chain_check_kernel<<<cdiv(B, 64), 64, 0, S.stream>>>((uint32_t)B, S.st_hashes.as<uint32_t>(), (const uint32_t *)j->expected_hashes,
                                                                         (const uint32_t *)j->state_records, (const uint8_t *)j->precheck, S.st_ok.as<uint32_t>());
    } else {
        // no state leg: chain_ok = all ones
        if (j->precheck) return fail(MINA_ERR_ARG, "precheck needs the protocol-state section");
        mb::fill_u32_kernel<<<cdiv(B, 256), 256, 0, S.stream>>>((uint32_t)B, 1u, S.st_ok.as<uint32_t>());
    }
    HIPC(hipGetLastError());
    c->L = L0;
    if ((phase & MB_JOB_LEGS) && LA == LI && (rc = accumulator_leg())) { c->L = L0; return rc; }
    c->L = LI;                                                  // ---- wrap-proof leg
    if (phase & MB_JOB_LEGS) {
    Lane &L = *LI;
    if ((rc = L.st_flags.ensure(16 * 4))) { c->L = L0; return rc; }
    const uint32_t *pub = (const uint32_t *)j->public_inputs;
    if (j->kimchi && j->kimchi->statements) {      // the Pickles statement -> the wrap circuit's public inputs, on this lane ahead of everything that reads them
        if ((rc = L.pk_pub.ensure(B * 40 * 32)) || (rc = L.pk_ok.ensure(B * 4))) { c->L = L0; return rc; }
        if ((rc = mb_pickles_statements_dev(c, B, j->kimchi->statements, L.pk_pub.as<uint32_t>(), L.pk_ok.as<uint32_t>()))) { c->L = L0; return rc; }
        pub = L.pk_pub.as<uint32_t>(); stmt_ok = L.pk_ok.as<uint32_t>();
    }
    if (j->npub || j->kimchi) {
        if ((rc = L.st_pubcomm.ensure(B * 64))) { c->L = L0; return rc; }
        if ((rc = mb_pubcomm_dev(c, B, j->log2_domain, j->npub, pub, L.st_pubcomm.as<uint32_t>()))) { c->L = L0; return rc; }
        comm_override = L.st_pubcomm.as<uint32_t>();
    }
    if (j->with_ipa) {
        mb::IpaShape sh; sh.batch = (uint32_t)B; sh.k = j->k; sh.npts = j->n_evalpoints; sh.ncomms = j->n_comms; sh.per = 2 * j->k + j->n_comms + 4;
        auto W = [](const void *p) { return (const uint32_t *)p; };
        ipa_v = L.st_flags.as<uint32_t>() + 4;
        if (j->kimchi) {
            // kimchi oracles + to_batch produce the BatchEvaluationProof rows in lane buffers (the public-input commitment is already in the list)
            const mina_kimchi_proofs &kp = *j->kimchi;
            if ((rc = L.kc_state.ensure(B * 96)) || (rc = L.kc_pos.ensure(B * 8)) || (rc = L.kc_cip.ensure(B * 32)) || (rc = L.kc_pts.ensure(B * 64)) ||
                (rc = L.kc_v.ensure(B * 32)) || (rc = L.kc_u.ensure(B * 32)) || (rc = L.kc_comms.ensure(B * (size_t)j->n_comms * 64))) { c->L = L0; return rc; }
            kimchi_bad = L.st_flags.as<uint32_t>() + 12;
            HIPC(hipMemsetAsync(kimchi_bad, 0, 4, L.stream));
            const uint32_t *prev_chals = W(kp.prev_chals);
            // no recursion challenges given beside a statement: they ARE the statement's messages_for_next_wrap_proof.old_bulletproof_challenges
            // (2 x 15; the one source a verifier has), and their digest was taken on the way by the statement stage
            const bool from_statement = !kp.prev_chals && !kp.prev_prechallenges && kp.statements && kp.n_prev == 2 && j->k == 15;
            const void *pre = from_statement ? kp.statements->wrap_old_challenges : kp.prev_prechallenges;
            const void *pf_digest = nullptr; uint32_t pf_stride = 0;
            if (from_statement && mb_tune().kimchi_shared_digest) mb_pickles_kimchi_digest(c, kp.statements, &pf_digest, &pf_stride);
            if (pre && kp.n_prev) {                             // 128-bit prechallenges -> scalar-field challenges, on this lane ahead of the sponges
                const size_t cnt = B * kp.n_prev * j->k;
                if ((rc = L.kc_pch.ensure(cnt * 32))) { c->L = L0; return rc; }
                challenge_to_field_kernel<FIELD_FQ><<<cdiv(cnt, 64), 64, 0, L.stream>>>((uint32_t)cnt, c->fk[FIELD_FQ], W(pre), L.kc_pch.as<uint32_t>());
                prev_chals = L.kc_pch.as<uint32_t>();
            }
            mb::KimchiIn in{pub, prev_chals, W(kp.prev_comms), W(kp.w_comm), W(kp.z_comm), W(kp.t_comm), W(kp.evals), W(kp.ft_eval1), comm_override};
            mb::KimchiOut out{L.kc_state.as<uint32_t>(), L.kc_pos.as<uint32_t>(), L.kc_cip.as<uint32_t>(), L.kc_pts.as<uint32_t>(), L.kc_v.as<uint32_t>(), L.kc_u.as<uint32_t>(),
                              L.kc_comms.as<uint32_t>(), nullptr};
            mb::IpaExpand ex;
            if ((rc = mb_kimchi_to_batch_dev(c, B, kp.n_prev, kp.npub, in, out, kimchi_bad, &ex, pf_digest, pf_stride))) { c->L = L0; return rc; }
            sh.expand_slot = kp.n_prev + 1; sh.per += mb::IPA_EXPAND - 1;          // the ft commitment enters the MSM as its 8 terms
            if (mb_tune().ipa_shared_points && kp.n_prev + 45 <= 64) {   // h, the 27 index columns and the index point of the ft combination are the same
                uint64_t m = 0;                                                       // points for every proof: their scalars are summed first (29 of 88 entries per proof)
                for (uint32_t i = kp.n_prev + 3; i < kp.n_prev + 9; ++i) m |= (uint64_t)1 << i;        // 6 selectors
                for (uint32_t i = kp.n_prev + 24; i < kp.n_prev + 45; ++i) m |= (uint64_t)1 << i;      // 15 coefficients + 6 sigma
                sh.shared_lo = (uint32_t)m; sh.shared_hi = (uint32_t)(m >> 32); sh.shared_h = 1; sh.shared_expand0 = 1; sh.nshared = 29;
            }
            mb::IpaDevIn iin{out.sponge_state, out.sponge_pos, out.cip, W(j->lr), W(j->delta), W(j->sg), W(j->z1), W(j->z2), out.evalpoints, out.evalscale, out.polyscale,
                             out.comms, nullptr, W(j->rand_base), W(j->sg_rand_base)};
            iin.expand = ex;
            if ((rc = mb_ipa_batch_check_dev(c, CURVE_PALLAS, sh, iin, ipa_v))) { c->L = L0; return rc; }
        } else {
            sh.override_slot = j->npub ? j->pub_comm_slot : 0xffffffffu;
            mb::IpaDevIn in{W(j->sponge_state), W(j->sponge_pos), W(j->cip), W(j->lr), W(j->delta), W(j->sg), W(j->z1), W(j->z2), W(j->evalpoints), W(j->evalscale),
                            W(j->polyscale), W(j->comms), comm_override, W(j->rand_base), W(j->sg_rand_base)};
            if ((rc = mb_ipa_batch_check_dev(c, CURVE_PALLAS, sh, in, ipa_v))) { c->L = L0; return rc; }
        }
    }
    }
    if ((phase & MB_JOB_LEGS) && LA != LI && !acc_ahead && (rc = accumulator_leg())) { c->L = L0; return rc; }
    c->L = L0;
    if (phase == MB_JOB_LEGS) { carry->ipa_v = ipa_v; carry->acc_v = acc_v; carry->kimchi_bad = kimchi_bad; carry->stmt_ok = stmt_ok; return MINA_OK; }
    if (phase == MB_JOB_FINISH) { ipa_v = carry->ipa_v; acc_v = carry->acc_v; kimchi_bad = carry->kimchi_bad; stmt_ok = carry->stmt_ok; }
    if (LI != L0) { int jrc; if ((jrc = leg_join(*LI, *L0)) || (jrc = leg_join(*LA, *L0)) || (LS != L0 && (jrc = leg_join(*LS, *L0)))) return jrc; }
    mb::state_job_verdict_kernel<<<cdiv(B, 64), 64, 0, L.stream>>>((uint32_t)B, S.st_ok.as<uint32_t>(), ipa_v, acc_v, kimchi_bad, stmt_ok, d_verdicts, d_flags, d_stmt_out);
    HIPC(hipGetLastError());
    return MINA_OK;
}

// The legs of a device-resident job on streams of their own (mina_verify_tuning.dev_fork).  One stream per job needs ~20 jobs in flight to
// fill the chip: the wrap-proof chain is ~15 dependent kernels of 781 waves (16 384 proofs in the 3-lane form) on 1024 SIMDs, and nothing of the SAME job may run
// beside them -- 47.6 GiB of workspaces and 71 ms of call latency for the headline.  The three legs of a job are independent until the verdict kernel (the same
// split the boundary makes per chunk, api_verify.hip setup_legs): pipeline lane i keeps the fork / join and the verdict kernel, its helper lanes
// MB_DEV_HELPER0 + 3 i + {0, 1, 2} run the wrap-proof chain, the accumulator check and the state hashes.  Streams are created once per context under the tuning
// then in force (a stream keeps its CU mask / priority for life).  A pinned lane (mina_ctx_pin_lane: the caller queues its own work on that lane's stream) forks as well: the legs
// start behind an event recorded on the lane and the lane waits for them before its verdict kernel, so everything the caller queued before the call is seen by every leg and
// everything it queues after the call sees every leg's output -- the exchange variant (mina_state_job_fold_dev under sharded.py's `ordered()` scope) keeps its ONE ordering stream.
static int dev_fork_lanes(mina_ctx *c, Lane **LI, Lane **LA, Lane **LS) {
    const mina_verify_tuning tu = mb_tune();
    const int li = (int)(c->L - c->lanes);
    if (!(tu.dev_fork & 1u) || c->nlanes > MB_DEV_FORK_MAX || li < 0 || li >= MB_DEV_FORK_MAX) return MINA_OK;
    const int in_flight = c->pinned >= 0 ? 1 : c->nlanes;          // a pinned context runs one job at a time
    Lane *h = &c->lanes[MB_DEV_HELPER0 + 3 * li];
    if (!h[0].stream || !h[1].stream || !h[2].stream) {
        const uint32_t mode = c->dev_fork_made ? c->dev_fork_made : tu.dev_fork;
        c->dev_fork_made = mode;
        int ncu = 0; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
        int prio_lo = 0, prio_hi = 0; (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);      // numerically lower = higher priority
        auto make = [&](Lane &ln, bool chain, bool masked, int prio) -> int {
            if (ln.stream) return MINA_OK;
            if (masked && tu.dev_chain_cus > 0 && tu.dev_chain_cus < (uint32_t)ncu && ncu <= 256) {
                // bit i of a mask = CU i / 8 of XCD i % 8 (tools/probes/cumask_probe.hip): the first dev_chain_cus bits are the same CUs of every XCD
                uint32_t mk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (uint32_t b2 = 0; b2 < (uint32_t)ncu; ++b2) if ((b2 < tu.dev_chain_cus) == chain) mk[b2 >> 5] |= 1u << (b2 & 31);
                if (hipExtStreamCreateWithCUMask(&ln.stream, 8, mk) == hipSuccess) return MINA_OK;
                (void)hipGetLastError(); ln.stream = nullptr;
            }
            if (prio_lo != prio_hi && (mode & 4u)) HIPC(hipStreamCreateWithPriority(&ln.stream, hipStreamNonBlocking, prio));
            else HIPC(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
            return MINA_OK;
        };
        int rc;
        if ((rc = make(h[0], true, (mode & 2u) != 0, prio_hi)) || (rc = make(h[1], true, false, (prio_lo + prio_hi) / 2)) || (rc = make(h[2], false, (mode & 2u) != 0, prio_lo))) return rc;
    }
    *LI = &h[0]; *LA = tu.dev_acc_lane ? &h[2] : &h[1]; *LS = &h[2];      // dev_acc_lane: the accumulator leg on the hashes' stream (1: behind them, 2: ahead)
    c->acc_first = tu.dev_acc_lane == 2;
    // The hashes of a forked job go out in pieces, so that the jobs in flight together ask for ~6 state-hash waves per SIMD (five fit beside nothing else, 96 VGPRs):
    // measured with the wave priorities on (lanes x piece grid at 4096 / 8192 / 16 384 proofs per call, profiles/r06_dev_fork.md) the best piece is ~6144 / lanes waves
    // whatever the call size -- 2 lanes 3072, 3: 2048, 4: 1536, 6: 1024 -- and a lone call is best left whole.
    uint32_t piece = tu.dev_piece_waves;
    if (piece == 0 && in_flight >= 2) piece = 6144u / (uint32_t)in_flight;
    if (piece == 0xffffffffu) piece = 0;
    c->hash_piece_waves = piece;
    // A lone forked job: its hashes would hold every wave slot their 96 VGPRs allow (5 per SIMD) and the chain's waves would wait for one to retire (~13 ms): the
    // hash workgroups reserve 41 KiB of LDS each -- three per CU = 3 waves per SIMD, 224 VGPRs left: room for a wave of every chain kernel but the two PolishToken
    // interpreters (244 / 194 VGPRs + 64 KiB of LDS: they wait for a CU to drain either way).  Lone calls
    // 16 384: 63.6 -> 61.5 ms, 8192: 38.7 -> 36.2 ms; with several jobs in flight the pieces do that job and the reservation costs 1 - 3 % (profiles/r06_dev_fork.md).
    uint32_t lds_kb = tu.dev_hash_lds_kb;
    if (lds_kb == 0 && in_flight == 1) lds_kb = 41;
    if (lds_kb == 0xffffffffu || lds_kb > 160) lds_kb = 0;
    c->hash_lds_bytes = lds_kb * 1024u;
    return MINA_OK;
}

extern "C" int mina_state_job_batch_dev(mina_ctx *c, const mina_state_jobs *jobs, void *d_verdicts, void *d_flags) {
    int rc = check_jobs(c, jobs);
    if (rc) return rc;
    if (!d_verdicts) return fail(MINA_ERR_ARG, "null argument");
    if (!c->have_state_salts && jobs->with_states) return fail(MINA_ERR_STATE, "call mina_state_jobs_prepare first");
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    Lane *LI = nullptr, *LA = nullptr, *LS = nullptr;
    if ((rc = dev_fork_lanes(c, &LI, &LA, &LS))) return rc;
    rc = mb_state_jobs_on_lane(c, jobs, (uint32_t *)d_verdicts, (uint32_t *)d_flags, LI, LA, nullptr, LS);
    c->hash_piece_waves = 0; c->hash_lds_bytes = 0; c->acc_first = false;
    return rc;
}

// SURVEY.md 8e.2 for the WHOLE job (the `north_star` variant: "a single reduce of partial sums over xGMI"): this shard's proofs go through every stage of the
// job -- state hashes, statements, kimchi, the opening transcripts, both folds -- but the two fixed-base MSMs and their comparisons are left to the caller,
// who exchanges the shards' folded scalar vectors (all-to-all), commits over its slice of the bases and reduces the partial points (mina_bridge_amd/sharded.py
// ShardedStateJob).  Written here, all in HBM: d_ipa_scalars 2^k x 32 B (Pallas, the wrap openings' sum_b sigma_b s_b), d_ipa_point 17 words (the shard's
// variable-base partial sum of the opening check, to be ADDED to the fixed-base part: the batch passes iff the total is the point at infinity),
// d_acc_scalars 2^acc_k x 32 B (Vesta, sum_b rho_b s_b of the step accumulators), d_acc_point 17 words (sum_b rho_b sg_b: must EQUAL the fixed-base part).
// d_verdicts[b] = every per-proof check of proof b (chain, linkage, statement, well-formed inputs) -- the folded checks are the caller's to AND in.
// Needs at least 2 proofs and both folded legs (with_ipa, with_accumulator).  The folding randomisers of the job are the shard's own (independent of the
// other shards'), and the opening fold's powers start at 1 (rho_b = rand_base^(b+1): IpaShape::pow_first) -- the shards' partials are ADDED, so no proof of any
// shard may carry a fixed coefficient (round 4 shipped ^b: first proofs of two shards with discrepancies +tH / -tH cancelled).
extern "C" int mina_state_job_fold_dev(mina_ctx *c, const mina_state_jobs *jobs, void *d_verdicts, void *d_flags, void *d_ipa_scalars, void *d_ipa_point,
                                       void *d_acc_scalars, void *d_acc_point) {
    int rc = check_jobs(c, jobs);
    if (rc) return rc;
    if (!d_verdicts || !d_ipa_scalars || !d_ipa_point || !d_acc_scalars || !d_acc_point) return fail(MINA_ERR_ARG, "null argument");
    if (jobs->batch < 2 || !jobs->with_ipa || !jobs->with_accumulator || !jobs->acc_rho) return fail(MINA_ERR_ARG, "the exchange variant needs >= 2 proofs and both folded legs");
    if (!c->have_state_salts && jobs->with_states) return fail(MINA_ERR_STATE, "call mina_state_jobs_prepare first");
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    mina_ctx::FoldExport fe; fe.ipa_scalars = (uint32_t *)d_ipa_scalars; fe.ipa_point = (uint32_t *)d_ipa_point; fe.acc_scalars = (uint32_t *)d_acc_scalars; fe.acc_point = (uint32_t *)d_acc_point;
    c->fold_export = &fe;
    Lane *LI = nullptr, *LA = nullptr, *LS = nullptr;
    if (!(rc = dev_fork_lanes(c, &LI, &LA, &LS))) rc = mb_state_jobs_on_lane(c, jobs, (uint32_t *)d_verdicts, (uint32_t *)d_flags, LI, LA, nullptr, LS);
    c->hash_piece_waves = 0; c->hash_lds_bytes = 0; c->acc_first = false;
    c->fold_export = nullptr;
    return rc;
}

// host-buffer form: one upload of every section, the pipeline, one download; when a folded check fails the proofs are
// re-checked in parts (32-way cuts, the parts of a round concurrently) so that every proof gets its own verdict (README.md:281-310: every failure is `false`)
namespace {
struct Section { const void **slot; size_t bytes; };
}
extern "C" int mina_state_job_batch(mina_ctx *c, const mina_state_jobs *jobs, uint8_t *verdicts) {
    int rc = check_jobs(c, jobs);
    if (rc) return rc;
    if (!verdicts) return fail(MINA_ERR_ARG, "null argument");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    if ((rc = mina_state_jobs_prepare(c, jobs->log2_domain, jobs->npub))) return rc;
    c->use_lane0();
    Lane &L = *c->L;
    mina_state_jobs d = *jobs;
    const size_t B = jobs->batch, k = jobs->k, m = jobs->n_comms, np = jobs->n_evalpoints, S = MINA_STATES_PER_PROOF;
    std::vector<Section> secs;
    auto add = [&](const void *&slot, size_t bytes) { if (slot && bytes) secs.push_back({&slot, bytes}); };
    if (d.with_states) { add(d.state_records, B * S * MINA_PSTATE_SLOTS * 32); add(d.state_nfields, B * S * 4); add(d.expected_hashes, B * S * 32); add(d.precheck, B); }
    if (d.npub) add(d.public_inputs, B * d.npub * 32);
    mina_kimchi_proofs kd{}; mina_pickles_statements sd{};
    if (d.with_ipa && d.kimchi) {
        kd = *d.kimchi; d.kimchi = &kd;
        kd.public_inputs = nullptr;                                   // the job's own public_inputs section is the one uploaded
        add(kd.prev_chals, B * kd.n_prev * k * 32); add(kd.prev_prechallenges, B * kd.n_prev * k * 16); add(kd.prev_comms, B * kd.n_prev * 64); add(kd.w_comm, B * 15 * 64); add(kd.z_comm, B * 64);
        add(kd.t_comm, B * 7 * 64); add(kd.evals, B * 43 * 64); add(kd.ft_eval1, B * 32);
        if (kd.statements) {
            const void **slots[12]; size_t strides[12];
            mb_pickles_sections(kd.statements, slots, strides, &sd);
            kd.statements = &sd;
            for (int i = 0; i < 12; ++i) add(*slots[i], B * strides[i]);
        }
        d.sponge_state = d.sponge_pos = d.cip = d.evalpoints = d.evalscale = d.polyscale = d.comms = nullptr;
    }
    if (d.with_ipa) {
        add(d.sponge_state, B * 96); add(d.sponge_pos, B * 8); add(d.cip, B * 32); add(d.lr, B * 2 * k * 64); add(d.delta, B * 64); add(d.sg, B * 64);
        add(d.z1, B * 32); add(d.z2, B * 32); add(d.evalpoints, B * np * 32); add(d.evalscale, B * 32); add(d.polyscale, B * 32); add(d.comms, B * m * 64);
        add(d.rand_base, 32); add(d.sg_rand_base, 32);
    }
    if (d.with_accumulator) { add(d.acc_prechallenges, B * d.acc_k * 16); add(d.acc_sg, B * 64); add(d.acc_rho, B * 32); }
    size_t total = 0;
    std::vector<size_t> offs;
    for (auto &s : secs) { offs.push_back(total); total += (s.bytes + 255) & ~(size_t)255; }
    if ((rc = L.host_stage.ensure(total + B * 4))) return rc;
    uint8_t *blob = (uint8_t *)L.host_stage.p;
    for (size_t i = 0; i < secs.size(); ++i) memcpy(blob + offs[i], *secs[i].slot, secs[i].bytes);
    if ((rc = L.st_in.ensure(total))) return rc;
    HIPC(hipMemcpyAsync(L.st_in.p, blob, total, hipMemcpyHostToDevice, L.stream));
    for (size_t i = 0; i < secs.size(); ++i) *secs[i].slot = L.st_in.as<uint8_t>() + offs[i];
    if (d.kimchi) kd.public_inputs = d.public_inputs;
    if ((rc = L.st_verdicts.ensure(2 * B * 4 + 16))) return rc;
    uint32_t *dv = L.st_verdicts.as<uint32_t>(), *df = dv + B, *ds = df + 4;
    // small, latency-bound batches: the three independent legs go to three lanes (lane 0 plus two helpers), joined by events
    const bool split = B <= 1024 && c->nlanes == 1;
    if ((rc = mb_state_jobs_on_lane(c, &d, dv, df, split ? &c->lanes[1] : nullptr, split ? &c->lanes[2] : nullptr, ds, nullptr))) return rc;
    std::vector<uint32_t> hv(2 * B + 4);
    if ((rc = d2h_sync(c, hv.data(), L.st_verdicts, (2 * B + 4) * 4))) return rc;
    std::vector<uint8_t> stmt_each(B); for (size_t b = 0; b < B; ++b) stmt_each[b] = hv[B + 4 + b] ? 1 : 0;
    const bool ipa_ok = hv[B] != 0, acc_ok = hv[B + 2] != 0;
    if (ipa_ok && acc_ok) { for (size_t b = 0; b < B; ++b) verdicts[b] = hv[b] ? 1 : 0; return MINA_OK; }
    // a folded check failed somewhere: find the culprits.  chain_ok per proof comes from a run without the folded legs.
    std::vector<uint8_t> ipa_each(B, 1), acc_each(B, 1), chain_each(B, 1);
    {
        mina_state_jobs only = d; only.with_ipa = 0; only.with_accumulator = 0; only.npub = 0; only.kimchi = nullptr;
        if (only.with_states) {
            if ((rc = mb_state_jobs_on_lane(c, &only, dv, df, nullptr, nullptr, nullptr, nullptr))) return rc;
            if ((rc = d2h_sync(c, hv.data(), L.st_verdicts, B * 4))) return rc;
            for (size_t b = 0; b < B; ++b) chain_each[b] = hv[b] ? 1 : 0;
        }
    }
    mina_kimchi_proofs kslice{}; mina_pickles_statements sslice{};
    auto slice = [&](const mina_state_jobs &src, size_t lo, size_t cnt) {
        mina_state_jobs s = src; s.batch = cnt; s.with_states = 0; s.precheck = nullptr;
        auto adv = [&](const void *&p, size_t stride) { if (p) p = (const uint8_t *)p + lo * stride; };
        if (s.kimchi) {
            kslice = *s.kimchi; kslice.batch = cnt; s.kimchi = &kslice;
            adv(kslice.public_inputs, (size_t)kslice.npub * 32); adv(kslice.prev_chals, (size_t)kslice.n_prev * k * 32); adv(kslice.prev_prechallenges, (size_t)kslice.n_prev * k * 16); adv(kslice.prev_comms, (size_t)kslice.n_prev * 64);
            adv(kslice.w_comm, 15 * 64); adv(kslice.z_comm, 64); adv(kslice.t_comm, 7 * 64); adv(kslice.evals, 43 * 64); adv(kslice.ft_eval1, 32);
            if (kslice.statements) {
                const void **slots[12]; size_t strides[12];
                mb_pickles_sections(kslice.statements, slots, strides, &sslice);
                kslice.statements = &sslice;
                for (int i = 0; i < 12; ++i) adv(*slots[i], strides[i]);
            }
        }
        adv(s.public_inputs, (size_t)s.npub * 32);
        adv(s.sponge_state, 96); adv(s.sponge_pos, 8); adv(s.cip, 32); adv(s.lr, 2 * k * 64); adv(s.delta, 64); adv(s.sg, 64); adv(s.z1, 32); adv(s.z2, 32);
        adv(s.evalpoints, np * 32); adv(s.evalscale, 32); adv(s.polyscale, 32); adv(s.comms, m * 64);
        adv(s.acc_prechallenges, (size_t)s.acc_k * 16); adv(s.acc_sg, 64); adv(s.acc_rho, 32);
        return s;
    };
    // the culprits of one folded leg: a failing range is cut into FAN parts whose jobs run CONCURRENTLY on lanes 0..FAN-1 of the context
    // (inputs stay where lane 0 uploaded them; every lane has its own verdict words); parts that fail are cut again: depth log_FAN(B) rounds
    // of ~one job latency each, where a bisection ran log2(B) jobs one after the other per culprit.  FAN = 4 (mina_verify_tuning.search_fan): with 32 -- the
    // fan-out until the end of round 3 -- a search over 8192 proofs took 370 ms instead of 165, and the 28 streams it created left the process
    // with more streams than the runtime has hardware queues: every later call of the boundary was 30 % slower, for the life of the process
    // (tools/after_search.py; destroying the streams afterwards does not undo it).
    // The opening leg of well-formed proofs does not repeat its transcripts: the prepared rows of the failed batch are still on their
    // lane and any slice of them is the folded check of that slice (mb_ipa_recheck_rows): a round costs a fold + two MSMs per part.
    const bool rows_ok = hv[B + 1] == 0 && c->ipa_rows && c->ipa_rows_batch == B && !mb_tune().search_full;
    auto search = [&](bool ipa_leg, std::vector<uint8_t> &each) -> int {
        const size_t FAN = std::min<size_t>(MB_PIPE_LANES, std::max<size_t>(2, (size_t)mb_tune().search_fan));
        static const bool timing = getenv("MINA_VERIFY_TIMING") != nullptr;
        // streams this search had to create are destroyed when it is done
        struct Restore {
            mina_ctx *c; std::vector<size_t> made;
            ~Restore() { for (size_t i : made) if (c->lanes[i].stream) { (void)hipStreamSynchronize(c->lanes[i].stream); (void)hipStreamDestroy(c->lanes[i].stream); c->lanes[i].stream = nullptr; } c->use_lane0(); }
        } restore{c, {}};
        for (size_t i = 0; i < FAN; ++i)
            if (!c->lanes[i].stream) { HIPC(hipStreamCreateWithFlags(&c->lanes[i].stream, hipStreamNonBlocking)); restore.made.push_back(i); }
        std::vector<std::pair<size_t, size_t>> failing{{0, B}};
        while (!failing.empty()) {
            std::vector<std::pair<size_t, size_t>> parts;
            for (auto [lo, cnt] : failing) {
                if (cnt == 1) { each[lo] = 0; continue; }
                const size_t np_ = std::min(FAN, cnt);
                for (size_t q = 0; q < np_; ++q) { const size_t a = lo + cnt * q / np_, e = lo + cnt * (q + 1) / np_; parts.push_back({a, e - a}); }
            }
            failing.clear();
            const auto t_round = std::chrono::steady_clock::now();
            for (size_t base = 0; base < parts.size(); base += FAN) {
                const size_t w = std::min(FAN, parts.size() - base);
                uint32_t *flags_at[MB_PIPE_LANES];
                for (size_t q = 0; q < w; ++q) {                      // issue: nothing here waits for the GPU
                    const auto [lo, cnt] = parts[base + q];
                    c->L = &c->lanes[q];
                    int r = c->L->st_verdicts.ensure(2 * cnt * 4 + 16);
                    if (r) return r;
                    uint32_t *v = c->L->st_verdicts.as<uint32_t>();
                    flags_at[q] = v + cnt;
                    if (ipa_leg && rows_ok) { if ((r = mb_ipa_recheck_rows(c, lo, cnt, flags_at[q]))) return r; continue; }
                    mina_state_jobs sj = slice(d, lo, cnt);
                    if (ipa_leg) sj.with_accumulator = 0; else { sj.with_ipa = 0; sj.npub = 0; sj.kimchi = nullptr; }
                    if ((r = mb_state_jobs_on_lane(c, &sj, v, flags_at[q], nullptr, nullptr, nullptr, nullptr))) return r;
                }
                for (size_t q = 0; q < w; ++q) {
                    uint32_t f[4];
                    HIPC(hipStreamSynchronize(c->lanes[q].stream));
                    HIPC(hipMemcpy(f, flags_at[q], 16, hipMemcpyDeviceToHost));
                    if (!(ipa_leg ? f[0] != 0 : f[2] != 0)) failing.push_back(parts[base + q]);
                }
            }
            if (timing) fprintf(stderr, "mina_state_job_batch: culprit search (%s), %zu parts -> %zu failing, %.2f ms\n", ipa_leg ? "opening" : "accumulator", parts.size(), failing.size(),
                                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_round).count());
        }
        return MINA_OK;
    };
    if (!ipa_ok && (rc = search(true, ipa_each))) return rc;
    if (!acc_ok && (rc = search(false, acc_each))) return rc;
    for (size_t b = 0; b < B; ++b) verdicts[b] = (chain_each[b] && ipa_each[b] && acc_each[b] && stmt_each[b]) ? 1 : 0;
    return MINA_OK;
}
