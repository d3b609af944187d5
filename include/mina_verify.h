/*
 * mina_verify.h -- C-ABI of libminaverify.so, the MI355X-native (gfx950) Kimchi/Pickles IPA batch
 * verifier that sits behind the byte contract of lambdaclass/mina_bridge.
 *
 * What each entry point replaces (reference = /root/reference; "pin" = un-vendored crate pinned in
 * core/Cargo.toml:14-25, whose source is not in the tree -- see SURVEY.md section 0 and 8b):
 *
 *   (whole library)
 *       the verifier the bytes built at core/src/aligned.rs:31-58 are handed to
 *       (`ProvingSystemId::Mina` / `MinaAccount`; README.md:275-279,358-362: Aligned's
 *       `verify_mina_state_ffi` / `verify_account_inclusion_ffi`); this header exports the hot path
 *       of that verifier -- the combined IPA check and its kernels.
 *   mina_srs_*
 *       poly-commitment `SRS::create` and the loader for srs/vesta.srs, srs/pallas.srs
 *       (MessagePack `SRS{g,h}`; format in SURVEY.md section 0).
 *   mina_msm*
 *       ark-ec 0.3 `VariableBaseMSM::multi_scalar_mul` (pin core/Cargo.toml:20,49; README.md:538-540).
 *   mina_b_poly*, mina_ipa_*
 *       poly-commitment `b_poly`, `b_poly_coefficients`, `SRS::verify` (pin core/Cargo.toml:16;
 *       README.md:469-475,534-544).
 *   mina_poseidon_*, mina_challenge_to_field
 *       mina-poseidon `ArithmeticSponge` (PlonkSpongeConstantsKimchi), kimchi `ScalarChallenge::to_field`
 *       (pin core/Cargo.toml:14; README.md:443).
 *   mina_to_group
 *       groupmap `BWParameters::to_group` (core/Cargo.lock:2825-2827).
 *
 * Conventions
 *   - plain C types only; caller owns every host buffer; the library owns device memory, SRS tables
 *     and streams inside a `mina_ctx`.  No exceptions cross the ABI: every function returns 0 (MINA_OK)
 *     or a negative error code; verdicts are separate outputs.
 *   - field element: 32-byte little-endian canonical integer (< modulus), the ark `CanonicalSerialize`
 *     form used at core/src/sol/serialization.rs:63-86.  Values >= modulus are a caller error (upstream's
 *     deserialiser rejects them before this boundary); the arithmetic entry points (MSM, b_poly, Poseidon, Merkle) do
 *     not re-check and the result is then unspecified.  The proof-level entry points (mina_ipa_batch_check, the wire
 *     parsers) do validate and reject.
 *   - affine point: x || y, 64 bytes; the point at infinity is 64 zero bytes.
 *   - `field`: 0 = Fp (Pallas base, Vesta scalar), 1 = Fq (Vesta base, Pallas scalar).
 *   - `curve`: 0 = Pallas (base Fp, scalar Fq), 1 = Vesta (base Fq, scalar Fp).
 *   - entry points ending in `_dev` take device pointers (HBM-resident inputs) and run on the
 *     context's stream without synchronising unless stated.
 *   - a context is bound to one GPU; calls on one context are serialised by the caller
 *     (one context per thread / per rank).
 */
#ifndef MINA_VERIFY_H
#define MINA_VERIFY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MINA_OK 0
#define MINA_ERR_ARG (-1)
#define MINA_ERR_HIP (-2)
#define MINA_ERR_STATE (-3)   /* e.g. SRS not loaded */
#define MINA_ERR_FORMAT (-4)  /* malformed bytes */

#define MINA_FIELD_FP 0
#define MINA_FIELD_FQ 1
#define MINA_CURVE_PALLAS 0
#define MINA_CURVE_VESTA 1

typedef struct mina_ctx mina_ctx;

/* ---- context -------------------------------------------------------------------------------- */
int mina_ctx_create(int device_id, mina_ctx **out);
void mina_ctx_destroy(mina_ctx *ctx);
/* last error text of the calling thread (never NULL) */
const char *mina_last_error(void);
/* block until everything queued on the context's stream has finished */
int mina_ctx_synchronize(mina_ctx *ctx);
/* the hipStream_t the `_dev` entry points are queued on: lane 0's, or the pinned lane's (for event timing and for stream-ordered work of the caller) */
void *mina_ctx_stream(mina_ctx *ctx);
/* Pin every `_dev` entry point to pipeline lane `lane` (0 <= lane < lanes; negative: back to round-robin).  While pinned, all work the library queues is
 * ordered on mina_ctx_stream(): a caller that queues its own copies / kernels / collectives on that stream needs no host synchronisation between calls
 * (the multi-GPU exchange step, SURVEY.md 8e.2: mina_bridge_amd/sharded.py).  mina_ctx_set_pipeline unpins. */
int mina_ctx_pin_lane(mina_ctx *ctx, int lane);

/* Pipelining: the `_dev` entry points are issued round-robin over `lanes` internal streams (1..32, default 1),
 * each with its own workspace, so independent calls overlap on the GPU.  mina_ctx_synchronize waits for all. */
int mina_ctx_set_pipeline(mina_ctx *ctx, int lanes);
/* Device memory for the `_dev` entry points, for callers without a HIP binding of their own (the copies are synchronous;
 * download first waits for everything queued on the context). */
int mina_dev_malloc(mina_ctx *ctx, size_t bytes, void **out);
int mina_dev_free(mina_ctx *ctx, void *p);
int mina_dev_upload(mina_ctx *ctx, void *dst, const void *src, size_t bytes);
int mina_dev_download(mina_ctx *ctx, void *dst, const void *src, size_t bytes);

/* HIP-event stage timing on the context stream.  `stage_mask` has one bit per pipeline stage (0 = off,
 * -1 = all; bit 3 = the MSM bucket-accumulate kernel).  mina_prof_read synchronises and writes a JSON object
 * {"stage": [launches, total_ms], ...} covering everything recorded since the previous read. */
int mina_prof_enable(mina_ctx *ctx, int stage_mask);
int mina_prof_read(mina_ctx *ctx, char *buf, size_t cap);

/* ---- SRS (a6) -------------------------------------------------------------------------------- */
/* Regenerate SRS{g[0..depth), h} exactly as poly-commitment `SRS::create(depth)` does
 * (BLAKE2b-512 -> field -> BW group map, K4 on the GPU) and build the MSM window tables in HBM. */
int mina_srs_create(mina_ctx *ctx, int curve, uint32_t depth);
/* Build (on != 0) or drop the pre-split form of the curve's window table: 128-byte records holding x, y and p - y as nine 29-bit limbs each, read by the accumulate
 * kernels when mina_verify_tuning.msm_fp29 = 2 (no limb conversion, no negation; twice the gather traffic, 128 MiB per curve at depth 2^16).  Without it
 * msm_fp29 = 2 behaves as 1.  Waits for the context's queued work. */
int mina_srs_split_table(mina_ctx *ctx, int curve, int on);
/* Load from the MessagePack bytes of srs/vesta.srs / srs/pallas.srs (33-byte compressed points). */
int mina_srs_load(mina_ctx *ctx, int curve, const uint8_t *msgpack, size_t len);
/* depth of the loaded SRS (0 if none) */
uint32_t mina_srs_depth(mina_ctx *ctx, int curve);
/* copy g[first..first+count) (affine, canonical bytes) / h to host */
int mina_srs_get_g(mina_ctx *ctx, int curve, uint32_t first, uint32_t count, uint8_t *out_affine);
int mina_srs_get_h(mina_ctx *ctx, int curve, uint8_t *out_affine);
/* Lagrange-basis commitments of g[0..2^k) over the radix-2 domain of size 2^k (poly-commitment `SRS::add_lagrange_basis`,
 * used by kimchi for the public-input commitment): out[i] = (1/n) sum_j w^(-ij) g[j], affine canonical bytes, n*64. */
int mina_srs_lagrange_basis(mina_ctx *ctx, int curve, uint32_t log2_domain, uint8_t *out_affine);
/* kimchi verifier's public-input commitment: h - sum_i public[i] * lagrange[i]  (i < npub; `mask_custom` with blinder 1).
 * The Lagrange basis of the domain is computed once and cached in the context. */
int mina_public_input_commitment(mina_ctx *ctx, int curve, uint32_t log2_domain, size_t npub, const uint8_t *public_inputs /* npub*32 */,
                                 uint8_t *out_affine);
/* `batch` public-input commitments at once (batch_verify computes one per proof): out[m] = h - sum_i public[m][i] * lagrange[i].
 * The first npub basis points get a fixed-base window table once; every proof is one problem of the multi-problem MSM. */
int mina_public_input_commitment_batch(mina_ctx *ctx, int curve, uint32_t log2_domain, size_t npub, size_t batch,
                                       const uint8_t *public_inputs /* batch*npub*32 */, uint8_t *out_affine /* batch*64 */);
/* serialise back to the reference's file format; *len receives the size (2 293 801 for depth 2^16) */
int mina_srs_serialize(mina_ctx *ctx, int curve, uint8_t *out, size_t cap, size_t *len);

/* ---- K1: multi-scalar multiplication (a7) ---------------------------------------------------- */
/* out = sum_i scalars[i] * bases[i]; any points (variable-base Pippenger). */
int mina_msm(mina_ctx *ctx, int curve, size_t n, const uint8_t *bases_affine, const uint8_t *scalars,
             uint8_t *out_affine);
/* out = sum_i scalars[i] * g[i], g = loaded SRS of `curve`, n <= depth (fixed-base window tables). */
int mina_msm_srs(mina_ctx *ctx, int curve, size_t n, const uint8_t *scalars, uint8_t *out_affine);
/* out = sum_i scalars[i] * g[first + i]: the slice of the SRS owned by one rank when an MSM is sharded by bases. */
int mina_msm_srs_range(mina_ctx *ctx, int curve, uint32_t first, size_t n, const uint8_t *scalars, uint8_t *out_affine);
/* nprob MSMs over the same SRS bases in ONE kernel pipeline: out[m] = sum_i scalars[m][i] * g[i], i < n.
 * (poly-commitment `SRS::commit_non_hiding` of nprob polynomials -- kimchi commits ~45 per proof; also nprob un-folded
 * accumulator MSMs.)  Each problem owns one set of 2^15 buckets; everything else is shared. */
int mina_msm_srs_multi(mina_ctx *ctx, int curve, size_t n, size_t nprob, const uint8_t *scalars /* nprob*n*32 */,
                       uint8_t *out_affine /* nprob*64 */);
/* Same with scalars already in HBM (n x 32 bytes, canonical).  Queued on the context stream; the
 * 68-byte result record {x[32], y[32], u32 is_infinity} is written to `d_out` (device memory). */
int mina_msm_srs_dev(mina_ctx *ctx, int curve, size_t n, const void *d_scalars, void *d_out);

/* ---- K2: IPA challenge polynomial (a9) ------------------------------------------------------- */
/* b_poly(chals, x) for `npoints` evaluation points */
int mina_b_poly(mina_ctx *ctx, int field, uint32_t k, const uint8_t *chals, size_t npoints,
                const uint8_t *xs, uint8_t *out);
/* s[0..2^k) = b_poly_coefficients(chals) */
int mina_b_poly_coefficients(mina_ctx *ctx, int field, uint32_t k, const uint8_t *chals, uint8_t *out);
/* folded[j] = sum_b weights[b] * b_poly_coefficients(chals[b])[j], j < 2^k  (batch fold of a8) */
int mina_b_poly_fold(mina_ctx *ctx, int field, uint32_t k, size_t batch, const uint8_t *chals /* batch*k */,
                     const uint8_t *weights /* batch */, uint8_t *out /* 2^k */);
int mina_b_poly_fold_dev(mina_ctx *ctx, int field, uint32_t k, size_t batch, const void *d_chals,
                         const void *d_weights, void *d_out);

/* ---- K3: Poseidon (a12) / challenges (a13) --------------------------------------------------- */
/* Install the sponge constants of `field`: mds[9] row-major then rc[55*3], 32-byte LE each.
 * The constants are parameters of the engine (the real fp_kimchi/fq_kimchi tables are not in the
 * reference tree). */
int mina_poseidon_set_params(mina_ctx *ctx, int field, const uint8_t *params /* (9+165)*32 */);
/* The same from the text forms upstream publishes the tables in: the JSON object of o1js' constants.ts ({"mds": [[...]], "roundConstants":
 * [[...]], ...}) or the Rust table of mina-poseidon's fp_kimchi.rs / fq_kimchi.rs (`mds: vec![...]`, `round_constants: vec![...]`); decimal
 * or 0x-hex literals, quoted or not.  Exactly 9 + 55 x 3 canonical field elements, else MINA_ERR_FORMAT.  The process-wide contexts of
 * mina_verify_* load $MINA_POSEIDON_PARAMS_FP / $MINA_POSEIDON_PARAMS_FQ (file paths) at start-up instead of the compiled-in surrogate. */
int mina_poseidon_params_parse(int field, const char *text, size_t len, uint8_t *params_out /* (9+165)*32 */);   /* host-side, no context */
int mina_poseidon_load_params(mina_ctx *ctx, int field, const char *text, size_t len);
int mina_poseidon_load_params_file(mina_ctx *ctx, int field, const char *path);
/* in-place permutation of n states (3 field elements each) */
int mina_poseidon_permute(mina_ctx *ctx, int field, size_t n, uint8_t *states /* n*96 */);
int mina_poseidon_permute_dev(mina_ctx *ctx, int field, size_t n, void *d_states);
/* n independent sponges: absorb `len` elements each (inputs: n*len*32 bytes), squeeze one. */
int mina_poseidon_hash(mina_ctx *ctx, int field, size_t n, size_t len, const uint8_t *inputs, uint8_t *out);
/* ScalarChallenge::to_field: n 128-bit challenges (16 bytes LE each) -> field, endo = endo_r of the
 * curve whose scalar field is `field`. */
int mina_challenge_to_field(mina_ctx *ctx, int field, size_t n, const uint8_t *chal128, uint8_t *out);

/* ---- a16: Merkle-path fold of Proof-of-Account (core/src/proof/account_proof.rs:9-14,30-35; README.md:358-362) ------
 * roots[i] = fold of leaves[i] along its path: node <- H_h(node, sib) for dirs = 0 (`MerkleNode::Left(sib)`: the node is
 * the left input) or H_h(sib, node) for dirs = 1 (`MerkleNode::Right(sib)`), h = 0..depth-1, with the per-height salt
 * "MinaMklTree%03d" (mina `hash_with_kimchi`).  depth <= 64.  Runs 4 lanes per path (cooperative Poseidon). */
int mina_merkle_roots(mina_ctx *ctx, int field, size_t n, uint32_t depth, const uint8_t *leaves /* n*32 */,
                      const uint8_t *siblings /* n*depth*32 */, const uint8_t *dirs /* n*depth */, uint8_t *roots_out /* n*32 */);
/* verdicts[i] = (root of path i == expected_roots[i]) -- the ledger-hash comparison of verify_account_inclusion */
int mina_merkle_verify_batch(mina_ctx *ctx, int field, size_t n, uint32_t depth, const uint8_t *leaves, const uint8_t *siblings,
                             const uint8_t *dirs, const uint8_t *expected_roots, uint8_t *verdicts);

/* ---- a12: batched Fq-sponge transcripts (mina-poseidon `DefaultFqSponge` over the base field of `curve`) -------------
 * Every proof of the batch runs the same `tape` of opcodes over its own input stream (inputs: per proof, the operands in
 * tape order, 32 B per field element, 64 B per point; outputs: per proof, 32 B per squeeze opcode).  4 lanes per proof.
 * init_state/init_pos (may both be NULL = fresh sponge): 3 field elements + {mode, count} per proof, as in mina_ipa_opening;
 * final_state/final_pos (may be NULL) receive the sponge after the tape. */
#define MINA_TAPE_ABSORB_FQ 0      /* base-field element */
#define MINA_TAPE_ABSORB_G 1       /* affine point (x, y); infinity absorbs (0, 0) */
#define MINA_TAPE_ABSORB_FR 2      /* scalar-field element: whole if r < q, else (x >> 1) then (x & 1) */
#define MINA_TAPE_CHALLENGE 3      /* squeeze -> low 128 bits */
#define MINA_TAPE_CHALLENGE_FQ 4   /* squeeze -> full base-field element */
#define MINA_TAPE_CHALLENGE_ENDO 5 /* squeeze 128 bits -> ScalarChallenge::to_field */
#define MINA_TAPE_DIGEST 6         /* squeeze -> scalar-field element if it fits, else 0 */
#define MINA_TAPE_CHALLENGE_ENDO_OWN 7 /* squeeze 128 bits -> to_field in the sponge's own field: kimchi's Fr-sponge
                                        (`DefaultFrSponge`) = this tape on the curve whose base field is the proof's scalar field */
int mina_fq_sponge_run(mina_ctx *ctx, int curve, size_t batch, const uint8_t *tape, size_t tape_len, const uint8_t *init_state,
                       const uint32_t *init_pos, const uint8_t *inputs, uint8_t *outputs, uint8_t *final_state, uint32_t *final_pos);

/* ---- K4: group map (a14) --------------------------------------------------------------------- */
int mina_to_group(mina_ctx *ctx, int curve, size_t n, const uint8_t *t, uint8_t *out_affine);

/* ---- field self-test hooks (used by the parity tests only) ----------------------------------- */
int mina_field_mul(mina_ctx *ctx, int field, size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out);
int mina_field_inv(mina_ctx *ctx, int field, size_t n, const uint8_t *a, uint8_t *out);
int mina_field_sqrt(mina_ctx *ctx, int field, size_t n, const uint8_t *a, uint8_t *out, uint8_t *out_is_square);
/* group law: the same chain of XYZZ additions/doublings on (P_i, Q_i) with the single-lane and with the 4-lane cooperative
 * routines; both results as affine points, same[i] = 1 iff the XYZZ coordinates agreed word for word at every step */
int mina_selftest_group_law(mina_ctx *ctx, int curve, size_t n, const uint8_t *p_affine, const uint8_t *q_affine,
                            uint8_t *out_serial, uint8_t *out_quad, uint8_t *same);

/* ---- a8 / a10: combined IPA check ------------------------------------------------------------ */
/* Accumulator check (a10) for `batch` proofs on the SRS of `curve`:
 *   verdicts[b] = ( MSM(g[0..2^k), b_poly_coefficients(chals_b)) == sg_b ).
 * `prechallenges`: batch*k 128-bit values (16 bytes LE), expanded with ScalarChallenge::to_field.
 * Implemented as ONE folded MSM with the caller-supplied batching randomisers `rho` (batch field
 * elements; upstream draws them from an RNG) and a search for the culprits on failure. */
int mina_accumulator_check_batch(mina_ctx *ctx, int curve, uint32_t k, size_t batch,
                                 const uint8_t *prechallenges /* batch*k*16 */, const uint8_t *sg /* batch*64 */,
                                 const uint8_t *rho /* batch*32 */, uint8_t *verdicts /* batch */);

/* Same with everything in HBM: prechallenges (batch*k*16 B), sg (batch*64 B, canonical), rho (batch*32 B,
 * may be NULL when batch == 1).  Queued on the context stream, no host synchronisation: the u32 word at
 * `d_verdict` becomes 1 iff the (folded) check holds. */
int mina_accumulator_check_dev(mina_ctx *ctx, int curve, uint32_t k, size_t batch, const void *d_prechallenges,
                               const void *d_sg, const void *d_rho, void *d_verdict);

/* `count` (<= 64) INDEPENDENT accumulator checks -- no random folding, one u32 verdict each at d_verdicts[i] -- issued as ONE
 * kernel pipeline (the MSMs are problems of the multi-problem pipeline): the fixed per-check dispatch latency is paid once
 * per group.  prechallenges count*k*16 B, sg count*64 B, all in HBM; queued, no host synchronisation. */
int mina_accumulator_check_multi_dev(mina_ctx *ctx, int curve, uint32_t k, size_t count, const void *d_prechallenges,
                                     const void *d_sg, void *d_verdicts);
/* Host-buffer form, any count (groups of 16 per pipeline): verdicts[i] = 1 iff proof i's check holds.  Deterministic --
 * the alternative to the folded check when the caller has no randomness to hand over. */
int mina_accumulator_check_multi(mina_ctx *ctx, int curve, uint32_t k, size_t count, const uint8_t *prechallenges /* count*k*16 */,
                                 const uint8_t *sg /* count*64 */, uint8_t *verdicts /* count */);

/* Combined IPA opening check `SRS::verify` (a8).  One entry = one upstream `BatchEvaluationProof`
 * (sponge, evaluation_points, polyscale, evalscale, evaluations[].commitment, opening,
 * combined_inner_product), single-chunk commitments without degree bounds. */
typedef struct {
    uint32_t k;                 /* IPA rounds; the opening is over g[0..2^k) */
    const uint8_t *lr;          /* k pairs (L_j, R_j): 2*k*64 bytes */
    const uint8_t *delta;       /* 64 */
    const uint8_t *sg;          /* 64 */
    const uint8_t *z1, *z2;     /* 32 each (scalar field) */
    uint32_t n_evalpoints;      /* evaluation points */
    const uint8_t *evalpoints;  /* n_evalpoints * 32 (scalar field) */
    uint32_t n_comms;           /* commitments being opened */
    const uint8_t *comms;       /* n_comms * 64 */
    const uint8_t *combined_inner_product; /* 32 (scalar field) */
    const uint8_t *polyscale;   /* xi, 32 */
    const uint8_t *evalscale;   /* r, 32 */
    const uint8_t *sponge_state;/* 3 * 32: Fq-sponge state (base field) handed over by the caller ... */
    uint32_t sponge_mode;       /* ... 0 = Absorbed(count), 1 = Squeezed(count) (mina-poseidon SpongeState) */
    uint32_t sponge_count;
} mina_ipa_opening;

/* poly-commitment `combined_inner_product` (single-chunk polynomials, no degree bounds): sum_i xi^i sum_j r^j evals[i][j].
 * Host-side (needs no context): the value goes into mina_ipa_opening.combined_inner_product. */
int mina_combined_inner_product(int field, size_t n_polys, size_t n_points, const uint8_t *evals /* n_polys*n_points*32 */,
                                const uint8_t *polyscale, const uint8_t *evalscale, uint8_t *out /* 32 */);

/* The openings of one call share k and n_evalpoints; n_comms may differ (proofs of different circuits), at most 4096 per opening.
 * `lr` is a bare pointer: the caller guarantees it holds 2*k points -- k is bounded (1..20, 2^k <= SRS depth) but the length of
 * the caller's buffer cannot be checked here (the container readers, mina_verify_state, do check the proof's own L/R count).
 * The proof data is validated the way upstream's deserialiser validates it before `SRS::verify` runs: every field element
 * canonical, every point on the curve (or the all-zero encoding of infinity); a malformed opening anywhere in the batch
 * gives verdict 0, never an error code (README.md:281-310: every failure is `false`). */
int mina_ipa_batch_check(mina_ctx *ctx, int curve, size_t batch, const mina_ipa_opening *openings,
                         const uint8_t *rand_base /* 32 */, const uint8_t *sg_rand_base /* 32 */,
                         uint8_t *verdict /* 1 byte: 1 = all openings valid */);

/* ---- wire formats of the reference (a2, a3, a4) and the inclusion part of Proof-of-Account (a16) -----------------
 * Host-side, no GPU needed for the parsers.  Layouts: api_wire.hip header / SURVEY.md 8a.  Malformed input -> MINA_ERR_FORMAT. */
typedef struct {
    uint8_t is_state_proof_from_devnet;
    uint8_t bridge_tip_state_hash[32];
    uint8_t candidate_chain_state_hashes[16][32];
    uint8_t candidate_chain_ledger_hashes[16][32];
} mina_state_pub_inputs;                       /* core/src/proof/state_proof.rs:10-25, 1057 bytes on the wire */
int mina_parse_state_pub_inputs(const uint8_t *bytes, size_t len, mina_state_pub_inputs *out);
/* MinaAccountPubInputs (account_proof.rs:18-25): ledger hash + where the ABI-encoded account sits in `bytes` */
int mina_parse_account_pub_inputs(const uint8_t *bytes, size_t len, uint8_t *ledger_hash /* 32 */, size_t *encoded_offset,
                                  size_t *encoded_len);
/* merkle_path prefix of a bincode MinaAccountProof (account_proof.rs:9-14,30-35): dirs[i] = 0 Left / 1 Right */
int mina_parse_merkle_path(const uint8_t *proof, size_t len, uint32_t max_depth, uint8_t *siblings /* max_depth*32 */,
                           uint8_t *dirs /* max_depth */, uint32_t *depth, size_t *account_offset);
/* Batched inclusion check of `verify_account_inclusion` (README.md:358-362): for each i parse proof i's Merkle path and
 * pub input i's ledger hash, fold leaf_hashes[i] along the path on the GPU and compare.  The leaf (account) hash is
 * supplied by the caller: hashing the binprot account is a "next" row.  Malformed entries -> verdict 0. */
int mina_verify_account_inclusion(mina_ctx *ctx, size_t n, const uint8_t *const *proofs, const size_t *proof_lens,
                                  const uint8_t *const *pub_inputs, const size_t *pub_lens, const uint8_t *leaf_hashes /* n*32 */,
                                  uint8_t *verdicts /* n */);

/* ---- Samasika chain selection (spec: reference README.md:619-735 + img/consensus0{3,7,8}.png) -------
 * Host-side.  The verifier runs it between the candidate tip and the bridge's tip (README.md:290-294). */
#define MINA_MAX_SUB_WINDOWS 16
typedef struct {
    uint32_t slots_per_sub_window;      /* v: window shift, in slots (mainnet 7) */
    uint32_t sub_windows_per_window;    /* window length in sub-windows (mainnet 11) */
} mina_consensus_params;
typedef struct {                        /* fields of ProtocolState.body.consensus_state used by chain selection */
    uint32_t blockchain_length;
    uint32_t epoch_count;
    uint32_t curr_global_slot;
    uint32_t min_window_density;
    uint32_t sub_window_densities[MINA_MAX_SUB_WINDOWS];
    uint8_t staking_lock_checkpoint[32];   /* previous-epoch data */
    uint8_t next_lock_checkpoint[32];      /* current-epoch data */
    uint8_t last_vrf_output_hash[32];      /* digest compared lexicographically (hashLastVRF) */
    uint8_t state_hash[32];                /* hashState, compared lexicographically */
} mina_consensus_state;
int mina_consensus_project_window(const mina_consensus_params *p, const mina_consensus_state *s, uint32_t next_global_slot,
                                  uint32_t *out_window /* sub_windows_per_window entries */);
int mina_consensus_relative_min_window_density(const mina_consensus_params *p, const mina_consensus_state *a,
                                               const mina_consensus_state *b, uint32_t *out);
int mina_consensus_is_short_range(const mina_consensus_state *a, const mina_consensus_state *b);   /* 1 / 0 */
/* selectSecureChain for one candidate: *candidate_selected = 1 if the candidate replaces the current tip */
int mina_consensus_select_secure_chain(const mina_consensus_params *p, const mina_consensus_state *tip,
                                       const mina_consensus_state *candidate, int *candidate_selected);

/* ---- protocol states: `MinaHash(ProtocolState)` (core/src/mina.rs:143-168, core/src/utils/constants.rs:22-24) -------
 * A serialized `MinaStateProtocolStateValueStableV2` (bin_prot as the reference reads it at core/src/mina.rs:164, or the bincode-of-serde
 * form inside `MinaStateProof`, core/src/proof/state_proof.rs:28-41) is flattened on the host into a fixed-size RECORD of
 * MINA_PSTATE_SLOTS field elements: slot 0 = previous_state_hash, slots 1..1+n = the body's `to_input` fields
 * (`Protocol_state.Body.to_input` packed as openmina's `Inputs::to_fields`).  The GPU then computes
 *     state_hash = H_"MinaProtoState"(previous_state_hash, H_"MinaProtoStateBody"(body fields)).
 * Host-side parsers need no context. */
#define MINA_PSTATE_SLOTS 64
#define MINA_STATES_PER_PROOF 17   /* 16 candidate-chain states (oldest .. tip) + the bridge tip state (state_proof.rs:28-41) */
#define MINA_ENC_BINPROT 0
#define MINA_ENC_BINCODE 1
typedef struct {
    uint8_t previous_state_hash[32];
    uint8_t genesis_state_hash[32];
    uint8_t snarked_ledger_hash[32];       /* body.blockchain_state.ledger_proof_statement.target.first_pass_ledger */
    uint32_t n_body_fields;
    uint32_t k, slots_per_epoch, slots_per_sub_window, sub_windows_per_window, grace_period_slots, delta;   /* body.constants (+ window length) */
    mina_consensus_state consensus;        /* state_hash is left zero: fill it with the computed hash before chain selection */
} mina_protocol_state_info;
/* *consumed (may be NULL: then the state must fill `len` exactly) receives the number of bytes read */
int mina_protocol_state_pack(const uint8_t *bytes, size_t len, int encoding, uint8_t *record /* MINA_PSTATE_SLOTS*32 */,
                             uint32_t *n_body_fields, mina_protocol_state_info *info /* may be NULL */, size_t *consumed);
/* n records -> n state hashes (and, if body_hashes_out != NULL, the n body hashes) on the GPU */
int mina_protocol_state_hash_batch(mina_ctx *ctx, size_t n, const uint8_t *records /* n*MINA_PSTATE_SLOTS*32 */,
                                   const uint32_t *n_body_fields /* n */, uint8_t *hashes_out /* n*32 */, uint8_t *body_hashes_out /* n*32 or NULL */);
/* serialized states in, state hashes out */
int mina_protocol_state_hash_bytes(mina_ctx *ctx, int encoding, size_t n, const uint8_t *const *states, const size_t *lens,
                                   uint8_t *hashes_out /* n*32 */);

/* ---- the Proof-of-State job (BASELINE config C3; README.md:281-310) ------------------------------------------------
 * `batch` state proofs through ONE device pipeline: per proof the 17 protocol-state hashes + public-input comparison + chain
 * linkage, the wrap proof's public-input commitment (Pallas Lagrange MSM; its result replaces commitment `pub_comm_slot` of
 * the opening), the wrap proof's combined IPA opening (folded over the batch) and the step accumulator check (Vesta 2^acc_k
 * bases, folded with acc_rho).  Sections are structure-of-arrays over the batch; layouts as in mina_ipa_opening /
 * mina_accumulator_check_batch.  A section is skipped when its `with_*` flag (or npub) is 0. */
struct mina_kimchi_proofs;
typedef struct {
    size_t batch;
    /* (1) protocol states */
    int with_states;
    const void *state_records;    /* batch*17 records (mina_protocol_state_pack) */
    const void *state_nfields;    /* batch*17 u32 */
    const void *expected_hashes;  /* batch*17*32: candidate_chain_state_hashes[16] then bridge_tip_state_hash (state_proof.rs:10-25) */
    const void *precheck;         /* batch bytes from host-side checks (ledger hashes, consensus), NULL = all pass */
    /* (2) wrap proof public input (scalar field of Pallas) */
    uint32_t log2_domain, npub, pub_comm_slot;
    const void *public_inputs;    /* batch*npub*32 */
    /* (3) wrap proof opening (Pallas) */
    int with_ipa;
    uint32_t k, n_evalpoints, n_comms;
    const void *sponge_state /* b*96 */, *sponge_pos /* b*2 u32 {mode, count} */, *cip /* b*32 */, *lr /* b*2k*64 */, *delta /* b*64 */,
               *sg /* b*64 */, *z1, *z2 /* b*32 */, *evalpoints /* b*n_evalpoints*32 */, *evalscale, *polyscale /* b*32 */, *comms /* b*n_comms*64 */;
    const void *rand_base, *sg_rand_base;   /* 32 each */
    /* (3') instead of the derived rows above (sponge_state, sponge_pos, cip, evalpoints, evalscale, polyscale, comms): the raw wrap
     * proofs; kimchi `oracles` + `to_batch` then run on the GPU against the installed verifier index (k = its domain, n_evalpoints = 2,
     * n_comms = n_prev + 45; lr, delta, sg, z1, z2 still come from the fields above).  The struct lives on the host; its inner
     * pointers are host or device addresses like every other section. */
    const struct mina_kimchi_proofs *kimchi;
    /* (4) step accumulator (Vesta) */
    int with_accumulator;
    uint32_t acc_k;
    const void *acc_prechallenges /* b*acc_k*16 */, *acc_sg /* b*64 */, *acc_rho /* b*32 */;
} mina_state_jobs;
/* one-off set-up that needs host synchronisation (prefix salts; Lagrange basis of the domain + window table of its first npub points) */
int mina_state_jobs_prepare(mina_ctx *ctx, uint32_t log2_domain, uint32_t npub);
/* Every pointer of `jobs` is a DEVICE pointer.  Queued on the next pipeline lane, no host synchronisation.
 * d_verdicts: batch u32, 1 = proof accepted.  d_flags (may be NULL): 4 u32 {folded IPA ok, IPA input malformed, folded accumulator ok, 0};
 * when a folded check fails every verdict of the batch is 0 (the host-buffer form below then finds the culprits). */
int mina_state_job_batch_dev(mina_ctx *ctx, const mina_state_jobs *jobs, void *d_verdicts, void *d_flags);
/* SURVEY.md 8e.2 for the whole job -- several GPUs, ONE exchange step (the multi-GPU variant north_star names): this shard's proofs through every stage of the
 * job except the two fixed-base MSMs and their comparisons; the shard's folded scalar vectors and variable-base partial sums are handed to the caller instead
 * (device buffers: d_ipa_scalars 2^k * 32 B and d_ipa_point 17 words for the wrap openings on Pallas -- fixed-base part + point must be infinity;
 * d_acc_scalars 2^acc_k * 32 B and d_acc_point 17 words for the step accumulators on Vesta -- fixed-base part must equal the point).  The caller
 * exchanges the vectors (all-to-all), commits over its slice of the SRS (mina_msm_srs_range_dev) and reduces the partial points (mina_points_sum_dev):
 * mina_bridge_amd/sharded.py ShardedStateJob.  d_verdicts[b] = the per-proof checks only.  batch >= 2, with_ipa and with_accumulator set.
 * Soundness of the sum over shards: here the opening fold runs with rho_b = rand_base^(b + 1), sigma_b = sg_rand_base^(b + 1) -- NOT upstream's ^b, whose first
 * proof carries coefficient 1: partial sums of G shards are added, and G coefficient-1 proofs could cancel each other's discrepancies.  Every shard draws its own
 * rand_base / sg_rand_base from a CSPRNG after its proofs are fixed.  The accumulator fold no longer depends on the caller for that (round 6): the library multiplies
 * every acc_rho[b] by ONE scalar it draws from the OS CSPRNG per call (both sides of the shard's check scale by it), so upstream's rho_0 = 1 or fixed test
 * randomisers cannot re-open the cancellation between shards; random acc_rho (b = 0 included) remain the recommendation. */
int mina_state_job_fold_dev(mina_ctx *ctx, const mina_state_jobs *jobs, void *d_verdicts, void *d_flags, void *d_ipa_scalars, void *d_ipa_point,
                            void *d_acc_scalars, void *d_acc_point);
/* Host-buffer form: one upload, the pipeline, one download; on a folded failure the failing range is cut into four parts ($MINA_SEARCH_FAN) that are
 * re-checked concurrently (and failing parts cut again) so that every proof gets its own verdict byte; the opening check of well-formed
 * proofs is re-checked from the rows of the failed batch, without repeating the transcripts. */
int mina_state_job_batch(mina_ctx *ctx, const mina_state_jobs *jobs, uint8_t *verdicts /* batch */);

/* ---- multi-GPU building blocks --------------------------------------------------------------------
 * One process per GPU; RCCL moves bytes (all-to-all of scalar slices, all-gather of partial points), these entry points are the
 * reduction operators RCCL lacks.  Device pointers; queued on the next pipeline lane, no host synchronisation.
 * Point record: 68 bytes {x[32], y[32], u32 is_infinity} as written by mina_msm_srs_dev. */
int mina_challenge_to_field_dev(mina_ctx *ctx, int field, size_t n, const void *d_chal128 /* n*16 */, void *d_out /* n*32 */);
/* out[j] = sum_r in[r][j] mod p for `rows` vectors of m canonical field elements: the fold of the peers' scalar slices */
int mina_field_sum_rows_dev(mina_ctx *ctx, int field, size_t rows, size_t m, const void *d_in /* rows*m*32 */, void *d_out /* m*32 */);
/* sum_i scalars[i] * g[first + i] over this rank's slice of the SRS */
int mina_msm_srs_range_dev(mina_ctx *ctx, int curve, uint32_t first, size_t n, const void *d_scalars, void *d_out /* record */);
/* variable-base MSM over canonical affine points in HBM */
int mina_msm_dev(mina_ctx *ctx, int curve, size_t n, const void *d_bases /* n*64 */, const void *d_scalars /* n*32 */, void *d_out /* record */);
/* sum of n point records (the all-gathered partial results) */
int mina_points_sum_dev(mina_ctx *ctx, int curve, size_t n, const void *d_records /* n*68 */, void *d_out /* record */);
int mina_point_records_equal_dev(mina_ctx *ctx, const void *d_a, const void *d_b, void *d_verdict /* u32 */);

struct mina_pickles_statements;
/* ---- kimchi verifier for the Pickles wrap proof (a11): `oracles` + `to_batch` on the GPU ---------------------------------
 * The verifier index is DATA: the reference tree does not hold the blockchain-snark index, so the engine takes domain, shifts,
 * commitments and the linearization's constant term (a PolishToken program in the byte-code below) as a parameter. */
#define MINA_TOK_ALPHA 0
#define MINA_TOK_BETA 1
#define MINA_TOK_GAMMA 2
#define MINA_TOK_JOINT_COMBINER 3
#define MINA_TOK_ENDO_COEFFICIENT 4
#define MINA_TOK_MDS 5                    /* + u8 row, u8 col */
#define MINA_TOK_LITERAL 6                /* + 32-byte scalar */
#define MINA_TOK_CELL 7                   /* + u8 column (0 z, 1..6 selectors, 7..21 w, 22..36 coefficients, 37..42 sigma), u8 row (0 = zeta, 1 = zeta*omega) */
#define MINA_TOK_DUP 8
#define MINA_TOK_POW 9                    /* + u64 exponent */
#define MINA_TOK_ADD 10
#define MINA_TOK_MUL 11
#define MINA_TOK_SUB 12
#define MINA_TOK_VANISHES_ON_ZK_ROWS 13   /* VanishesOnZeroKnowledgeAndPreviousRows */
#define MINA_TOK_UNNORMALIZED_LAGRANGE 14 /* + i32 row offset (negative: counted back from the first zero-knowledge row; INT32_MIN: that row itself) */
#define MINA_TOK_STORE 15
#define MINA_TOK_LOAD 16                  /* + u16 cache slot */
#define MINA_TOK_SKIP_IF 17               /* + u8 feature code, u16 n: if the proof has the feature, push 0 and skip the next n tokens (kimchi SkipIf) */
#define MINA_TOK_SKIP_IF_NOT 18           /* + u8 feature code, u16 n: the same when the proof does NOT have it.  Feature codes: 0..5 the optional gates
                                             (range_check0, range_check1, foreign_field_add, foreign_field_mul, xor, rot), 6 LookupTables, 7 RuntimeLookupTables,
                                             8..11 LookupPattern Xor / Lookup / RangeCheck / ForeignFieldMul, 12 + w TableWidth(w <= 3), 16 + n LookupsPerRow(n <= 4);
                                             derived per proof from the statement's eight feature flags (csrc/polish.h).  A skipped region must net one value. */
typedef struct {
    uint32_t log2_domain;              /* evaluation domain = SRS chunk size = 2^log2_domain (wrap: 15) */
    uint32_t zk_rows;                  /* 3 */
    uint32_t perm_alpha_offset;        /* first power of alpha of the permutation argument (21) */
    const uint8_t *shifts;             /* 7 * 32, scalar field (Fq) */
    const uint8_t *sigma_comm;         /* 7 * 64 */
    const uint8_t *coefficients_comm;  /* 15 * 64 */
    const uint8_t *selector_comm;      /* 6 * 64: generic, poseidon, complete_add, mul, emul, endomul_scalar */
    const uint8_t *constant_term;      /* PolishToken byte-code of linearization.constant_term */
    size_t constant_term_len;
} mina_verifier_index;
int mina_verifier_index_install(mina_ctx *ctx, const mina_verifier_index *index);
int mina_verifier_index_digest(mina_ctx *ctx, uint8_t *out32);   /* `VerifierIndex::digest` (base-field element) */
/* File-drop forms [UPSTREAM-RECALL for every key / enum spelling; pinned by round trips against independent writers, tests/test_loaders.py]:
 * `serde_json` of kimchi's `Vec<PolishToken>` (externally tagged: "Alpha", {"Mds": {"row": r, "col": c}}, {"Literal": "<hex>"},
 * {"Cell": {"col": {"Witness": i} | "Z" | {"Index": "<gate>"} | {"Coefficient": i} | {"Permutation": i}, "row": "Curr" | "Next"}}, "Dup",
 * {"Pow": n}, "Add", "Mul", "Sub", "VanishesOnZeroKnowledgeAndPreviousRows", {"UnnormalizedLagrangeBasis": {"zk_rows": b, "offset": i}},
 * "Store", {"Load": i}, {"SkipIf" | "SkipIfNot": [<feature>, n]}; also {"Challenge": ..} / {"Constant": ..}) -> the byte-code above.
 * SkipIf / SkipIfNot: with `enabled_features` = MINA_FEATURES_RUNTIME they become MINA_TOK_SKIP_IF / _NOT and every proof's own flags decide
 * (the step index); otherwise they are resolved here against `enabled_features` (bit = feature code, see MINA_TOK_SKIP_IF_NOT; a fixed circuit
 * such as the wrap index: 0).  `optional_present`: bit i = the proofs carry optional evaluation i (wire order) -- naming another one is an
 * error -- or MINA_FEATURES_RUNTIME for "any" (the proof's presence mask decides).  `out` may be NULL to query the length.  Host-side. */
#define MINA_FEATURES_RUNTIME 0xffffffffu
int mina_polish_tokens_from_json(int field, const char *json, size_t len, uint32_t enabled_features, uint32_t optional_present, uint8_t *out,
                                 size_t cap, size_t *out_len);
/* `serde_json` of kimchi `VerifierIndex<Pallas>` (domain, zk_rows, shift, sigma_comm, coefficients_comm, generic_comm, psm_comm,
 * complete_add_comm, mul_comm, emul_comm, endomul_scalar_comm; o1-utils SerdeAs hex; 33-byte compressed points) + the constant term of its
 * linearization (`serde_json::to_string(&index.linearization.constant_term)` -- the index itself skips it) -> mina_verifier_index_install.
 * An index that enables optional gates or lookups is refused. */
int mina_verifier_index_load_json(mina_ctx *ctx, const char *index_json, size_t index_len, const char *constant_term_json, size_t ct_len,
                                  uint32_t perm_alpha_offset /* 21 for kimchi's gate set */);
typedef struct mina_kimchi_proofs {
    size_t batch;
    uint32_t n_prev, npub;             /* recursion challenges per proof (wrap: 2), public inputs per proof */
    const void *public_inputs;         /* b * npub * 32 */
    const void *prev_chals;            /* b * n_prev * k * 32 (expanded challenges, scalar field) */
    const void *prev_comms;            /* b * n_prev * 64 */
    const void *w_comm /* b*15*64 */, *z_comm /* b*64 */, *t_comm /* b*7*64 */;
    const void *evals;                 /* b * 43 * 2 * 32: column order of MINA_TOK_CELL, [zeta, zeta*omega] */
    const void *ft_eval1;              /* b * 32 */
    /* NULL, or the Pickles statements the public inputs are DERIVED from on the GPU (needs mina_step_index_install; npub must be 40
     * and public_inputs is ignored): compute_deferred_values + the message digests + the packing run ahead of `oracles`, and a
     * malformed statement fails its proof.  Host struct; inner pointers host or device like the sections above. */
    const struct mina_pickles_statements *statements;
    /* NULL, or instead of prev_chals (which may then be NULL) the 128-bit prechallenges, b * n_prev * k * 16: expanded on the GPU
     * (`ScalarChallenge::to_field` with the scalar field's endo coefficient) inside mina_state_job_batch[_dev].
     * BOTH may be NULL when `statements` is given (n_prev = 2, k = 15): the recursion challenges are then the statement's
     * messages_for_next_wrap_proof.old_bulletproof_challenges (`wrap_old_challenges`) -- the one source a verifier has -- and kimchi's
     * digest of them comes out of the statement's own sponge (15 permutations per proof that are not repeated). */
    const void *prev_prechallenges;
} mina_kimchi_proofs;
typedef struct {                       /* one `BatchEvaluationProof` row per proof, host buffers */
    uint8_t *sponge_state /* b*96 */; uint32_t *sponge_pos /* b*2 */; uint8_t *cip /* b*32 */, *evalpoints /* b*64: zeta, zeta*omega */,
            *polyscale /* b*32: v */, *evalscale /* b*32: u */, *comms /* b*(n_prev+45)*64 */, *ft_eval0 /* b*32 or NULL */;
    uint8_t *malformed;                /* 1 byte or NULL: some input was not canonical / not on the curve */
} mina_kimchi_batch_out;
int mina_kimchi_to_batch(mina_ctx *ctx, const mina_kimchi_proofs *proofs, mina_kimchi_batch_out *out);

/* ---- Pickles glue of `verify_block` (a15): statement -> deferred values -> the wrap circuit's 40 public inputs ---------------
 * openmina `compute_deferred_values` + the two message digests + `PreparedStatement::to_public_input`.  Needs, besides the wrap
 * index, the STEP circuit's data (also absent from the reference tree): zero-knowledge rows, the 7 permutation shifts of every
 * step domain in use, and the step linearization's constant term in the same PolishToken byte-code (columns: as MINA_TOK_CELL,
 * optional evaluations following at 43, 44, ... in wire order).  With a step index installed the kimchi step of mina_verify_state
 * runs on these public inputs (every statement field is then bound to the proof); without one it runs with none. */
typedef struct {
    uint32_t zk_rows;                 /* 3 */
    uint32_t n_domains;               /* step domains in use (<= 8) */
    const uint32_t *domain_log2;      /* n_domains */
    const uint8_t *shifts;            /* n_domains * 7 * 32 (Fp) */
    const uint8_t *constant_term;     /* PolishToken byte-code over Fp.  A CELL column c >= 43 names optional evaluation SLOT c - 43 (wire order);
                                         the proof's presence mask (`misc`) says where it sits among the evaluations the proof carries; naming a
                                         slot the proof lacks, outside a skipped region, fails that proof.  MINA_TOK_JOINT_COMBINER is the statement's
                                         joint combiner (endo-expanded; 0 without one); SkipIf / SkipIfNot look at the statement's feature flags. */
    size_t constant_term_len;
} mina_step_index;
int mina_step_index_install(mina_ctx *ctx, const mina_step_index *index);
/* the step side from files: one `VerifierIndex<Vesta>` JSON per step domain in use (domain, zk_rows and shift are read) + the step
 * linearization's constant term -> mina_step_index_install */
int mina_step_index_load_json(mina_ctx *ctx, size_t n_indexes, const char *const *index_jsons, const size_t *index_lens, const char *constant_term_json,
                              size_t ct_len, uint32_t enabled_features, uint32_t optional_present);
/* The statements of `batch` wrap proofs, structure-of-arrays (what `compute_deferred_values` and the two message digests read).
 * 128-bit challenges are 16 little-endian bytes; field elements 32.  Every proof of one call has the same n_old / n_evals. */
typedef struct mina_pickles_statements {
    uint32_t n_old;                     /* step-side previous accumulators per proof (0..4; Mina blockchain proof: 2) */
    uint32_t n_evals;                   /* evaluation pairs of the step proof: 43 + the optional ones present (<= 62) */
    const void *plonk;                  /* b * 64: alpha, beta, gamma, zeta */
    const void *bulletproof_challenges; /* b * 16 * 16: deferred_values.bulletproof_challenges (step, Tick rounds) */
    const void *step_old_challenges;    /* b * n_old * 16 * 16: messages_for_next_step_proof.old_bulletproof_challenges */
    const void *step_comms;             /* b * n_old * 64: messages_for_next_step_proof.challenge_polynomial_commitments */
    const void *wrap_old_challenges;    /* b * 2 * 15 * 16: messages_for_next_wrap_proof.old_bulletproof_challenges */
    const void *wrap_sg;                /* b * 64: messages_for_next_wrap_proof.challenge_polynomial_commitment */
    const void *sponge_digest;          /* b * 32: sponge_digest_before_evaluations (4 x u64, not reduced) */
    const void *prev_evals;             /* b * n_evals * 64 (Fp): kimchi column order (MINA_TOK_CELL), then the optional ones; (zeta, zeta*omega)
                                           pairs, chunked evaluations already combined with zeta^(2^16) */
    const void *prev_public_input;      /* b * 64 (Fp): the step proof's public-input evaluations */
    const void *prev_ft_eval1;          /* b * 32 (Fp) */
    const void *app_state;              /* b * 32 (Fp): the application state = hash of the tip protocol state */
    const void *misc;                   /* b * 32: [0] domain_log2, [1] proofs_verified (0..2), [2..10) feature flags, [10] has joint
                                           combiner, [11..14) little-endian presence mask of the 19 optional evaluations (wire order: 6 optional
                                           gate selectors, lookup aggregation, lookup table, 5 lookup sorted, runtime table, runtime selector,
                                           4 lookup-pattern selectors); its popcount must be n_evals - 43, [16..32) joint combiner */
} mina_pickles_statements;
/* statements -> the wrap circuit's 40 public inputs, entirely on the GPU (three sponge kernels, one scalar kernel).  Host buffers.
 * ok[b] = 0: statement b is malformed (non-canonical element, unknown step domain); its public inputs are then unspecified. */
int mina_pickles_public_inputs_batch(mina_ctx *ctx, const mina_pickles_statements *st, size_t batch, uint8_t *public_inputs_out /* b*40*32 */,
                                     uint8_t *ok /* b */);
/* one serialized wrap proof + the application state (the tip's protocol-state hash) -> public_input_out[40*32] (Fq) and, if
 * derived_out != NULL, 7*32 bytes: combined_inner_product, b, zeta^(2^16), zeta^n, perm, xi, r (Fp).  Sponges run on the GPU. */
int mina_pickles_public_input(mina_ctx *ctx, const uint8_t *wrap_proof, size_t len, int encoding, const uint8_t *app_state /* 32 */,
                              uint8_t *public_input_out, uint8_t *derived_out);

/* ---- containers (a1, a5, 8f-2) -------------------------------------------------------------------------------------
 * Serialized Pickles wrap proof (`MinaBaseProofStableV2`; bin_prot as core/src/mina.rs:235-248 reads it, or the serde/bincode form
 * at the head of `MinaStateProof`) -> one flat byte string in the fixed order the kernels consume:
 *   alpha beta gamma zeta (16 each) | has_joint_combiner (1) joint_combiner (16) | feature_flags (8) | 16 step prechallenges (16 each)
 *   | proofs_verified (1) domain_log2 (1) | sponge_digest (32) | step accumulator sg (64) | 2 x 15 wrap prechallenges
 *   | u32 n, n points | u32 n, n x 16 prechallenges | prev public-input evals | u32 count, prev evals (each: u32 n, n x 32, u32 n, n x 32)
 *   | optional-presence bytes | prev ft_eval1 | 15 w_comm, z_comm, 7 t_comm | evals w(15) coefficients(15) z s(6) selectors(6), (zeta, zeta*omega) each
 *   | ft_eval1 | u32 n, n x (L, R) | z1 z2 delta sg.
 * `out` may be NULL to query the length.  Host-side, no context. */
int mina_wrap_proof_flatten(const uint8_t *bytes, size_t len, int encoding, uint8_t *out, size_t cap, size_t *out_len, size_t *consumed /* may be NULL: exact */);
/* bincode `MinaStateProof` (core/src/proof/state_proof.rs:28-41): length of the leading proof, then where each of the 17 states sits */
int mina_state_proof_split(const uint8_t *bytes, size_t len, size_t *proof_len, size_t *state_offsets /* 17 */, size_t *state_lens /* 17 */);

/* ---- the reference-shaped boundary (README.md:275-279, 281-310, 358-362) --------------------------------
 * Same (ptr, len, ptr, len) -> bool shape as Aligned's `verify_mina_state_ffi` / `verify_account_inclusion_ffi`, fed with exactly
 * the bytes core/src/aligned.rs:31-58 produces.  Every failure is `false`; nothing unwinds; callable from any thread (one
 * process-wide context on GPU $MINA_VERIFY_DEVICE (default 0), created on first use).  Concurrent calls of the single-proof entry
 * points are merged: calls that arrive while a job runs on the GPU leave together as the next job (one proof is a 23 ms dependent
 * chain that leaves the chip idle; 256 threads calling at once see 9 k proofs/s instead of 40, each call lasting about one job).  Every caller still gets the
 * verdict of its own proof.  Batch calls of up to mina_verify_tuning.merge_batch_max (default 512) proofs share jobs with concurrent callers the same
 * way; bigger ones are pipelined on their own (chunks of 8192: parsing, uploads and the GPU job of consecutive chunks overlap; at most four jobs
 * per device in flight over all callers).  mina_verify_tuning (below): .merge = 0 sends each call through on its own; .linger_us (default 500, + 2 us per
 * caller of the previous job) bounds how long the leader of a job waits for the callers of the previous job to come back. */
#define MINA_CHECK_FORMAT 1u        /* pub inputs (1057 B) and bincode MinaStateProof parse */
#define MINA_CHECK_LEDGER 2u        /* ledger hashes of the public input == the states' snarked ledger hashes   (README.md:287) */
#define MINA_CHECK_CHAIN 4u         /* 17 state hashes == public input, states linked                          (README.md:285-288) */
#define MINA_CHECK_CONSENSUS 8u     /* candidate tip selected over the bridge tip                              (README.md:290-294) */
#define MINA_CHECK_ACCUMULATOR 16u  /* step accumulator: MSM(vesta.g, b_poly_coefficients) == challenge_polynomial_commitment */
#define MINA_CHECK_KIMCHI 32u       /* kimchi verification of the wrap proof (needs an installed verifier index) */
#define MINA_CHECK_ACCOUNT_ABI 64u  /* Proof of Account: encoded_account == ABI encoding re-derived from `account`    (README.md:349-352) */
#define MINA_CHECK_MERKLE 128u      /* Proof of Account: account hash folded along the path == ledger hash            (README.md:345-347) */
#define MINA_VERIFY_ALLOW_MISSING_KIMCHI 1u   /* verdict ignores MINA_CHECK_KIMCHI when the kimchi step cannot run: NOT a full verification */
#define MINA_VERIFY_ALLOW_UNBOUND_STATEMENT 2u /* a wrap index WITHOUT a step index: run the kimchi step with an EMPTY public input (the proof is then
                                                  not bound to the Pickles statement / the candidate tip).  Test hook; never in a deployment. */
#define MINA_VERIFY_ALLOW_SURROGATE 4u        /* verify although the installed Poseidon tables are the library's surrogate (mina_poseidon_params_name()
                                                  contains "UNPINNED"): hashes then agree with this repo's oracle, not with the Mina network.  Without the
                                                  flag every mina_verify_* verdict is `false` on such a context. */
#include <stdbool.h>
bool mina_verify_state(const uint8_t *proof, size_t proof_len, const uint8_t *pub_input, size_t pub_len);
int mina_verify_state_batch(size_t n, const uint8_t *const *proofs, const size_t *proof_lens, const uint8_t *const *pub_inputs,
                            const size_t *pub_lens, uint8_t *verdicts_out /* n bytes 0/1 */);
/* which steps ran and which passed (bit masks of MINA_CHECK_*) */
int mina_verify_state_checks(const uint8_t *proof, size_t proof_len, const uint8_t *pub_input, size_t pub_len, uint32_t *passed_mask, uint32_t *ran_mask);
/* the `--save-proof` files of the reference CLI (core/src/aligned.rs:60-69): mina_state.proof / mina_state.pub */
bool mina_verify_state_files(const char *proof_path, const char *pub_path);
bool mina_verify_account(const uint8_t *proof, size_t proof_len, const uint8_t *pub_input, size_t pub_len);
int mina_verify_account_batch(size_t n, const uint8_t *const *proofs, const size_t *proof_lens, const uint8_t *const *pub_inputs,
                              const size_t *pub_lens, uint8_t *verdicts_out);
bool mina_verify_account_files(const char *proof_path, const char *pub_path);   /* mina_account.proof / mina_account.pub */
/* The symbol names Aligned's operator binds (README.md:277-279, 358-362; proving-system tags `Mina` / `MinaAccount`, core/src/aligned.rs:40,53):
 * the library can be linked in place of the Rust verifier crate without a shim.  The caller passes its fixed-size buffers (48 KiB proof, 6 KiB public
 * input at the pinned revision: MINA_FFI_MAX_PROOF_SIZE / MINA_FFI_MAX_PUB_INPUT_SIZE) and the used lengths; a Proof-of-State length beyond the buffer
 * is `false` (the account entry reads `len` bytes whatever the buffer: its operator's buffer sizes are not in the tree).
 * Later Aligned versions pass the lengths as u32: the `_u32` forms.  Same semantics as mina_verify_state / mina_verify_account. */
#define MINA_FFI_MAX_PROOF_SIZE (48u * 1024u)
#define MINA_FFI_MAX_PUB_INPUT_SIZE (6u * 1024u)
bool verify_mina_state_ffi(const uint8_t *proof_buffer, size_t proof_len, const uint8_t *pub_input_buffer, size_t pub_input_len);
bool verify_account_inclusion_ffi(const uint8_t *proof_buffer, size_t proof_len, const uint8_t *pub_input_buffer, size_t pub_input_len);
bool verify_mina_state_ffi_u32(const uint8_t *proof_buffer, uint32_t proof_len, const uint8_t *pub_input_buffer, uint32_t pub_input_len);
bool verify_account_inclusion_ffi_u32(const uint8_t *proof_buffer, uint32_t proof_len, const uint8_t *pub_input_buffer, uint32_t pub_input_len);
int mina_verify_account_checks(const uint8_t *proof, size_t proof_len, const uint8_t *pub_input, size_t pub_len, uint32_t *passed_mask, uint32_t *ran_mask);
/* the same on a caller-owned context (n pairs, masks per pair) */
int mina_verify_account_ctx(mina_ctx *ctx, size_t n, const uint8_t *const *proofs, const size_t *proof_lens, const uint8_t *const *pub_inputs,
                            const size_t *pub_lens, uint32_t *passed_masks, uint32_t *ran_masks);
/* serialized `MinaBaseAccountBinableArgStableV2` (bin_prot as core/src/mina.rs:307-313 reads it, or the bincode form inside
 * MinaAccountProof) -> account hashes on the GPU / the ABI bytes of sol/account.rs:25-314 (host only; out may be NULL to query the length) */
int mina_account_hash_batch(mina_ctx *ctx, int encoding, size_t n, const uint8_t *const *accounts, const size_t *lens, uint8_t *hashes_out /* n*32 */);
int mina_account_abi_encode(const uint8_t *account, size_t len, int encoding, uint8_t *out, size_t cap, size_t *out_len);
int mina_verify_configure(uint32_t flags);       /* MINA_VERIFY_* */
/* Tuning: every knob of the library in ONE struct instead of environment switches (the defaults are the measured optima on one MI355X, DESIGN.md
 * section 5; tests force other shapes to prove the verdicts do not depend on them).  Process-wide; set it before the first verification or between
 * calls -- not while calls are in flight.  mina_verify_configure_ex(NULL) restores the defaults.  The environment is read only for deployment facts:
 * MINA_VERIFY_DEVICES / MINA_VERIFY_DEVICE (which GPUs), MINA_HOST_THREADS (parser pool), MINA_POSEIDON_PARAMS_FP / _FQ (table files),
 * MINA_VERIFY_TIMING (stderr timeline). */
typedef struct mina_verify_tuning {
    uint32_t struct_size;          /* sizeof(mina_verify_tuning), filled by mina_verify_tuning_default */
    /* bytes -> bools pipeline of mina_verify_state_batch (api_verify.hip run_device) */
    uint32_t chunk;                /* 8192  proofs per chunk of a call beyond `single_max` */
    uint32_t single_max;           /* 8192  calls up to this many proofs are ONE chunk */
    uint32_t slots;                /* 4     chunks in flight per device over all callers (<= 16) */
    uint32_t window;               /* 4     chunks of one call on the GPU at a time */
    uint32_t ahead;                /* 0     chunks parsed ahead of the window */
    uint32_t early_min;            /* 2048  chunks from this many entries are streamed: records uploaded and hashed run by run */
    uint32_t early_sub;            /* 1024  entries per run (0 = no streaming) */
    uint32_t head_min;             /* 6144  streamed chunks from this many entries parse their first run whole, ahead of everything else */
    uint32_t split_max;            /* 4     up to this many chunks in flight the three legs of a job fork onto three streams */
    uint32_t chain_cus;            /* 128   CUs of the wrap-proof chain's stream mask (the state hashes get the rest; 0 = no CU masks) */
    uint32_t cu_period;            /* 256   period of the mask pattern over the CU index */
    uint32_t acc_mask;             /* 0     accumulator leg under: 0 the chain's mask, 1 the hashes', 2 none */
    uint32_t hash_piece_waves;     /* 1024  state hashes are launched in pieces of this many waves */
    uint32_t up_stream;            /* 1     uploads on a stream of their own */
    uint32_t min_shard;            /* 64    fewer proofs per device than this: the call stays on one device */
    uint32_t pace_us;              /* 0     gap between the first `window` chunk issues of a long call */
    /* merging of concurrent single-proof / small-batch callers */
    uint32_t merge;                /* 1     0 = every call runs on its own */
    uint32_t merge_batch_max;      /* 512   batch calls up to this size join merged jobs */
    uint32_t linger_us;            /* 500   how long a job's leader waits for the previous job's callers to come back */
    uint32_t max_jobs;             /* 1     merged jobs overlapping on the device */
    /* lane forms of the Poseidon kernels: sponges (or proofs) in flight up to which the 16- / 8-lane latency forms are used */
    uint32_t coop16_max;           /* 64 */
    uint32_t coop8_max;            /* 8192 */
    uint32_t coop8_per_call;       /* 0     1 = the limits look at one call, not at calls x lanes in flight */
    uint32_t transcript_coop8_max; /* 0     0 = built-in rule (ctx.h use_coop8_transcripts) */
    uint32_t ipa_coop8_max;        /* 1024 */
    uint32_t kimchi_coop8_max;     /* 1024 */
    /* shortcuts, each with a slower equivalent path kept for cross-checking */
    uint32_t bpoly_mfma;           /* 1     b_poly fold of batches >= 256 on the matrix cores (0: VALU fold) */
    uint32_t pubcomm_direct;       /* 1     public-input commitments by digit-table lookups (0: bucket MSM) */
    uint32_t ipa_shared_points;    /* 1     batch-shared points of the opening check enter the MSM once */
    uint32_t kimchi_shared_digest; /* 1     kimchi's challenge digest taken from the statement's sponge */
    uint32_t ipa_side_stream;      /* 1     U = to_group(t) beside the transcript on a second stream */
    uint32_t search_fan;           /* 4     fan-out of the culprit search (2 .. 32) */
    uint32_t search_full;          /* 0     1 = every part of a culprit search repeats its transcripts */
    uint32_t msm_fp29;             /* 1     SRS-table MSMs accumulate their buckets on 29-bit limbs: 1 = points gathered from the 64-byte twin table (8 x 32 words in the 2^261 domain),
                                            2 = from the pre-split 128-byte records (x, y, p - y as 29-bit limbs: no conversion, no negation, twice the gather traffic; needs mina_srs_split_table),
                                            3 = as 1, and in the multi-MSM form the buckets STAY on 29-bit limbs through the 2-D bucket reduction; 0: the 8 x 32-bit law */
    uint32_t search_ctx;           /* 1     the culprit search of a failed chunk runs on a SECOND context of its device (its own lanes, workspaces and lock): valid traffic keeps
                                            flowing beside it.  0 = round 4: on the device's one context, with the device drained and its lock held for the length of the search */
    /* device-resident jobs (mina_state_job_batch_dev): the three independent legs of ONE job -- protocol-state hashes / wrap-proof chain / accumulator -- on streams
     * of their own, joined before the verdict kernel, so that a few jobs in flight fill the chip (round 6; until then one stream per job and ~20 jobs in flight) */
    uint32_t dev_fork;             /* 1     bit 0: fork the legs (pipelines of up to 8 lanes; a pinned lane and mina_state_job_fold_dev fork as well: fork and join are events on the lane's stream); bit 1: the chain's and the hashes' streams get disjoint CU masks (`dev_chain_cus`);
                                            bit 2: the wrap-proof chain's stream is created with the highest stream priority, the hashes' with the lowest */
    uint32_t dev_chain_cus;        /* 96    CUs of the chain's mask when bit 1 of dev_fork is set */
    uint32_t dev_piece_waves;      /* 0     a forked job's state hashes are launched in pieces of this many waves; 0 = 6144 / pipeline lanes (whole for a lone lane), 0xffffffff = never */
    uint32_t dev_hash_lds_kb;      /* 0     KiB of LDS a forked job's state-hash workgroup reserves (33 = four waves per SIMD instead of five, 41 = three: room for the waves of the other legs);
                                            0 = 41 for a lone lane, none with several lanes; 0xffffffff = never */
    uint32_t dev_acc_lane;         /* 0     the accumulator leg of a forked job: 0 = on a stream of its own, 1 = on the hashes' stream behind them, 2 = on the hashes' stream ahead of them */
} mina_verify_tuning;
void mina_verify_tuning_default(mina_verify_tuning *out);
int mina_verify_tuning_get(mina_verify_tuning *out);
int mina_verify_configure_ex(const mina_verify_tuning *tuning);
/* How many RETIRED tuning environment variables (MINA_VERIFY_CHUNK, MINA_VERIFY_NO_MERGE, MINA_COOP8_MAX ... -- the switches of rounds 1 - 3, now fields of
 * mina_verify_tuning) the process environment still sets.  They are not read; the library says so once on stderr when it is loaded, naming the field that
 * replaces each.  A strict deployment checks this for 0 at start-up. */
int mina_verify_retired_env(void);
int mina_verify_shutdown(void);                  /* destroy the process-wide contexts */
mina_ctx *mina_verify_global_ctx(void);          /* the first device's context, e.g. to install a verifier index; NULL without a GPU.  Installing through this handle with the
                                                    kernel-level installers (mina_verifier_index_install, mina_srs_*, mina_poseidon_set_params ...) takes none of the boundary's locks: do it
                                                    BEFORE the first mina_verify_* call only.  While verification calls may be in flight use the mina_verify_install_* / mina_verify_set_*
                                                    functions below: they serialise against running jobs and culprit searches (lock order g_mu, search_mu, mu) on every device. */
/* Multi-GPU behind the boundary: the process holds one context per GPU named by $MINA_VERIFY_DEVICES ("all" | comma list of
 * ordinals; an ordinal may repeat = several logical contexts on one GPU; default: $MINA_VERIFY_DEVICE or 0).  mina_verify_state_batch cuts
 * its proofs into contiguous shards, one per device, each with its own folding randomisers and its own culprit search; merged single-proof
 * jobs are dealt round-robin.  The installers below put the same data on EVERY device. */
/* which network the installed indexes belong to: 0 = mainnet, 1 = devnet, -1 (default) = not declared.  Once declared, a proof whose public
 * input claims the other network (`is_state_proof_from_devnet`, core/src/proof/state_proof.rs:10-25) fails the kimchi step. */
int mina_verify_set_network(int devnet);
int mina_verify_device_count(void);
mina_ctx *mina_verify_device_ctx(int i);
int mina_verify_install_verifier_index(const mina_verifier_index *index);
int mina_verify_install_step_index(const mina_step_index *index);
int mina_verify_set_poseidon_params(int field, const uint8_t *params /* (9+165)*32 */);
/* the Poseidon constant set compiled into the library (name contains "UNPINNED" while it is a surrogate for fp_kimchi/fq_kimchi) */
const char *mina_poseidon_params_name(void);
int mina_poseidon_install_default_params(mina_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MINA_VERIFY_H */
