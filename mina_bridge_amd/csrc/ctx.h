// ctx.h -- shared host-side state of libminaverify.so (one mina_ctx = one GPU).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/mina_verify.h"
#include "groupmap.cuh"

using namespace mb;

struct mina_ctx;
// A VIEW of a context (round 5): its own lanes, streams and workspaces, while everything INSTALLED -- field constants, both SRS with their window tables, Poseidon
// tables, salts, verifier / step index -- is borrowed from the parent (no second copy of the tables).  The boundary runs the culprit search of a failed chunk on a
// view, so that the search needs neither the parent's lock nor a drained device (api_verify.hip Device::sc).  `refresh` re-reads the parent's installs; the caller
// holds whatever serialises installs on the parent.  Destroy the view BEFORE the parent.
int mb_ctx_create_view(mina_ctx *parent, mina_ctx **out);
void mb_ctx_refresh_view(mina_ctx *view, mina_ctx *parent);
int mb_fail(int code, const std::string &msg);      // records the thread-local error text, returns code
mina_verify_tuning mb_tune();                        // the process-wide tuning (api_core.hip; mina_verify_configure_ex), by value
#define fail mb_fail

#define HIPC(expr)                                                                                 \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(MINA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));         \
    } while (0)

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    bool borrowed = false;             // the memory belongs to another context's DevBuf (a VIEW context: mb_ctx_refresh_view): never freed here; outgrown -> replaced by an own allocation
    int ensure(size_t bytes) {
        if (bytes <= cap) return MINA_OK;
        if (p) { if (!borrowed && hipFree(p) != hipSuccess) return MINA_ERR_HIP; p = nullptr; cap = 0; borrowed = false; }
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) return fail(MINA_ERR_HIP, "hipMalloc failed");
        cap = want; return MINA_OK;
    }
    void release() { if (p && !borrowed) (void)hipFree(p); p = nullptr; cap = 0; borrowed = false; }
    void alias(const DevBuf &o) { release(); p = o.p; cap = o.cap; borrowed = o.p != nullptr; }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

// page-locked host staging (grow-only): packing straight into it makes the H2D copy a real DMA at PCIe speed
struct PinnedBuf {
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return MINA_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return fail(MINA_ERR_HIP, "hipHostMalloc failed");
        cap = want; return MINA_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct SrsState {
    uint32_t depth = 0;
    uint32_t c = 16, W = 16;           // fixed-base window shape
    DevBuf table;                      // W * depth affine_t; window 0 = g itself
    DevBuf table29;                    // the same points with coordinates x * 2^261 mod p (8 x 32-bit words): what the fp29 accumulate kernels gather (ec29.cuh)
    DevBuf table29s;                   // ... pre-split: x, y, p - y as 9 x 29-bit limbs + an infinity flag, 128 B per point (msm.cuh tab29_t; mina_verify_tuning.msm_fp29 = 2)
    DevBuf h;                          // 1 affine_t
    int lagrange_log2 = -1;            // cached Lagrange basis (canonical affine bytes, host side)
    std::vector<uint8_t> lagrange_host;
    uint64_t srs_gen = 0;              // process-unique stamp of the SRS these tables were built from (api_srs.hip srs_alloc): a view context re-copies what it derived on a change
    DevBuf lagrange_table;             // window table (c = 8, W = 32) of the first lagrange_table_n basis points, for
    uint32_t lagrange_table_n = 0;     //   batched public-input commitments
    int lagrange_table_log2 = -1;
    DevBuf lagrange_digits;            // d * 2^(8w) * L_i, d = 1..128, of the first lagrange_digits_n (<= 64) basis points (lagrange.cuh: direct commitments)
    uint32_t lagrange_digits_n = 0;
    DevBuf lagrange_digits29;          // ... its 2^261-domain twin: the direct commitments' mixed adds run on 29-bit limbs (lagrange.cuh pubcomm_direct29_kernel)
};

struct MsmWorkspace {
    DevBuf scalars, points, points29 /* the 2^261-domain twin of a variable-base MSM's points (api_msm.hip mb_msm_variable) */, ekey, eval, eoff, count, start, task_start, rem_pos, rem_bucket, info, sorted, partial, heavy, order, redo, ghist, stage, buckets, red_r, red_ws, red2_r, red2_w, set_total, out_words, out_xyzz, buckets29, seg_bad;
};

// HIP-event stage timing on the context stream (off by default; bench.py turns it on for the timed region)
enum ProfStage : int { PS_DIGITS = 0, PS_SCAN, PS_SCATTER, PS_ACCUMULATE, PS_BUCKET_SUM, PS_REDUCE_A, PS_REDUCE_BC, PS_FINISH,
                       PS_BPOLY_TABLES, PS_BPOLY_FOLD, PS_BPOLY_FINISH, PS_STATE_HASH, PS_IPA_TRANSCRIPT, PS_KIMCHI, PS_PICKLES, PS_COUNT };
struct ProfState {
    int mask = 0;                                   // bit per stage; 0 = off
    struct Rec { hipEvent_t a, b; int stage; };
    std::vector<Rec> recs; size_t used = 0;
};

// A pipeline lane: one HIP stream + every scratch buffer a call needs.  Independent `_dev` calls are
// issued round-robin over the lanes so that the low-occupancy tail of one MSM overlaps the wide phases of
// the next (SRS tables, field constants and Poseidon constants are shared, read-only).
struct Lane {
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;               // second stream for work that can run beside the lane's main stream (IPA to_group)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_leg = nullptr;   // ev_leg: fork / join of the Proof-of-State job's legs
    MsmWorkspace ws;
    DevBuf tmp_a, tmp_b, tmp_c, tmp_d;       // staging for the host-buffer entry points
    PinnedBuf host_stage;                    // pinned host side of the big H2D blobs (synchronous entry points only)
    DevBuf bp_ltab, bp_htab, bp_partial, bp_ldig, bp_hdig, bp_colsum;
    DevBuf ipa_chals, ipa_folded, ipa_xyzz_a, ipa_xyzz_b, ipa_points, ipa_scalars, ipa_sigma, ipa_in_a, ipa_in_b, ipa_in_c, ipa_verdict, ipa_xfer, ipa_shared, ipa_shared_off, acc_rho_scaled /* exchange variant: the caller's acc_rho times the shard's own CSPRNG scalar */;
    DevBuf st_ok, st_hashes, st_pub_xyzz, st_pubcomm, st_flags, st_in, st_verdicts;   // Proof-of-State job (api_state.hip)
    DevBuf kc_state, kc_pos, kc_cip, kc_pts, kc_v, kc_u, kc_comms, kc_xfer, kc_pch, pk_xe, pk_pub, pk_ok;                  // kimchi to_batch output rows (api_kimchi.hip)
    void release_all() {
        MsmWorkspace &w = ws;
        DevBuf *all[] = {&w.scalars, &w.points, &w.points29, &w.ekey, &w.eval, &w.eoff, &w.count, &w.start, &w.task_start, &w.rem_pos, &w.rem_bucket, &w.info, &w.sorted, &w.partial, &w.heavy, &w.order, &w.redo, &w.ghist, &w.stage,
                         &w.buckets, &w.buckets29, &w.seg_bad, &w.red_r, &w.red_ws, &w.red2_r, &w.red2_w, &w.set_total, &w.out_words, &w.out_xyzz, &tmp_a, &tmp_b, &tmp_c, &tmp_d,
                         &bp_ltab, &bp_htab, &bp_partial, &bp_ldig, &bp_hdig, &bp_colsum, &ipa_chals, &ipa_folded, &ipa_xyzz_a, &ipa_xyzz_b, &ipa_points, &ipa_scalars,
                         &ipa_sigma, &ipa_in_a, &ipa_in_b, &ipa_in_c, &ipa_verdict, &ipa_xfer, &ipa_shared, &ipa_shared_off, &acc_rho_scaled,
                         &st_ok, &st_hashes, &st_pub_xyzz, &st_pubcomm, &st_flags, &st_in, &st_verdicts,
                         &kc_state, &kc_pos, &kc_cip, &kc_pts, &kc_v, &kc_u, &kc_comms, &kc_xfer, &kc_pch, &pk_xe, &pk_pub, &pk_ok};
        for (DevBuf *b : all) b->release();
        host_stage.release();
    }
};
static constexpr int MB_PIPE_LANES = 32;                   // lanes a caller may pipeline over (mina_ctx_set_pipeline) = the fan-out of the culprit search
static constexpr int MB_DEV_FORK_MAX = 8;                  // pipelines of up to this many lanes fork the legs of a device-resident job (mina_verify_tuning.dev_fork)
static constexpr int MB_DEV_HELPER0 = MB_PIPE_LANES + 3 * 16;   // helper lanes of pipeline lane i: MB_DEV_HELPER0 + 3 i .. (wrap-proof chain / accumulator / state hashes)
static constexpr int MB_MAX_LANES = MB_DEV_HELPER0 + 3 * MB_DEV_FORK_MAX;  // pipeline lanes + the helper lanes of the boundary's 16 slots (api_verify.hip: legs of slot s on lanes 32 + 3 s ..) + those of the forked device-resident jobs
enum : int { MB_SALT_PSTATE_BODY = 0, MB_SALT_PSTATE, MB_SALT_ACCOUNT, MB_SALT_ZKAPP_ACCOUNT, MB_SALT_ZKAPP_URI, MB_SALT_SIDE_LOADED_VK, MB_N_PREFIX_SALTS };

struct mina_ctx {
    int device = 0;
    ProfState prof;
    Lane lanes[MB_MAX_LANES];
    int nlanes = 1;
    unsigned rr = 0;                 // round-robin cursor of the `_dev` entry points
    int pinned = -1;                 // >= 0: every `_dev` entry point runs on this lane (mina_ctx_pin_lane): a caller that queues its own work on that lane's stream needs no host synchronisation
    Lane *L = nullptr;               // lane the current call runs on
    Lane *ipa_rows = nullptr; uint32_t ipa_rows_batch = 0, ipa_rows_k = 0, ipa_rows_per = 0, ipa_rows_nshared = 0; int ipa_rows_curve = -1;   // lane holding the prepared rows of the last folded opening check (mb_ipa_recheck_rows)
    FieldK fk[2];
    SrsState srs[2];
    DevBuf pparams[2]; bool have_pparams[2] = {false, false};
    DevBuf merkle_salts[2]; uint32_t merkle_depth[2] = {0, 0};   // salted initial states of the Merkle hash per height
    DevBuf kimchi_index, kimchi_tokens, kimchi_literals; bool have_kimchi = false; uint32_t kimchi_log2 = 0; uint8_t kimchi_digest[32] = {0};   // installed wrap verifier index
    uint8_t kimchi_comms_host[28 * 64] = {0};   // its commitments as installed: sigma 7, coefficients 15, selectors 6 (messages_for_next_step_proof hashes them)
    DevBuf pickles_index, pickles_tokens, pickles_literals; bool have_pickles_dev = false, pickles_ms_valid = false;   // installed step index (api_pickles.hip); ms = the Tick sponge after the wrap index commitments
    void *step_host = nullptr; void (*step_host_free)(void *) = nullptr;   // host half of the installed step index (api_pickles.hip), owned by the context
    bool pparams_surrogate[2] = {false, false};                  // the installed Poseidon tables are the library's UNPINNED surrogate set
    DevBuf state_salts; bool have_state_salts = false;           // salted initial states of the named hash prefixes (Fp): MB_SALT_*
    bool legs_forked = false;        // the job being queued runs its legs on separate streams (api_state.hip)
    bool is_view = false;            // a view of another context (mb_ctx_create_view: the culprit search's): creates no stream beyond its lane 0 -- its lanes 1 .. 3 borrow the failed chunk's, and the opening check's side stream is off
    size_t state_hashes_early = 0;   // states of the next job's protocol-state leg already queued on its lane (mb_state_hashes_early), consumed by mb_state_jobs_on_lane
    uint32_t hash_piece_waves = 0;   // > 0: the protocol-state hashes of a job are launched in pieces of this many waves (api_state.hip pstate_hash_dev)
    uint32_t hash_lds_bytes = 0;     // > 0: dynamic LDS a 3-lane state-hash workgroup reserves, to cap its waves per SIMD beside the other legs of a forked job (mina_verify_tuning.dev_hash_lds_kb)
    bool acc_first = false;          // forked device-resident job whose accumulator leg shares the hashes' stream: queue it AHEAD of them (mina_verify_tuning.dev_acc_lane = 2)
    uint32_t dev_fork_made = 0;      // the dev_fork value the helper lanes' streams were created under (streams keep their mask / priority for life)
    // SURVEY.md 8e.2 (one exchange step over several GPUs): while set, the folded checks of a job do NOT run their fixed-base MSM and comparison -- they hand out
    // this shard's folded scalar vector and the 17-word record of its variable-base partial sum instead (mina_state_job_fold_dev); device pointers
    struct FoldExport { uint32_t *ipa_scalars = nullptr, *ipa_point = nullptr, *acc_scalars = nullptr, *acc_point = nullptr; } *fold_export = nullptr;
    void use_lane0() { L = &lanes[0]; }
    void next_lane() { L = pinned >= 0 ? &lanes[pinned] : &lanes[rr++ % (unsigned)nlanes]; }
};

// lane-cooperative Poseidon: batches of at most this many sponges use 8 lanes each (shortest dependency chain, 2.6x the issue
// slots), larger ones the wave-packed 3-lane form (21 sponges per wave); both run their rounds on the 29-bit limbs (fp29.cuh).
// The per-proof transcripts of a job (kimchi, Pickles statement, opening) switch at 1024 proofs per call instead (api_kimchi.hip,
// api_pickles.hip, api_ipa.hip: measured with 16 calls in flight).
static constexpr size_t COOP8_MAX_GROUPS = 8192;   // (the default of mina_verify_tuning.coop8_max)
// `groups` sponges per call, `nlanes` calls in flight: the choice looks at the work in flight (256 proofs per call on 16 lanes are 69 k state
// hashes at once -- the 8-lane form then spends 2.1x the issue slots of a saturated chip: 480 proofs per call 66 -> 75 k/s).
// mina_verify_tuning.coop8_max overrides the limit, .coop8_per_call = 1 ignores the lanes.
static inline bool use_coop8(const mina_ctx *c, size_t groups) {
    const mina_verify_tuning t = mb_tune();
    return groups * (size_t)((c && !t.coop8_per_call) ? c->nlanes : 1) <= (size_t)t.coop8_max;
}

// One proof or a handful in flight (the reference's call pattern): 16 lanes per sponge -- the shortest dependent chain, at 5.3x the issue
// slots of the 3-lane form (poseidon_permute_hex).  `proofs` per call x lanes in flight, up to 64 (one proof 25.5 -> 23.1 ms, 16 proofs
// 22.9 -> 21.2 ms; at 256 the 8-lane form is as fast); mina_verify_tuning.coop16_max overrides (0 disables).
static inline bool use_coop16(const mina_ctx *c, size_t proofs) {
    return proofs * (size_t)(c ? c->nlanes : 1) <= (size_t)mb_tune().coop16_max;
}

// The per-proof transcripts (statement, kimchi, opening): 8 lanes per sponge while the proofs in flight (per call x lanes) leave the chip
// latency-bound (<= 1024 per call and <= 2048 in flight: with 16 lanes 1024 per call ran 123 k/s in the 8-lane form, 138 k/s in the 3-lane
// form; 512: 83 / 91 k; 256: 54 / 58 k), the wave-packed 3-lane form beyond.  mina_verify_tuning.transcript_coop8_max overrides (proofs in flight).
static inline bool use_coop8_transcripts(const mina_ctx *c, size_t batch, size_t per_call_limit) {
    const size_t lim = (size_t)mb_tune().transcript_coop8_max;
    const size_t in_flight = batch * (size_t)(c ? c->nlanes : 1);
    return lim ? in_flight <= lim : ((batch <= per_call_limit && in_flight <= 2048) || in_flight <= 2560);   // a call alone on the GPU: 1536 proofs 29.2 -> 26.8 ms, 2048: 30.2 -> 28.4, 3072: flat, 4096: +3 ms
}

// ---- the Proof-of-State job on the lanes of a context (api_state.hip); every pointer of `j` is a device pointer
enum : uint32_t { MB_JOB_LEGS = 1, MB_JOB_FINISH = 2, MB_JOB_ALL = 3 };
struct StateJobCarry { uint32_t *ipa_v = nullptr, *acc_v = nullptr, *kimchi_bad = nullptr; const uint32_t *stmt_ok = nullptr; };
int mb_state_jobs_on_lane(mina_ctx *c, const mina_state_jobs *j, uint32_t *d_verdicts, uint32_t *d_flags, Lane *LI, Lane *LA, uint32_t *d_stmt_out, Lane *LS,
                          uint32_t phase = MB_JOB_ALL, StateJobCarry *carry = nullptr);
int mb_verify_account_on(mina_ctx *c, size_t n, const uint8_t *const *proofs, const size_t *proof_lens, const uint8_t *const *pubs, const size_t *pub_lens,
                         uint32_t *passed, uint32_t *ran, Lane *lane, std::mutex *enq_mu);   // api_account.hip: Proof-of-Account on a lane of the caller's choice
int mb_state_hashes_early(mina_ctx *c, Lane *LS, size_t ns_total, size_t lo, size_t cnt, const uint32_t *d_records, const uint32_t *d_nfields, hipEvent_t after);

// ---- host-side worker pool (api_core.hip): persistent threads, created on first use -- min(hardware threads / 2, 64), $MINA_HOST_THREADS
// overrides.  A job is `n` independent items handed out in index order; `mb_pool_submit` returns at once (the boundary's pipeline parses
// chunk i + 1 while chunk i is on the GPU), `mb_pool_wait` joins in on the remaining items and returns when all are done.
struct MbPoolJob;
std::shared_ptr<MbPoolJob> mb_pool_submit(size_t n, std::function<void(size_t)> fn);
void mb_pool_wait(const std::shared_ptr<MbPoolJob> &job);
size_t mb_pool_threads();
// independent per-item host work (items are ~0.01 - 0.1 ms each: the pool only when there are enough of them)
template <class Fn> static inline void mb_parallel_for(size_t n, Fn fn) {
    if (n < 128 || mb_pool_threads() <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    mb_pool_wait(mb_pool_submit(n, std::function<void(size_t)>(fn)));
}
// `n` bytes from the operating system's CSPRNG (getrandom(2), /dev/urandom as the fallback); false = none available: callers fail closed
bool mb_secure_random(void *buf, size_t n);

static inline int base_field_of(int curve) { return curve == CURVE_PALLAS ? FIELD_FP : FIELD_FQ; }
static inline int scalar_field_of(int curve) { return curve == CURVE_PALLAS ? FIELD_FQ : FIELD_FP; }
static inline bool bad_field(int f) { return f != 0 && f != 1; }

#define DISPATCH_FIELD(field, ...)                           \
    do { if ((field) == FIELD_FP) { constexpr int F_ = FIELD_FP; __VA_ARGS__; } else { constexpr int F_ = FIELD_FQ; __VA_ARGS__; } } while (0)

static inline uint32_t cdiv(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }


// small host<->device helpers for the byte-buffer entry points
static inline int h2d(mina_ctx *c, DevBuf &b, const void *src, size_t bytes) {
    int rc = b.ensure(bytes ? bytes : 4);
    if (rc) return rc;
    if (bytes) HIPC(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->L->stream));
    return MINA_OK;
}
static inline int d2h_sync(mina_ctx *c, void *dst, const DevBuf &b, size_t bytes) {
    if (bytes) HIPC(hipMemcpyAsync(dst, b.p, bytes, hipMemcpyDeviceToHost, c->L->stream));
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(c->L->stream));
    return MINA_OK;
}

// The signed-digit recoding of the MSM has no window for a carry out of bit 255: scalars must be < 2^255 (every canonical field
// element is).  The host-buffer entry points reject anything else instead of returning a point that is off by 2^256 * P.
static inline bool scalars_below_2_255(const uint8_t *scalars, size_t n) { for (size_t i = 0; i < n; ++i) if (scalars[i * 32 + 31] & 0x80) return false; return true; }

void mb_prof_begin(mina_ctx *c, int stage);
void mb_prof_end(mina_ctx *c, int stage);
struct ProfScope {
    mina_ctx *c; int stage;
    ProfScope(mina_ctx *c_, int s_) : c(c_), stage(s_) { if (c->prof.mask & (1 << stage)) mb_prof_begin(c, stage); }
    ~ProfScope() { if (c->prof.mask & (1 << stage)) mb_prof_end(c, stage); }
};

// combined IPA opening check with inputs in HBM (api_ipa.hip)
namespace mb {
struct IpaShape { uint32_t batch, k, npts, ncomms, per; uint32_t override_slot = 0xffffffffu, expand_slot = 0xffffffffu;
                  uint32_t shared_lo = 0, shared_hi = 0, shared_h = 0, shared_expand0 = 0, nshared = 0;
                  uint32_t pow_first = 0; };   // per = 2k + ncomms + 4 points per proof
// pow_first: rho_b = rand_base^(b + pow_first), sigma_b = sg_rand_base^(b + pow_first).  0 = upstream's batch_verify (proof 0 has coefficient 1: fine when the
//   batch is ONE combination); 1 = the exchange variant (mina_state_job_fold_dev): the shards' partial sums are ADDED by the caller, so every coefficient of
//   every shard must be random -- G shards whose first proofs all carry coefficient 1 let two first proofs with discrepancies +tH / -tH cancel
// shared_*: list entries whose POINT is the same for every proof of the batch (the caller vouches: h, verifier-index commitments -- bits of
//   shared_lo/hi over the commitment index, ncomms <= 64 --, the index point of the expanded slot).  Their scalars are summed over the batch
//   first and enter the MSM ONCE, as `nshared` entries behind the per-proof lists (a third of a kimchi batch's 88 points per proof);
//   the per-proof entries keep a zero scalar (the MSM drops zero digits) and the real one in a side matrix (culprit search restores it)
// override_slot: commitment index whose point comes from `comm_override` (b*16 canonical words, e.g. the public-input commitment computed on the GPU)
// expand_slot: commitment index given as a linear combination instead of a point (kimchi's chunked ft commitment): its list entry becomes
//   IPA_EXPAND entries (point_j, weight * expand_sc[j]), so the combination is evaluated by the batch MSM; per grows by IPA_EXPAND - 1
static constexpr uint32_t IPA_EXPAND = 8;
struct IpaExpand {    // point 0 = *p0 (Montgomery, trusted index data), points 1..7 = pts + b*7*16 (canonical words, checked); scalars sc[b*stride + j] (Montgomery)
    const affine_t *p0 = nullptr; const uint32_t *pts = nullptr; const fe_t *sc = nullptr; uint32_t stride = 0;
};
struct IpaDevIn {     // structure-of-arrays over the batch, canonical little-endian words; layouts as in mina_ipa_opening
    const uint32_t *state /* b*24 */, *pos /* b*2 */, *cip /* b*8 */, *lr /* b*2k*16 */, *delta /* b*16 */, *sg /* b*16 */, *z1, *z2 /* b*8 */,
                   *pts /* b*npts*8 */, *r /* b*8 */, *xi /* b*8 */, *comms /* b*ncomms*16 */, *comm_override /* b*16 or null */, *rb /* 8 */, *sb /* 8 */;
    IpaExpand expand;
};
}
int mb_ipa_batch_check_dev(mina_ctx *c, int curve, mb::IpaShape sh, const mb::IpaDevIn &in, uint32_t *d_verdict /* [0] verdict, [1] malformed flag */);
int mb_ipa_recheck_rows(mina_ctx *c, size_t lo, size_t cnt, uint32_t *d_verdict /* [0] verdict */);   // folded check of proofs [lo, lo + cnt) of the batch prepared last, from its rows; on the current lane
int mb_accumulator_check_dev(mina_ctx *c, int curve, uint32_t k, size_t batch, const uint32_t *d_prechal, const uint32_t *d_sg_words, const uint32_t *d_rho, uint32_t *d_verdict);

// cross-file entry points (C++ linkage)
struct xyzz_dev;   // opaque: mb::xyzz_t in HBM
// fixed-base MSM over the SRS table of `curve`; writes the 17-word affine record and/or the XYZZ value
int mb_msm_fixed(mina_ctx *c, int curve, uint32_t n, const uint32_t *d_scalars, uint32_t *d_out_words, void *d_out_xyzz, uint32_t first = 0);
// variable-base MSM over Montgomery affine points already in HBM
// K2 fold on the current lane (no lane switch)
int mb_msm_table(mina_ctx *c, int curve, const void *d_table, uint32_t stride, uint32_t cbits, uint32_t W, uint32_t first, uint32_t n,
                 uint32_t nprob, const uint32_t *d_scalars, uint32_t *d_out_words, void *d_out_xyzz);
int mb_bpoly_single_from_prechallenges(mina_ctx *c, int field, uint32_t k, const uint32_t *d_prechal, uint32_t *d_out, uint32_t count);
int mb_bpoly_fold(mina_ctx *c, int field, uint32_t k, size_t batch, const uint32_t *d_chals, const uint32_t *d_weights, uint32_t *d_out);
int mb_msm_variable(mina_ctx *c, int curve, uint32_t n, const uint32_t *d_scalars, const void *d_points_mont,
                    uint32_t *d_out_words, void *d_out_xyzz);
