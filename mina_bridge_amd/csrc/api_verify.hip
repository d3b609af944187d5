// api_verify.hip -- the reference-shaped boundary: what Aligned's operator / batcher binds.
//
//   bool mina_verify_state  (proof, len, pub, len)    <->  `verify_mina_state_ffi`          (README.md:275-279, 281-310)
//   bool mina_verify_account(proof, len, pub, len)    <->  `verify_account_inclusion_ffi`   (README.md:358-362)
// with the bytes `core/src/aligned.rs:31-58` produces (`bincode::serialize` of `MinaStateProof` / `MinaStatePubInputs`,
// `MinaAccountProof` / `MinaAccountPubInputs`) -- also the content of the `--save-proof` files `mina_state.proof/.pub`,
// `mina_account.proof/.pub` (core/src/aligned.rs:60-69).  Every failure is `false`; nothing unwinds; a process-wide lazily
// created context (GPU 0 or $MINA_VERIFY_DEVICE) with both SRS and the Poseidon tables serves all callers under one mutex.
//
// Steps of mina_verify_state (README.md:281-310), and what runs where:
//   FORMAT      host   pub inputs = 1057 bytes, proof = bincode MinaStateProof (wire_proof.h / wire_state.h)
//   LEDGER      host   pub.candidate_chain_ledger_hashes[i] == states[i] snarked ledger hash          (README.md:287)
//   CHAIN       GPU    17 x MinaHash(state) == pub hashes, state i+1 names state i                    (README.md:285-288)
//   CONSENSUS   host   candidate tip selected over the bridge tip by `select_secure_chain`            (README.md:290-294)
//   ACCUMULATOR GPU    MSM(vesta.g, b_poly_coefficients(step bulletproof challenges)) == challenge_polynomial_commitment
//   KIMCHI      GPU    kimchi::verifier::verify of the wrap proof -- needs the blockchain-snark verifier index, which the
//                      reference tree does not hold; it runs when an index has been installed (mina_verifier_index_install),
//                      otherwise the step cannot run and mina_verify_state answers `false` (mina_verify_state_checks tells
//                      which steps ran and passed; MINA_VERIFY_ALLOW_MISSING_KIMCHI relaxes the verdict for integration tests).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "ctx.h"
#include "sponge.cuh"
#include "wire_proof.h"
#include "poseidon_tables.inc"

int mb_pack_protocol_state(const mw::ProtocolState &s, uint8_t *record, uint32_t *n_body_fields, mina_protocol_state_info *info);   // api_state.hip
int mb_kimchi_available(mina_ctx *c);                                                                                                  // api_kimchi.hip
int mb_poseidon_env_params(mina_ctx *c);                                                                                               // api_loaders.hip
int mb_step_index_feature_aware(mina_ctx *c);                                                                                         // api_pickles.hip
int mb_step_index_installed(mina_ctx *c);                                                                                              // api_pickles.hip

extern "C" const char *mina_poseidon_params_name(void) { return MB_POSEIDON_SET_NAME; }
// the compiled-in tables are a surrogate while their name says UNPINNED: a context running on them is flagged (mina_verify_state refuses)
bool mb_params_are_surrogate(int field, const uint8_t *params) {
    return strstr(MB_POSEIDON_SET_NAME, "UNPINNED") != nullptr && (field == 0 || field == 1) && memcmp(params, MB_POSEIDON_TABLES[field], (9 + 165) * 32) == 0;
}
extern "C" int mina_poseidon_install_default_params(mina_ctx *c) {
    if (!c) return fail(MINA_ERR_ARG, "null argument");
    for (int f = 0; f < 2; ++f) { int rc = mina_poseidon_set_params(c, f, MB_POSEIDON_TABLES[f]); if (rc) return rc; }
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------ process-wide devices
// One context per GPU the process may use: $MINA_VERIFY_DEVICES = "all" | comma list of device ordinals (an ordinal may repeat: several
// logical contexts on one GPU, the test hook for the sharding code on a 1-GPU box); default: GPU $MINA_VERIFY_DEVICE, or 0.  Every
// context holds both SRS, the Poseidon tables and -- once installed through mina_verify_install_* -- the verifier / step index.
// mina_verify_state_batch cuts its proofs into contiguous shards, one per device (zero exchange, verdict bytes gathered
// on the host); merged single-proof jobs are dealt round-robin.
namespace {
constexpr int NSLOT = 16;                  // most chunks in flight per device ($MINA_VERIFY_SLOTS of them in use: 4): slot s runs on lane s of the context (helper lanes MB_PIPE_LANES + 3 s .. for its forked legs)
// `up`: the slot's upload stream.  No copy of the pipeline waits for a kernel: the copy engines take their commands in order, and one that waits for a
// kernel of its stream holds up the uploads of every other chunk queued behind it (calls of 65 536 proofs: the jobs of the 8 chunks started up to 300 ms
// apart, rocprofv3 timeline).  So uploads have a stream of their own, and the verdict words go back through a kernel that writes the page-locked buffer.
struct Slot { PinnedBuf host, out; DevBuf dev; hipStream_t up = nullptr; hipEvent_t ev = nullptr, ev_up = nullptr; hipEvent_t tev[3] = {nullptr, nullptr, nullptr}; std::vector<hipEvent_t> rec_ev; bool busy = false; };
struct Device {
    mina_ctx *c = nullptr; int ordinal = 0;
    std::mutex mu;                         // serialises every call into `c` (a context has ONE current-lane cursor)
    // The culprit search of a chunk whose folded check failed (mina_state_job_batch: synchronous jobs fanned over a few lanes) has a context of its OWN on the same
    // GPU (round 5): until then it ran on `c` with the device drained and `mu` held for its whole length -- one bad proof from any caller stalled every
    // other caller of the device (measured: a clean caller of 8192 proofs kept 51 of its 117 k proofs/s beside a caller whose every call carried one bad opening).
    // A VIEW of `c` (ctx.h mb_ctx_create_view): own lanes and workspaces, tables and indexes borrowed.  Locked by `search_mu`: searches queue behind each other,
    // traffic does not; the installers take `search_mu` before `mu` so that no search reads a buffer they replace.
    mina_ctx *sc = nullptr; std::mutex search_mu;
    std::mutex acct_mu;                    // one Proof-of-Account job at a time (its lane's buffers)
    std::mutex slot_mu; std::condition_variable slot_cv; Slot slots[NSLOT];
    uint32_t prepared_npub = 0xffffffffu;
    std::atomic<unsigned> inflight{0};
};
std::mutex g_mu;                           // guards the device list, the flags and set-up
std::vector<Device *> g_devs; bool g_init_failed = false;
uint32_t g_flags = 0;
int g_network = -1;                        // which network the installed indexes belong to: -1 = not declared, 0 = mainnet, 1 = devnet (mina_verify_set_network)
std::atomic<unsigned> g_rr{0};

int create_device(int ordinal, Device **out) {
    mina_ctx *c = nullptr;
    int rc = mina_ctx_create(ordinal, &c);
    if (rc) return rc;
    if ((rc = mina_poseidon_install_default_params(c)) || (rc = mb_poseidon_env_params(c)) || (rc = mina_srs_create(c, CURVE_VESTA, 1u << 16)) || (rc = mina_srs_create(c, CURVE_PALLAS, 1u << 16))) { mina_ctx_destroy(c); return rc; }
    mina_ctx *sc = nullptr;                                  // a VIEW of c (ctx.h): own lanes and workspaces, the tables and indexes borrowed -- no second copy
    if ((rc = mb_ctx_create_view(c, &sc))) { mina_ctx_destroy(c); return rc; }
    Device *d = new Device(); d->c = c; d->sc = sc; d->ordinal = ordinal;
    *out = d;
    return MINA_OK;
}
std::vector<Device *> &devices() {          // caller holds g_mu
    if (!g_devs.empty() || g_init_failed) return g_devs;
    std::vector<int> ords;
    if (const char *e = getenv("MINA_VERIFY_DEVICES")) {
        if (!strcmp(e, "all")) { int n = 0; if (hipGetDeviceCount(&n) == hipSuccess) for (int i = 0; i < n; ++i) ords.push_back(i); }
        else for (const char *p = e; *p;) { char *q; const long v = strtol(p, &q, 10); if (q == p) break; ords.push_back((int)v); p = *q == ',' ? q + 1 : q; }
    }
    if (ords.empty()) ords.push_back(getenv("MINA_VERIFY_DEVICE") ? atoi(getenv("MINA_VERIFY_DEVICE")) : 0);
    for (int o : ords) {
        Device *d = nullptr;
        if (create_device(o, &d) != MINA_OK) { for (Device *x : g_devs) { mina_ctx_destroy(x->sc); mina_ctx_destroy(x->c); delete x; } g_devs.clear(); g_init_failed = true; break; }
        g_devs.push_back(d);
    }
    return g_devs;
}
void destroy_devices() {                    // caller holds g_mu
    for (Device *d : g_devs) {
        { std::lock_guard<std::mutex> sl(d->search_mu);       // lock order: search_mu before mu (the installers', the fallback's)
          std::lock_guard<std::mutex> lk(d->mu);
          (void)hipSetDevice(d->c->device);
          for (Slot &s : d->slots) { if (s.ev) { (void)hipEventSynchronize(s.ev); (void)hipEventDestroy(s.ev); } for (auto &e : s.tev) if (e) (void)hipEventDestroy(e); for (auto &e : s.rec_ev) if (e) (void)hipEventDestroy(e); s.rec_ev.clear(); if (s.ev_up) (void)hipEventDestroy(s.ev_up); if (s.up) (void)hipStreamDestroy(s.up); s.ev_up = nullptr; s.up = nullptr; s.host.release(); s.out.release(); s.dev.release(); }
          mina_ctx_destroy(d->sc);                            // the view first: it borrows c's buffers
          mina_ctx_destroy(d->c); }
        delete d;
    }
    g_devs.clear(); g_init_failed = false;
}
}  // namespace

extern "C" int mina_verify_configure(uint32_t flags) { std::lock_guard<std::mutex> lk(g_mu); g_flags = flags; return MINA_OK; }
// `is_state_proof_from_devnet` of the public input (core/src/proof/state_proof.rs:10-25) selects, upstream, the devnet or the mainnet
// verifier index.  This library holds ONE index pair: declare which network it belongs to and proofs that claim the other one are rejected
// at the kimchi step (their own verdict only); -1 (the default) = not declared, the flag is not looked at.
extern "C" int mina_verify_set_network(int devnet) { if (devnet < -1 || devnet > 1) return fail(MINA_ERR_ARG, "network must be -1, 0 or 1"); std::lock_guard<std::mutex> lk(g_mu); g_network = devnet; return MINA_OK; }
extern "C" int mina_verify_shutdown(void) { std::lock_guard<std::mutex> lk(g_mu); destroy_devices(); return MINA_OK; }
// the first device's context (a process with one GPU: THE context), e.g. to install a verifier index or other Poseidon tables; NULL if no
// GPU / set-up failed.  Install before the first verification: installing is not synchronised against calls in flight.
extern "C" mina_ctx *mina_verify_global_ctx(void) { std::lock_guard<std::mutex> lk(g_mu); auto &d = devices(); return d.empty() ? nullptr : d[0]->c; }
extern "C" int mina_verify_device_count(void) { std::lock_guard<std::mutex> lk(g_mu); return (int)devices().size(); }
extern "C" mina_ctx *mina_verify_device_ctx(int i) { std::lock_guard<std::mutex> lk(g_mu); auto &d = devices(); return (i < 0 || (size_t)i >= d.size()) ? nullptr : d[(size_t)i]->c; }
// the same data on EVERY device of the process (what a multi-GPU deployment calls instead of the per-context installers)
extern "C" int mina_verify_install_verifier_index(const mina_verifier_index *ix) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto &ds = devices(); if (ds.empty()) return fail(MINA_ERR_HIP, "no device");
    for (Device *d : ds) {
        std::lock_guard<std::mutex> sl(d->search_mu);      // no culprit search is reading the buffers being replaced (lock order: g_mu, search_mu, mu)
        std::lock_guard<std::mutex> dl(d->mu); int rc = mina_verifier_index_install(d->c, ix); if (rc) return rc; d->prepared_npub = 0xffffffffu;
    }
    return MINA_OK;
}
extern "C" int mina_verify_install_step_index(const mina_step_index *ix) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto &ds = devices(); if (ds.empty()) return fail(MINA_ERR_HIP, "no device");
    for (Device *d : ds) {
        std::lock_guard<std::mutex> sl(d->search_mu);      // no culprit search is reading the buffers being replaced (lock order: g_mu, search_mu, mu)
        std::lock_guard<std::mutex> dl(d->mu); int rc = mina_step_index_install(d->c, ix); if (rc) return rc;
    }
    return MINA_OK;
}
extern "C" int mina_verify_set_poseidon_params(int field, const uint8_t *params) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto &ds = devices(); if (ds.empty()) return fail(MINA_ERR_HIP, "no device");
    for (Device *d : ds) {
        std::lock_guard<std::mutex> sl(d->search_mu);      // no culprit search is reading the buffers being replaced (lock order: g_mu, search_mu, mu)
        std::lock_guard<std::mutex> dl(d->mu); int rc = mina_poseidon_set_params(d->c, field, params); if (rc) return rc;
    }
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------ merging of concurrent single-proof calls
// The reference's entry points take ONE proof and are called from many goroutines / tokio tasks at once.  One proof is
// a 25 ms dependent chain that leaves the chip idle, so concurrent callers are merged (group commit): the first caller runs a job with
// everything queued at that moment; calls arriving while it runs wait and leave together as the next job, led by one of them.  A lone
// caller pays nothing; N concurrent callers share one job of N proofs.  (mina_verify_tuning.max_jobs lets that many merged jobs overlap on the
// device; measured with 256 calling threads: 1 job at a time 4.7 k proofs/s, 2: 3.8 k, 4: 2.4 k -- small jobs are latency-bound chains that
// slow each other down and split the batches, so the default stays 1.)  Verdicts are per proof either way: the folded checks of a job use
// randomisers drawn from the operating system's CSPRNG after every proof of the job is fixed (as upstream's `batch_verify` draws its own),
// so one caller's proof cannot be built to cancel another's.  mina_verify_tuning.merge = 0 sends every call through on its own;
// .linger_us (default 500) is how long the leader of a job waits for the callers of the previous job to come back before it leaves.
namespace {
struct PendingCall { const uint8_t *proof; size_t proof_len; const uint8_t *pub; size_t pub_len; uint8_t verdict = 0; bool done = false, claimed = false; int rc = MINA_OK; size_t owner = 0; };
typedef void (*exec_fn_t)(std::vector<PendingCall *> &job);           // sets `verdict` of every call of the job
struct CallMerger {
    std::mutex mu; std::condition_variable cv, arrived; std::vector<PendingCall *> waiting;
    bool collecting = false;             // a leader is gathering its job (the linger): arrivals join it instead of leading jobs of their own
    size_t active = 0;                   // jobs running: up to MAX_ACTIVE overlap on the device (the pipeline's slots; a job is a latency-bound chain)
    size_t last_job = 0;                 // calls merged into the previous job: its callers return together and call again within microseconds
    std::vector<std::pair<size_t, size_t>> last_owners;   // ... by calling thread
    static constexpr size_t MAX_JOB = 8192;
    bool run(exec_fn_t exec, PendingCall &me) { run_group(exec, &me, 1); return me.verdict == 1; }
    // the calls of one caller -- one proof (the reference's entry point) or a small batch -- wait here for a job to take them
    void run_group(exec_fn_t exec, PendingCall *mine, size_t count) {
        const mina_verify_tuning tune = mb_tune();
        if (!tune.merge || count == 0) { std::vector<PendingCall *> job; for (size_t i = 0; i < count; ++i) job.push_back(&mine[i]); if (count) exec(job); return; }
        const long linger_us = (long)tune.linger_us;
        const size_t max_active = std::max<size_t>(1, tune.max_jobs);
        auto all_done = [&] { for (size_t i = 0; i < count; ++i) if (!mine[i].done) return false; return true; };
        auto all_claimed = [&] { for (size_t i = 0; i < count; ++i) if (!mine[i].claimed) return false; return true; };
        const size_t me = std::hash<std::thread::id>()(std::this_thread::get_id());
        std::unique_lock<std::mutex> lk(mu);
        for (size_t i = 0; i < count; ++i) { mine[i].owner = me; waiting.push_back(&mine[i]); }
        arrived.notify_all();
        while (!all_done()) {
            if (all_claimed() || collecting || active >= max_active) { cv.wait(lk); continue; }   // my calls are in jobs / a leader is gathering / every job slot is taken: woken on every change
            ++active; collecting = true;                                      // lead the next job: everything queued by the time it leaves (these calls included)
            size_t foreign = last_job;                                        // calls of OTHER threads in the job that just ended (a caller alone with the library waits for nobody)
            { auto it = std::lower_bound(last_owners.begin(), last_owners.end(), std::make_pair(me, (size_t)0)); if (it != last_owners.end() && it->first == me) foreign -= std::min(foreign, it->second); }
            if ((foreign > 0 || active > 1) && linger_us > 0) {               // other callers are about: those of the job that just ended are on their way back (in
                // ADDITION to the ones that queued up while it ran -- leaving without them splits the callers into two groups that take turns, each call
                // then lasting two jobs: 16 threads saw 43 ms per call, 23 ms once the leader waits), and while another job runs a moment's wait costs nothing
                // (waking N threads and getting them back here takes longer the more there are: 2 us per caller of the last job on top, 4 ms at most)
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(linger_us + (long)std::min<size_t>(2 * foreign, 4000));
                const size_t expect = waiting.size() + std::max<size_t>(foreign, 1);
                while (waiting.size() < expect && arrived.wait_until(lk, deadline) != std::cv_status::timeout) {}
            }
            const size_t n = std::min(waiting.size(), MAX_JOB);
            std::vector<PendingCall *> job(waiting.begin(), waiting.begin() + n);
            waiting.erase(waiting.begin(), waiting.begin() + n);
            for (PendingCall *p : job) p->claimed = true;
            collecting = false;
            cv.notify_all();                                                  // whoever arrives from now on may lead the next job
            lk.unlock();
            exec(job);
            lk.lock();
            for (PendingCall *p : job) p->done = true;
            last_job = n;
            last_owners.clear();
            { std::vector<size_t> ow; ow.reserve(n); for (PendingCall *p : job) ow.push_back(p->owner); std::sort(ow.begin(), ow.end());
              for (size_t i = 0; i < ow.size();) { size_t j = i; while (j < ow.size() && ow[j] == ow[i]) ++j; last_owners.push_back({ow[i], j - i}); i = j; } }
            --active;
            cv.notify_all();
        }
    }
};
CallMerger g_state_calls, g_account_calls;
}  // namespace

// verdict words of a job, device staging -> the slot's page-locked buffer (see Slot::up)
__global__ void words_out_kernel(uint32_t n, const uint32_t *__restrict__ src, uint32_t *__restrict__ dst) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i]; }
// (the one kernel launch of this file, behind a function: the ThreadSanitizer tier builds the rest of the file -- pure host logic -- with g++ against a stand-in
// runtime, tests/fuzz/hip_stub + tests/fuzz/tsan_boundary.cpp, where "device" memory is host memory and the copy is a memcpy)
static inline void launch_words_out(hipStream_t st, uint32_t words, const uint32_t *src, uint32_t *dst) {
#if defined(MB_HIP_STUB)
    (void)st; memcpy(dst, src, (size_t)words * 4);
#else
    words_out_kernel<<<(words + 255) / 256, 256, 0, st>>>(words, src, dst);
#endif
}

// ------------------------------------------------------------------------------------------------ Proof of State
// bytes -> bools, pipelined (core/src/aligned.rs:31-58 builds the bytes; core/src/proof/state_proof.rs:10-41 their layout):
//   a call is cut into chunks of 8192 proofs; the host pool parses a chunk STRAIGHT INTO the page-locked structure-of-arrays staging of its
//   slot, in two halves -- the wrap proofs, then the 17 protocol states per proof (`to_input` flattening, ledger / consensus checks) -- and each
//   half goes to the GPU as soon as it exists (issue_legs / stream_records / finish in run_device; mb_state_jobs_on_lane: no host synchronisation
//   inside); up to four chunks are on the GPU at a time, verdict words come back through page-locked memory.  No per-call allocation, no gather
//   copy.  A chunk whose folded check fails goes through the culprit search of mina_state_job_batch from the same staging.
namespace {
struct Shape { bool kimchi = false, statements = false, feature_aware = false; uint32_t k = 0, n_prev = 2, n_old = 0, n_ev = 0; int network = -1; };
enum Sec : int { S_REC = 0, S_NF, S_EXP, S_PRE, S_APRE, S_ASG, S_ARHO, S_LR, S_DELTA, S_SG, S_Z1, S_Z2, S_PCM, S_WC, S_ZC, S_TC, S_EV, S_FT1, S_PCH,
                 S_PLONK, S_BP, S_OLD, S_CM, S_WOLD, S_WSG, S_DG, S_SEV, S_PI, S_SFT, S_APP, S_MISC, S_RB, S_SB, NSEC };
struct Layout {
    size_t stride[NSEC] = {0}, off[NSEC] = {0}, total = 0, cap = 0;
    void build(const Shape &sh, size_t cap_) {
        cap = cap_;
        for (size_t &x : stride) x = 0;
        stride[S_REC] = (size_t)MINA_STATES_PER_PROOF * MINA_PSTATE_SLOTS * 32; stride[S_NF] = MINA_STATES_PER_PROOF * 4; stride[S_EXP] = MINA_STATES_PER_PROOF * 32; stride[S_PRE] = 1;
        stride[S_APRE] = 16 * 16; stride[S_ASG] = 64; stride[S_ARHO] = 32;
        if (sh.kimchi) {
            stride[S_LR] = 2 * (size_t)sh.k * 64; stride[S_DELTA] = 64; stride[S_SG] = 64; stride[S_Z1] = 32; stride[S_Z2] = 32;
            stride[S_PCM] = sh.n_prev * 64; stride[S_WC] = 15 * 64; stride[S_ZC] = 64; stride[S_TC] = 7 * 64; stride[S_EV] = 43 * 64; stride[S_FT1] = 32;
            if (!sh.statements || sh.k != 15 || sh.n_prev != 2) stride[S_PCH] = (size_t)sh.n_prev * sh.k * 16;   // else the statement's wrap_old_challenges ARE the recursion challenges
            if (sh.statements) {
                stride[S_PLONK] = 64; stride[S_BP] = 256; stride[S_OLD] = (size_t)sh.n_old * 256; stride[S_CM] = (size_t)sh.n_old * 64; stride[S_WOLD] = 480; stride[S_WSG] = 64;
                stride[S_DG] = 32; stride[S_SEV] = (size_t)sh.n_ev * 64; stride[S_PI] = 64; stride[S_SFT] = 32; stride[S_APP] = 32; stride[S_MISC] = 32;
            }
        }
        total = 0;
        for (int i = 0; i < NSEC; ++i) {
            off[i] = total;
            const size_t bytes = (i == S_RB || i == S_SB) ? (sh.kimchi ? 32 : 0) : stride[i] * cap;
            total += (bytes + 255) & ~(size_t)255;
        }
    }
    uint8_t *at(uint8_t *base, int sec, size_t b) const { return base + off[sec] + b * stride[sec]; }
    const uint8_t *at(const uint8_t *base, int sec, size_t b) const { return base + off[sec] + b * stride[sec]; }
    size_t out_off() const { return total; }                         // device side only: verdict words behind the inputs
    static size_t out_bytes(size_t B) { return (2 * B + 4) * 4; }
};
// per-proof outcome of the host side
struct HostBits { uint8_t proof_ok = 0, parsed = 0, ledger = 0, consensus = 0, shape = 0, deferred = 0; uint32_t states_at = 0; };   // proof_ok: the wrap-proof half parsed; parsed: all of it

void put_chal(uint8_t *o, const mw::Chal128 &c) { memcpy(o, &c.lo, 8); memcpy(o + 8, &c.hi, 8); }
void chal_bytes(const mw::Chal128 &c, uint8_t *o) { put_chal(o, c); }
template <int F> fe_t host_mont(const uint8_t *b, const FieldK &k) { fe_t a; memcpy(a.v, b, 32); return fe_to_mont<F>(a, k.r2); }
void put_pt(uint8_t *o, const mw::Pt &p) { memcpy(o, p.x.b, 32); memcpy(o + 32, p.y.b, 32); }

// kimchi gates whose constraints read lookup tables (or the joint combiner).  A step linearization installed WITHOUT feature-dependent
// tokens (no SkipIf / SkipIfNot, no joint combiner, no optional column) cannot evaluate such a proof faithfully, so a statement that switches
// one on is then REJECTED at the kimchi step instead of being evaluated as if the feature were off; a feature-aware program
// (mb_step_index_feature_aware) is evaluated with the statement's flags, joint combiner and optional evaluations (kimchi_dev.cuh).
// feature_flags: range_check0, range_check1, foreign_field_add, foreign_field_mul, xor, rot, lookup, runtime_tables.
bool uses_lookups(const mw::WrapProof &w) {
    return w.has_joint_combiner || w.feature_flags[0] || w.feature_flags[1] || w.feature_flags[3] || w.feature_flags[4] || w.feature_flags[5] || w.feature_flags[6] || w.feature_flags[7];
}

mw::StateProofContainer &tl_box() { static thread_local std::unique_ptr<mw::StateProofContainer> b; if (!b) b.reset(new mw::StateProofContainer()); return *b; }

// host part of one proof: FORMAT, LEDGER, CONSENSUS + its entry of the staging, in two halves -- a MinaStateProof is the wrap proof followed by
// the 17 protocol states (core/src/proof/state_proof.rs:28-41), and the pipeline sends the first half to the GPU while the second is parsed.
// `sh.n_old / n_ev == 0xffffffff`: take them from the proof (single-proof diagnostic form).
// First half: the public inputs and the wrap proof -> every section but the protocol-state records, their field counts and `precheck`.
// hb.proof_ok = 0 (entry untouched) when the bytes are malformed.
void parse_proof_half(const Shape &sh, const Layout &lay, uint8_t *base, size_t b, const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len, HostBits &hb, const mina_ctx *c) {
    hb = HostBits{};
    if (!proof || !pub) return;
    mina_state_pub_inputs pi;
    if (mina_parse_state_pub_inputs(pub, pub_len, &pi) != MINA_OK) return;
    mw::StateProofContainer &box = tl_box();
    mw::Bincode cur(proof, proof_len);
    if (!mw::read_wrap_proof(cur, box.tip_proof) || !cur.ok || cur.pos > 0xffffffffu) return;
    const mw::WrapProof &w = box.tip_proof;
    if (w.lr.empty() || w.lr.size() > 20 || w.step_challenge_polynomial_commitments.size() != w.step_old_bulletproof_challenges.size()) return;
    hb.states_at = (uint32_t)cur.pos;
    hb.proof_ok = 1;
    uint8_t *exp = lay.at(base, S_EXP, b);
    memcpy(exp, pi.candidate_chain_state_hashes, 512); memcpy(exp + 512, pi.bridge_tip_state_hash, 32);
    uint8_t *apre = lay.at(base, S_APRE, b);
    for (int i = 0; i < 16; ++i) put_chal(apre + 16 * i, w.bulletproof_challenges[i]);
    put_pt(lay.at(base, S_ASG, b), w.challenge_polynomial_commitment);
    hb.shape = 1;
    if (sh.kimchi) {
        // the wrap proof must have the shape of the installed index: k rounds, n_prev step-side accumulators, no lookup features; a well-formed
        // proof of another evaluation / recursion shape is verified in a job of its own (`deferred`), a malformed one fails here
        const size_t n_old = w.step_old_bulletproof_challenges.size(), n_ev = w.prev_evals.size();
        if (w.lr.size() != sh.k || w.step_challenge_polynomial_commitments.size() != sh.n_prev || (uses_lookups(w) && !sh.feature_aware) || n_old > 4 || n_ev < 43 || n_ev > 62 ||
            w.prev_public_input.zeta.empty() || w.prev_public_input.zeta_omega.empty() || (sh.network >= 0 && (int)pi.is_state_proof_from_devnet != sh.network)) hb.shape = 0;
        else if (sh.statements && (n_old != sh.n_old || n_ev != sh.n_ev)) { hb.shape = 0; hb.deferred = 1; }
    }
    if (sh.kimchi && hb.shape) {
        uint8_t *lr = lay.at(base, S_LR, b);
        for (size_t i = 0; i < w.lr.size(); ++i) { put_pt(lr + 128 * i, w.lr[i].first); put_pt(lr + 128 * i + 64, w.lr[i].second); }
        put_pt(lay.at(base, S_DELTA, b), w.delta); put_pt(lay.at(base, S_SG, b), w.sg);
        memcpy(lay.at(base, S_Z1, b), w.z1.b, 32); memcpy(lay.at(base, S_Z2, b), w.z2.b, 32);
        uint8_t *pcm = lay.at(base, S_PCM, b);
        for (uint32_t a = 0; a < sh.n_prev; ++a) put_pt(pcm + 64 * a, w.step_challenge_polynomial_commitments[a]);
        uint8_t *wc = lay.at(base, S_WC, b); for (int i = 0; i < 15; ++i) put_pt(wc + 64 * i, w.w_comm[i]);
        put_pt(lay.at(base, S_ZC, b), w.z_comm);
        uint8_t *tc = lay.at(base, S_TC, b); for (int i = 0; i < 7; ++i) put_pt(tc + 64 * i, w.t_comm[i]);
        uint8_t *ev = lay.at(base, S_EV, b);       // kimchi column order: z, 6 selectors, 15 w, 15 coefficients, 6 sigma; (zeta, zeta * omega) each
        auto pair = [&](const mw::B32 (&e)[2]) { memcpy(ev, e[0].b, 32); memcpy(ev + 32, e[1].b, 32); ev += 64; };
        pair(w.z_eval); for (int i = 0; i < 6; ++i) pair(w.selector_eval[i]); for (int i = 0; i < 15; ++i) pair(w.w_eval[i]);
        for (int i = 0; i < 15; ++i) pair(w.coefficients_eval[i]); for (int i = 0; i < 6; ++i) pair(w.s_eval[i]);
        memcpy(lay.at(base, S_FT1, b), w.ft_eval1.b, 32);
        if (lay.stride[S_PCH]) {                   // recursion challenges of the wrap proof: messages_for_next_wrap_proof.old_bulletproof_challenges (expanded on the GPU)
            uint8_t *pch = lay.at(base, S_PCH, b);
            for (uint32_t a = 0; a < sh.n_prev; ++a) for (uint32_t j = 0; j < sh.k; ++j) put_chal(pch + 16 * (a * sh.k + j), w.old_bulletproof_challenges[a < 2 ? a : 1][j < 15 ? j : 14]);
        }
        if (sh.statements) {
            // the wrap circuit's public input = the Pickles statement: derived on the GPU inside the job (api_pickles.hip)
            uint8_t *pl = lay.at(base, S_PLONK, b); put_chal(pl, w.alpha); put_chal(pl + 16, w.beta); put_chal(pl + 32, w.gamma); put_chal(pl + 48, w.zeta);
            uint8_t *bp = lay.at(base, S_BP, b); for (int j = 0; j < 16; ++j) put_chal(bp + 16 * j, w.bulletproof_challenges[j]);
            uint8_t *old = lay.at(base, S_OLD, b); for (size_t a = 0; a < sh.n_old; ++a) for (int j = 0; j < 16; ++j) put_chal(old + 256 * a + 16 * j, w.step_old_bulletproof_challenges[a][j]);
            uint8_t *cm = lay.at(base, S_CM, b); for (size_t a = 0; a < sh.n_old; ++a) put_pt(cm + 64 * a, w.step_challenge_polynomial_commitments[a]);
            uint8_t *wo = lay.at(base, S_WOLD, b); for (int a = 0; a < 2; ++a) for (int j = 0; j < 15; ++j) put_chal(wo + 16 * (15 * a + j), w.old_bulletproof_challenges[a][j]);
            put_pt(lay.at(base, S_WSG, b), w.challenge_polynomial_commitment);
            memcpy(lay.at(base, S_DG, b), w.sponge_digest_before_evaluations, 32);
            // evaluations: one chunk each in every Mina step proof; chunked ones are combined here with zeta^(2^16) (host field arithmetic)
            static const size_t order[43] = {30, 37, 38, 39, 40, 41, 42, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 31, 32, 33, 34, 35, 36};
            bool chunked = w.prev_public_input.zeta.size() != 1 || w.prev_public_input.zeta_omega.size() != 1;
            for (const mw::EvalPair &e : w.prev_evals) chunked = chunked || e.zeta.size() != 1 || e.zeta_omega.size() != 1;
            const FieldK &kpf = c->fk[FIELD_FP];
            fe_t zn = fe_zero(), zwn = fe_zero();
            if (chunked) {
                const fe_t zeta = challenge_to_field<FIELD_FP>(w.zeta.lo, w.zeta.hi, kpf);
                fe_t om = kpf.root; for (uint32_t j = 0; j + w.domain_log2 < 32; ++j) om = fe_sqr<FIELD_FP>(om);
                zn = zeta; zwn = fe_mul<FIELD_FP>(zeta, om);
                for (int j = 0; j < 16; ++j) { zn = fe_sqr<FIELD_FP>(zn); zwn = fe_sqr<FIELD_FP>(zwn); }
            }
            uint8_t *sev = lay.at(base, S_SEV, b);
            auto comb = [&](const mw::SmallVec<mw::B32, 16> &chunks, const fe_t &ptn) {
                if (chunks.size() == 1) { memcpy(sev, chunks[0].b, 32); sev += 32; return; }
                fe_t acc = fe_zero();
                for (size_t j = chunks.size(); j-- > 0;) { if (!mw::fp_canonical(chunks[j].b)) hb.shape = 0; acc = fe_add<FIELD_FP>(fe_mul<FIELD_FP>(acc, ptn), host_mont<FIELD_FP>(chunks[j].b, kpf)); }
                const fe_t pl_ = fe_from_mont<FIELD_FP>(acc); memcpy(sev, pl_.v, 32); sev += 32;
            };
            for (size_t j = 0; j < sh.n_ev; ++j) { const mw::EvalPair &e = w.prev_evals[j < 43 ? order[j] : j]; comb(e.zeta, zn); comb(e.zeta_omega, zwn); }
            uint8_t *ppi = lay.at(base, S_PI, b); memcpy(ppi, w.prev_public_input.zeta[0].b, 32); memcpy(ppi + 32, w.prev_public_input.zeta_omega[0].b, 32);
            memcpy(lay.at(base, S_SFT, b), w.prev_ft_eval1.b, 32);
            memcpy(lay.at(base, S_APP, b), pi.candidate_chain_state_hashes[15], 32);          // the application state = hash of the candidate tip
            uint8_t *misc = lay.at(base, S_MISC, b); memset(misc, 0, 32);
            misc[0] = w.domain_log2; misc[1] = w.proofs_verified; for (int j = 0; j < 8; ++j) misc[2 + j] = w.feature_flags[j] ? 1 : 0;
            misc[10] = w.has_joint_combiner ? 1 : 0; if (w.has_joint_combiner) put_chal(misc + 16, w.joint_combiner);
            { uint32_t present = 0; for (size_t j = 0; j < w.prev_evals_present.size() && j < 19; ++j) if (w.prev_evals_present[j]) present |= 1u << j;
              misc[11] = (uint8_t)present; misc[12] = (uint8_t)(present >> 8); misc[13] = (uint8_t)(present >> 16); }
        }
    }
}

// Second half: the 17 protocol states behind the wrap proof -> records + field counts, the LEDGER and CONSENSUS checks, `precheck`.
// hb.parsed = 1 when the whole container was well-formed; otherwise the records of the entry are left undefined (the caller zeroes them).
void parse_states_half(const Layout &lay, uint8_t *base, size_t b, const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len, HostBits &hb) {
    *lay.at(base, S_PRE, b) = 0;
    if (!hb.proof_ok) return;
    mina_state_pub_inputs pi;
    if (mina_parse_state_pub_inputs(pub, pub_len, &pi) != MINA_OK) return;
    mw::StateProofContainer &box = tl_box();
    mw::Bincode cur(proof, proof_len);
    cur.pos = hb.states_at;
    for (int i = 0; i < MINA_STATES_PER_PROOF; ++i) if (!mw::read_protocol_state(cur, box.states[i])) return;
    if (!cur.ok || cur.pos != proof_len) return;
    mina_protocol_state_info info[MINA_STATES_PER_PROOF];
    uint8_t *recs = lay.at(base, S_REC, b); uint32_t *nf = (uint32_t *)lay.at(base, S_NF, b);
    for (int i = 0; i < MINA_STATES_PER_PROOF; ++i)
        if (mb_pack_protocol_state(box.states[i], recs + (size_t)i * MINA_PSTATE_SLOTS * 32, &nf[i], &info[i]) != MINA_OK) return;
    hb.parsed = 1;
    bool ledger = true;
    for (int i = 0; i < 16; ++i) ledger = ledger && memcmp(pi.candidate_chain_ledger_hashes[i], info[i].snarked_ledger_hash, 32) == 0;
    hb.ledger = ledger;
    // chain selection between the bridge tip (state 16) and the candidate tip (state 15); tie-breaks use the hashes the public
    // input names (the CHAIN step proves them) and Blake2b-256 of the last VRF output (`hashLastVRF`)
    mina_consensus_state tip = info[16].consensus, cand = info[15].consensus;
    memcpy(tip.state_hash, pi.bridge_tip_state_hash, 32); memcpy(cand.state_hash, pi.candidate_chain_state_hashes[15], 32);
    blake2b_short(box.states[16].last_vrf_output.data(), 32, tip.last_vrf_output_hash, 32);
    blake2b_short(box.states[15].last_vrf_output.data(), 32, cand.last_vrf_output_hash, 32);
    mina_consensus_params cp{info[15].slots_per_sub_window, info[15].sub_windows_per_window};
    int sel = 0;
    if (info[16].sub_windows_per_window == cp.sub_windows_per_window && cp.sub_windows_per_window >= 1 && cp.sub_windows_per_window <= MINA_MAX_SUB_WINDOWS &&
        cp.slots_per_sub_window >= 1 && mina_consensus_select_secure_chain(&cp, &tip, &cand, &sel) == MINA_OK)
        hb.consensus = sel == 1;
    *lay.at(base, S_PRE, b) = (hb.ledger && hb.consensus && hb.shape) ? 1 : 0;
}
void parse_into(const Shape &sh, const Layout &lay, uint8_t *base, size_t b, const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len, HostBits &hb, const mina_ctx *c) {
    parse_proof_half(sh, lay, base, b, proof, proof_len, pub, pub_len, hb, c);
    parse_states_half(lay, base, b, proof, proof_len, pub, pub_len, hb);
}

// the job over entries [0, B) of a staging at `base` (host or device addresses alike)
struct JobStructs { mina_state_jobs j; mina_kimchi_proofs k; mina_pickles_statements s; };
void make_jobs(const Shape &sh, const Layout &lay, uint8_t *base, size_t B, bool with_states, bool with_acc, bool with_kimchi, JobStructs &o) {
    memset(&o.j, 0, sizeof o.j); memset(&o.k, 0, sizeof o.k); memset(&o.s, 0, sizeof o.s);
    mina_state_jobs &j = o.j; j.batch = B;
    if (with_states) { j.with_states = 1; j.state_records = lay.at(base, S_REC, 0); j.state_nfields = lay.at(base, S_NF, 0); j.expected_hashes = lay.at(base, S_EXP, 0); j.precheck = lay.at(base, S_PRE, 0); }
    if (with_acc) { j.with_accumulator = 1; j.acc_k = 16; j.acc_prechallenges = lay.at(base, S_APRE, 0); j.acc_sg = lay.at(base, S_ASG, 0); j.acc_rho = lay.at(base, S_ARHO, 0); }
    if (with_kimchi && sh.kimchi) {
        mina_kimchi_proofs &k = o.k; k.batch = B; k.n_prev = sh.n_prev; k.npub = sh.statements ? 40 : 0;
        k.prev_comms = lay.at(base, S_PCM, 0); k.w_comm = lay.at(base, S_WC, 0); k.z_comm = lay.at(base, S_ZC, 0); k.t_comm = lay.at(base, S_TC, 0); k.evals = lay.at(base, S_EV, 0); k.ft_eval1 = lay.at(base, S_FT1, 0);
        if (sh.statements) {
            mina_pickles_statements &s = o.s; s.n_old = sh.n_old; s.n_evals = sh.n_ev;
            s.plonk = lay.at(base, S_PLONK, 0); s.bulletproof_challenges = lay.at(base, S_BP, 0); s.step_old_challenges = lay.at(base, S_OLD, 0); s.step_comms = lay.at(base, S_CM, 0);
            s.wrap_old_challenges = lay.at(base, S_WOLD, 0); s.wrap_sg = lay.at(base, S_WSG, 0); s.sponge_digest = lay.at(base, S_DG, 0); s.prev_evals = lay.at(base, S_SEV, 0);
            s.prev_public_input = lay.at(base, S_PI, 0); s.prev_ft_eval1 = lay.at(base, S_SFT, 0); s.app_state = lay.at(base, S_APP, 0); s.misc = lay.at(base, S_MISC, 0);
            k.statements = &o.s;
        }
        if (lay.stride[S_PCH]) k.prev_prechallenges = lay.at(base, S_PCH, 0);
        j.with_ipa = 1; j.kimchi = &o.k; j.k = sh.k; j.n_evalpoints = 2; j.n_comms = sh.n_prev + 2 + 43; j.log2_domain = sh.k; j.npub = k.npub;
        j.lr = lay.at(base, S_LR, 0); j.delta = lay.at(base, S_DELTA, 0); j.sg = lay.at(base, S_SG, 0); j.z1 = lay.at(base, S_Z1, 0); j.z2 = lay.at(base, S_Z2, 0);
        j.rand_base = lay.at(base, S_RB, 0); j.sg_rand_base = lay.at(base, S_SB, 0);
    }
}

// The folding randomisers of one job: rand_base, sg_rand_base (powers fold the opening checks, `SRS::verify`) and one rho per accumulator
// check, all from the operating system's CSPRNG, drawn AFTER the job's proofs are fixed and never reused across jobs -- upstream
// `batch_verify` draws its own the same way.  Field elements below 2^254 (< both moduli).  false = no entropy: the job fails closed.
bool draw_randomisers(const Shape &sh, const Layout &lay, uint8_t *base, size_t B) {
    if (!mb_secure_random(lay.at(base, S_ARHO, 0), B * 32)) return false;
    for (size_t b = 0; b < B; ++b) lay.at(base, S_ARHO, b)[31] &= 0x3f;
    if (sh.kimchi) {
        if (!mb_secure_random(lay.at(base, S_RB, 0), 32) || !mb_secure_random(lay.at(base, S_SB, 0), 32)) return false;
        lay.at(base, S_RB, 0)[31] &= 0x3f; lay.at(base, S_SB, 0)[31] &= 0x3f;
    }
    return true;
}

// entries that did not parse / do not have the job's shape borrow a well-formed entry's bytes (their own verdict is already 0 through
// `precheck`), so that they do not fail the job's folded checks for everyone else
void copy_proof_half(const Layout &lay, uint8_t *base, size_t dst, size_t src) {      // what parse_proof_half writes; the records of such an entry are zeroed (clear_states_half)
    for (int i = 0; i < NSEC; ++i) if (lay.stride[i] && i != S_PRE && i != S_REC && i != S_NF) memcpy(lay.at(base, i, dst), lay.at(base, i, src), lay.stride[i]);
}
void clear_states_half(const Layout &lay, uint8_t *base, size_t b) { memset(lay.at(base, S_REC, b), 0, lay.stride[S_REC]); memset(lay.at(base, S_NF, b), 0, lay.stride[S_NF]); *lay.at(base, S_PRE, b) = 0; }

struct Config { bool usable = false, kimchi = false, statements = false, feature_aware = false; uint32_t k = 0; int network = -1; };
// Lock order everywhere: g_mu, then a device's mu.  `network` (and `flags`) are copied under g_mu by the caller BEFORE this takes D.mu -- the
// installers hold g_mu while they take every D.mu, so taking g_mu in here would invert the order (a job and an install in flight: deadlock).
Config read_config(Device &D, uint32_t flags, int network) {
    std::lock_guard<std::mutex> lk(D.mu);
    mina_ctx *c = D.c; Config cf;
    // a deployment on the library's surrogate Poseidon tables agrees with nothing on the Mina network: refuse unless the caller said so
    const bool surrogate = c->pparams_surrogate[0] || c->pparams_surrogate[1];
    cf.usable = !surrogate || (flags & MINA_VERIFY_ALLOW_SURROGATE);
    cf.statements = mb_step_index_installed(c) != 0;
    // without a step index the wrap proof's public input cannot be derived from the statement: the kimchi step is then NOT runnable
    // (the proof would not be bound to the candidate tip) unless the caller explicitly accepts the unbound form
    cf.kimchi = mb_kimchi_available(c) && (cf.statements || (flags & MINA_VERIFY_ALLOW_UNBOUND_STATEMENT));
    cf.k = c->kimchi_log2;
    cf.feature_aware = cf.statements && mb_step_index_feature_aware(c) != 0;
    cf.network = network;
    return cf;
}

struct CallIn { const uint8_t *const *proofs; const size_t *proof_lens; const uint8_t *const *pubs; const size_t *pub_lens; };

struct Chunk {
    size_t lo = 0, n = 0; Slot *slot = nullptr; int slot_ix = -1; std::vector<HostBits> hb; bool issued = false, skipped = false, harvested = false;
    // two pool jobs per chunk: the wrap-proof halves of its entries (A), then the protocol-state halves (B) in `nsub` runs of `sub` entries -- the
    // records of a run go to the GPU, and their hashes are queued, as soon as the run is parsed
    std::shared_ptr<MbPoolJob> jobH, jobA, jobB; size_t head = 0;
    size_t sub = 0, nsub = 0; std::unique_ptr<std::atomic<uint32_t>[]> sub_left; std::mutex mu; std::condition_variable cv;
    size_t hashed = 0;                        // protocol states whose hashes are queued
    bool legs_set = false, queued = false, counted = false, randomised = false; Lane *LI = nullptr, *LA = nullptr, *LS = nullptr; StateJobCarry carry;
};

const bool g_timing = getenv("MINA_VERIFY_TIMING") != nullptr;
double ms_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); }

// what a caller of the pipeline copied under g_mu (lock order: g_mu before any D.mu) + the tuning of the call
struct CallEnv { uint32_t flags = 0; int network = -1; mina_verify_tuning tu; };

// verdict bytes of the proofs idx[0..m) of the call on device D that share ONE evaluation / recursion shape; well-formed proofs of another
// shape are returned in `deferred` (their verdict bytes stay 0 until the caller runs them as a job of their own)
int run_device_shape(Device &D, const CallIn &in, const std::vector<size_t> &idx, uint8_t *verdicts, const CallEnv &env, std::vector<size_t> &deferred) {
    const size_t m = idx.size();
    const uint32_t flags = env.flags; const mina_verify_tuning &tu = env.tu;
    deferred.clear();
    for (size_t i = 0; i < m; ++i) verdicts[idx[i]] = 0;
    if (m == 0) return MINA_OK;
    const auto t_call = std::chrono::steady_clock::now();
    const Config cf = read_config(D, flags, env.network);
    if (!cf.usable) return MINA_OK;
    if (!cf.kimchi && !(flags & MINA_VERIFY_ALLOW_MISSING_KIMCHI)) return MINA_OK;          // the kimchi step cannot run: nothing can pass
    Shape sh; sh.kimchi = cf.kimchi; sh.statements = cf.kimchi && cf.statements; sh.k = cf.k; sh.network = cf.network; sh.feature_aware = cf.feature_aware;
    if (sh.statements) {           // evaluation / recursion shape of the job (Mina's blockchain proofs all share one): the most frequent shape among up to 9
        // proofs spread over the call -- one odd-shaped proof at the head of a merged job must not send every other caller's proof through a second job --
        // and, when none of the sampled ones parses, the first proof that does
        auto shape_of = [&](size_t q, uint32_t &n_old, uint32_t &n_ev) -> bool {
            if (!in.proofs[q]) return false;
            mw::StateProofContainer &box = tl_box();
            mw::Bincode cur(in.proofs[q], in.proof_lens[q]);
            if (!mw::read_wrap_proof(cur, box.tip_proof)) return false;
            const size_t a = box.tip_proof.step_old_bulletproof_challenges.size(), b = box.tip_proof.prev_evals.size();
            if (a > 4 || b < 43 || b > 62) return false;
            n_old = (uint32_t)a; n_ev = (uint32_t)b; return true;
        };
        std::vector<std::pair<uint32_t, uint32_t>> votes;
        const size_t samples = std::min<size_t>(m, 9);
        for (size_t i = 0; i < samples; ++i) { uint32_t a, b; if (shape_of(idx[samples > 1 ? i * (m - 1) / (samples - 1) : 0], a, b)) votes.push_back({a, b}); }
        bool found = false;
        if (!votes.empty()) {
            size_t best = 0;
            for (auto &v : votes) { const size_t cnt = (size_t)std::count(votes.begin(), votes.end(), v); if (cnt > best) { best = cnt; sh.n_old = v.first; sh.n_ev = v.second; } }
            found = true;
        }
        for (size_t i = 0; i < m && !found; ++i) found = shape_of(idx[i], sh.n_old, sh.n_ev);
        if (!found) return MINA_OK;                                                          // nothing parses
    }
    // read per call (tests force tiny chunks / shards to drive the pipeline's slot recycling with a handful of proofs)
    // Measured on one MI355X (tools/boundary_sweep.sh, bench.py --boundary-only): 8192 full-size proofs as ONE chunk 55 - 58 ms, 2 x 4096: 74 ms,
    // 8 x 1024: 87 ms, 16 x 512: 92 ms -- a job is a ~30 ms dependent chain of small kernels whatever its size, and jobs that start together do
    // not fill each other's gaps the way the staggered steps of a long-running pipeline do.  Bigger calls in chunks of 8192 (16 384 proofs: 92 ms
    // against 121 ms in chunks of 4096; 65 536: 305 against 336 ms = 215 k proofs/s): their parsing and upload overlap the previous chunk's job,
    // and one bad proof costs a culprit search over 8192, not over everything.
    const size_t chunk_target = std::max<size_t>(1, tu.chunk);
    const size_t single_max = std::max<size_t>(1, tu.single_max);
    const size_t nchunks = m <= single_max ? 1 : (m + chunk_target - 1) / chunk_target;
    // chunks of ONE call on the GPU at a time (the next ones are parsed meanwhile)
    // A call of more chunks than the window is a pipeline: jobs that enter the GPU together also leave it together (4 jobs started within 15 ms: all done
    // at ~185 ms, then the next 4 -- 65 536 proofs: 345 ms), jobs one period apart keep the chip filled while one of them drains (four caller threads
    // with 8192 proofs each reach 238 k proofs/s that way).  So the first `window` chunks of such a call are issued `pace_ms` apart.
    const double pace_ms = tu.pace_us * 1e-3;
    // (no parsing ahead of the window: a slot is a staging buffer AND four streams, and with more streams in use than the runtime's 16 hardware queues the
    // jobs' legs wait for each other -- 65 536 proofs per call: 342 ms with two chunks parsed ahead, 285 ms with none; the wrap-proof halves of a chunk take 1 ms)
    // slots of the device in use, over all callers: 4 jobs with their legs forked = 16 streams = the runtime's hardware queues (two callers of 32 768 proofs
    // with 16 slots: 125 k proofs/s, six callers of 8192: 205 k -- against 235 k for four)
    const size_t head_min = tu.head_min;      // 0: every streamed chunk; huge: never
    const int nslot = std::min(NSLOT, std::max(1, (int)tu.slots));
    const size_t ahead = tu.ahead;
    const size_t window = std::max<size_t>(1, tu.window);
    // streamed form of a chunk (stream_records below): from `early_min` entries, in runs of `early_sub` (0 = off)
    const size_t early_min = std::max<size_t>(1, tu.early_min);
    const size_t early_sub = tu.early_sub;
    std::vector<Chunk> chunks(nchunks);
    for (size_t q = 0; q < nchunks; ++q) { chunks[q].lo = m * q / nchunks; chunks[q].n = m * (q + 1) / nchunks - chunks[q].lo; chunks[q].hb.resize(chunks[q].n); }
    const size_t cap = (m + nchunks - 1) / nchunks;
    Layout lay; lay.build(sh, cap);
    mina_ctx *c = D.c;
    int rc_all = MINA_OK;

    auto try_acquire = [&]() -> int {
        std::lock_guard<std::mutex> lk(D.slot_mu);
        for (int s = 0; s < nslot; ++s) if (!D.slots[s].busy) { D.slots[s].busy = true; return s; }
        return -1;
    };
    auto release = [&](int s) { { std::lock_guard<std::mutex> lk(D.slot_mu); D.slots[s].busy = false; } D.slot_cv.notify_all(); };

    auto fallback = [&](Chunk &ch) -> int {     // a folded check of the chunk failed: per-proof verdicts through the culprit search, from the same staging
        JobStructs js; make_jobs(sh, lay, (uint8_t *)ch.slot->host.p, ch.n, true, true, true, js);
        std::vector<uint8_t> v(ch.n, 0);
        const auto t = std::chrono::steady_clock::now();
        int rc;
        if (tu.search_ctx) {
            // on the device's SECOND context (Device::sc): its own lanes, workspaces and lock -- nothing of `c` is touched, nothing is drained, D.mu is not taken:
            // the other callers of the device keep queueing while the search runs beside their jobs
            std::lock_guard<std::mutex> lk(D.search_mu);
            (void)hipSetDevice(D.sc->device);
            // The search fans its parts over lanes 1 .. 3 of the view beside lane 0.  They run on the streams of the failed chunk's OWN slot -- its lane and its three leg
            // streams, idle now (the chunk has been harvested) and held by the slot until this returns -- so that a search creates NO stream: a process with more
            // streams than hardware queues stays slower for its whole life (round 3: after_search.py; round 5: BASELINE C5's call 45 -> 52 - 54 ms after a few searches
            // that created theirs, with either search form).
            hipStream_t borrowed[3] = {nullptr, nullptr, nullptr};
            {
                std::lock_guard<std::mutex> dl(D.mu);
                mb_ctx_refresh_view(D.sc, c);                        // what is installed on c NOW (an install since the last search; tables prepared by the jobs)
                Lane *own[3] = {&c->lanes[MB_PIPE_LANES + 3 * ch.slot_ix], &c->lanes[MB_PIPE_LANES + 3 * ch.slot_ix + 1], &c->lanes[MB_PIPE_LANES + 3 * ch.slot_ix + 2]};
                for (int q = 0; q < 3; ++q) {
                    if (!own[q]->stream && hipStreamCreateWithFlags(&own[q]->stream, hipStreamNonBlocking) != hipSuccess) return fail(MINA_ERR_HIP, "hipStreamCreate for the culprit search");
                    borrowed[q] = own[q]->stream;
                }
            }
            for (int q = 0; q < 3; ++q) D.sc->lanes[1 + q].stream = borrowed[q];
            rc = mina_state_job_batch(D.sc, &js.j, v.data());
            for (int q = 0; q < 3; ++q) { if (D.sc->lanes[1 + q].stream) (void)hipStreamSynchronize(D.sc->lanes[1 + q].stream); D.sc->lanes[1 + q].stream = nullptr; }
        } else {
            std::lock_guard<std::mutex> lk(D.mu);
            // (round 4) The search runs synchronous jobs on lane 0 and fans out over lanes 0 .. search_fan - 1 (and the forked-leg helpers of lane 0) -- the lanes of the
            // slots other chunks / callers have in flight.  So the device is drained first: everything queued so far completes (queued work needs no host
            // action), and nothing new can be queued while D.mu is held.  The search then has the GPU to itself: its lane forms are those of ONE call.
            (void)hipSetDevice(c->device);
            if (hipDeviceSynchronize() != hipSuccess) return fail(MINA_ERR_HIP, "hipDeviceSynchronize before the culprit search");
            c->nlanes = 1;
            rc = mina_state_job_batch(c, &js.j, v.data());
        }
        if (g_timing) fprintf(stderr, "mina_verify: chunk of %zu: folded check failed, culprit search %.2f ms (rc %d)\n", ch.n, ms_since(t), rc);
        if (rc) return rc;
        for (size_t b = 0; b < ch.n; ++b) verdicts[idx[ch.lo + b]] = (v[b] && ch.hb[b].parsed && ch.hb[b].shape) ? 1 : 0;
        return MINA_OK;
    };
    auto harvest = [&](Chunk &ch) {
        if (ch.harvested) return;
        ch.harvested = true;
        if (ch.issued) {
            bool ok = hipEventSynchronize(ch.slot->ev) == hipSuccess;
            if (g_timing && ok && ch.slot->tev[2]) { float a = 0, b = 0; (void)hipEventElapsedTime(&a, ch.slot->tev[0], ch.slot->tev[1]); (void)hipEventElapsedTime(&b, ch.slot->tev[1], ch.slot->tev[2]);
                fprintf(stderr, "mina_verify:   chunk at %zu: upload of %.1f MB %.2f ms, job %.2f ms (on its lane)\n", ch.lo, lay.total / 1e6, a, b); }
            const uint32_t *o = (const uint32_t *)ch.slot->out.p;
            if (ok) {
                const bool ipa_ok = !sh.kimchi || o[ch.n] != 0, acc_ok = o[ch.n + 2] != 0;
                if (ipa_ok && acc_ok) { for (size_t b = 0; b < ch.n; ++b) verdicts[idx[ch.lo + b]] = (o[b] && ch.hb[b].parsed && ch.hb[b].shape) ? 1 : 0; }
                else { int rc = fallback(ch); if (rc && !rc_all) rc_all = rc; }
            } else if (!rc_all) rc_all = MINA_ERR_HIP;
        }
        else if (ch.queued) {                                         // the job was not completed, but parts of it are queued: nothing of them may outlive the slot
            std::lock_guard<std::mutex> lk(D.mu);
            (void)hipSetDevice(c->device);
            (void)hipDeviceSynchronize();
        }
        if (ch.slot_ix >= 0) release(ch.slot_ix);
        if (ch.counted) D.inflight.fetch_sub(1);
    };

    // D.mu, taken with the tables of this call's (domain, npub) prepared.  A re-prepare (another index since the last call) frees and rewrites the Lagrange tables --
    // buffers a culprit search on the view context D.sc may be reading through its aliases, holding search_mu only (round 5 re-prepared under
    // D.mu alone).  So the re-prepare takes search_mu as well, FIRST (lock order g_mu, search_mu, mu: the installers', the fallback's); the common path -- already
    // prepared -- never touches search_mu and never waits for a search.
    const uint32_t prep_key = ((sh.k ? sh.k : 15u) << 16) | (sh.statements ? 40u : 0u);          // the Lagrange table belongs to (domain, npub): another index -> prepare again
    auto lock_prepared = [&](std::unique_lock<std::mutex> &lk) -> int {
        for (;;) {
            lk = std::unique_lock<std::mutex>(D.mu);
            if (D.prepared_npub == prep_key) return MINA_OK;
            lk.unlock();
            std::lock_guard<std::mutex> sl(D.search_mu);
            std::lock_guard<std::mutex> dl(D.mu);
            if (D.prepared_npub == prep_key) continue;
            HIPC(hipSetDevice(c->device));
            int rc = mina_state_jobs_prepare(c, sh.k ? sh.k : 15, sh.statements ? 40 : 0);
            if (rc) return rc;
            D.prepared_npub = prep_key;
        }
    };
    // device side of a chunk's slot, before anything is queued on it (idempotent).  Caller holds D.mu through lock_prepared.
    auto setup_slot = [&](Chunk &ch) -> int {
        HIPC(hipSetDevice(c->device));
        int rc;
        if (D.prepared_npub != prep_key) return fail(MINA_ERR_STATE, "slot set up without the prepared tables");
        Slot &S = *ch.slot;
        if ((rc = S.dev.ensure(lay.total + Layout::out_bytes(lay.cap)))) return rc;
        if (!S.ev) HIPC(hipEventCreateWithFlags(&S.ev, hipEventDisableTiming));
        if (!S.ev_up) HIPC(hipEventCreateWithFlags(&S.ev_up, hipEventDisableTiming));
        if (!S.up) HIPC(hipStreamCreateWithFlags(&S.up, hipStreamNonBlocking));
        Lane &L = c->lanes[ch.slot_ix];
        if (!L.stream) HIPC(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        if (g_timing) for (auto &e : S.tev) if (!e) HIPC(hipEventCreate(&e));
        return MINA_OK;
    };
    // the lanes of a chunk's job, decided once per chunk.  Caller holds D.mu.
    auto setup_legs = [&](Chunk &ch) -> int {
        if (ch.legs_set) return MINA_OK;
        ch.legs_set = true;
        // few chunks in flight: the three legs of a job (state hashes / wrap proof / accumulator) go to three streams
        // (up to 4 chunks in flight -- 32 768 proofs per call 186 -> 152 ms, three callers of 8192: 138 -> 158 k proofs/s; with 8 the streams outnumber the hardware queues: 322 -> 360 ms)
        if (D.inflight.load() > tu.split_max) return MINA_OK;
        Lane *LI = &c->lanes[MB_PIPE_LANES + 3 * ch.slot_ix], *LA = &c->lanes[MB_PIPE_LANES + 3 * ch.slot_ix + 1], *LS = &c->lanes[MB_PIPE_LANES + 3 * ch.slot_ix + 2];
        // Alone on the GPU a job is a latency-bound chain of small kernels (~390 waves each, one behind the other) beside 20 ms of chip-filling
        // hashes; where their waves share a SIMD both run at half speed, and the call waits for the chain (rocprofv3 timeline: the statement
        // digests 10.7 ms beside the hashes against 4.1 ms alone).  So the chain's stream and the hashes' stream get DISJOINT CU masks: the chain
        // `chain_cus` CUs -- 128 = a SIMD per wave; fewer and a kernel lasts as long as its doubled-up SIMDs: 96 CUs cost +16 ms -- the hashes the
        // rest (bit i of a mask = CU i / 8 of XCD i % 8, tools/probes/cumask_probe.hip).  8192 proofs per call: 62.6 -> 57.7 ms.
        const uint32_t chain_cus = tu.chain_cus;
        int ncu = 0; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
        if (chain_cus > 0 && chain_cus < (uint32_t)ncu && ncu <= 256) {
            const uint32_t period = std::max(1u, tu.cu_period);
            auto masked = [&](Lane &ln, bool chain) -> int {
                if (ln.stream) return MINA_OK;
                uint32_t mk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (uint32_t b2 = 0; b2 < (uint32_t)ncu; ++b2) { const bool in_chain = (b2 % period) < chain_cus * period / 256; if (in_chain == chain) mk[b2 >> 5] |= 1u << (b2 & 31); }
                if (hipExtStreamCreateWithCUMask(&ln.stream, 8, mk) != hipSuccess) {       // a runtime without CU masks: plain streams, the legs share the CUs
                    (void)hipGetLastError(); ln.stream = nullptr;
                    HIPC(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
                }
                return MINA_OK;
            };
            int rc;
            if ((rc = masked(*LI, true)) || (rc = masked(*LS, false))) return rc;
            // the accumulator leg (fold GEMM + one MSM: chip-filling kernels, done within ~8 ms) shares the chain's CUs: the hashes are the later leg of a lone
            // job, and unmasked this leg slowed their first pieces (8192 per call: 45.4 -> 44.5 ms; mina_verify_tuning.acc_mask = 0 chain | 1 hash | 2 none)
            if (tu.acc_mask == 0) { if ((rc = masked(*LA, true))) return rc; } else if (tu.acc_mask == 1) { if ((rc = masked(*LA, false))) return rc; }
        } else LS = nullptr;
        ch.LI = LI; ch.LA = LA; ch.LS = LS;
        return MINA_OK;
    };
    const bool own_up = tu.up_stream != 0;
    auto lane_forms = [&]() {       // the lane forms of the sponge kernels follow the work in flight on the device (ctx.h use_coop*)
        c->nlanes = (int)std::max(1u, std::min<unsigned>(D.inflight.load(), NSLOT));
        c->hash_piece_waves = tu.hash_piece_waves;
    };

    // A chunk goes to the GPU in three steps, each as soon as its input is parsed (8192 full-size proofs per call: the job's wrap-proof chain
    // takes ~42 ms on its CUs and the state hashes ~43 ms on theirs, so the call is as long as the later of the two STARTS):
    //   issue_legs      after the wrap-proof halves (pool job A, ~1/3 of the parsing): patch, draw the randomisers, upload everything but the records,
    //                   queue the accumulator and wrap-proof legs
    //   stream_records  run by run of pool job B: upload the run's protocol-state records (34 KB of an entry's 47 KB) and queue their hashes
    //   finish          upload `precheck`, queue the rest of the state leg, the joins and the verdict kernel, the download
    auto issue_legs = [&](Chunk &ch) -> int {
        uint8_t *hbase = (uint8_t *)ch.slot->host.p;
        // host side of the chunk: collect the other shapes, patch what cannot go to the GPU as it is
        size_t donor = SIZE_MAX;
        for (size_t b = 0; b < ch.n; ++b) { if (ch.hb[b].proof_ok && ch.hb[b].shape && donor == SIZE_MAX) donor = b; if (ch.hb[b].deferred) deferred.push_back(idx[ch.lo + b]); }
        if (donor == SIZE_MAX) { ch.skipped = true; return MINA_OK; }                         // nothing of this chunk can pass
        for (size_t b = 0; b < ch.n; ++b) if (!(ch.hb[b].proof_ok && ch.hb[b].shape)) copy_proof_half(lay, hbase, b, donor);
        if (!ch.randomised) return fail(MINA_ERR_STATE, "no entropy for the folding randomisers");
        std::unique_lock<std::mutex> lk;
        int rc;
        if ((rc = lock_prepared(lk)) || (rc = setup_slot(ch)) || (rc = setup_legs(ch))) return rc;
        Slot &S = *ch.slot;
        Lane &L = c->lanes[ch.slot_ix];
        lane_forms();
        c->L = &L;
        uint8_t *dbase = S.dev.as<uint8_t>();
        hipStream_t up = own_up ? S.up : L.stream;
        if (g_timing && !ch.queued) HIPC(hipEventRecord(S.tev[0], up));
        ch.queued = true;
        // everything behind the records and their field counts, EXCEPT `precheck`: the pool is still writing those bytes (parse_states_half) -- finish() sends them.
        // (Until round 5 the one copy ran over them too and finish() re-sent them: harmless, since only the FINISH phase reads `precheck`, but a read of memory
        // another thread is writing -- the one report of the ThreadSanitizer tier, tests/fuzz/tsan_boundary.cpp.)
        HIPC(hipMemcpyAsync(dbase + lay.off[S_EXP], hbase + lay.off[S_EXP], lay.off[S_PRE] - lay.off[S_EXP], hipMemcpyHostToDevice, up));
        HIPC(hipMemcpyAsync(dbase + lay.off[S_APRE], hbase + lay.off[S_APRE], lay.total - lay.off[S_APRE], hipMemcpyHostToDevice, up));
        if (own_up) { HIPC(hipEventRecord(S.ev_up, up)); HIPC(hipStreamWaitEvent(L.stream, S.ev_up, 0)); }
        JobStructs js; make_jobs(sh, lay, dbase, ch.n, true, true, true, js);
        uint32_t *dv = (uint32_t *)(dbase + lay.out_off()), *df = dv + ch.n, *ds = df + 4;
        rc = mb_state_jobs_on_lane(c, &js.j, dv, df, ch.LI, ch.LA, ds, ch.LS, MB_JOB_LEGS, &ch.carry);
        c->hash_piece_waves = 0;
        c->use_lane0();
        return rc;
    };
    auto stream_records = [&](Chunk &ch, size_t r_lo, size_t r_hi) -> int {
        uint8_t *hbase = (uint8_t *)ch.slot->host.p;
        Slot &S = *ch.slot;
        for (size_t r = r_lo; r < r_hi && r < ch.nsub; ++r) {
            { std::unique_lock<std::mutex> lk(ch.mu); ch.cv.wait(lk, [&] { return ch.sub_left[r].load() == 0; }); }
            const size_t lo = r * ch.sub, hi = std::min(ch.n, lo + ch.sub);
            for (size_t b = lo; b < hi; ++b) if (!ch.hb[b].parsed) clear_states_half(lay, hbase, b);      // malformed (or of another shape / to be patched): defined records, verdict 0 through `precheck`
            std::unique_lock<std::mutex> lk;
            { int src; if ((src = lock_prepared(lk)) || (src = setup_slot(ch)) || (src = setup_legs(ch))) return src; }
            Lane &L = c->lanes[ch.slot_ix];
            if (g_timing && !ch.queued) HIPC(hipEventRecord(S.tev[0], own_up ? S.up : L.stream));
            ch.queued = true;
            uint8_t *dbase = S.dev.as<uint8_t>();
            if (S.rec_ev.size() < ch.nsub) S.rec_ev.resize(ch.nsub, nullptr);
            if (!S.rec_ev[r]) HIPC(hipEventCreateWithFlags(&S.rec_ev[r], hipEventDisableTiming));
            hipStream_t up = own_up ? S.up : L.stream;
            HIPC(hipMemcpyAsync(lay.at(dbase, S_REC, lo), lay.at(hbase, S_REC, lo), (hi - lo) * lay.stride[S_REC], hipMemcpyHostToDevice, up));
            HIPC(hipMemcpyAsync(lay.at(dbase, S_NF, lo), lay.at(hbase, S_NF, lo), (hi - lo) * lay.stride[S_NF], hipMemcpyHostToDevice, up));
            HIPC(hipEventRecord(S.rec_ev[r], up));
            lane_forms();
            // hashes in WHOLE pieces (a piece = `hash_piece_waves` waves of 21 states = two waves on every SIMD of the state leg's 128 CUs): a run of
            // 1024 entries is 829 waves -- launched run by run, a fifth of the leg's SIMDs would hold one wave where the others hold two, for as long
            // The odd piece goes FIRST: it is complete after fewer runs, so the leg starts earlier, and it ends with a full piece instead of a half-empty one.
            const size_t piece = (size_t)c->hash_piece_waves * 21, ready = hi * MINA_STATES_PER_PROOF, total = ch.n * MINA_STATES_PER_PROOF;
            const size_t first = piece ? (total % piece ? total % piece : piece) : 0;
            const size_t upto = (r + 1 == ch.nsub || !piece) ? ready : (ready < first ? 0 : first + (ready - first) / piece * piece);
            int rc = MINA_OK;
            if (upto > ch.hashed)
                rc = mb_state_hashes_early(c, ch.LS ? ch.LS : &L, ch.n * MINA_STATES_PER_PROOF, ch.hashed, upto - ch.hashed,
                                           (const uint32_t *)lay.at(dbase, S_REC, 0), (const uint32_t *)lay.at(dbase, S_NF, 0), S.rec_ev[r]);
            c->hash_piece_waves = 0;
            c->use_lane0();
            if (rc) return rc;
            ch.hashed = std::max(ch.hashed, upto);
        }
        return MINA_OK;
    };
    auto finish = [&](Chunk &ch) -> int {
        uint8_t *hbase = (uint8_t *)ch.slot->host.p;
        std::lock_guard<std::mutex> lk(D.mu);
        HIPC(hipSetDevice(c->device));
        Slot &S = *ch.slot;
        Lane &L = c->lanes[ch.slot_ix];
        lane_forms();
        c->L = &L;
        uint8_t *dbase = S.dev.as<uint8_t>();
        hipStream_t up = own_up ? S.up : L.stream;
        HIPC(hipMemcpyAsync(lay.at(dbase, S_PRE, 0), lay.at(hbase, S_PRE, 0), ch.n * lay.stride[S_PRE], hipMemcpyHostToDevice, up));
        if (g_timing) HIPC(hipEventRecord(S.tev[1], up));
        if (own_up) { HIPC(hipEventRecord(S.ev_up, up)); HIPC(hipStreamWaitEvent(L.stream, S.ev_up, 0)); }
        JobStructs js; make_jobs(sh, lay, dbase, ch.n, true, true, true, js);
        uint32_t *dv = (uint32_t *)(dbase + lay.out_off()), *df = dv + ch.n, *ds = df + 4;
        c->state_hashes_early = ch.hashed;
        int rc = mb_state_jobs_on_lane(c, &js.j, dv, df, ch.LI, ch.LA, ds, ch.LS, MB_JOB_FINISH, &ch.carry);
        c->state_hashes_early = 0;
        c->hash_piece_waves = 0;
        c->use_lane0();
        if (rc) return rc;
        if (g_timing) HIPC(hipEventRecord(S.tev[2], L.stream));
        if (own_up) { const uint32_t words = (uint32_t)(Layout::out_bytes(ch.n) / 4); launch_words_out(L.stream, words, dv, (uint32_t *)S.out.p); HIPC(hipGetLastError()); }
        else HIPC(hipMemcpyAsync(S.out.p, dv, Layout::out_bytes(ch.n), hipMemcpyDeviceToHost, L.stream));
        HIPC(hipEventRecord(S.ev, L.stream));
        ch.issued = true;
        return MINA_OK;
    };

    size_t next_submit = 0, next_issue = 0, oldest = 0;
    auto t_issued = t_call;
    while (next_issue < nchunks && !rc_all) {
        while (next_submit < nchunks && next_submit < oldest + window + ahead) {        // parsing runs `ahead` chunks ahead of the window
            const int s = try_acquire();
            if (s < 0) break;
            Chunk &ch = chunks[next_submit];
            ch.slot = &D.slots[s]; ch.slot_ix = s;
            if (ch.slot->host.ensure(lay.total) || ch.slot->out.ensure(Layout::out_bytes(lay.cap))) { release(s); ch.slot = nullptr; ch.slot_ix = -1; rc_all = MINA_ERR_HIP; break; }
            uint8_t *hbase = (uint8_t *)ch.slot->host.p;
            Chunk *chp = &ch;
            ch.sub = (early_sub && ch.n >= early_min) ? early_sub : ch.n; ch.nsub = (ch.n + ch.sub - 1) / ch.sub;
            ch.sub_left.reset(new std::atomic<uint32_t>[ch.nsub]);
            for (size_t r = 0; r < ch.nsub; ++r) ch.sub_left[r].store((uint32_t)(std::min(ch.n, (r + 1) * ch.sub) - r * ch.sub));
            // the pool's order per chunk: the FIRST run whole (both halves: the state leg is the later one of a lone job, its first piece of hashes needs
            // ~600 entries), then the wrap-proof halves of the rest (A), then their protocol states (B)
            // (from ~6000 entries: below, the wrap-proof chain is the later leg and must not wait -- 4096 per call: 34.6 ms without, 35.4 with; 8192: 45.8 / 44.4)
            const size_t head = (ch.nsub > 1 && ch.n >= head_min) ? ch.sub : 0;
            ch.head = head;
            auto run_done = [chp](size_t b) { if (chp->sub_left[b / chp->sub].fetch_sub(1) == 1) { std::lock_guard<std::mutex> lk(chp->mu); chp->cv.notify_all(); } };
            ch.jobH = mb_pool_submit(head, [&, chp, hbase, run_done](size_t b) {
                const size_t q = idx[chp->lo + b];
                parse_proof_half(sh, lay, hbase, b, in.proofs[q], in.proof_lens[q], in.pubs[q], in.pub_lens[q], chp->hb[b], c);
                parse_states_half(lay, hbase, b, in.proofs[q], in.proof_lens[q], in.pubs[q], in.pub_lens[q], chp->hb[b]);
                run_done(b);
            });
            ch.jobA = mb_pool_submit(ch.n - head, [&, chp, hbase, head](size_t i) {
                const size_t b = head + i, q = idx[chp->lo + b];
                parse_proof_half(sh, lay, hbase, b, in.proofs[q], in.proof_lens[q], in.pubs[q], in.pub_lens[q], chp->hb[b], c);
            });
            ch.jobB = mb_pool_submit(ch.n - head, [&, chp, hbase, head, run_done](size_t i) {
                mb_pool_wait(chp->jobA);                                  // the pool hands jobs out in order, but the last items of A may still be running
                const size_t b = head + i, q = idx[chp->lo + b];
                parse_states_half(lay, hbase, b, in.proofs[q], in.proof_lens[q], in.pubs[q], in.pub_lens[q], chp->hb[b]);
                run_done(b);
            });
            ++next_submit;
        }
        if (rc_all) break;
        if (next_issue == next_submit) {                 // no slot for the next chunk: take back the oldest of this call, or wait for another caller's
            if (oldest < next_issue) { harvest(chunks[oldest++]); continue; }
            std::unique_lock<std::mutex> lk(D.slot_mu);
            D.slot_cv.wait_for(lk, std::chrono::milliseconds(2));
            continue;
        }
        if (next_issue - oldest >= window) { harvest(chunks[oldest++]); continue; }
        if (pace_ms > 0 && nchunks > window && next_issue > 0 && next_issue < window) {      // fill the window one job period apart (see `pace_ms`)
            const double wait = pace_ms - ms_since(t_issued);
            if (wait > 0) std::this_thread::sleep_for(std::chrono::duration<double, std::milli>(wait));
        }
        t_issued = std::chrono::steady_clock::now();
        Chunk &ch = chunks[next_issue];
        ch.counted = true; D.inflight.fetch_add(1);
        // the folding randomisers of the chunk's job (262 KB from the OS for 8192 proofs: ~0.5 ms), while the pool parses: the proofs are the caller's
        // bytes, fixed since the call was made, and the values never leave the process
        ch.randomised = draw_randomisers(sh, lay, (uint8_t *)ch.slot->host.p, ch.n);
        int rc = ch.head ? stream_records(ch, 0, 1) : MINA_OK;          // the first run's records and hashes, ahead of everything else
        mb_pool_wait(ch.jobH); mb_pool_wait(ch.jobA);
        const double t_a = g_timing ? ms_since(t_call) : 0;
        if (!rc) rc = issue_legs(ch);
        const double t_legs = g_timing ? ms_since(t_call) : 0;
        if (!rc && !ch.skipped) rc = stream_records(ch, ch.head ? 1 : 0, ch.nsub);
        mb_pool_wait(ch.jobB);
        const double t_parsed = g_timing ? ms_since(t_call) : 0;
        if (!rc && !ch.skipped) rc = finish(ch);
        if (g_timing) fprintf(stderr, "mina_verify:   chunk %zu (%zu proofs): wrap proofs parsed at %.2f ms, legs queued at %.2f, states parsed at %.2f, all queued at %.2f ms\n", next_issue, ch.n, t_a, t_legs, t_parsed, ms_since(t_call));
        if (rc && !rc_all) rc_all = rc;
        ++next_issue;
    }
    for (size_t q = 0; q < next_submit; ++q) { mb_pool_wait(chunks[q].jobH); mb_pool_wait(chunks[q].jobA); mb_pool_wait(chunks[q].jobB); }     // nothing may still write into a slot (error paths)
    for (size_t q = 0; q < nchunks; ++q) {
        if (q < next_submit) { harvest(chunks[q]); if (g_timing) fprintf(stderr, "mina_verify:   chunk %zu harvested at %.2f ms\n", q, ms_since(t_call)); }
    }
    if (g_timing) fprintf(stderr, "mina_verify: device %d: %zu proofs in %zu chunk(s), %.2f ms\n", D.ordinal, m, nchunks, ms_since(t_call));
    if (rc_all) { for (size_t i = 0; i < m; ++i) verdicts[idx[i]] = 0; deferred.clear(); return rc_all; }
    return MINA_OK;
}

// verdict bytes of the proofs idx[0..m) of the call on device D: one pass per evaluation / recursion shape among them (Mina's blockchain proofs share
// one; every pass takes at least the proofs of its own shape, so there are at most as many passes as distinct shapes: 5 x 20 by the parser's bounds)
int run_device(Device &D, const CallIn &in, const std::vector<size_t> &idx, uint8_t *verdicts, const CallEnv &env) {
    std::vector<size_t> cur, next;
    int rc = run_device_shape(D, in, idx, verdicts, env, next);
    for (int pass = 1; !rc && !next.empty(); ++pass) {
        cur.swap(next);
        if (pass > 128) { for (size_t q : cur) verdicts[q] = 0; rc = fail(MINA_ERR_STATE, "more than 128 distinct proof shapes in one call"); break; }
        rc = run_device_shape(D, in, cur, verdicts, env, next);
        if (!rc && next.size() >= cur.size()) { for (size_t q : next) verdicts[q] = 0; rc = fail(MINA_ERR_STATE, "a pass over deferred proofs made no progress"); break; }
    }
    if (rc) for (size_t q : idx) verdicts[q] = 0;
    return rc;
}

// n proofs over the devices of the process: contiguous shards, one host thread per extra device
int verify_state_many(const CallIn &in, size_t n, uint8_t *verdicts) {
    std::vector<Device *> devs; CallEnv env;
    { std::lock_guard<std::mutex> lk(g_mu); devs = devices(); env.flags = g_flags; env.network = g_network; }
    env.tu = mb_tune();
    for (size_t i = 0; i < n; ++i) verdicts[i] = 0;
    if (devs.empty()) return MINA_ERR_HIP;
    if (n == 0) return MINA_OK;
    const size_t G = devs.size();
    const size_t min_shard = std::max<size_t>(1, env.tu.min_shard);
    const size_t use = std::max<size_t>(1, std::min(G, n / min_shard));    // tiny calls stay on one device (dealt round-robin)
    if (use == 1) {
        std::vector<size_t> idx(n); for (size_t i = 0; i < n; ++i) idx[i] = i;
        return run_device(*devs[g_rr.fetch_add(1) % G], in, idx, verdicts, env);
    }
    std::vector<int> rcs(use, MINA_OK); std::vector<std::thread> th;
    auto shard = [&](size_t g) {
        const size_t lo = n * g / use, hi = n * (g + 1) / use;
        std::vector<size_t> idx(hi - lo); for (size_t i = lo; i < hi; ++i) idx[i - lo] = i;
        rcs[g] = run_device(*devs[g], in, idx, verdicts, env);
    };
    for (size_t g = 1; g < use; ++g) th.emplace_back(shard, g);
    shard(0);
    for (auto &t : th) t.join();
    for (int rc : rcs) if (rc) return rc;
    return MINA_OK;
}

// single-proof diagnostic form: one job per step so that every step gets its own bit
int state_checks(const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len, uint32_t *passed_out, uint32_t *ran_out) {
    Device *D; uint32_t flags; int network;
    { std::lock_guard<std::mutex> lk(g_mu); auto &ds = devices(); if (ds.empty()) return MINA_ERR_HIP; D = ds[0]; flags = g_flags; network = g_network; }
    const Config cf = read_config(*D, flags | MINA_VERIFY_ALLOW_SURROGATE, network);
    Shape sh; sh.kimchi = cf.kimchi; sh.statements = cf.kimchi && cf.statements; sh.k = cf.k; sh.network = cf.network; sh.feature_aware = cf.feature_aware;
    uint32_t passed = 0, ran = MINA_CHECK_FORMAT;
    *passed_out = 0; *ran_out = ran;
    if (sh.statements) {
        mw::StateProofContainer &box = tl_box();
        mw::Bincode cur(proof, proof ? proof_len : 0);
        if (!proof || !mw::read_wrap_proof(cur, box.tip_proof)) return MINA_OK;
        sh.n_old = (uint32_t)std::min<size_t>(box.tip_proof.step_old_bulletproof_challenges.size(), 4); sh.n_ev = (uint32_t)std::min<size_t>(std::max<size_t>(box.tip_proof.prev_evals.size(), 43), 62);
    }
    Layout lay; lay.build(sh, 1);
    std::vector<uint8_t> stage(lay.total + 256);
    HostBits hb;
    parse_into(sh, lay, stage.data(), 0, proof, proof_len, pub, pub_len, hb, D->c);
    if (!hb.parsed) return MINA_OK;
    passed |= MINA_CHECK_FORMAT; ran |= MINA_CHECK_LEDGER | MINA_CHECK_CONSENSUS;
    if (hb.ledger) passed |= MINA_CHECK_LEDGER;
    if (hb.consensus) passed |= MINA_CHECK_CONSENSUS;
    *lay.at(stage.data(), S_PRE, 0) = 1;                               // the host checks have their own bits here
    if (!draw_randomisers(sh, lay, stage.data(), 1)) return fail(MINA_ERR_STATE, "no entropy for the folding randomisers");
    std::lock_guard<std::mutex> lk(D->mu);
    mina_ctx *c = D->c;
    auto leg = [&](bool st, bool acc, bool kim, uint32_t bit, bool extra_ok) -> int {
        JobStructs js; make_jobs(sh, lay, stage.data(), 1, st, acc, kim, js);
        uint8_t v = 0;
        int rc = mina_state_job_batch(c, &js.j, &v);
        if (rc) return rc;
        ran |= bit; if (v && extra_ok) passed |= bit;
        return MINA_OK;
    };
    int rc;
    if ((rc = leg(true, false, false, MINA_CHECK_CHAIN, true))) return rc;
    if ((rc = leg(false, true, false, MINA_CHECK_ACCUMULATOR, true))) return rc;
    if (sh.kimchi) {
        if (hb.shape) { if ((rc = leg(false, false, true, MINA_CHECK_KIMCHI, true))) return rc; }
        else ran |= MINA_CHECK_KIMCHI;                                 // wrong shape for the installed index / a lookup feature switched on: the step ran and failed
    }
    *passed_out = passed; *ran_out = ran;
    return MINA_OK;
}
}  // namespace

extern "C" int mina_verify_state_checks(const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len, uint32_t *passed_mask, uint32_t *ran_mask) {
    if (!passed_mask || !ran_mask) return fail(MINA_ERR_ARG, "null argument");
    return state_checks(proof, proof_len, pub, pub_len, passed_mask, ran_mask);
}

// the merged job of single-proof callers (and of small batches)
static void exec_state_calls(std::vector<PendingCall *> &job) {
    const size_t n = job.size();
    std::vector<const uint8_t *> pr(n), pu(n); std::vector<size_t> pl(n), ul(n); std::vector<uint8_t> v(n, 0);
    for (size_t i = 0; i < n; ++i) { pr[i] = job[i]->proof; pl[i] = job[i]->proof_len; pu[i] = job[i]->pub; ul[i] = job[i]->pub_len; }
    CallIn in{pr.data(), pl.data(), pu.data(), ul.data()};
    const int rc = verify_state_many(in, n, v.data());
    for (size_t i = 0; i < n; ++i) { job[i]->verdict = rc == MINA_OK ? v[i] : 0; job[i]->rc = rc; }
}
extern "C" int mina_verify_state_batch(size_t n, const uint8_t *const *proofs, const size_t *proof_lens, const uint8_t *const *pubs, const size_t *pub_lens,
                                       uint8_t *verdicts_out) {
    if (n && (!proofs || !proof_lens || !pubs || !pub_lens || !verdicts_out)) return fail(MINA_ERR_ARG, "null argument");
    // a small batch is a latency-bound job whatever its size (64 proofs: 21 ms, 1024: 25 ms): concurrent small batches share jobs the way
    // single-proof calls do (16 callers of 64 proofs: 8.7 k -> proofs/s of one 1024-proof job per ~25 ms)
    const mina_verify_tuning tune = mb_tune();
    if (n && n <= tune.merge_batch_max && tune.merge) {
        std::vector<PendingCall> calls(n);
        for (size_t i = 0; i < n; ++i) { calls[i].proof = proofs[i]; calls[i].proof_len = proof_lens[i]; calls[i].pub = pubs[i]; calls[i].pub_len = pub_lens[i]; }
        g_state_calls.run_group(exec_state_calls, calls.data(), n);
        int rc = MINA_OK;
        for (size_t i = 0; i < n; ++i) { verdicts_out[i] = calls[i].verdict; if (calls[i].rc && !rc) rc = calls[i].rc; }
        if (rc) for (size_t i = 0; i < n; ++i) verdicts_out[i] = 0;
        return rc;
    }
    CallIn in{proofs, proof_lens, pubs, pub_lens};
    return verify_state_many(in, n, verdicts_out);
}
extern "C" bool mina_verify_state(const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len) {
    PendingCall me{proof, proof_len, pub, pub_len};
    return g_state_calls.run(exec_state_calls, me);
}

// the `--save-proof` form (core/src/aligned.rs:60-69): two files holding exactly the two byte strings
static bool read_file(const char *path, std::vector<uint8_t> &out, size_t cap) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    out.clear(); uint8_t buf[65536]; size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) { out.insert(out.end(), buf, buf + k); if (out.size() > cap) { fclose(f); return false; } }
    fclose(f);
    return true;
}
extern "C" bool mina_verify_state_files(const char *proof_path, const char *pub_path) {
    std::vector<uint8_t> p, q;
    if (!proof_path || !pub_path || !read_file(proof_path, p, 1u << 20) || !read_file(pub_path, q, 1u << 16)) return false;
    return mina_verify_state(p.data(), p.size(), q.data(), q.size());
}

// ------------------------------------------------------------------------------------------------ container introspection (tests, tooling)
// Flattens a serialized wrap proof into the fixed order the kernels consume; see include/mina_verify.h for the layout.
extern "C" int mina_wrap_proof_flatten(const uint8_t *bytes, size_t len, int encoding, uint8_t *out, size_t cap, size_t *out_len, size_t *consumed) {
    if (!bytes || !out_len) return fail(MINA_ERR_ARG, "null argument");
    mw::WrapProof p; bool ok; size_t used;
    if (encoding == MINA_ENC_BINPROT) { mw::Binprot c(bytes, len); ok = mw::read_wrap_proof(c, p); used = c.pos; }
    else if (encoding == MINA_ENC_BINCODE) { mw::Bincode c(bytes, len); ok = mw::read_wrap_proof(c, p); used = c.pos; }
    else return fail(MINA_ERR_ARG, "bad encoding");
    if (!ok) return fail(MINA_ERR_FORMAT, "malformed wrap proof");
    if (consumed) *consumed = used; else if (used != len) return fail(MINA_ERR_FORMAT, "trailing bytes after the proof");
    std::vector<uint8_t> o;
    auto u32 = [&](uint32_t v) { for (int i = 0; i < 4; ++i) o.push_back((uint8_t)(v >> (8 * i))); };
    auto chal = [&](const mw::Chal128 &c) { uint8_t b[16]; chal_bytes(c, b); o.insert(o.end(), b, b + 16); };
    auto b32 = [&](const mw::B32 &x) { o.insert(o.end(), x.b, x.b + 32); };
    auto pt = [&](const mw::Pt &q) { b32(q.x); b32(q.y); };
    chal(p.alpha); chal(p.beta); chal(p.gamma); chal(p.zeta); o.push_back(p.has_joint_combiner); chal(p.joint_combiner);
    for (int i = 0; i < 8; ++i) o.push_back(p.feature_flags[i]);
    for (int i = 0; i < 16; ++i) chal(p.bulletproof_challenges[i]);
    o.push_back(p.proofs_verified); o.push_back(p.domain_log2);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) o.push_back((uint8_t)(p.sponge_digest_before_evaluations[i] >> (8 * j)));
    pt(p.challenge_polynomial_commitment);
    for (int a = 0; a < 2; ++a) for (int i = 0; i < 15; ++i) chal(p.old_bulletproof_challenges[a][i]);
    u32((uint32_t)p.step_challenge_polynomial_commitments.size()); for (auto &q : p.step_challenge_polynomial_commitments) pt(q);
    u32((uint32_t)p.step_old_bulletproof_challenges.size()); for (auto &a : p.step_old_bulletproof_challenges) for (int i = 0; i < 16; ++i) chal(a[i]);
    auto ev = [&](const mw::EvalPair &e) { u32((uint32_t)e.zeta.size()); for (auto &x : e.zeta) b32(x); u32((uint32_t)e.zeta_omega.size()); for (auto &x : e.zeta_omega) b32(x); };
    ev(p.prev_public_input);
    u32((uint32_t)p.prev_evals.size()); for (auto &e : p.prev_evals) ev(e);
    for (uint8_t f : p.prev_evals_present) o.push_back(f);
    b32(p.prev_ft_eval1);
    for (int i = 0; i < 15; ++i) pt(p.w_comm[i]); pt(p.z_comm); for (int i = 0; i < 7; ++i) pt(p.t_comm[i]);
    for (int i = 0; i < 15; ++i) { b32(p.w_eval[i][0]); b32(p.w_eval[i][1]); }
    for (int i = 0; i < 15; ++i) { b32(p.coefficients_eval[i][0]); b32(p.coefficients_eval[i][1]); }
    b32(p.z_eval[0]); b32(p.z_eval[1]);
    for (int i = 0; i < 6; ++i) { b32(p.s_eval[i][0]); b32(p.s_eval[i][1]); }
    for (int i = 0; i < 6; ++i) { b32(p.selector_eval[i][0]); b32(p.selector_eval[i][1]); }
    b32(p.ft_eval1);
    u32((uint32_t)p.lr.size()); for (auto &q : p.lr) { pt(q.first); pt(q.second); }
    b32(p.z1); b32(p.z2); pt(p.delta); pt(p.sg);
    *out_len = o.size();
    if (out) { if (cap < o.size()) return fail(MINA_ERR_ARG, "output buffer too small"); memcpy(out, o.data(), o.size()); }
    return MINA_OK;
}

// splits a bincode MinaStateProof: *proof_len = bytes of the leading wrap proof; state_offsets[17] / state_lens[17] locate the states
extern "C" int mina_state_proof_split(const uint8_t *bytes, size_t len, size_t *proof_len, size_t *state_offsets, size_t *state_lens) {
    if (!bytes || !proof_len || !state_offsets || !state_lens) return fail(MINA_ERR_ARG, "null argument");
    mw::Bincode c(bytes, len);
    mw::WrapProof p;
    if (!mw::read_wrap_proof(c, p)) return fail(MINA_ERR_FORMAT, "malformed wrap proof");
    *proof_len = c.pos;
    for (int i = 0; i < MINA_STATES_PER_PROOF; ++i) {
        mw::ProtocolState s; state_offsets[i] = c.pos;
        if (!mw::read_protocol_state(c, s)) return fail(MINA_ERR_FORMAT, "malformed protocol state in the container");
        state_lens[i] = c.pos - state_offsets[i];
    }
    if (c.pos != len) return fail(MINA_ERR_FORMAT, "trailing bytes after MinaStateProof");
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------ Proof of Account
extern "C" int mina_verify_account_checks(const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len, uint32_t *passed_mask, uint32_t *ran_mask) {
    if (!passed_mask || !ran_mask) return fail(MINA_ERR_ARG, "null argument");
    Device *D;
    { std::lock_guard<std::mutex> lk(g_mu); auto &ds = devices(); if (ds.empty()) return MINA_ERR_HIP; D = ds[0]; }
    std::lock_guard<std::mutex> lk(D->mu);
    return mina_verify_account_ctx(D->c, 1, &proof, &proof_len, &pub, &pub_len, passed_mask, ran_mask);
}
static int account_batch_direct(size_t n, const uint8_t *const *proofs, const size_t *proof_lens, const uint8_t *const *pubs, const size_t *pub_lens, uint8_t *verdicts_out) {
    for (size_t i = 0; i < n; ++i) verdicts_out[i] = 0;
    if (n == 0) return MINA_OK;
    std::vector<uint32_t> passed(n), ran(n);
    {
        Device *D; uint32_t flags;
        { std::lock_guard<std::mutex> lk(g_mu); auto &ds = devices(); if (ds.empty()) return MINA_ERR_HIP; D = ds[g_rr.fetch_add(1) % ds.size()]; flags = g_flags; }
        { std::lock_guard<std::mutex> lk(D->mu);
          if ((D->c->pparams_surrogate[0] || D->c->pparams_surrogate[1]) && !(flags & MINA_VERIFY_ALLOW_SURROGATE)) return MINA_OK; }   // surrogate Poseidon tables: refuse (see read_config)
        // one account job per device at a time, on a lane of its own (the last pipeline lane: no slot, no culprit search uses it); the device's lock is
        // taken only while its kernels are queued, so the state-proof pipeline of the same process keeps its jobs coming
        std::lock_guard<std::mutex> acct(D->acct_mu);
        int rc = mb_verify_account_on(D->c, n, proofs, proof_lens, pubs, pub_lens, passed.data(), ran.data(), &D->c->lanes[MB_PIPE_LANES - 1], &D->mu);
        if (rc) return rc;
    }
    const uint32_t need = MINA_CHECK_FORMAT | MINA_CHECK_ACCOUNT_ABI | MINA_CHECK_MERKLE;
    for (size_t i = 0; i < n; ++i) verdicts_out[i] = (passed[i] & need) == need ? 1 : 0;
    return MINA_OK;
}
static void exec_account_calls(std::vector<PendingCall *> &job) {
    const size_t n = job.size();
    std::vector<const uint8_t *> pr(n), pu(n); std::vector<size_t> pl(n), ul(n); std::vector<uint8_t> v(n, 0);
    for (size_t i = 0; i < n; ++i) { pr[i] = job[i]->proof; pl[i] = job[i]->proof_len; pu[i] = job[i]->pub; ul[i] = job[i]->pub_len; }
    const int rc = account_batch_direct(n, pr.data(), pl.data(), pu.data(), ul.data(), v.data());
    for (size_t i = 0; i < n; ++i) { job[i]->verdict = rc == MINA_OK ? v[i] : 0; job[i]->rc = rc; }
}
extern "C" int mina_verify_account_batch(size_t n, const uint8_t *const *proofs, const size_t *proof_lens, const uint8_t *const *pubs, const size_t *pub_lens,
                                         uint8_t *verdicts_out) {
    if (n && (!proofs || !proof_lens || !pubs || !pub_lens || !verdicts_out)) return fail(MINA_ERR_ARG, "null argument");
    // BASELINE C4's batch of 256 is a latency-bound job (a dependent chain of ~70 permutations: 8 ms for 1 or for 1024 proofs): concurrent small
    // batches share jobs the way single-proof calls do
    const mina_verify_tuning tune = mb_tune();
    if (n && n <= tune.merge_batch_max && tune.merge) {
        std::vector<PendingCall> calls(n);
        for (size_t i = 0; i < n; ++i) { calls[i].proof = proofs[i]; calls[i].proof_len = proof_lens[i]; calls[i].pub = pubs[i]; calls[i].pub_len = pub_lens[i]; }
        g_account_calls.run_group(exec_account_calls, calls.data(), n);
        int rc = MINA_OK;
        for (size_t i = 0; i < n; ++i) { verdicts_out[i] = calls[i].verdict; if (calls[i].rc && !rc) rc = calls[i].rc; }
        if (rc) for (size_t i = 0; i < n; ++i) verdicts_out[i] = 0;
        return rc;
    }
    return account_batch_direct(n, proofs, proof_lens, pubs, pub_lens, verdicts_out);
}
extern "C" bool mina_verify_account(const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len) {
    PendingCall me{proof, proof_len, pub, pub_len};
    return g_account_calls.run(exec_account_calls, me);
}
extern "C" bool mina_verify_account_files(const char *proof_path, const char *pub_path) {
    std::vector<uint8_t> p, q;
    if (!proof_path || !pub_path || !read_file(proof_path, p, 1u << 20) || !read_file(pub_path, q, 1u << 20)) return false;
    return mina_verify_account(p.data(), p.size(), q.data(), q.size());
}

// ------------------------------------------------------------------------------------------------ the reference's own symbol names
// What Aligned's operator (cgo) and batcher (Rust) bind: README.md:277-279 (`verify_mina_state_ffi`), :358-362 (`verify_account_inclusion_ffi`);
// proving-system tags core/src/aligned.rs:40,53.  Fixed-size caller-owned buffers + used lengths: a length beyond the buffer is
// `false`, like every other failure; nothing unwinds (the callees catch nothing because nothing throws across them: allocation failure aside, the
// pipeline reports through return codes).  The `_u32` forms are the length type of later Aligned versions.
extern "C" bool verify_mina_state_ffi(const uint8_t *proof_buffer, size_t proof_len, const uint8_t *pub_input_buffer, size_t pub_input_len) {
    if (!proof_buffer || !pub_input_buffer || proof_len > MINA_FFI_MAX_PROOF_SIZE || pub_input_len > MINA_FFI_MAX_PUB_INPUT_SIZE) return false;
    try { return mina_verify_state(proof_buffer, proof_len, pub_input_buffer, pub_input_len); } catch (...) { return false; }
}
extern "C" bool verify_account_inclusion_ffi(const uint8_t *proof_buffer, size_t proof_len, const uint8_t *pub_input_buffer, size_t pub_input_len) {
    // (no cap of its own: the account operator's buffer sizes are not in the tree -- a zkApp account's ABI form can exceed 6 KiB -- and only `len` bytes are read)
    if (!proof_buffer || !pub_input_buffer) return false;
    try { return mina_verify_account(proof_buffer, proof_len, pub_input_buffer, pub_input_len); } catch (...) { return false; }
}
extern "C" bool verify_mina_state_ffi_u32(const uint8_t *proof_buffer, uint32_t proof_len, const uint8_t *pub_input_buffer, uint32_t pub_input_len) {
    return verify_mina_state_ffi(proof_buffer, (size_t)proof_len, pub_input_buffer, (size_t)pub_input_len);
}
extern "C" bool verify_account_inclusion_ffi_u32(const uint8_t *proof_buffer, uint32_t proof_len, const uint8_t *pub_input_buffer, uint32_t pub_input_len) {
    return verify_account_inclusion_ffi(proof_buffer, (size_t)proof_len, pub_input_buffer, (size_t)pub_input_len);
}
