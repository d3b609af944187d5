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
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "ctx.h"
#include "wire_proof.h"
#include "poseidon_tables.inc"

int mb_pack_protocol_state(const mw::ProtocolState &s, uint8_t *record, uint32_t *n_body_fields, mina_protocol_state_info *info);   // api_state.hip
int mb_kimchi_available(mina_ctx *c);                                                                                                  // api_kimchi.hip
int mb_kimchi_fill_jobs(mina_ctx *c, const mw::WrapProof *const *proofs, const uint8_t *const *tip_hashes, size_t n, mina_state_jobs *jobs,
                        std::vector<std::vector<uint8_t>> &storage, std::vector<uint8_t> &statement_ok);

extern "C" const char *mina_poseidon_params_name(void) { return MB_POSEIDON_SET_NAME; }
extern "C" int mina_poseidon_install_default_params(mina_ctx *c) {
    if (!c) return fail(MINA_ERR_ARG, "null argument");
    for (int f = 0; f < 2; ++f) { int rc = mina_poseidon_set_params(c, f, MB_POSEIDON_TABLES[f]); if (rc) return rc; }
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------ process-wide context
static std::mutex g_mu;
static mina_ctx *g_ctx = nullptr;
static uint32_t g_flags = 0;

static mina_ctx *global_ctx() {            // caller holds g_mu
    if (g_ctx) return g_ctx;
    int dev = 0;
    if (const char *e = getenv("MINA_VERIFY_DEVICE")) dev = atoi(e);
    mina_ctx *c = nullptr;
    if (mina_ctx_create(dev, &c) != MINA_OK) return nullptr;
    if (mina_poseidon_install_default_params(c) != MINA_OK || mina_srs_create(c, CURVE_VESTA, 1u << 16) != MINA_OK ||
        mina_srs_create(c, CURVE_PALLAS, 1u << 16) != MINA_OK) { mina_ctx_destroy(c); return nullptr; }
    g_ctx = c;
    return g_ctx;
}
extern "C" int mina_verify_configure(uint32_t flags) { std::lock_guard<std::mutex> lk(g_mu); g_flags = flags; return MINA_OK; }
extern "C" int mina_verify_shutdown(void) { std::lock_guard<std::mutex> lk(g_mu); if (g_ctx) mina_ctx_destroy(g_ctx); g_ctx = nullptr; return MINA_OK; }
// the process-wide context, e.g. to install a verifier index or different Poseidon tables; NULL if no GPU / set-up failed
extern "C" mina_ctx *mina_verify_global_ctx(void) { std::lock_guard<std::mutex> lk(g_mu); return global_ctx(); }

// ------------------------------------------------------------------------------------------------ merging of concurrent single-proof calls
// The reference's entry points take ONE proof and are called from many goroutines / tokio tasks at once.  One proof is
// a 25 ms dependent chain that leaves the chip idle, so concurrent callers are merged (group commit): the first caller runs a job with
// everything queued at that moment; calls arriving while it runs wait and leave together as the next job, led by one of them.  A lone
// caller pays nothing; N concurrent callers share one job of N proofs.  Verdicts are per proof either way (the batch entry points
// isolate failing proofs).  MINA_VERIFY_NO_MERGE=1 sends every call through on its own; MINA_VERIFY_LINGER_US (default 500) is how long
// the leader of a job waits for the callers of the previous job to come back before it leaves.
namespace {
struct PendingCall { const uint8_t *proof; size_t proof_len; const uint8_t *pub; size_t pub_len; void *parsed = nullptr; uint8_t verdict = 0; bool done = false; };
typedef void (*exec_fn_t)(std::vector<PendingCall *> &job);           // sets `verdict` of every call of the job
struct CallMerger {
    std::mutex mu; std::condition_variable cv, arrived; std::vector<PendingCall *> waiting; bool leader = false;
    size_t last_job = 0;                 // calls merged into the previous job: its callers return together and call again within microseconds
    static constexpr size_t MAX_JOB = 8192;
    bool run(exec_fn_t exec, PendingCall &me) {
        static const bool off = getenv("MINA_VERIFY_NO_MERGE") != nullptr;
        if (off) { std::vector<PendingCall *> job{&me}; exec(job); return me.verdict == 1; }
        static const long linger_us = getenv("MINA_VERIFY_LINGER_US") ? atol(getenv("MINA_VERIFY_LINGER_US")) : 500;
        std::unique_lock<std::mutex> lk(mu);
        waiting.push_back(&me);
        arrived.notify_one();
        while (!me.done) {
            if (leader) { cv.wait(lk); continue; }
            leader = true;                                                    // lead the next job: everything queued so far (this call included, unless MAX_JOB cut it off)
            if (last_job > 1 && linger_us > 0) {                              // the callers of the job that just ended are on their way back: give them a moment,
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(linger_us);   // else this call leaves alone and they wait two latencies
                while (waiting.size() < last_job && arrived.wait_until(lk, deadline) != std::cv_status::timeout) {}
            }
            const size_t n = std::min(waiting.size(), MAX_JOB);
            std::vector<PendingCall *> job(waiting.begin(), waiting.begin() + n);
            waiting.erase(waiting.begin(), waiting.begin() + n);
            lk.unlock();
            exec(job);
            lk.lock();
            for (PendingCall *p : job) p->done = true;
            last_job = n;
            leader = false;
            cv.notify_all();
        }
        return me.verdict == 1;
    }
};
CallMerger g_state_calls, g_account_calls;
}  // namespace

// ------------------------------------------------------------------------------------------------ Proof of State
namespace {
struct ParsedState {
    ParsedState() {}                     // user-provided: a vector of these is NOT zero-filled (40 KB each)
    bool format_ok = false, ledger_ok = false, consensus_ok = false;
    mina_state_pub_inputs pub;
    mw::StateProofContainer box;
    uint8_t records[MINA_STATES_PER_PROOF][MINA_PSTATE_SLOTS * 32]; uint32_t nfields[MINA_STATES_PER_PROOF];
    mina_protocol_state_info info[MINA_STATES_PER_PROOF];
};

void chal_bytes(const mw::Chal128 &c, uint8_t *o) { for (int i = 0; i < 8; ++i) { o[i] = (uint8_t)(c.lo >> (8 * i)); o[8 + i] = (uint8_t)(c.hi >> (8 * i)); } }

// host part of one proof: FORMAT, LEDGER, CONSENSUS
void parse_state(const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len, ParsedState &ps) {
    if (!proof || !pub) return;
    if (mina_parse_state_pub_inputs(pub, pub_len, &ps.pub) != MINA_OK) return;
    if (!mw::read_state_proof(proof, proof_len, ps.box)) return;
    for (int i = 0; i < MINA_STATES_PER_PROOF; ++i)
        if (mb_pack_protocol_state(ps.box.states[i], ps.records[i], &ps.nfields[i], &ps.info[i]) != MINA_OK) return;
    if (ps.box.tip_proof.lr.empty() || ps.box.tip_proof.lr.size() > 20 || ps.box.tip_proof.step_challenge_polynomial_commitments.size() != ps.box.tip_proof.step_old_bulletproof_challenges.size()) return;
    ps.format_ok = true;
    bool ledger = true;
    for (int i = 0; i < 16; ++i) ledger = ledger && memcmp(ps.pub.candidate_chain_ledger_hashes[i], ps.info[i].snarked_ledger_hash, 32) == 0;
    ps.ledger_ok = ledger;
    // chain selection between the bridge tip (state 16) and the candidate tip (state 15); tie-breaks use the hashes the public
    // input names (the CHAIN step proves them) and Blake2b-256 of the last VRF output (`hashLastVRF`)
    mina_consensus_state tip = ps.info[16].consensus, cand = ps.info[15].consensus;
    memcpy(tip.state_hash, ps.pub.bridge_tip_state_hash, 32); memcpy(cand.state_hash, ps.pub.candidate_chain_state_hashes[15], 32);
    blake2b_short(ps.box.states[16].last_vrf_output.data(), 32, tip.last_vrf_output_hash, 32);
    blake2b_short(ps.box.states[15].last_vrf_output.data(), 32, cand.last_vrf_output_hash, 32);
    mina_consensus_params cp{ps.info[15].slots_per_sub_window, ps.info[15].sub_windows_per_window};
    int sel = 0;
    if (ps.info[16].sub_windows_per_window == cp.sub_windows_per_window && cp.sub_windows_per_window >= 1 && cp.sub_windows_per_window <= MINA_MAX_SUB_WINDOWS &&
        cp.slots_per_sub_window >= 1 && mina_consensus_select_secure_chain(&cp, &tip, &cand, &sel) == MINA_OK)
        ps.consensus_ok = sel == 1;
}

// GPU part of n proofs (those whose FORMAT passed): CHAIN + ACCUMULATOR (+ KIMCHI) through the Proof-of-State job
// `masks`: one job per step so that every step gets its own bit (mina_verify_state_checks); otherwise ONE job with all legs -- the
// verdict-only entry points need nothing finer, and the legs overlap on the GPU
int run_state_jobs(mina_ctx *c, std::vector<ParsedState *> &ps, std::vector<uint32_t> &passed, std::vector<uint32_t> &ran, bool masks) {
    const size_t n = ps.size();
    if (n == 0) return MINA_OK;
    const auto tg = std::chrono::steady_clock::now();
    std::unique_ptr<uint8_t[]> recs_(new uint8_t[n * MINA_STATES_PER_PROOF * MINA_PSTATE_SLOTS * 32]);      // 34 KB per proof: not zero-filled, every byte is written below
    uint8_t *recs = recs_.get();
    std::vector<uint8_t> exp(n * MINA_STATES_PER_PROOF * 32), pre(n * 16 * 16), sg(n * 64), rho(n * 32);
    std::vector<uint32_t> nf(n * MINA_STATES_PER_PROOF);
    mb_parallel_for(n, [&](size_t b) {
        memcpy(&recs[b * sizeof ps[b]->records], ps[b]->records, sizeof ps[b]->records);
        memcpy(&nf[b * MINA_STATES_PER_PROOF], ps[b]->nfields, sizeof ps[b]->nfields);
        memcpy(&exp[b * MINA_STATES_PER_PROOF * 32], ps[b]->pub.candidate_chain_state_hashes, 512);
        memcpy(&exp[b * MINA_STATES_PER_PROOF * 32 + 512], ps[b]->pub.bridge_tip_state_hash, 32);
        const mw::WrapProof &w = ps[b]->box.tip_proof;
        for (int i = 0; i < 16; ++i) chal_bytes(w.bulletproof_challenges[i], &pre[(b * 16 + i) * 16]);
        memcpy(&sg[b * 64], w.challenge_polynomial_commitment.x.b, 32); memcpy(&sg[b * 64 + 32], w.challenge_polynomial_commitment.y.b, 32);
    });
    // batching randomisers of the folded accumulator check: SplitMix64 over the proof bytes' digest would make them unpredictable to
    // a prover; here a per-call counter-seeded stream (the folded check only needs them independent of the proofs' contents)
    { static uint64_t ctr = 0x6d696e61ULL; uint64_t st = (ctr += 0x9E3779B97F4A7C15ULL);
      for (size_t i = 0; i < rho.size(); i += 8) { uint64_t z = (st += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31; memcpy(&rho[i], &z, 8); }
      for (size_t b = 0; b < n; ++b) rho[b * 32 + 31] &= 0x3f; }
    auto run = [&](mina_state_jobs &j, std::vector<uint8_t> &v) { v.assign(n, 0); return mina_state_job_batch(c, &j, v.data()); };
    mina_state_jobs base{}; base.batch = n;
    int rc;
    std::vector<uint8_t> v;
    if (!masks) {
        mina_state_jobs j = base; std::vector<std::vector<uint8_t>> storage; std::vector<uint8_t> stmt_ok(n, 1);
        uint32_t steps = MINA_CHECK_CHAIN | MINA_CHECK_ACCUMULATOR;
        if (mb_kimchi_available(c)) {
            std::vector<const mw::WrapProof *> wp(n); std::vector<const uint8_t *> th(n);
            for (size_t b = 0; b < n; ++b) { wp[b] = &ps[b]->box.tip_proof; th[b] = ps[b]->pub.candidate_chain_state_hashes[15]; }
            if ((rc = mb_kimchi_fill_jobs(c, wp.data(), th.data(), n, &j, storage, stmt_ok))) return rc;
            steps |= MINA_CHECK_KIMCHI;
        }
        j.with_states = 1; j.state_records = recs; j.state_nfields = nf.data(); j.expected_hashes = exp.data();
        j.with_accumulator = 1; j.acc_k = 16; j.acc_prechallenges = pre.data(); j.acc_sg = sg.data(); j.acc_rho = rho.data();
        const auto tj = std::chrono::steady_clock::now();
        if ((rc = run(j, v))) return rc;
        if (getenv("MINA_VERIFY_TIMING")) fprintf(stderr, "mina_verify:   gather+fill %.2f ms, mina_state_job_batch %.2f ms\n", std::chrono::duration<double, std::milli>(tj - tg).count(),
                                                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj).count());
        for (size_t b = 0; b < n; ++b) { ran[b] |= steps; if (v[b] && stmt_ok[b]) passed[b] |= steps; }
        return MINA_OK;
    }
    {   // CHAIN
        mina_state_jobs j = base; j.with_states = 1; j.state_records = recs; j.state_nfields = nf.data(); j.expected_hashes = exp.data();
        if ((rc = run(j, v))) return rc;
        for (size_t b = 0; b < n; ++b) { ran[b] |= MINA_CHECK_CHAIN; if (v[b]) passed[b] |= MINA_CHECK_CHAIN; }
    }
    {   // ACCUMULATOR
        mina_state_jobs j = base; j.with_accumulator = 1; j.acc_k = 16; j.acc_prechallenges = pre.data(); j.acc_sg = sg.data(); j.acc_rho = rho.data();
        if ((rc = run(j, v))) return rc;
        for (size_t b = 0; b < n; ++b) { ran[b] |= MINA_CHECK_ACCUMULATOR; if (v[b]) passed[b] |= MINA_CHECK_ACCUMULATOR; }
    }
    if (mb_kimchi_available(c)) {   // KIMCHI: oracles + to_batch on the GPU, then the combined opening check
        std::vector<const mw::WrapProof *> wp(n); std::vector<const uint8_t *> th(n);
        for (size_t b = 0; b < n; ++b) { wp[b] = &ps[b]->box.tip_proof; th[b] = ps[b]->pub.candidate_chain_state_hashes[15]; }
        mina_state_jobs j = base; std::vector<std::vector<uint8_t>> storage; std::vector<uint8_t> stmt_ok;
        if ((rc = mb_kimchi_fill_jobs(c, wp.data(), th.data(), n, &j, storage, stmt_ok))) return rc;
        if ((rc = run(j, v))) return rc;
        for (size_t b = 0; b < n; ++b) { ran[b] |= MINA_CHECK_KIMCHI; if (v[b] && stmt_ok[b]) passed[b] |= MINA_CHECK_KIMCHI; }
    }
    return MINA_OK;
}

// n parsed proofs: host verdict bits, then ONE pass over the GPU for those whose FORMAT passed
int verify_parsed(std::vector<ParsedState *> &ps, uint32_t *passed_out, uint32_t *ran_out, bool masks) {
    const size_t n = ps.size();
    std::vector<uint32_t> passed(n, 0), ran(n, 0);
    std::vector<ParsedState *> live; std::vector<size_t> live_idx;
    for (size_t i = 0; i < n; ++i) {
        ran[i] |= MINA_CHECK_FORMAT;
        if (!ps[i]->format_ok) continue;
        passed[i] |= MINA_CHECK_FORMAT; ran[i] |= MINA_CHECK_LEDGER | MINA_CHECK_CONSENSUS;
        if (ps[i]->ledger_ok) passed[i] |= MINA_CHECK_LEDGER;
        if (ps[i]->consensus_ok) passed[i] |= MINA_CHECK_CONSENSUS;
        live.push_back(ps[i]); live_idx.push_back(i);
    }
    if (!live.empty()) {
        std::lock_guard<std::mutex> lk(g_mu);
        mina_ctx *c = global_ctx();
        if (!c) return MINA_ERR_HIP;
        std::vector<uint32_t> lp(live.size(), 0), lr(live.size(), 0);
        int rc = run_state_jobs(c, live, lp, lr, masks);
        if (rc) return rc;
        for (size_t k = 0; k < live.size(); ++k) { passed[live_idx[k]] |= lp[k]; ran[live_idx[k]] |= lr[k]; }
    }
    for (size_t i = 0; i < n; ++i) { passed_out[i] = passed[i]; ran_out[i] = ran[i]; }
    return MINA_OK;
}

int verify_state_many(size_t n, const uint8_t *const *proofs, const size_t *proof_lens, const uint8_t *const *pubs, const size_t *pub_lens,
                      uint32_t *passed_out, uint32_t *ran_out, bool masks) {
    static const bool timing = getenv("MINA_VERIFY_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t0 = now();
    std::vector<ParsedState> ps(n);
    // host side of every proof (parse both containers, flatten 17 states, ledger + consensus checks): independent, ~0.1 ms each -> threads
    mb_parallel_for(n, [&](size_t i) { parse_state(proofs[i], proof_lens[i], pubs[i], pub_lens[i], ps[i]); });
    std::vector<ParsedState *> ptr(n); for (size_t i = 0; i < n; ++i) ptr[i] = &ps[i];
    const auto t1 = now();
    int rc = verify_parsed(ptr, passed_out, ran_out, masks);
    if (timing) fprintf(stderr, "mina_verify: n=%zu parse %.2f ms, jobs %.2f ms\n", n, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(now() - t1).count());
    return rc;
}

bool verdict_of(uint32_t passed, uint32_t ran, uint32_t flags) {
    uint32_t need = MINA_CHECK_FORMAT | MINA_CHECK_LEDGER | MINA_CHECK_CHAIN | MINA_CHECK_CONSENSUS | MINA_CHECK_ACCUMULATOR | MINA_CHECK_KIMCHI;
    if ((flags & MINA_VERIFY_ALLOW_MISSING_KIMCHI) && !(ran & MINA_CHECK_KIMCHI)) need &= ~(uint32_t)MINA_CHECK_KIMCHI;
    return (passed & need) == need;
}
}  // namespace

extern "C" int mina_verify_state_checks(const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len, uint32_t *passed_mask, uint32_t *ran_mask) {
    if (!passed_mask || !ran_mask) return fail(MINA_ERR_ARG, "null argument");
    return verify_state_many(1, &proof, &proof_len, &pub, &pub_len, passed_mask, ran_mask, /*masks=*/true);
}

extern "C" int mina_verify_state_batch(size_t n, const uint8_t *const *proofs, const size_t *proof_lens, const uint8_t *const *pubs, const size_t *pub_lens,
                                       uint8_t *verdicts_out) {
    if (n && (!proofs || !proof_lens || !pubs || !pub_lens || !verdicts_out)) return fail(MINA_ERR_ARG, "null argument");
    std::vector<uint32_t> passed(n), ran(n);
    int rc = verify_state_many(n, proofs, proof_lens, pubs, pub_lens, passed.data(), ran.data(), /*masks=*/false);
    if (rc) { for (size_t i = 0; i < n; ++i) verdicts_out[i] = 0; return rc; }
    uint32_t flags; { std::lock_guard<std::mutex> lk(g_mu); flags = g_flags; }
    for (size_t i = 0; i < n; ++i) verdicts_out[i] = verdict_of(passed[i], ran[i], flags) ? 1 : 0;
    return MINA_OK;
}

// the merged job of single-proof callers: every caller parsed its own proof on its own thread before queueing
static void exec_state_calls(std::vector<PendingCall *> &job) {
    const size_t n = job.size();
    std::vector<ParsedState *> ps(n); for (size_t i = 0; i < n; ++i) ps[i] = (ParsedState *)job[i]->parsed;
    std::vector<uint32_t> passed(n), ran(n);
    const int rc = verify_parsed(ps, passed.data(), ran.data(), /*masks=*/false);
    uint32_t flags; { std::lock_guard<std::mutex> lk(g_mu); flags = g_flags; }
    for (size_t i = 0; i < n; ++i) job[i]->verdict = (rc == MINA_OK && verdict_of(passed[i], ran[i], flags)) ? 1 : 0;
}
extern "C" bool mina_verify_state(const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len) {
    std::unique_ptr<ParsedState> ps(new (std::nothrow) ParsedState);
    if (!ps) return false;
    parse_state(proof, proof_len, pub, pub_len, *ps);
    PendingCall me{proof, proof_len, pub, pub_len};
    me.parsed = ps.get();
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
    std::lock_guard<std::mutex> lk(g_mu);
    mina_ctx *c = global_ctx();
    if (!c) return MINA_ERR_HIP;
    return mina_verify_account_ctx(c, 1, &proof, &proof_len, &pub, &pub_len, passed_mask, ran_mask);
}
extern "C" int mina_verify_account_batch(size_t n, const uint8_t *const *proofs, const size_t *proof_lens, const uint8_t *const *pubs, const size_t *pub_lens,
                                         uint8_t *verdicts_out) {
    if (n && (!proofs || !proof_lens || !pubs || !pub_lens || !verdicts_out)) return fail(MINA_ERR_ARG, "null argument");
    for (size_t i = 0; i < n; ++i) verdicts_out[i] = 0;
    if (n == 0) return MINA_OK;
    std::vector<uint32_t> passed(n), ran(n);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        mina_ctx *c = global_ctx();
        if (!c) return MINA_ERR_HIP;
        int rc = mina_verify_account_ctx(c, n, proofs, proof_lens, pubs, pub_lens, passed.data(), ran.data());
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
    const int rc = mina_verify_account_batch(n, pr.data(), pl.data(), pu.data(), ul.data(), v.data());
    for (size_t i = 0; i < n; ++i) job[i]->verdict = rc == MINA_OK ? v[i] : 0;
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
