// host_core.hip -- the library's host-only infrastructure: the calling thread's error text, the one tuning struct, the CSPRNG, the host worker pool.
// No kernel, no device pointer: this file (with api_verify.hip, api_wire.hip, api_consensus.hip) also builds with plain g++ against the stand-in runtime of
// tests/fuzz/hip_stub -- the ThreadSanitizer tier of the boundary's host logic (tests/fuzz/tsan_boundary.cpp).  Until round 5 it was the
// head of api_core.hip.
#include "ctx.h"

#include <sys/random.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>

static thread_local std::string g_err = "";
int mb_fail(int code, const std::string &msg) { g_err = msg; return code; }

// ------------------------------------------------------------------------------------------------ tuning (include/mina_verify.h)
static mina_verify_tuning tuning_defaults() {
    mina_verify_tuning t; memset(&t, 0, sizeof t);
    t.struct_size = (uint32_t)sizeof t;
    t.chunk = 8192; t.single_max = 8192; t.slots = 4; t.window = 4; t.ahead = 0; t.early_min = 2048; t.early_sub = 1024; t.head_min = 6144; t.split_max = 4;
    t.chain_cus = 128; t.cu_period = 256; t.acc_mask = 0; t.hash_piece_waves = 1024; t.up_stream = 1; t.min_shard = 64; t.pace_us = 0;
    t.merge = 1; t.merge_batch_max = 512; t.linger_us = 500; t.max_jobs = 1;
    t.coop16_max = 64; t.coop8_max = 8192; t.coop8_per_call = 0; t.transcript_coop8_max = 0; t.ipa_coop8_max = 1024; t.kimchi_coop8_max = 1024;
    t.bpoly_mfma = 1; t.pubcomm_direct = 1; t.ipa_shared_points = 1; t.kimchi_shared_digest = 1; t.ipa_side_stream = 1; t.search_fan = 4; t.search_full = 0; t.msm_fp29 = 1; t.search_ctx = 1;
    t.dev_fork = 1; t.dev_chain_cus = 96; t.dev_piece_waves = 0; t.dev_hash_lds_kb = 0; t.dev_acc_lane = 0;
    return t;
}
// The environment switches of rounds 1 - 3 became fields of mina_verify_tuning (round 4).  A deployment that still exports one gets the DEFAULT now: say so, once,
// at load time, naming the field -- silence would look like the switch still worked.  mina_verify_retired_env() returns the count: what a strict
// deployment's start-up check reads.
static const struct { const char *env, *field; } RETIRED_ENV[] = {
    {"MINA_VERIFY_CHUNK", "chunk"}, {"MINA_VERIFY_SINGLE_MAX", "single_max"}, {"MINA_VERIFY_SLOTS", "slots"}, {"MINA_VERIFY_WINDOW", "window"}, {"MINA_VERIFY_AHEAD", "ahead"},
    {"MINA_VERIFY_EARLY_MIN", "early_min"}, {"MINA_VERIFY_EARLY_SUB", "early_sub"}, {"MINA_VERIFY_HEAD_MIN", "head_min"}, {"MINA_VERIFY_SPLIT_MAX", "split_max"},
    {"MINA_VERIFY_CHAIN_CUS", "chain_cus"}, {"MINA_VERIFY_CU_PERIOD", "cu_period"}, {"MINA_VERIFY_ACC_MASK", "acc_mask"}, {"MINA_VERIFY_HASH_PIECE", "hash_piece_waves"},
    {"MINA_VERIFY_UP_STREAM", "up_stream"}, {"MINA_VERIFY_MIN_SHARD", "min_shard"}, {"MINA_VERIFY_PACE_MS", "pace_us"}, {"MINA_VERIFY_NO_MERGE", "merge (= 0)"},
    {"MINA_VERIFY_MERGE_BATCH_MAX", "merge_batch_max"}, {"MINA_VERIFY_LINGER_US", "linger_us"}, {"MINA_VERIFY_MAX_JOBS", "max_jobs"}, {"MINA_COOP16_MAX", "coop16_max"},
    {"MINA_COOP8_MAX", "coop8_max"}, {"MINA_COOP8_PER_CALL", "coop8_per_call"}, {"MINA_TRANSCRIPT_COOP8_MAX", "transcript_coop8_max"}, {"MINA_IPA_COOP8_MAX", "ipa_coop8_max"},
    {"MINA_KIMCHI_COOP8_MAX", "kimchi_coop8_max"}, {"MINA_BPOLY_MFMA", "bpoly_mfma"}, {"MINA_PUBCOMM_GENERIC_MSM", "pubcomm_direct (= 0)"}, {"MINA_IPA_NO_SHARED", "ipa_shared_points (= 0)"},
    {"MINA_KIMCHI_OWN_DIGEST", "kimchi_shared_digest (= 0)"}, {"MINA_IPA_NO_SIDE_STREAM", "ipa_side_stream (= 0)"}, {"MINA_SEARCH_FAN", "search_fan"}, {"MINA_STATE_SEARCH_FULL", "search_full"},
    {"MINA_MSM_ATOMIC_SORT", "(removed: the LDS-only sort is the only one)"}, {"MINA_MSM_TASKS", "(removed)"}};
static int g_retired_env_seen = -1;
extern "C" int mina_verify_retired_env(void) {                      // how many retired tuning variables the environment still sets (the notice is printed on the first call)
    static std::once_flag once;
    std::call_once(once, [] {
        int n = 0;
        for (const auto &r : RETIRED_ENV)
            if (getenv(r.env)) {
                fprintf(stderr, "libminaverify: environment variable %s is no longer read -- set mina_verify_tuning.%s through mina_verify_configure_ex (include/mina_verify.h); the default is in effect\n", r.env, r.field);
                ++n;
            }
        g_retired_env_seen = n;
    });
    return g_retired_env_seen;
}
__attribute__((constructor)) static void mb_warn_retired_env() { (void)mina_verify_retired_env(); }

static mina_verify_tuning g_tune = tuning_defaults();
static std::mutex g_tune_mu;
mina_verify_tuning mb_tune() { std::lock_guard<std::mutex> lk(g_tune_mu); return g_tune; }
extern "C" void mina_verify_tuning_default(mina_verify_tuning *out) { if (out) *out = tuning_defaults(); }
extern "C" int mina_verify_tuning_get(mina_verify_tuning *out) { if (!out) return fail(MINA_ERR_ARG, "null argument"); *out = mb_tune(); return MINA_OK; }
extern "C" int mina_verify_configure_ex(const mina_verify_tuning *t) {
    mina_verify_tuning n = tuning_defaults();
    if (t) {
        if (t->struct_size != sizeof n) return fail(MINA_ERR_ARG, "mina_verify_tuning.struct_size does not match this library: start from mina_verify_tuning_default");
        n = *t;
        if (!n.chunk || !n.single_max || !n.slots || n.slots > 16 || !n.window || !n.early_min || !n.min_shard || !n.max_jobs || n.acc_mask > 2 || !n.cu_period || n.search_fan < 2 || n.search_fan > 32)
            return fail(MINA_ERR_ARG, "mina_verify_tuning: chunk, single_max, slots (<= 16), window, early_min, min_shard, max_jobs, cu_period must be positive; acc_mask <= 2; search_fan in 2..32");
    }
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tune = n;
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------ CSPRNG
bool mb_secure_random(void *buf, size_t n) {
    uint8_t *p = (uint8_t *)buf; size_t got = 0;
    while (got < n) {
        const ssize_t r = getrandom(p + got, n - got, 0);
        if (r > 0) { got += (size_t)r; continue; }
        if (r < 0 && errno == EINTR) continue;
        break;
    }
    if (got == n) return true;
    FILE *f = fopen("/dev/urandom", "rb");                       // kernels without the system call
    if (!f) return false;
    const size_t k = fread(p + got, 1, n - got, f);
    fclose(f);
    return got + k == n;
}

// ------------------------------------------------------------------------------------------------ host worker pool
struct MbPoolJob { std::function<void(size_t)> fn; size_t n = 0; std::atomic<size_t> next{0}, done{0}; std::mutex mu; std::condition_variable cv; };
namespace {
struct HostPool {
    std::mutex mu; std::condition_variable cv; std::deque<std::shared_ptr<MbPoolJob>> jobs; std::vector<std::thread> th;
    static void drain(MbPoolJob &j) {
        for (;;) {
            const size_t i = j.next.fetch_add(1);
            if (i >= j.n) return;
            j.fn(i);
            if (j.done.fetch_add(1) + 1 == j.n) { std::lock_guard<std::mutex> lk(j.mu); j.cv.notify_all(); }
        }
    }
    void worker() {
        for (;;) {
            std::shared_ptr<MbPoolJob> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    while (!jobs.empty() && jobs.front()->next.load() >= jobs.front()->n) jobs.pop_front();
                    if (!jobs.empty()) { j = jobs.front(); break; }
                    cv.wait(lk);
                }
            }
            drain(*j);
        }
    }
    explicit HostPool(size_t nt) { for (size_t t = 0; t < nt; ++t) { th.emplace_back([this] { worker(); }); th.back().detach(); } }
};
HostPool *host_pool() {                                              // never destroyed: its threads outlive static destructors
    static HostPool *p = [] {
        size_t nt;
        if (const char *e = getenv("MINA_HOST_THREADS")) nt = (size_t)std::max(1L, atol(e));
        else { const size_t hw = std::thread::hardware_concurrency(); nt = std::max<size_t>(1, std::min<size_t>(hw / 2, 64)); }
        return new HostPool(nt);
    }();
    return p;
}
}  // namespace
size_t mb_pool_threads() { return host_pool()->th.size(); }
std::shared_ptr<MbPoolJob> mb_pool_submit(size_t n, std::function<void(size_t)> fn) {
    auto j = std::make_shared<MbPoolJob>(); j->fn = std::move(fn); j->n = n;
    if (n == 0) return j;
    HostPool *p = host_pool();
    { std::lock_guard<std::mutex> lk(p->mu); p->jobs.push_back(j); }
    p->cv.notify_all();
    return j;
}
void mb_pool_wait(const std::shared_ptr<MbPoolJob> &j) {
    if (!j || j->n == 0 || j->done.load() >= j->n) return;
    HostPool::drain(*j);                                             // the caller works too
    std::unique_lock<std::mutex> lk(j->mu);
    j->cv.wait(lk, [&] { return j->done.load() >= j->n; });
}

extern "C" const char *mina_last_error(void) { return g_err.c_str(); }
