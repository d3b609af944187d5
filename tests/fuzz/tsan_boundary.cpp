// tsan_boundary.cpp -- the race / stress tier of the boundary's HOST logic (contract: "callable concurrently ... must be re-entrant",
// SURVEY.md 8b; the reference's callers: /root/reference/README.md:277-279 -- an operator's goroutines, a batcher's tokio tasks).
//
// What is built: the PRODUCT's own host sources -- mina_bridge_amd/csrc/api_verify.hip (slots, chunking, group commit of small callers, the shape vote, device
// sharding, the fallback that drains the device before a culprit search), host_core.hip (worker pool, tuning, error text), api_wire.hip and api_consensus.hip (parsers,
// fork choice) -- compiled with g++ -fsanitize=thread against a stand-in HIP runtime (hip_stub/hip/hip_runtime.h: streams run their commands at once, "device"
// memory is host memory).  What is stubbed, below: the DEVICE layer only -- context creation, the kernel pipelines (`mb_state_jobs_on_lane`, `mina_state_job_batch`,
// `mb_verify_account_on` ...) -- with fixed answers that depend on the job's own bytes:
//     * a proof whose `ft_eval1` starts with the four bytes "BAD!" fails the job's FOLDED opening check (every verdict of the job is then 0 and the boundary must
//       run its fallback, which finds exactly that proof);
//     * a proof whose protocol states do not parse is rejected by the real host parser before any device code.
// What runs: N state callers (batches of mixed sizes: single proofs that merge into shared jobs, small batches, multi-chunk calls), M account callers, an installer
// thread that re-installs the verifier / step index and flips the tuning while calls are in flight, and a bad proof in every K-th call -- for `seconds`, under
// ThreadSanitizer.  Every verdict is checked against what the stub device must produce.  Exit code 0 = no race report (TSAN aborts with 66), no wrong verdict.
//
//   make -C tests/fuzz tsan && tests/fuzz/tsan_boundary tests/golden/state_proofs_k15_bytes.json tests/golden/account_proofs_bytes.json 30
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../mina_bridge_amd/csrc/ctx.h"
#include "../../mina_bridge_amd/csrc/wire_proof.h"

// ------------------------------------------------------------------------------------------------ the stub device layer
static std::atomic<long> g_jobs{0}, g_searches{0}, g_account_jobs{0}, g_installs{0};
static bool marked_bad(const void *ft_eval1, size_t b) { return ft_eval1 && memcmp((const uint8_t *)ft_eval1 + 32 * b, "BAD!", 4) == 0; }

extern "C" int mina_ctx_create(int device_id, mina_ctx **out) {
    mina_ctx *c = new mina_ctx(); c->device = device_id; c->nlanes = 1;
    for (int i = 0; i < 4; ++i) (void)hipStreamCreateWithFlags(&c->lanes[i].stream, hipStreamNonBlocking);
    c->use_lane0(); *out = c; return MINA_OK;
}
int mb_ctx_create_view(mina_ctx *parent, mina_ctx **out) { int rc = mina_ctx_create(parent->device, out); if (!rc) mb_ctx_refresh_view(*out, parent); return rc; }
void mb_ctx_refresh_view(mina_ctx *v, mina_ctx *p) { v->have_kimchi = p->have_kimchi; v->kimchi_log2 = p->kimchi_log2; v->have_state_salts = p->have_state_salts; }   // (read under the parent's lock by the caller)
extern "C" void mina_ctx_destroy(mina_ctx *c) { if (!c) return; for (auto &l : c->lanes) { if (l.stream) (void)hipStreamDestroy(l.stream); l.release_all(); } delete c; }
extern "C" int mina_poseidon_set_params(mina_ctx *c, int field, const uint8_t *) { c->have_pparams[field] = true; c->pparams_surrogate[field] = false; return MINA_OK; }
extern "C" int mina_srs_create(mina_ctx *c, int curve, uint32_t depth) { c->srs[curve].depth = depth; return MINA_OK; }
int mb_poseidon_env_params(mina_ctx *) { return MINA_OK; }
extern "C" int mina_state_jobs_prepare(mina_ctx *c, uint32_t, uint32_t) { c->have_state_salts = true; return MINA_OK; }
// the installed indexes: flags the installer thread flips under the boundary's own locks (mina_verify_install_* takes g_mu and every device's mu)
extern "C" int mina_verifier_index_install(mina_ctx *c, const mina_verifier_index *ix) {
    c->have_kimchi = true; c->kimchi_log2 = ix->log2_domain; g_installs.fetch_add(1); return MINA_OK;
}
static std::atomic<int> g_step_installed{0};
extern "C" int mina_step_index_install(mina_ctx *, const mina_step_index *) { g_step_installed.store(1); g_installs.fetch_add(1); return MINA_OK; }
int mb_step_index_installed(mina_ctx *) { return g_step_installed.load(); }
int mb_step_index_feature_aware(mina_ctx *) { return 0; }
int mb_kimchi_available(mina_ctx *c) { return c->have_kimchi ? 1 : 0; }
extern "C" int mina_merkle_verify_batch(mina_ctx *, int, size_t n, uint32_t, const uint8_t *, const uint8_t *, const uint8_t *, const uint8_t *, uint8_t *ok) { for (size_t i = 0; i < n; ++i) ok[i] = 1; return MINA_OK; }

int mb_state_hashes_early(mina_ctx *c, Lane *LS, size_t ns_total, size_t lo, size_t cnt, const uint32_t *, const uint32_t *, hipEvent_t) {
    if (!LS || lo + cnt > ns_total) return mb_fail(MINA_ERR_ARG, "bad early state range");
    c->state_hashes_early = lo + cnt;       // (what the real one leaves for mb_state_jobs_on_lane: written under the device's lock, like the real field)
    return MINA_OK;
}
// the job on a lane: verdict words as the verdict kernel writes them -- d_verdicts[b] = precheck[b] AND no folded failure; flags = {opening fold ok, 0, accumulator fold ok, 0}
int mb_state_jobs_on_lane(mina_ctx *c, const mina_state_jobs *j, uint32_t *d_verdicts, uint32_t *d_flags, Lane *, Lane *, uint32_t *d_stmt_out, Lane *, uint32_t phase, StateJobCarry *carry) {
    if (phase != MB_JOB_ALL && !carry) return mb_fail(MINA_ERR_ARG, "a split job needs a carry");
    const size_t B = j->batch;
    Lane &L = *c->L;
    int rc;
    if ((rc = L.st_flags.ensure(16 * 4))) return rc;          // the lane's own scratch words, as in the real pipeline: two jobs on one lane at once would race HERE
    uint32_t *w = L.st_flags.as<uint32_t>();
    if (phase & MB_JOB_LEGS) {
        bool any_bad = false;
        if (j->with_ipa && j->kimchi) for (size_t b = 0; b < B; ++b) any_bad = any_bad || marked_bad(j->kimchi->ft_eval1, b);
        w[4] = any_bad ? 0u : 1u; w[8] = 1u;
        if (phase == MB_JOB_LEGS) { carry->ipa_v = w + 4; carry->acc_v = w + 8; return MINA_OK; }
    }
    const uint32_t ipa_ok = j->with_ipa ? (phase == MB_JOB_FINISH ? carry->ipa_v[0] : w[4]) : 1u, acc_ok = 1u;
    c->state_hashes_early = 0;
    for (size_t b = 0; b < B; ++b) {
        const uint32_t pre = j->precheck ? ((const uint8_t *)j->precheck)[b] : 1u;
        d_verdicts[b] = (pre && ipa_ok && acc_ok) ? 1u : 0u;
        if (d_stmt_out) d_stmt_out[b] = 1u;
    }
    if (d_flags) { d_flags[0] = ipa_ok; d_flags[1] = 0u; d_flags[2] = acc_ok; d_flags[3] = 0u; }
    g_jobs.fetch_add(1);
    return MINA_OK;
}
// the culprit search of a failed job (host-buffer form): every proof's own verdict
extern "C" int mina_state_job_batch(mina_ctx *, const mina_state_jobs *j, uint8_t *verdicts) {
    for (size_t b = 0; b < j->batch; ++b) {
        const uint8_t pre = j->precheck ? ((const uint8_t *)j->precheck)[b] : 1;
        verdicts[b] = (pre && !(j->with_ipa && j->kimchi && marked_bad(j->kimchi->ft_eval1, b))) ? 1 : 0;
    }
    g_searches.fetch_add(1);
    return MINA_OK;
}
// Proof of Account: every pair passes every check unless its public input starts with 0xBD (the harness' marker); queues on the given lane under the given mutex,
// as the real one does around its kernel launches
int mb_verify_account_on(mina_ctx *c, size_t n, const uint8_t *const *, const size_t *, const uint8_t *const *pubs, const size_t *pub_lens, uint32_t *passed, uint32_t *ran, Lane *lane, std::mutex *enq_mu) {
    for (size_t i = 0; i < n; ++i) {
        ran[i] = MINA_CHECK_FORMAT | MINA_CHECK_ACCOUNT_ABI | MINA_CHECK_MERKLE;
        passed[i] = (pub_lens[i] && pubs[i][0] == 0xBD) ? (MINA_CHECK_FORMAT | MINA_CHECK_ACCOUNT_ABI) : ran[i];
    }
    if (lane) {
        std::unique_lock<std::mutex> lk;
        if (enq_mu) lk = std::unique_lock<std::mutex>(*enq_mu);
        Lane *const keep = c->L; c->L = lane;                 // the real one switches the context's current lane while it queues: the field the device's lock protects
        (void)lane->tmp_a.ensure(64);
        c->L = keep;
    }
    g_account_jobs.fetch_add(1);
    return MINA_OK;
}
extern "C" int mina_verify_account_ctx(mina_ctx *c, size_t n, const uint8_t *const *proofs, const size_t *pl, const uint8_t *const *pubs, const size_t *ql, uint32_t *passed, uint32_t *ran) {
    return mb_verify_account_on(c, n, proofs, pl, pubs, ql, passed, ran, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------ the harness
static std::vector<std::string> json_strings(const std::string &text, const std::string &key) {     // every value of "key": "..." in a flat fixture file
    std::vector<std::string> out; const std::string pat = "\"" + key + "\": \"";
    for (size_t at = text.find(pat); at != std::string::npos; at = text.find(pat, at + 1)) { const size_t a = at + pat.size(), b = text.find('"', a); out.push_back(text.substr(a, b - a)); }
    return out;
}
static std::string unhex(const std::string &h) { std::string o(h.size() / 2, '\0'); for (size_t i = 0; i < o.size(); ++i) o[i] = (char)strtol(h.substr(2 * i, 2).c_str(), nullptr, 16); return o; }
static std::string unb64(const std::string &s) {
    static const std::string T = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string o; uint32_t acc = 0; int bits = 0;
    for (char ch : s) { if (ch == '=') break; const size_t v = T.find(ch); if (v == std::string::npos) continue; acc = (acc << 6) | (uint32_t)v; bits += 6; if (bits >= 8) { bits -= 8; o.push_back((char)((acc >> bits) & 0xff)); } }
    return o;
}
static std::string slurp(const char *path) { std::ifstream f(path, std::ios::binary); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }


int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s state_proofs_k15_bytes.json account_proofs_bytes.json seconds [state_callers account_callers bad_every]\n", argv[0]); return 2; }
    const double seconds = atof(argv[3]);
    const int n_state = argc > 4 ? atoi(argv[4]) : 6, n_acct = argc > 5 ? atoi(argv[5]) : 3, bad_every = argc > 6 ? atoi(argv[6]) : 5;
    setenv("MINA_VERIFY_DEVICES", "0,0", 1);                        // two logical contexts: a large call is cut into shards, one pipeline per device
    setenv("MINA_HOST_THREADS", "4", 1);
    const std::string st = slurp(argv[1]), ac = slurp(argv[2]);
    std::vector<std::string> proofs, pubs, aproofs, apubs;
    for (auto &h : json_strings(st, "proof")) proofs.push_back(unhex(h));
    for (auto &h : json_strings(st, "pub")) pubs.push_back(unhex(h));
    { auto p = json_strings(ac, "proof"), q = json_strings(ac, "pub"); for (size_t i = 0; i < p.size() && i < 32; ++i) { aproofs.push_back(unb64(p[i])); apubs.push_back(unb64(q[i])); } }
    if (proofs.size() < 4 || proofs.size() != pubs.size() || aproofs.empty()) { fprintf(stderr, "fixtures not readable\n"); return 2; }
    // the bad variant of every state proof: the four bytes "BAD!" over the start of ft_eval1.  ft_eval1 is found by parsing with the product's own reader and
    // searching for its 32 bytes in the serialized form (bincode writes field elements as their little-endian bytes)
    std::vector<std::string> bad_proofs;
    for (auto &p : proofs) {
        mw::StateProofContainer *box = new mw::StateProofContainer();
        mw::Bincode cur((const uint8_t *)p.data(), p.size());
        if (!mw::read_wrap_proof(cur, box->tip_proof)) { fprintf(stderr, "fixture proof does not parse\n"); return 2; }
        const std::string needle((const char *)box->tip_proof.ft_eval1.b, 32);
        const size_t at = p.find(needle);
        if (at == std::string::npos || p.find(needle, at + 1) != std::string::npos) { fprintf(stderr, "ft_eval1 not found exactly once in the serialized proof\n"); return 2; }
        std::string b = p; memcpy(&b[at], "BAD!", 4); bad_proofs.push_back(b);
        delete box;
    }
    std::string truncated = proofs[0].substr(0, proofs[0].size() - 3000);      // the protocol states cut short: the host parser rejects it alone
    mina_verify_configure(MINA_VERIFY_ALLOW_SURROGATE | MINA_VERIFY_ALLOW_UNBOUND_STATEMENT);   // (no step index: the kimchi step runs on the wrap index alone)
    static uint8_t zeros[8192] = {0};
    mina_verifier_index ix; memset(&ix, 0, sizeof ix);
    ix.log2_domain = 15; ix.zk_rows = 3;
    if (mina_verify_install_verifier_index(&ix) != MINA_OK) { fprintf(stderr, "install: %s\n", mina_last_error()); return 2; }

    std::atomic<bool> stop{false}; std::atomic<long> wrong{0}, calls{0}, verdicts{0}, errors{0};
    auto state_caller = [&](int id) {
        std::mt19937 rng(1234 + id);
        const size_t sizes[] = {1, 1, 2, 7, 64, 300, 1, 900};
        for (long it = 0; !stop.load(); ++it) {
            const size_t n = sizes[(it + id) % 8];
            std::vector<const uint8_t *> P(n), Q(n); std::vector<size_t> PL(n), QL(n); std::vector<uint8_t> want(n, 1), got(n, 7);
            const bool with_bad = bad_every > 0 && it % bad_every == 0;
            for (size_t i = 0; i < n; ++i) {
                const size_t k = rng() % proofs.size();
                const std::string *p = &proofs[k];
                if (with_bad && i == n / 2) { if (rng() & 1) { p = &bad_proofs[k]; want[i] = 0; } else { p = &truncated; want[i] = 0; } }
                P[i] = (const uint8_t *)p->data(); PL[i] = p->size(); Q[i] = (const uint8_t *)pubs[k].data(); QL[i] = pubs[k].size();
                if (p == &truncated) { Q[i] = (const uint8_t *)pubs[0].data(); QL[i] = pubs[0].size(); }
            }
            int rc;
            if (n == 1 && (it & 1)) { got[0] = mina_verify_state(P[0], PL[0], Q[0], QL[0]) ? 1 : 0; rc = MINA_OK; }
            else rc = mina_verify_state_batch(n, P.data(), PL.data(), Q.data(), QL.data(), got.data());
            if (rc != MINA_OK) { errors.fetch_add(1); fprintf(stderr, "state call failed: %s\n", mina_last_error()); continue; }
            for (size_t i = 0; i < n; ++i) if (got[i] != want[i]) { wrong.fetch_add(1); if (wrong.load() < 5) fprintf(stderr, "caller %d call %ld: proof %zu of %zu: verdict %d, expected %d\n", id, it, i, n, got[i], want[i]); }
            calls.fetch_add(1); verdicts.fetch_add((long)n);
        }
    };
    auto account_caller = [&](int id) {
        std::mt19937 rng(99 + id);
        for (long it = 0; !stop.load(); ++it) {
            const size_t n = 1 + rng() % 40;
            std::vector<const uint8_t *> P(n), Q(n); std::vector<size_t> PL(n), QL(n); std::vector<uint8_t> want(n, 1), got(n, 7); std::vector<std::string> keep;
            keep.reserve(n);
            for (size_t i = 0; i < n; ++i) {
                const size_t k = rng() % aproofs.size();
                P[i] = (const uint8_t *)aproofs[k].data(); PL[i] = aproofs[k].size(); Q[i] = (const uint8_t *)apubs[k].data(); QL[i] = apubs[k].size();
                if (bad_every > 0 && it % bad_every == 0 && i == 0) { keep.push_back(apubs[k]); keep.back()[0] = (char)0xBD; Q[i] = (const uint8_t *)keep.back().data(); want[i] = 0; }
            }
            if (mina_verify_account_batch(n, P.data(), PL.data(), Q.data(), QL.data(), got.data()) != MINA_OK) { errors.fetch_add(1); continue; }
            for (size_t i = 0; i < n; ++i) if (got[i] != want[i]) wrong.fetch_add(1);
            calls.fetch_add(1); verdicts.fetch_add((long)n);
        }
    };
    auto installer = [&]() {                                          // re-installs the index and flips tuning fields while calls are in flight
        for (long it = 0; !stop.load(); ++it) {
            std::this_thread::sleep_for(std::chrono::milliseconds(40));
            (void)mina_verify_install_verifier_index(&ix);
            mina_verify_tuning t; mina_verify_tuning_get(&t);
            t.merge = (it & 1) ? 1 : 0; t.chunk = (it & 2) ? 256 : 8192; t.slots = (it & 4) ? 2 : 4; t.linger_us = (it & 1) ? 200 : 500;
            (void)mina_verify_configure_ex(&t);
        }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < n_state; ++i) th.emplace_back(state_caller, i);
    for (int i = 0; i < n_acct; ++i) th.emplace_back(account_caller, i);
    th.emplace_back(installer);
    std::this_thread::sleep_for(std::chrono::milliseconds((long)(seconds * 1000)));
    stop.store(true);
    for (auto &t : th) t.join();
    mina_verify_shutdown();
    printf("{\"seconds\": %.1f, \"state_callers\": %d, \"account_callers\": %d, \"calls\": %ld, \"verdicts\": %ld, \"device_jobs\": %ld, \"culprit_searches\": %ld, \"account_jobs\": %ld, \"installs\": %ld, "
           "\"wrong_verdicts\": %ld, \"failed_calls\": %ld}\n", seconds, n_state, n_acct, calls.load(), verdicts.load(), g_jobs.load(), g_searches.load(), g_account_jobs.load(), g_installs.load(), wrong.load(), errors.load());
    return (wrong.load() || errors.load() || g_searches.load() == 0 || g_jobs.load() == 0) ? 1 : 0;
}
