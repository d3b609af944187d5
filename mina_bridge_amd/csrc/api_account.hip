// api_account.hip -- Proof-of-Account (8f-3; BASELINE config C4): `verify_account_inclusion_ffi`
// (README.md:343-362) for the bytes core/src/aligned.rs:46-58 produces.
//
// Per proof:   FORMAT   host  bincode MinaAccountProof (merkle path + account), MinaAccountPubInputs (ledger hash + ABI bytes)
//              ABI      host  re-derive `encoded_account` from `account` (sol/account.rs:25-314) and compare bytes   (README.md:349-352)
//              HASH     GPU   account hash = H_"MinaAccount"(to_input) with the zkapp-uri, verification-key and zkapp sub-hashes
//              MERKLE   GPU   fold the hash along the path with H_"MinaMklTree%03d", compare with the ledger hash    (README.md:345-347)
// The four Poseidon stages of a batch run as three salted-hash launches + the Merkle fold, all on one lane without host
// synchronisation in between; the hash of one stage is patched into its slot of the next stage's input on the device.
#include <mutex>

#include "ctx.h"
#include <chrono>
#include <new>
#include "sponge.cuh"
#include "wire_account.h"

int mb_merkle_fold_dev(mina_ctx *c, int field, size_t n, uint32_t depth, const uint32_t *d_leaves, const uint32_t *d_sib, const uint8_t *d_dirs, uint32_t *d_roots);
int mb_merkle_prepare_salts(mina_ctx *c, int field, uint32_t depth);
int mb_ensure_state_salts(mina_ctx *c);

namespace mb {

// n sponges started from the salted state `salts[salt_idx[i]]`, absorbing records[i][0 .. nf[i]) (canonical words), squeezed once.
// patch_a / patch_b (optional): hashes of an earlier stage that replace slot `slot_a` / `slot_b` of every record first.
template <int F, int LANES>
__global__ void __launch_bounds__(256)
salted_hash_kernel(uint32_t n, FieldK fk, const PoseidonParams *__restrict__ pp, const fe_t *__restrict__ salts, const uint32_t *__restrict__ salt_idx,
                   const uint32_t *__restrict__ records /* n * MINA_PSTATE_SLOTS * 8 */, const uint32_t *__restrict__ nfields,
                   const uint32_t *__restrict__ patch_a, uint32_t slot_a, const uint32_t *__restrict__ patch_b, uint32_t slot_b, uint32_t *__restrict__ out /* n*8 */) {
    // a Proof-of-Account job is a short dependent chain (~70 permutations); beside the state-proof pipeline its waves share SIMDs with chip-filling
    // hash kernels: issue priority keeps the chain at its own pace, the others take the cycles it leaves
    __builtin_amdgcn_s_setprio(3);
    bool writer;
    const uint32_t sp = coop_sponge_index<LANES>(writer), e = coop_elem<LANES>();
    const bool live = sp < n;
    const uint32_t idx = live ? sp : 0;
    const uint32_t *rec = records + (size_t)idx * MINA_PSTATE_SLOTS * 8;
    uint32_t nf = nfields[idx]; if (nf > MINA_PSTATE_SLOTS) nf = MINA_PSTATE_SLOTS;
    fe_t s = salts[salt_idx[idx] * 3 + e];
    uint32_t count = 0;
    for (uint32_t el = 0; el < nf; ++el) {
        if (count == 2) { poseidon_permute_coop<F, LANES>(s, pp); count = 0; }
        if (e == count) {
            const uint32_t *src = (patch_a && el == slot_a) ? patch_a + (size_t)idx * 8 : ((patch_b && el == slot_b) ? patch_b + (size_t)idx * 8 : rec + (size_t)el * 8);
            s = fe_add<F>(s, fe_to_mont<F>(load_fe<F>(src), fk.r2));
        }
        ++count;
    }
    poseidon_permute_coop<F, LANES>(s, pp);
    s = coop_get<LANES>(s, 0);
    if (live && writer) { const fe_t w = fe_from_mont<F>(s); for (int i = 0; i < 8; ++i) out[(size_t)sp * 8 + i] = w.v[i]; }
}

}  // namespace mb

static int salted_hash_dev(mina_ctx *c, size_t n, const uint32_t *salt_idx, const uint32_t *recs, const uint32_t *nf, const uint32_t *pa, uint32_t sa,
                           const uint32_t *pb, uint32_t sb, uint32_t *out) {
    const PoseidonParams *pp = c->pparams[FIELD_FP].as<PoseidonParams>();
    const fe_t *salts = c->state_salts.as<fe_t>();
    ProfScope ps_(c, PS_STATE_HASH);
    if (use_coop16(c, n))
        mb::salted_hash_kernel<FIELD_FP, 16><<<cdiv(n * 16, 256), 256, 0, c->L->stream>>>((uint32_t)n, c->fk[FIELD_FP], pp, salts, salt_idx, recs, nf, pa, sa, pb, sb, out);
    else if (use_coop8(c, n))
        mb::salted_hash_kernel<FIELD_FP, 8><<<cdiv(n * 8, 256), 256, 0, c->L->stream>>>((uint32_t)n, c->fk[FIELD_FP], pp, salts, salt_idx, recs, nf, pa, sa, pb, sb, out);
    else
        mb::salted_hash_kernel<FIELD_FP, 3><<<cdiv(coop_threads<3>(n), 256), 256, 0, c->L->stream>>>((uint32_t)n, c->fk[FIELD_FP], pp, salts, salt_idx, recs, nf, pa, sa, pb, sb, out);
    HIPC(hipGetLastError());
    return MINA_OK;
}

namespace {
struct ParsedAccount { bool ok = false, abi_ok = false; mw::Account acc; uint32_t depth = 0; uint8_t sib[64 * 32]; uint8_t dirs[64]; uint8_t ledger[32]; };

void parse_account(const uint8_t *proof, size_t proof_len, const uint8_t *pub, size_t pub_len, ParsedAccount &pa) {
    if (!proof || !pub) return;
    size_t acc_off = 0, eo = 0, el = 0;
    if (mina_parse_merkle_path(proof, proof_len, 64, pa.sib, pa.dirs, &pa.depth, &acc_off) != MINA_OK) return;
    if (mina_parse_account_pub_inputs(pub, pub_len, pa.ledger, &eo, &el) != MINA_OK) return;
    mw::Bincode c(proof + acc_off, proof_len - acc_off);
    if (!mw::read_account(c, pa.acc) || c.pos != proof_len - acc_off) return;
    pa.ok = true;
    std::vector<uint8_t> enc;
    mw::abi_encode_account(pa.acc, enc);
    pa.abi_ok = enc.size() == el && memcmp(enc.data(), pub + eo, el) == 0;
}

}  // namespace

// account hashes (and optionally Merkle roots along per-account paths of one common depth) of n parsed accounts, on the GPU
// `lane` / `enq_mu` (the boundary): the job runs on a lane of its own, and `enq_mu` -- the lock of everything that touches the context's lane cursor --
// is held only while kernels are QUEUED, not while the host flattens the inputs or waits for the GPU (an account job is a latency-bound chain of
// ~8 ms: held throughout, state-proof callers of the same process could not queue their jobs meanwhile -- and their rate fell by a third)
static int account_hashes(mina_ctx *c, const std::vector<const mw::Account *> &accs, uint8_t *hashes_out, uint32_t depth, const uint8_t *sib, const uint8_t *dirs, uint8_t *roots_out,
                          Lane *lane = nullptr, std::mutex *enq_mu = nullptr) {
    const size_t n = accs.size();
    if (n == 0) return MINA_OK;
    int rc;
    struct Enq {                                                 // the lane cursor is set inside the lock and put back before it is released
        mina_ctx *c; Lane *lane; std::mutex *mu; bool held = false;
        void lock() { if (mu) mu->lock(); held = true; if (lane) c->L = lane; else c->use_lane0(); }
        void unlock() { if (held) { c->use_lane0(); if (mu) mu->unlock(); held = false; } }
        ~Enq() { unlock(); }
    } enq{c, lane, enq_mu};
    enq.lock();
    if (lane && !lane->stream) HIPC(hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking));
    if ((rc = mb_ensure_state_salts(c))) return rc;
    if (lane) c->L = lane;
    if (depth && (rc = mb_merkle_prepare_salts(c, FIELD_FP, depth))) return rc;
    if (lane) c->L = lane;
    Lane &L = *c->L;
    enq.unlock();
    // stage records straight into the pinned upload blob: [0, n) zkapp-uri, [n, 2n) verification key, [2n, 3n) zkapp, [3n, 4n) account
    const size_t rec_bytes = 4 * n * MINA_PSTATE_SLOTS * 32;
    const size_t o_nf = rec_bytes, o_salt = o_nf + 4 * n * 4, o_sib = o_salt + 4 * n * 4, o_dir = o_sib + n * depth * 32,
                 o_h = (o_dir + n * depth + 255) & ~(size_t)255, total = o_h + 5 * n * 32;      // hashes: uri, vk, zkapp, account, roots
    if ((rc = L.host_stage.ensure(o_h))) return rc;
    uint8_t *blob = (uint8_t *)L.host_stage.p;
    uint32_t *nf = (uint32_t *)(blob + o_nf), *salt = (uint32_t *)(blob + o_salt);
    static const mw::ZkappAccount DEFAULT_ZKAPP;
    auto put = [&](size_t slot, const std::vector<mw::B32> &f, uint32_t salt_id) {
        const size_t k = f.size() < MINA_PSTATE_SLOTS ? f.size() : MINA_PSTATE_SLOTS;
        uint8_t *r = blob + slot * MINA_PSTATE_SLOTS * 32;
        for (size_t j = 0; j < k; ++j) memcpy(r + j * 32, f[j].b, 32);
        memset(r + k * 32, 0, (MINA_PSTATE_SLOTS - k) * 32);
        nf[slot] = (uint32_t)k; salt[slot] = salt_id;
    };
    mb_parallel_for(n, [&](size_t i) {                          // to_input flattening of the four hashes' inputs: independent per account
        const mw::Account &a = *accs[i];
        const mw::ZkappAccount &z = a.has_zkapp ? a.zkapp : DEFAULT_ZKAPP;
        std::vector<mw::B32> f;
        mw::zkapp_uri_fields(z.zkapp_uri, f); put(i, f, MB_SALT_ZKAPP_URI);
        mw::vk_fields(z.has_vk ? z.vk : mw::dummy_vk(), f); put(n + i, f, MB_SALT_SIDE_LOADED_VK);
        mw::zkapp_fields(z, f); put(2 * n + i, f, MB_SALT_ZKAPP_ACCOUNT);
        mw::account_fields(a, f); put(3 * n + i, f, MB_SALT_ACCOUNT);
    });
    if (depth) { memcpy(blob + o_sib, sib, n * depth * 32); memcpy(blob + o_dir, dirs, n * depth); }
    static const bool timing = getenv("MINA_VERIFY_TIMING") != nullptr;
    const auto t_flat = std::chrono::steady_clock::now();
    if ((rc = L.st_in.ensure(total))) return rc;
    uint8_t *d = L.st_in.as<uint8_t>();
    enq.lock();
    HIPC(hipMemcpyAsync(d, blob, o_h, hipMemcpyHostToDevice, L.stream));
    auto R = [&](size_t stage) { return (const uint32_t *)(d + stage * n * MINA_PSTATE_SLOTS * 32); };
    auto NF = [&](size_t stage) { return (const uint32_t *)(d + o_nf) + stage * n; };
    auto SL = [&](size_t stage) { return (const uint32_t *)(d + o_salt) + stage * n; };
    auto H = [&](size_t stage) { return (uint32_t *)(d + o_h) + stage * n * 8; };
    if ((rc = salted_hash_dev(c, 2 * n, SL(0), R(0), NF(0), nullptr, 0, nullptr, 0, H(0)))) return rc;                       // uri + vk in one launch
    if ((rc = salted_hash_dev(c, n, SL(2), R(2), NF(2), H(0), mw::ZK_SLOT_URI, H(1), mw::ZK_SLOT_VK, H(2)))) return rc;       // zkapp
    if ((rc = salted_hash_dev(c, n, SL(3), R(3), NF(3), H(2), 0, nullptr, 0, H(3)))) return rc;                               // account
    if (roots_out) {
        if (depth) { if ((rc = mb_merkle_fold_dev(c, FIELD_FP, n, depth, H(3), (const uint32_t *)(d + o_sib), d + o_dir, H(4)))) return rc; }
        else HIPC(hipMemcpyAsync(H(4), H(3), n * 32, hipMemcpyDeviceToDevice, L.stream));
        HIPC(hipMemcpyAsync(roots_out, H(4), n * 32, hipMemcpyDeviceToHost, L.stream));
    }
    if (hashes_out) HIPC(hipMemcpyAsync(hashes_out, H(3), n * 32, hipMemcpyDeviceToHost, L.stream));
    enq.unlock();
    HIPC(hipStreamSynchronize(L.stream));
    if (timing) fprintf(stderr, "mina_verify:   %zu accounts: upload + GPU + download %.2f ms (%.1f MB up)\n", n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_flat).count(), o_h / 1e6);
    return MINA_OK;
}

// serialized accounts -> account hashes (parity hook; encoding MINA_ENC_BINPROT / MINA_ENC_BINCODE)
extern "C" int mina_account_hash_batch(mina_ctx *c, int encoding, size_t n, const uint8_t *const *accounts, const size_t *lens, uint8_t *hashes_out) {
    if (!c || (n && (!accounts || !lens || !hashes_out))) return fail(MINA_ERR_ARG, "null argument");
    if (!c->have_pparams[FIELD_FP]) return fail(MINA_ERR_STATE, "Poseidon constants not installed for Fp");
    std::vector<mw::Account> accs(n); std::vector<const mw::Account *> ptr(n);
    for (size_t i = 0; i < n; ++i) {
        bool ok;
        if (!accounts[i]) return fail(MINA_ERR_ARG, "null account");
        if (encoding == MINA_ENC_BINPROT) { mw::Binprot r(accounts[i], lens[i]); ok = mw::read_account(r, accs[i]) && r.pos == lens[i]; }
        else if (encoding == MINA_ENC_BINCODE) { mw::Bincode r(accounts[i], lens[i]); ok = mw::read_account(r, accs[i]) && r.pos == lens[i]; }
        else return fail(MINA_ERR_ARG, "bad encoding");
        if (!ok) return fail(MINA_ERR_FORMAT, "malformed account");
        ptr[i] = &accs[i];
    }
    HIPC(hipSetDevice(c->device));
    return account_hashes(c, ptr, hashes_out, 0, nullptr, nullptr, nullptr);
}

// serialized account -> the ABI bytes the reference's `Account::abi_encode()` produces (host only)
extern "C" int mina_account_abi_encode(const uint8_t *account, size_t len, int encoding, uint8_t *out, size_t cap, size_t *out_len) {
    if (!account || !out_len) return fail(MINA_ERR_ARG, "null argument");
    mw::Account a; bool ok;
    if (encoding == MINA_ENC_BINPROT) { mw::Binprot r(account, len); ok = mw::read_account(r, a) && r.pos == len; }
    else if (encoding == MINA_ENC_BINCODE) { mw::Bincode r(account, len); ok = mw::read_account(r, a) && r.pos == len; }
    else return fail(MINA_ERR_ARG, "bad encoding");
    if (!ok) return fail(MINA_ERR_FORMAT, "malformed account");
    std::vector<uint8_t> enc; mw::abi_encode_account(a, enc);
    *out_len = enc.size();
    if (out) { if (cap < enc.size()) return fail(MINA_ERR_ARG, "output buffer too small"); memcpy(out, enc.data(), enc.size()); }
    return MINA_OK;
}

// Proof-of-Account for n (proof, pub) pairs on context c: passed / ran masks per proof
extern "C" int mina_verify_account_ctx(mina_ctx *c, size_t n, const uint8_t *const *proofs, const size_t *proof_lens, const uint8_t *const *pubs, const size_t *pub_lens,
                                       uint32_t *passed, uint32_t *ran) {
    return mb_verify_account_on(c, n, proofs, proof_lens, pubs, pub_lens, passed, ran, nullptr, nullptr);
}
int mb_verify_account_on(mina_ctx *c, size_t n, const uint8_t *const *proofs, const size_t *proof_lens, const uint8_t *const *pubs, const size_t *pub_lens,
                         uint32_t *passed, uint32_t *ran, Lane *lane, std::mutex *enq_mu) {
    if (!c || (n && (!proofs || !proof_lens || !pubs || !pub_lens || !passed || !ran))) return fail(MINA_ERR_ARG, "null argument");
    if (!c->have_pparams[FIELD_FP]) return fail(MINA_ERR_STATE, "Poseidon constants not installed for Fp");
    HIPC(hipSetDevice(c->device));
    static const bool timing = getenv("MINA_VERIFY_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    std::unique_ptr<ParsedAccount[]> pa(new ParsedAccount[n]);       // (constructing / destroying the entries on the pool's threads instead: 49 -> 76 ms for 16 384 -- the allocator's arenas)
    const double t_alloc = ms();
    mb_parallel_for(n, [&](size_t i) { parse_account(proofs[i], proof_lens[i], pubs[i], pub_lens[i], pa[i]); });
    const double t_parse = ms();
    for (size_t i = 0; i < n; ++i) {
        ran[i] = MINA_CHECK_FORMAT; passed[i] = 0;
        if (!pa[i].ok) continue;
        passed[i] |= MINA_CHECK_FORMAT; ran[i] |= MINA_CHECK_ACCOUNT_ABI | MINA_CHECK_MERKLE;
        if (pa[i].abi_ok) passed[i] |= MINA_CHECK_ACCOUNT_ABI;
    }
    for (uint32_t d = 0; d <= 64; ++d) {                           // paths of one depth go to the GPU together (one group in practice)
        std::vector<size_t> idx;
        for (size_t i = 0; i < n; ++i) if (pa[i].ok && pa[i].depth == d) idx.push_back(i);
        if (idx.empty()) continue;
        const size_t m = idx.size();
        std::vector<const mw::Account *> accs(m); std::vector<uint8_t> sib(m * d * 32 + 1), dirs(m * d + 1), roots(m * 32);
        for (size_t j = 0; j < m; ++j) { accs[j] = &pa[idx[j]].acc; if (d) { memcpy(&sib[j * d * 32], pa[idx[j]].sib, (size_t)d * 32); memcpy(&dirs[j * d], pa[idx[j]].dirs, d); } }
        int rc = account_hashes(c, accs, nullptr, d, sib.data(), dirs.data(), roots.data(), lane, enq_mu);
        if (rc) return rc;
        for (size_t j = 0; j < m; ++j) if (memcmp(&roots[j * 32], pa[idx[j]].ledger, 32) == 0) passed[idx[j]] |= MINA_CHECK_MERKLE;
    }
    if (timing) fprintf(stderr, "mina_verify: %zu account proofs: containers allocated at %.2f ms, parsed at %.2f, hashed + folded at %.2f\n", n, t_alloc, t_parse, ms());
    return MINA_OK;
}
