// fuzz_parsers.cpp -- the library's untrusted-bytes surface, standalone under sanitizers (the proof / public-input bytes arrive over
// the network: core/src/aligned.rs:31-58 hands them to Aligned's batcher and operators).  Every reader here is the SAME header libminaverify.so
// compiles (mina_bridge_amd/csrc/wire_*.h, loaders_text.h) -- plain C++17, so g++ / clang++ build it without HIP.
//
// Two front ends over one dispatcher (`one_input`: byte 0 picks the reader, the rest is its input):
//   * -DFUZZ_LIBFUZZER : clang++ -fsanitize=fuzzer,address,undefined  -> LLVMFuzzerTestOneInput (coverage-guided, seeded from the golden fixtures)
//   * default          : g++ -fsanitize=address,undefined            -> main(): for every seed file, every reader it is meant for sees the seed itself,
//                        every truncation of it and single-bit flips at every byte (strided for the big ones): the deterministic sweep
// A finding is a sanitizer report (heap / stack overflow, signed overflow, misaligned or out-of-range access ...) or an assertion of the invariants below.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "../../mina_bridge_amd/csrc/loaders_text.h"
#include "../../mina_bridge_amd/csrc/wire_account.h"
#include "../../mina_bridge_amd/csrc/wire_proof.h"
#include "../../mina_bridge_amd/csrc/wire_pub.h"

namespace {
enum Target : uint8_t { T_WRAP_BINCODE, T_WRAP_BINPROT, T_STATE_PROOF, T_PSTATE_BINCODE, T_PSTATE_BINPROT, T_ACCOUNT_BINCODE, T_ACCOUNT_BINPROT, T_ACCOUNT_PROOF,
                        T_STATE_PUB, T_ACCOUNT_PUB, T_POSEIDON_TEXT, T_TOKENS_JSON, T_INDEX_JSON, T_COUNT };

uint64_t g_sink = 0;                     // keeps the parsed values alive
void sink(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; ++i) g_sink = g_sink * 1099511628211ull + b[i]; }

template <class C> void wrap_proof(const uint8_t *d, size_t n) {
    auto w = std::make_unique<mw::WrapProof>();
    C c(d, n);
    if (mw::read_wrap_proof(c, *w)) {
        assert(c.pos <= n);
        sink(&w->prev_evals.n, sizeof w->prev_evals.n);
        for (size_t i = 0; i < w->prev_evals.size(); ++i) { assert(w->prev_evals[i].zeta.size() <= 16); sink(w->prev_evals[i].zeta.data(), w->prev_evals[i].zeta.size() * 32); }
        sink(&w->step_old_bulletproof_challenges.n, sizeof(size_t));
    }
}
template <class C> void protocol_state(const uint8_t *d, size_t n) {
    auto s = std::make_unique<mw::ProtocolState>();
    C c(d, n);
    if (mw::read_protocol_state(c, *s)) {
        assert(c.pos <= n);
        std::vector<mw::B32> f; mw::protocol_state_body_fields(*s, f);        // `to_input` flattening + bit packing
        assert(f.size() < 64);                                                // a record has 64 slots (include/mina_verify.h MINA_PSTATE_SLOTS)
        for (auto &x : f) sink(x.b, 32);
    }
}
template <class C> void account(const uint8_t *d, size_t n) {
    auto a = std::make_unique<mw::Account>();
    C c(d, n);
    if (mw::read_account(c, *a)) {
        assert(c.pos <= n);
        std::vector<uint8_t> abi; mw::abi_encode_account(*a, abi); sink(abi.data(), abi.size());     // sol/account.rs:25-314
        std::vector<mw::B32> f; mw::account_fields(*a, f); for (auto &x : f) sink(x.b, 32);
    }
}
}  // namespace

extern "C" int one_input(const uint8_t *data, size_t size) {
    if (size == 0) return 0;
    const uint8_t t = data[0] % T_COUNT; const uint8_t *d = data + 1; const size_t n = size - 1;
    std::vector<uint8_t> copy(d, d + n);          // exact-size heap copy: reading one byte past the input is an ASan report
    d = copy.data();
    uint8_t empty_byte = 0; if (n == 0) d = &empty_byte;
    const char *why = "";
    switch (t) {
    case T_WRAP_BINCODE: wrap_proof<mw::Bincode>(d, n); break;
    case T_WRAP_BINPROT: wrap_proof<mw::Binprot>(d, n); break;
    case T_STATE_PROOF: { auto box = std::make_unique<mw::StateProofContainer>(); if (mw::read_state_proof(d, n, *box)) sink(&box->states[16].blockchain_length, 4); break; }
    case T_PSTATE_BINCODE: protocol_state<mw::Bincode>(d, n); break;
    case T_PSTATE_BINPROT: protocol_state<mw::Binprot>(d, n); break;
    case T_ACCOUNT_BINCODE: account<mw::Bincode>(d, n); break;
    case T_ACCOUNT_BINPROT: account<mw::Binprot>(d, n); break;
    case T_ACCOUNT_PROOF: {                        // MinaAccountProof: the merkle path, then the bincode account behind it (api_account.hip)
        uint8_t sib[64 * 32], dirs[64]; uint32_t depth = 0; size_t off = 0;
        if (mw::parse_merkle_path(d, n, 64, sib, dirs, &depth, &off, &why) == MINA_OK) { assert(depth <= 64 && off <= n); sink(sib, depth * 32); account<mw::Bincode>(d + off, n - off); }
        break; }
    case T_STATE_PUB: { mina_state_pub_inputs out; if (mw::parse_state_pub_inputs(d, n, &out, &why) == MINA_OK) sink(&out, sizeof out); break; }
    case T_ACCOUNT_PUB: { uint8_t lh[32]; size_t eo = 0, el = 0; if (mw::parse_account_pub_inputs(d, n, lh, &eo, &el, &why) == MINA_OK) { assert(eo + el == n); sink(d + eo, el); } break; }
    case T_POSEIDON_TEXT: { std::vector<uint8_t> out((9 + 165) * 32); std::string err; for (int f = 0; f < 2; ++f) if (mbl::poseidon_params_parse(f, (const char *)d, n, out.data(), err) == MINA_OK) sink(out.data(), out.size()); break; }
    case T_TOKENS_JSON: {
        mbl::JVal root;
        if (mbl::parse_json((const char *)d, n, root))
            for (uint32_t feat : {0u, 0x00ffu, 0xffffffffu}) for (uint32_t present : {0u, 0xffffffffu}) { mbl::TokOut o; if (mbl::tokens_from_json(root, feat & 1, feat, present, o)) sink(o.code.data(), o.code.size()); }
        break; }
    case T_INDEX_JSON: {
        mbl::JVal root;
        if (mbl::parse_json((const char *)d, n, root)) {
            const mbl::PointReader point = [](const std::vector<uint8_t> &b, uint8_t *out64) { if (b.size() != 33 && b.size() != 64) return false; memset(out64, 0, 64); memcpy(out64, b.data(), b.size() < 64 ? 32 : 64); return true; };
            for (int with_points = 0; with_points < 2; ++with_points) { mbl::IndexFields f; std::string err; if (mbl::index_fields_from_json(root, with_points, with_points ? &point : nullptr, f, err)) sink(&f, sizeof f); }
        }
        break; }
    }
    return 0;
}

#ifdef FUZZ_LIBFUZZER
extern "C" int LLVMFuzzerTestOneInput(const uint8_t *data, size_t size) { return one_input(data, size); }
#else
// deterministic sweep: argv = seed files named <target index>_<anything>; the seed, every truncation, one bit flipped at every (strided) byte
static bool slurp(const char *path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path, "rb"); if (!f) return false;
    out.clear(); uint8_t buf[65536]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + k);
    fclose(f); return true;
}
int main(int argc, char **argv) {
    size_t cases = 0;
    for (int a = 1; a < argc; ++a) {
        const char *base = strrchr(argv[a], '/'); base = base ? base + 1 : argv[a];
        const int t = atoi(base);
        std::vector<uint8_t> seed;
        if (t < 0 || t >= T_COUNT || !slurp(argv[a], seed)) { fprintf(stderr, "bad seed %s\n", argv[a]); return 2; }
        std::vector<uint8_t> in(seed.size() + 1); in[0] = (uint8_t)t; if (!seed.empty()) memcpy(in.data() + 1, seed.data(), seed.size());
        one_input(in.data(), in.size()); ++cases;
        const size_t step_trunc = seed.size() > 8192 ? 7 : 1, step_flip = seed.size() > 8192 ? 13 : 1;     // the 40 KB containers: every 7th length, every 13th byte
        for (size_t len = 0; len < seed.size(); len += step_trunc) { one_input(in.data(), 1 + len); ++cases; }
        for (size_t i = 0; i < seed.size(); i += step_flip) { const uint8_t keep = in[1 + i]; in[1 + i] = keep ^ (uint8_t)(1u << (i % 8)); one_input(in.data(), in.size()); in[1 + i] = keep ^ 0xff; one_input(in.data(), in.size()); in[1 + i] = keep; cases += 2; }
        // length fields: the first 16 bytes set to 0xff one at a time (huge counts must be bounded before anything is allocated or indexed)
        for (size_t i = 0; i < seed.size() && i < 64; ++i) { const uint8_t keep = in[1 + i]; in[1 + i] = 0xff; one_input(in.data(), in.size()); in[1 + i] = keep; ++cases; }
    }
    printf("sweep ok: %zu cases over %d seeds (sink %llx)\n", cases, argc - 1, (unsigned long long)g_sink);
    return 0;
}
#endif
