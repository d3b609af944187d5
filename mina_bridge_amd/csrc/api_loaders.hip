// api_loaders.hip -- file-drop loaders for the data the reference tree does not hold (README.md:306-310):
//   * the Poseidon tables of mina-poseidon (`fp_kimchi` / `fq_kimchi`: 3 x 3 MDS + 55 x 3 round constants per field), in the text forms
//     upstream publishes them: the JSON object of o1js' `constants.ts` ({"mds": [[dec, ...], ...], "roundConstants": [[dec, ...], ...], ...})
//     or the Rust table (`mds: vec![vec![...]]`, `round_constants: vec![...]`), decimal or 0x-hex literals, quoted or not;
//   * a kimchi `VerifierIndex` as `serde_json` writes it (o1-utils `SerdeAs`: ark `CanonicalSerialize` bytes as hex strings; points are
//     33-byte compressed: x little-endian + flag byte) and its linearization's constant term as `serde_json` writes `Vec<PolishToken>`
//     -> the library's `mina_verifier_index` + PolishToken byte-code; the same token reader serves the step index.
// [UPSTREAM-RECALL] for every key name and enum spelling (pins core/Cargo.toml:14-18; nothing in the tree holds such a file): the readers
// are tolerant (unknown keys skipped, alternative spellings accepted) and are pinned only by round trips against independent Python
// writers of the same layouts (tests/test_loaders.py).  Host-side parsing; only the install calls touch the GPU.
#include <cctype>
#include <map>

#include "ctx.h"
#include "loaders_text.h"

using mbl::JVal; using mbl::IndexFields; using mbl::TokOut; using mbl::parse_json; using mbl::tokens_from_json; using mbl::index_fields_from_json;

namespace {

bool read_file_text(const char *path, std::string &out, size_t cap) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char buf[65536]; size_t k;
    out.clear();
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) { out.append(buf, k); if (out.size() > cap) { fclose(f); return false; } }
    fclose(f);
    return true;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ Poseidon tables
// text -> the (9 + 165) x 32-byte layout of mina_poseidon_set_params (mds row-major, then rc[round][element]); the reader: loaders_text.h
extern "C" int mina_poseidon_params_parse(int field, const char *text, size_t len, uint8_t *params_out) {
    std::string err;
    const int rc = mbl::poseidon_params_parse(field, text, len, params_out, err);
    return rc ? fail(rc, err) : MINA_OK;
}
extern "C" int mina_poseidon_load_params(mina_ctx *c, int field, const char *text, size_t len) {
    if (!c) return fail(MINA_ERR_ARG, "null argument");
    std::vector<uint8_t> p((9 + 165) * 32);
    int rc = mina_poseidon_params_parse(field, text, len, p.data());
    if (rc) return rc;
    return mina_poseidon_set_params(c, field, p.data());
}
extern "C" int mina_poseidon_load_params_file(mina_ctx *c, int field, const char *path) {
    std::string t;
    if (!path || !read_file_text(path, t, 4u << 20)) return fail(MINA_ERR_ARG, "cannot read the Poseidon table file");
    return mina_poseidon_load_params(c, field, t.data(), t.size());
}
// used by the process-wide contexts: $MINA_POSEIDON_PARAMS_FP / $MINA_POSEIDON_PARAMS_FQ name table files to load instead of the compiled-in surrogate
int mb_poseidon_env_params(mina_ctx *c) {
    static const char *names[2] = {"MINA_POSEIDON_PARAMS_FP", "MINA_POSEIDON_PARAMS_FQ"};
    for (int f = 0; f < 2; ++f) if (const char *p = getenv(names[f])) { int rc = mina_poseidon_load_params_file(c, f, p); if (rc) return rc; }
    return MINA_OK;
}

// PolishToken JSON -> byte-code: loaders_text.h (a small JSON reader + the token table)
extern "C" int mina_polish_tokens_from_json(int field, const char *json, size_t len, uint32_t enabled_features, uint32_t optional_present, uint8_t *out, size_t cap, size_t *out_len) {
    std::string err;
    const int rc = mbl::polish_tokens_from_json(field, json, len, enabled_features, optional_present, out, cap, out_len, err);
    return rc ? fail(rc, err) : MINA_OK;
}

// ------------------------------------------------------------------------------------------------ VerifierIndex JSON
namespace {
// 33-byte ark-serialize 0.3 compressed short-Weierstrass point (the SRS files' codec): x little-endian, then the flag byte --
// 0x40 = infinity, 0x80 = y is the larger of {y, p - y}.  Pallas points (base field Fp): y^2 = x^3 + 5.
bool decompress_pallas(const std::vector<uint8_t> &b, const FieldK &k, uint8_t *out64) {
    if (b.size() == 64) { memcpy(out64, b.data(), 64); return mw::fp_canonical(out64) && mw::fp_canonical(out64 + 32); }      // uncompressed x || y
    if (b.size() != 33) return false;
    if (b[32] & 0x40) { memset(out64, 0, 64); return true; }
    if (b[32] & 0x3f || !mw::fp_canonical(b.data())) return false;
    fe_t x; memcpy(x.v, b.data(), 32); x = fe_to_mont<FIELD_FP>(x, k.r2);
    const fe_t rhs = fe_add<FIELD_FP>(fe_mul<FIELD_FP>(fe_sqr<FIELD_FP>(x), x), k.five);
    fe_t y;
    if (!fe_sqrt<FIELD_FP>(y, rhs, k)) return false;
    fe_t yc = fe_from_mont<FIELD_FP>(y), ny = fe_from_mont<FIELD_FP>(fe_neg<FIELD_FP>(y));
    bool y_larger = false; for (int i = 7; i >= 0; --i) if (yc.v[i] != ny.v[i]) { y_larger = yc.v[i] > ny.v[i]; break; }
    const bool want_larger = (b[32] & 0x80) != 0;
    const fe_t &pick = (y_larger == want_larger) ? yc : ny;
    memcpy(out64, b.data(), 32); memcpy(out64 + 32, pick.v, 32);
    return true;
}
}  // namespace

// kimchi `VerifierIndex<Pallas>` (the Pickles wrap index) as serde_json writes it + its linearization's constant term as serde_json writes
// `Vec<PolishToken>` (`serde_json::to_string(&index.linearization.constant_term)`: the linearization itself is `#[serde(skip)]` in the index)
extern "C" int mina_verifier_index_load_json(mina_ctx *c, const char *index_json, size_t index_len, const char *constant_term_json, size_t ct_len, uint32_t perm_alpha_offset) {
    if (!c || !index_json || !constant_term_json) return fail(MINA_ERR_ARG, "null argument");
    JVal root, prog;
    if (!parse_json(index_json, index_len, root)) return fail(MINA_ERR_FORMAT, "the verifier index is not valid JSON");
    if (!parse_json(constant_term_json, ct_len, prog)) return fail(MINA_ERR_FORMAT, "the constant term is not valid JSON");
    IndexFields f; std::string err;
    const FieldK &kb = c->fk[FIELD_FP];
    const mbl::PointReader point = [&kb](const std::vector<uint8_t> &b, uint8_t *out64) { return decompress_pallas(b, kb, out64); };
    if (!index_fields_from_json(root, FIELD_FQ, &point, f, err)) return fail(MINA_ERR_FORMAT, err);
    TokOut t;
    if (!tokens_from_json(prog, FIELD_FQ, 0, 0, t)) return fail(MINA_ERR_FORMAT, t.err);
    mina_verifier_index vi{};
    vi.log2_domain = f.log2_domain; vi.zk_rows = f.zk_rows; vi.perm_alpha_offset = perm_alpha_offset;
    vi.shifts = f.shifts; vi.sigma_comm = f.sigma; vi.coefficients_comm = f.coeff; vi.selector_comm = f.sel; vi.constant_term = t.code.data(); vi.constant_term_len = t.code.size();
    return mina_verifier_index_install(c, &vi);
}
// the step side: one `VerifierIndex<Vesta>` JSON per step domain in use (domain + shifts + zk_rows are read, commitments are not needed) and
// the step linearization's constant term; `optional_present`: bit i = the step proofs carry optional evaluation i (wire order)
extern "C" int mina_step_index_load_json(mina_ctx *c, size_t n_indexes, const char *const *index_jsons, const size_t *index_lens, const char *constant_term_json, size_t ct_len,
                                         uint32_t enabled_features, uint32_t optional_present) {
    if (!c || !index_jsons || !index_lens || !constant_term_json || n_indexes == 0 || n_indexes > 8) return fail(MINA_ERR_ARG, "bad argument");
    std::vector<uint32_t> domains; std::vector<uint8_t> shifts; uint32_t zk = 3;
    for (size_t i = 0; i < n_indexes; ++i) {
        JVal root; IndexFields f; std::string err;
        if (!index_jsons[i] || !parse_json(index_jsons[i], index_lens[i], root)) return fail(MINA_ERR_FORMAT, "a step verifier index is not valid JSON");
        if (!index_fields_from_json(root, FIELD_FP, nullptr, f, err)) return fail(MINA_ERR_FORMAT, err);
        if (i && f.zk_rows != zk) return fail(MINA_ERR_FORMAT, "the step indexes disagree on zk_rows");
        zk = f.zk_rows; domains.push_back(f.log2_domain); shifts.insert(shifts.end(), f.shifts, f.shifts + 7 * 32);
    }
    JVal prog; TokOut t;
    if (!parse_json(constant_term_json, ct_len, prog)) return fail(MINA_ERR_FORMAT, "the constant term is not valid JSON");
    if (!tokens_from_json(prog, FIELD_FP, enabled_features, optional_present, t)) return fail(MINA_ERR_FORMAT, t.err);
    mina_step_index si{};
    si.zk_rows = zk; si.n_domains = (uint32_t)domains.size(); si.domain_log2 = domains.data(); si.shifts = shifts.data(); si.constant_term = t.code.data(); si.constant_term_len = t.code.size();
    return mina_step_index_install(c, &si);
}
