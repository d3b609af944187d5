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
#include "wire_state.h"

namespace {

// ---------------------------------------------------------------------------------------------- 256-bit literals
struct U256 { uint64_t w[4] = {0, 0, 0, 0}; };
bool mul_small_add(U256 &a, uint32_t m, uint32_t add) {
    unsigned __int128 c = add;
    for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a.w[i] * m; a.w[i] = (uint64_t)c; c >>= 64; }
    return c == 0;
}
void to_le32(const U256 &a, uint8_t *o) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) o[8 * i + j] = (uint8_t)(a.w[i] >> (8 * j)); }

// next numeric literal at or after `pos`: decimal digits or 0x + hex digits (big-endian text), optionally quoted; false = none left / overflow
bool next_number(const char *s, size_t n, size_t &pos, U256 &out) {
    while (pos < n) {
        const unsigned char c = (unsigned char)s[pos];
        if (isdigit(c) && (pos == 0 || !(isalnum((unsigned char)s[pos - 1]) || s[pos - 1] == '_'))) break;     // not the tail of an identifier (vec3, u64, ...)
        ++pos;
    }
    if (pos >= n) return false;
    out = U256{};
    if (s[pos] == '0' && pos + 1 < n && (s[pos + 1] == 'x' || s[pos + 1] == 'X')) {
        pos += 2; size_t digits = 0;
        while (pos < n && isxdigit((unsigned char)s[pos])) {
            const char c = s[pos++]; const uint32_t d = isdigit((unsigned char)c) ? (uint32_t)(c - '0') : (uint32_t)(tolower(c) - 'a' + 10);
            if (!mul_small_add(out, 16, d)) return false;
            ++digits;
        }
        return digits > 0;
    }
    while (pos < n && (isdigit((unsigned char)s[pos]) || s[pos] == '_')) { if (s[pos] != '_' && !mul_small_add(out, 10, (uint32_t)(s[pos] - '0'))) return false; ++pos; }
    return true;
}
// position just behind the first occurrence of any of the key spellings (as a whole word), or npos
size_t find_key(const char *s, size_t n, std::initializer_list<const char *> keys, size_t from = 0) {
    size_t best = std::string::npos;
    for (const char *k : keys) {
        const size_t kl = strlen(k);
        for (size_t i = from; i + kl <= n; ++i) {
            if (memcmp(s + i, k, kl)) continue;
            const bool left = i == 0 || !(isalnum((unsigned char)s[i - 1]) || s[i - 1] == '_'), right = i + kl == n || !(isalnum((unsigned char)s[i + kl]) || s[i + kl] == '_');
            if (left && right) { if (i + kl < best) best = i + kl; break; }
        }
    }
    return best;
}

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
// text -> the (9 + 165) x 32-byte layout of mina_poseidon_set_params (mds row-major, then rc[round][element])
extern "C" int mina_poseidon_params_parse(int field, const char *text, size_t len, uint8_t *params_out) {
    if (!text || !params_out) return fail(MINA_ERR_ARG, "null argument");
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    const size_t at_mds = find_key(text, len, {"mds", "MDS"}), at_rc = find_key(text, len, {"round_constants", "roundConstants", "ROUND_CONSTANTS", "rc"});
    if (at_mds == std::string::npos || at_rc == std::string::npos) return fail(MINA_ERR_FORMAT, "no `mds` / `round_constants` key in the Poseidon table text");
    auto take = [&](size_t pos, size_t count, uint8_t *dst, size_t stop) -> bool {
        for (size_t i = 0; i < count; ++i) {
            U256 v;
            if (!next_number(text, len, pos, v) || pos > stop) return false;
            to_le32(v, dst + 32 * i);
            if (!(field == FIELD_FP ? mw::fp_canonical(dst + 32 * i) : mw::fq_canonical(dst + 32 * i))) return false;
        }
        return true;
    };
    // each table ends where the other begins (whichever comes second runs to the end of the text)
    const size_t mds_stop = at_mds < at_rc ? at_rc : len, rc_stop = at_rc < at_mds ? at_mds : len;
    if (!take(at_mds, 9, params_out, mds_stop)) return fail(MINA_ERR_FORMAT, "the MDS matrix needs 9 canonical field elements");
    if (!take(at_rc, 165, params_out + 9 * 32, rc_stop)) return fail(MINA_ERR_FORMAT, "the round constants need 55 x 3 canonical field elements");
    { size_t pos = at_rc; U256 v; size_t cnt = 0; while (next_number(text, len, pos, v) && pos <= rc_stop) ++cnt;      // a table of another shape (e.g. the 100-round legacy one) is refused
      // trailing scalars of the o1js object (fullRounds: 55, stateSize: 3, ...) may follow the table when it comes last: allow up to 8
      if (cnt > 165 + 8) return fail(MINA_ERR_FORMAT, "more than 55 x 3 round constants: not the Kimchi parameter shape"); }
    return MINA_OK;
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
