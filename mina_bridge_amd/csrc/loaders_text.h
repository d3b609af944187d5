// loaders_text.h -- the host-only text / JSON readers of api_loaders.hip: plain C++17, no HIP, no context.  Split out so that the code which parses
// operator-supplied files (Poseidon tables, serde_json verifier indexes, PolishToken programs) builds standalone under g++ -fsanitize=address,undefined
// and under the fuzz loop of tests/fuzz/ (the same header is what libminaverify.so compiles).  Key names and enum spellings are
// [UPSTREAM-RECALL] (pins core/Cargo.toml:14-18): the readers are tolerant and pinned by round trips against independent writers (tests/test_loaders.py).
#pragma once
#include <cctype>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <string>
#include <vector>

#include "../../include/mina_verify.h"
#include "wire_state.h"

namespace mbl {


// ---------------------------------------------------------------------------------------------- 256-bit literals
struct U256 { uint64_t w[4] = {0, 0, 0, 0}; };
inline bool mul_small_add(U256 &a, uint32_t m, uint32_t add) {
    unsigned __int128 c = add;
    for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a.w[i] * m; a.w[i] = (uint64_t)c; c >>= 64; }
    return c == 0;
}
inline void to_le32(const U256 &a, uint8_t *o) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) o[8 * i + j] = (uint8_t)(a.w[i] >> (8 * j)); }

// next numeric literal at or after `pos`: decimal digits or 0x + hex digits (big-endian text), optionally quoted -- or a QUOTED string of exactly 64 hex
// digits without 0x: the byte-hex form of o1-utils `FieldHelpers::from_hex` / serde (`Fp::from_hex("...")` in Rust tables), i.e. the 32 bytes of the
// element in LITTLE-endian order (the same rule as json_field below: 64 quoted hex digits are never a decimal).  false = none left / overflow
inline bool next_number(const char *s, size_t n, size_t &pos, U256 &out) {
    while (pos < n) {
        const unsigned char c = (unsigned char)s[pos];
        if (c == '"' && n - pos >= 66 && s[pos + 65] == '"') {
            bool hex = true; for (size_t i = 1; i <= 64 && hex; ++i) hex = isxdigit((unsigned char)s[pos + i]) != 0;
            // 64 quoted characters that are ALL decimal digits are both a valid little-endian byte-hex element and a valid 64-digit decimal literal.  The
            // spelling in front decides: `from_hex(` = byte-hex, `from_str(` = decimal (falls through to the decimal reader below); with neither the literal is
            // AMBIGUOUS and the table is refused (false) rather than loaded as a constant that may be the wrong one (a random element is all-decimal with
            // probability (10/16)^64 ~ 1e-13: no real table is refused)
            bool all_decimal = hex;
            for (size_t i = 1; i <= 64 && all_decimal; ++i) all_decimal = isdigit((unsigned char)s[pos + i]) != 0;
            if (hex && all_decimal) {
                size_t q = pos; while (q > 0 && isspace((unsigned char)s[q - 1])) --q;
                auto ends_with = [&](const char *w) { const size_t wl = strlen(w); return q >= wl && memcmp(s + q - wl, w, wl) == 0; };
                if (ends_with("from_str(")) hex = false;
                else if (!ends_with("from_hex(")) { pos = n; return false; }
            }
            if (hex) {
                auto d = [](char ch) { return (uint64_t)(isdigit((unsigned char)ch) ? ch - '0' : tolower((unsigned char)ch) - 'a' + 10); };
                out = U256{};
                for (size_t i = 0; i < 32; ++i) out.w[i / 8] |= (d(s[pos + 1 + 2 * i]) * 16 + d(s[pos + 2 + 2 * i])) << (8 * (i % 8));
                pos += 66;
                return true;
            }
        }
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
inline size_t find_key(const char *s, size_t n, std::initializer_list<const char *> keys, size_t from = 0) {
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


// ------------------------------------------------------------------------------------------------ Poseidon tables
// text -> the (9 + 165) x 32-byte layout of mina_poseidon_set_params (mds row-major, then rc[round][element]); MINA_OK or MINA_ERR_FORMAT + `err`
inline int poseidon_params_parse(int field, const char *text, size_t len, uint8_t *params_out, std::string &err) {
    if (!text || !params_out) { err = "null argument"; return MINA_ERR_ARG; }
    if (field != MINA_FIELD_FP && field != MINA_FIELD_FQ) { err = "bad field"; return MINA_ERR_ARG; }
    const size_t at_mds = find_key(text, len, {"mds", "MDS"}), at_rc = find_key(text, len, {"round_constants", "roundConstants", "ROUND_CONSTANTS", "rc"});
    if (at_mds == std::string::npos || at_rc == std::string::npos) { err = "no `mds` / `round_constants` key in the Poseidon table text"; return MINA_ERR_FORMAT; }
    auto take = [&](size_t pos, size_t count, uint8_t *dst, size_t stop) -> bool {
        for (size_t i = 0; i < count; ++i) {
            U256 v;
            if (!next_number(text, len, pos, v) || pos > stop) return false;
            to_le32(v, dst + 32 * i);
            if (!(field == MINA_FIELD_FP ? mw::fp_canonical(dst + 32 * i) : mw::fq_canonical(dst + 32 * i))) return false;
        }
        return true;
    };
    // each table ends where the other begins (whichever comes second runs to the end of the text)
    const size_t mds_stop = at_mds < at_rc ? at_rc : len, rc_stop = at_rc < at_mds ? at_mds : len;
    if (!take(at_mds, 9, params_out, mds_stop)) { err = "the MDS matrix needs 9 canonical field elements"; return MINA_ERR_FORMAT; }
    if (!take(at_rc, 165, params_out + 9 * 32, rc_stop)) { err = "the round constants need 55 x 3 canonical field elements"; return MINA_ERR_FORMAT; }
    { size_t pos = at_rc; U256 v; size_t cnt = 0; while (next_number(text, len, pos, v) && pos <= rc_stop) ++cnt;      // a table of another shape (e.g. the 100-round legacy one) is refused
      // trailing scalars of the o1js object (fullRounds: 55, stateSize: 3, ...) may follow the table when it comes last: allow up to 8
      if (cnt > 165 + 8) { err = "more than 55 x 3 round constants: not the Kimchi parameter shape"; return MINA_ERR_FORMAT; } }
    return MINA_OK;
}

// ================================================================================================ a small JSON reader
struct JVal {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    bool b = false; double num = 0; bool num_is_int = false; long long inum = 0; std::string str;
    std::vector<JVal> arr; std::vector<std::pair<std::string, JVal>> obj;
    const JVal *get(const char *key) const { if (kind != OBJ) return nullptr; for (auto &kv : obj) if (kv.first == key) return &kv.second; return nullptr; }
    const JVal *get_any(std::initializer_list<const char *> keys) const { for (const char *k : keys) if (const JVal *v = get(k)) return v; return nullptr; }
};
struct JParser {
    const char *s; size_t n, p = 0; int depth = 0; bool ok = true;
    void ws() { while (p < n && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) ++p; }
    bool lit(const char *w) { const size_t l = strlen(w); if (n - p >= l && !memcmp(s + p, w, l)) { p += l; return true; } return false; }
    JVal value() {
        JVal v; ws();
        if (!ok || p >= n || ++depth > 64) { ok = false; return v; }
        const char c = s[p];
        if (c == '{') {
            ++p; v.kind = JVal::OBJ; ws();
            if (p < n && s[p] == '}') { ++p; --depth; return v; }
            for (;;) {
                ws(); JVal k = value(); if (!ok || k.kind != JVal::STR) { ok = false; break; }
                ws(); if (p >= n || s[p] != ':') { ok = false; break; } ++p;
                JVal x = value(); if (!ok) break;
                v.obj.emplace_back(std::move(k.str), std::move(x));
                ws(); if (p < n && s[p] == ',') { ++p; continue; }
                if (p < n && s[p] == '}') { ++p; break; }
                ok = false; break;
            }
        } else if (c == '[') {
            ++p; v.kind = JVal::ARR; ws();
            if (p < n && s[p] == ']') { ++p; --depth; return v; }
            for (;;) {
                JVal x = value(); if (!ok) break;
                v.arr.push_back(std::move(x));
                ws(); if (p < n && s[p] == ',') { ++p; continue; }
                if (p < n && s[p] == ']') { ++p; break; }
                ok = false; break;
            }
        } else if (c == '"') {
            ++p; v.kind = JVal::STR;
            while (p < n && s[p] != '"') {
                if (s[p] == '\\') { if (p + 1 >= n) { ok = false; break; } const char e = s[p + 1]; p += 2;
                    if (e == 'n') v.str.push_back('\n'); else if (e == 't') v.str.push_back('\t'); else if (e == 'u') { if (n - p < 4) { ok = false; break; } v.str.push_back('?'); p += 4; } else v.str.push_back(e); }
                else v.str.push_back(s[p++]);
            }
            if (p >= n) ok = false; else ++p;
        } else if (lit("true")) { v.kind = JVal::BOOL; v.b = true; }
        else if (lit("false")) { v.kind = JVal::BOOL; v.b = false; }
        else if (lit("null")) { v.kind = JVal::NUL; }
        else if (c == '-' || isdigit((unsigned char)c)) {
            const size_t st = p; if (c == '-') ++p;
            bool isint = true;
            while (p < n && (isdigit((unsigned char)s[p]) || s[p] == '.' || s[p] == 'e' || s[p] == 'E' || s[p] == '+' || s[p] == '-')) { if (!isdigit((unsigned char)s[p])) isint = false; ++p; }
            v.kind = JVal::NUM; const std::string t(s + st, p - st); v.num = atof(t.c_str()); v.num_is_int = isint; if (isint) v.inum = atoll(t.c_str());
        } else ok = false;
        --depth;
        return v;
    }
};
inline bool parse_json(const char *s, size_t n, JVal &out) { JParser p{s, n}; out = p.value(); p.ws(); return p.ok && p.p == n; }

inline bool hex_bytes(const std::string &h, std::vector<uint8_t> &out) {
    size_t st = (h.size() >= 2 && h[0] == '0' && (h[1] == 'x' || h[1] == 'X')) ? 2 : 0;
    if ((h.size() - st) % 2) return false;
    out.clear();
    for (size_t i = st; i < h.size(); i += 2) {
        if (!isxdigit((unsigned char)h[i]) || !isxdigit((unsigned char)h[i + 1])) return false;
        auto d = [](char c) { return isdigit((unsigned char)c) ? c - '0' : tolower(c) - 'a' + 10; };
        out.push_back((uint8_t)(d(h[i]) * 16 + d(h[i + 1])));
    }
    return true;
}
// a field element as serde_json writes it through o1-utils `SerdeAs`: hex of the 32 little-endian bytes; also accepted: a decimal string.
// The rule is explicit: a string of exactly 64 hex digits (optionally behind 0x) is ALWAYS the serde byte-hex form -- also when every digit happens to
// be decimal (a 64-digit decimal literal is not accepted: pad or shorten it) -- and any other all-decimal string is a decimal integer.
inline bool json_field(const JVal &v, int field, uint8_t *out32) {
    if (v.kind != JVal::STR) return false;
    std::vector<uint8_t> b;
    bool all_dec = !v.str.empty(); for (char ch : v.str) if (!isdigit((unsigned char)ch)) all_dec = false;
    if (all_dec && v.str.size() != 64) { size_t pos = 0; U256 x; if (!next_number(v.str.data(), v.str.size(), pos, x) || pos != v.str.size()) return false; to_le32(x, out32); }
    else { if (!hex_bytes(v.str, b) || b.size() != 32) return false; memcpy(out32, b.data(), 32); }
    return field == MINA_FIELD_FP ? mw::fp_canonical(out32) : mw::fq_canonical(out32);
}

// ---------------------------------------------------------------------------------------------- PolishToken JSON -> byte-code
// serde_json of kimchi `Vec<PolishToken<F, Column>>` (externally tagged enums): "Alpha", {"Mds": {"row": 0, "col": 1}}, {"Literal": "<hex>"},
// {"Cell": {"col": {"Witness": 3}, "row": "Curr"}}, "Dup", {"Pow": 7}, "Add", "Mul", "Sub", "VanishesOnZeroKnowledgeAndPreviousRows",
// {"UnnormalizedLagrangeBasis": {"zk_rows": true, "offset": -1}}, "Store", {"Load": 2}, {"SkipIf": [<feature>, n]}, {"SkipIfNot": [<feature>, n]};
// the later spelling {"Challenge": "Alpha"} / {"Constant": "EndoCoefficient" | {"Mds": ..} | {"Literal": ..}} is accepted too.
struct TokOut { std::vector<uint8_t> code; std::string err; };
static const char *const GATE_SELECTORS[6] = {"Generic", "Poseidon", "CompleteAdd", "VarBaseMul", "EndoMul", "EndoMulScalar"};
// optional evaluations in wire order (wire_proof.h rd_all_evals): 6 gate selectors, lookup aggregation / table, 5 sorted, runtime table + selector, 4 lookup selectors
static const char *const OPTIONAL_GATES[6] = {"RangeCheck0", "RangeCheck1", "ForeignFieldAdd", "ForeignFieldMul", "Xor16", "Rot64"};
static const char *const LOOKUP_PATTERNS[4] = {"Xor", "Lookup", "RangeCheck", "ForeignFieldMul"};

inline int feature_bit(const JVal &f) {            // kimchi FeatureFlag -> the byte-code's feature code (csrc/polish.h): gates 0..5, LookupTables 6, RuntimeLookupTables 7,
    if (f.kind == JVal::STR) {              // LookupPattern(p) 8 + p, TableWidth(w) 12 + w, LookupsPerRow(n) 16 + n
        for (int i = 0; i < 6; ++i) if (f.str == OPTIONAL_GATES[i]) return i;
        if (f.str == "Xor") return 4;
        if (f.str == "Rot") return 5;
        if (f.str == "LookupTables") return 6;
        if (f.str == "RuntimeLookupTables") return 7;
        return -1;
    }
    if (f.kind == JVal::OBJ && f.obj.size() == 1) {
        const JVal &v = f.obj[0].second;
        if (f.obj[0].first == "LookupPattern" && v.kind == JVal::STR) { for (int i = 0; i < 4; ++i) if (v.str == LOOKUP_PATTERNS[i]) return 8 + i; }
        if (f.obj[0].first == "TableWidth" && v.kind == JVal::NUM && v.num_is_int && v.inum >= 0 && v.inum <= 3) return 12 + (int)v.inum;
        if (f.obj[0].first == "LookupsPerRow" && v.kind == JVal::NUM && v.num_is_int && v.inum >= 0 && v.inum <= 4) return 16 + (int)v.inum;
    }
    return -1;
}
inline bool column_index(const JVal &col, uint32_t optional_present, uint32_t &out, std::string &err) {
    auto optional = [&](int slot) -> bool {                          // the byte-code names the SLOT (43 + slot, wire order); the proof's presence mask places it
        if (optional_present != 0xffffffffu && !(optional_present >> slot & 1)) { err = "the program names an optional evaluation the proofs do not carry"; return false; }
        out = 43 + (uint32_t)slot; return true;
    };
    if (col.kind == JVal::STR) {
        if (col.str == "Z") { out = 0; return true; }
        if (col.str == "LookupAggreg") return optional(6);
        if (col.str == "LookupTable") return optional(7);
        if (col.str == "LookupRuntimeTable") return optional(13);
        if (col.str == "LookupRuntimeSelector") return optional(14);
        err = "unknown column " + col.str; return false;
    }
    if (col.kind != JVal::OBJ || col.obj.size() != 1) { err = "malformed column"; return false; }
    const std::string &k = col.obj[0].first; const JVal &v = col.obj[0].second;
    if (k == "Witness" && v.kind == JVal::NUM && v.num_is_int && v.inum >= 0 && v.inum < 15) { out = 7 + (uint32_t)v.inum; return true; }
    if (k == "Coefficient" && v.kind == JVal::NUM && v.num_is_int && v.inum >= 0 && v.inum < 15) { out = 22 + (uint32_t)v.inum; return true; }
    if (k == "Permutation" && v.kind == JVal::NUM && v.num_is_int && v.inum >= 0 && v.inum < 6) { out = 37 + (uint32_t)v.inum; return true; }
    if (k == "Index" && v.kind == JVal::STR) {
        for (int i = 0; i < 6; ++i) if (v.str == GATE_SELECTORS[i]) { out = 1 + i; return true; }
        for (int i = 0; i < 6; ++i) if (v.str == OPTIONAL_GATES[i]) return optional(i);
        err = "unknown gate selector " + v.str; return false;
    }
    if (k == "LookupSorted" && v.kind == JVal::NUM && v.num_is_int && v.inum >= 0 && v.inum < 5) return optional(8 + (int)v.inum);
    if (k == "LookupKindIndex" && v.kind == JVal::STR) { for (int i = 0; i < 4; ++i) if (v.str == LOOKUP_PATTERNS[i]) return optional(15 + i); }
    err = "unknown column " + k; return false;
}

inline bool tokens_from_json(const JVal &root, int field, uint32_t enabled_features, uint32_t optional_present, TokOut &o) {
    if (root.kind != JVal::ARR) { o.err = "the token program must be a JSON array"; return false; }
    auto put = [&](uint8_t b) { o.code.push_back(b); };
    auto challenge = [&](const std::string &s) -> bool {
        if (s == "Alpha") put(MINA_TOK_ALPHA); else if (s == "Beta") put(MINA_TOK_BETA); else if (s == "Gamma") put(MINA_TOK_GAMMA); else if (s == "JointCombiner") put(MINA_TOK_JOINT_COMBINER); else return false;
        return true;
    };
    auto mds = [&](const JVal &v) -> bool {
        const JVal *r = v.get("row"), *c = v.get("col");
        if (!r || !c || r->kind != JVal::NUM || c->kind != JVal::NUM || r->inum < 0 || r->inum > 2 || c->inum < 0 || c->inum > 2) return false;
        put(MINA_TOK_MDS); put((uint8_t)r->inum); put((uint8_t)c->inum); return true;
    };
    auto literal = [&](const JVal &v) -> bool { uint8_t b[32]; if (!json_field(v, field, b)) return false; put(MINA_TOK_LITERAL); o.code.insert(o.code.end(), b, b + 32); return true; };
    for (size_t i = 0; i < root.arr.size(); ++i) {
        const JVal &t = root.arr[i];
        auto bad = [&](const std::string &why) { o.err = "token " + std::to_string(i) + ": " + why; return false; };
        if (t.kind == JVal::STR) {
            const std::string &s = t.str;
            if (challenge(s)) continue;
            if (s == "EndoCoefficient") put(MINA_TOK_ENDO_COEFFICIENT);
            else if (s == "Dup") put(MINA_TOK_DUP); else if (s == "Add") put(MINA_TOK_ADD); else if (s == "Mul") put(MINA_TOK_MUL); else if (s == "Sub") put(MINA_TOK_SUB);
            else if (s == "VanishesOnZeroKnowledgeAndPreviousRows") put(MINA_TOK_VANISHES_ON_ZK_ROWS);
            else if (s == "Store") put(MINA_TOK_STORE);
            else return bad("unknown token " + s);
            continue;
        }
        if (t.kind != JVal::OBJ || t.obj.size() != 1) return bad("malformed token");
        const std::string &k = t.obj[0].first; const JVal &v = t.obj[0].second;
        if (k == "Challenge") { if (v.kind != JVal::STR || !challenge(v.str)) return bad("unknown challenge"); }
        else if (k == "Constant") {
            if (v.kind == JVal::STR && v.str == "EndoCoefficient") put(MINA_TOK_ENDO_COEFFICIENT);
            else if (v.kind == JVal::OBJ && v.obj.size() == 1 && v.obj[0].first == "Mds") { if (!mds(v.obj[0].second)) return bad("malformed Mds"); }
            else if (v.kind == JVal::OBJ && v.obj.size() == 1 && v.obj[0].first == "Literal") { if (!literal(v.obj[0].second)) return bad("malformed literal"); }
            else return bad("unknown constant");
        }
        else if (k == "Mds") { if (!mds(v)) return bad("malformed Mds"); }
        else if (k == "Literal") { if (!literal(v)) return bad("literal is not a canonical field element"); }
        else if (k == "Cell") {
            const JVal *col = v.get("col"), *row = v.get("row");
            uint32_t ci = 0; std::string err;
            if (!col || !row || row->kind != JVal::STR || (row->str != "Curr" && row->str != "Next")) return bad("malformed cell");
            if (!column_index(*col, optional_present, ci, err)) return bad(err);
            put(MINA_TOK_CELL); put((uint8_t)ci); put(row->str == "Next" ? 1 : 0);
        }
        else if (k == "Pow") { if (v.kind != JVal::NUM || !v.num_is_int || v.inum < 0) return bad("malformed Pow"); put(MINA_TOK_POW); const uint64_t e = (uint64_t)v.inum; for (int j = 0; j < 8; ++j) put((uint8_t)(e >> (8 * j))); }
        else if (k == "UnnormalizedLagrangeBasis") {
            // RowOffset {zk_rows, offset}: with zk_rows the offset counts back from the first zero-knowledge row -- the byte-code's negative form
            long long off = 0; bool zk = false;
            if (v.kind == JVal::NUM && v.num_is_int) off = v.inum;                       // older spelling: a bare i32
            else { const JVal *z = v.get("zk_rows"), *f = v.get("offset"); if (!z || !f || z->kind != JVal::BOOL || f->kind != JVal::NUM || !f->num_is_int) return bad("malformed row offset"); zk = z->b; off = f->inum; }
            if (zk) { if (off > 0) return bad("a zero-knowledge-relative row offset must be <= 0"); off = off == 0 ? INT32_MIN : off; }
            else if (off < 0) return bad("an absolute row offset must be >= 0");
            if (off != INT32_MIN && (off < -(1 << 30) || off > (1 << 30))) return bad("row offset out of range");
            put(MINA_TOK_UNNORMALIZED_LAGRANGE); const int32_t o32 = (int32_t)off; uint8_t b[4]; memcpy(b, &o32, 4); o.code.insert(o.code.end(), b, b + 4);
        }
        else if (k == "Load") { if (v.kind != JVal::NUM || !v.num_is_int || v.inum < 0 || v.inum > 65535) return bad("malformed Load"); put(MINA_TOK_LOAD); put((uint8_t)v.inum); put((uint8_t)(v.inum >> 8)); }
        else if (k == "SkipIf" || k == "SkipIfNot") {
            // resolved here, against the feature set the index was built for: kimchi pushes zero and skips `n` tokens when the condition holds
            if (v.kind != JVal::ARR || v.arr.size() != 2 || v.arr[1].kind != JVal::NUM || !v.arr[1].num_is_int || v.arr[1].inum < 0) return bad("malformed skip");
            const int bit = feature_bit(v.arr[0]);
            if (bit < 0) return bad("unknown feature flag");
            if (enabled_features == 0xffffffffu) {                   // run-time form: the interpreter looks at each proof's own flags
                if (v.arr[1].inum == 0 || v.arr[1].inum > 65535) return bad("skip count out of range");
                put(k == "SkipIf" ? MINA_TOK_SKIP_IF : MINA_TOK_SKIP_IF_NOT); put((uint8_t)bit); put((uint8_t)v.arr[1].inum); put((uint8_t)(v.arr[1].inum >> 8));
                continue;
            }
            const bool on = (enabled_features >> bit) & 1;
            if ((k == "SkipIf") == on) {
                const size_t cnt = (size_t)v.arr[1].inum;
                if (cnt > root.arr.size() - 1 - i) return bad("skip runs past the end of the program");
                put(MINA_TOK_LITERAL); o.code.insert(o.code.end(), 32, 0);
                i += cnt;
            }
        }
        else return bad("unknown token " + k);
    }
    return true;
}

inline int polish_tokens_from_json(int field, const char *json, size_t len, uint32_t enabled_features, uint32_t optional_present, uint8_t *out, size_t cap, size_t *out_len, std::string &err) {
    if (!json || !out_len) { err = "null argument"; return MINA_ERR_ARG; }
    if (field != MINA_FIELD_FP && field != MINA_FIELD_FQ) { err = "bad field"; return MINA_ERR_ARG; }
    JVal root;
    if (!parse_json(json, len, root)) { err = "the token program is not valid JSON"; return MINA_ERR_FORMAT; }
    TokOut o;
    if (!tokens_from_json(root, field, enabled_features, optional_present, o)) { err = o.err; return MINA_ERR_FORMAT; }
    *out_len = o.code.size();
    if (out) { if (cap < o.code.size()) { err = "output buffer too small"; return MINA_ERR_ARG; } memcpy(out, o.code.data(), o.code.size()); }
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------ VerifierIndex JSON
struct IndexFields { uint32_t log2_domain = 0, zk_rows = 3; uint8_t shifts[7 * 32]; uint8_t sigma[7 * 64], coeff[15 * 64], sel[6 * 64]; };
// field of the shifts = the circuit's scalar field
// `point`: decompresses / validates one commitment (33-byte compressed or 64-byte x || y) into 64 canonical bytes -- field arithmetic, supplied by the caller
// (api_loaders.hip: ark's Tonelli-Shanks on the host; the fuzz harness: a size check); commitments are read only when it is given (the wrap index)
typedef std::function<bool(const std::vector<uint8_t> &, uint8_t *)> PointReader;
// PolyComm: {"elems": ["<hex>"]} (one chunk) or the older {"unshifted": ["<hex>"], "shifted": null}
inline bool json_polycomm(const JVal *v, const PointReader &point, uint8_t *out64) {
    if (!v) return false;
    const JVal *e = v->get_any({"elems", "unshifted", "chunks"});
    if (!e || e->kind != JVal::ARR || e->arr.size() != 1 || e->arr[0].kind != JVal::STR) return false;       // the wrap / step domains fit one SRS chunk
    std::vector<uint8_t> b;
    return hex_bytes(e->arr[0].str, b) && point(b, out64);
}
inline bool index_fields_from_json(const JVal &root, int scalar_field, const PointReader *point, IndexFields &o, std::string &err) {
    if (root.kind != JVal::OBJ) { err = "the verifier index must be a JSON object"; return false; }
    const JVal *dom = root.get("domain");
    if (dom && dom->kind == JVal::STR) {                        // ark bytes of Radix2EvaluationDomain: size u64, log_size_of_group u32, then five field elements
        std::vector<uint8_t> b; if (!hex_bytes(dom->str, b) || b.size() < 12) { err = "malformed domain"; return false; }
        memcpy(&o.log2_domain, b.data() + 8, 4);
        uint64_t size; memcpy(&size, b.data(), 8);
        if (o.log2_domain > 32 || size != ((uint64_t)1 << o.log2_domain)) { err = "domain size and log size disagree"; return false; }
    } else if (dom && dom->kind == JVal::OBJ) {
        const JVal *l = dom->get_any({"log_size_of_group", "log2_size"}); if (!l || l->kind != JVal::NUM) { err = "malformed domain"; return false; }
        if (!l->num_is_int || l->inum < 0 || l->inum > 32) { err = "malformed domain"; return false; }
        o.log2_domain = (uint32_t)l->inum;
    } else { err = "no `domain`"; return false; }
    if (const JVal *z = root.get("zk_rows")) { if (z->kind != JVal::NUM || z->inum < 1 || z->inum > 8) { err = "bad zk_rows"; return false; } o.zk_rows = (uint32_t)z->inum; }
    const JVal *sh = root.get_any({"shift", "shifts"});
    if (!sh || sh->kind != JVal::ARR || sh->arr.size() != 7) { err = "`shift` must hold 7 field elements"; return false; }
    for (int i = 0; i < 7; ++i) if (!json_field(sh->arr[i], scalar_field, o.shifts + 32 * i)) { err = "shift is not a canonical field element"; return false; }
    if (!point) return true;
    const JVal *sg = root.get("sigma_comm"), *cf = root.get("coefficients_comm");
    if (!sg || sg->kind != JVal::ARR || sg->arr.size() != 7 || !cf || cf->kind != JVal::ARR || cf->arr.size() != 15) { err = "sigma_comm / coefficients_comm must hold 7 / 15 commitments"; return false; }
    for (int i = 0; i < 7; ++i) if (!json_polycomm(&sg->arr[i], *point, o.sigma + 64 * i)) { err = "malformed sigma commitment"; return false; }
    for (int i = 0; i < 15; ++i) if (!json_polycomm(&cf->arr[i], *point, o.coeff + 64 * i)) { err = "malformed coefficient commitment"; return false; }
    static const char *names[6] = {"generic_comm", "psm_comm", "complete_add_comm", "mul_comm", "emul_comm", "endomul_scalar_comm"};
    for (int i = 0; i < 6; ++i) if (!json_polycomm(root.get(names[i]), *point, o.sel + 64 * i)) { err = std::string("malformed / missing ") + names[i]; return false; }
    // an index built with optional gates or lookups needs terms this library's linearization interpreter does not carry: refuse it here
    static const char *optional[] = {"range_check0_comm", "range_check1_comm", "foreign_field_add_comm", "foreign_field_mul_comm", "xor_comm", "rot_comm", "lookup_index"};
    for (const char *nm : optional) if (const JVal *v = root.get(nm)) if (v->kind != JVal::NUL) { err = std::string("the index enables `") + nm + "`: optional gates / lookups are not supported"; return false; }
    return true;
}
}  // namespace mbl
