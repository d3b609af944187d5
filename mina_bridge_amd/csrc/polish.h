// polish.h -- kimchi `PolishToken` programs (the linearization's constant term) in the engine's byte-code (include/mina_verify.h
// MINA_TOK_*): host-side decoder / validator shared by the wrap index (api_kimchi.hip: interpreted on the GPU, one lane group per
// proof) and the step index (api_pickles.hip: interpreted on the host while deriving the deferred values), plus the host interpreter.
#pragma once
#include <array>
#include <vector>

#include "../../include/mina_verify.h"
#include "groupmap.cuh"
#include "wire_state.h"

namespace mb {

static constexpr uint32_t KC_STACK = 24, KC_CACHE = 8;
struct KimchiToken { uint32_t op, a, b, c; };

// ---- feature flags of a (step) proof, as kimchi's `FeatureFlag` sees them (SkipIf / SkipIfNot tokens).  [UPSTREAM-RECALL]
// Pickles statement flags, in order: range_check0, range_check1, foreign_field_add, foreign_field_mul, xor, rot, lookup, runtime_tables.
// Feature codes of the byte-code: 0..5 the optional gates (same order); 6 LookupTables; 7 RuntimeLookupTables; 8..11 LookupPattern
// Xor / Lookup / RangeCheck / ForeignFieldMul; 12 + w: TableWidth(w), w <= 3; 16 + n: LookupsPerRow(n), n <= 4.
// Lookup patterns in use: Xor <- xor; Lookup <- lookup; RangeCheck <- range_check0 | range_check1 | rot; ForeignFieldMul <- foreign_field_mul;
// joint table width / lookups per row = the maximum over the patterns in use (Xor 3 / 4, Lookup 2 / 3, RangeCheck 1 / 4, ForeignFieldMul 2 / 2).
#ifdef __HIPCC__
#define MB_POLISH_HD __host__ __device__
#else
#define MB_POLISH_HD
#endif
MB_POLISH_HD static inline uint32_t feature_mask_of_flags(const uint8_t *f /* 8 */) {
    const bool rc0 = f[0], rc1 = f[1], ffadd = f[2], ffmul = f[3], x = f[4], rot = f[5], lk = f[6], rt = f[7];
    const bool pxor = x, plk = lk, prc = rc0 || rc1 || rot, pff = ffmul;
    const bool any = pxor || plk || prc || pff;
    const uint32_t width = pxor ? 3 : (plk || pff) ? 2 : prc ? 1 : 0, per_row = (pxor || prc) ? 4 : plk ? 3 : pff ? 2 : 0;
    uint32_t m = (rc0 ? 1u : 0) | (rc1 ? 2u : 0) | (ffadd ? 4u : 0) | (ffmul ? 8u : 0) | (x ? 16u : 0) | (rot ? 32u : 0) | (any ? 64u : 0) | (rt ? 128u : 0) |
                 (pxor ? 256u : 0) | (plk ? 512u : 0) | (prc ? 1024u : 0) | (pff ? 2048u : 0);
    for (uint32_t w = 0; w <= 3; ++w) if (width >= w && (w == 0 || any)) m |= 1u << (12 + w);
    for (uint32_t n = 0; n <= 4; ++n) if (per_row >= n && (n == 0 || any)) m |= 1u << (16 + n);
    return m;
}
static constexpr uint32_t MB_N_FEATURE_CODES = 21;

// byte-code -> fixed-width tokens + literal table; validates operand ranges, stack depth and cache use so that the interpreters
// need no run-time checks.  `field`: the field the literals live in; `ncols`: evaluation columns a CELL may name.
static inline bool decode_tokens(const uint8_t *code, size_t len, int field, uint32_t ncols, std::vector<KimchiToken> &toks, std::vector<std::array<uint8_t, 32>> &lits) {
    size_t p = 0; int depth = 0, cache = 0;
    auto need = [&](size_t k) { return len - p >= k; };
    // SkipIf / SkipIfNot regions: (index of the region's last token, depth the stack must have there).  kimchi pushes ZERO and skips the
    // region when the condition holds, so a region must net exactly one value whichever way it goes.  ONE convention everywhere, upstream's
    // (kimchi `PolishToken::evaluate`: `if skip_count > 0 { skip_count -= 1; continue; }`, `Store => cache.push(top)`, `Load(i) => cache[i]`):
    // SKIPPED TOKENS HAVE NO EFFECT -- a skipped STORE takes no cache slot, so after a region with a STORE in it the index a later STORE lands on
    // depends on the proof's flags.  The static resolution of api_loaders.hip (skipped tokens dropped for a fixed feature set) is the same function.
    // Here `cache` counts the STOREs so far as if every region ran: the capacity bound, and a NECESSARY bound for LOAD indices; the interpreters
    // check every LOAD against the slots really filled and fail the proof otherwise (upstream would panic on the out-of-range index).
    std::vector<std::pair<size_t, int>> regions;
    while (p < len) {
        KimchiToken t{code[p++], 0, 0, 0};
        switch (t.op) {
            case MINA_TOK_SKIP_IF: case MINA_TOK_SKIP_IF_NOT:
                if (!need(3)) return false;
                t.a = code[p]; t.b = code[p + 1] | (code[p + 2] << 8); p += 3;
                if (t.a >= MB_N_FEATURE_CODES || t.b == 0) return false;
                regions.emplace_back(toks.size() + t.b, depth + 1);
                break;
            case MINA_TOK_ALPHA: case MINA_TOK_BETA: case MINA_TOK_GAMMA: case MINA_TOK_JOINT_COMBINER: case MINA_TOK_ENDO_COEFFICIENT: case MINA_TOK_VANISHES_ON_ZK_ROWS: ++depth; break;
            case MINA_TOK_MDS: if (!need(2)) return false; t.a = code[p]; t.b = code[p + 1]; p += 2; if (t.a > 2 || t.b > 2) return false; ++depth; break;
            case MINA_TOK_LITERAL: { if (!need(32) || !(field == FIELD_FP ? mw::fp_canonical(code + p) : mw::fq_canonical(code + p))) return false;
                                     std::array<uint8_t, 32> l; memcpy(l.data(), code + p, 32); p += 32; t.a = (uint32_t)lits.size(); lits.push_back(l); ++depth; break; }
            case MINA_TOK_CELL: if (!need(2)) return false; t.a = code[p]; t.b = code[p + 1]; p += 2; if (t.a >= ncols || t.b > 1) return false; ++depth; break;
            case MINA_TOK_DUP: if (depth < 1) return false; ++depth; break;
            case MINA_TOK_POW: if (!need(8) || depth < 1) return false; memcpy(&t.a, code + p, 4); memcpy(&t.b, code + p + 4, 4); p += 8; break;
            case MINA_TOK_ADD: case MINA_TOK_MUL: case MINA_TOK_SUB: if (depth < 2) return false; --depth; break;
            case MINA_TOK_UNNORMALIZED_LAGRANGE: if (!need(4)) return false; memcpy(&t.a, code + p, 4); p += 4; ++depth; break;
            case MINA_TOK_STORE: if (depth < 1 || cache >= (int)KC_CACHE) return false; ++cache; break;
            case MINA_TOK_LOAD: if (!need(2)) return false; t.a = code[p] | (code[p + 1] << 8); p += 2; if ((int)t.a >= cache) return false; ++depth; break;
            default: return false;
        }
        if (depth > (int)KC_STACK) return false;
        toks.push_back(t);
        while (!regions.empty() && regions.back().first == toks.size() - 1) { if (depth != regions.back().second) return false; regions.pop_back(); }
        for (const auto &r : regions) if (depth < r.second - 1) return false;       // a region may not consume what was on the stack before it
    }
    return regions.empty() && (toks.empty() || depth == 1);
}

// host interpreter (Montgomery values): evals[col][row]; returns false on a program that names a column the proof does not carry
template <int F> struct PolishEnv {
    fe_t alpha, beta, gamma, endo_coeff, zkpm, zeta, zeta1, omega; const fe_t *mds; uint32_t log2_domain, zk_rows;
    const std::vector<std::array<fe_t, 2>> *evals;
    fe_t joint_combiner = fe_zero(); uint32_t features = 0;           // the proof's joint combiner (0 without one) and feature mask (feature_mask_of_flags)
    uint32_t present = 0; bool slots = false;                          // slots: a CELL column c >= 43 names optional SLOT c - 43, found through the presence mask
};
template <int F> static inline fe_t host_pow_u64(fe_t base, uint64_t e, const fe_t &one) { fe_t r = one; for (; e; e >>= 1) { if (e & 1) r = fe_mul<F>(r, base); base = fe_sqr<F>(base); } return r; }
template <int F> static inline bool polish_eval_host(const std::vector<KimchiToken> &toks, const std::vector<fe_t> &lits, const PolishEnv<F> &env, const FieldK &k, fe_t &out) {
    fe_t stack[KC_STACK], cache[KC_CACHE]; int sp = 0, nc = 0; uint32_t skip = 0;
    for (const KimchiToken &tk : toks) {
        if (skip) { --skip; continue; }                                  // a skipped token has no effect (no cache slot for a skipped STORE)
        switch (tk.op) {
            case MINA_TOK_SKIP_IF: case MINA_TOK_SKIP_IF_NOT: {
                const bool on = (env.features >> tk.a) & 1u;
                if (on == (tk.op == MINA_TOK_SKIP_IF)) { skip = tk.b; stack[sp++] = fe_zero(); }
                break; }
            case MINA_TOK_ALPHA: stack[sp++] = env.alpha; break;
            case MINA_TOK_BETA: stack[sp++] = env.beta; break;
            case MINA_TOK_GAMMA: stack[sp++] = env.gamma; break;
            case MINA_TOK_JOINT_COMBINER: stack[sp++] = env.joint_combiner; break;
            case MINA_TOK_ENDO_COEFFICIENT: stack[sp++] = env.endo_coeff; break;
            case MINA_TOK_MDS: stack[sp++] = env.mds[tk.a * 3 + tk.b]; break;
            case MINA_TOK_LITERAL: stack[sp++] = lits[tk.a]; break;
            case MINA_TOK_CELL: {
                uint32_t col = tk.a;
                if (env.slots && col >= 43) { const uint32_t slot = col - 43; if (slot >= 19 || !((env.present >> slot) & 1u)) return false; col = 43 + (uint32_t)__builtin_popcount(env.present & ((1u << slot) - 1)); }
                if (col >= env.evals->size()) return false;
                stack[sp++] = (*env.evals)[col][tk.b]; break; }
            case MINA_TOK_DUP: stack[sp] = stack[sp - 1]; ++sp; break;
            case MINA_TOK_POW: stack[sp - 1] = host_pow_u64<F>(stack[sp - 1], (uint64_t)tk.a | ((uint64_t)tk.b << 32), k.one); break;
            case MINA_TOK_ADD: stack[sp - 2] = fe_add<F>(stack[sp - 2], stack[sp - 1]); --sp; break;
            case MINA_TOK_MUL: stack[sp - 2] = fe_mul<F>(stack[sp - 2], stack[sp - 1]); --sp; break;
            case MINA_TOK_SUB: stack[sp - 2] = fe_sub<F>(stack[sp - 2], stack[sp - 1]); --sp; break;
            case MINA_TOK_VANISHES_ON_ZK_ROWS: stack[sp++] = env.zkpm; break;
            case MINA_TOK_UNNORMALIZED_LAGRANGE: {
                const int32_t off = (int32_t)tk.a;
                const uint64_t row = off >= 0 ? (uint64_t)off : ((uint64_t)1 << env.log2_domain) - env.zk_rows - (off == INT32_MIN ? 0 : (uint64_t)(-(int64_t)off));   // INT32_MIN: the first zero-knowledge row itself
                stack[sp++] = fe_mul<F>(fe_sub<F>(env.zeta1, k.one), fe_inv<F>(fe_sub<F>(env.zeta, host_pow_u64<F>(env.omega, row, k.one)), k)); break; }
            case MINA_TOK_STORE: cache[nc++] = stack[sp - 1]; break;
            case MINA_TOK_LOAD: if ((int)tk.a >= nc) return false; stack[sp++] = cache[tk.a]; break;      // a slot no executed STORE has filled
            default: return false;
        }
    }
    if (sp != 1) return false;
    out = stack[0];
    return true;
}

}  // namespace mb
