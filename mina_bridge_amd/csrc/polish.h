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

// byte-code -> fixed-width tokens + literal table; validates operand ranges, stack depth and cache use so that the interpreters
// need no run-time checks.  `field`: the field the literals live in; `ncols`: evaluation columns a CELL may name.
static inline bool decode_tokens(const uint8_t *code, size_t len, int field, uint32_t ncols, std::vector<KimchiToken> &toks, std::vector<std::array<uint8_t, 32>> &lits) {
    size_t p = 0; int depth = 0, cache = 0;
    auto need = [&](size_t k) { return len - p >= k; };
    while (p < len) {
        KimchiToken t{code[p++], 0, 0, 0};
        switch (t.op) {
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
    }
    return toks.empty() || depth == 1;
}

// host interpreter (Montgomery values): evals[col][row]; returns false on a program that names a column the proof does not carry
template <int F> struct PolishEnv {
    fe_t alpha, beta, gamma, endo_coeff, zkpm, zeta, zeta1, omega; const fe_t *mds; uint32_t log2_domain, zk_rows;
    const std::vector<std::array<fe_t, 2>> *evals;
};
template <int F> static inline fe_t host_pow_u64(fe_t base, uint64_t e, const fe_t &one) { fe_t r = one; for (; e; e >>= 1) { if (e & 1) r = fe_mul<F>(r, base); base = fe_sqr<F>(base); } return r; }
template <int F> static inline bool polish_eval_host(const std::vector<KimchiToken> &toks, const std::vector<fe_t> &lits, const PolishEnv<F> &env, const FieldK &k, fe_t &out) {
    fe_t stack[KC_STACK], cache[KC_CACHE]; int sp = 0, nc = 0;
    for (const KimchiToken &tk : toks) {
        switch (tk.op) {
            case MINA_TOK_ALPHA: stack[sp++] = env.alpha; break;
            case MINA_TOK_BETA: stack[sp++] = env.beta; break;
            case MINA_TOK_GAMMA: stack[sp++] = env.gamma; break;
            case MINA_TOK_JOINT_COMBINER: stack[sp++] = fe_zero(); break;
            case MINA_TOK_ENDO_COEFFICIENT: stack[sp++] = env.endo_coeff; break;
            case MINA_TOK_MDS: stack[sp++] = env.mds[tk.a * 3 + tk.b]; break;
            case MINA_TOK_LITERAL: stack[sp++] = lits[tk.a]; break;
            case MINA_TOK_CELL: if (tk.a >= env.evals->size()) return false; stack[sp++] = (*env.evals)[tk.a][tk.b]; break;
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
            case MINA_TOK_LOAD: stack[sp++] = cache[tk.a]; break;
            default: return false;
        }
    }
    if (sp != 1) return false;
    out = stack[0];
    return true;
}

}  // namespace mb
