// ec29.cuh -- the XYZZ mixed add of the MSM's accumulate kernels on 9 limbs of 29 bits (fp29.cuh: Montgomery 2^261, no carry instructions, lazy
// reduction).  Round 4: the accumulate kernel -- >= 1 M mixed adds per 2^16 MSM -- is bound by multiply-accumulate issue; in the saturated 8 x 32 form
// a limb product is a `v_mad_u64_u32` + a `v_addc`, here it is the multiply-accumulate alone, a squaring is 99 of them instead of 135, and no
// conditional subtraction exists.  Measured on MI355X (tools/probes/batch_affine_probe.hip, 8.4 M gathered adds per launch, bit-identical buckets):
// 16.2 - 16.8 G adds/s against 12.4 - 13.2 for the 8 x 32 law (+ 27 - 31 %); profiles/r04_group_law_probe.md.
//
// Value discipline (p = 2^254 + c; 2^256 ~ 4 p; a normalised 9-limb value holds up to 2^261) -- PROVEN, not asserted: tools/fe29_bounds.py runs this law on
// intervals (a maximum per limb and per value), tools/gen_fe29.py refuses to emit fp29.cuh when a rule fails, and the constants below (`EC29::*`, generated into
// fp29.cuh) are the ones the proof ran with:
//   * six of the nine products run LAZY (quotient digits unmasked: result < a b / 2^261 + 8.0001 p instead of + p; nine masks less each): pd, r, pp, ppp, zz3, zzz3 --
//     every one whose result only feeds products or a "K p - b"; q, x3 and the dot product y3 stay strict (with a seventh lazy product the invariants have no fixed point)
//   * a - b is a + K p - b with K p in a redundant limb form whose every limb exceeds any normalised limb of b: limb-wise, NO carry pass -- so the top limb must
//     hold by itself: K at least one more than b's bound in units of p (the round-4 bug: ONE p under a canonical y)
//   * accumulator coordinates stay below EC29::INV_X / INV_Y / INV_ZZ / INV_ZZZ times p (26, 6, 10, 9: the least fixed point of the lazy law); table coordinates
//     are canonical (< p), in the 2^261 domain
// The exceptional cases of the group law (the two points equal or opposite: P = 0 mod p) are found EXACTLY: a multiple k p = k 2^254 + k c of p below
// EC29::PD_MAX p has limbs 5..7 and the low 22 bits of limb 8 zero (k c < 2^131) -- four instructions per add -- and only then limbs 0..4 are compared with k c;
// the unit of work (bucket / task) that meets one is handed to the 8 x 32 law (never on SRS points; it keeps the kernels exact on any input).
#pragma once
#include "ec.cuh"
#include "fp29.cuh"

namespace mb {

struct xyzz29_t { fe29_t x, y, zz, zzz; };

template <int F, uint32_t MULT> struct KP29 {   // MULT * p in normalised 29-bit limbs n_0 .. n_5, n_8 (n_6 = n_7 = 0)
    static constexpr uint64_t t0 = (uint64_t)MULT * 1u, n0 = t0 & M29, t1 = (uint64_t)MULT * P29<F>::L1 + (t0 >> 29), n1 = t1 & M29, t2 = (uint64_t)MULT * P29<F>::L2 + (t1 >> 29), n2 = t2 & M29,
                              t3 = (uint64_t)MULT * P29<F>::L3 + (t2 >> 29), n3 = t3 & M29, t4 = (uint64_t)MULT * P29<F>::L4 + (t3 >> 29), n4 = t4 & M29, n5 = t4 >> 29, n8 = (uint64_t)MULT * P29<F>::L8;
    static_assert(n5 < (1u << 29) && n8 >= 2 && n8 < (1u << 29), "limb shape of MULT * p");
};
// a + MULT p - b, normalised; needs b < MULT p (both operands normalised).  K_0 = n_0 + 2^30, K_i = n_i + 2^30 - 2 (0 < i < 8), K_8 = n_8 - 2: the same integer
template <int F, uint32_t MULT> MB_HD fe29_t fe29_sub_kp(const fe29_t &a, const fe29_t &b) {
    typedef KP29<F, MULT> K;
    const uint32_t k[9] = {(uint32_t)K::n0 + (1u << 30), (uint32_t)K::n1 + (1u << 30) - 2, (uint32_t)K::n2 + (1u << 30) - 2, (uint32_t)K::n3 + (1u << 30) - 2, (uint32_t)K::n4 + (1u << 30) - 2,
                           (uint32_t)K::n5 + (1u << 30) - 2, (1u << 30) - 2, (1u << 30) - 2, (uint32_t)K::n8 - 2};
    fe29_t r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L29; ++i) { const uint32_t t = a.v[i] + k[i] - b.v[i] + c; if (i < L29 - 1) { r.v[i] = t & M29; c = t >> 29; } else r.v[i] = t; }
    return r;
}
// MULT p - a - 2 b limb by limb, every limb non-negative (NOT normalised: limbs up to 2^31 + 2^29): MULT p in the redundant form K_0 = n_0 + 2^31, K_i = n_i + 2^31 - 4
// (0 < i < 8), K_8 = n_8 - 4 -- the same integer.  Needs a, b normalised with a_8 + 2 b_8 <= n_8 - 4 (fe29_bounds.py kp_minus_a_minus_2b).  The operand h of
// fe29_sqr_hi_asm: r^2 / 2^261 + (MULT p - ppp - 2 q) in one reduction, where two normalised additions and a fe29_sub_kp took 81 instructions (27 + 9 now).
template <int F, uint32_t MULT> MB_HD fe29_t fe29_kp_minus_a_minus_2b(const fe29_t &a, const fe29_t &b) {
    typedef KP29<F, MULT> K;
    const uint32_t k[9] = {(uint32_t)K::n0 + (1u << 31), (uint32_t)K::n1 + (1u << 31) - 4, (uint32_t)K::n2 + (1u << 31) - 4, (uint32_t)K::n3 + (1u << 31) - 4, (uint32_t)K::n4 + (1u << 31) - 4,
                           (uint32_t)K::n5 + (1u << 31) - 4, (1u << 31) - 4, (1u << 31) - 4, (uint32_t)K::n8 - 4};
    fe29_t r;
#pragma unroll
    for (int i = 0; i < L29; ++i) r.v[i] = k[i] - a.v[i] - 2u * b.v[i];
    return r;
}
// MULT p - b limb by limb, every limb non-negative (not normalised), in the redundant form fe29_sub_kp uses: the operand h of fe29_mul_hi_asm -- a product and
// "+ MULT p - b" in one reduction (9 subtractions + 9 multiply-accumulates by 1 where fe29_sub_kp's carry pass took 36 instructions).  Needs b normalised and -- there
// being no carry pass to lend to the top limb -- b_8 <= (MULT p)_8 - 2, i.e. b < (MULT - 2^-20) p: MULT at least one more than b's bound in units of p
template <int F, uint32_t MULT> MB_HD fe29_t fe29_kp_minus(const fe29_t &b) {
    typedef KP29<F, MULT> K;
    const uint32_t k[9] = {(uint32_t)K::n0 + (1u << 30), (uint32_t)K::n1 + (1u << 30) - 2, (uint32_t)K::n2 + (1u << 30) - 2, (uint32_t)K::n3 + (1u << 30) - 2, (uint32_t)K::n4 + (1u << 30) - 2,
                           (uint32_t)K::n5 + (1u << 30) - 2, (1u << 30) - 2, (1u << 30) - 2, (uint32_t)K::n8 - 2};
    fe29_t r;
#pragma unroll
    for (int i = 0; i < L29; ++i) r.v[i] = k[i] - b.v[i];
    return r;
}
// a + MULT p - b limb by limb, not normalised (limbs up to 2^31): a product's operand as it is -- the strict two-term dot product of the group law holds
// r (q + 8 p - x3) + (8 p - y1) ppp with BOTH differences in this raw form (column maximum 0.69 x 2^64: tests/test_fe29_lazy_model.py)
template <int F, uint32_t MULT> MB_HD fe29_t fe29_add_kp_minus(const fe29_t &a, const fe29_t &b) {
    const fe29_t k = fe29_kp_minus<F, MULT>(b);
    fe29_t r;
#pragma unroll
    for (int i = 0; i < L29; ++i) r.v[i] = a.v[i] + k.v[i];
    return r;
}
MB_HD fe29_t fe29_zero() { fe29_t r; for (int i = 0; i < L29; ++i) r.v[i] = 0; return r; }

// a (normalised, below EC29::PD_MAX p) == 0 mod p?  exact
template <int F> MB_HD bool fe29_is_multiple_of_p(const fe29_t &a) {
    if ((a.v[5] | a.v[6] | a.v[7] | (a.v[8] & 0x3fffffu)) != 0u) return false;          // the cheap necessary test, inlined again at the call site
    const uint64_t k = a.v[8] >> 22;                                                      // k p = k 2^254 + k c
    const uint64_t t0 = k, t1 = k * P29<F>::L1 + (t0 >> 29), t2 = k * P29<F>::L2 + (t1 >> 29), t3 = k * P29<F>::L3 + (t2 >> 29), t4 = k * P29<F>::L4 + (t3 >> 29);
    return a.v[0] == (uint32_t)(t0 & M29) && a.v[1] == (uint32_t)(t1 & M29) && a.v[2] == (uint32_t)(t2 & M29) && a.v[3] == (uint32_t)(t3 & M29) && a.v[4] == (uint32_t)(t4 & M29) && (t4 >> 29) == 0;
}

#if defined(__HIP_DEVICE_COMPILE__)
// x * 2^261 (lazy, below 16 p) -> x * 2^256, canonical words; and back
template <int F> __device__ __forceinline__ fe_t fe29_leave(const fe29_t &a, const fe29_t &leave /* the integer 2^256 mod p */) { return fe_cond_sub_p<F>(fe29_to_words(fe29_mul_asm<F>(a, leave))); }

// acc += (qx, qy): affine, not infinity, qx canonical in the 2^261 domain; qy = the y coordinate of the point ADDED, either normalised (< p: the pre-split table holds
// y and p - y) or the raw limb form 2 p - y of a negated point (the 8-word twin table: fe29_kp_minus)  (madd-2008-s; ec.cuh xyzz_add_affine on the other limbs).
// `first_y()`: that coordinate NORMALISED, called only for the first point of a bucket (the accumulator becomes the point).
// `operands_done()`: called once qx and qy have been consumed (after the first two products, 290 of the ~1750 instructions of an add): the caller issues its NEXT
// gather there, INTO the registers of qx / qy -- the pipelined loop then needs no register copies (a `p = nxt` hand-over cost the pre-split table's kernel 111 v_mov).
// Returns FALSE -- acc untouched -- in the exceptional case (the points are equal or opposite: P = 0 mod p): the caller hands its unit of work to the
// 8 x 32 law (msm.cuh: the redo queue).  No call, no second code path inside the hot loop: an out-of-line fallback cost the kernel 53 VGPRs and 288 B of
// scratch per lane (the call ABI) and made it SLOWER than the 8 x 32 kernel (C2: 8.6 k checks/s against 11.4 k).
template <int F, class FirstY, class Done> __device__ __forceinline__ bool xyzz29_add_affine(xyzz29_t &acc, bool &inf, const fe29_t &qx, const fe29_t &qy, FirstY first_y, Done operands_done, const fe_t &m32) {
    if (inf) { acc.x = qx; acc.y = first_y(); acc.zz = fe29_from_words(m32); acc.zzz = acc.zz; inf = false; operands_done(); return true; }   // 1 in the 2^261 domain = the integer 2^261 mod p
    const fe29_t pd = fe29_mul_hi_sg<F>(qx, acc.zz, fe29_kp_minus<F, EC29::SUB_X1_MULT>(acc.x)), r = fe29_mul_hi_sg<F>(qy, acc.zzz, fe29_kp_minus<F, EC29::SUB_Y1_MULT>(acc.y));   // u2 + K p - x1 < 13.1 p, s2 + K p - y1 < 5.1 p (signed quotient digits: fp29.cuh)
    operands_done();                                                                                                                           // qx, qy are dead from here
    if (__builtin_expect((pd.v[5] | pd.v[6] | pd.v[7] | (pd.v[8] & 0x3fffffu)) == 0u, 0))
        if (fe29_is_multiple_of_p<F>(pd)) return false;
    const fe29_t pp = fe29_sqr_sg<F>(pd), ppp = fe29_mul_sg<F>(pd, pp), q = fe29_mul_sg<F>(acc.x, pp);                                       // < 3.4 p, < 2.4 p, < 2.3 p
    const fe29_t x3 = fe29_sqr_hi_sg<F>(r, fe29_kp_minus_a_minus_2b<F, EC29::X3_SUB_MULT>(ppp, q));                                           // r^2 + K p - (ppp + 2 q) < 9.2 p, inside the square's reduction
    const fe29_t y3 = fe29_dot2_asm<F>(r, fe29_add_kp_minus<F, EC29::SUB_X3_MULT>(q, x3), fe29_kp_minus<F, EC29::SUB_Y1_MULT>(acc.y), ppp);   // r (q - x3) - y1 ppp, one reduction, the differences not normalised (the strict unsigned form: raw limbs up to 2^31 leave a signed column no room): < 1.6 p
    acc.zz = fe29_mul_sg<F>(acc.zz, pp); acc.zzz = fe29_mul_sg<F>(acc.zzz, ppp);                                                               // < 2.1 p, < 2.1 p
    acc.x = x3; acc.y = y3;
    return true;
}
// the 8-word twin table's entry (qx, py): `neg` adds (qx, -py); -py = 2 p - py enters its product in the raw limb form (9 subtractions and a select per limb;
// normalised -- p - py with the carry pass -- only where it becomes the accumulator itself).  TWO p: the raw form has no carries, so its top limb n_8 - 2 - py_8 must
// not go negative by itself, and a canonical py reaches 2^22 = the top limb of ONE p (py >= 2^254 - 2^233: one table point in 2^21)
template <int F> __device__ __forceinline__ bool xyzz29_add_affine(xyzz29_t &acc, bool &inf, const fe29_t &qx, const fe29_t &py, bool neg, const fe_t &m32) {
    fe29_t qy = fe29_kp_minus<F, EC29::NEG_Y_MULT>(py);
#pragma unroll
    for (int i = 0; i < L29; ++i) qy.v[i] = neg ? qy.v[i] : py.v[i];
    return xyzz29_add_affine<F>(acc, inf, qx, qy, [&]() { return neg ? fe29_sub_kp<F, 1>(fe29_zero(), py) : py; }, []() {}, m32);
}
// a += b for two accumulators on 29-bit limbs, NEITHER infinity (add-2008-s on XYZZ: 12 products + 2 squares; the callers handle infinity).  Both within the
// invariants of EC29, the sum within them again (fe29_bounds.py prove_group_add; multiples EC29::G_*).  Twelve of the thirteen reductions with signed quotient digits.
// Returns FALSE -- a untouched -- when the points are equal or opposite (P = 0 mod p, found exactly as in the mixed add): the caller recomputes its unit of work
// with the complete 8 x 32 law (msm.cuh: msm_segsum29_redo_kernel).
template <int F> __device__ __forceinline__ bool xyzz29_add(xyzz29_t &a, const xyzz29_t &b) {
    const fe29_t u1 = fe29_mul_sg<F>(a.x, b.zz), s1 = fe29_mul_sg<F>(a.y, b.zzz);                                                              // < 2.3 p, < 2.1 p
    const fe29_t pd = fe29_mul_hi_sg<F>(b.x, a.zz, fe29_kp_minus<F, EC29::G_U1_MULT>(u1)), r = fe29_mul_hi_sg<F>(b.y, a.zzz, fe29_kp_minus<F, EC29::G_S1_MULT>(s1));   // u2 + K p - u1 < 5.3 p, s2 + K p - s1 < 5.1 p
    if (__builtin_expect((pd.v[5] | pd.v[6] | pd.v[7] | (pd.v[8] & 0x3fffffu)) == 0u, 0))
        if (fe29_is_multiple_of_p<F>(pd)) return false;
    const fe29_t pp = fe29_sqr_sg<F>(pd), ppp = fe29_mul_sg<F>(pd, pp), q = fe29_mul_sg<F>(u1, pp);                                            // < 2.3 p, < 2.1 p, < 2.1 p
    const fe29_t x3 = fe29_sqr_hi_sg<F>(r, fe29_kp_minus_a_minus_2b<F, EC29::G_X3_SUB_MULT>(ppp, q));                                         // < 9.2 p
    const fe29_t y3 = fe29_dot2_asm<F>(r, fe29_add_kp_minus<F, EC29::G_SUB_X3_MULT>(q, x3), fe29_kp_minus<F, EC29::G_S1_MULT>(s1), ppp);       // r (q - x3) - s1 ppp < 1.6 p
    a.zz = fe29_mul_sg<F>(fe29_mul_sg<F>(a.zz, b.zz), pp); a.zzz = fe29_mul_sg<F>(fe29_mul_sg<F>(a.zzz, b.zzz), ppp);                          // < 2.1 p, < 2.1 p
    a.x = x3; a.y = y3;
    return true;
}
// the bucket value in the 8 x 32 form the rest of the MSM reads (canonical Montgomery-2^256 XYZZ; infinity = zz 0)
template <int F> __device__ __forceinline__ xyzz_t xyzz29_leave(const xyzz29_t &acc, bool inf, const fe_t &one) {
    if (inf) return xyzz_inf();
    const fe29_t leave = fe29_from_words(one);
    xyzz_t o; o.x = fe29_leave<F>(acc.x, leave); o.y = fe29_leave<F>(acc.y, leave); o.zz = fe29_leave<F>(acc.zz, leave); o.zzz = fe29_leave<F>(acc.zzz, leave);
    return o;
}
#endif

}  // namespace mb
