// fp29.cuh -- the Poseidon rounds' own field representation: 9 limbs of 29 bits in 32-bit registers, Montgomery with R = 2^261.
//
// Why a second representation.  In the saturated 8 x 32-bit form (fp.cuh) every limb product is TWO VALU instructions: `v_mad_u64_u32` into
// a 64-bit accumulator plus a `v_addc` that collects its carry-out, because sixteen 64-bit products do not fit 64 bits -- and the VALU port
// is what the sponge kernels saturate (profiles/r02f: 86 % of a step's instructions are sponges, issue port ~90 % busy).  With 29-bit limbs
// a column of <= 27 products of < 2^58 each plus 5 reduction terms stays below 2^64: the 64-bit accumulator of `v_mad_u64_u32` never
// overflows, NO carry instruction exists, and a column's carry into the next one is a shift.  Per product: 81 limb products + 54 reduction
// products (p = 2^254 + c has 29-bit limbs 0..4 and limb 8 = 2^22; p_0 = 1) and ~43 shifts / masks; a squaring needs 45 limb products
// (cross terms once, against doubled limbs); no conditional subtraction anywhere (t < a b / 2^261 + p < 1.13 p for operands below 2^256).
// A round of the 3-lane permutation: 765 multiply-accumulates + ~290 simple instructions, against 608 + 608 carry adds + ~350 in the
// 8 x 32 form -- measured: v_mad_u64_u32 issues at ~6.5 cycles per wave, simple VALU at 4 (10 limbs of 28 bits were tried first: 910
// multiply-accumulates per round, only -5 %).
//
// LAZY forms (fe29_sqr_lz / fe29_mul_lz / fe29_dot3rc_lz, round 4): the quotient digit of column k is m_k = -col mod 2^32, NOT masked to 29 bits.
// Its three extra bits add h 2^29 p 2^(29 k) to the sum -- a multiple of p, and the low limb still cancels -- so the result is the same field
// element, only larger: < (sum of products) / 2^261 + 8.0001 p.  Bounds: operands < 10 p (limbs 0..7 < 2^29, limb 8 < 2^26): a column holds at most
// 27 products of < 2^58, four quotient terms m p_j < 2^61, m 2^22, m and the carry: < 0.93 x 2^64.  One mask per quotient digit saved (45 per
// round), the accumulator starts from its first product, and the dot product adds a tenth operand before reducing (the round constant times
// 2^261: 9 multiply-accumulates by 1 instead of a 27-instruction normalised addition).  The STRICT forms stay for everything whose result must be
// below 1.13 p (the way out of the permutation, the group law of ec29.cuh).
//
// SIGNED-DIGIT forms (fe29_mul_sg / fe29_sqr_sg / fe29_mul_hi_sg / fe29_sqr_hi_sg / fe29_mulrc_sg / fe29_dot2rc_sg / fe29_dot3rc_sg, late round 5) -- what the Poseidon rounds
// and the group law of ec29.cuh run on now.  The quotient digit of column k < 8 is the column's own low word READ AS AN int32 and SUBTRACTED (`v_mad_i64_i32` against the
// negated prime limbs, which ride the constant bus): no instruction makes the digit -- the statement that cancels the low limb writes the new column to other registers, so
// the old low word simply stays where it is for the digit's five later uses.  Digit 8 is (col & M29) - 2^30 (one v_and_or): its sign is fixed, so the quotient
// M = - sum s_k 2^(29 k) lies in ((1 - 2^-27) R, (2 + 2^-27) R) and the result is T / R + (1 p, 2 p] -- positive with no offset term, limbs 0..7 normalised, and TIGHTER
// than the lazy forms' + 8 p (the invariants of every caller shrank: states below 2.1 p, the accumulator's x below 10 p).  The accumulator is a two's-complement 64-bit
// value; tools/fe29_bounds.py `product_signed` proves the true column sum inside [-2^63, 2^63) for every caller (worst: the MDS row's 27 products per column, 0.92 of
// it); what does not fit -- the group law's two-term dot product on raw operands -- keeps the strict unsigned form.  Per Poseidon lane-round 925 VALU instructions
// (965 with the lazy forms), per mixed add 1673 (1745); measured on one dependent chain x <- x^3: + 4.6 % at 5 - 8 waves per SIMD, + 8.5 % at 2 (tools/probes/sg_probe.hip).
//
// Only the wave-packed 3-lane permutation uses it (sponge.cuh): state enters as 8 x 32 Montgomery-2^256, is re-based with one product by
// 2^266 mod p (x 2^256 -> x 2^261), runs its 55 rounds here, leaves with one product by 2^256 mod p and one conditional subtraction.
// Bit-identical results (the value computed is the same field element): every sponge parity test runs through it.
#pragma once
#include "fp.cuh"

namespace mb {

static constexpr int L29 = 9;
struct fe29_t { uint32_t v[L29]; };
static constexpr uint32_t M29 = 0x1FFFFFFFu;

// 29-bit limb i of the 256-bit little-endian word array w[8]
MB_HD constexpr uint32_t limb29_of(const uint32_t w0, const uint32_t w1, const uint32_t w2, const uint32_t w3, const uint32_t w4, const uint32_t w5, const uint32_t w6,
                                   const uint32_t w7, int i) {
    const uint32_t w[9] = {w0, w1, w2, w3, w4, w5, w6, w7, 0u};
    const int bit = 29 * i, q = bit >> 5, r = bit & 31;
    if (q >= 8) return 0u;
    const uint64_t two = (uint64_t)w[q] | ((uint64_t)w[q + 1] << 32);
    return (uint32_t)(two >> r) & M29;
}
template <int F> struct P29 {          // p = 2^254 + c, c < 2^125: limbs 0..4 carry c (limb 0 = 1), limbs 5..7 are zero, limb 8 = 2^(254 - 232) = 2^22
#define MB_P29(i) limb29_of(1u, FieldP<F>::P1, FieldP<F>::P2, FieldP<F>::P3, 0, 0, 0, P7, i)
    static constexpr uint32_t L1 = MB_P29(1), L2 = MB_P29(2), L3 = MB_P29(3), L4 = MB_P29(4), L8 = MB_P29(8);
    static_assert(MB_P29(0) == 1u && MB_P29(8) == (1u << 22) && MB_P29(5) == 0u && MB_P29(6) == 0u && MB_P29(7) == 0u, "Pasta prime shape");
#undef MB_P29
};

MB_HD fe29_t fe29_from_words(const fe_t &a) {            // 8 x 32 -> 9 x 29 (the integer is unchanged)
    fe29_t r;
#pragma unroll
    for (int i = 0; i < L29; ++i) r.v[i] = limb29_of(a.v[0], a.v[1], a.v[2], a.v[3], a.v[4], a.v[5], a.v[6], a.v[7], i);
    return r;
}
// 9 x 29 (limbs below 2^29, the integer below 2^256) -> 8 x 32
MB_HD fe_t fe29_to_words(const fe29_t &a) {
    fe_t r = fe_zero();
#pragma unroll
    for (int i = 0; i < L29; ++i) {
        const int q = (29 * i) >> 5, s = (29 * i) & 31;
        r.v[q] |= a.v[i] << s;
        if (s > 3 && q + 1 < 8) r.v[q + 1] |= a.v[i] >> (32 - s);
    }
    return r;
}
// limb-wise sum with the carries propagated (both operands normalised: the result is again)
MB_HD fe29_t fe29_add(const fe29_t &a, const fe29_t &b) {
    fe29_t r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L29; ++i) { const uint32_t t = a.v[i] + b.v[i] + c; if (i < L29 - 1) { r.v[i] = t & M29; c = t >> 29; } else r.v[i] = t; }
    return r;
}

// a + b + c with the carries propagated (limbs 0..7 of the operands below 2^29): one pass of 4 instructions per limb where two fe29_add take 6
MB_HD fe29_t fe29_add3(const fe29_t &a, const fe29_t &b, const fe29_t &c) {
    fe29_t r; uint32_t k = 0;
#pragma unroll
    for (int i = 0; i < L29; ++i) { const uint32_t t = a.v[i] + b.v[i] + c.v[i] + k; if (i < L29 - 1) { r.v[i] = t & M29; k = t >> 29; } else r.v[i] = t; }
    return r;
}

// ---- PROVEN CONSTANTS (tools/gen_fe29.py <- tools/fe29_bounds.py): do not edit by hand
// proven by tools/fe29_bounds.py (interval model of every routine and of the callers' value discipline; gen_fe29.py refuses to write this file otherwise):
//   group law: worst column 0.697 x 2^64; Poseidon lane forms: worst column 0.457 x 2^64
struct EC29 {      // xyzz29_add_affine: accumulator invariants (units of p) and the multiple of p in every limb-wise "K p - b" (no limb may go negative)
    static constexpr uint32_t INV_X = 10;
    static constexpr uint32_t INV_Y = 2;
    static constexpr uint32_t INV_ZZ = 3;
    static constexpr uint32_t INV_ZZZ = 3;
    static constexpr uint32_t NEG_Y_MULT = 2;
    static constexpr uint32_t SUB_X1_MULT = 11;
    static constexpr uint32_t SUB_Y1_MULT = 3;
    static constexpr uint32_t X3_SUB_MULT = 7;
    static constexpr uint32_t SUB_X3_MULT = 11;
    static constexpr uint32_t PD_MAX = 14;
    // xyzz29_add (two accumulators): the multiples under u1 = x1 zz2, s1 = y1 zzz2, ppp + 2 q and x3
    static constexpr uint32_t G_U1_MULT = 3;
    static constexpr uint32_t G_S1_MULT = 3;
    static constexpr uint32_t G_X3_SUB_MULT = 7;
    static constexpr uint32_t G_SUB_X3_MULT = 10;
};
struct SPONGE29 {   // the Poseidon lane forms' state bounds between rounds, in thousandths of p (fixed points of a lazy round)
    static constexpr uint32_t LANES3_STATE_MILLI_P = 4100;
    static constexpr uint32_t LANES8_STATE_MILLI_P = 4100;
    static constexpr uint32_t LANES16_STATE_MILLI_P = 6100;
};
// ---- END PROVEN CONSTANTS

#if defined(__HIP_DEVICE_COMPILE__)
// Montgomery products, R = 2^261, operands normalised (limbs < 2^29), results normalised; every multiply-accumulate of a column pinned
// in one asm statement (left to itself the compiler spreads a column over several accumulators and re-adds them).
// ---- GENERATED by tools/gen_fe29.py: do not edit by hand
template <int F> __device__ __forceinline__ fe29_t fe29_mul_asm(const fe29_t &a, const fe29_t &b) {
    uint64_t col, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;
    // column 0: 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[0]));
    m0 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m0));
    col >>= 29;
    // column 1: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[1]), "v"(b.v[0]), "v"(m0), "v"(p1));
    m1 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1));
    col >>= 29;
    // column 2: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[0]), "v"(m1), "v"(p1), "v"(m0), "v"(p2));
    m2 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2));
    col >>= 29;
    // column 3: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[2]), "v"(b.v[1]), "v"(a.v[3]), "v"(b.v[0]), "v"(m2), "v"(p1), "v"(m1), "v"(p2), "v"(m0), "v"(p3));
    m3 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3));
    col >>= 29;
    // column 4: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[3]), "v"(b.v[1]), "v"(a.v[4]), "v"(b.v[0]), "v"(m3), "v"(p1), "v"(m2), "v"(p2), "v"(m1), "v"(p3), "v"(m0), "v"(p4));
    m4 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4));
    col >>= 29;
    // column 5: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[0]), "v"(m4), "v"(p1), "v"(m3), "v"(p2), "v"(m2), "v"(p3), "v"(m1), "v"(p4));
    m5 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5));
    col >>= 29;
    // column 6: 11 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[0]), "v"(m5), "v"(p1), "v"(m4), "v"(p2), "v"(m3), "v"(p3), "v"(m2), "v"(p4));
    m6 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6));
    col >>= 29;
    // column 7: 12 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0]), "v"(m6), "v"(p1), "v"(m5), "v"(p2), "v"(m4), "v"(p3), "v"(m3), "v"(p4));
    m7 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m7));
    col >>= 29;
    // column 8: 14 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[0]), "v"(m7), "v"(p1), "v"(m6), "v"(p2), "v"(m5), "v"(p3));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4), "v"(p4), "v"(m0), "v"(p8));
    m8 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8));
    col >>= 29;
    // column 9: 13 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[1]), "v"(b.v[8]), "v"(a.v[2]), "v"(b.v[7]), "v"(a.v[3]), "v"(b.v[6]), "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[2]), "v"(a.v[8]), "v"(b.v[1]), "v"(m8), "v"(p1), "v"(m7), "v"(p2), "v"(m6), "v"(p3), "v"(m5), "v"(p4));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1), "v"(p8));
    r.v[0] = (uint32_t)col & M29; col >>= 29;
    // column 10: 11 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[2]), "v"(b.v[8]), "v"(a.v[3]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[5]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[2]), "v"(m8), "v"(p2), "v"(m7), "v"(p3), "v"(m6), "v"(p4), "v"(m2), "v"(p8));
    r.v[1] = (uint32_t)col & M29; col >>= 29;
    // column 11: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[3]), "v"(b.v[8]), "v"(a.v[4]), "v"(b.v[7]), "v"(a.v[5]), "v"(b.v[6]), "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[7]), "v"(b.v[4]), "v"(a.v[8]), "v"(b.v[3]), "v"(m8), "v"(p3), "v"(m7), "v"(p4), "v"(m3), "v"(p8));
    r.v[2] = (uint32_t)col & M29; col >>= 29;
    // column 12: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[4]), "v"(b.v[8]), "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[4]), "v"(m8), "v"(p4), "v"(m4), "v"(p8));
    r.v[3] = (uint32_t)col & M29; col >>= 29;
    // column 13: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[5]), "v"(b.v[8]), "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[7]), "v"(b.v[6]), "v"(a.v[8]), "v"(b.v[5]), "v"(m5), "v"(p8));
    r.v[4] = (uint32_t)col & M29; col >>= 29;
    // column 14: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[6]), "v"(b.v[8]), "v"(a.v[7]), "v"(b.v[7]), "v"(a.v[8]), "v"(b.v[6]), "v"(m6), "v"(p8));
    r.v[5] = (uint32_t)col & M29; col >>= 29;
    // column 15: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[7]), "v"(b.v[8]), "v"(a.v[8]), "v"(b.v[7]), "v"(m7), "v"(p8));
    r.v[6] = (uint32_t)col & M29; col >>= 29;
    // column 16: 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(b.v[8]), "v"(m8), "v"(p8));
    r.v[7] = (uint32_t)col & M29; col >>= 29;
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_sqr_asm(const fe29_t &a) {
    const uint32_t d0 = a.v[0] << 1, d1 = a.v[1] << 1, d2 = a.v[2] << 1, d3 = a.v[3] << 1, d4 = a.v[4] << 1, d5 = a.v[5] << 1, d6 = a.v[6] << 1, d7 = a.v[7] << 1;   // limbs < 2^29: the doubled ones fit 32 bits
    uint64_t col, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;
    // column 0: 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(a.v[0]));
    m0 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m0));
    col >>= 29;
    // column 1: 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[1]), "v"(m0), "v"(p1));
    m1 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1));
    col >>= 29;
    // column 2: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[2]), "v"(a.v[1]), "v"(a.v[1]), "v"(m1), "v"(p1), "v"(m0), "v"(p2));
    m2 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2));
    col >>= 29;
    // column 3: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[3]), "v"(d1), "v"(a.v[2]), "v"(m2), "v"(p1), "v"(m1), "v"(p2), "v"(m0), "v"(p3));
    m3 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3));
    col >>= 29;
    // column 4: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[4]), "v"(d1), "v"(a.v[3]), "v"(a.v[2]), "v"(a.v[2]), "v"(m3), "v"(p1), "v"(m2), "v"(p2), "v"(m1), "v"(p3), "v"(m0), "v"(p4));
    m4 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4));
    col >>= 29;
    // column 5: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[5]), "v"(d1), "v"(a.v[4]), "v"(d2), "v"(a.v[3]), "v"(m4), "v"(p1), "v"(m3), "v"(p2), "v"(m2), "v"(p3), "v"(m1), "v"(p4));
    m5 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5));
    col >>= 29;
    // column 6: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[6]), "v"(d1), "v"(a.v[5]), "v"(d2), "v"(a.v[4]), "v"(a.v[3]), "v"(a.v[3]), "v"(m5), "v"(p1), "v"(m4), "v"(p2), "v"(m3), "v"(p3), "v"(m2), "v"(p4));
    m6 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6));
    col >>= 29;
    // column 7: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[7]), "v"(d1), "v"(a.v[6]), "v"(d2), "v"(a.v[5]), "v"(d3), "v"(a.v[4]), "v"(m6), "v"(p1), "v"(m5), "v"(p2), "v"(m4), "v"(p3), "v"(m3), "v"(p4));
    m7 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m7));
    col >>= 29;
    // column 8: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[8]), "v"(d1), "v"(a.v[7]), "v"(d2), "v"(a.v[6]), "v"(d3), "v"(a.v[5]), "v"(a.v[4]), "v"(a.v[4]), "v"(m7), "v"(p1), "v"(m6), "v"(p2), "v"(m5), "v"(p3), "v"(m4), "v"(p4), "v"(m0), "v"(p8));
    m8 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8));
    col >>= 29;
    // column 9: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d1), "v"(a.v[8]), "v"(d2), "v"(a.v[7]), "v"(d3), "v"(a.v[6]), "v"(d4), "v"(a.v[5]), "v"(m8), "v"(p1), "v"(m7), "v"(p2), "v"(m6), "v"(p3), "v"(m5), "v"(p4), "v"(m1), "v"(p8));
    r.v[0] = (uint32_t)col & M29; col >>= 29;
    // column 10: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d2), "v"(a.v[8]), "v"(d3), "v"(a.v[7]), "v"(d4), "v"(a.v[6]), "v"(a.v[5]), "v"(a.v[5]), "v"(m8), "v"(p2), "v"(m7), "v"(p3), "v"(m6), "v"(p4), "v"(m2), "v"(p8));
    r.v[1] = (uint32_t)col & M29; col >>= 29;
    // column 11: 6 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d3), "v"(a.v[8]), "v"(d4), "v"(a.v[7]), "v"(d5), "v"(a.v[6]), "v"(m8), "v"(p3), "v"(m7), "v"(p4), "v"(m3), "v"(p8));
    r.v[2] = (uint32_t)col & M29; col >>= 29;
    // column 12: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d4), "v"(a.v[8]), "v"(d5), "v"(a.v[7]), "v"(a.v[6]), "v"(a.v[6]), "v"(m8), "v"(p4), "v"(m4), "v"(p8));
    r.v[3] = (uint32_t)col & M29; col >>= 29;
    // column 13: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d5), "v"(a.v[8]), "v"(d6), "v"(a.v[7]), "v"(m5), "v"(p8));
    r.v[4] = (uint32_t)col & M29; col >>= 29;
    // column 14: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d6), "v"(a.v[8]), "v"(a.v[7]), "v"(a.v[7]), "v"(m6), "v"(p8));
    r.v[5] = (uint32_t)col & M29; col >>= 29;
    // column 15: 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d7), "v"(a.v[8]), "v"(m7), "v"(p8));
    r.v[6] = (uint32_t)col & M29; col >>= 29;
    // column 16: 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(a.v[8]), "v"(m8), "v"(p8));
    r.v[7] = (uint32_t)col & M29; col >>= 29;
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_dot2_asm(const fe29_t &a0, const fe29_t &b0, const fe29_t &a1, const fe29_t &b1) {
    uint64_t col, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;
    // column 0: 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "=&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[0]));
    m0 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m0));
    col >>= 29;
    // column 1: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[1]), "v"(a0.v[1]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[1]), "v"(a1.v[1]), "v"(b1.v[0]), "v"(m0), "v"(p1));
    m1 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1));
    col >>= 29;
    // column 2: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[2]), "v"(a0.v[1]), "v"(b0.v[1]), "v"(a0.v[2]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[2]), "v"(a1.v[1]), "v"(b1.v[1]), "v"(a1.v[2]), "v"(b1.v[0]), "v"(m1), "v"(p1), "v"(m0), "v"(p2));
    m2 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2));
    col >>= 29;
    // column 3: 11 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[3]), "v"(a0.v[1]), "v"(b0.v[2]), "v"(a0.v[2]), "v"(b0.v[1]), "v"(a0.v[3]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[3]), "v"(a1.v[1]), "v"(b1.v[2]), "v"(a1.v[2]), "v"(b1.v[1]), "v"(a1.v[3]), "v"(b1.v[0]), "v"(m2), "v"(p1), "v"(m1), "v"(p2), "v"(m0), "v"(p3));
    m3 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3));
    col >>= 29;
    // column 4: 14 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[4]), "v"(a0.v[1]), "v"(b0.v[3]), "v"(a0.v[2]), "v"(b0.v[2]), "v"(a0.v[3]), "v"(b0.v[1]), "v"(a0.v[4]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[4]), "v"(a1.v[1]), "v"(b1.v[3]), "v"(a1.v[2]), "v"(b1.v[2]), "v"(a1.v[3]), "v"(b1.v[1]), "v"(a1.v[4]), "v"(b1.v[0]), "v"(m3), "v"(p1), "v"(m2), "v"(p2));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1), "v"(p3), "v"(m0), "v"(p4));
    m4 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4));
    col >>= 29;
    // column 5: 16 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[5]), "v"(a0.v[1]), "v"(b0.v[4]), "v"(a0.v[2]), "v"(b0.v[3]), "v"(a0.v[3]), "v"(b0.v[2]), "v"(a0.v[4]), "v"(b0.v[1]), "v"(a0.v[5]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[5]), "v"(a1.v[1]), "v"(b1.v[4]), "v"(a1.v[2]), "v"(b1.v[3]), "v"(a1.v[3]), "v"(b1.v[2]), "v"(a1.v[4]), "v"(b1.v[1]), "v"(a1.v[5]), "v"(b1.v[0]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4), "v"(p1), "v"(m3), "v"(p2), "v"(m2), "v"(p3), "v"(m1), "v"(p4));
    m5 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5));
    col >>= 29;
    // column 6: 18 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[6]), "v"(a0.v[1]), "v"(b0.v[5]), "v"(a0.v[2]), "v"(b0.v[4]), "v"(a0.v[3]), "v"(b0.v[3]), "v"(a0.v[4]), "v"(b0.v[2]), "v"(a0.v[5]), "v"(b0.v[1]), "v"(a0.v[6]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[6]), "v"(a1.v[1]), "v"(b1.v[5]), "v"(a1.v[2]), "v"(b1.v[4]), "v"(a1.v[3]), "v"(b1.v[3]), "v"(a1.v[4]), "v"(b1.v[2]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[5]), "v"(b1.v[1]), "v"(a1.v[6]), "v"(b1.v[0]), "v"(m5), "v"(p1), "v"(m4), "v"(p2), "v"(m3), "v"(p3), "v"(m2), "v"(p4));
    m6 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6));
    col >>= 29;
    // column 7: 20 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[7]), "v"(a0.v[1]), "v"(b0.v[6]), "v"(a0.v[2]), "v"(b0.v[5]), "v"(a0.v[3]), "v"(b0.v[4]), "v"(a0.v[4]), "v"(b0.v[3]), "v"(a0.v[5]), "v"(b0.v[2]), "v"(a0.v[6]), "v"(b0.v[1]), "v"(a0.v[7]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[7]), "v"(a1.v[1]), "v"(b1.v[6]), "v"(a1.v[2]), "v"(b1.v[5]), "v"(a1.v[3]), "v"(b1.v[4]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[4]), "v"(b1.v[3]), "v"(a1.v[5]), "v"(b1.v[2]), "v"(a1.v[6]), "v"(b1.v[1]), "v"(a1.v[7]), "v"(b1.v[0]), "v"(m6), "v"(p1), "v"(m5), "v"(p2), "v"(m4), "v"(p3), "v"(m3), "v"(p4));
    m7 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m7));
    col >>= 29;
    // column 8: 23 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[8]), "v"(a0.v[1]), "v"(b0.v[7]), "v"(a0.v[2]), "v"(b0.v[6]), "v"(a0.v[3]), "v"(b0.v[5]), "v"(a0.v[4]), "v"(b0.v[4]), "v"(a0.v[5]), "v"(b0.v[3]), "v"(a0.v[6]), "v"(b0.v[2]), "v"(a0.v[7]), "v"(b0.v[1]), "v"(a0.v[8]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[8]), "v"(a1.v[1]), "v"(b1.v[7]), "v"(a1.v[2]), "v"(b1.v[6]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[3]), "v"(b1.v[5]), "v"(a1.v[4]), "v"(b1.v[4]), "v"(a1.v[5]), "v"(b1.v[3]), "v"(a1.v[6]), "v"(b1.v[2]), "v"(a1.v[7]), "v"(b1.v[1]), "v"(a1.v[8]), "v"(b1.v[0]), "v"(m7), "v"(p1), "v"(m6), "v"(p2), "v"(m5), "v"(p3), "v"(m4), "v"(p4), "v"(m0), "v"(p8));
    m8 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8));
    col >>= 29;
    // column 9: 21 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[1]), "v"(b0.v[8]), "v"(a0.v[2]), "v"(b0.v[7]), "v"(a0.v[3]), "v"(b0.v[6]), "v"(a0.v[4]), "v"(b0.v[5]), "v"(a0.v[5]), "v"(b0.v[4]), "v"(a0.v[6]), "v"(b0.v[3]), "v"(a0.v[7]), "v"(b0.v[2]), "v"(a0.v[8]), "v"(b0.v[1]), "v"(a1.v[1]), "v"(b1.v[8]), "v"(a1.v[2]), "v"(b1.v[7]), "v"(a1.v[3]), "v"(b1.v[6]), "v"(a1.v[4]), "v"(b1.v[5]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[5]), "v"(b1.v[4]), "v"(a1.v[6]), "v"(b1.v[3]), "v"(a1.v[7]), "v"(b1.v[2]), "v"(a1.v[8]), "v"(b1.v[1]), "v"(m8), "v"(p1), "v"(m7), "v"(p2), "v"(m6), "v"(p3), "v"(m5), "v"(p4), "v"(m1), "v"(p8));
    r.v[0] = (uint32_t)col & M29; col >>= 29;
    // column 10: 18 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[2]), "v"(b0.v[8]), "v"(a0.v[3]), "v"(b0.v[7]), "v"(a0.v[4]), "v"(b0.v[6]), "v"(a0.v[5]), "v"(b0.v[5]), "v"(a0.v[6]), "v"(b0.v[4]), "v"(a0.v[7]), "v"(b0.v[3]), "v"(a0.v[8]), "v"(b0.v[2]), "v"(a1.v[2]), "v"(b1.v[8]), "v"(a1.v[3]), "v"(b1.v[7]), "v"(a1.v[4]), "v"(b1.v[6]), "v"(a1.v[5]), "v"(b1.v[5]), "v"(a1.v[6]), "v"(b1.v[4]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[7]), "v"(b1.v[3]), "v"(a1.v[8]), "v"(b1.v[2]), "v"(m8), "v"(p2), "v"(m7), "v"(p3), "v"(m6), "v"(p4), "v"(m2), "v"(p8));
    r.v[1] = (uint32_t)col & M29; col >>= 29;
    // column 11: 15 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[3]), "v"(b0.v[8]), "v"(a0.v[4]), "v"(b0.v[7]), "v"(a0.v[5]), "v"(b0.v[6]), "v"(a0.v[6]), "v"(b0.v[5]), "v"(a0.v[7]), "v"(b0.v[4]), "v"(a0.v[8]), "v"(b0.v[3]), "v"(a1.v[3]), "v"(b1.v[8]), "v"(a1.v[4]), "v"(b1.v[7]), "v"(a1.v[5]), "v"(b1.v[6]), "v"(a1.v[6]), "v"(b1.v[5]), "v"(a1.v[7]), "v"(b1.v[4]), "v"(a1.v[8]), "v"(b1.v[3]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8), "v"(p3), "v"(m7), "v"(p4), "v"(m3), "v"(p8));
    r.v[2] = (uint32_t)col & M29; col >>= 29;
    // column 12: 12 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[4]), "v"(b0.v[8]), "v"(a0.v[5]), "v"(b0.v[7]), "v"(a0.v[6]), "v"(b0.v[6]), "v"(a0.v[7]), "v"(b0.v[5]), "v"(a0.v[8]), "v"(b0.v[4]), "v"(a1.v[4]), "v"(b1.v[8]), "v"(a1.v[5]), "v"(b1.v[7]), "v"(a1.v[6]), "v"(b1.v[6]), "v"(a1.v[7]), "v"(b1.v[5]), "v"(a1.v[8]), "v"(b1.v[4]), "v"(m8), "v"(p4), "v"(m4), "v"(p8));
    r.v[3] = (uint32_t)col & M29; col >>= 29;
    // column 13: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[5]), "v"(b0.v[8]), "v"(a0.v[6]), "v"(b0.v[7]), "v"(a0.v[7]), "v"(b0.v[6]), "v"(a0.v[8]), "v"(b0.v[5]), "v"(a1.v[5]), "v"(b1.v[8]), "v"(a1.v[6]), "v"(b1.v[7]), "v"(a1.v[7]), "v"(b1.v[6]), "v"(a1.v[8]), "v"(b1.v[5]), "v"(m5), "v"(p8));
    r.v[4] = (uint32_t)col & M29; col >>= 29;
    // column 14: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[6]), "v"(b0.v[8]), "v"(a0.v[7]), "v"(b0.v[7]), "v"(a0.v[8]), "v"(b0.v[6]), "v"(a1.v[6]), "v"(b1.v[8]), "v"(a1.v[7]), "v"(b1.v[7]), "v"(a1.v[8]), "v"(b1.v[6]), "v"(m6), "v"(p8));
    r.v[5] = (uint32_t)col & M29; col >>= 29;
    // column 15: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[7]), "v"(b0.v[8]), "v"(a0.v[8]), "v"(b0.v[7]), "v"(a1.v[7]), "v"(b1.v[8]), "v"(a1.v[8]), "v"(b1.v[7]), "v"(m7), "v"(p8));
    r.v[6] = (uint32_t)col & M29; col >>= 29;
    // column 16: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[8]), "v"(b0.v[8]), "v"(a1.v[8]), "v"(b1.v[8]), "v"(m8), "v"(p8));
    r.v[7] = (uint32_t)col & M29; col >>= 29;
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_dot3_asm(const fe29_t &a0, const fe29_t &b0, const fe29_t &a1, const fe29_t &b1, const fe29_t &a2, const fe29_t &b2) {
    uint64_t col, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;
    // column 0: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "=&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[0]));
    m0 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m0));
    col >>= 29;
    // column 1: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[1]), "v"(a0.v[1]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[1]), "v"(a1.v[1]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[1]), "v"(a2.v[1]), "v"(b2.v[0]), "v"(m0), "v"(p1));
    m1 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1));
    col >>= 29;
    // column 2: 11 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[2]), "v"(a0.v[1]), "v"(b0.v[1]), "v"(a0.v[2]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[2]), "v"(a1.v[1]), "v"(b1.v[1]), "v"(a1.v[2]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[2]), "v"(a2.v[1]), "v"(b2.v[1]), "v"(a2.v[2]), "v"(b2.v[0]), "v"(m1), "v"(p1), "v"(m0), "v"(p2));
    m2 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2));
    col >>= 29;
    // column 3: 15 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[3]), "v"(a0.v[1]), "v"(b0.v[2]), "v"(a0.v[2]), "v"(b0.v[1]), "v"(a0.v[3]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[3]), "v"(a1.v[1]), "v"(b1.v[2]), "v"(a1.v[2]), "v"(b1.v[1]), "v"(a1.v[3]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[3]), "v"(a2.v[1]), "v"(b2.v[2]), "v"(a2.v[2]), "v"(b2.v[1]), "v"(a2.v[3]), "v"(b2.v[0]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2), "v"(p1), "v"(m1), "v"(p2), "v"(m0), "v"(p3));
    m3 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3));
    col >>= 29;
    // column 4: 19 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[4]), "v"(a0.v[1]), "v"(b0.v[3]), "v"(a0.v[2]), "v"(b0.v[2]), "v"(a0.v[3]), "v"(b0.v[1]), "v"(a0.v[4]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[4]), "v"(a1.v[1]), "v"(b1.v[3]), "v"(a1.v[2]), "v"(b1.v[2]), "v"(a1.v[3]), "v"(b1.v[1]), "v"(a1.v[4]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[4]), "v"(a2.v[1]), "v"(b2.v[3]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[2]), "v"(b2.v[2]), "v"(a2.v[3]), "v"(b2.v[1]), "v"(a2.v[4]), "v"(b2.v[0]), "v"(m3), "v"(p1), "v"(m2), "v"(p2), "v"(m1), "v"(p3), "v"(m0), "v"(p4));
    m4 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4));
    col >>= 29;
    // column 5: 22 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[5]), "v"(a0.v[1]), "v"(b0.v[4]), "v"(a0.v[2]), "v"(b0.v[3]), "v"(a0.v[3]), "v"(b0.v[2]), "v"(a0.v[4]), "v"(b0.v[1]), "v"(a0.v[5]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[5]), "v"(a1.v[1]), "v"(b1.v[4]), "v"(a1.v[2]), "v"(b1.v[3]), "v"(a1.v[3]), "v"(b1.v[2]), "v"(a1.v[4]), "v"(b1.v[1]), "v"(a1.v[5]), "v"(b1.v[0]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[0]), "v"(b2.v[5]), "v"(a2.v[1]), "v"(b2.v[4]), "v"(a2.v[2]), "v"(b2.v[3]), "v"(a2.v[3]), "v"(b2.v[2]), "v"(a2.v[4]), "v"(b2.v[1]), "v"(a2.v[5]), "v"(b2.v[0]), "v"(m4), "v"(p1), "v"(m3), "v"(p2), "v"(m2), "v"(p3), "v"(m1), "v"(p4));
    m5 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5));
    col >>= 29;
    // column 6: 25 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[6]), "v"(a0.v[1]), "v"(b0.v[5]), "v"(a0.v[2]), "v"(b0.v[4]), "v"(a0.v[3]), "v"(b0.v[3]), "v"(a0.v[4]), "v"(b0.v[2]), "v"(a0.v[5]), "v"(b0.v[1]), "v"(a0.v[6]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[6]), "v"(a1.v[1]), "v"(b1.v[5]), "v"(a1.v[2]), "v"(b1.v[4]), "v"(a1.v[3]), "v"(b1.v[3]), "v"(a1.v[4]), "v"(b1.v[2]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[5]), "v"(b1.v[1]), "v"(a1.v[6]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[6]), "v"(a2.v[1]), "v"(b2.v[5]), "v"(a2.v[2]), "v"(b2.v[4]), "v"(a2.v[3]), "v"(b2.v[3]), "v"(a2.v[4]), "v"(b2.v[2]), "v"(a2.v[5]), "v"(b2.v[1]), "v"(a2.v[6]), "v"(b2.v[0]), "v"(m5), "v"(p1), "v"(m4), "v"(p2), "v"(m3), "v"(p3));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2), "v"(p4));
    m6 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6));
    col >>= 29;
    // column 7: 28 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[7]), "v"(a0.v[1]), "v"(b0.v[6]), "v"(a0.v[2]), "v"(b0.v[5]), "v"(a0.v[3]), "v"(b0.v[4]), "v"(a0.v[4]), "v"(b0.v[3]), "v"(a0.v[5]), "v"(b0.v[2]), "v"(a0.v[6]), "v"(b0.v[1]), "v"(a0.v[7]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[7]), "v"(a1.v[1]), "v"(b1.v[6]), "v"(a1.v[2]), "v"(b1.v[5]), "v"(a1.v[3]), "v"(b1.v[4]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[4]), "v"(b1.v[3]), "v"(a1.v[5]), "v"(b1.v[2]), "v"(a1.v[6]), "v"(b1.v[1]), "v"(a1.v[7]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[7]), "v"(a2.v[1]), "v"(b2.v[6]), "v"(a2.v[2]), "v"(b2.v[5]), "v"(a2.v[3]), "v"(b2.v[4]), "v"(a2.v[4]), "v"(b2.v[3]), "v"(a2.v[5]), "v"(b2.v[2]), "v"(a2.v[6]), "v"(b2.v[1]), "v"(a2.v[7]), "v"(b2.v[0]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6), "v"(p1), "v"(m5), "v"(p2), "v"(m4), "v"(p3), "v"(m3), "v"(p4));
    m7 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m7));
    col >>= 29;
    // column 8: 32 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[8]), "v"(a0.v[1]), "v"(b0.v[7]), "v"(a0.v[2]), "v"(b0.v[6]), "v"(a0.v[3]), "v"(b0.v[5]), "v"(a0.v[4]), "v"(b0.v[4]), "v"(a0.v[5]), "v"(b0.v[3]), "v"(a0.v[6]), "v"(b0.v[2]), "v"(a0.v[7]), "v"(b0.v[1]), "v"(a0.v[8]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[8]), "v"(a1.v[1]), "v"(b1.v[7]), "v"(a1.v[2]), "v"(b1.v[6]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[3]), "v"(b1.v[5]), "v"(a1.v[4]), "v"(b1.v[4]), "v"(a1.v[5]), "v"(b1.v[3]), "v"(a1.v[6]), "v"(b1.v[2]), "v"(a1.v[7]), "v"(b1.v[1]), "v"(a1.v[8]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[8]), "v"(a2.v[1]), "v"(b2.v[7]), "v"(a2.v[2]), "v"(b2.v[6]), "v"(a2.v[3]), "v"(b2.v[5]), "v"(a2.v[4]), "v"(b2.v[4]), "v"(a2.v[5]), "v"(b2.v[3]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[6]), "v"(b2.v[2]), "v"(a2.v[7]), "v"(b2.v[1]), "v"(a2.v[8]), "v"(b2.v[0]), "v"(m7), "v"(p1), "v"(m6), "v"(p2), "v"(m5), "v"(p3), "v"(m4), "v"(p4), "v"(m0), "v"(p8));
    m8 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8));
    col >>= 29;
    // column 9: 29 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[1]), "v"(b0.v[8]), "v"(a0.v[2]), "v"(b0.v[7]), "v"(a0.v[3]), "v"(b0.v[6]), "v"(a0.v[4]), "v"(b0.v[5]), "v"(a0.v[5]), "v"(b0.v[4]), "v"(a0.v[6]), "v"(b0.v[3]), "v"(a0.v[7]), "v"(b0.v[2]), "v"(a0.v[8]), "v"(b0.v[1]), "v"(a1.v[1]), "v"(b1.v[8]), "v"(a1.v[2]), "v"(b1.v[7]), "v"(a1.v[3]), "v"(b1.v[6]), "v"(a1.v[4]), "v"(b1.v[5]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[5]), "v"(b1.v[4]), "v"(a1.v[6]), "v"(b1.v[3]), "v"(a1.v[7]), "v"(b1.v[2]), "v"(a1.v[8]), "v"(b1.v[1]), "v"(a2.v[1]), "v"(b2.v[8]), "v"(a2.v[2]), "v"(b2.v[7]), "v"(a2.v[3]), "v"(b2.v[6]), "v"(a2.v[4]), "v"(b2.v[5]), "v"(a2.v[5]), "v"(b2.v[4]), "v"(a2.v[6]), "v"(b2.v[3]), "v"(a2.v[7]), "v"(b2.v[2]), "v"(a2.v[8]), "v"(b2.v[1]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8), "v"(p1), "v"(m7), "v"(p2), "v"(m6), "v"(p3), "v"(m5), "v"(p4), "v"(m1), "v"(p8));
    r.v[0] = (uint32_t)col & M29; col >>= 29;
    // column 10: 25 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[2]), "v"(b0.v[8]), "v"(a0.v[3]), "v"(b0.v[7]), "v"(a0.v[4]), "v"(b0.v[6]), "v"(a0.v[5]), "v"(b0.v[5]), "v"(a0.v[6]), "v"(b0.v[4]), "v"(a0.v[7]), "v"(b0.v[3]), "v"(a0.v[8]), "v"(b0.v[2]), "v"(a1.v[2]), "v"(b1.v[8]), "v"(a1.v[3]), "v"(b1.v[7]), "v"(a1.v[4]), "v"(b1.v[6]), "v"(a1.v[5]), "v"(b1.v[5]), "v"(a1.v[6]), "v"(b1.v[4]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[7]), "v"(b1.v[3]), "v"(a1.v[8]), "v"(b1.v[2]), "v"(a2.v[2]), "v"(b2.v[8]), "v"(a2.v[3]), "v"(b2.v[7]), "v"(a2.v[4]), "v"(b2.v[6]), "v"(a2.v[5]), "v"(b2.v[5]), "v"(a2.v[6]), "v"(b2.v[4]), "v"(a2.v[7]), "v"(b2.v[3]), "v"(a2.v[8]), "v"(b2.v[2]), "v"(m8), "v"(p2), "v"(m7), "v"(p3), "v"(m6), "v"(p4));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2), "v"(p8));
    r.v[1] = (uint32_t)col & M29; col >>= 29;
    // column 11: 21 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[3]), "v"(b0.v[8]), "v"(a0.v[4]), "v"(b0.v[7]), "v"(a0.v[5]), "v"(b0.v[6]), "v"(a0.v[6]), "v"(b0.v[5]), "v"(a0.v[7]), "v"(b0.v[4]), "v"(a0.v[8]), "v"(b0.v[3]), "v"(a1.v[3]), "v"(b1.v[8]), "v"(a1.v[4]), "v"(b1.v[7]), "v"(a1.v[5]), "v"(b1.v[6]), "v"(a1.v[6]), "v"(b1.v[5]), "v"(a1.v[7]), "v"(b1.v[4]), "v"(a1.v[8]), "v"(b1.v[3]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[3]), "v"(b2.v[8]), "v"(a2.v[4]), "v"(b2.v[7]), "v"(a2.v[5]), "v"(b2.v[6]), "v"(a2.v[6]), "v"(b2.v[5]), "v"(a2.v[7]), "v"(b2.v[4]), "v"(a2.v[8]), "v"(b2.v[3]), "v"(m8), "v"(p3), "v"(m7), "v"(p4), "v"(m3), "v"(p8));
    r.v[2] = (uint32_t)col & M29; col >>= 29;
    // column 12: 17 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[4]), "v"(b0.v[8]), "v"(a0.v[5]), "v"(b0.v[7]), "v"(a0.v[6]), "v"(b0.v[6]), "v"(a0.v[7]), "v"(b0.v[5]), "v"(a0.v[8]), "v"(b0.v[4]), "v"(a1.v[4]), "v"(b1.v[8]), "v"(a1.v[5]), "v"(b1.v[7]), "v"(a1.v[6]), "v"(b1.v[6]), "v"(a1.v[7]), "v"(b1.v[5]), "v"(a1.v[8]), "v"(b1.v[4]), "v"(a2.v[4]), "v"(b2.v[8]), "v"(a2.v[5]), "v"(b2.v[7]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[6]), "v"(b2.v[6]), "v"(a2.v[7]), "v"(b2.v[5]), "v"(a2.v[8]), "v"(b2.v[4]), "v"(m8), "v"(p4), "v"(m4), "v"(p8));
    r.v[3] = (uint32_t)col & M29; col >>= 29;
    // column 13: 13 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[5]), "v"(b0.v[8]), "v"(a0.v[6]), "v"(b0.v[7]), "v"(a0.v[7]), "v"(b0.v[6]), "v"(a0.v[8]), "v"(b0.v[5]), "v"(a1.v[5]), "v"(b1.v[8]), "v"(a1.v[6]), "v"(b1.v[7]), "v"(a1.v[7]), "v"(b1.v[6]), "v"(a1.v[8]), "v"(b1.v[5]), "v"(a2.v[5]), "v"(b2.v[8]), "v"(a2.v[6]), "v"(b2.v[7]), "v"(a2.v[7]), "v"(b2.v[6]), "v"(a2.v[8]), "v"(b2.v[5]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5), "v"(p8));
    r.v[4] = (uint32_t)col & M29; col >>= 29;
    // column 14: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[6]), "v"(b0.v[8]), "v"(a0.v[7]), "v"(b0.v[7]), "v"(a0.v[8]), "v"(b0.v[6]), "v"(a1.v[6]), "v"(b1.v[8]), "v"(a1.v[7]), "v"(b1.v[7]), "v"(a1.v[8]), "v"(b1.v[6]), "v"(a2.v[6]), "v"(b2.v[8]), "v"(a2.v[7]), "v"(b2.v[7]), "v"(a2.v[8]), "v"(b2.v[6]), "v"(m6), "v"(p8));
    r.v[5] = (uint32_t)col & M29; col >>= 29;
    // column 15: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[7]), "v"(b0.v[8]), "v"(a0.v[8]), "v"(b0.v[7]), "v"(a1.v[7]), "v"(b1.v[8]), "v"(a1.v[8]), "v"(b1.v[7]), "v"(a2.v[7]), "v"(b2.v[8]), "v"(a2.v[8]), "v"(b2.v[7]), "v"(m7), "v"(p8));
    r.v[6] = (uint32_t)col & M29; col >>= 29;
    // column 16: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[8]), "v"(b0.v[8]), "v"(a1.v[8]), "v"(b1.v[8]), "v"(a2.v[8]), "v"(b2.v[8]), "v"(m8), "v"(p8));
    r.v[7] = (uint32_t)col & M29; col >>= 29;
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_mul_lz(const fe29_t &a, const fe29_t &b) {
    uint64_t col, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;
    // column 0: 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[0]));
    m0 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m0));
    col >>= 29;
    // column 1: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[1]), "v"(b.v[0]), "v"(m0), "v"(p1));
    m1 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1));
    col >>= 29;
    // column 2: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[0]), "v"(m1), "v"(p1), "v"(m0), "v"(p2));
    m2 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2));
    col >>= 29;
    // column 3: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[2]), "v"(b.v[1]), "v"(a.v[3]), "v"(b.v[0]), "v"(m2), "v"(p1), "v"(m1), "v"(p2), "v"(m0), "v"(p3));
    m3 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3));
    col >>= 29;
    // column 4: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[3]), "v"(b.v[1]), "v"(a.v[4]), "v"(b.v[0]), "v"(m3), "v"(p1), "v"(m2), "v"(p2), "v"(m1), "v"(p3), "v"(m0), "v"(p4));
    m4 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4));
    col >>= 29;
    // column 5: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[0]), "v"(m4), "v"(p1), "v"(m3), "v"(p2), "v"(m2), "v"(p3), "v"(m1), "v"(p4));
    m5 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5));
    col >>= 29;
    // column 6: 11 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[0]), "v"(m5), "v"(p1), "v"(m4), "v"(p2), "v"(m3), "v"(p3), "v"(m2), "v"(p4));
    m6 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6));
    col >>= 29;
    // column 7: 12 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0]), "v"(m6), "v"(p1), "v"(m5), "v"(p2), "v"(m4), "v"(p3), "v"(m3), "v"(p4));
    m7 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m7));
    col >>= 29;
    // column 8: 14 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[0]), "v"(m7), "v"(p1), "v"(m6), "v"(p2), "v"(m5), "v"(p3));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4), "v"(p4), "v"(m0), "v"(p8));
    m8 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8));
    col >>= 29;
    // column 9: 13 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[1]), "v"(b.v[8]), "v"(a.v[2]), "v"(b.v[7]), "v"(a.v[3]), "v"(b.v[6]), "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[2]), "v"(a.v[8]), "v"(b.v[1]), "v"(m8), "v"(p1), "v"(m7), "v"(p2), "v"(m6), "v"(p3), "v"(m5), "v"(p4));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1), "v"(p8));
    r.v[0] = (uint32_t)col & M29; col >>= 29;
    // column 10: 11 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[2]), "v"(b.v[8]), "v"(a.v[3]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[5]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[2]), "v"(m8), "v"(p2), "v"(m7), "v"(p3), "v"(m6), "v"(p4), "v"(m2), "v"(p8));
    r.v[1] = (uint32_t)col & M29; col >>= 29;
    // column 11: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[3]), "v"(b.v[8]), "v"(a.v[4]), "v"(b.v[7]), "v"(a.v[5]), "v"(b.v[6]), "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[7]), "v"(b.v[4]), "v"(a.v[8]), "v"(b.v[3]), "v"(m8), "v"(p3), "v"(m7), "v"(p4), "v"(m3), "v"(p8));
    r.v[2] = (uint32_t)col & M29; col >>= 29;
    // column 12: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[4]), "v"(b.v[8]), "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[4]), "v"(m8), "v"(p4), "v"(m4), "v"(p8));
    r.v[3] = (uint32_t)col & M29; col >>= 29;
    // column 13: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[5]), "v"(b.v[8]), "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[7]), "v"(b.v[6]), "v"(a.v[8]), "v"(b.v[5]), "v"(m5), "v"(p8));
    r.v[4] = (uint32_t)col & M29; col >>= 29;
    // column 14: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[6]), "v"(b.v[8]), "v"(a.v[7]), "v"(b.v[7]), "v"(a.v[8]), "v"(b.v[6]), "v"(m6), "v"(p8));
    r.v[5] = (uint32_t)col & M29; col >>= 29;
    // column 15: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[7]), "v"(b.v[8]), "v"(a.v[8]), "v"(b.v[7]), "v"(m7), "v"(p8));
    r.v[6] = (uint32_t)col & M29; col >>= 29;
    // column 16: 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(b.v[8]), "v"(m8), "v"(p8));
    r.v[7] = (uint32_t)col & M29; col >>= 29;
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_sqr_lz(const fe29_t &a) {
    const uint32_t d0 = a.v[0] << 1, d1 = a.v[1] << 1, d2 = a.v[2] << 1, d3 = a.v[3] << 1, d4 = a.v[4] << 1, d5 = a.v[5] << 1, d6 = a.v[6] << 1, d7 = a.v[7] << 1;   // limbs < 2^29: the doubled ones fit 32 bits
    uint64_t col, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;
    // column 0: 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(a.v[0]));
    m0 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m0));
    col >>= 29;
    // column 1: 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[1]), "v"(m0), "v"(p1));
    m1 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1));
    col >>= 29;
    // column 2: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[2]), "v"(a.v[1]), "v"(a.v[1]), "v"(m1), "v"(p1), "v"(m0), "v"(p2));
    m2 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2));
    col >>= 29;
    // column 3: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[3]), "v"(d1), "v"(a.v[2]), "v"(m2), "v"(p1), "v"(m1), "v"(p2), "v"(m0), "v"(p3));
    m3 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3));
    col >>= 29;
    // column 4: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[4]), "v"(d1), "v"(a.v[3]), "v"(a.v[2]), "v"(a.v[2]), "v"(m3), "v"(p1), "v"(m2), "v"(p2), "v"(m1), "v"(p3), "v"(m0), "v"(p4));
    m4 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4));
    col >>= 29;
    // column 5: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[5]), "v"(d1), "v"(a.v[4]), "v"(d2), "v"(a.v[3]), "v"(m4), "v"(p1), "v"(m3), "v"(p2), "v"(m2), "v"(p3), "v"(m1), "v"(p4));
    m5 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5));
    col >>= 29;
    // column 6: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[6]), "v"(d1), "v"(a.v[5]), "v"(d2), "v"(a.v[4]), "v"(a.v[3]), "v"(a.v[3]), "v"(m5), "v"(p1), "v"(m4), "v"(p2), "v"(m3), "v"(p3), "v"(m2), "v"(p4));
    m6 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6));
    col >>= 29;
    // column 7: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[7]), "v"(d1), "v"(a.v[6]), "v"(d2), "v"(a.v[5]), "v"(d3), "v"(a.v[4]), "v"(m6), "v"(p1), "v"(m5), "v"(p2), "v"(m4), "v"(p3), "v"(m3), "v"(p4));
    m7 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m7));
    col >>= 29;
    // column 8: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[8]), "v"(d1), "v"(a.v[7]), "v"(d2), "v"(a.v[6]), "v"(d3), "v"(a.v[5]), "v"(a.v[4]), "v"(a.v[4]), "v"(m7), "v"(p1), "v"(m6), "v"(p2), "v"(m5), "v"(p3), "v"(m4), "v"(p4), "v"(m0), "v"(p8));
    m8 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8));
    col >>= 29;
    // column 9: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d1), "v"(a.v[8]), "v"(d2), "v"(a.v[7]), "v"(d3), "v"(a.v[6]), "v"(d4), "v"(a.v[5]), "v"(m8), "v"(p1), "v"(m7), "v"(p2), "v"(m6), "v"(p3), "v"(m5), "v"(p4), "v"(m1), "v"(p8));
    r.v[0] = (uint32_t)col & M29; col >>= 29;
    // column 10: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d2), "v"(a.v[8]), "v"(d3), "v"(a.v[7]), "v"(d4), "v"(a.v[6]), "v"(a.v[5]), "v"(a.v[5]), "v"(m8), "v"(p2), "v"(m7), "v"(p3), "v"(m6), "v"(p4), "v"(m2), "v"(p8));
    r.v[1] = (uint32_t)col & M29; col >>= 29;
    // column 11: 6 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d3), "v"(a.v[8]), "v"(d4), "v"(a.v[7]), "v"(d5), "v"(a.v[6]), "v"(m8), "v"(p3), "v"(m7), "v"(p4), "v"(m3), "v"(p8));
    r.v[2] = (uint32_t)col & M29; col >>= 29;
    // column 12: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d4), "v"(a.v[8]), "v"(d5), "v"(a.v[7]), "v"(a.v[6]), "v"(a.v[6]), "v"(m8), "v"(p4), "v"(m4), "v"(p8));
    r.v[3] = (uint32_t)col & M29; col >>= 29;
    // column 13: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d5), "v"(a.v[8]), "v"(d6), "v"(a.v[7]), "v"(m5), "v"(p8));
    r.v[4] = (uint32_t)col & M29; col >>= 29;
    // column 14: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d6), "v"(a.v[8]), "v"(a.v[7]), "v"(a.v[7]), "v"(m6), "v"(p8));
    r.v[5] = (uint32_t)col & M29; col >>= 29;
    // column 15: 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d7), "v"(a.v[8]), "v"(m7), "v"(p8));
    r.v[6] = (uint32_t)col & M29; col >>= 29;
    // column 16: 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(a.v[8]), "v"(m8), "v"(p8));
    r.v[7] = (uint32_t)col & M29; col >>= 29;
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_dot3rc_lz(const fe29_t &a0, const fe29_t &b0, const fe29_t &a1, const fe29_t &b1, const fe29_t &a2, const fe29_t &b2, const fe29_t &c) {
    uint64_t col, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;
    // column 0: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0"
        : "=&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[0]), "v"(c.v[0]));
    m0 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m0));
    col >>= 29;
    // column 1: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, 1, %0\n\tv_mad_u64_u32 %0, %1, %15, %16, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[1]), "v"(a0.v[1]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[1]), "v"(a1.v[1]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[1]), "v"(a2.v[1]), "v"(b2.v[0]), "v"(c.v[1]), "v"(m0), "v"(p1));
    m1 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1));
    col >>= 29;
    // column 2: 12 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, 1, %0\n\tv_mad_u64_u32 %0, %1, %21, %22, %0\n\tv_mad_u64_u32 %0, %1, %23, %24, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[2]), "v"(a0.v[1]), "v"(b0.v[1]), "v"(a0.v[2]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[2]), "v"(a1.v[1]), "v"(b1.v[1]), "v"(a1.v[2]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[2]), "v"(a2.v[1]), "v"(b2.v[1]), "v"(a2.v[2]), "v"(b2.v[0]), "v"(c.v[2]), "v"(m1), "v"(p1), "v"(m0), "v"(p2));
    m2 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2));
    col >>= 29;
    // column 3: 16 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[3]), "v"(a0.v[1]), "v"(b0.v[2]), "v"(a0.v[2]), "v"(b0.v[1]), "v"(a0.v[3]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[3]), "v"(a1.v[1]), "v"(b1.v[2]), "v"(a1.v[2]), "v"(b1.v[1]), "v"(a1.v[3]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[3]), "v"(a2.v[1]), "v"(b2.v[2]), "v"(a2.v[2]), "v"(b2.v[1]), "v"(a2.v[3]), "v"(b2.v[0]));
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0\n\tv_mad_u64_u32 %0, %1, %3, %4, %0\n\tv_mad_u64_u32 %0, %1, %5, %6, %0\n\tv_mad_u64_u32 %0, %1, %7, %8, %0"
        : "+&v"(col), "=&s"(cc) : "v"(c.v[3]), "v"(m2), "v"(p1), "v"(m1), "v"(p2), "v"(m0), "v"(p3));
    m3 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3));
    col >>= 29;
    // column 4: 20 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[4]), "v"(a0.v[1]), "v"(b0.v[3]), "v"(a0.v[2]), "v"(b0.v[2]), "v"(a0.v[3]), "v"(b0.v[1]), "v"(a0.v[4]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[4]), "v"(a1.v[1]), "v"(b1.v[3]), "v"(a1.v[2]), "v"(b1.v[2]), "v"(a1.v[3]), "v"(b1.v[1]), "v"(a1.v[4]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[4]), "v"(a2.v[1]), "v"(b2.v[3]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0\n\tv_mad_u64_u32 %0, %1, %9, %10, %0\n\tv_mad_u64_u32 %0, %1, %11, %12, %0\n\tv_mad_u64_u32 %0, %1, %13, %14, %0\n\tv_mad_u64_u32 %0, %1, %15, %16, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[2]), "v"(b2.v[2]), "v"(a2.v[3]), "v"(b2.v[1]), "v"(a2.v[4]), "v"(b2.v[0]), "v"(c.v[4]), "v"(m3), "v"(p1), "v"(m2), "v"(p2), "v"(m1), "v"(p3), "v"(m0), "v"(p4));
    m4 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4));
    col >>= 29;
    // column 5: 23 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[5]), "v"(a0.v[1]), "v"(b0.v[4]), "v"(a0.v[2]), "v"(b0.v[3]), "v"(a0.v[3]), "v"(b0.v[2]), "v"(a0.v[4]), "v"(b0.v[1]), "v"(a0.v[5]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[5]), "v"(a1.v[1]), "v"(b1.v[4]), "v"(a1.v[2]), "v"(b1.v[3]), "v"(a1.v[3]), "v"(b1.v[2]), "v"(a1.v[4]), "v"(b1.v[1]), "v"(a1.v[5]), "v"(b1.v[0]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, 1, %0\n\tv_mad_u64_u32 %0, %1, %15, %16, %0\n\tv_mad_u64_u32 %0, %1, %17, %18, %0\n\tv_mad_u64_u32 %0, %1, %19, %20, %0\n\tv_mad_u64_u32 %0, %1, %21, %22, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[0]), "v"(b2.v[5]), "v"(a2.v[1]), "v"(b2.v[4]), "v"(a2.v[2]), "v"(b2.v[3]), "v"(a2.v[3]), "v"(b2.v[2]), "v"(a2.v[4]), "v"(b2.v[1]), "v"(a2.v[5]), "v"(b2.v[0]), "v"(c.v[5]), "v"(m4), "v"(p1), "v"(m3), "v"(p2), "v"(m2), "v"(p3), "v"(m1), "v"(p4));
    m5 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5));
    col >>= 29;
    // column 6: 26 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[6]), "v"(a0.v[1]), "v"(b0.v[5]), "v"(a0.v[2]), "v"(b0.v[4]), "v"(a0.v[3]), "v"(b0.v[3]), "v"(a0.v[4]), "v"(b0.v[2]), "v"(a0.v[5]), "v"(b0.v[1]), "v"(a0.v[6]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[6]), "v"(a1.v[1]), "v"(b1.v[5]), "v"(a1.v[2]), "v"(b1.v[4]), "v"(a1.v[3]), "v"(b1.v[3]), "v"(a1.v[4]), "v"(b1.v[2]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, 1, %0\n\tv_mad_u64_u32 %0, %1, %21, %22, %0\n\tv_mad_u64_u32 %0, %1, %23, %24, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[5]), "v"(b1.v[1]), "v"(a1.v[6]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[6]), "v"(a2.v[1]), "v"(b2.v[5]), "v"(a2.v[2]), "v"(b2.v[4]), "v"(a2.v[3]), "v"(b2.v[3]), "v"(a2.v[4]), "v"(b2.v[2]), "v"(a2.v[5]), "v"(b2.v[1]), "v"(a2.v[6]), "v"(b2.v[0]), "v"(c.v[6]), "v"(m5), "v"(p1), "v"(m4), "v"(p2));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3), "v"(p3), "v"(m2), "v"(p4));
    m6 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6));
    col >>= 29;
    // column 7: 29 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[7]), "v"(a0.v[1]), "v"(b0.v[6]), "v"(a0.v[2]), "v"(b0.v[5]), "v"(a0.v[3]), "v"(b0.v[4]), "v"(a0.v[4]), "v"(b0.v[3]), "v"(a0.v[5]), "v"(b0.v[2]), "v"(a0.v[6]), "v"(b0.v[1]), "v"(a0.v[7]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[7]), "v"(a1.v[1]), "v"(b1.v[6]), "v"(a1.v[2]), "v"(b1.v[5]), "v"(a1.v[3]), "v"(b1.v[4]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[4]), "v"(b1.v[3]), "v"(a1.v[5]), "v"(b1.v[2]), "v"(a1.v[6]), "v"(b1.v[1]), "v"(a1.v[7]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[7]), "v"(a2.v[1]), "v"(b2.v[6]), "v"(a2.v[2]), "v"(b2.v[5]), "v"(a2.v[3]), "v"(b2.v[4]), "v"(a2.v[4]), "v"(b2.v[3]), "v"(a2.v[5]), "v"(b2.v[2]), "v"(a2.v[6]), "v"(b2.v[1]), "v"(a2.v[7]), "v"(b2.v[0]));
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0\n\tv_mad_u64_u32 %0, %1, %3, %4, %0\n\tv_mad_u64_u32 %0, %1, %5, %6, %0\n\tv_mad_u64_u32 %0, %1, %7, %8, %0\n\tv_mad_u64_u32 %0, %1, %9, %10, %0"
        : "+&v"(col), "=&s"(cc) : "v"(c.v[7]), "v"(m6), "v"(p1), "v"(m5), "v"(p2), "v"(m4), "v"(p3), "v"(m3), "v"(p4));
    m7 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m7));
    col >>= 29;
    // column 8: 33 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[8]), "v"(a0.v[1]), "v"(b0.v[7]), "v"(a0.v[2]), "v"(b0.v[6]), "v"(a0.v[3]), "v"(b0.v[5]), "v"(a0.v[4]), "v"(b0.v[4]), "v"(a0.v[5]), "v"(b0.v[3]), "v"(a0.v[6]), "v"(b0.v[2]), "v"(a0.v[7]), "v"(b0.v[1]), "v"(a0.v[8]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[8]), "v"(a1.v[1]), "v"(b1.v[7]), "v"(a1.v[2]), "v"(b1.v[6]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[3]), "v"(b1.v[5]), "v"(a1.v[4]), "v"(b1.v[4]), "v"(a1.v[5]), "v"(b1.v[3]), "v"(a1.v[6]), "v"(b1.v[2]), "v"(a1.v[7]), "v"(b1.v[1]), "v"(a1.v[8]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[8]), "v"(a2.v[1]), "v"(b2.v[7]), "v"(a2.v[2]), "v"(b2.v[6]), "v"(a2.v[3]), "v"(b2.v[5]), "v"(a2.v[4]), "v"(b2.v[4]), "v"(a2.v[5]), "v"(b2.v[3]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0\n\tv_mad_u64_u32 %0, %1, %9, %10, %0\n\tv_mad_u64_u32 %0, %1, %11, %12, %0\n\tv_mad_u64_u32 %0, %1, %13, %14, %0\n\tv_mad_u64_u32 %0, %1, %15, %16, %0\n\tv_mad_u64_u32 %0, %1, %17, %18, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[6]), "v"(b2.v[2]), "v"(a2.v[7]), "v"(b2.v[1]), "v"(a2.v[8]), "v"(b2.v[0]), "v"(c.v[8]), "v"(m7), "v"(p1), "v"(m6), "v"(p2), "v"(m5), "v"(p3), "v"(m4), "v"(p4), "v"(m0), "v"(p8));
    m8 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8));
    col >>= 29;
    // column 9: 29 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[1]), "v"(b0.v[8]), "v"(a0.v[2]), "v"(b0.v[7]), "v"(a0.v[3]), "v"(b0.v[6]), "v"(a0.v[4]), "v"(b0.v[5]), "v"(a0.v[5]), "v"(b0.v[4]), "v"(a0.v[6]), "v"(b0.v[3]), "v"(a0.v[7]), "v"(b0.v[2]), "v"(a0.v[8]), "v"(b0.v[1]), "v"(a1.v[1]), "v"(b1.v[8]), "v"(a1.v[2]), "v"(b1.v[7]), "v"(a1.v[3]), "v"(b1.v[6]), "v"(a1.v[4]), "v"(b1.v[5]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[5]), "v"(b1.v[4]), "v"(a1.v[6]), "v"(b1.v[3]), "v"(a1.v[7]), "v"(b1.v[2]), "v"(a1.v[8]), "v"(b1.v[1]), "v"(a2.v[1]), "v"(b2.v[8]), "v"(a2.v[2]), "v"(b2.v[7]), "v"(a2.v[3]), "v"(b2.v[6]), "v"(a2.v[4]), "v"(b2.v[5]), "v"(a2.v[5]), "v"(b2.v[4]), "v"(a2.v[6]), "v"(b2.v[3]), "v"(a2.v[7]), "v"(b2.v[2]), "v"(a2.v[8]), "v"(b2.v[1]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8), "v"(p1), "v"(m7), "v"(p2), "v"(m6), "v"(p3), "v"(m5), "v"(p4), "v"(m1), "v"(p8));
    r.v[0] = (uint32_t)col & M29; col >>= 29;
    // column 10: 25 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[2]), "v"(b0.v[8]), "v"(a0.v[3]), "v"(b0.v[7]), "v"(a0.v[4]), "v"(b0.v[6]), "v"(a0.v[5]), "v"(b0.v[5]), "v"(a0.v[6]), "v"(b0.v[4]), "v"(a0.v[7]), "v"(b0.v[3]), "v"(a0.v[8]), "v"(b0.v[2]), "v"(a1.v[2]), "v"(b1.v[8]), "v"(a1.v[3]), "v"(b1.v[7]), "v"(a1.v[4]), "v"(b1.v[6]), "v"(a1.v[5]), "v"(b1.v[5]), "v"(a1.v[6]), "v"(b1.v[4]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[7]), "v"(b1.v[3]), "v"(a1.v[8]), "v"(b1.v[2]), "v"(a2.v[2]), "v"(b2.v[8]), "v"(a2.v[3]), "v"(b2.v[7]), "v"(a2.v[4]), "v"(b2.v[6]), "v"(a2.v[5]), "v"(b2.v[5]), "v"(a2.v[6]), "v"(b2.v[4]), "v"(a2.v[7]), "v"(b2.v[3]), "v"(a2.v[8]), "v"(b2.v[2]), "v"(m8), "v"(p2), "v"(m7), "v"(p3), "v"(m6), "v"(p4));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2), "v"(p8));
    r.v[1] = (uint32_t)col & M29; col >>= 29;
    // column 11: 21 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[3]), "v"(b0.v[8]), "v"(a0.v[4]), "v"(b0.v[7]), "v"(a0.v[5]), "v"(b0.v[6]), "v"(a0.v[6]), "v"(b0.v[5]), "v"(a0.v[7]), "v"(b0.v[4]), "v"(a0.v[8]), "v"(b0.v[3]), "v"(a1.v[3]), "v"(b1.v[8]), "v"(a1.v[4]), "v"(b1.v[7]), "v"(a1.v[5]), "v"(b1.v[6]), "v"(a1.v[6]), "v"(b1.v[5]), "v"(a1.v[7]), "v"(b1.v[4]), "v"(a1.v[8]), "v"(b1.v[3]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[3]), "v"(b2.v[8]), "v"(a2.v[4]), "v"(b2.v[7]), "v"(a2.v[5]), "v"(b2.v[6]), "v"(a2.v[6]), "v"(b2.v[5]), "v"(a2.v[7]), "v"(b2.v[4]), "v"(a2.v[8]), "v"(b2.v[3]), "v"(m8), "v"(p3), "v"(m7), "v"(p4), "v"(m3), "v"(p8));
    r.v[2] = (uint32_t)col & M29; col >>= 29;
    // column 12: 17 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[4]), "v"(b0.v[8]), "v"(a0.v[5]), "v"(b0.v[7]), "v"(a0.v[6]), "v"(b0.v[6]), "v"(a0.v[7]), "v"(b0.v[5]), "v"(a0.v[8]), "v"(b0.v[4]), "v"(a1.v[4]), "v"(b1.v[8]), "v"(a1.v[5]), "v"(b1.v[7]), "v"(a1.v[6]), "v"(b1.v[6]), "v"(a1.v[7]), "v"(b1.v[5]), "v"(a1.v[8]), "v"(b1.v[4]), "v"(a2.v[4]), "v"(b2.v[8]), "v"(a2.v[5]), "v"(b2.v[7]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[6]), "v"(b2.v[6]), "v"(a2.v[7]), "v"(b2.v[5]), "v"(a2.v[8]), "v"(b2.v[4]), "v"(m8), "v"(p4), "v"(m4), "v"(p8));
    r.v[3] = (uint32_t)col & M29; col >>= 29;
    // column 13: 13 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[5]), "v"(b0.v[8]), "v"(a0.v[6]), "v"(b0.v[7]), "v"(a0.v[7]), "v"(b0.v[6]), "v"(a0.v[8]), "v"(b0.v[5]), "v"(a1.v[5]), "v"(b1.v[8]), "v"(a1.v[6]), "v"(b1.v[7]), "v"(a1.v[7]), "v"(b1.v[6]), "v"(a1.v[8]), "v"(b1.v[5]), "v"(a2.v[5]), "v"(b2.v[8]), "v"(a2.v[6]), "v"(b2.v[7]), "v"(a2.v[7]), "v"(b2.v[6]), "v"(a2.v[8]), "v"(b2.v[5]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5), "v"(p8));
    r.v[4] = (uint32_t)col & M29; col >>= 29;
    // column 14: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[6]), "v"(b0.v[8]), "v"(a0.v[7]), "v"(b0.v[7]), "v"(a0.v[8]), "v"(b0.v[6]), "v"(a1.v[6]), "v"(b1.v[8]), "v"(a1.v[7]), "v"(b1.v[7]), "v"(a1.v[8]), "v"(b1.v[6]), "v"(a2.v[6]), "v"(b2.v[8]), "v"(a2.v[7]), "v"(b2.v[7]), "v"(a2.v[8]), "v"(b2.v[6]), "v"(m6), "v"(p8));
    r.v[5] = (uint32_t)col & M29; col >>= 29;
    // column 15: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[7]), "v"(b0.v[8]), "v"(a0.v[8]), "v"(b0.v[7]), "v"(a1.v[7]), "v"(b1.v[8]), "v"(a1.v[8]), "v"(b1.v[7]), "v"(a2.v[7]), "v"(b2.v[8]), "v"(a2.v[8]), "v"(b2.v[7]), "v"(m7), "v"(p8));
    r.v[6] = (uint32_t)col & M29; col >>= 29;
    // column 16: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[8]), "v"(b0.v[8]), "v"(a1.v[8]), "v"(b1.v[8]), "v"(a2.v[8]), "v"(b2.v[8]), "v"(m8), "v"(p8));
    r.v[7] = (uint32_t)col & M29; col >>= 29;
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_mulrc_lz(const fe29_t &a, const fe29_t &b, const fe29_t &c) {
    uint64_t col, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;
    // column 0: 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, 1, %0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[0]), "v"(c.v[0]));
    m0 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m0));
    col >>= 29;
    // column 1: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, 1, %0\n\tv_mad_u64_u32 %0, %1, %7, %8, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[1]), "v"(b.v[0]), "v"(c.v[1]), "v"(m0), "v"(p1));
    m1 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1));
    col >>= 29;
    // column 2: 6 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0\n\tv_mad_u64_u32 %0, %1, %9, %10, %0\n\tv_mad_u64_u32 %0, %1, %11, %12, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[0]), "v"(c.v[2]), "v"(m1), "v"(p1), "v"(m0), "v"(p2));
    m2 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2));
    col >>= 29;
    // column 3: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, 1, %0\n\tv_mad_u64_u32 %0, %1, %11, %12, %0\n\tv_mad_u64_u32 %0, %1, %13, %14, %0\n\tv_mad_u64_u32 %0, %1, %15, %16, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[2]), "v"(b.v[1]), "v"(a.v[3]), "v"(b.v[0]), "v"(c.v[3]), "v"(m2), "v"(p1), "v"(m1), "v"(p2), "v"(m0), "v"(p3));
    m3 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3));
    col >>= 29;
    // column 4: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, 1, %0\n\tv_mad_u64_u32 %0, %1, %13, %14, %0\n\tv_mad_u64_u32 %0, %1, %15, %16, %0\n\tv_mad_u64_u32 %0, %1, %17, %18, %0\n\tv_mad_u64_u32 %0, %1, %19, %20, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[3]), "v"(b.v[1]), "v"(a.v[4]), "v"(b.v[0]), "v"(c.v[4]), "v"(m3), "v"(p1), "v"(m2), "v"(p2), "v"(m1), "v"(p3), "v"(m0), "v"(p4));
    m4 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4));
    col >>= 29;
    // column 5: 11 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, 1, %0\n\tv_mad_u64_u32 %0, %1, %15, %16, %0\n\tv_mad_u64_u32 %0, %1, %17, %18, %0\n\tv_mad_u64_u32 %0, %1, %19, %20, %0\n\tv_mad_u64_u32 %0, %1, %21, %22, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[0]), "v"(c.v[5]), "v"(m4), "v"(p1), "v"(m3), "v"(p2), "v"(m2), "v"(p3), "v"(m1), "v"(p4));
    m5 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5));
    col >>= 29;
    // column 6: 12 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, 1, %0\n\tv_mad_u64_u32 %0, %1, %17, %18, %0\n\tv_mad_u64_u32 %0, %1, %19, %20, %0\n\tv_mad_u64_u32 %0, %1, %21, %22, %0\n\tv_mad_u64_u32 %0, %1, %23, %24, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[0]), "v"(c.v[6]), "v"(m5), "v"(p1), "v"(m4), "v"(p2), "v"(m3), "v"(p3), "v"(m2), "v"(p4));
    m6 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6));
    col >>= 29;
    // column 7: 13 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, 1, %0\n\tv_mad_u64_u32 %0, %1, %19, %20, %0\n\tv_mad_u64_u32 %0, %1, %21, %22, %0\n\tv_mad_u64_u32 %0, %1, %23, %24, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0]), "v"(c.v[7]), "v"(m6), "v"(p1), "v"(m5), "v"(p2), "v"(m4), "v"(p3));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3), "v"(p4));
    m7 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m7));
    col >>= 29;
    // column 8: 15 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, 1, %0\n\tv_mad_u64_u32 %0, %1, %21, %22, %0\n\tv_mad_u64_u32 %0, %1, %23, %24, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[0]), "v"(c.v[8]), "v"(m7), "v"(p1), "v"(m6), "v"(p2));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5), "v"(p3), "v"(m4), "v"(p4), "v"(m0), "v"(p8));
    m8 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8));
    col >>= 29;
    // column 9: 13 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[1]), "v"(b.v[8]), "v"(a.v[2]), "v"(b.v[7]), "v"(a.v[3]), "v"(b.v[6]), "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[2]), "v"(a.v[8]), "v"(b.v[1]), "v"(m8), "v"(p1), "v"(m7), "v"(p2), "v"(m6), "v"(p3), "v"(m5), "v"(p4));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1), "v"(p8));
    r.v[0] = (uint32_t)col & M29; col >>= 29;
    // column 10: 11 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[2]), "v"(b.v[8]), "v"(a.v[3]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[5]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[2]), "v"(m8), "v"(p2), "v"(m7), "v"(p3), "v"(m6), "v"(p4), "v"(m2), "v"(p8));
    r.v[1] = (uint32_t)col & M29; col >>= 29;
    // column 11: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[3]), "v"(b.v[8]), "v"(a.v[4]), "v"(b.v[7]), "v"(a.v[5]), "v"(b.v[6]), "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[7]), "v"(b.v[4]), "v"(a.v[8]), "v"(b.v[3]), "v"(m8), "v"(p3), "v"(m7), "v"(p4), "v"(m3), "v"(p8));
    r.v[2] = (uint32_t)col & M29; col >>= 29;
    // column 12: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[4]), "v"(b.v[8]), "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[4]), "v"(m8), "v"(p4), "v"(m4), "v"(p8));
    r.v[3] = (uint32_t)col & M29; col >>= 29;
    // column 13: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[5]), "v"(b.v[8]), "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[7]), "v"(b.v[6]), "v"(a.v[8]), "v"(b.v[5]), "v"(m5), "v"(p8));
    r.v[4] = (uint32_t)col & M29; col >>= 29;
    // column 14: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[6]), "v"(b.v[8]), "v"(a.v[7]), "v"(b.v[7]), "v"(a.v[8]), "v"(b.v[6]), "v"(m6), "v"(p8));
    r.v[5] = (uint32_t)col & M29; col >>= 29;
    // column 15: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[7]), "v"(b.v[8]), "v"(a.v[8]), "v"(b.v[7]), "v"(m7), "v"(p8));
    r.v[6] = (uint32_t)col & M29; col >>= 29;
    // column 16: 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(b.v[8]), "v"(m8), "v"(p8));
    r.v[7] = (uint32_t)col & M29; col >>= 29;
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_dot2rc_lz(const fe29_t &a0, const fe29_t &b0, const fe29_t &a1, const fe29_t &b1, const fe29_t &c) {
    uint64_t col, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;
    // column 0: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, 1, %0"
        : "=&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[0]), "v"(c.v[0]));
    m0 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m0));
    col >>= 29;
    // column 1: 6 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, 1, %0\n\tv_mad_u64_u32 %0, %1, %11, %12, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[1]), "v"(a0.v[1]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[1]), "v"(a1.v[1]), "v"(b1.v[0]), "v"(c.v[1]), "v"(m0), "v"(p1));
    m1 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1));
    col >>= 29;
    // column 2: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, 1, %0\n\tv_mad_u64_u32 %0, %1, %15, %16, %0\n\tv_mad_u64_u32 %0, %1, %17, %18, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[2]), "v"(a0.v[1]), "v"(b0.v[1]), "v"(a0.v[2]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[2]), "v"(a1.v[1]), "v"(b1.v[1]), "v"(a1.v[2]), "v"(b1.v[0]), "v"(c.v[2]), "v"(m1), "v"(p1), "v"(m0), "v"(p2));
    m2 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2));
    col >>= 29;
    // column 3: 12 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, 1, %0\n\tv_mad_u64_u32 %0, %1, %19, %20, %0\n\tv_mad_u64_u32 %0, %1, %21, %22, %0\n\tv_mad_u64_u32 %0, %1, %23, %24, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[3]), "v"(a0.v[1]), "v"(b0.v[2]), "v"(a0.v[2]), "v"(b0.v[1]), "v"(a0.v[3]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[3]), "v"(a1.v[1]), "v"(b1.v[2]), "v"(a1.v[2]), "v"(b1.v[1]), "v"(a1.v[3]), "v"(b1.v[0]), "v"(c.v[3]), "v"(m2), "v"(p1), "v"(m1), "v"(p2), "v"(m0), "v"(p3));
    m3 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3));
    col >>= 29;
    // column 4: 15 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, 1, %0\n\tv_mad_u64_u32 %0, %1, %23, %24, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[4]), "v"(a0.v[1]), "v"(b0.v[3]), "v"(a0.v[2]), "v"(b0.v[2]), "v"(a0.v[3]), "v"(b0.v[1]), "v"(a0.v[4]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[4]), "v"(a1.v[1]), "v"(b1.v[3]), "v"(a1.v[2]), "v"(b1.v[2]), "v"(a1.v[3]), "v"(b1.v[1]), "v"(a1.v[4]), "v"(b1.v[0]), "v"(c.v[4]), "v"(m3), "v"(p1));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2), "v"(p2), "v"(m1), "v"(p3), "v"(m0), "v"(p4));
    m4 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4));
    col >>= 29;
    // column 5: 17 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[5]), "v"(a0.v[1]), "v"(b0.v[4]), "v"(a0.v[2]), "v"(b0.v[3]), "v"(a0.v[3]), "v"(b0.v[2]), "v"(a0.v[4]), "v"(b0.v[1]), "v"(a0.v[5]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[5]), "v"(a1.v[1]), "v"(b1.v[4]), "v"(a1.v[2]), "v"(b1.v[3]), "v"(a1.v[3]), "v"(b1.v[2]), "v"(a1.v[4]), "v"(b1.v[1]), "v"(a1.v[5]), "v"(b1.v[0]));
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0\n\tv_mad_u64_u32 %0, %1, %3, %4, %0\n\tv_mad_u64_u32 %0, %1, %5, %6, %0\n\tv_mad_u64_u32 %0, %1, %7, %8, %0\n\tv_mad_u64_u32 %0, %1, %9, %10, %0"
        : "+&v"(col), "=&s"(cc) : "v"(c.v[5]), "v"(m4), "v"(p1), "v"(m3), "v"(p2), "v"(m2), "v"(p3), "v"(m1), "v"(p4));
    m5 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5));
    col >>= 29;
    // column 6: 19 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[6]), "v"(a0.v[1]), "v"(b0.v[5]), "v"(a0.v[2]), "v"(b0.v[4]), "v"(a0.v[3]), "v"(b0.v[3]), "v"(a0.v[4]), "v"(b0.v[2]), "v"(a0.v[5]), "v"(b0.v[1]), "v"(a0.v[6]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[6]), "v"(a1.v[1]), "v"(b1.v[5]), "v"(a1.v[2]), "v"(b1.v[4]), "v"(a1.v[3]), "v"(b1.v[3]), "v"(a1.v[4]), "v"(b1.v[2]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, 1, %0\n\tv_mad_u64_u32 %0, %1, %7, %8, %0\n\tv_mad_u64_u32 %0, %1, %9, %10, %0\n\tv_mad_u64_u32 %0, %1, %11, %12, %0\n\tv_mad_u64_u32 %0, %1, %13, %14, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[5]), "v"(b1.v[1]), "v"(a1.v[6]), "v"(b1.v[0]), "v"(c.v[6]), "v"(m5), "v"(p1), "v"(m4), "v"(p2), "v"(m3), "v"(p3), "v"(m2), "v"(p4));
    m6 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6));
    col >>= 29;
    // column 7: 21 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[7]), "v"(a0.v[1]), "v"(b0.v[6]), "v"(a0.v[2]), "v"(b0.v[5]), "v"(a0.v[3]), "v"(b0.v[4]), "v"(a0.v[4]), "v"(b0.v[3]), "v"(a0.v[5]), "v"(b0.v[2]), "v"(a0.v[6]), "v"(b0.v[1]), "v"(a0.v[7]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[7]), "v"(a1.v[1]), "v"(b1.v[6]), "v"(a1.v[2]), "v"(b1.v[5]), "v"(a1.v[3]), "v"(b1.v[4]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, 1, %0\n\tv_mad_u64_u32 %0, %1, %11, %12, %0\n\tv_mad_u64_u32 %0, %1, %13, %14, %0\n\tv_mad_u64_u32 %0, %1, %15, %16, %0\n\tv_mad_u64_u32 %0, %1, %17, %18, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[4]), "v"(b1.v[3]), "v"(a1.v[5]), "v"(b1.v[2]), "v"(a1.v[6]), "v"(b1.v[1]), "v"(a1.v[7]), "v"(b1.v[0]), "v"(c.v[7]), "v"(m6), "v"(p1), "v"(m5), "v"(p2), "v"(m4), "v"(p3), "v"(m3), "v"(p4));
    m7 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m7));
    col >>= 29;
    // column 8: 24 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[8]), "v"(a0.v[1]), "v"(b0.v[7]), "v"(a0.v[2]), "v"(b0.v[6]), "v"(a0.v[3]), "v"(b0.v[5]), "v"(a0.v[4]), "v"(b0.v[4]), "v"(a0.v[5]), "v"(b0.v[3]), "v"(a0.v[6]), "v"(b0.v[2]), "v"(a0.v[7]), "v"(b0.v[1]), "v"(a0.v[8]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[8]), "v"(a1.v[1]), "v"(b1.v[7]), "v"(a1.v[2]), "v"(b1.v[6]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, 1, %0\n\tv_mad_u64_u32 %0, %1, %15, %16, %0\n\tv_mad_u64_u32 %0, %1, %17, %18, %0\n\tv_mad_u64_u32 %0, %1, %19, %20, %0\n\tv_mad_u64_u32 %0, %1, %21, %22, %0\n\tv_mad_u64_u32 %0, %1, %23, %24, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[3]), "v"(b1.v[5]), "v"(a1.v[4]), "v"(b1.v[4]), "v"(a1.v[5]), "v"(b1.v[3]), "v"(a1.v[6]), "v"(b1.v[2]), "v"(a1.v[7]), "v"(b1.v[1]), "v"(a1.v[8]), "v"(b1.v[0]), "v"(c.v[8]), "v"(m7), "v"(p1), "v"(m6), "v"(p2), "v"(m5), "v"(p3), "v"(m4), "v"(p4), "v"(m0), "v"(p8));
    m8 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8));
    col >>= 29;
    // column 9: 21 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[1]), "v"(b0.v[8]), "v"(a0.v[2]), "v"(b0.v[7]), "v"(a0.v[3]), "v"(b0.v[6]), "v"(a0.v[4]), "v"(b0.v[5]), "v"(a0.v[5]), "v"(b0.v[4]), "v"(a0.v[6]), "v"(b0.v[3]), "v"(a0.v[7]), "v"(b0.v[2]), "v"(a0.v[8]), "v"(b0.v[1]), "v"(a1.v[1]), "v"(b1.v[8]), "v"(a1.v[2]), "v"(b1.v[7]), "v"(a1.v[3]), "v"(b1.v[6]), "v"(a1.v[4]), "v"(b1.v[5]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[5]), "v"(b1.v[4]), "v"(a1.v[6]), "v"(b1.v[3]), "v"(a1.v[7]), "v"(b1.v[2]), "v"(a1.v[8]), "v"(b1.v[1]), "v"(m8), "v"(p1), "v"(m7), "v"(p2), "v"(m6), "v"(p3), "v"(m5), "v"(p4), "v"(m1), "v"(p8));
    r.v[0] = (uint32_t)col & M29; col >>= 29;
    // column 10: 18 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[2]), "v"(b0.v[8]), "v"(a0.v[3]), "v"(b0.v[7]), "v"(a0.v[4]), "v"(b0.v[6]), "v"(a0.v[5]), "v"(b0.v[5]), "v"(a0.v[6]), "v"(b0.v[4]), "v"(a0.v[7]), "v"(b0.v[3]), "v"(a0.v[8]), "v"(b0.v[2]), "v"(a1.v[2]), "v"(b1.v[8]), "v"(a1.v[3]), "v"(b1.v[7]), "v"(a1.v[4]), "v"(b1.v[6]), "v"(a1.v[5]), "v"(b1.v[5]), "v"(a1.v[6]), "v"(b1.v[4]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a1.v[7]), "v"(b1.v[3]), "v"(a1.v[8]), "v"(b1.v[2]), "v"(m8), "v"(p2), "v"(m7), "v"(p3), "v"(m6), "v"(p4), "v"(m2), "v"(p8));
    r.v[1] = (uint32_t)col & M29; col >>= 29;
    // column 11: 15 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[3]), "v"(b0.v[8]), "v"(a0.v[4]), "v"(b0.v[7]), "v"(a0.v[5]), "v"(b0.v[6]), "v"(a0.v[6]), "v"(b0.v[5]), "v"(a0.v[7]), "v"(b0.v[4]), "v"(a0.v[8]), "v"(b0.v[3]), "v"(a1.v[3]), "v"(b1.v[8]), "v"(a1.v[4]), "v"(b1.v[7]), "v"(a1.v[5]), "v"(b1.v[6]), "v"(a1.v[6]), "v"(b1.v[5]), "v"(a1.v[7]), "v"(b1.v[4]), "v"(a1.v[8]), "v"(b1.v[3]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8), "v"(p3), "v"(m7), "v"(p4), "v"(m3), "v"(p8));
    r.v[2] = (uint32_t)col & M29; col >>= 29;
    // column 12: 12 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[4]), "v"(b0.v[8]), "v"(a0.v[5]), "v"(b0.v[7]), "v"(a0.v[6]), "v"(b0.v[6]), "v"(a0.v[7]), "v"(b0.v[5]), "v"(a0.v[8]), "v"(b0.v[4]), "v"(a1.v[4]), "v"(b1.v[8]), "v"(a1.v[5]), "v"(b1.v[7]), "v"(a1.v[6]), "v"(b1.v[6]), "v"(a1.v[7]), "v"(b1.v[5]), "v"(a1.v[8]), "v"(b1.v[4]), "v"(m8), "v"(p4), "v"(m4), "v"(p8));
    r.v[3] = (uint32_t)col & M29; col >>= 29;
    // column 13: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[5]), "v"(b0.v[8]), "v"(a0.v[6]), "v"(b0.v[7]), "v"(a0.v[7]), "v"(b0.v[6]), "v"(a0.v[8]), "v"(b0.v[5]), "v"(a1.v[5]), "v"(b1.v[8]), "v"(a1.v[6]), "v"(b1.v[7]), "v"(a1.v[7]), "v"(b1.v[6]), "v"(a1.v[8]), "v"(b1.v[5]), "v"(m5), "v"(p8));
    r.v[4] = (uint32_t)col & M29; col >>= 29;
    // column 14: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[6]), "v"(b0.v[8]), "v"(a0.v[7]), "v"(b0.v[7]), "v"(a0.v[8]), "v"(b0.v[6]), "v"(a1.v[6]), "v"(b1.v[8]), "v"(a1.v[7]), "v"(b1.v[7]), "v"(a1.v[8]), "v"(b1.v[6]), "v"(m6), "v"(p8));
    r.v[5] = (uint32_t)col & M29; col >>= 29;
    // column 15: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[7]), "v"(b0.v[8]), "v"(a0.v[8]), "v"(b0.v[7]), "v"(a1.v[7]), "v"(b1.v[8]), "v"(a1.v[8]), "v"(b1.v[7]), "v"(m7), "v"(p8));
    r.v[6] = (uint32_t)col & M29; col >>= 29;
    // column 16: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[8]), "v"(b0.v[8]), "v"(a1.v[8]), "v"(b1.v[8]), "v"(m8), "v"(p8));
    r.v[7] = (uint32_t)col & M29; col >>= 29;
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_sqr_hi_asm(const fe29_t &a, const fe29_t &h) {
    const uint32_t d0 = a.v[0] << 1, d1 = a.v[1] << 1, d2 = a.v[2] << 1, d3 = a.v[3] << 1, d4 = a.v[4] << 1, d5 = a.v[5] << 1, d6 = a.v[6] << 1, d7 = a.v[7] << 1;
    uint64_t col, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;
    // column 0: 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(a.v[0]));
    m0 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m0));
    col >>= 29;
    // column 1: 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[1]), "v"(m0), "v"(p1));
    m1 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1));
    col >>= 29;
    // column 2: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[2]), "v"(a.v[1]), "v"(a.v[1]), "v"(m1), "v"(p1), "v"(m0), "v"(p2));
    m2 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2));
    col >>= 29;
    // column 3: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[3]), "v"(d1), "v"(a.v[2]), "v"(m2), "v"(p1), "v"(m1), "v"(p2), "v"(m0), "v"(p3));
    m3 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3));
    col >>= 29;
    // column 4: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[4]), "v"(d1), "v"(a.v[3]), "v"(a.v[2]), "v"(a.v[2]), "v"(m3), "v"(p1), "v"(m2), "v"(p2), "v"(m1), "v"(p3), "v"(m0), "v"(p4));
    m4 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4));
    col >>= 29;
    // column 5: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[5]), "v"(d1), "v"(a.v[4]), "v"(d2), "v"(a.v[3]), "v"(m4), "v"(p1), "v"(m3), "v"(p2), "v"(m2), "v"(p3), "v"(m1), "v"(p4));
    m5 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5));
    col >>= 29;
    // column 6: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[6]), "v"(d1), "v"(a.v[5]), "v"(d2), "v"(a.v[4]), "v"(a.v[3]), "v"(a.v[3]), "v"(m5), "v"(p1), "v"(m4), "v"(p2), "v"(m3), "v"(p3), "v"(m2), "v"(p4));
    m6 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6));
    col >>= 29;
    // column 7: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[7]), "v"(d1), "v"(a.v[6]), "v"(d2), "v"(a.v[5]), "v"(d3), "v"(a.v[4]), "v"(m6), "v"(p1), "v"(m5), "v"(p2), "v"(m4), "v"(p3), "v"(m3), "v"(p4));
    m7 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m7));
    col >>= 29;
    // column 8: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d0), "v"(a.v[8]), "v"(d1), "v"(a.v[7]), "v"(d2), "v"(a.v[6]), "v"(d3), "v"(a.v[5]), "v"(a.v[4]), "v"(a.v[4]), "v"(m7), "v"(p1), "v"(m6), "v"(p2), "v"(m5), "v"(p3), "v"(m4), "v"(p4), "v"(m0), "v"(p8));
    m8 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8));
    col >>= 29;
    // column 9: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d1), "v"(a.v[8]), "v"(d2), "v"(a.v[7]), "v"(d3), "v"(a.v[6]), "v"(d4), "v"(a.v[5]), "v"(m8), "v"(p1), "v"(m7), "v"(p2), "v"(m6), "v"(p3), "v"(m5), "v"(p4), "v"(m1), "v"(p8), "v"(h.v[0]));
    r.v[0] = (uint32_t)col & M29; col >>= 29;
    // column 10: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d2), "v"(a.v[8]), "v"(d3), "v"(a.v[7]), "v"(d4), "v"(a.v[6]), "v"(a.v[5]), "v"(a.v[5]), "v"(m8), "v"(p2), "v"(m7), "v"(p3), "v"(m6), "v"(p4), "v"(m2), "v"(p8), "v"(h.v[1]));
    r.v[1] = (uint32_t)col & M29; col >>= 29;
    // column 11: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d3), "v"(a.v[8]), "v"(d4), "v"(a.v[7]), "v"(d5), "v"(a.v[6]), "v"(m8), "v"(p3), "v"(m7), "v"(p4), "v"(m3), "v"(p8), "v"(h.v[2]));
    r.v[2] = (uint32_t)col & M29; col >>= 29;
    // column 12: 6 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d4), "v"(a.v[8]), "v"(d5), "v"(a.v[7]), "v"(a.v[6]), "v"(a.v[6]), "v"(m8), "v"(p4), "v"(m4), "v"(p8), "v"(h.v[3]));
    r.v[3] = (uint32_t)col & M29; col >>= 29;
    // column 13: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d5), "v"(a.v[8]), "v"(d6), "v"(a.v[7]), "v"(m5), "v"(p8), "v"(h.v[4]));
    r.v[4] = (uint32_t)col & M29; col >>= 29;
    // column 14: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d6), "v"(a.v[8]), "v"(a.v[7]), "v"(a.v[7]), "v"(m6), "v"(p8), "v"(h.v[5]));
    r.v[5] = (uint32_t)col & M29; col >>= 29;
    // column 15: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d7), "v"(a.v[8]), "v"(m7), "v"(p8), "v"(h.v[6]));
    r.v[6] = (uint32_t)col & M29; col >>= 29;
    // column 16: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(a.v[8]), "v"(m8), "v"(p8), "v"(h.v[7]));
    r.v[7] = (uint32_t)col & M29; col >>= 29;
    r.v[8] = (uint32_t)col + h.v[8];
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_mul_hi_asm(const fe29_t &a, const fe29_t &b, const fe29_t &h) {
    uint64_t col, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;
    // column 0: 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[0]));
    m0 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m0));
    col >>= 29;
    // column 1: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[1]), "v"(b.v[0]), "v"(m0), "v"(p1));
    m1 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1));
    col >>= 29;
    // column 2: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[0]), "v"(m1), "v"(p1), "v"(m0), "v"(p2));
    m2 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2));
    col >>= 29;
    // column 3: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[2]), "v"(b.v[1]), "v"(a.v[3]), "v"(b.v[0]), "v"(m2), "v"(p1), "v"(m1), "v"(p2), "v"(m0), "v"(p3));
    m3 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3));
    col >>= 29;
    // column 4: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[3]), "v"(b.v[1]), "v"(a.v[4]), "v"(b.v[0]), "v"(m3), "v"(p1), "v"(m2), "v"(p2), "v"(m1), "v"(p3), "v"(m0), "v"(p4));
    m4 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4));
    col >>= 29;
    // column 5: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[0]), "v"(m4), "v"(p1), "v"(m3), "v"(p2), "v"(m2), "v"(p3), "v"(m1), "v"(p4));
    m5 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5));
    col >>= 29;
    // column 6: 11 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[0]), "v"(m5), "v"(p1), "v"(m4), "v"(p2), "v"(m3), "v"(p3), "v"(m2), "v"(p4));
    m6 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6));
    col >>= 29;
    // column 7: 12 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0]), "v"(m6), "v"(p1), "v"(m5), "v"(p2), "v"(m4), "v"(p3), "v"(m3), "v"(p4));
    m7 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m7));
    col >>= 29;
    // column 8: 14 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[0]), "v"(m7), "v"(p1), "v"(m6), "v"(p2), "v"(m5), "v"(p3));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4), "v"(p4), "v"(m0), "v"(p8));
    m8 = (0u - (uint32_t)col) & M29;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8));
    col >>= 29;
    // column 9: 14 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[1]), "v"(b.v[8]), "v"(a.v[2]), "v"(b.v[7]), "v"(a.v[3]), "v"(b.v[6]), "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[2]), "v"(a.v[8]), "v"(b.v[1]), "v"(m8), "v"(p1), "v"(m7), "v"(p2), "v"(m6), "v"(p3), "v"(m5), "v"(p4));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1), "v"(p8), "v"(h.v[0]));
    r.v[0] = (uint32_t)col & M29; col >>= 29;
    // column 10: 12 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[2]), "v"(b.v[8]), "v"(a.v[3]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[5]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[2]), "v"(m8), "v"(p2), "v"(m7), "v"(p3), "v"(m6), "v"(p4), "v"(m2), "v"(p8), "v"(h.v[1]));
    r.v[1] = (uint32_t)col & M29; col >>= 29;
    // column 11: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[3]), "v"(b.v[8]), "v"(a.v[4]), "v"(b.v[7]), "v"(a.v[5]), "v"(b.v[6]), "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[7]), "v"(b.v[4]), "v"(a.v[8]), "v"(b.v[3]), "v"(m8), "v"(p3), "v"(m7), "v"(p4), "v"(m3), "v"(p8), "v"(h.v[2]));
    r.v[2] = (uint32_t)col & M29; col >>= 29;
    // column 12: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[4]), "v"(b.v[8]), "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[4]), "v"(m8), "v"(p4), "v"(m4), "v"(p8), "v"(h.v[3]));
    r.v[3] = (uint32_t)col & M29; col >>= 29;
    // column 13: 6 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[5]), "v"(b.v[8]), "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[7]), "v"(b.v[6]), "v"(a.v[8]), "v"(b.v[5]), "v"(m5), "v"(p8), "v"(h.v[4]));
    r.v[4] = (uint32_t)col & M29; col >>= 29;
    // column 14: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[6]), "v"(b.v[8]), "v"(a.v[7]), "v"(b.v[7]), "v"(a.v[8]), "v"(b.v[6]), "v"(m6), "v"(p8), "v"(h.v[5]));
    r.v[5] = (uint32_t)col & M29; col >>= 29;
    // column 15: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[7]), "v"(b.v[8]), "v"(a.v[8]), "v"(b.v[7]), "v"(m7), "v"(p8), "v"(h.v[6]));
    r.v[6] = (uint32_t)col & M29; col >>= 29;
    // column 16: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(b.v[8]), "v"(m8), "v"(p8), "v"(h.v[7]));
    r.v[7] = (uint32_t)col & M29; col >>= 29;
    r.v[8] = (uint32_t)col + h.v[8];
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_mul_hi_lz(const fe29_t &a, const fe29_t &b, const fe29_t &h) {
    uint64_t col, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const uint32_t p1 = P29<F>::L1, p2 = P29<F>::L2, p3 = P29<F>::L3, p4 = P29<F>::L4, p8 = P29<F>::L8;
    // column 0: 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[0]));
    m0 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m0));
    col >>= 29;
    // column 1: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[1]), "v"(b.v[0]), "v"(m0), "v"(p1));
    m1 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1));
    col >>= 29;
    // column 2: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[0]), "v"(m1), "v"(p1), "v"(m0), "v"(p2));
    m2 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m2));
    col >>= 29;
    // column 3: 7 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[2]), "v"(b.v[1]), "v"(a.v[3]), "v"(b.v[0]), "v"(m2), "v"(p1), "v"(m1), "v"(p2), "v"(m0), "v"(p3));
    m3 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3));
    col >>= 29;
    // column 4: 9 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[3]), "v"(b.v[1]), "v"(a.v[4]), "v"(b.v[0]), "v"(m3), "v"(p1), "v"(m2), "v"(p2), "v"(m1), "v"(p3), "v"(m0), "v"(p4));
    m4 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4));
    col >>= 29;
    // column 5: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[0]), "v"(m4), "v"(p1), "v"(m3), "v"(p2), "v"(m2), "v"(p3), "v"(m1), "v"(p4));
    m5 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m5));
    col >>= 29;
    // column 6: 11 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[0]), "v"(m5), "v"(p1), "v"(m4), "v"(p2), "v"(m3), "v"(p3), "v"(m2), "v"(p4));
    m6 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m6));
    col >>= 29;
    // column 7: 12 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0]), "v"(m6), "v"(p1), "v"(m5), "v"(p2), "v"(m4), "v"(p3), "v"(m3), "v"(p4));
    m7 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m7));
    col >>= 29;
    // column 8: 14 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[0]), "v"(m7), "v"(p1), "v"(m6), "v"(p2), "v"(m5), "v"(p3));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m4), "v"(p4), "v"(m0), "v"(p8));
    m8 = 0u - (uint32_t)col;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8));
    col >>= 29;
    // column 9: 14 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[1]), "v"(b.v[8]), "v"(a.v[2]), "v"(b.v[7]), "v"(a.v[3]), "v"(b.v[6]), "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[2]), "v"(a.v[8]), "v"(b.v[1]), "v"(m8), "v"(p1), "v"(m7), "v"(p2), "v"(m6), "v"(p3), "v"(m5), "v"(p4));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m1), "v"(p8), "v"(h.v[0]));
    r.v[0] = (uint32_t)col & M29; col >>= 29;
    // column 10: 12 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[2]), "v"(b.v[8]), "v"(a.v[3]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[5]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[2]), "v"(m8), "v"(p2), "v"(m7), "v"(p3), "v"(m6), "v"(p4), "v"(m2), "v"(p8), "v"(h.v[1]));
    r.v[1] = (uint32_t)col & M29; col >>= 29;
    // column 11: 10 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[3]), "v"(b.v[8]), "v"(a.v[4]), "v"(b.v[7]), "v"(a.v[5]), "v"(b.v[6]), "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[7]), "v"(b.v[4]), "v"(a.v[8]), "v"(b.v[3]), "v"(m8), "v"(p3), "v"(m7), "v"(p4), "v"(m3), "v"(p8), "v"(h.v[2]));
    r.v[2] = (uint32_t)col & M29; col >>= 29;
    // column 12: 8 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[4]), "v"(b.v[8]), "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[4]), "v"(m8), "v"(p4), "v"(m4), "v"(p8), "v"(h.v[3]));
    r.v[3] = (uint32_t)col & M29; col >>= 29;
    // column 13: 6 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[5]), "v"(b.v[8]), "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[7]), "v"(b.v[6]), "v"(a.v[8]), "v"(b.v[5]), "v"(m5), "v"(p8), "v"(h.v[4]));
    r.v[4] = (uint32_t)col & M29; col >>= 29;
    // column 14: 5 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[6]), "v"(b.v[8]), "v"(a.v[7]), "v"(b.v[7]), "v"(a.v[8]), "v"(b.v[6]), "v"(m6), "v"(p8), "v"(h.v[5]));
    r.v[5] = (uint32_t)col & M29; col >>= 29;
    // column 15: 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[7]), "v"(b.v[8]), "v"(a.v[8]), "v"(b.v[7]), "v"(m7), "v"(p8), "v"(h.v[6]));
    r.v[6] = (uint32_t)col & M29; col >>= 29;
    // column 16: 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, 1, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(b.v[8]), "v"(m8), "v"(p8), "v"(h.v[7]));
    r.v[7] = (uint32_t)col & M29; col >>= 29;
    r.v[8] = (uint32_t)col + h.v[8];
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_mul_sg(const fe29_t &a, const fe29_t &b) {
    uint64_t col, nc, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const int32_t n1 = -(int32_t)P29<F>::L1, n2 = -(int32_t)P29<F>::L2, n3 = -(int32_t)P29<F>::L3, n4 = -(int32_t)P29<F>::L4, n8 = -(int32_t)P29<F>::L8;
    // column 0: 1 + 0 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[0]));
    m0 = (uint32_t)col;
    // column 1: 2 + 1 products; first: - s_0 p_0 (the low limb of column 0 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m0), "v"(col), "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[1]), "v"(b.v[0]), "v"(m0), "s"(n1));
    col = nc;
    m1 = (uint32_t)col;
    // column 2: 3 + 2 products; first: - s_1 p_0 (the low limb of column 1 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m1), "v"(col), "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[0]), "v"(m1), "s"(n1), "v"(m0), "s"(n2));
    col = nc;
    m2 = (uint32_t)col;
    // column 3: 4 + 3 products; first: - s_2 p_0 (the low limb of column 2 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m2), "v"(col), "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[2]), "v"(b.v[1]), "v"(a.v[3]), "v"(b.v[0]), "v"(m2), "s"(n1), "v"(m1), "s"(n2), "v"(m0), "s"(n3));
    col = nc;
    m3 = (uint32_t)col;
    // column 4: 5 + 4 products; first: - s_3 p_0 (the low limb of column 3 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m3), "v"(col), "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[3]), "v"(b.v[1]), "v"(a.v[4]), "v"(b.v[0]), "v"(m3), "s"(n1), "v"(m2), "s"(n2), "v"(m1), "s"(n3), "v"(m0), "s"(n4));
    col = nc;
    m4 = (uint32_t)col;
    // column 5: 6 + 4 products; first: - s_4 p_0 (the low limb of column 4 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m4), "v"(col), "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[0]), "v"(m4), "s"(n1), "v"(m3), "s"(n2), "v"(m2), "s"(n3), "v"(m1), "s"(n4));
    col = nc;
    m5 = (uint32_t)col;
    // column 6: 7 + 4 products; first: - s_5 p_0 (the low limb of column 5 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0\n\tv_mad_i64_i32 %0, %1, %24, %25, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m5), "v"(col), "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[0]), "v"(m5), "s"(n1), "v"(m4), "s"(n2), "v"(m3), "s"(n3), "v"(m2), "s"(n4));
    col = nc;
    m6 = (uint32_t)col;
    // column 7: 8 + 4 products; first: - s_6 p_0 (the low limb of column 6 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0\n\tv_mad_i64_i32 %0, %1, %24, %25, %0\n\tv_mad_i64_i32 %0, %1, %26, %27, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m6), "v"(col), "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0]), "v"(m6), "s"(n1), "v"(m5), "s"(n2), "v"(m4), "s"(n3), "v"(m3), "s"(n4));
    col = nc;
    m7 = (uint32_t)col;
    // column 8: 9 + 5 products; first: - s_7 p_0 (the low limb of column 7 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0\n\tv_mad_i64_i32 %0, %1, %24, %25, %0\n\tv_mad_i64_i32 %0, %1, %26, %27, %0\n\tv_mad_i64_i32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m7), "v"(col), "v"(a.v[0]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[0]), "v"(m7), "s"(n1), "v"(m6), "s"(n2), "v"(m5), "s"(n3), "v"(m4), "s"(n4));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(m0), "s"(n8));
    col = nc;
    m8 = ((uint32_t)col & M29) | 0xC0000000u;        // (col & M29) - 2^30: the one digit with a fixed sign
    // column 9: 8 + 5 products; first: - s_8 p_0 (the low limb of column 8 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0\n\tv_mad_i64_i32 %0, %1, %24, %25, %0\n\tv_mad_i64_i32 %0, %1, %26, %27, %0\n\tv_mad_i64_i32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m8), "v"(col), "v"(a.v[1]), "v"(b.v[8]), "v"(a.v[2]), "v"(b.v[7]), "v"(a.v[3]), "v"(b.v[6]), "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[2]), "v"(a.v[8]), "v"(b.v[1]), "v"(m8), "s"(n1), "v"(m7), "s"(n2), "v"(m6), "s"(n3), "v"(m5), "s"(n4), "v"(m1), "s"(n8));
    col = nc;
    r.v[0] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 10: 7 + 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[2]), "v"(b.v[8]), "v"(a.v[3]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[5]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[2]), "v"(m8), "s"(n2), "v"(m7), "s"(n3), "v"(m6), "s"(n4), "v"(m2), "s"(n8));
    r.v[1] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 11: 6 + 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[3]), "v"(b.v[8]), "v"(a.v[4]), "v"(b.v[7]), "v"(a.v[5]), "v"(b.v[6]), "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[7]), "v"(b.v[4]), "v"(a.v[8]), "v"(b.v[3]), "v"(m8), "s"(n3), "v"(m7), "s"(n4), "v"(m3), "s"(n8));
    r.v[2] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 12: 5 + 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[4]), "v"(b.v[8]), "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[4]), "v"(m8), "s"(n4), "v"(m4), "s"(n8));
    r.v[3] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 13: 4 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[5]), "v"(b.v[8]), "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[7]), "v"(b.v[6]), "v"(a.v[8]), "v"(b.v[5]), "v"(m5), "s"(n8));
    r.v[4] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 14: 3 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[6]), "v"(b.v[8]), "v"(a.v[7]), "v"(b.v[7]), "v"(a.v[8]), "v"(b.v[6]), "v"(m6), "s"(n8));
    r.v[5] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 15: 2 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[7]), "v"(b.v[8]), "v"(a.v[8]), "v"(b.v[7]), "v"(m7), "s"(n8));
    r.v[6] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 16: 1 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(b.v[8]), "v"(m8), "s"(n8));
    r.v[7] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_sqr_sg(const fe29_t &a) {
    const uint32_t d0 = a.v[0] << 1, d1 = a.v[1] << 1, d2 = a.v[2] << 1, d3 = a.v[3] << 1, d4 = a.v[4] << 1, d5 = a.v[5] << 1, d6 = a.v[6] << 1, d7 = a.v[7] << 1;
    uint64_t col, nc, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const int32_t n1 = -(int32_t)P29<F>::L1, n2 = -(int32_t)P29<F>::L2, n3 = -(int32_t)P29<F>::L3, n4 = -(int32_t)P29<F>::L4, n8 = -(int32_t)P29<F>::L8;
    // column 0: 1 + 0 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(a.v[0]));
    m0 = (uint32_t)col;
    // column 1: 1 + 1 products; first: - s_0 p_0 (the low limb of column 0 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m0), "v"(col), "v"(d0), "v"(a.v[1]), "v"(m0), "s"(n1));
    col = nc;
    m1 = (uint32_t)col;
    // column 2: 2 + 2 products; first: - s_1 p_0 (the low limb of column 1 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m1), "v"(col), "v"(d0), "v"(a.v[2]), "v"(a.v[1]), "v"(a.v[1]), "v"(m1), "s"(n1), "v"(m0), "s"(n2));
    col = nc;
    m2 = (uint32_t)col;
    // column 3: 2 + 3 products; first: - s_2 p_0 (the low limb of column 2 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m2), "v"(col), "v"(d0), "v"(a.v[3]), "v"(d1), "v"(a.v[2]), "v"(m2), "s"(n1), "v"(m1), "s"(n2), "v"(m0), "s"(n3));
    col = nc;
    m3 = (uint32_t)col;
    // column 4: 3 + 4 products; first: - s_3 p_0 (the low limb of column 3 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m3), "v"(col), "v"(d0), "v"(a.v[4]), "v"(d1), "v"(a.v[3]), "v"(a.v[2]), "v"(a.v[2]), "v"(m3), "s"(n1), "v"(m2), "s"(n2), "v"(m1), "s"(n3), "v"(m0), "s"(n4));
    col = nc;
    m4 = (uint32_t)col;
    // column 5: 3 + 4 products; first: - s_4 p_0 (the low limb of column 4 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m4), "v"(col), "v"(d0), "v"(a.v[5]), "v"(d1), "v"(a.v[4]), "v"(d2), "v"(a.v[3]), "v"(m4), "s"(n1), "v"(m3), "s"(n2), "v"(m2), "s"(n3), "v"(m1), "s"(n4));
    col = nc;
    m5 = (uint32_t)col;
    // column 6: 4 + 4 products; first: - s_5 p_0 (the low limb of column 5 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m5), "v"(col), "v"(d0), "v"(a.v[6]), "v"(d1), "v"(a.v[5]), "v"(d2), "v"(a.v[4]), "v"(a.v[3]), "v"(a.v[3]), "v"(m5), "s"(n1), "v"(m4), "s"(n2), "v"(m3), "s"(n3), "v"(m2), "s"(n4));
    col = nc;
    m6 = (uint32_t)col;
    // column 7: 4 + 4 products; first: - s_6 p_0 (the low limb of column 6 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m6), "v"(col), "v"(d0), "v"(a.v[7]), "v"(d1), "v"(a.v[6]), "v"(d2), "v"(a.v[5]), "v"(d3), "v"(a.v[4]), "v"(m6), "s"(n1), "v"(m5), "s"(n2), "v"(m4), "s"(n3), "v"(m3), "s"(n4));
    col = nc;
    m7 = (uint32_t)col;
    // column 8: 5 + 5 products; first: - s_7 p_0 (the low limb of column 7 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m7), "v"(col), "v"(d0), "v"(a.v[8]), "v"(d1), "v"(a.v[7]), "v"(d2), "v"(a.v[6]), "v"(d3), "v"(a.v[5]), "v"(a.v[4]), "v"(a.v[4]), "v"(m7), "s"(n1), "v"(m6), "s"(n2), "v"(m5), "s"(n3), "v"(m4), "s"(n4), "v"(m0), "s"(n8));
    col = nc;
    m8 = ((uint32_t)col & M29) | 0xC0000000u;        // (col & M29) - 2^30: the one digit with a fixed sign
    // column 9: 4 + 5 products; first: - s_8 p_0 (the low limb of column 8 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m8), "v"(col), "v"(d1), "v"(a.v[8]), "v"(d2), "v"(a.v[7]), "v"(d3), "v"(a.v[6]), "v"(d4), "v"(a.v[5]), "v"(m8), "s"(n1), "v"(m7), "s"(n2), "v"(m6), "s"(n3), "v"(m5), "s"(n4), "v"(m1), "s"(n8));
    col = nc;
    r.v[0] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 10: 4 + 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d2), "v"(a.v[8]), "v"(d3), "v"(a.v[7]), "v"(d4), "v"(a.v[6]), "v"(a.v[5]), "v"(a.v[5]), "v"(m8), "s"(n2), "v"(m7), "s"(n3), "v"(m6), "s"(n4), "v"(m2), "s"(n8));
    r.v[1] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 11: 3 + 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d3), "v"(a.v[8]), "v"(d4), "v"(a.v[7]), "v"(d5), "v"(a.v[6]), "v"(m8), "s"(n3), "v"(m7), "s"(n4), "v"(m3), "s"(n8));
    r.v[2] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 12: 3 + 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d4), "v"(a.v[8]), "v"(d5), "v"(a.v[7]), "v"(a.v[6]), "v"(a.v[6]), "v"(m8), "s"(n4), "v"(m4), "s"(n8));
    r.v[3] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 13: 2 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d5), "v"(a.v[8]), "v"(d6), "v"(a.v[7]), "v"(m5), "s"(n8));
    r.v[4] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 14: 2 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d6), "v"(a.v[8]), "v"(a.v[7]), "v"(a.v[7]), "v"(m6), "s"(n8));
    r.v[5] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 15: 1 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d7), "v"(a.v[8]), "v"(m7), "s"(n8));
    r.v[6] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 16: 1 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(a.v[8]), "v"(m8), "s"(n8));
    r.v[7] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_mul_hi_sg(const fe29_t &a, const fe29_t &b, const fe29_t &h) {
    uint64_t col, nc, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const int32_t n1 = -(int32_t)P29<F>::L1, n2 = -(int32_t)P29<F>::L2, n3 = -(int32_t)P29<F>::L3, n4 = -(int32_t)P29<F>::L4, n8 = -(int32_t)P29<F>::L8;
    // column 0: 1 + 0 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[0]));
    m0 = (uint32_t)col;
    // column 1: 2 + 1 products; first: - s_0 p_0 (the low limb of column 0 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m0), "v"(col), "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[1]), "v"(b.v[0]), "v"(m0), "s"(n1));
    col = nc;
    m1 = (uint32_t)col;
    // column 2: 3 + 2 products; first: - s_1 p_0 (the low limb of column 1 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m1), "v"(col), "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[0]), "v"(m1), "s"(n1), "v"(m0), "s"(n2));
    col = nc;
    m2 = (uint32_t)col;
    // column 3: 4 + 3 products; first: - s_2 p_0 (the low limb of column 2 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m2), "v"(col), "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[2]), "v"(b.v[1]), "v"(a.v[3]), "v"(b.v[0]), "v"(m2), "s"(n1), "v"(m1), "s"(n2), "v"(m0), "s"(n3));
    col = nc;
    m3 = (uint32_t)col;
    // column 4: 5 + 4 products; first: - s_3 p_0 (the low limb of column 3 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m3), "v"(col), "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[3]), "v"(b.v[1]), "v"(a.v[4]), "v"(b.v[0]), "v"(m3), "s"(n1), "v"(m2), "s"(n2), "v"(m1), "s"(n3), "v"(m0), "s"(n4));
    col = nc;
    m4 = (uint32_t)col;
    // column 5: 6 + 4 products; first: - s_4 p_0 (the low limb of column 4 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m4), "v"(col), "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[0]), "v"(m4), "s"(n1), "v"(m3), "s"(n2), "v"(m2), "s"(n3), "v"(m1), "s"(n4));
    col = nc;
    m5 = (uint32_t)col;
    // column 6: 7 + 4 products; first: - s_5 p_0 (the low limb of column 5 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0\n\tv_mad_i64_i32 %0, %1, %24, %25, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m5), "v"(col), "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[0]), "v"(m5), "s"(n1), "v"(m4), "s"(n2), "v"(m3), "s"(n3), "v"(m2), "s"(n4));
    col = nc;
    m6 = (uint32_t)col;
    // column 7: 8 + 4 products; first: - s_6 p_0 (the low limb of column 6 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0\n\tv_mad_i64_i32 %0, %1, %24, %25, %0\n\tv_mad_i64_i32 %0, %1, %26, %27, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m6), "v"(col), "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0]), "v"(m6), "s"(n1), "v"(m5), "s"(n2), "v"(m4), "s"(n3), "v"(m3), "s"(n4));
    col = nc;
    m7 = (uint32_t)col;
    // column 8: 9 + 5 products; first: - s_7 p_0 (the low limb of column 7 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0\n\tv_mad_i64_i32 %0, %1, %24, %25, %0\n\tv_mad_i64_i32 %0, %1, %26, %27, %0\n\tv_mad_i64_i32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m7), "v"(col), "v"(a.v[0]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[0]), "v"(m7), "s"(n1), "v"(m6), "s"(n2), "v"(m5), "s"(n3), "v"(m4), "s"(n4));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(m0), "s"(n8));
    col = nc;
    m8 = ((uint32_t)col & M29) | 0xC0000000u;        // (col & M29) - 2^30: the one digit with a fixed sign
    // column 9: 9 + 5 products; first: - s_8 p_0 (the low limb of column 8 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, 1, %0\n\tv_mad_i64_i32 %0, %1, %21, %22, %0\n\tv_mad_i64_i32 %0, %1, %23, %24, %0\n\tv_mad_i64_i32 %0, %1, %25, %26, %0\n\tv_mad_i64_i32 %0, %1, %27, %28, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m8), "v"(col), "v"(a.v[1]), "v"(b.v[8]), "v"(a.v[2]), "v"(b.v[7]), "v"(a.v[3]), "v"(b.v[6]), "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[2]), "v"(a.v[8]), "v"(b.v[1]), "v"(h.v[0]), "v"(m8), "s"(n1), "v"(m7), "s"(n2), "v"(m6), "s"(n3), "v"(m5), "s"(n4));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(m1), "s"(n8));
    col = nc;
    r.v[0] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 10: 8 + 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, 1, %0\n\tv_mad_i64_i32 %0, %1, %17, %18, %0\n\tv_mad_i64_i32 %0, %1, %19, %20, %0\n\tv_mad_i64_i32 %0, %1, %21, %22, %0\n\tv_mad_i64_i32 %0, %1, %23, %24, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[2]), "v"(b.v[8]), "v"(a.v[3]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[5]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[2]), "v"(h.v[1]), "v"(m8), "s"(n2), "v"(m7), "s"(n3), "v"(m6), "s"(n4), "v"(m2), "s"(n8));
    r.v[1] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 11: 7 + 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, 1, %0\n\tv_mad_i64_i32 %0, %1, %15, %16, %0\n\tv_mad_i64_i32 %0, %1, %17, %18, %0\n\tv_mad_i64_i32 %0, %1, %19, %20, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[3]), "v"(b.v[8]), "v"(a.v[4]), "v"(b.v[7]), "v"(a.v[5]), "v"(b.v[6]), "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[7]), "v"(b.v[4]), "v"(a.v[8]), "v"(b.v[3]), "v"(h.v[2]), "v"(m8), "s"(n3), "v"(m7), "s"(n4), "v"(m3), "s"(n8));
    r.v[2] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 12: 6 + 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, 1, %0\n\tv_mad_i64_i32 %0, %1, %13, %14, %0\n\tv_mad_i64_i32 %0, %1, %15, %16, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[4]), "v"(b.v[8]), "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[4]), "v"(h.v[3]), "v"(m8), "s"(n4), "v"(m4), "s"(n8));
    r.v[3] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 13: 5 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, 1, %0\n\tv_mad_i64_i32 %0, %1, %11, %12, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[5]), "v"(b.v[8]), "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[7]), "v"(b.v[6]), "v"(a.v[8]), "v"(b.v[5]), "v"(h.v[4]), "v"(m5), "s"(n8));
    r.v[4] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 14: 4 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0\n\tv_mad_i64_i32 %0, %1, %9, %10, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[6]), "v"(b.v[8]), "v"(a.v[7]), "v"(b.v[7]), "v"(a.v[8]), "v"(b.v[6]), "v"(h.v[5]), "v"(m6), "s"(n8));
    r.v[5] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 15: 3 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, 1, %0\n\tv_mad_i64_i32 %0, %1, %7, %8, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[7]), "v"(b.v[8]), "v"(a.v[8]), "v"(b.v[7]), "v"(h.v[6]), "v"(m7), "s"(n8));
    r.v[6] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 16: 2 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, 1, %0\n\tv_mad_i64_i32 %0, %1, %5, %6, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(b.v[8]), "v"(h.v[7]), "v"(m8), "s"(n8));
    r.v[7] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    r.v[8] = (uint32_t)col + h.v[8];
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_sqr_hi_sg(const fe29_t &a, const fe29_t &h) {
    const uint32_t d0 = a.v[0] << 1, d1 = a.v[1] << 1, d2 = a.v[2] << 1, d3 = a.v[3] << 1, d4 = a.v[4] << 1, d5 = a.v[5] << 1, d6 = a.v[6] << 1, d7 = a.v[7] << 1;
    uint64_t col, nc, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const int32_t n1 = -(int32_t)P29<F>::L1, n2 = -(int32_t)P29<F>::L2, n3 = -(int32_t)P29<F>::L3, n4 = -(int32_t)P29<F>::L4, n8 = -(int32_t)P29<F>::L8;
    // column 0: 1 + 0 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(a.v[0]));
    m0 = (uint32_t)col;
    // column 1: 1 + 1 products; first: - s_0 p_0 (the low limb of column 0 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m0), "v"(col), "v"(d0), "v"(a.v[1]), "v"(m0), "s"(n1));
    col = nc;
    m1 = (uint32_t)col;
    // column 2: 2 + 2 products; first: - s_1 p_0 (the low limb of column 1 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m1), "v"(col), "v"(d0), "v"(a.v[2]), "v"(a.v[1]), "v"(a.v[1]), "v"(m1), "s"(n1), "v"(m0), "s"(n2));
    col = nc;
    m2 = (uint32_t)col;
    // column 3: 2 + 3 products; first: - s_2 p_0 (the low limb of column 2 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m2), "v"(col), "v"(d0), "v"(a.v[3]), "v"(d1), "v"(a.v[2]), "v"(m2), "s"(n1), "v"(m1), "s"(n2), "v"(m0), "s"(n3));
    col = nc;
    m3 = (uint32_t)col;
    // column 4: 3 + 4 products; first: - s_3 p_0 (the low limb of column 3 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m3), "v"(col), "v"(d0), "v"(a.v[4]), "v"(d1), "v"(a.v[3]), "v"(a.v[2]), "v"(a.v[2]), "v"(m3), "s"(n1), "v"(m2), "s"(n2), "v"(m1), "s"(n3), "v"(m0), "s"(n4));
    col = nc;
    m4 = (uint32_t)col;
    // column 5: 3 + 4 products; first: - s_4 p_0 (the low limb of column 4 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m4), "v"(col), "v"(d0), "v"(a.v[5]), "v"(d1), "v"(a.v[4]), "v"(d2), "v"(a.v[3]), "v"(m4), "s"(n1), "v"(m3), "s"(n2), "v"(m2), "s"(n3), "v"(m1), "s"(n4));
    col = nc;
    m5 = (uint32_t)col;
    // column 6: 4 + 4 products; first: - s_5 p_0 (the low limb of column 5 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m5), "v"(col), "v"(d0), "v"(a.v[6]), "v"(d1), "v"(a.v[5]), "v"(d2), "v"(a.v[4]), "v"(a.v[3]), "v"(a.v[3]), "v"(m5), "s"(n1), "v"(m4), "s"(n2), "v"(m3), "s"(n3), "v"(m2), "s"(n4));
    col = nc;
    m6 = (uint32_t)col;
    // column 7: 4 + 4 products; first: - s_6 p_0 (the low limb of column 6 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m6), "v"(col), "v"(d0), "v"(a.v[7]), "v"(d1), "v"(a.v[6]), "v"(d2), "v"(a.v[5]), "v"(d3), "v"(a.v[4]), "v"(m6), "s"(n1), "v"(m5), "s"(n2), "v"(m4), "s"(n3), "v"(m3), "s"(n4));
    col = nc;
    m7 = (uint32_t)col;
    // column 8: 5 + 5 products; first: - s_7 p_0 (the low limb of column 7 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m7), "v"(col), "v"(d0), "v"(a.v[8]), "v"(d1), "v"(a.v[7]), "v"(d2), "v"(a.v[6]), "v"(d3), "v"(a.v[5]), "v"(a.v[4]), "v"(a.v[4]), "v"(m7), "s"(n1), "v"(m6), "s"(n2), "v"(m5), "s"(n3), "v"(m4), "s"(n4), "v"(m0), "s"(n8));
    col = nc;
    m8 = ((uint32_t)col & M29) | 0xC0000000u;        // (col & M29) - 2^30: the one digit with a fixed sign
    // column 9: 5 + 5 products; first: - s_8 p_0 (the low limb of column 8 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, 1, %0\n\tv_mad_i64_i32 %0, %1, %13, %14, %0\n\tv_mad_i64_i32 %0, %1, %15, %16, %0\n\tv_mad_i64_i32 %0, %1, %17, %18, %0\n\tv_mad_i64_i32 %0, %1, %19, %20, %0\n\tv_mad_i64_i32 %0, %1, %21, %22, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m8), "v"(col), "v"(d1), "v"(a.v[8]), "v"(d2), "v"(a.v[7]), "v"(d3), "v"(a.v[6]), "v"(d4), "v"(a.v[5]), "v"(h.v[0]), "v"(m8), "s"(n1), "v"(m7), "s"(n2), "v"(m6), "s"(n3), "v"(m5), "s"(n4), "v"(m1), "s"(n8));
    col = nc;
    r.v[0] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 10: 5 + 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, 1, %0\n\tv_mad_i64_i32 %0, %1, %11, %12, %0\n\tv_mad_i64_i32 %0, %1, %13, %14, %0\n\tv_mad_i64_i32 %0, %1, %15, %16, %0\n\tv_mad_i64_i32 %0, %1, %17, %18, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d2), "v"(a.v[8]), "v"(d3), "v"(a.v[7]), "v"(d4), "v"(a.v[6]), "v"(a.v[5]), "v"(a.v[5]), "v"(h.v[1]), "v"(m8), "s"(n2), "v"(m7), "s"(n3), "v"(m6), "s"(n4), "v"(m2), "s"(n8));
    r.v[1] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 11: 4 + 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0\n\tv_mad_i64_i32 %0, %1, %9, %10, %0\n\tv_mad_i64_i32 %0, %1, %11, %12, %0\n\tv_mad_i64_i32 %0, %1, %13, %14, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d3), "v"(a.v[8]), "v"(d4), "v"(a.v[7]), "v"(d5), "v"(a.v[6]), "v"(h.v[2]), "v"(m8), "s"(n3), "v"(m7), "s"(n4), "v"(m3), "s"(n8));
    r.v[2] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 12: 4 + 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0\n\tv_mad_i64_i32 %0, %1, %9, %10, %0\n\tv_mad_i64_i32 %0, %1, %11, %12, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d4), "v"(a.v[8]), "v"(d5), "v"(a.v[7]), "v"(a.v[6]), "v"(a.v[6]), "v"(h.v[3]), "v"(m8), "s"(n4), "v"(m4), "s"(n8));
    r.v[3] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 13: 3 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, 1, %0\n\tv_mad_i64_i32 %0, %1, %7, %8, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d5), "v"(a.v[8]), "v"(d6), "v"(a.v[7]), "v"(h.v[4]), "v"(m5), "s"(n8));
    r.v[4] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 14: 3 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, 1, %0\n\tv_mad_i64_i32 %0, %1, %7, %8, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d6), "v"(a.v[8]), "v"(a.v[7]), "v"(a.v[7]), "v"(h.v[5]), "v"(m6), "s"(n8));
    r.v[5] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 15: 2 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, 1, %0\n\tv_mad_i64_i32 %0, %1, %5, %6, %0"
        : "+&v"(col), "=&s"(cc) : "v"(d7), "v"(a.v[8]), "v"(h.v[6]), "v"(m7), "s"(n8));
    r.v[6] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 16: 2 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, 1, %0\n\tv_mad_i64_i32 %0, %1, %5, %6, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(a.v[8]), "v"(h.v[7]), "v"(m8), "s"(n8));
    r.v[7] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    r.v[8] = (uint32_t)col + h.v[8];
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_mulrc_sg(const fe29_t &a, const fe29_t &b, const fe29_t &c) {
    uint64_t col, nc, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const int32_t n1 = -(int32_t)P29<F>::L1, n2 = -(int32_t)P29<F>::L2, n3 = -(int32_t)P29<F>::L3, n4 = -(int32_t)P29<F>::L4, n8 = -(int32_t)P29<F>::L8;
    // column 0: 2 + 0 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, 1, %0"
        : "=&v"(col), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[0]), "v"(c.v[0]));
    m0 = (uint32_t)col;
    // column 1: 3 + 1 products; first: - s_0 p_0 (the low limb of column 0 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0\n\tv_mad_i64_i32 %0, %1, %9, %10, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m0), "v"(col), "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[1]), "v"(b.v[0]), "v"(c.v[1]), "v"(m0), "s"(n1));
    col = nc;
    m1 = (uint32_t)col;
    // column 2: 4 + 2 products; first: - s_1 p_0 (the low limb of column 1 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, 1, %0\n\tv_mad_i64_i32 %0, %1, %11, %12, %0\n\tv_mad_i64_i32 %0, %1, %13, %14, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m1), "v"(col), "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[0]), "v"(c.v[2]), "v"(m1), "s"(n1), "v"(m0), "s"(n2));
    col = nc;
    m2 = (uint32_t)col;
    // column 3: 5 + 3 products; first: - s_2 p_0 (the low limb of column 2 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, 1, %0\n\tv_mad_i64_i32 %0, %1, %13, %14, %0\n\tv_mad_i64_i32 %0, %1, %15, %16, %0\n\tv_mad_i64_i32 %0, %1, %17, %18, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m2), "v"(col), "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[2]), "v"(b.v[1]), "v"(a.v[3]), "v"(b.v[0]), "v"(c.v[3]), "v"(m2), "s"(n1), "v"(m1), "s"(n2), "v"(m0), "s"(n3));
    col = nc;
    m3 = (uint32_t)col;
    // column 4: 6 + 4 products; first: - s_3 p_0 (the low limb of column 3 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, 1, %0\n\tv_mad_i64_i32 %0, %1, %15, %16, %0\n\tv_mad_i64_i32 %0, %1, %17, %18, %0\n\tv_mad_i64_i32 %0, %1, %19, %20, %0\n\tv_mad_i64_i32 %0, %1, %21, %22, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m3), "v"(col), "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[3]), "v"(b.v[1]), "v"(a.v[4]), "v"(b.v[0]), "v"(c.v[4]), "v"(m3), "s"(n1), "v"(m2), "s"(n2), "v"(m1), "s"(n3), "v"(m0), "s"(n4));
    col = nc;
    m4 = (uint32_t)col;
    // column 5: 7 + 4 products; first: - s_4 p_0 (the low limb of column 4 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, 1, %0\n\tv_mad_i64_i32 %0, %1, %17, %18, %0\n\tv_mad_i64_i32 %0, %1, %19, %20, %0\n\tv_mad_i64_i32 %0, %1, %21, %22, %0\n\tv_mad_i64_i32 %0, %1, %23, %24, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m4), "v"(col), "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[0]), "v"(c.v[5]), "v"(m4), "s"(n1), "v"(m3), "s"(n2), "v"(m2), "s"(n3), "v"(m1), "s"(n4));
    col = nc;
    m5 = (uint32_t)col;
    // column 6: 8 + 4 products; first: - s_5 p_0 (the low limb of column 5 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, 1, %0\n\tv_mad_i64_i32 %0, %1, %19, %20, %0\n\tv_mad_i64_i32 %0, %1, %21, %22, %0\n\tv_mad_i64_i32 %0, %1, %23, %24, %0\n\tv_mad_i64_i32 %0, %1, %25, %26, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m5), "v"(col), "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[0]), "v"(c.v[6]), "v"(m5), "s"(n1), "v"(m4), "s"(n2), "v"(m3), "s"(n3), "v"(m2), "s"(n4));
    col = nc;
    m6 = (uint32_t)col;
    // column 7: 9 + 4 products; first: - s_6 p_0 (the low limb of column 6 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, 1, %0\n\tv_mad_i64_i32 %0, %1, %21, %22, %0\n\tv_mad_i64_i32 %0, %1, %23, %24, %0\n\tv_mad_i64_i32 %0, %1, %25, %26, %0\n\tv_mad_i64_i32 %0, %1, %27, %28, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m6), "v"(col), "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0]), "v"(c.v[7]), "v"(m6), "s"(n1), "v"(m5), "s"(n2), "v"(m4), "s"(n3), "v"(m3), "s"(n4));
    col = nc;
    m7 = (uint32_t)col;
    // column 8: 10 + 5 products; first: - s_7 p_0 (the low limb of column 7 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, 1, %0\n\tv_mad_i64_i32 %0, %1, %23, %24, %0\n\tv_mad_i64_i32 %0, %1, %25, %26, %0\n\tv_mad_i64_i32 %0, %1, %27, %28, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m7), "v"(col), "v"(a.v[0]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[0]), "v"(c.v[8]), "v"(m7), "s"(n1), "v"(m6), "s"(n2), "v"(m5), "s"(n3));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(m4), "s"(n4), "v"(m0), "s"(n8));
    col = nc;
    m8 = ((uint32_t)col & M29) | 0xC0000000u;        // (col & M29) - 2^30: the one digit with a fixed sign
    // column 9: 8 + 5 products; first: - s_8 p_0 (the low limb of column 8 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0\n\tv_mad_i64_i32 %0, %1, %24, %25, %0\n\tv_mad_i64_i32 %0, %1, %26, %27, %0\n\tv_mad_i64_i32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m8), "v"(col), "v"(a.v[1]), "v"(b.v[8]), "v"(a.v[2]), "v"(b.v[7]), "v"(a.v[3]), "v"(b.v[6]), "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[2]), "v"(a.v[8]), "v"(b.v[1]), "v"(m8), "s"(n1), "v"(m7), "s"(n2), "v"(m6), "s"(n3), "v"(m5), "s"(n4), "v"(m1), "s"(n8));
    col = nc;
    r.v[0] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 10: 7 + 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[2]), "v"(b.v[8]), "v"(a.v[3]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[5]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[2]), "v"(m8), "s"(n2), "v"(m7), "s"(n3), "v"(m6), "s"(n4), "v"(m2), "s"(n8));
    r.v[1] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 11: 6 + 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[3]), "v"(b.v[8]), "v"(a.v[4]), "v"(b.v[7]), "v"(a.v[5]), "v"(b.v[6]), "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[7]), "v"(b.v[4]), "v"(a.v[8]), "v"(b.v[3]), "v"(m8), "s"(n3), "v"(m7), "s"(n4), "v"(m3), "s"(n8));
    r.v[2] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 12: 5 + 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[4]), "v"(b.v[8]), "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[4]), "v"(m8), "s"(n4), "v"(m4), "s"(n8));
    r.v[3] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 13: 4 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[5]), "v"(b.v[8]), "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[7]), "v"(b.v[6]), "v"(a.v[8]), "v"(b.v[5]), "v"(m5), "s"(n8));
    r.v[4] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 14: 3 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[6]), "v"(b.v[8]), "v"(a.v[7]), "v"(b.v[7]), "v"(a.v[8]), "v"(b.v[6]), "v"(m6), "s"(n8));
    r.v[5] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 15: 2 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[7]), "v"(b.v[8]), "v"(a.v[8]), "v"(b.v[7]), "v"(m7), "s"(n8));
    r.v[6] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 16: 1 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a.v[8]), "v"(b.v[8]), "v"(m8), "s"(n8));
    r.v[7] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_dot2rc_sg(const fe29_t &a0, const fe29_t &b0, const fe29_t &a1, const fe29_t &b1, const fe29_t &c) {
    uint64_t col, nc, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const int32_t n1 = -(int32_t)P29<F>::L1, n2 = -(int32_t)P29<F>::L2, n3 = -(int32_t)P29<F>::L3, n4 = -(int32_t)P29<F>::L4, n8 = -(int32_t)P29<F>::L8;
    // column 0: 3 + 0 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, 1, %0"
        : "=&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[0]), "v"(c.v[0]));
    m0 = (uint32_t)col;
    // column 1: 5 + 1 products; first: - s_0 p_0 (the low limb of column 0 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, 1, %0\n\tv_mad_i64_i32 %0, %1, %13, %14, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m0), "v"(col), "v"(a0.v[0]), "v"(b0.v[1]), "v"(a0.v[1]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[1]), "v"(a1.v[1]), "v"(b1.v[0]), "v"(c.v[1]), "v"(m0), "s"(n1));
    col = nc;
    m1 = (uint32_t)col;
    // column 2: 7 + 2 products; first: - s_1 p_0 (the low limb of column 1 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, 1, %0\n\tv_mad_i64_i32 %0, %1, %17, %18, %0\n\tv_mad_i64_i32 %0, %1, %19, %20, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m1), "v"(col), "v"(a0.v[0]), "v"(b0.v[2]), "v"(a0.v[1]), "v"(b0.v[1]), "v"(a0.v[2]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[2]), "v"(a1.v[1]), "v"(b1.v[1]), "v"(a1.v[2]), "v"(b1.v[0]), "v"(c.v[2]), "v"(m1), "s"(n1), "v"(m0), "s"(n2));
    col = nc;
    m2 = (uint32_t)col;
    // column 3: 9 + 3 products; first: - s_2 p_0 (the low limb of column 2 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, 1, %0\n\tv_mad_i64_i32 %0, %1, %21, %22, %0\n\tv_mad_i64_i32 %0, %1, %23, %24, %0\n\tv_mad_i64_i32 %0, %1, %25, %26, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m2), "v"(col), "v"(a0.v[0]), "v"(b0.v[3]), "v"(a0.v[1]), "v"(b0.v[2]), "v"(a0.v[2]), "v"(b0.v[1]), "v"(a0.v[3]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[3]), "v"(a1.v[1]), "v"(b1.v[2]), "v"(a1.v[2]), "v"(b1.v[1]), "v"(a1.v[3]), "v"(b1.v[0]), "v"(c.v[3]), "v"(m2), "s"(n1), "v"(m1), "s"(n2), "v"(m0), "s"(n3));
    col = nc;
    m3 = (uint32_t)col;
    // column 4: 11 + 4 products; first: - s_3 p_0 (the low limb of column 3 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, 1, %0\n\tv_mad_i64_i32 %0, %1, %25, %26, %0\n\tv_mad_i64_i32 %0, %1, %27, %28, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m3), "v"(col), "v"(a0.v[0]), "v"(b0.v[4]), "v"(a0.v[1]), "v"(b0.v[3]), "v"(a0.v[2]), "v"(b0.v[2]), "v"(a0.v[3]), "v"(b0.v[1]), "v"(a0.v[4]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[4]), "v"(a1.v[1]), "v"(b1.v[3]), "v"(a1.v[2]), "v"(b1.v[2]), "v"(a1.v[3]), "v"(b1.v[1]), "v"(a1.v[4]), "v"(b1.v[0]), "v"(c.v[4]), "v"(m3), "s"(n1), "v"(m2), "s"(n2));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(m1), "s"(n3), "v"(m0), "s"(n4));
    col = nc;
    m4 = (uint32_t)col;
    // column 5: 13 + 4 products; first: - s_4 p_0 (the low limb of column 4 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, 1, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m4), "v"(col), "v"(a0.v[0]), "v"(b0.v[5]), "v"(a0.v[1]), "v"(b0.v[4]), "v"(a0.v[2]), "v"(b0.v[3]), "v"(a0.v[3]), "v"(b0.v[2]), "v"(a0.v[4]), "v"(b0.v[1]), "v"(a0.v[5]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[5]), "v"(a1.v[1]), "v"(b1.v[4]), "v"(a1.v[2]), "v"(b1.v[3]), "v"(a1.v[3]), "v"(b1.v[2]), "v"(a1.v[4]), "v"(b1.v[1]), "v"(a1.v[5]), "v"(b1.v[0]), "v"(c.v[5]));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(m4), "s"(n1), "v"(m3), "s"(n2), "v"(m2), "s"(n3), "v"(m1), "s"(n4));
    col = nc;
    m5 = (uint32_t)col;
    // column 6: 15 + 4 products; first: - s_5 p_0 (the low limb of column 5 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m5), "v"(col), "v"(a0.v[0]), "v"(b0.v[6]), "v"(a0.v[1]), "v"(b0.v[5]), "v"(a0.v[2]), "v"(b0.v[4]), "v"(a0.v[3]), "v"(b0.v[3]), "v"(a0.v[4]), "v"(b0.v[2]), "v"(a0.v[5]), "v"(b0.v[1]), "v"(a0.v[6]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[6]), "v"(a1.v[1]), "v"(b1.v[5]), "v"(a1.v[2]), "v"(b1.v[4]), "v"(a1.v[3]), "v"(b1.v[3]), "v"(a1.v[4]), "v"(b1.v[2]), "v"(a1.v[5]), "v"(b1.v[1]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, 1, %0\n\tv_mad_i64_i32 %0, %1, %5, %6, %0\n\tv_mad_i64_i32 %0, %1, %7, %8, %0\n\tv_mad_i64_i32 %0, %1, %9, %10, %0\n\tv_mad_i64_i32 %0, %1, %11, %12, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(a1.v[6]), "v"(b1.v[0]), "v"(c.v[6]), "v"(m5), "s"(n1), "v"(m4), "s"(n2), "v"(m3), "s"(n3), "v"(m2), "s"(n4));
    col = nc;
    m6 = (uint32_t)col;
    // column 7: 17 + 4 products; first: - s_6 p_0 (the low limb of column 6 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m6), "v"(col), "v"(a0.v[0]), "v"(b0.v[7]), "v"(a0.v[1]), "v"(b0.v[6]), "v"(a0.v[2]), "v"(b0.v[5]), "v"(a0.v[3]), "v"(b0.v[4]), "v"(a0.v[4]), "v"(b0.v[3]), "v"(a0.v[5]), "v"(b0.v[2]), "v"(a0.v[6]), "v"(b0.v[1]), "v"(a0.v[7]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[7]), "v"(a1.v[1]), "v"(b1.v[6]), "v"(a1.v[2]), "v"(b1.v[5]), "v"(a1.v[3]), "v"(b1.v[4]), "v"(a1.v[4]), "v"(b1.v[3]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0\n\tv_mad_i64_i32 %0, %1, %9, %10, %0\n\tv_mad_i64_i32 %0, %1, %11, %12, %0\n\tv_mad_i64_i32 %0, %1, %13, %14, %0\n\tv_mad_i64_i32 %0, %1, %15, %16, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(a1.v[5]), "v"(b1.v[2]), "v"(a1.v[6]), "v"(b1.v[1]), "v"(a1.v[7]), "v"(b1.v[0]), "v"(c.v[7]), "v"(m6), "s"(n1), "v"(m5), "s"(n2), "v"(m4), "s"(n3), "v"(m3), "s"(n4));
    col = nc;
    m7 = (uint32_t)col;
    // column 8: 19 + 5 products; first: - s_7 p_0 (the low limb of column 7 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m7), "v"(col), "v"(a0.v[0]), "v"(b0.v[8]), "v"(a0.v[1]), "v"(b0.v[7]), "v"(a0.v[2]), "v"(b0.v[6]), "v"(a0.v[3]), "v"(b0.v[5]), "v"(a0.v[4]), "v"(b0.v[4]), "v"(a0.v[5]), "v"(b0.v[3]), "v"(a0.v[6]), "v"(b0.v[2]), "v"(a0.v[7]), "v"(b0.v[1]), "v"(a0.v[8]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[8]), "v"(a1.v[1]), "v"(b1.v[7]), "v"(a1.v[2]), "v"(b1.v[6]), "v"(a1.v[3]), "v"(b1.v[5]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, 1, %0\n\tv_mad_i64_i32 %0, %1, %13, %14, %0\n\tv_mad_i64_i32 %0, %1, %15, %16, %0\n\tv_mad_i64_i32 %0, %1, %17, %18, %0\n\tv_mad_i64_i32 %0, %1, %19, %20, %0\n\tv_mad_i64_i32 %0, %1, %21, %22, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(a1.v[4]), "v"(b1.v[4]), "v"(a1.v[5]), "v"(b1.v[3]), "v"(a1.v[6]), "v"(b1.v[2]), "v"(a1.v[7]), "v"(b1.v[1]), "v"(a1.v[8]), "v"(b1.v[0]), "v"(c.v[8]), "v"(m7), "s"(n1), "v"(m6), "s"(n2), "v"(m5), "s"(n3), "v"(m4), "s"(n4), "v"(m0), "s"(n8));
    col = nc;
    m8 = ((uint32_t)col & M29) | 0xC0000000u;        // (col & M29) - 2^30: the one digit with a fixed sign
    // column 9: 16 + 5 products; first: - s_8 p_0 (the low limb of column 8 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m8), "v"(col), "v"(a0.v[1]), "v"(b0.v[8]), "v"(a0.v[2]), "v"(b0.v[7]), "v"(a0.v[3]), "v"(b0.v[6]), "v"(a0.v[4]), "v"(b0.v[5]), "v"(a0.v[5]), "v"(b0.v[4]), "v"(a0.v[6]), "v"(b0.v[3]), "v"(a0.v[7]), "v"(b0.v[2]), "v"(a0.v[8]), "v"(b0.v[1]), "v"(a1.v[1]), "v"(b1.v[8]), "v"(a1.v[2]), "v"(b1.v[7]), "v"(a1.v[3]), "v"(b1.v[6]), "v"(a1.v[4]), "v"(b1.v[5]), "v"(a1.v[5]), "v"(b1.v[4]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(a1.v[6]), "v"(b1.v[3]), "v"(a1.v[7]), "v"(b1.v[2]), "v"(a1.v[8]), "v"(b1.v[1]), "v"(m8), "s"(n1), "v"(m7), "s"(n2), "v"(m6), "s"(n3), "v"(m5), "s"(n4), "v"(m1), "s"(n8));
    col = nc;
    r.v[0] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 10: 14 + 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[2]), "v"(b0.v[8]), "v"(a0.v[3]), "v"(b0.v[7]), "v"(a0.v[4]), "v"(b0.v[6]), "v"(a0.v[5]), "v"(b0.v[5]), "v"(a0.v[6]), "v"(b0.v[4]), "v"(a0.v[7]), "v"(b0.v[3]), "v"(a0.v[8]), "v"(b0.v[2]), "v"(a1.v[2]), "v"(b1.v[8]), "v"(a1.v[3]), "v"(b1.v[7]), "v"(a1.v[4]), "v"(b1.v[6]), "v"(a1.v[5]), "v"(b1.v[5]), "v"(a1.v[6]), "v"(b1.v[4]), "v"(a1.v[7]), "v"(b1.v[3]), "v"(a1.v[8]), "v"(b1.v[2]));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m8), "s"(n2), "v"(m7), "s"(n3), "v"(m6), "s"(n4), "v"(m2), "s"(n8));
    r.v[1] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 11: 12 + 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_i64_i32 %0, %1, %26, %27, %0\n\tv_mad_i64_i32 %0, %1, %28, %29, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[3]), "v"(b0.v[8]), "v"(a0.v[4]), "v"(b0.v[7]), "v"(a0.v[5]), "v"(b0.v[6]), "v"(a0.v[6]), "v"(b0.v[5]), "v"(a0.v[7]), "v"(b0.v[4]), "v"(a0.v[8]), "v"(b0.v[3]), "v"(a1.v[3]), "v"(b1.v[8]), "v"(a1.v[4]), "v"(b1.v[7]), "v"(a1.v[5]), "v"(b1.v[6]), "v"(a1.v[6]), "v"(b1.v[5]), "v"(a1.v[7]), "v"(b1.v[4]), "v"(a1.v[8]), "v"(b1.v[3]), "v"(m8), "s"(n3), "v"(m7), "s"(n4));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0"
        : "+&v"(col), "=&s"(cc) : "v"(m3), "s"(n8));
    r.v[2] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 12: 10 + 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0\n\tv_mad_i64_i32 %0, %1, %24, %25, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[4]), "v"(b0.v[8]), "v"(a0.v[5]), "v"(b0.v[7]), "v"(a0.v[6]), "v"(b0.v[6]), "v"(a0.v[7]), "v"(b0.v[5]), "v"(a0.v[8]), "v"(b0.v[4]), "v"(a1.v[4]), "v"(b1.v[8]), "v"(a1.v[5]), "v"(b1.v[7]), "v"(a1.v[6]), "v"(b1.v[6]), "v"(a1.v[7]), "v"(b1.v[5]), "v"(a1.v[8]), "v"(b1.v[4]), "v"(m8), "s"(n4), "v"(m4), "s"(n8));
    r.v[3] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 13: 8 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[5]), "v"(b0.v[8]), "v"(a0.v[6]), "v"(b0.v[7]), "v"(a0.v[7]), "v"(b0.v[6]), "v"(a0.v[8]), "v"(b0.v[5]), "v"(a1.v[5]), "v"(b1.v[8]), "v"(a1.v[6]), "v"(b1.v[7]), "v"(a1.v[7]), "v"(b1.v[6]), "v"(a1.v[8]), "v"(b1.v[5]), "v"(m5), "s"(n8));
    r.v[4] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 14: 6 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[6]), "v"(b0.v[8]), "v"(a0.v[7]), "v"(b0.v[7]), "v"(a0.v[8]), "v"(b0.v[6]), "v"(a1.v[6]), "v"(b1.v[8]), "v"(a1.v[7]), "v"(b1.v[7]), "v"(a1.v[8]), "v"(b1.v[6]), "v"(m6), "s"(n8));
    r.v[5] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 15: 4 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[7]), "v"(b0.v[8]), "v"(a0.v[8]), "v"(b0.v[7]), "v"(a1.v[7]), "v"(b1.v[8]), "v"(a1.v[8]), "v"(b1.v[7]), "v"(m7), "s"(n8));
    r.v[6] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 16: 2 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[8]), "v"(b0.v[8]), "v"(a1.v[8]), "v"(b1.v[8]), "v"(m8), "s"(n8));
    r.v[7] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    r.v[8] = (uint32_t)col;
    return r;
}
template <int F> __device__ __forceinline__ fe29_t fe29_dot3rc_sg(const fe29_t &a0, const fe29_t &b0, const fe29_t &a1, const fe29_t &b1, const fe29_t &a2, const fe29_t &b2, const fe29_t &c) {
    uint64_t col, nc, cc; fe29_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    const int32_t n1 = -(int32_t)P29<F>::L1, n2 = -(int32_t)P29<F>::L2, n3 = -(int32_t)P29<F>::L3, n4 = -(int32_t)P29<F>::L4, n8 = -(int32_t)P29<F>::L8;
    // column 0: 4 + 0 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, 1, %0"
        : "=&v"(col), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[0]), "v"(c.v[0]));
    m0 = (uint32_t)col;
    // column 1: 7 + 1 products; first: - s_0 p_0 (the low limb of column 0 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, 1, %0\n\tv_mad_i64_i32 %0, %1, %17, %18, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m0), "v"(col), "v"(a0.v[0]), "v"(b0.v[1]), "v"(a0.v[1]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[1]), "v"(a1.v[1]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[1]), "v"(a2.v[1]), "v"(b2.v[0]), "v"(c.v[1]), "v"(m0), "s"(n1));
    col = nc;
    m1 = (uint32_t)col;
    // column 2: 10 + 2 products; first: - s_1 p_0 (the low limb of column 1 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, 1, %0\n\tv_mad_i64_i32 %0, %1, %23, %24, %0\n\tv_mad_i64_i32 %0, %1, %25, %26, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m1), "v"(col), "v"(a0.v[0]), "v"(b0.v[2]), "v"(a0.v[1]), "v"(b0.v[1]), "v"(a0.v[2]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[2]), "v"(a1.v[1]), "v"(b1.v[1]), "v"(a1.v[2]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[2]), "v"(a2.v[1]), "v"(b2.v[1]), "v"(a2.v[2]), "v"(b2.v[0]), "v"(c.v[2]), "v"(m1), "s"(n1), "v"(m0), "s"(n2));
    col = nc;
    m2 = (uint32_t)col;
    // column 3: 13 + 3 products; first: - s_2 p_0 (the low limb of column 2 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, 1, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m2), "v"(col), "v"(a0.v[0]), "v"(b0.v[3]), "v"(a0.v[1]), "v"(b0.v[2]), "v"(a0.v[2]), "v"(b0.v[1]), "v"(a0.v[3]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[3]), "v"(a1.v[1]), "v"(b1.v[2]), "v"(a1.v[2]), "v"(b1.v[1]), "v"(a1.v[3]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[3]), "v"(a2.v[1]), "v"(b2.v[2]), "v"(a2.v[2]), "v"(b2.v[1]), "v"(a2.v[3]), "v"(b2.v[0]), "v"(c.v[3]));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(m2), "s"(n1), "v"(m1), "s"(n2), "v"(m0), "s"(n3));
    col = nc;
    m3 = (uint32_t)col;
    // column 4: 16 + 4 products; first: - s_3 p_0 (the low limb of column 3 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m3), "v"(col), "v"(a0.v[0]), "v"(b0.v[4]), "v"(a0.v[1]), "v"(b0.v[3]), "v"(a0.v[2]), "v"(b0.v[2]), "v"(a0.v[3]), "v"(b0.v[1]), "v"(a0.v[4]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[4]), "v"(a1.v[1]), "v"(b1.v[3]), "v"(a1.v[2]), "v"(b1.v[2]), "v"(a1.v[3]), "v"(b1.v[1]), "v"(a1.v[4]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[4]), "v"(a2.v[1]), "v"(b2.v[3]), "v"(a2.v[2]), "v"(b2.v[2]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, 1, %0\n\tv_mad_i64_i32 %0, %1, %7, %8, %0\n\tv_mad_i64_i32 %0, %1, %9, %10, %0\n\tv_mad_i64_i32 %0, %1, %11, %12, %0\n\tv_mad_i64_i32 %0, %1, %13, %14, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(a2.v[3]), "v"(b2.v[1]), "v"(a2.v[4]), "v"(b2.v[0]), "v"(c.v[4]), "v"(m3), "s"(n1), "v"(m2), "s"(n2), "v"(m1), "s"(n3), "v"(m0), "s"(n4));
    col = nc;
    m4 = (uint32_t)col;
    // column 5: 19 + 4 products; first: - s_4 p_0 (the low limb of column 4 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m4), "v"(col), "v"(a0.v[0]), "v"(b0.v[5]), "v"(a0.v[1]), "v"(b0.v[4]), "v"(a0.v[2]), "v"(b0.v[3]), "v"(a0.v[3]), "v"(b0.v[2]), "v"(a0.v[4]), "v"(b0.v[1]), "v"(a0.v[5]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[5]), "v"(a1.v[1]), "v"(b1.v[4]), "v"(a1.v[2]), "v"(b1.v[3]), "v"(a1.v[3]), "v"(b1.v[2]), "v"(a1.v[4]), "v"(b1.v[1]), "v"(a1.v[5]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[5]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, 1, %0\n\tv_mad_i64_i32 %0, %1, %13, %14, %0\n\tv_mad_i64_i32 %0, %1, %15, %16, %0\n\tv_mad_i64_i32 %0, %1, %17, %18, %0\n\tv_mad_i64_i32 %0, %1, %19, %20, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(a2.v[1]), "v"(b2.v[4]), "v"(a2.v[2]), "v"(b2.v[3]), "v"(a2.v[3]), "v"(b2.v[2]), "v"(a2.v[4]), "v"(b2.v[1]), "v"(a2.v[5]), "v"(b2.v[0]), "v"(c.v[5]), "v"(m4), "s"(n1), "v"(m3), "s"(n2), "v"(m2), "s"(n3), "v"(m1), "s"(n4));
    col = nc;
    m5 = (uint32_t)col;
    // column 6: 22 + 4 products; first: - s_5 p_0 (the low limb of column 5 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m5), "v"(col), "v"(a0.v[0]), "v"(b0.v[6]), "v"(a0.v[1]), "v"(b0.v[5]), "v"(a0.v[2]), "v"(b0.v[4]), "v"(a0.v[3]), "v"(b0.v[3]), "v"(a0.v[4]), "v"(b0.v[2]), "v"(a0.v[5]), "v"(b0.v[1]), "v"(a0.v[6]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[6]), "v"(a1.v[1]), "v"(b1.v[5]), "v"(a1.v[2]), "v"(b1.v[4]), "v"(a1.v[3]), "v"(b1.v[3]), "v"(a1.v[4]), "v"(b1.v[2]), "v"(a1.v[5]), "v"(b1.v[1]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, 1, %0\n\tv_mad_i64_i32 %0, %1, %19, %20, %0\n\tv_mad_i64_i32 %0, %1, %21, %22, %0\n\tv_mad_i64_i32 %0, %1, %23, %24, %0\n\tv_mad_i64_i32 %0, %1, %25, %26, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(a1.v[6]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[6]), "v"(a2.v[1]), "v"(b2.v[5]), "v"(a2.v[2]), "v"(b2.v[4]), "v"(a2.v[3]), "v"(b2.v[3]), "v"(a2.v[4]), "v"(b2.v[2]), "v"(a2.v[5]), "v"(b2.v[1]), "v"(a2.v[6]), "v"(b2.v[0]), "v"(c.v[6]), "v"(m5), "s"(n1), "v"(m4), "s"(n2), "v"(m3), "s"(n3), "v"(m2), "s"(n4));
    col = nc;
    m6 = (uint32_t)col;
    // column 7: 25 + 4 products; first: - s_6 p_0 (the low limb of column 6 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m6), "v"(col), "v"(a0.v[0]), "v"(b0.v[7]), "v"(a0.v[1]), "v"(b0.v[6]), "v"(a0.v[2]), "v"(b0.v[5]), "v"(a0.v[3]), "v"(b0.v[4]), "v"(a0.v[4]), "v"(b0.v[3]), "v"(a0.v[5]), "v"(b0.v[2]), "v"(a0.v[6]), "v"(b0.v[1]), "v"(a0.v[7]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[7]), "v"(a1.v[1]), "v"(b1.v[6]), "v"(a1.v[2]), "v"(b1.v[5]), "v"(a1.v[3]), "v"(b1.v[4]), "v"(a1.v[4]), "v"(b1.v[3]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, 1, %0\n\tv_mad_i64_i32 %0, %1, %25, %26, %0\n\tv_mad_i64_i32 %0, %1, %27, %28, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(a1.v[5]), "v"(b1.v[2]), "v"(a1.v[6]), "v"(b1.v[1]), "v"(a1.v[7]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[7]), "v"(a2.v[1]), "v"(b2.v[6]), "v"(a2.v[2]), "v"(b2.v[5]), "v"(a2.v[3]), "v"(b2.v[4]), "v"(a2.v[4]), "v"(b2.v[3]), "v"(a2.v[5]), "v"(b2.v[2]), "v"(a2.v[6]), "v"(b2.v[1]), "v"(a2.v[7]), "v"(b2.v[0]), "v"(c.v[7]), "v"(m6), "s"(n1), "v"(m5), "s"(n2));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(m4), "s"(n3), "v"(m3), "s"(n4));
    col = nc;
    m7 = (uint32_t)col;
    // column 8: 28 + 5 products; first: - s_7 p_0 (the low limb of column 7 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m7), "v"(col), "v"(a0.v[0]), "v"(b0.v[8]), "v"(a0.v[1]), "v"(b0.v[7]), "v"(a0.v[2]), "v"(b0.v[6]), "v"(a0.v[3]), "v"(b0.v[5]), "v"(a0.v[4]), "v"(b0.v[4]), "v"(a0.v[5]), "v"(b0.v[3]), "v"(a0.v[6]), "v"(b0.v[2]), "v"(a0.v[7]), "v"(b0.v[1]), "v"(a0.v[8]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[8]), "v"(a1.v[1]), "v"(b1.v[7]), "v"(a1.v[2]), "v"(b1.v[6]), "v"(a1.v[3]), "v"(b1.v[5]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(a1.v[4]), "v"(b1.v[4]), "v"(a1.v[5]), "v"(b1.v[3]), "v"(a1.v[6]), "v"(b1.v[2]), "v"(a1.v[7]), "v"(b1.v[1]), "v"(a1.v[8]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[8]), "v"(a2.v[1]), "v"(b2.v[7]), "v"(a2.v[2]), "v"(b2.v[6]), "v"(a2.v[3]), "v"(b2.v[5]), "v"(a2.v[4]), "v"(b2.v[4]), "v"(a2.v[5]), "v"(b2.v[3]), "v"(a2.v[6]), "v"(b2.v[2]), "v"(a2.v[7]), "v"(b2.v[1]), "v"(a2.v[8]), "v"(b2.v[0]));
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0\n\tv_mad_i64_i32 %0, %1, %3, %4, %0\n\tv_mad_i64_i32 %0, %1, %5, %6, %0\n\tv_mad_i64_i32 %0, %1, %7, %8, %0\n\tv_mad_i64_i32 %0, %1, %9, %10, %0\n\tv_mad_i64_i32 %0, %1, %11, %12, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(c.v[8]), "v"(m7), "s"(n1), "v"(m6), "s"(n2), "v"(m5), "s"(n3), "v"(m4), "s"(n4), "v"(m0), "s"(n8));
    col = nc;
    m8 = ((uint32_t)col & M29) | 0xC0000000u;        // (col & M29) - 2^30: the one digit with a fixed sign
    // column 9: 24 + 5 products; first: - s_8 p_0 (the low limb of column 8 cancels) and its carry
    asm("v_mad_i64_i32 %0, %1, %2, -1, %3\n\tv_ashrrev_i64 %0, 29, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "=&v"(nc), "=&s"(cc) : "v"(m8), "v"(col), "v"(a0.v[1]), "v"(b0.v[8]), "v"(a0.v[2]), "v"(b0.v[7]), "v"(a0.v[3]), "v"(b0.v[6]), "v"(a0.v[4]), "v"(b0.v[5]), "v"(a0.v[5]), "v"(b0.v[4]), "v"(a0.v[6]), "v"(b0.v[3]), "v"(a0.v[7]), "v"(b0.v[2]), "v"(a0.v[8]), "v"(b0.v[1]), "v"(a1.v[1]), "v"(b1.v[8]), "v"(a1.v[2]), "v"(b1.v[7]), "v"(a1.v[3]), "v"(b1.v[6]), "v"(a1.v[4]), "v"(b1.v[5]), "v"(a1.v[5]), "v"(b1.v[4]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_i64_i32 %0, %1, %24, %25, %0\n\tv_mad_i64_i32 %0, %1, %26, %27, %0\n\tv_mad_i64_i32 %0, %1, %28, %29, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(a1.v[6]), "v"(b1.v[3]), "v"(a1.v[7]), "v"(b1.v[2]), "v"(a1.v[8]), "v"(b1.v[1]), "v"(a2.v[1]), "v"(b2.v[8]), "v"(a2.v[2]), "v"(b2.v[7]), "v"(a2.v[3]), "v"(b2.v[6]), "v"(a2.v[4]), "v"(b2.v[5]), "v"(a2.v[5]), "v"(b2.v[4]), "v"(a2.v[6]), "v"(b2.v[3]), "v"(a2.v[7]), "v"(b2.v[2]), "v"(a2.v[8]), "v"(b2.v[1]), "v"(m8), "s"(n1), "v"(m7), "s"(n2), "v"(m6), "s"(n3));
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0"
        : "+&v"(nc), "=&s"(cc) : "v"(m5), "s"(n4), "v"(m1), "s"(n8));
    col = nc;
    r.v[0] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 10: 21 + 4 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[2]), "v"(b0.v[8]), "v"(a0.v[3]), "v"(b0.v[7]), "v"(a0.v[4]), "v"(b0.v[6]), "v"(a0.v[5]), "v"(b0.v[5]), "v"(a0.v[6]), "v"(b0.v[4]), "v"(a0.v[7]), "v"(b0.v[3]), "v"(a0.v[8]), "v"(b0.v[2]), "v"(a1.v[2]), "v"(b1.v[8]), "v"(a1.v[3]), "v"(b1.v[7]), "v"(a1.v[4]), "v"(b1.v[6]), "v"(a1.v[5]), "v"(b1.v[5]), "v"(a1.v[6]), "v"(b1.v[4]), "v"(a1.v[7]), "v"(b1.v[3]), "v"(a1.v[8]), "v"(b1.v[2]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_i64_i32 %0, %1, %16, %17, %0\n\tv_mad_i64_i32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0\n\tv_mad_i64_i32 %0, %1, %22, %23, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[2]), "v"(b2.v[8]), "v"(a2.v[3]), "v"(b2.v[7]), "v"(a2.v[4]), "v"(b2.v[6]), "v"(a2.v[5]), "v"(b2.v[5]), "v"(a2.v[6]), "v"(b2.v[4]), "v"(a2.v[7]), "v"(b2.v[3]), "v"(a2.v[8]), "v"(b2.v[2]), "v"(m8), "s"(n2), "v"(m7), "s"(n3), "v"(m6), "s"(n4), "v"(m2), "s"(n8));
    r.v[1] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 11: 18 + 3 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[3]), "v"(b0.v[8]), "v"(a0.v[4]), "v"(b0.v[7]), "v"(a0.v[5]), "v"(b0.v[6]), "v"(a0.v[6]), "v"(b0.v[5]), "v"(a0.v[7]), "v"(b0.v[4]), "v"(a0.v[8]), "v"(b0.v[3]), "v"(a1.v[3]), "v"(b1.v[8]), "v"(a1.v[4]), "v"(b1.v[7]), "v"(a1.v[5]), "v"(b1.v[6]), "v"(a1.v[6]), "v"(b1.v[5]), "v"(a1.v[7]), "v"(b1.v[4]), "v"(a1.v[8]), "v"(b1.v[3]), "v"(a2.v[3]), "v"(b2.v[8]), "v"(a2.v[4]), "v"(b2.v[7]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_i64_i32 %0, %1, %10, %11, %0\n\tv_mad_i64_i32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[5]), "v"(b2.v[6]), "v"(a2.v[6]), "v"(b2.v[5]), "v"(a2.v[7]), "v"(b2.v[4]), "v"(a2.v[8]), "v"(b2.v[3]), "v"(m8), "s"(n3), "v"(m7), "s"(n4), "v"(m3), "s"(n8));
    r.v[2] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 12: 15 + 2 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_u64_u32 %0, %1, %26, %27, %0\n\tv_mad_u64_u32 %0, %1, %28, %29, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[4]), "v"(b0.v[8]), "v"(a0.v[5]), "v"(b0.v[7]), "v"(a0.v[6]), "v"(b0.v[6]), "v"(a0.v[7]), "v"(b0.v[5]), "v"(a0.v[8]), "v"(b0.v[4]), "v"(a1.v[4]), "v"(b1.v[8]), "v"(a1.v[5]), "v"(b1.v[7]), "v"(a1.v[6]), "v"(b1.v[6]), "v"(a1.v[7]), "v"(b1.v[5]), "v"(a1.v[8]), "v"(b1.v[4]), "v"(a2.v[4]), "v"(b2.v[8]), "v"(a2.v[5]), "v"(b2.v[7]), "v"(a2.v[6]), "v"(b2.v[6]), "v"(a2.v[7]), "v"(b2.v[5]));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a2.v[8]), "v"(b2.v[4]), "v"(m8), "s"(n4), "v"(m4), "s"(n8));
    r.v[3] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 13: 12 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0\n\tv_mad_i64_i32 %0, %1, %26, %27, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[5]), "v"(b0.v[8]), "v"(a0.v[6]), "v"(b0.v[7]), "v"(a0.v[7]), "v"(b0.v[6]), "v"(a0.v[8]), "v"(b0.v[5]), "v"(a1.v[5]), "v"(b1.v[8]), "v"(a1.v[6]), "v"(b1.v[7]), "v"(a1.v[7]), "v"(b1.v[6]), "v"(a1.v[8]), "v"(b1.v[5]), "v"(a2.v[5]), "v"(b2.v[8]), "v"(a2.v[6]), "v"(b2.v[7]), "v"(a2.v[7]), "v"(b2.v[6]), "v"(a2.v[8]), "v"(b2.v[5]), "v"(m5), "s"(n8));
    r.v[4] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 14: 9 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_i64_i32 %0, %1, %20, %21, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[6]), "v"(b0.v[8]), "v"(a0.v[7]), "v"(b0.v[7]), "v"(a0.v[8]), "v"(b0.v[6]), "v"(a1.v[6]), "v"(b1.v[8]), "v"(a1.v[7]), "v"(b1.v[7]), "v"(a1.v[8]), "v"(b1.v[6]), "v"(a2.v[6]), "v"(b2.v[8]), "v"(a2.v[7]), "v"(b2.v[7]), "v"(a2.v[8]), "v"(b2.v[6]), "v"(m6), "s"(n8));
    r.v[5] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 15: 6 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_i64_i32 %0, %1, %14, %15, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[7]), "v"(b0.v[8]), "v"(a0.v[8]), "v"(b0.v[7]), "v"(a1.v[7]), "v"(b1.v[8]), "v"(a1.v[8]), "v"(b1.v[7]), "v"(a2.v[7]), "v"(b2.v[8]), "v"(a2.v[8]), "v"(b2.v[7]), "v"(m7), "s"(n8));
    r.v[6] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    // column 16: 3 + 1 products
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0"
        : "+&v"(col), "=&s"(cc) : "v"(a0.v[8]), "v"(b0.v[8]), "v"(a1.v[8]), "v"(b1.v[8]), "v"(a2.v[8]), "v"(b2.v[8]), "v"(m8), "s"(n8));
    r.v[7] = (uint32_t)col & M29; col = (uint64_t)((int64_t)col >> 29);
    r.v[8] = (uint32_t)col;
    return r;
}
// ---- END GENERATED
#endif

}  // namespace mb
