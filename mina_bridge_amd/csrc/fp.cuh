// fp.cuh -- Pasta prime-field arithmetic for gfx950 (and the host driver).
//
// Replaces (on the device) ark-ff 0.3 `Fp256<FpParameters>` Montgomery arithmetic used by every
// routine on the verifier path (pin: core/Cargo.toml:19-22,47-57 -> lambdaclass/openmina_algebra).
//
// Representation: 8 x u32 little-endian limbs, Montgomery form with R = 2^256, always fully
// reduced to [0, p).  Both Pasta primes have the shape p = 2^254 + t (t < 2^126) and p = 1 mod 2^32:
//   limbs   = { 1, p1, p2, p3, 0, 0, 0, 0x40000000 }
//   -p^-1 mod 2^32 = 0xffffffff  ->  the Montgomery quotient digit is m = -t0 (no multiply)
//   m * p needs only 3 real 32x32 products (p1..p3); p0 = 1 is an add, p7 = 2^30 is a shift.
// so one Montgomery product costs 64 + 24 = 88 `v_mad_u64_u32` instead of 128.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MB_HD __host__ __device__ __forceinline__
#else
#define MB_HD inline
#endif

// Wave priority of every kernel of a job that is NOT the chip-filling state hash (round 6).  The legs of a device-resident job run on streams of their own
// (api_state.hip dev_fork_lanes); where a wave of the wrap-proof chain -- ~15 dependent kernels of <= 1 wave per SIMD -- shares a SIMD with the four or five
// resident state-hash waves, the instruction arbiter takes waves oldest first and the chain ran at a fifth of its speed: a lone 16 384-proof call took as long
// forked as on one stream (68.7 against 70.5 ms; timeline: profiles/r06_dev_fork.md).  `s_setprio` raises the issuing wave's priority at the arbiter (0 = the
// default every other wave keeps, 3 = highest): the chain keeps its own pace and the hashes take the issue slots it leaves.  -DMB_CHAIN_PRIO=0 builds without.
#ifndef MB_CHAIN_PRIO
#define MB_CHAIN_PRIO 2
#endif
#ifndef MB_LEG_PRIO                       // the chip-filling kernels of the accumulator leg and of the MSMs (b_poly tables / GEMM, sort, accumulate, reductions)
#define MB_LEG_PRIO 2
#endif
#if defined(__HIPCC__)
template <int LEG = 0> __device__ __forceinline__ void mb_wave_prio() {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int prio = LEG ? MB_LEG_PRIO : MB_CHAIN_PRIO;
    if constexpr (prio != 0) __builtin_amdgcn_s_setprio(prio);
#endif
}
#else
template <int LEG = 0> static inline void mb_wave_prio() {}      // the host-only builds of the ThreadSanitizer tier (tests/fuzz): kernels are stubs
#endif

namespace mb {

enum : int { FIELD_FP = 0, FIELD_FQ = 1 };     // Fp: Pallas base / Vesta scalar.  Fq: Vesta base / Pallas scalar
enum : int { CURVE_PALLAS = 0, CURVE_VESTA = 1 };

struct alignas(16) fe_t { uint32_t v[8]; };

template <int F> struct FieldP;
template <> struct FieldP<FIELD_FP> {
    static constexpr uint32_t P1 = 0x992d30edu, P2 = 0x094cf91bu, P3 = 0x224698fcu;
};
template <> struct FieldP<FIELD_FQ> {
    static constexpr uint32_t P1 = 0x8c46eb21u, P2 = 0x0994a8ddu, P3 = 0x224698fcu;
};
static constexpr uint32_t P7 = 0x40000000u;

template <int F> MB_HD uint32_t modulus_limb(int i) {
    switch (i) {
        case 0: return 1u;
        case 1: return FieldP<F>::P1;
        case 2: return FieldP<F>::P2;
        case 3: return FieldP<F>::P3;
        case 7: return P7;
        default: return 0u;
    }
}

MB_HD bool fe_is_zero(const fe_t &a) {
    return (a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4] | a.v[5] | a.v[6] | a.v[7]) == 0;
}
MB_HD bool fe_eq(const fe_t &a, const fe_t &b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d |= a.v[i] ^ b.v[i];
    return d == 0;
}
MB_HD fe_t fe_zero() { fe_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0;
    return r; }

// r = a - p if a >= p else a        (a < 2p)
// plain 256-bit integer (not Montgomery) strictly below the modulus?  (ark's `CanonicalDeserialize` admits nothing else)
template <int F> MB_HD bool fe_words_canonical(const fe_t &a) {
    for (int i = 7; i >= 0; --i) { const uint32_t m = modulus_limb<F>(i); if (a.v[i] != m) return a.v[i] < m; }
    return false;
}
template <int F> MB_HD fe_t fe_cond_sub_p_portable(const fe_t &a) {
    fe_t d; uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.v[i] - modulus_limb<F>(i) - br;
        d.v[i] = (uint32_t)t; br = (uint32_t)(t >> 32) & 1u;
    }
    fe_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = br ? a.v[i] : d.v[i];
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
template <int F> __device__ __forceinline__ fe_t fe_cond_sub_p_device(const fe_t &a) {
    fe_t r;
    uint32_t d0, d1, d2, d3, d4, d5, d6, d7;
    asm("v_subrev_co_u32_e32 %0, vcc, 1, %8\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, %16, %9, vcc\n\t"
        "v_subbrev_co_u32_e32 %2, vcc, %17, %10, vcc\n\t"
        "v_subbrev_co_u32_e32 %3, vcc, %18, %11, vcc\n\t"
        "v_subbrev_co_u32_e32 %4, vcc, 0, %12, vcc\n\t"
        "v_subbrev_co_u32_e32 %5, vcc, 0, %13, vcc\n\t"
        "v_subbrev_co_u32_e32 %6, vcc, 0, %14, vcc\n\t"
        "v_subbrev_co_u32_e32 %7, vcc, %19, %15, vcc\n\t"
        "v_cndmask_b32_e32 %0, %0, %8, vcc\n\t"
        "v_cndmask_b32_e32 %1, %1, %9, vcc\n\t"
        "v_cndmask_b32_e32 %2, %2, %10, vcc\n\t"
        "v_cndmask_b32_e32 %3, %3, %11, vcc\n\t"
        "v_cndmask_b32_e32 %4, %4, %12, vcc\n\t"
        "v_cndmask_b32_e32 %5, %5, %13, vcc\n\t"
        "v_cndmask_b32_e32 %6, %6, %14, vcc\n\t"
        "v_cndmask_b32_e32 %7, %7, %15, vcc"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(FieldP<F>::P1), "v"(FieldP<F>::P2), "v"(FieldP<F>::P3), "v"(P7)
        : "vcc");
    r.v[0] = d0; r.v[1] = d1; r.v[2] = d2; r.v[3] = d3; r.v[4] = d4; r.v[5] = d5; r.v[6] = d6; r.v[7] = d7;
    return r;
}
#endif
template <int F> MB_HD fe_t fe_cond_sub_p(const fe_t &a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_cond_sub_p_device<F>(a);
#else
    return fe_cond_sub_p_portable<F>(a);
#endif
}

template <int F> MB_HD fe_t fe_add_portable(const fe_t &a, const fe_t &b) {
    fe_t s; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.v[i] + b.v[i] + c;
        s.v[i] = (uint32_t)t; c = (uint32_t)(t >> 32);
    }
    // a + b < 2p < 2^256: no carry out of limb 7
    return fe_cond_sub_p<F>(s);
}

template <int F> MB_HD fe_t fe_sub_portable(const fe_t &a, const fe_t &b) {
    fe_t d; uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
        d.v[i] = (uint32_t)t; br = (uint32_t)(t >> 32) & 1u;
    }
    // if borrow: add p back
    uint32_t mask = 0u - br, c = 0;
    fe_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)d.v[i] + (modulus_limb<F>(i) & mask) + c;
        r.v[i] = (uint32_t)t; c = (uint32_t)(t >> 32);
    }
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950: straight VCC carry chains (24 / 22 VALU instructions) instead of compiler-emulated 64-bit carries.
template <int F> __device__ __forceinline__ fe_t fe_add_device(const fe_t &a, const fe_t &b) {
    fe_t r;
    uint32_t s0, s1, s2, s3, s4, s5, s6, s7, d0, d1, d2, d3, d4, d5, d6, d7;
    asm("v_add_co_u32_e32 %0, vcc, %16, %24\n\t"
        "v_addc_co_u32_e32 %1, vcc, %17, %25, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %18, %26, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, %19, %27, vcc\n\t"
        "v_addc_co_u32_e32 %4, vcc, %20, %28, vcc\n\t"
        "v_addc_co_u32_e32 %5, vcc, %21, %29, vcc\n\t"
        "v_addc_co_u32_e32 %6, vcc, %22, %30, vcc\n\t"
        "v_addc_co_u32_e32 %7, vcc, %23, %31, vcc\n\t"
        "v_subrev_co_u32_e32 %8, vcc, 1, %0\n\t"
        "v_subbrev_co_u32_e32 %9, vcc, %32, %1, vcc\n\t"
        "v_subbrev_co_u32_e32 %10, vcc, %33, %2, vcc\n\t"
        "v_subbrev_co_u32_e32 %11, vcc, %34, %3, vcc\n\t"
        "v_subbrev_co_u32_e32 %12, vcc, 0, %4, vcc\n\t"
        "v_subbrev_co_u32_e32 %13, vcc, 0, %5, vcc\n\t"
        "v_subbrev_co_u32_e32 %14, vcc, 0, %6, vcc\n\t"
        "v_subbrev_co_u32_e32 %15, vcc, %35, %7, vcc\n\t"
        "v_cndmask_b32_e32 %8, %8, %0, vcc\n\t"
        "v_cndmask_b32_e32 %9, %9, %1, vcc\n\t"
        "v_cndmask_b32_e32 %10, %10, %2, vcc\n\t"
        "v_cndmask_b32_e32 %11, %11, %3, vcc\n\t"
        "v_cndmask_b32_e32 %12, %12, %4, vcc\n\t"
        "v_cndmask_b32_e32 %13, %13, %5, vcc\n\t"
        "v_cndmask_b32_e32 %14, %14, %6, vcc\n\t"
        "v_cndmask_b32_e32 %15, %15, %7, vcc"
        : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(s4), "=&v"(s5), "=&v"(s6), "=&v"(s7),
          "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]),
          "v"(FieldP<F>::P1), "v"(FieldP<F>::P2), "v"(FieldP<F>::P3), "v"(P7)
        : "vcc");
    r.v[0] = d0; r.v[1] = d1; r.v[2] = d2; r.v[3] = d3; r.v[4] = d4; r.v[5] = d5; r.v[6] = d6; r.v[7] = d7;
    return r;
}
template <int F> __device__ __forceinline__ fe_t fe_sub_device(const fe_t &a, const fe_t &b) {
    fe_t r;
    uint32_t d0, d1, d2, d3, d4, d5, d6, d7, m, t1, t2, t3, t7;
    asm("v_sub_co_u32_e32 %0, vcc, %13, %21\n\t"
        "v_subb_co_u32_e32 %1, vcc, %14, %22, vcc\n\t"
        "v_subb_co_u32_e32 %2, vcc, %15, %23, vcc\n\t"
        "v_subb_co_u32_e32 %3, vcc, %16, %24, vcc\n\t"
        "v_subb_co_u32_e32 %4, vcc, %17, %25, vcc\n\t"
        "v_subb_co_u32_e32 %5, vcc, %18, %26, vcc\n\t"
        "v_subb_co_u32_e32 %6, vcc, %19, %27, vcc\n\t"
        "v_subb_co_u32_e32 %7, vcc, %20, %28, vcc\n\t"
        "v_cndmask_b32_e64 %8, 0, -1, vcc\n\t"
        "v_and_b32_e32 %9, %29, %8\n\t"
        "v_and_b32_e32 %10, %30, %8\n\t"
        "v_and_b32_e32 %11, %31, %8\n\t"
        "v_and_b32_e32 %12, %32, %8\n\t"
        "v_and_b32_e32 %8, 1, %8\n\t"
        "v_add_co_u32_e32 %0, vcc, %0, %8\n\t"
        "v_addc_co_u32_e32 %1, vcc, %1, %9, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %10, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, %3, %11, vcc\n\t"
        "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
        "v_addc_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
        "v_addc_co_u32_e32 %7, vcc, %7, %12, vcc"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7),
          "=&v"(m), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t7)
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]),
          "v"(FieldP<F>::P1), "v"(FieldP<F>::P2), "v"(FieldP<F>::P3), "v"(P7)
        : "vcc");
    r.v[0] = d0; r.v[1] = d1; r.v[2] = d2; r.v[3] = d3; r.v[4] = d4; r.v[5] = d5; r.v[6] = d6; r.v[7] = d7;
    return r;
}
#endif
template <int F> MB_HD fe_t fe_add(const fe_t &a, const fe_t &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_add_device<F>(a, b);
#else
    return fe_add_portable<F>(a, b);
#endif
}
template <int F> MB_HD fe_t fe_sub(const fe_t &a, const fe_t &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_sub_device<F>(a, b);
#else
    return fe_sub_portable<F>(a, b);
#endif
}

template <int F> MB_HD fe_t fe_neg(const fe_t &a) { return fe_sub<F>(fe_zero(), a); }
template <int F> MB_HD fe_t fe_dbl(const fe_t &a) { return fe_add<F>(a, a); }

// Montgomery product, CIOS over 32-bit digits, specialised to the Pasta prime shape (portable form:
// host code and the reference for the device version below).
template <int F> MB_HD fe_t fe_mul_portable(const fe_t &a, const fe_t &b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
        const uint32_t bi = b.v[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c = (uint64_t)a.v[j] * bi + t[j] + c;
            t[j] = (uint32_t)c; c >>= 32;
        }
        t[8] += (uint32_t)c;                                 // t < 2^288 here, no further carry
        const uint32_t m = 0u - t[0];
        uint64_t k = (t[0] != 0) ? 1u : 0u;                  // carry of t0 + m * 1
        k = (uint64_t)m * FieldP<F>::P1 + t[1] + k; t[0] = (uint32_t)k; k >>= 32;
        k = (uint64_t)m * FieldP<F>::P2 + t[2] + k; t[1] = (uint32_t)k; k >>= 32;
        k = (uint64_t)m * FieldP<F>::P3 + t[3] + k; t[2] = (uint32_t)k; k >>= 32;
        k = (uint64_t)t[4] + k; t[3] = (uint32_t)k; k >>= 32;
        k = (uint64_t)t[5] + k; t[4] = (uint32_t)k; k >>= 32;
        k = (uint64_t)t[6] + k; t[5] = (uint32_t)k; k >>= 32;
        k = ((uint64_t)m << 30) + t[7] + k; t[6] = (uint32_t)k; k >>= 32;
        k = (uint64_t)t[8] + k; t[7] = (uint32_t)k; t[8] = (uint32_t)(k >> 32);
    }
    fe_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = t[i];
    return fe_cond_sub_p<F>(r);                               // t < 2p
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 device version: product scanning (FIPS) with a 96-bit column accumulator (acc:64, hi:32).
// One `v_mad_u64_u32` per 32x32 product accumulates straight into `acc`; its carry-out goes to an SGPR pair
// that the following `v_addc_co_u32` folds into `hi` -- no 64-bit addend assembly, no v_mov traffic.
// 64 (a*b) + 32 (m*p1, p2, p3, p7) multiply-accumulates; p0 = 1 is a carry fold (mb_fold_shift).  `hi` is zero at the start of every
// column, so the column's first carry add computes 0 + 0 + carry INTO it: no per-column `v_mov hi, 0` (15 per product).
// acc(96 bit) += m * p0 where m = -lo: the low word becomes 0 and carries (lo != 0) into the upper 64 bits; then >> 32.
__device__ __forceinline__ void mb_fold_shift(uint64_t &acc, uint32_t &hi, uint32_t lo, uint32_t mid) {
    uint32_t nlo, nhi;
    asm("v_cmp_ne_u32_e32 vcc, 0, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %3, vcc\n\tv_addc_co_u32_e32 %1, vcc, 0, %4, vcc"
        : "=&v"(nlo), "=&v"(nhi) : "v"(lo), "v"(mid), "v"(hi) : "vcc");
    acc = ((uint64_t)nhi << 32) | nlo;                       // `hi` is dead here: the next column's first carry add writes it
}
template <int F, bool RED = true> __device__ __forceinline__ fe_t fe_mul_device(const fe_t &a, const fe_t &b) {
    // generated by tools/gen_fe_mul.py -- product scanning, one Montgomery reduction
    uint64_t acc = 0, cc; uint32_t hi, lo, mid; fe_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7;
    const uint32_t p1 = FieldP<F>::P1, p2 = FieldP<F>::P2, p3 = FieldP<F>::P3, p7 = P7;
    // column 0: 1 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[0]));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m0 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 1: 3 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[1]), "v"(b.v[0]), "v"(m0), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m1 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 2: 5 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[0]), "v"(m0), "v"(p2), "v"(m1), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m2 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 3: 7 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[2]), "v"(b.v[1]), "v"(a.v[3]), "v"(b.v[0]), "v"(m0), "v"(p3), "v"(m1), "v"(p2), "v"(m2), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m3 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 4: 8 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[3]), "v"(b.v[1]), "v"(a.v[4]), "v"(b.v[0]), "v"(m1), "v"(p3), "v"(m2), "v"(p2), "v"(m3), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m4 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 5: 9 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[0]), "v"(m2), "v"(p3), "v"(m3), "v"(p2), "v"(m4), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m5 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 6: 10 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[0]), "v"(m3), "v"(p3), "v"(m4), "v"(p2), "v"(m5), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m6 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 7: 12 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0]), "v"(m0), "v"(p7), "v"(m4), "v"(p3), "v"(m5), "v"(p2), "v"(m6), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m7 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 8: 11 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[1]), "v"(m1), "v"(p7), "v"(m5), "v"(p3), "v"(m6), "v"(p2), "v"(m7), "v"(p1));
    r.v[0] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 9: 9 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[2]), "v"(b.v[7]), "v"(a.v[3]), "v"(b.v[6]), "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[2]), "v"(m2), "v"(p7), "v"(m6), "v"(p3), "v"(m7), "v"(p2));
    r.v[1] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 10: 7 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[3]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[5]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[3]), "v"(m3), "v"(p7), "v"(m7), "v"(p3));
    r.v[2] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 11: 5 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[4]), "v"(b.v[7]), "v"(a.v[5]), "v"(b.v[6]), "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[7]), "v"(b.v[4]), "v"(m4), "v"(p7));
    r.v[3] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 12: 4 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[5]), "v"(m5), "v"(p7));
    r.v[4] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 13: 3 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[7]), "v"(b.v[6]), "v"(m6), "v"(p7));
    r.v[5] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 14: 2 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a.v[7]), "v"(b.v[7]), "v"(m7), "v"(p7));
    r.v[6] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    r.v[7] = (uint32_t)acc;                                       // result < 2p < 2^256
    return RED ? fe_cond_sub_p<F>(r) : r;                 // RED = false: result < sum(a_i b_i) / 2^256 + p, left to the caller
}
template <int F, bool RED = true> __device__ __forceinline__ fe_t fe_dot2_device(const fe_t &a0, const fe_t &b0, const fe_t &a1, const fe_t &b1) {
    // generated by tools/gen_fe_mul.py -- product scanning, one Montgomery reduction
    uint64_t acc = 0, cc; uint32_t hi, lo, mid; fe_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7;
    const uint32_t p1 = FieldP<F>::P1, p2 = FieldP<F>::P2, p3 = FieldP<F>::P3, p7 = P7;
    // column 0: 2 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[0]));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m0 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 1: 5 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[1]), "v"(a0.v[1]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[1]), "v"(a1.v[1]), "v"(b1.v[0]), "v"(m0), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m1 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 2: 8 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[2]), "v"(a0.v[1]), "v"(b0.v[1]), "v"(a0.v[2]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[2]), "v"(a1.v[1]), "v"(b1.v[1]), "v"(a1.v[2]), "v"(b1.v[0]), "v"(m0), "v"(p2), "v"(m1), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m2 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 3: 11 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[3]), "v"(a0.v[1]), "v"(b0.v[2]), "v"(a0.v[2]), "v"(b0.v[1]), "v"(a0.v[3]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[3]), "v"(a1.v[1]), "v"(b1.v[2]), "v"(a1.v[2]), "v"(b1.v[1]), "v"(a1.v[3]), "v"(b1.v[0]), "v"(m0), "v"(p3), "v"(m1), "v"(p2), "v"(m2), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m3 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 4: 13 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[4]), "v"(a0.v[1]), "v"(b0.v[3]), "v"(a0.v[2]), "v"(b0.v[2]), "v"(a0.v[3]), "v"(b0.v[1]), "v"(a0.v[4]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[4]), "v"(a1.v[1]), "v"(b1.v[3]), "v"(a1.v[2]), "v"(b1.v[2]), "v"(a1.v[3]), "v"(b1.v[1]), "v"(a1.v[4]), "v"(b1.v[0]), "v"(m1), "v"(p3), "v"(m2), "v"(p2));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(m3), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m4 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 5: 15 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[5]), "v"(a0.v[1]), "v"(b0.v[4]), "v"(a0.v[2]), "v"(b0.v[3]), "v"(a0.v[3]), "v"(b0.v[2]), "v"(a0.v[4]), "v"(b0.v[1]), "v"(a0.v[5]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[5]), "v"(a1.v[1]), "v"(b1.v[4]), "v"(a1.v[2]), "v"(b1.v[3]), "v"(a1.v[3]), "v"(b1.v[2]), "v"(a1.v[4]), "v"(b1.v[1]), "v"(a1.v[5]), "v"(b1.v[0]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(m2), "v"(p3), "v"(m3), "v"(p2), "v"(m4), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m5 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 6: 17 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[6]), "v"(a0.v[1]), "v"(b0.v[5]), "v"(a0.v[2]), "v"(b0.v[4]), "v"(a0.v[3]), "v"(b0.v[3]), "v"(a0.v[4]), "v"(b0.v[2]), "v"(a0.v[5]), "v"(b0.v[1]), "v"(a0.v[6]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[6]), "v"(a1.v[1]), "v"(b1.v[5]), "v"(a1.v[2]), "v"(b1.v[4]), "v"(a1.v[3]), "v"(b1.v[3]), "v"(a1.v[4]), "v"(b1.v[2]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(a1.v[5]), "v"(b1.v[1]), "v"(a1.v[6]), "v"(b1.v[0]), "v"(m3), "v"(p3), "v"(m4), "v"(p2), "v"(m5), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m6 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 7: 20 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[7]), "v"(a0.v[1]), "v"(b0.v[6]), "v"(a0.v[2]), "v"(b0.v[5]), "v"(a0.v[3]), "v"(b0.v[4]), "v"(a0.v[4]), "v"(b0.v[3]), "v"(a0.v[5]), "v"(b0.v[2]), "v"(a0.v[6]), "v"(b0.v[1]), "v"(a0.v[7]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[7]), "v"(a1.v[1]), "v"(b1.v[6]), "v"(a1.v[2]), "v"(b1.v[5]), "v"(a1.v[3]), "v"(b1.v[4]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(a1.v[4]), "v"(b1.v[3]), "v"(a1.v[5]), "v"(b1.v[2]), "v"(a1.v[6]), "v"(b1.v[1]), "v"(a1.v[7]), "v"(b1.v[0]), "v"(m0), "v"(p7), "v"(m4), "v"(p3), "v"(m5), "v"(p2), "v"(m6), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m7 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 8: 18 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[1]), "v"(b0.v[7]), "v"(a0.v[2]), "v"(b0.v[6]), "v"(a0.v[3]), "v"(b0.v[5]), "v"(a0.v[4]), "v"(b0.v[4]), "v"(a0.v[5]), "v"(b0.v[3]), "v"(a0.v[6]), "v"(b0.v[2]), "v"(a0.v[7]), "v"(b0.v[1]), "v"(a1.v[1]), "v"(b1.v[7]), "v"(a1.v[2]), "v"(b1.v[6]), "v"(a1.v[3]), "v"(b1.v[5]), "v"(a1.v[4]), "v"(b1.v[4]), "v"(a1.v[5]), "v"(b1.v[3]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(a1.v[6]), "v"(b1.v[2]), "v"(a1.v[7]), "v"(b1.v[1]), "v"(m1), "v"(p7), "v"(m5), "v"(p3), "v"(m6), "v"(p2), "v"(m7), "v"(p1));
    r.v[0] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 9: 15 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[2]), "v"(b0.v[7]), "v"(a0.v[3]), "v"(b0.v[6]), "v"(a0.v[4]), "v"(b0.v[5]), "v"(a0.v[5]), "v"(b0.v[4]), "v"(a0.v[6]), "v"(b0.v[3]), "v"(a0.v[7]), "v"(b0.v[2]), "v"(a1.v[2]), "v"(b1.v[7]), "v"(a1.v[3]), "v"(b1.v[6]), "v"(a1.v[4]), "v"(b1.v[5]), "v"(a1.v[5]), "v"(b1.v[4]), "v"(a1.v[6]), "v"(b1.v[3]), "v"(a1.v[7]), "v"(b1.v[2]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(m2), "v"(p7), "v"(m6), "v"(p3), "v"(m7), "v"(p2));
    r.v[1] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 10: 12 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[3]), "v"(b0.v[7]), "v"(a0.v[4]), "v"(b0.v[6]), "v"(a0.v[5]), "v"(b0.v[5]), "v"(a0.v[6]), "v"(b0.v[4]), "v"(a0.v[7]), "v"(b0.v[3]), "v"(a1.v[3]), "v"(b1.v[7]), "v"(a1.v[4]), "v"(b1.v[6]), "v"(a1.v[5]), "v"(b1.v[5]), "v"(a1.v[6]), "v"(b1.v[4]), "v"(a1.v[7]), "v"(b1.v[3]), "v"(m3), "v"(p7), "v"(m7), "v"(p3));
    r.v[2] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 11: 9 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[4]), "v"(b0.v[7]), "v"(a0.v[5]), "v"(b0.v[6]), "v"(a0.v[6]), "v"(b0.v[5]), "v"(a0.v[7]), "v"(b0.v[4]), "v"(a1.v[4]), "v"(b1.v[7]), "v"(a1.v[5]), "v"(b1.v[6]), "v"(a1.v[6]), "v"(b1.v[5]), "v"(a1.v[7]), "v"(b1.v[4]), "v"(m4), "v"(p7));
    r.v[3] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 12: 7 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[5]), "v"(b0.v[7]), "v"(a0.v[6]), "v"(b0.v[6]), "v"(a0.v[7]), "v"(b0.v[5]), "v"(a1.v[5]), "v"(b1.v[7]), "v"(a1.v[6]), "v"(b1.v[6]), "v"(a1.v[7]), "v"(b1.v[5]), "v"(m5), "v"(p7));
    r.v[4] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 13: 5 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[6]), "v"(b0.v[7]), "v"(a0.v[7]), "v"(b0.v[6]), "v"(a1.v[6]), "v"(b1.v[7]), "v"(a1.v[7]), "v"(b1.v[6]), "v"(m6), "v"(p7));
    r.v[5] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 14: 3 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[7]), "v"(b0.v[7]), "v"(a1.v[7]), "v"(b1.v[7]), "v"(m7), "v"(p7));
    r.v[6] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    r.v[7] = (uint32_t)acc;                                       // result < 2p < 2^256
    return RED ? fe_cond_sub_p<F>(r) : r;                 // RED = false: result < sum(a_i b_i) / 2^256 + p, left to the caller
}
template <int F, bool RED = true> __device__ __forceinline__ fe_t fe_dot3_device(const fe_t &a0, const fe_t &b0, const fe_t &a1, const fe_t &b1, const fe_t &a2, const fe_t &b2) {
    // generated by tools/gen_fe_mul.py -- product scanning, one Montgomery reduction
    uint64_t acc = 0, cc; uint32_t hi, lo, mid; fe_t r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7;
    const uint32_t p1 = FieldP<F>::P1, p2 = FieldP<F>::P2, p3 = FieldP<F>::P3, p7 = P7;
    // column 0: 3 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[0]));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m0 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 1: 7 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[1]), "v"(a0.v[1]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[1]), "v"(a1.v[1]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[1]), "v"(a2.v[1]), "v"(b2.v[0]), "v"(m0), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m1 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 2: 11 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[2]), "v"(a0.v[1]), "v"(b0.v[1]), "v"(a0.v[2]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[2]), "v"(a1.v[1]), "v"(b1.v[1]), "v"(a1.v[2]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[2]), "v"(a2.v[1]), "v"(b2.v[1]), "v"(a2.v[2]), "v"(b2.v[0]), "v"(m0), "v"(p2), "v"(m1), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m2 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 3: 15 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[3]), "v"(a0.v[1]), "v"(b0.v[2]), "v"(a0.v[2]), "v"(b0.v[1]), "v"(a0.v[3]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[3]), "v"(a1.v[1]), "v"(b1.v[2]), "v"(a1.v[2]), "v"(b1.v[1]), "v"(a1.v[3]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[3]), "v"(a2.v[1]), "v"(b2.v[2]), "v"(a2.v[2]), "v"(b2.v[1]), "v"(a2.v[3]), "v"(b2.v[0]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(m0), "v"(p3), "v"(m1), "v"(p2), "v"(m2), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m3 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 4: 18 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[4]), "v"(a0.v[1]), "v"(b0.v[3]), "v"(a0.v[2]), "v"(b0.v[2]), "v"(a0.v[3]), "v"(b0.v[1]), "v"(a0.v[4]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[4]), "v"(a1.v[1]), "v"(b1.v[3]), "v"(a1.v[2]), "v"(b1.v[2]), "v"(a1.v[3]), "v"(b1.v[1]), "v"(a1.v[4]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[4]), "v"(a2.v[1]), "v"(b2.v[3]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(a2.v[2]), "v"(b2.v[2]), "v"(a2.v[3]), "v"(b2.v[1]), "v"(a2.v[4]), "v"(b2.v[0]), "v"(m1), "v"(p3), "v"(m2), "v"(p2), "v"(m3), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m4 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 5: 21 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[5]), "v"(a0.v[1]), "v"(b0.v[4]), "v"(a0.v[2]), "v"(b0.v[3]), "v"(a0.v[3]), "v"(b0.v[2]), "v"(a0.v[4]), "v"(b0.v[1]), "v"(a0.v[5]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[5]), "v"(a1.v[1]), "v"(b1.v[4]), "v"(a1.v[2]), "v"(b1.v[3]), "v"(a1.v[3]), "v"(b1.v[2]), "v"(a1.v[4]), "v"(b1.v[1]), "v"(a1.v[5]), "v"(b1.v[0]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(a2.v[0]), "v"(b2.v[5]), "v"(a2.v[1]), "v"(b2.v[4]), "v"(a2.v[2]), "v"(b2.v[3]), "v"(a2.v[3]), "v"(b2.v[2]), "v"(a2.v[4]), "v"(b2.v[1]), "v"(a2.v[5]), "v"(b2.v[0]), "v"(m2), "v"(p3), "v"(m3), "v"(p2), "v"(m4), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m5 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 6: 24 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[6]), "v"(a0.v[1]), "v"(b0.v[5]), "v"(a0.v[2]), "v"(b0.v[4]), "v"(a0.v[3]), "v"(b0.v[3]), "v"(a0.v[4]), "v"(b0.v[2]), "v"(a0.v[5]), "v"(b0.v[1]), "v"(a0.v[6]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[6]), "v"(a1.v[1]), "v"(b1.v[5]), "v"(a1.v[2]), "v"(b1.v[4]), "v"(a1.v[3]), "v"(b1.v[3]), "v"(a1.v[4]), "v"(b1.v[2]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(a1.v[5]), "v"(b1.v[1]), "v"(a1.v[6]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[6]), "v"(a2.v[1]), "v"(b2.v[5]), "v"(a2.v[2]), "v"(b2.v[4]), "v"(a2.v[3]), "v"(b2.v[3]), "v"(a2.v[4]), "v"(b2.v[2]), "v"(a2.v[5]), "v"(b2.v[1]), "v"(a2.v[6]), "v"(b2.v[0]), "v"(m3), "v"(p3), "v"(m4), "v"(p2), "v"(m5), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m6 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 7: 28 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[0]), "v"(b0.v[7]), "v"(a0.v[1]), "v"(b0.v[6]), "v"(a0.v[2]), "v"(b0.v[5]), "v"(a0.v[3]), "v"(b0.v[4]), "v"(a0.v[4]), "v"(b0.v[3]), "v"(a0.v[5]), "v"(b0.v[2]), "v"(a0.v[6]), "v"(b0.v[1]), "v"(a0.v[7]), "v"(b0.v[0]), "v"(a1.v[0]), "v"(b1.v[7]), "v"(a1.v[1]), "v"(b1.v[6]), "v"(a1.v[2]), "v"(b1.v[5]), "v"(a1.v[3]), "v"(b1.v[4]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(a1.v[4]), "v"(b1.v[3]), "v"(a1.v[5]), "v"(b1.v[2]), "v"(a1.v[6]), "v"(b1.v[1]), "v"(a1.v[7]), "v"(b1.v[0]), "v"(a2.v[0]), "v"(b2.v[7]), "v"(a2.v[1]), "v"(b2.v[6]), "v"(a2.v[2]), "v"(b2.v[5]), "v"(a2.v[3]), "v"(b2.v[4]), "v"(a2.v[4]), "v"(b2.v[3]), "v"(a2.v[5]), "v"(b2.v[2]), "v"(a2.v[6]), "v"(b2.v[1]), "v"(a2.v[7]), "v"(b2.v[0]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(m0), "v"(p7), "v"(m4), "v"(p3), "v"(m5), "v"(p2), "v"(m6), "v"(p1));
    lo = (uint32_t)acc; mid = (uint32_t)(acc >> 32); m7 = 0u - lo;
    mb_fold_shift(acc, hi, lo, mid);                          // + m_k * p0 (low word -> 0), then >> 32
    // column 8: 25 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[1]), "v"(b0.v[7]), "v"(a0.v[2]), "v"(b0.v[6]), "v"(a0.v[3]), "v"(b0.v[5]), "v"(a0.v[4]), "v"(b0.v[4]), "v"(a0.v[5]), "v"(b0.v[3]), "v"(a0.v[6]), "v"(b0.v[2]), "v"(a0.v[7]), "v"(b0.v[1]), "v"(a1.v[1]), "v"(b1.v[7]), "v"(a1.v[2]), "v"(b1.v[6]), "v"(a1.v[3]), "v"(b1.v[5]), "v"(a1.v[4]), "v"(b1.v[4]), "v"(a1.v[5]), "v"(b1.v[3]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(a1.v[6]), "v"(b1.v[2]), "v"(a1.v[7]), "v"(b1.v[1]), "v"(a2.v[1]), "v"(b2.v[7]), "v"(a2.v[2]), "v"(b2.v[6]), "v"(a2.v[3]), "v"(b2.v[5]), "v"(a2.v[4]), "v"(b2.v[4]), "v"(a2.v[5]), "v"(b2.v[3]), "v"(a2.v[6]), "v"(b2.v[2]), "v"(a2.v[7]), "v"(b2.v[1]), "v"(m1), "v"(p7), "v"(m5), "v"(p3), "v"(m6), "v"(p2));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(m7), "v"(p1));
    r.v[0] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 9: 21 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[2]), "v"(b0.v[7]), "v"(a0.v[3]), "v"(b0.v[6]), "v"(a0.v[4]), "v"(b0.v[5]), "v"(a0.v[5]), "v"(b0.v[4]), "v"(a0.v[6]), "v"(b0.v[3]), "v"(a0.v[7]), "v"(b0.v[2]), "v"(a1.v[2]), "v"(b1.v[7]), "v"(a1.v[3]), "v"(b1.v[6]), "v"(a1.v[4]), "v"(b1.v[5]), "v"(a1.v[5]), "v"(b1.v[4]), "v"(a1.v[6]), "v"(b1.v[3]), "v"(a1.v[7]), "v"(b1.v[2]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(a2.v[2]), "v"(b2.v[7]), "v"(a2.v[3]), "v"(b2.v[6]), "v"(a2.v[4]), "v"(b2.v[5]), "v"(a2.v[5]), "v"(b2.v[4]), "v"(a2.v[6]), "v"(b2.v[3]), "v"(a2.v[7]), "v"(b2.v[2]), "v"(m2), "v"(p7), "v"(m6), "v"(p3), "v"(m7), "v"(p2));
    r.v[1] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 10: 17 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[3]), "v"(b0.v[7]), "v"(a0.v[4]), "v"(b0.v[6]), "v"(a0.v[5]), "v"(b0.v[5]), "v"(a0.v[6]), "v"(b0.v[4]), "v"(a0.v[7]), "v"(b0.v[3]), "v"(a1.v[3]), "v"(b1.v[7]), "v"(a1.v[4]), "v"(b1.v[6]), "v"(a1.v[5]), "v"(b1.v[5]), "v"(a1.v[6]), "v"(b1.v[4]), "v"(a1.v[7]), "v"(b1.v[3]), "v"(a2.v[3]), "v"(b2.v[7]), "v"(a2.v[4]), "v"(b2.v[6]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(a2.v[5]), "v"(b2.v[5]), "v"(a2.v[6]), "v"(b2.v[4]), "v"(a2.v[7]), "v"(b2.v[3]), "v"(m3), "v"(p7), "v"(m7), "v"(p3));
    r.v[2] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 11: 13 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %25, %26, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[4]), "v"(b0.v[7]), "v"(a0.v[5]), "v"(b0.v[6]), "v"(a0.v[6]), "v"(b0.v[5]), "v"(a0.v[7]), "v"(b0.v[4]), "v"(a1.v[4]), "v"(b1.v[7]), "v"(a1.v[5]), "v"(b1.v[6]), "v"(a1.v[6]), "v"(b1.v[5]), "v"(a1.v[7]), "v"(b1.v[4]), "v"(a2.v[4]), "v"(b2.v[7]), "v"(a2.v[5]), "v"(b2.v[6]), "v"(a2.v[6]), "v"(b2.v[5]), "v"(a2.v[7]), "v"(b2.v[4]));
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "+&v"(hi), "=&s"(cc) : "v"(m4), "v"(p7));
    r.v[3] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 12: 10 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %17, %18, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %21, %22, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[5]), "v"(b0.v[7]), "v"(a0.v[6]), "v"(b0.v[6]), "v"(a0.v[7]), "v"(b0.v[5]), "v"(a1.v[5]), "v"(b1.v[7]), "v"(a1.v[6]), "v"(b1.v[6]), "v"(a1.v[7]), "v"(b1.v[5]), "v"(a2.v[5]), "v"(b2.v[7]), "v"(a2.v[6]), "v"(b2.v[6]), "v"(a2.v[7]), "v"(b2.v[5]), "v"(m5), "v"(p7));
    r.v[4] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 13: 7 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %13, %14, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[6]), "v"(b0.v[7]), "v"(a0.v[7]), "v"(b0.v[6]), "v"(a1.v[6]), "v"(b1.v[7]), "v"(a1.v[7]), "v"(b1.v[6]), "v"(a2.v[6]), "v"(b2.v[7]), "v"(a2.v[7]), "v"(b2.v[6]), "v"(m6), "v"(p7));
    r.v[5] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    // column 14: 4 products
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, 0, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %9, %10, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+&v"(acc), "=&v"(hi), "=&s"(cc) : "v"(a0.v[7]), "v"(b0.v[7]), "v"(a1.v[7]), "v"(b1.v[7]), "v"(a2.v[7]), "v"(b2.v[7]), "v"(m7), "v"(p7));
    r.v[6] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)hi << 32);
    r.v[7] = (uint32_t)acc;                                       // result < 2p < 2^256
    return RED ? fe_cond_sub_p<F>(r) : r;                 // RED = false: result < sum(a_i b_i) / 2^256 + p, left to the caller
}

#endif
template <int F> MB_HD fe_t fe_mul(const fe_t &a, const fe_t &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_mul_device<F>(a, b);
#else
    return fe_mul_portable<F>(a, b);
#endif
}
template <int F> MB_HD fe_t fe_sqr(const fe_t &a) { return fe_mul<F>(a, a); }
// a0*b0 + a1*b1 (+ a2*b2) with a single Montgomery reduction on the device (see tools/gen_fe_mul.py)
template <int F> MB_HD fe_t fe_dot2(const fe_t &a0, const fe_t &b0, const fe_t &a1, const fe_t &b1) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_dot2_device<F>(a0, b0, a1, b1);
#else
    return fe_add<F>(fe_mul<F>(a0, b0), fe_mul<F>(a1, b1));
#endif
}
template <int F> MB_HD fe_t fe_dot3(const fe_t &a0, const fe_t &b0, const fe_t &a1, const fe_t &b1, const fe_t &a2, const fe_t &b2) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_dot3_device<F>(a0, b0, a1, b1, a2, b2);
#else
    return fe_add<F>(fe_add<F>(fe_mul<F>(a0, b0), fe_mul<F>(a1, b1)), fe_mul<F>(a2, b2));
#endif
}

// Constants (computed once on the host, see fp_host.h) that kernels need by value.
template <int F> struct FieldConsts {
    fe_t one;    // R mod p
    fe_t r2;     // R^2 mod p
};

// canonical little-endian bytes (as 8 u32 words) <-> Montgomery
// ---- lazy forms for the Poseidon rounds (sponge.cuh).  A Montgomery product of a, b returns t < a b / 2^256 + p; with p / 2^256 =
// 1/4 + 2^-131 the bound "< 2p + small" is closed under products of values below 2p + small, and everything stays far below
// 2^256 = 3.99 p.  So the x^7 chain and the MDS row skip their conditional subtractions (5 x 16 VALU instructions per round) and the
// round ends with ONE: s = row + round constant (< 2.5p + p, plain 256-bit add), minus 2p if s >= 2p  ->  s < 2p.  The permutation
// normalises its state to [0, p) before returning (fe_cond_sub_p), so nothing outside ever sees a non-canonical value.
template <int F> MB_HD fe_t fe_mul_nr(const fe_t &a, const fe_t &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_mul_device<F, false>(a, b);
#else
    return fe_mul_portable<F>(a, b);
#endif
}
template <int F> MB_HD fe_t fe_dot3_nr(const fe_t &a0, const fe_t &b0, const fe_t &a1, const fe_t &b1, const fe_t &a2, const fe_t &b2) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_dot3_device<F, false>(a0, b0, a1, b1, a2, b2);
#else
    return fe_add<F>(fe_add<F>(fe_mul<F>(a0, b0), fe_mul<F>(a1, b1)), fe_mul<F>(a2, b2));
#endif
}
template <int F> struct TwoP {      // 2p = (2, T1, T2, T3, 0, 0, 0, 0x80000000): P3 < 2^30, so nothing carries past limb 3
    static constexpr uint32_t T1 = FieldP<F>::P1 << 1, T2 = (FieldP<F>::P2 << 1) | (FieldP<F>::P1 >> 31), T3 = (FieldP<F>::P3 << 1) | (FieldP<F>::P2 >> 31);
};
// a + b (no carry out of 256 bits: caller's bound), minus 2p if the sum is >= 2p
template <int F> MB_HD fe_t fe_add_csub2p(const fe_t &a, const fe_t &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    fe_t r;
    uint32_t s0, s1, s2, s3, s4, s5, s6, s7, d0, d1, d2, d3, d4, d5, d6, d7;
    asm("v_add_co_u32_e32 %0, vcc, %16, %24\n\t"
        "v_addc_co_u32_e32 %1, vcc, %17, %25, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %18, %26, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, %19, %27, vcc\n\t"
        "v_addc_co_u32_e32 %4, vcc, %20, %28, vcc\n\t"
        "v_addc_co_u32_e32 %5, vcc, %21, %29, vcc\n\t"
        "v_addc_co_u32_e32 %6, vcc, %22, %30, vcc\n\t"
        "v_addc_co_u32_e32 %7, vcc, %23, %31, vcc\n\t"
        "v_subrev_co_u32_e32 %8, vcc, 2, %0\n\t"
        "v_subbrev_co_u32_e32 %9, vcc, %32, %1, vcc\n\t"
        "v_subbrev_co_u32_e32 %10, vcc, %33, %2, vcc\n\t"
        "v_subbrev_co_u32_e32 %11, vcc, %34, %3, vcc\n\t"
        "v_subbrev_co_u32_e32 %12, vcc, 0, %4, vcc\n\t"
        "v_subbrev_co_u32_e32 %13, vcc, 0, %5, vcc\n\t"
        "v_subbrev_co_u32_e32 %14, vcc, 0, %6, vcc\n\t"
        "v_subbrev_co_u32_e32 %15, vcc, %35, %7, vcc\n\t"
        "v_cndmask_b32_e32 %8, %8, %0, vcc\n\t"
        "v_cndmask_b32_e32 %9, %9, %1, vcc\n\t"
        "v_cndmask_b32_e32 %10, %10, %2, vcc\n\t"
        "v_cndmask_b32_e32 %11, %11, %3, vcc\n\t"
        "v_cndmask_b32_e32 %12, %12, %4, vcc\n\t"
        "v_cndmask_b32_e32 %13, %13, %5, vcc\n\t"
        "v_cndmask_b32_e32 %14, %14, %6, vcc\n\t"
        "v_cndmask_b32_e32 %15, %15, %7, vcc"
        : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(s4), "=&v"(s5), "=&v"(s6), "=&v"(s7),
          "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]),
          "v"(TwoP<F>::T1), "v"(TwoP<F>::T2), "v"(TwoP<F>::T3), "v"(0x80000000u)
        : "vcc");
    r.v[0] = d0; r.v[1] = d1; r.v[2] = d2; r.v[3] = d3; r.v[4] = d4; r.v[5] = d5; r.v[6] = d6; r.v[7] = d7;
    return r;
#else
    return fe_add<F>(a, b);      // host: canonical operands, canonical result (the lazy forms are a device-side economy)
#endif
}

template <int F> MB_HD fe_t fe_to_mont(const fe_t &a, const fe_t &r2) { return fe_mul<F>(a, r2); }
template <int F> MB_HD fe_t fe_from_mont(const fe_t &a) {
    fe_t one = fe_zero(); one.v[0] = 1; return fe_mul<F>(a, one);
}

// a^e, e given as 8 x u32 plain integer (fixed, public exponents only: inversion / sqrt / Legendre)
template <int F> MB_HD fe_t fe_pow(const fe_t &a, const fe_t &e, const fe_t &one) {
    fe_t acc = one; bool started = false;
    for (int i = 255; i >= 0; --i) {
        if (started) acc = fe_sqr<F>(acc);
        if ((e.v[i >> 5] >> (i & 31)) & 1u) { acc = fe_mul<F>(acc, a); started = true; }
    }
    return acc;
}

}  // namespace mb
