// ec.cuh -- Pallas / Vesta group law for the MSM kernels (y^2 = x^3 + 5 over Fp / Fq).
//
// Replaces ark-ec 0.3 `GroupAffine` / `GroupProjective` mixed/full addition used by
// `VariableBaseMSM::multi_scalar_mul` (pin: core/Cargo.toml:20,49).  The reference accumulates in
// Jacobian coordinates; here buckets are kept in extended Jacobian "XYZZ" (X, Y, ZZ, ZZZ with
// x = X/ZZ, y = Y/ZZZ): mixed add 8M+2S, full add 12M+2S, no field inversion until the very end.
// The group element computed is the same; results are compared in canonical affine form.
//
// All special cases (infinity, P = Q, P = -Q) are handled: bucket sums can collide.
#pragma once
#include "fp.cuh"

namespace mb {

struct alignas(16) affine_t { fe_t x, y; };            // Montgomery; infinity encoded as (0, 0)
struct alignas(16) xyzz_t { fe_t x, y, zz, zzz; };     // zz == 0  <=>  infinity

MB_HD bool aff_is_inf(const affine_t &p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
MB_HD bool xyzz_is_inf(const xyzz_t &p) { return fe_is_zero(p.zz); }
MB_HD xyzz_t xyzz_inf() { xyzz_t r; r.x = fe_zero(); r.y = fe_zero(); r.zz = fe_zero(); r.zzz = fe_zero(); return r; }

template <int F> MB_HD xyzz_t xyzz_from_affine(const affine_t &p, const fe_t &one) {
    xyzz_t r;
    if (aff_is_inf(p)) return xyzz_inf();
    r.x = p.x; r.y = p.y; r.zz = one; r.zzz = one; return r;
}

// 2 * (x, y) affine -> XYZZ  (mdbl-2008-s-1, a = 0)
template <int F> MB_HD xyzz_t xyzz_dbl_affine(const fe_t &x, const fe_t &y) {
    xyzz_t r;
    if (fe_is_zero(y)) return xyzz_inf();          // order-2 point (none on these curves, kept for completeness)
    fe_t u = fe_dbl<F>(y);
    fe_t v = fe_sqr<F>(u);
    fe_t w = fe_mul<F>(u, v);
    fe_t s = fe_mul<F>(x, v);
    fe_t x2 = fe_sqr<F>(x);
    fe_t m = fe_add<F>(fe_dbl<F>(x2), x2);
    r.x = fe_sub<F>(fe_sub<F>(fe_sqr<F>(m), s), s);
    r.y = fe_dot2<F>(m, fe_sub<F>(s, r.x), w, fe_neg<F>(y));       // m (s - x3) - w y, one reduction
    r.zz = v; r.zzz = w;
    return r;
}

// 2 * P, XYZZ (dbl-2008-s-1, a = 0)
template <int F> MB_HD xyzz_t xyzz_dbl(const xyzz_t &p) {
    if (xyzz_is_inf(p) || fe_is_zero(p.y)) return xyzz_inf();
    xyzz_t r;
    fe_t u = fe_dbl<F>(p.y);
    fe_t v = fe_sqr<F>(u);
    fe_t w = fe_mul<F>(u, v);
    fe_t s = fe_mul<F>(p.x, v);
    fe_t x2 = fe_sqr<F>(p.x);
    fe_t m = fe_add<F>(fe_dbl<F>(x2), x2);
    r.x = fe_sub<F>(fe_sub<F>(fe_sqr<F>(m), s), s);
    r.y = fe_dot2<F>(m, fe_sub<F>(s, r.x), w, fe_neg<F>(p.y));
    r.zz = fe_mul<F>(v, p.zz);
    r.zzz = fe_mul<F>(w, p.zzz);
    return r;
}

// acc += (qx, qy) affine, not infinity  (madd-2008-s)
template <int F> MB_HD void xyzz_add_affine(xyzz_t &acc, const fe_t &qx, const fe_t &qy, const fe_t &one) {
    if (xyzz_is_inf(acc)) { acc.x = qx; acc.y = qy; acc.zz = one; acc.zzz = one; return; }
    fe_t u2 = fe_mul<F>(qx, acc.zz);
    fe_t s2 = fe_mul<F>(qy, acc.zzz);
    fe_t p = fe_sub<F>(u2, acc.x);
    fe_t r = fe_sub<F>(s2, acc.y);
    if (fe_is_zero(p)) {
        if (fe_is_zero(r)) acc = xyzz_dbl_affine<F>(qx, qy);
        else acc = xyzz_inf();
        return;
    }
    fe_t pp = fe_sqr<F>(p);
    fe_t ppp = fe_mul<F>(p, pp);
    fe_t q = fe_mul<F>(acc.x, pp);
    fe_t x3 = fe_sub<F>(fe_sub<F>(fe_sub<F>(fe_sqr<F>(r), ppp), q), q);
    fe_t y3 = fe_dot2<F>(r, fe_sub<F>(q, x3), fe_neg<F>(acc.y), ppp);   // r (q - x3) - y1 ppp, one reduction
    acc.zz = fe_mul<F>(acc.zz, pp);
    acc.zzz = fe_mul<F>(acc.zzz, ppp);
    acc.x = x3; acc.y = y3;
}

// acc += q (XYZZ)  (add-2008-s)
template <int F> MB_HD void xyzz_add(xyzz_t &acc, const xyzz_t &q) {
    if (xyzz_is_inf(q)) return;
    if (xyzz_is_inf(acc)) { acc = q; return; }
    fe_t u1 = fe_mul<F>(acc.x, q.zz);
    fe_t u2 = fe_mul<F>(q.x, acc.zz);
    fe_t s1 = fe_mul<F>(acc.y, q.zzz);
    fe_t s2 = fe_mul<F>(q.y, acc.zzz);
    fe_t p = fe_sub<F>(u2, u1);
    fe_t r = fe_sub<F>(s2, s1);
    if (fe_is_zero(p)) {
        if (fe_is_zero(r)) acc = xyzz_dbl<F>(acc);
        else acc = xyzz_inf();
        return;
    }
    fe_t pp = fe_sqr<F>(p);
    fe_t ppp = fe_mul<F>(p, pp);
    fe_t qq = fe_mul<F>(u1, pp);
    fe_t x3 = fe_sub<F>(fe_sub<F>(fe_sub<F>(fe_sqr<F>(r), ppp), qq), qq);
    fe_t y3 = fe_dot2<F>(r, fe_sub<F>(qq, x3), fe_neg<F>(s1), ppp);
    acc.zz = fe_mul<F>(fe_mul<F>(acc.zz, q.zz), pp);
    acc.zzz = fe_mul<F>(fe_mul<F>(acc.zzz, q.zzz), ppp);
    acc.x = x3; acc.y = y3;
}

// ---------------------------------------------------------------- lane-cooperative group law (gfx950 device only)
// Four lanes of one DPP quad hold IDENTICAL copies of the operands and cooperate on one XYZZ operation: the
// independent field products of each dependency level run on different lanes (4 levels for add, 3 for double,
// instead of 14 / 9 products in sequence), results are exchanged with quad_perm DPP moves.  Every lane ends with the
// identical, bit-exact same XYZZ value the single-lane routines produce.  For the latency-bound stages of the MSM
// (few live values, long dependency chains); all branches below are uniform within a quad.
#if defined(__HIPCC__)
template <int K> __device__ __forceinline__ fe_t quad_bcast(const fe_t &a) {
    fe_t r = a;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < 8; ++i)
        r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], K * 0x55, 0xf, 0xf, true);   // quad_perm:[K,K,K,K]
#endif
    return r;
}
__device__ __forceinline__ fe_t quad_sel4(uint32_t rho, const fe_t &a, const fe_t &b, const fe_t &c, const fe_t &d) {
    fe_t r; const bool b0 = rho & 1u, b1 = rho & 2u;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint32_t x = b0 ? b.v[i] : a.v[i], y = b0 ? d.v[i] : c.v[i]; r.v[i] = b1 ? y : x; }
    return r;
}
template <int F> __device__ __forceinline__ xyzz_t xyzz_dbl_quad(const xyzz_t &p) {
    if (xyzz_is_inf(p) || fe_is_zero(p.y)) return xyzz_inf();
    const uint32_t rho = threadIdx.x & 3u;
    const fe_t u = fe_dbl<F>(p.y);
    fe_t a = quad_sel4(rho, u, p.x, u, u);
    fe_t m = fe_mul<F>(a, a);
    const fe_t v = quad_bcast<0>(m), xx = quad_bcast<1>(m);
    const fe_t mm1 = fe_add<F>(fe_dbl<F>(xx), xx);                       // M = 3 X^2
    a = quad_sel4(rho, u, p.x, p.zz, mm1);
    fe_t b = quad_sel4(rho, v, v, v, mm1);
    m = fe_mul<F>(a, b);
    const fe_t w = quad_bcast<0>(m), s = quad_bcast<1>(m), zz3 = quad_bcast<2>(m), msq = quad_bcast<3>(m);
    xyzz_t r;
    r.x = fe_sub<F>(fe_sub<F>(msq, s), s);
    a = quad_sel4(rho, mm1, w, w, w);
    b = quad_sel4(rho, fe_sub<F>(s, r.x), p.y, p.zzz, p.zzz);
    m = fe_mul<F>(a, b);
    r.y = fe_sub<F>(quad_bcast<0>(m), quad_bcast<1>(m));
    r.zz = zz3; r.zzz = quad_bcast<2>(m);
    return r;
}
template <int F> __device__ __forceinline__ void xyzz_add_quad(xyzz_t &acc, const xyzz_t &q) {
    if (xyzz_is_inf(q)) return;
    if (xyzz_is_inf(acc)) { acc = q; return; }
    const uint32_t rho = threadIdx.x & 3u;
    fe_t a = quad_sel4(rho, acc.x, q.x, acc.y, q.y);
    fe_t b = quad_sel4(rho, q.zz, acc.zz, q.zzz, acc.zzz);
    fe_t m = fe_mul<F>(a, b);
    const fe_t u1 = quad_bcast<0>(m), u2 = quad_bcast<1>(m), s1 = quad_bcast<2>(m), s2 = quad_bcast<3>(m);
    const fe_t p = fe_sub<F>(u2, u1), r = fe_sub<F>(s2, s1);
    if (fe_is_zero(p)) {
        if (fe_is_zero(r)) acc = xyzz_dbl_quad<F>(acc); else acc = xyzz_inf();
        return;
    }
    a = quad_sel4(rho, p, r, acc.zz, acc.zzz);
    b = quad_sel4(rho, p, r, q.zz, q.zzz);
    m = fe_mul<F>(a, b);
    const fe_t pp = quad_bcast<0>(m), rr = quad_bcast<1>(m), zz12 = quad_bcast<2>(m), zzz12 = quad_bcast<3>(m);
    a = quad_sel4(rho, p, u1, zz12, zz12);
    m = fe_mul<F>(a, pp);
    const fe_t ppp = quad_bcast<0>(m), qq = quad_bcast<1>(m), zz3 = quad_bcast<2>(m);
    const fe_t x3 = fe_sub<F>(fe_sub<F>(fe_sub<F>(rr, ppp), qq), qq);
    a = quad_sel4(rho, r, s1, zzz12, zzz12);
    b = quad_sel4(rho, fe_sub<F>(qq, x3), ppp, ppp, ppp);
    m = fe_mul<F>(a, b);
    acc.x = x3;
    acc.y = fe_sub<F>(quad_bcast<0>(m), quad_bcast<1>(m));
    acc.zz = zz3; acc.zzz = quad_bcast<2>(m);
}
#endif

}  // namespace mb
