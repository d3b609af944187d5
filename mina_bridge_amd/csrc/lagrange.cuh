// lagrange.cuh -- Lagrange-basis commitments of the SRS: L_i = (1/n) * sum_j w^(-ij) * g_j  (n = 2^k, w = primitive n-th root
// of unity of the scalar field).
//
// Replaces poly-commitment `SRS::add_lagrange_basis(domain)` = ark-poly `Radix2EvaluationDomain::ifft_in_place` applied to the
// SRS points as group elements (pin core/Cargo.toml:16,21).  kimchi's verifier takes the public-input commitment as an MSM
// over lagrange[0..npub] (SURVEY.md 8a row a11, appendix A last bullet).  The root of unity is ark's
// TWO_ADIC_ROOT_OF_UNITY^(2^(32-k)) with TWO_ADIC_ROOT_OF_UNITY = 5^t -- the same constant whose choice is pinned by the
// y-signs of the in-tree SRS files (it drives Tonelli-Shanks).
//
// Radix-2 decimation-in-time over XYZZ points; every butterfly needs one 255-bit scalar multiplication by a twiddle
// factor, done by ONE DPP quad with the lane-cooperative group law (double: 3 dependent products, add: 4).
#pragma once
#include "groupmap.cuh"
#include "ec29.cuh"

namespace mb {

// tw[j] = w_inv^j for j < n/2 and tw[n/2] = 1/n, canonical (non-Montgomery) words: they are used as scalars
template <int FS>
__global__ void lagrange_twiddles_kernel(uint32_t half, FieldK ks, fe_t w_inv /* Montgomery */, fe_t n_inv /* Montgomery */,
                                         uint32_t *__restrict__ tw_words) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > half) return;
    fe_t r;
    if (j == half) r = n_inv;
    else { fe_t e = fe_zero(); e.v[0] = j; r = fe_pow<FS>(w_inv, e, ks.one); }
    r = fe_from_mont<FS>(r);
    for (int i = 0; i < 8; ++i) tw_words[(size_t)j * 8 + i] = r.v[i];
}

template <int FB>
__global__ void lagrange_load_bitrev_kernel(uint32_t n, uint32_t k, FieldK kb, const affine_t *__restrict__ g, xyzz_t *__restrict__ a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = __brev(i) >> (32 - k);
    a[r] = xyzz_from_affine<FB>(g[i], kb.one);
}

// scalar * P with one quad (all four lanes hold identical copies; result identical on all four)
template <int FB>
__device__ __forceinline__ xyzz_t xyzz_scalar_mul_quad(const uint32_t *__restrict__ scalar_words, const xyzz_t &p) {
    uint32_t s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = scalar_words[i];
    int top = 255;
    while (top >= 0 && !((s[top >> 5] >> (top & 31)) & 1u)) --top;
    xyzz_t acc = xyzz_inf();
    for (int b = top; b >= 0; --b) {
        acc = xyzz_dbl_quad<FB>(acc);
        if ((s[b >> 5] >> (b & 31)) & 1u) xyzz_add_quad<FB>(acc, p);
    }
    return acc;
}

// one DIT stage: for butterfly (i, i + half) inside blocks of 2*half:  t = tw * a[i+half];  a[i] = a[i] + t;  a[i+half] = a[i] - t
template <int FB>
__global__ void __launch_bounds__(256)
lagrange_stage_kernel(uint32_t n, uint32_t half, uint32_t tw_stride, const uint32_t *__restrict__ tw_words, xyzz_t *__restrict__ a) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t bf = gid >> 2;                              // butterfly index
    if (bf >= n / 2) return;                                   // whole quads leave together
    const uint32_t pos = bf % half, blk = bf / half;
    const uint32_t i = blk * 2 * half + pos;
    const xyzz_t u = a[i], v = a[i + half];
    xyzz_t t = (pos == 0) ? v : xyzz_scalar_mul_quad<FB>(tw_words + (size_t)pos * tw_stride * 8, v);   // twiddle 1 for pos 0
    xyzz_t hi = u; xyzz_add_quad<FB>(hi, t);
    xyzz_t nt = t; nt.y = fe_neg<FB>(nt.y);
    xyzz_t lo = u; xyzz_add_quad<FB>(lo, nt);
    if ((gid & 3u) == 0) { a[i] = hi; a[i + half] = lo; }
}

// scale by 1/n and normalise to affine (Montgomery): one quad per point for the scalar multiplication, lane 0 inverts
template <int FB>
__global__ void __launch_bounds__(256)
lagrange_finish_kernel(uint32_t n, FieldK kb, const uint32_t *__restrict__ n_inv_words, const xyzz_t *__restrict__ a, affine_t *__restrict__ out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 2;
    if (i >= n) return;
    const xyzz_t t = xyzz_scalar_mul_quad<FB>(n_inv_words, a[i]);
    if ((gid & 3u) != 0) return;
    affine_t r; r.x = fe_zero(); r.y = fe_zero();
    if (!xyzz_is_inf(t)) {
        fe_t zi = fe_inv<FB>(fe_mul<FB>(t.zz, t.zzz), kb);
        r.x = fe_mul<FB>(t.x, fe_mul<FB>(zi, t.zzz));
        r.y = fe_mul<FB>(t.y, fe_mul<FB>(zi, t.zz));
    }
    out[i] = r;
}

// batched public-input commitments: out[m] = h - A[m], A[m] = sum_i pub[m][i] * lagrange_i from the table MSM.
// One lane per problem; 17 canonical words each (x || y, infinity flag).
template <int FB>
__global__ void __launch_bounds__(64)
pubcomm_finish_kernel(uint32_t batch, FieldK kb, const affine_t *__restrict__ h, const xyzz_t *__restrict__ a, uint32_t *__restrict__ out_words) { mb_wave_prio();
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= batch) return;
    xyzz_t t = a[m];
    t.y = fe_neg<FB>(t.y);
    const affine_t H = *h;
    xyzz_add_affine<FB>(t, H.x, H.y, kb.one);
    uint32_t *o = out_words + (size_t)m * 17;
    if (xyzz_is_inf(t)) { for (int i = 0; i < 16; ++i) o[i] = 0; o[16] = 1; return; }
    const fe_t zi = fe_inv<FB>(fe_mul<FB>(t.zz, t.zzz), kb);
    const fe_t x = fe_from_mont<FB>(fe_mul<FB>(t.x, fe_mul<FB>(zi, t.zzz))), y = fe_from_mont<FB>(fe_mul<FB>(t.y, fe_mul<FB>(zi, t.zz)));
    for (int i = 0; i < 8; ++i) { o[i] = x.v[i]; o[8 + i] = y.v[i]; }
    o[16] = 0;
}

// ---------------------------------------------------------------- direct public-input commitments (the batch's per-proof 40-term MSMs)
// The generic multi-problem MSM gives every proof its own 128 buckets: sort, accumulate, 2-D bucket reduction -- 1.64 G of a step's 19.4 G
// VALU instructions and ten launches, a third of it the reduction of buckets that hold ~10 points each.  With npub <= 64 fixed bases the
// multiples themselves fit a table (d * 2^(8w) * L_i for d = 1..128: npub x 32 x 128 points = 10.5 MB at npub = 40), so a commitment is
// 32 npub table lookups added straight into one accumulator: no buckets, no sort, no reduction, ONE launch.  Eight lanes per proof take
// the scalars round-robin (5 each at npub = 40) and their partial sums are added by shuffles.  Same signed-digit rule as msm_entry.
static constexpr uint32_t LAGD_MAX_POINTS = 64, LAGD_WINDOWS = 32, LAGD_DIGITS = 128;
// digits[((i * 32 + w) * 128) + (d - 1)] = d * window[w * stride + i], Montgomery affine, (0, 0) = infinity; one thread per (i, w)
template <int F>
__global__ void lagrange_digit_table_kernel(uint32_t n_pts, uint32_t stride, FieldK fk, const affine_t *__restrict__ window, affine_t *__restrict__ digits) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pts * LAGD_WINDOWS) return;
    const uint32_t i = t / LAGD_WINDOWS, w = t % LAGD_WINDOWS;
    const affine_t P = window[(size_t)w * stride + i];
    affine_t *out = digits + (size_t)t * LAGD_DIGITS;
    xyzz_t acc = xyzz_inf();
#pragma unroll 1
    for (uint32_t d = 0; d < LAGD_DIGITS; ++d) {
        affine_t o; o.x = fe_zero(); o.y = fe_zero();
        if (!aff_is_inf(P)) {
            xyzz_add_affine<F>(acc, P.x, P.y, fk.one);           // (d + 1) P
            if (!xyzz_is_inf(acc)) {
                const fe_t zi = fe_inv<F>(fe_mul<F>(acc.zz, acc.zzz), fk);
                o.x = fe_mul<F>(acc.x, fe_mul<F>(zi, acc.zzz)); o.y = fe_mul<F>(acc.y, fe_mul<F>(zi, acc.zz));
            }
        }
        out[d] = o;
    }
}
// out[b] = sum_i pub[b][i] * L_i (XYZZ); pub = canonical 32-byte scalars below 2^255 (words); LPP lanes per proof: 8 for chip-filling
// batches (5 scalars = 160 additions per lane at npub = 40), 64 for small ones (one scalar = 32 additions per lane, then a 6-level sum)
template <int F, int LPP>
__global__ void __launch_bounds__(64)
pubcomm_direct_kernel(uint32_t batch, uint32_t npub, FieldK fk, const affine_t *__restrict__ digits, const uint32_t *__restrict__ pub, xyzz_t *__restrict__ out) { mb_wave_prio();
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x, b = gid / LPP, l = gid % LPP;
    const bool live = b < batch;
    xyzz_t acc = xyzz_inf();
    if (live) {
#pragma unroll 1
        for (uint32_t i = l; i < npub; i += LPP) {
            const uint32_t *sc = pub + ((size_t)b * npub + i) * 8;
            const affine_t *row = digits + (size_t)i * LAGD_WINDOWS * LAGD_DIGITS;
            uint32_t carry = 0, word = 0;
#pragma unroll 1
            for (uint32_t w = 0; w < LAGD_WINDOWS; ++w) {
                if ((w & 3u) == 0) word = sc[w >> 2];
                uint32_t d = ((word >> (8 * (w & 3u))) & 0xffu) + carry;
                const bool neg = d > 128u;
                carry = neg ? 1u : 0u;
                if (neg) d = 256u - d;
                if (d == 0) continue;
                affine_t P = row[(size_t)w * LAGD_DIGITS + (d - 1)];
                if (aff_is_inf(P)) continue;
                if (neg) P.y = fe_neg<F>(P.y);
                xyzz_add_affine<F>(acc, P.x, P.y, fk.one);
            }
        }
    }
#pragma unroll 1
    for (int d = LPP / 2; d >= 1; d >>= 1) { const xyzz_t o = shfl_down_xyzz(acc, d); if ((int)l + d < LPP) xyzz_add<F>(acc, o); }
    if (live && l == 0) out[b] = acc;
}

// The same sums with the mixed adds on 29-bit limbs (ec29.cuh: the lazy XYZZ law of the MSM's accumulate kernels; round 5).  digits29 = the digit table's
// 2^261-domain twin (api_srs.hip build_lagrange_table).  A lane whose running sum meets one of the law's exceptional cases (P = +-acc; found exactly on the lazy limbs)
// starts over with the complete 8 x 32 law on the 2^256-domain table -- no queue: the lanes are independent and the case does not occur for honest inputs.
template <int F, int LPP>
__global__ void __launch_bounds__(64)
pubcomm_direct29_kernel(uint32_t batch, uint32_t npub, FieldK fk, const affine_t *__restrict__ digits, const affine_t *__restrict__ digits29, const uint32_t *__restrict__ pub,
                        xyzz_t *__restrict__ out) { mb_wave_prio();
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x, b = gid / LPP, l = gid % LPP;
    const bool live = b < batch;
    xyzz_t acc = xyzz_inf();
    if (live) {
        xyzz29_t a29; bool inf = true, exact = true;
#pragma unroll 1
        for (uint32_t i = l; i < npub && exact; i += LPP) {
            const uint32_t *sc = pub + ((size_t)b * npub + i) * 8;
            const affine_t *row = digits29 + (size_t)i * LAGD_WINDOWS * LAGD_DIGITS;
            uint32_t carry = 0, word = 0;
#pragma unroll 1
            for (uint32_t w = 0; w < LAGD_WINDOWS && exact; ++w) {
                if ((w & 3u) == 0) word = sc[w >> 2];
                uint32_t d = ((word >> (8 * (w & 3u))) & 0xffu) + carry;
                const bool neg = d > 128u;
                carry = neg ? 1u : 0u;
                if (neg) d = 256u - d;
                if (d == 0) continue;
                const affine_t P = row[(size_t)w * LAGD_DIGITS + (d - 1)];
                if (aff_is_inf(P)) continue;
                exact = xyzz29_add_affine<F>(a29, inf, fe29_from_words(P.x), fe29_from_words(P.y), neg, fk.m32);
            }
        }
        if (exact) acc = xyzz29_leave<F>(a29, inf, fk.one);
        else {
#pragma unroll 1
            for (uint32_t i = l; i < npub; i += LPP) {
                const uint32_t *sc = pub + ((size_t)b * npub + i) * 8;
                const affine_t *row = digits + (size_t)i * LAGD_WINDOWS * LAGD_DIGITS;
                uint32_t carry = 0, word = 0;
#pragma unroll 1
                for (uint32_t w = 0; w < LAGD_WINDOWS; ++w) {
                    if ((w & 3u) == 0) word = sc[w >> 2];
                    uint32_t d = ((word >> (8 * (w & 3u))) & 0xffu) + carry;
                    const bool neg = d > 128u;
                    carry = neg ? 1u : 0u;
                    if (neg) d = 256u - d;
                    if (d == 0) continue;
                    affine_t P = row[(size_t)w * LAGD_DIGITS + (d - 1)];
                    if (aff_is_inf(P)) continue;
                    if (neg) P.y = fe_neg<F>(P.y);
                    xyzz_add_affine<F>(acc, P.x, P.y, fk.one);
                }
            }
        }
    }
#pragma unroll 1
    for (int d = LPP / 2; d >= 1; d >>= 1) { const xyzz_t o = shfl_down_xyzz(acc, d); if ((int)l + d < LPP) xyzz_add<F>(acc, o); }
    if (live && l == 0) out[b] = acc;
#endif
}

}  // namespace mb
