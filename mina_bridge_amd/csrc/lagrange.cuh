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
pubcomm_finish_kernel(uint32_t batch, FieldK kb, const affine_t *__restrict__ h, const xyzz_t *__restrict__ a, uint32_t *__restrict__ out_words) {
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

}  // namespace mb
