// api_shard.hip -- device-resident building blocks of the multi-GPU variants: one process per GPU moves BYTES with
// RCCL (all-to-all of scalar slices, all-gather of partial points); the reduction operators -- modular addition of scalar vectors,
// Pasta point addition -- are these kernels, because RCCL has no field / curve reduction op.  Every pointer is a device pointer;
// everything is queued on the context's next pipeline lane without host synchronisation.
#include "ctx.h"
#include "msm.cuh"
#include "sponge.cuh"

namespace mb {

template <int F>
__global__ void field_sum_rows_kernel(uint32_t rows, uint32_t m, FieldK fk, const uint32_t *__restrict__ in /* rows*m*8 canonical */, uint32_t *__restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    fe_t acc = fe_zero();                                          // canonical integers added modulo p: no Montgomery form needed
    for (uint32_t r = 0; r < rows; ++r) { fe_t v; for (int i = 0; i < 8; ++i) v.v[i] = in[((size_t)r * m + j) * 8 + i]; acc = fe_add<F>(acc, v); }
    for (int i = 0; i < 8; ++i) out[(size_t)j * 8 + i] = acc.v[i];
}

// sum of n affine points given as 17-word records {x, y, is_infinity} -> one record
template <int F>
__global__ void points_sum_kernel(uint32_t n, FieldK kb, const uint32_t *__restrict__ recs, uint32_t *__restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    xyzz_t acc = xyzz_inf();
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t *r = recs + (size_t)i * 17;
        if (r[16]) continue;
        fe_t x, y; for (int q = 0; q < 8; ++q) { x.v[q] = r[q]; y.v[q] = r[8 + q]; }
        xyzz_add_affine<F>(acc, fe_to_mont<F>(x, kb.r2), fe_to_mont<F>(y, kb.r2), kb.one);
    }
    if (xyzz_is_inf(acc)) { for (int i = 0; i < 16; ++i) out[i] = 0; out[16] = 1; return; }
    const fe_t zi = fe_inv<F>(fe_mul<F>(acc.zz, acc.zzz), kb);
    const fe_t x = fe_from_mont<F>(fe_mul<F>(acc.x, fe_mul<F>(zi, acc.zzz))), y = fe_from_mont<F>(fe_mul<F>(acc.y, fe_mul<F>(zi, acc.zz)));
    for (int i = 0; i < 8; ++i) { out[i] = x.v[i]; out[8 + i] = y.v[i]; }
    out[16] = 0;
}

__global__ void records_equal_kernel(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint32_t *__restrict__ verdict) {
    if (threadIdx.x || blockIdx.x) return;
    bool eq = (a[16] != 0) == (b[16] != 0);
    if (eq && !a[16]) for (int i = 0; i < 16; ++i) eq = eq && a[i] == b[i];
    *verdict = eq ? 1u : 0u;
}

}  // namespace mb

extern "C" int mina_challenge_to_field_dev(mina_ctx *c, int field, size_t n, const void *d_chal128, void *d_out) {
    if (!c || (n && (!d_chal128 || !d_out))) return fail(MINA_ERR_ARG, "null argument");
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    if (n == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    DISPATCH_FIELD(field, { challenge_to_field_kernel<F_><<<cdiv(n, 64), 64, 0, c->L->stream>>>((uint32_t)n, c->fk[F_], (const uint32_t *)d_chal128, (uint32_t *)d_out); });
    HIPC(hipGetLastError());
    return MINA_OK;
}

extern "C" int mina_field_sum_rows_dev(mina_ctx *c, int field, size_t rows, size_t m, const void *d_in, void *d_out) {
    if (!c || !d_in || !d_out) return fail(MINA_ERR_ARG, "null argument");
    if (bad_field(field) || rows == 0 || m == 0 || rows > 4096 || m > (1u << 24)) return fail(MINA_ERR_ARG, "bad field / shape");
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    DISPATCH_FIELD(field, { mb::field_sum_rows_kernel<F_><<<cdiv(m, 256), 256, 0, c->L->stream>>>((uint32_t)rows, (uint32_t)m, c->fk[F_], (const uint32_t *)d_in, (uint32_t *)d_out); });
    HIPC(hipGetLastError());
    return MINA_OK;
}

extern "C" int mina_msm_srs_range_dev(mina_ctx *c, int curve, uint32_t first, size_t n, const void *d_scalars, void *d_out) {
    if (!c || !d_scalars || !d_out) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (n == 0 || n > 0xffffffffu) return fail(MINA_ERR_ARG, "bad n");
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    return mb_msm_fixed(c, curve, (uint32_t)n, (const uint32_t *)d_scalars, (uint32_t *)d_out, nullptr, first);
}

extern "C" int mina_msm_dev(mina_ctx *c, int curve, size_t n, const void *d_bases, const void *d_scalars, void *d_out) {
    if (!c || !d_bases || !d_scalars || !d_out) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (n == 0 || n > (1u << 24)) return fail(MINA_ERR_ARG, "bad n");
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    int rc;
    MsmWorkspace &w = c->L->ws;
    if ((rc = w.points.ensure(n * sizeof(affine_t)))) return rc;
    DISPATCH_FIELD(base_field_of(curve), { points_to_mont_kernel<F_><<<cdiv(n, 256), 256, 0, c->L->stream>>>((uint32_t)n, (const uint32_t *)d_bases, c->fk[F_].r2, w.points.as<affine_t>()); });
    return mb_msm_variable(c, curve, (uint32_t)n, (const uint32_t *)d_scalars, w.points.p, (uint32_t *)d_out, nullptr);
}

extern "C" int mina_points_sum_dev(mina_ctx *c, int curve, size_t n, const void *d_records, void *d_out) {
    if (!c || !d_records || !d_out) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (n == 0 || n > 65536) return fail(MINA_ERR_ARG, "bad n");
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    DISPATCH_FIELD(base_field_of(curve), { mb::points_sum_kernel<F_><<<1, 64, 0, c->L->stream>>>((uint32_t)n, c->fk[F_], (const uint32_t *)d_records, (uint32_t *)d_out); });
    HIPC(hipGetLastError());
    return MINA_OK;
}

extern "C" int mina_point_records_equal_dev(mina_ctx *c, const void *d_a, const void *d_b, void *d_verdict) {
    if (!c || !d_a || !d_b || !d_verdict) return fail(MINA_ERR_ARG, "null argument");
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    mb::records_equal_kernel<<<1, 64, 0, c->L->stream>>>((const uint32_t *)d_a, (const uint32_t *)d_b, (uint32_t *)d_verdict);
    HIPC(hipGetLastError());
    return MINA_OK;
}
