// api_srs.hip -- SRS generation / loading / export and the fixed-base window tables (a6).
#include <atomic>
#include "ctx.h"
#include "msm.cuh"
#include "lagrange.cuh"
#include <type_traits>
#include <vector>

// ------------------------------------------------------------------------------------------------
// SRS
template <int F> static int build_tables(mina_ctx *c, SrsState &s) {
    const FieldK &fk = c->fk[F];
    msm_build_table_kernel<F><<<cdiv(s.depth, 256), 256, 0, c->L->stream>>>(s.depth, s.depth, s.c, s.W, fk.one, fk.pm2, s.table.as<affine_t>());
    HIPC(hipGetLastError());
    const size_t npts = (size_t)s.W * s.depth;                     // the 2^261-domain twin the fp29 accumulate kernels gather from (ec29.cuh)
    msm_table29_kernel<F><<<cdiv(npts, 256), 256, 0, c->L->stream>>>(npts, s.table.as<affine_t>(), fk.m32, s.table29.as<affine_t>());
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(c->L->stream));
    s.table29s.release();                                          // the pre-split form is built on request (mina_srs_split_table)
    return MINA_OK;
}
// The pre-split form of the window table (msm.cuh tab29_t: x, y, p - y as 29-bit limbs, 128 B per point; mina_verify_tuning.msm_fp29 = 2 reads it).  Measured in
// round 5 (profiles/r05_k1.md): 1714 instead of 1745 instructions per mixed add, + 0.5 % checks/s for twice the gather traffic and 128 MiB per curve -- NOT the
// default; built only when a caller asks for it.
extern "C" int mina_srs_split_table(mina_ctx *c, int curve, int on) {
    if (!c) return fail(MINA_ERR_ARG, "null ctx");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded for this curve");
    HIPC(hipSetDevice(c->device));
    for (int i = 0; i < c->nlanes; ++i) HIPC(hipStreamSynchronize(c->lanes[i].stream));      // an MSM in flight may be reading the table
    if (!on) { s.table29s.release(); return MINA_OK; }
    if (s.table29s.p) return MINA_OK;
    c->use_lane0();
    const size_t npts = (size_t)s.W * s.depth;
    int rc;
    if ((rc = s.table29s.ensure(npts * sizeof(tab29_t)))) return rc;
    DISPATCH_FIELD(base_field_of(curve), { msm_table29s_kernel<F_><<<cdiv(npts, 256), 256, 0, c->L->stream>>>(npts, s.table.as<affine_t>(), c->fk[F_].m32, s.table29s.as<tab29_t>()); });
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(c->L->stream));
    return MINA_OK;
}

static int srs_alloc(mina_ctx *c, int curve, uint32_t depth) {
    SrsState &s = c->srs[curve];
    s.depth = 0; s.c = 16; s.W = 16;
    s.lagrange_log2 = -1; s.lagrange_host.clear();
    static std::atomic<uint64_t> gen{0};
    s.srs_gen = ++gen;
    int rc;
    if ((rc = s.table.ensure((size_t)s.W * depth * sizeof(affine_t)))) return rc;
    if ((rc = s.table29.ensure((size_t)s.W * depth * sizeof(affine_t)))) return rc;
    if ((rc = s.h.ensure(sizeof(affine_t)))) return rc;
    return MINA_OK;
}

extern "C" int mina_srs_create(mina_ctx *c, int curve, uint32_t depth) {
    if (!c) return fail(MINA_ERR_ARG, "null ctx");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (depth == 0 || depth > (1u << 20)) return fail(MINA_ERR_ARG, "depth must be in 1..2^20");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc = srs_alloc(c, curve, depth);
    if (rc) return rc;
    SrsState &s = c->srs[curve];
    const int F = base_field_of(curve);
    DISPATCH_FIELD(F, {
        srs_create_kernel<F_><<<cdiv((size_t)depth + 1, 256), 256, 0, c->L->stream>>>(depth, c->fk[F_], s.table.as<affine_t>(), s.h.as<affine_t>());
    });
    HIPC(hipGetLastError());
    s.depth = depth;
    DISPATCH_FIELD(F, { rc = build_tables<F_>(c, s); });
    if (rc) s.depth = 0;
    return rc;
}

extern "C" int mina_srs_load(mina_ctx *c, int curve, const uint8_t *d, size_t len) {
    if (!c || !d) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    // fixarray(2) [ array(n) [ bin8(33) ... ], bin8(33) ]; rmp writes the shortest array header: fixarray (< 16), array16 (< 65536), array32
    if (len < 3 || d[0] != 0x92) return fail(MINA_ERR_FORMAT, "not an SRS MessagePack blob");
    uint32_t n; size_t hdr;
    if ((d[1] & 0xf0) == 0x90) { n = d[1] & 0x0f; hdr = 2; }
    else if (d[1] == 0xdc && len >= 4) { n = ((uint32_t)d[2] << 8) | d[3]; hdr = 4; }
    else if (d[1] == 0xdd && len >= 6) { n = ((uint32_t)d[2] << 24) | ((uint32_t)d[3] << 16) | ((uint32_t)d[4] << 8) | d[5]; hdr = 6; }
    else return fail(MINA_ERR_FORMAT, "not an SRS MessagePack blob");
    if (n == 0 || n > (1u << 20) || len != hdr + (size_t)(n + 1) * 35) return fail(MINA_ERR_FORMAT, "bad SRS length");
    std::vector<uint8_t> blobs((size_t)(n + 1) * 33);
    for (size_t i = 0; i <= n; ++i) {
        const uint8_t *e = d + hdr + i * 35;
        if (e[0] != 0xc4 || e[1] != 33) return fail(MINA_ERR_FORMAT, "bad point header");
        memcpy(&blobs[i * 33], e + 2, 33);
    }
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc = srs_alloc(c, curve, n);
    if (rc) return rc;
    SrsState &s = c->srs[curve];
    if ((rc = c->L->tmp_a.ensure(blobs.size()))) return rc;
    if ((rc = c->L->tmp_b.ensure(4))) return rc;
    HIPC(hipMemcpyAsync(c->L->tmp_a.p, blobs.data(), blobs.size(), hipMemcpyHostToDevice, c->L->stream));
    HIPC(hipMemsetAsync(c->L->tmp_b.p, 0, 4, c->L->stream));
    const int F = base_field_of(curve);
    DISPATCH_FIELD(F, {
        decompress_kernel<F_><<<cdiv(n, 256), 256, 0, c->L->stream>>>(n, c->fk[F_], c->L->tmp_a.as<uint8_t>(), s.table.as<affine_t>(), c->L->tmp_b.as<uint32_t>());
        decompress_kernel<F_><<<1, 64, 0, c->L->stream>>>(1, c->fk[F_], c->L->tmp_a.as<uint8_t>() + (size_t)n * 33, s.h.as<affine_t>(), c->L->tmp_b.as<uint32_t>());
    });
    uint32_t bad = 0;
    HIPC(hipMemcpyAsync(&bad, c->L->tmp_b.p, 4, hipMemcpyDeviceToHost, c->L->stream));
    HIPC(hipStreamSynchronize(c->L->stream));
    if (bad) return fail(MINA_ERR_FORMAT, "SRS contains a point that is not on the curve");
    s.depth = n;
    DISPATCH_FIELD(F, { rc = build_tables<F_>(c, s); });
    if (rc) s.depth = 0;
    return rc;
}

extern "C" uint32_t mina_srs_depth(mina_ctx *c, int curve) {
    if (!c || (curve != 0 && curve != 1)) return 0;
    return c->srs[curve].depth;
}

extern "C" int mina_srs_get_g(mina_ctx *c, int curve, uint32_t first, uint32_t count, uint8_t *out) {
    if (!c || !out) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded");
    if ((uint64_t)first + count > s.depth) return fail(MINA_ERR_ARG, "range outside SRS");
    if (count == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = c->L->tmp_a.ensure((size_t)count * 64))) return rc;
    const int F = base_field_of(curve);
    DISPATCH_FIELD(F, { points_from_mont_kernel<F_><<<cdiv(count, 256), 256, 0, c->L->stream>>>(count, s.table.as<affine_t>() + first, c->L->tmp_a.as<uint32_t>()); });
    HIPC(hipMemcpyAsync(out, c->L->tmp_a.p, (size_t)count * 64, hipMemcpyDeviceToHost, c->L->stream));
    HIPC(hipStreamSynchronize(c->L->stream));
    return MINA_OK;
}

extern "C" int mina_srs_get_h(mina_ctx *c, int curve, uint8_t *out) {
    if (!c || !out) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = c->L->tmp_a.ensure(64))) return rc;
    const int F = base_field_of(curve);
    DISPATCH_FIELD(F, { points_from_mont_kernel<F_><<<1, 64, 0, c->L->stream>>>(1, s.h.as<affine_t>(), c->L->tmp_a.as<uint32_t>()); });
    HIPC(hipMemcpyAsync(out, c->L->tmp_a.p, 64, hipMemcpyDeviceToHost, c->L->stream));
    HIPC(hipStreamSynchronize(c->L->stream));
    return MINA_OK;
}

extern "C" int mina_srs_serialize(mina_ctx *c, int curve, uint8_t *out, size_t cap, size_t *len) {
    if (!c || !len) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded");
    const size_t hdr = s.depth < 16 ? 2 : (s.depth < 65536 ? 4 : 6);          // the minimal array header, as rmp-serde writes it
    const size_t need = hdr + (size_t)(s.depth + 1) * 35;
    *len = need;
    if (!out || cap < need) return fail(MINA_ERR_ARG, "output buffer too small");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = c->L->tmp_a.ensure((size_t)(s.depth + 1) * 33))) return rc;
    const int F = base_field_of(curve);
    DISPATCH_FIELD(F, {
        compress_kernel<F_><<<cdiv(s.depth, 256), 256, 0, c->L->stream>>>(s.depth, c->fk[F_], s.table.as<affine_t>(), c->L->tmp_a.as<uint8_t>());
        compress_kernel<F_><<<1, 64, 0, c->L->stream>>>(1, c->fk[F_], s.h.as<affine_t>(), c->L->tmp_a.as<uint8_t>() + (size_t)s.depth * 33);
    });
    std::vector<uint8_t> blobs((size_t)(s.depth + 1) * 33);
    HIPC(hipMemcpyAsync(blobs.data(), c->L->tmp_a.p, blobs.size(), hipMemcpyDeviceToHost, c->L->stream));
    HIPC(hipStreamSynchronize(c->L->stream));
    out[0] = 0x92;
    if (hdr == 2) out[1] = (uint8_t)(0x90 | s.depth);
    else if (hdr == 4) { out[1] = 0xdc; out[2] = (uint8_t)(s.depth >> 8); out[3] = (uint8_t)s.depth; }
    else { out[1] = 0xdd; out[2] = (uint8_t)(s.depth >> 24); out[3] = (uint8_t)(s.depth >> 16); out[4] = (uint8_t)(s.depth >> 8); out[5] = (uint8_t)s.depth; }
    for (size_t i = 0; i <= s.depth; ++i) {
        uint8_t *e = out + hdr + i * 35;
        e[0] = 0xc4; e[1] = 33; memcpy(e + 2, &blobs[i * 33], 33);
    }
    return MINA_OK;
}


// ------------------------------------------------------------------------------------------------
// Lagrange-basis commitments (poly-commitment `SRS::add_lagrange_basis`)
template <int FB, int FS>
static int run_lagrange(mina_ctx *c, SrsState &s, uint32_t k, uint8_t *out_host) {
    const uint32_t n = 1u << k, half = n / 2;
    const FieldK &kb = c->fk[FB], &ks = c->fk[FS];
    // w = root^(2^(32-k)) (primitive n-th root of unity), w_inv = w^(n-1), n_inv = n^(p-2)
    fe_t w = ks.root;
    for (uint32_t i = 0; i < 32 - k; ++i) w = fe_sqr<FS>(w);
    fe_t e = fe_zero(); e.v[0] = n - 1;
    const fe_t w_inv = fe_pow<FS>(w, e, ks.one);
    fe_t nn = fe_zero(); nn.v[0] = n;
    const fe_t n_inv = fe_inv<FS>(fe_to_mont<FS>(nn, ks.r2), ks);
    int rc;
    if ((rc = c->L->tmp_a.ensure(((size_t)half + 1) * 32))) return rc;          // twiddles
    if ((rc = c->L->tmp_b.ensure((size_t)n * sizeof(xyzz_t)))) return rc;       // working array
    if ((rc = c->L->tmp_c.ensure((size_t)n * sizeof(affine_t)))) return rc;     // affine result (Montgomery)
    if ((rc = c->L->tmp_d.ensure((size_t)n * 64))) return rc;                   // canonical bytes
    hipStream_t st = c->L->stream;
    lagrange_twiddles_kernel<FS><<<cdiv((size_t)half + 1, 256), 256, 0, st>>>(half, ks, w_inv, n_inv, c->L->tmp_a.as<uint32_t>());
    lagrange_load_bitrev_kernel<FB><<<cdiv(n, 256), 256, 0, st>>>(n, k, kb, s.table.as<affine_t>(), c->L->tmp_b.as<xyzz_t>());
    for (uint32_t h = 1; h < n; h <<= 1)
        lagrange_stage_kernel<FB><<<cdiv((size_t)half * 4, 256), 256, 0, st>>>(n, h, half / h, c->L->tmp_a.as<uint32_t>(), c->L->tmp_b.as<xyzz_t>());
    lagrange_finish_kernel<FB><<<cdiv((size_t)n * 4, 256), 256, 0, st>>>(n, kb, c->L->tmp_a.as<uint32_t>() + (size_t)half * 8, c->L->tmp_b.as<xyzz_t>(), c->L->tmp_c.as<affine_t>());
    points_from_mont_kernel<FB><<<cdiv(n, 256), 256, 0, st>>>(n, c->L->tmp_c.as<affine_t>(), c->L->tmp_d.as<uint32_t>());
    HIPC(hipGetLastError());
    return d2h_sync(c, out_host, c->L->tmp_d, (size_t)n * 64);
}

extern "C" int mina_srs_lagrange_basis(mina_ctx *c, int curve, uint32_t log2_domain, uint8_t *out_affine) {
    if (!c || !out_affine) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded");
    if (log2_domain > 20 || ((uint64_t)1 << log2_domain) > s.depth) return fail(MINA_ERR_ARG, "domain larger than the SRS");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    if (log2_domain == 0) return mina_srs_get_g(c, curve, 0, 1, out_affine);   // L_0 = g_0
    if (curve == CURVE_PALLAS) return run_lagrange<FIELD_FP, FIELD_FQ>(c, s, log2_domain, out_affine);
    return run_lagrange<FIELD_FQ, FIELD_FP>(c, s, log2_domain, out_affine);
}

// Lagrange basis of this domain, cached on the host in canonical form (one-off group iFFT on the GPU)
static int ensure_lagrange_host(mina_ctx *c, int curve, uint32_t log2_domain) {
    SrsState &s = c->srs[curve];
    if (s.lagrange_log2 == (int)log2_domain && !s.lagrange_host.empty()) return MINA_OK;
    int rc;
    s.lagrange_host.assign(((size_t)1 << log2_domain) * 64, 0);
    s.lagrange_log2 = -1; s.lagrange_table_log2 = -1;
    if ((rc = mina_srs_lagrange_basis(c, curve, log2_domain, s.lagrange_host.data()))) { s.lagrange_host.clear(); return rc; }
    s.lagrange_log2 = (int)log2_domain;
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------
// kimchi verifier: public-input commitment  public_comm = h - sum_i pub_i * lagrange_i   (a11; `mask_custom` with blinder 1)
extern "C" int mina_public_input_commitment(mina_ctx *c, int curve, uint32_t log2_domain, size_t npub, const uint8_t *public_inputs,
                                            uint8_t *out_affine) {
    if (!c || !out_affine || (npub && !public_inputs)) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded");
    if (log2_domain > 20 || ((uint64_t)1 << log2_domain) > s.depth || npub > ((size_t)1 << log2_domain)) return fail(MINA_ERR_ARG, "bad domain / npub");
    if (npub && !scalars_below_2_255(public_inputs, npub)) return fail(MINA_ERR_ARG, "scalar >= 2^255");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = ensure_lagrange_host(c, curve, log2_domain))) return rc;
    uint8_t acc[64]; memset(acc, 0, 64);
    if (npub && (rc = mina_msm(c, curve, npub, s.lagrange_host.data(), public_inputs, acc))) return rc;
    uint8_t hbytes[64];
    if ((rc = mina_srs_get_h(c, curve, hbytes))) return rc;
    // h - acc on the host with the same (host-compiled) field / group code the kernels use
    auto finish = [&](auto tag) {
        constexpr int F = decltype(tag)::value;
        const FieldK &k = c->fk[F];
        auto load = [&](const uint8_t *b) { affine_t a; memcpy(a.x.v, b, 32); memcpy(a.y.v, b + 32, 32); a.x = fe_to_mont<F>(a.x, k.r2); a.y = fe_to_mont<F>(a.y, k.r2); return a; };
        affine_t H = load(hbytes), A = load(acc);
        xyzz_t t = xyzz_from_affine<F>(H, k.one);
        if (!aff_is_inf(A)) { A.y = fe_neg<F>(A.y); xyzz_add_affine<F>(t, A.x, A.y, k.one); }
        if (xyzz_is_inf(t)) { memset(out_affine, 0, 64); return; }
        fe_t zi = fe_inv<F>(fe_mul<F>(t.zz, t.zzz), k);
        fe_t x = fe_from_mont<F>(fe_mul<F>(t.x, fe_mul<F>(zi, t.zzz))), y = fe_from_mont<F>(fe_mul<F>(t.y, fe_mul<F>(zi, t.zz)));
        memcpy(out_affine, x.v, 32); memcpy(out_affine + 32, y.v, 32);
    };
    if (base_field_of(curve) == FIELD_FP) finish(std::integral_constant<int, FIELD_FP>{}); else finish(std::integral_constant<int, FIELD_FQ>{});
    return MINA_OK;
}

// Batched form: `batch` proofs' public-input commitments in one pipeline.  The first npub Lagrange points get a window
// table (c = 8, W = 32: 2^(8w) * L_i) once; every commitment is then one fixed-base problem of the multi-problem MSM
// (128 signed buckets per proof, no doubling chain), finished by h - A on the device.
static constexpr uint32_t LAG_C = 8, LAG_W = 32;
template <int F> static int build_lagrange_table(mina_ctx *c, SrsState &s, uint32_t n_tab) {
    const FieldK &fk = c->fk[F];
    int rc;
    if ((rc = s.lagrange_table.ensure((size_t)LAG_W * n_tab * sizeof(affine_t)))) return rc;
    if ((rc = c->L->tmp_a.ensure((size_t)n_tab * 64))) return rc;
    if ((rc = h2d(c, c->L->tmp_a, s.lagrange_host.data(), (size_t)n_tab * 64))) return rc;
    points_to_mont_kernel<F><<<cdiv(n_tab, 256), 256, 0, c->L->stream>>>(n_tab, c->L->tmp_a.as<uint32_t>(), fk.r2, s.lagrange_table.as<affine_t>());
    msm_build_table_kernel<F><<<cdiv(n_tab, 256), 256, 0, c->L->stream>>>(n_tab, n_tab, LAG_C, LAG_W, fk.one, fk.pm2, s.lagrange_table.as<affine_t>());
    HIPC(hipGetLastError());
    // the first <= 64 points also get their 128 digit multiples per window (direct commitments, lagrange.cuh)
    const uint32_t n_dig = std::min<uint32_t>(n_tab, mb::LAGD_MAX_POINTS);
    s.lagrange_digits_n = 0;
    if ((rc = s.lagrange_digits.ensure((size_t)n_dig * mb::LAGD_WINDOWS * mb::LAGD_DIGITS * sizeof(affine_t)))) return rc;
    mb::lagrange_digit_table_kernel<F><<<cdiv(n_dig * mb::LAGD_WINDOWS, 64), 64, 0, c->L->stream>>>(n_dig, n_tab, fk, s.lagrange_table.as<affine_t>(), s.lagrange_digits.as<affine_t>());
    HIPC(hipGetLastError());
    {   // ... and the digit table its 2^261-domain twin (the direct commitments add on 29-bit limbs)
        const size_t npts = (size_t)n_dig * mb::LAGD_WINDOWS * mb::LAGD_DIGITS;
        if ((rc = s.lagrange_digits29.ensure(npts * sizeof(affine_t)))) return rc;
        msm_table29_kernel<F><<<cdiv(npts, 256), 256, 0, c->L->stream>>>(npts, s.lagrange_digits.as<affine_t>(), fk.m32, s.lagrange_digits29.as<affine_t>());
        HIPC(hipGetLastError());
    }
    HIPC(hipStreamSynchronize(c->L->stream));
    s.lagrange_digits_n = n_dig;
    return MINA_OK;
}

// Lagrange basis of the domain (host cache) + the window table of its first npub points; synchronises when it has to build
int mb_ensure_lagrange_table(mina_ctx *c, int curve, uint32_t log2_domain, uint32_t npub) {
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded");
    if (log2_domain > 20 || ((uint64_t)1 << log2_domain) > s.depth || npub > ((size_t)1 << log2_domain)) return fail(MINA_ERR_ARG, "bad domain / npub");
    int rc;
    if ((rc = ensure_lagrange_host(c, curve, log2_domain))) return rc;
    const int FB = base_field_of(curve);
    if (s.lagrange_table_log2 != (int)log2_domain || s.lagrange_table_n < npub) {
        uint32_t n_tab = (uint32_t)(npub < 64 ? 64 : npub);
        if (n_tab > (1u << log2_domain)) n_tab = 1u << log2_domain;
        s.lagrange_table_log2 = -1;
        DISPATCH_FIELD(FB, { rc = build_lagrange_table<F_>(c, s, n_tab); });
        if (rc) return rc;
        s.lagrange_table_n = n_tab; s.lagrange_table_log2 = (int)log2_domain;
    }
    return MINA_OK;
}

// sum_i scalars[b][i] * L_i for `batch` proofs (XYZZ, one per proof) on the current lane; the table must be current (mb_ensure_lagrange_table).
// <= 64 inputs: straight from the digit table (lagrange.cuh), else one bucket problem per proof of the multi-problem MSM.
int mb_lagrange_sums_dev(mina_ctx *c, int curve, uint32_t npub, size_t batch, const uint32_t *d_scalars, void *d_out_xyzz) {
    SrsState &s = c->srs[curve];
    const bool generic = mb_tune().pubcomm_direct == 0;                             // cross-check switch: always the bucket MSM
    if (generic || npub > s.lagrange_digits_n)
        return mb_msm_table(c, curve, s.lagrange_table.p, s.lagrange_table_n, LAG_C, LAG_W, 0, npub, (uint32_t)batch, d_scalars, nullptr, d_out_xyzz);
    ProfScope ps_(c, PS_ACCUMULATE);
    const int FB = base_field_of(curve);
    hipStream_t st = c->L->stream;
    if (mb_tune().msm_fp29 && s.lagrange_digits29.p) {
        if (batch * (size_t)c->nlanes <= 1024) {
            DISPATCH_FIELD(FB, { mb::pubcomm_direct29_kernel<F_, 64><<<(uint32_t)batch, 64, 0, st>>>((uint32_t)batch, npub, c->fk[F_], s.lagrange_digits.as<affine_t>(), s.lagrange_digits29.as<affine_t>(), d_scalars, (xyzz_t *)d_out_xyzz); });
        } else {
            DISPATCH_FIELD(FB, { mb::pubcomm_direct29_kernel<F_, 8><<<cdiv(batch * 8, 64), 64, 0, st>>>((uint32_t)batch, npub, c->fk[F_], s.lagrange_digits.as<affine_t>(), s.lagrange_digits29.as<affine_t>(), d_scalars, (xyzz_t *)d_out_xyzz); });
        }
    } else if (batch * (size_t)c->nlanes <= 1024) {
        DISPATCH_FIELD(FB, { mb::pubcomm_direct_kernel<F_, 64><<<(uint32_t)batch, 64, 0, st>>>((uint32_t)batch, npub, c->fk[F_], s.lagrange_digits.as<affine_t>(), d_scalars, (xyzz_t *)d_out_xyzz); });
    } else {
        DISPATCH_FIELD(FB, { mb::pubcomm_direct_kernel<F_, 8><<<cdiv(batch * 8, 64), 64, 0, st>>>((uint32_t)batch, npub, c->fk[F_], s.lagrange_digits.as<affine_t>(), d_scalars, (xyzz_t *)d_out_xyzz); });
    }
    HIPC(hipGetLastError());
    return MINA_OK;
}

extern "C" int mina_public_input_commitment_batch(mina_ctx *c, int curve, uint32_t log2_domain, size_t npub, size_t batch,
                                                  const uint8_t *public_inputs, uint8_t *out_affine) {
    if (!c || (batch && !out_affine) || (npub && batch && !public_inputs)) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded");
    if (log2_domain > 20 || ((uint64_t)1 << log2_domain) > s.depth || npub > ((size_t)1 << log2_domain)) return fail(MINA_ERR_ARG, "bad domain / npub");
    if (npub > 4096 || batch > 65536) return fail(MINA_ERR_ARG, "npub / batch out of range");
    if (npub && batch && !scalars_below_2_255(public_inputs, npub * batch)) return fail(MINA_ERR_ARG, "scalar >= 2^255");
    if (batch == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if (npub == 0) {                                            // empty public input: h itself
        uint8_t hb[64];
        if ((rc = mina_srs_get_h(c, curve, hb))) return rc;
        for (size_t m = 0; m < batch; ++m) memcpy(out_affine + m * 64, hb, 64);
        return MINA_OK;
    }
    if ((rc = mb_ensure_lagrange_table(c, curve, log2_domain, (uint32_t)npub))) return rc;
    const int FB = base_field_of(curve);
    MsmWorkspace &w = c->L->ws;
    if ((rc = w.scalars.ensure(batch * npub * 32))) return rc;
    if ((rc = c->L->tmp_b.ensure(batch * sizeof(xyzz_t)))) return rc;
    if ((rc = w.out_words.ensure(batch * 17 * 4))) return rc;
    if ((rc = h2d(c, w.scalars, public_inputs, batch * npub * 32))) return rc;
    if ((rc = mb_lagrange_sums_dev(c, curve, (uint32_t)npub, batch, w.scalars.as<uint32_t>(), c->L->tmp_b.p))) return rc;
    DISPATCH_FIELD(FB, { pubcomm_finish_kernel<F_><<<cdiv(batch, 64), 64, 0, c->L->stream>>>((uint32_t)batch, c->fk[F_], s.h.as<affine_t>(), c->L->tmp_b.as<xyzz_t>(), w.out_words.as<uint32_t>()); });
    HIPC(hipGetLastError());
    std::vector<uint32_t> hw(batch * 17);
    if ((rc = d2h_sync(c, hw.data(), w.out_words, hw.size() * 4))) return rc;
    for (size_t m = 0; m < batch; ++m) {
        if (hw[m * 17 + 16]) memset(out_affine + m * 64, 0, 64); else memcpy(out_affine + m * 64, &hw[m * 17], 64);
    }
    return MINA_OK;
}
