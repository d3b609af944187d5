// api_core.hip -- context lifetime, field constants, field self-test hooks, K4 group map entry point.  (The host-only infrastructure -- error text, tuning, CSPRNG,
// worker pool -- lives in host_core.hip.)
#include "ctx.h"
#include "sponge.cuh"
#include "msm.cuh"

#include <sys/random.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>

// ------------------------------------------------------------------------------------------------
// Host-side derivation of the per-field constants (no table of magic numbers: everything follows
// from the modulus and the generator 5).
template <int F> static fe_t host_modulus() { fe_t p; for (int i = 0; i < 8; ++i) p.v[i] = modulus_limb<F>(i); return p; }
static void fe_shr1(fe_t &a) { for (int i = 0; i < 7; ++i) a.v[i] = (a.v[i] >> 1) | (a.v[i + 1] << 31); a.v[7] >>= 1; }
static fe_t fe_sub_small(const fe_t &a, uint32_t k) { fe_t r = a; uint64_t br = k; for (int i = 0; i < 8 && br; ++i) { uint64_t t = (uint64_t)r.v[i] - br; r.v[i] = (uint32_t)t; br = (t >> 32) & 1u; } return r; }
static fe_t fe_div3(const fe_t &a) { fe_t r; uint64_t rem = 0; for (int i = 7; i >= 0; --i) { uint64_t cur = (rem << 32) | a.v[i]; r.v[i] = (uint32_t)(cur / 3); rem = cur % 3; } return r; }

template <int F> static FieldK make_field_consts() {
    FieldK k;
    fe_t p = host_modulus<F>();
    // R mod p and R^2 mod p by modular doubling of 1 (fe_add works on any residues)
    fe_t a = fe_zero(); a.v[0] = 1;
    for (int i = 0; i < 512; ++i) { a = fe_add<F>(a, a); if (i == 255) k.one = a; }
    k.r2 = a;
    fe_t pm1 = fe_sub_small(p, 1);
    k.pm2 = fe_sub_small(p, 2);
    k.pm1d2 = pm1; fe_shr1(k.pm1d2);
    k.half = k.pm1d2;
    fe_t t = pm1; for (int i = 0; i < 32; ++i) fe_shr1(t);
    k.tm1d2 = fe_sub_small(t, 1); fe_shr1(k.tm1d2);
    fe_t five = fe_zero(); five.v[0] = 5; k.five = fe_to_mont<F>(five, k.r2);
    k.root = fe_pow<F>(k.five, t, k.one);
    fe_t three = fe_zero(); three.v[0] = 3; three = fe_to_mont<F>(three, k.r2);
    fe_t six = fe_zero(); six.v[0] = 6; k.bw_fu = fe_to_mont<F>(six, k.r2);
    fe_t two = fe_dbl<F>(k.one);
    fe_sqrt<F>(k.bw_s, fe_neg<F>(three), k);
    k.bw_c = fe_mul<F>(fe_sub<F>(k.bw_s, k.one), fe_inv<F>(two, k));
    k.bw_inv3 = fe_inv<F>(three, k);
    fe_t w = fe_pow<F>(k.five, fe_div3(pm1), k.one);
    k.endo = fe_sqr<F>(w);
    k.inv2 = fe_inv<F>(two, k);
    k.two255 = k.one; for (int i = 0; i < 255; ++i) k.two255 = fe_dbl<F>(k.two255);
    { fe_t t = fe_zero(); t.v[0] = 32; k.m32 = fe_to_mont<F>(t, k.r2); }
    return k;
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------

extern "C" int mina_ctx_create(int device_id, mina_ctx **out) {
    if (!out) return fail(MINA_ERR_ARG, "out is null");
    *out = nullptr;
    int ndev = 0;
    HIPC(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(MINA_ERR_HIP, "no HIP device: libminaverify has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(MINA_ERR_ARG, "bad device id");
    HIPC(hipSetDevice(device_id));
    mina_ctx *c = new mina_ctx();
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->lanes[0].stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(MINA_ERR_HIP, "hipStreamCreate failed"); }
    c->use_lane0();
    c->fk[FIELD_FP] = make_field_consts<FIELD_FP>();
    c->fk[FIELD_FQ] = make_field_consts<FIELD_FQ>();
    *out = c;
    return MINA_OK;
}

int mb_ctx_create_view(mina_ctx *parent, mina_ctx **out) {
    if (!parent || !out) return fail(MINA_ERR_ARG, "null argument");
    int rc = mina_ctx_create(parent->device, out);
    if (rc) return rc;
    (*out)->is_view = true;
    mb_ctx_refresh_view(*out, parent);
    return MINA_OK;
}
void mb_ctx_refresh_view(mina_ctx *v, mina_ctx *p) {
    for (int i = 0; i < 2; ++i) {
        v->fk[i] = p->fk[i];
        SrsState &a = v->srs[i]; const SrsState &b = p->srs[i];
        a.depth = b.depth; a.c = b.c; a.W = b.W;
        a.table.alias(b.table); a.table29.alias(b.table29); a.table29s.alias(b.table29s); a.h.alias(b.h);
        // the host-side Lagrange basis is derived from the SRS: the view keeps its copy (or one it computed itself) until the parent's SRS or domain changes
        if (a.srs_gen != b.srs_gen || a.lagrange_log2 != b.lagrange_log2 || (a.lagrange_host.empty() && !b.lagrange_host.empty())) { a.lagrange_host = b.lagrange_host; a.lagrange_log2 = b.lagrange_log2; a.srs_gen = b.srs_gen; }
        a.lagrange_table.alias(b.lagrange_table); a.lagrange_table_n = b.lagrange_table_n; a.lagrange_table_log2 = b.lagrange_table_log2;
        a.lagrange_digits.alias(b.lagrange_digits); a.lagrange_digits29.alias(b.lagrange_digits29); a.lagrange_digits_n = b.lagrange_digits_n;
        v->pparams[i].alias(p->pparams[i]); v->have_pparams[i] = p->have_pparams[i]; v->pparams_surrogate[i] = p->pparams_surrogate[i];
        v->merkle_salts[i].alias(p->merkle_salts[i]); v->merkle_depth[i] = p->merkle_depth[i];
    }
    v->kimchi_index.alias(p->kimchi_index); v->kimchi_tokens.alias(p->kimchi_tokens); v->kimchi_literals.alias(p->kimchi_literals);
    v->have_kimchi = p->have_kimchi; v->kimchi_log2 = p->kimchi_log2; memcpy(v->kimchi_digest, p->kimchi_digest, sizeof v->kimchi_digest); memcpy(v->kimchi_comms_host, p->kimchi_comms_host, sizeof v->kimchi_comms_host);
    v->pickles_index.alias(p->pickles_index); v->pickles_tokens.alias(p->pickles_tokens); v->pickles_literals.alias(p->pickles_literals);
    v->have_pickles_dev = p->have_pickles_dev; v->pickles_ms_valid = p->pickles_ms_valid;
    // borrowed: the parent frees it.  A host half the view had built for itself (step_of(view) before the parent had one) is freed first (it leaked)
    if (v->step_host && v->step_host_free && v->step_host != p->step_host) v->step_host_free(v->step_host);
    v->step_host = p->step_host; v->step_host_free = nullptr;
    v->state_salts.alias(p->state_salts); v->have_state_salts = p->have_state_salts;
}

extern "C" void mina_ctx_destroy(mina_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (int i = 0; i < MB_MAX_LANES; ++i) if (c->lanes[i].stream) (void)hipStreamSynchronize(c->lanes[i].stream);
    for (int i = 0; i < 2; ++i) { c->srs[i].table.release(); c->srs[i].table29.release(); c->srs[i].table29s.release(); c->srs[i].h.release(); c->srs[i].lagrange_table.release(); c->srs[i].lagrange_digits.release(); c->srs[i].lagrange_digits29.release(); c->pparams[i].release(); c->merkle_salts[i].release(); }
    c->state_salts.release(); c->kimchi_index.release(); c->kimchi_tokens.release(); c->kimchi_literals.release();
    c->pickles_index.release(); c->pickles_tokens.release(); c->pickles_literals.release();
    if (c->step_host && c->step_host_free) c->step_host_free(c->step_host);
    c->step_host = nullptr;
    for (int i = 0; i < MB_MAX_LANES; ++i) c->lanes[i].release_all();
    for (auto &r : c->prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (int i = 0; i < MB_MAX_LANES; ++i) {
        Lane &L = c->lanes[i];
        if (L.aux) { (void)hipStreamSynchronize(L.aux); (void)hipStreamDestroy(L.aux); (void)hipEventDestroy(L.ev_fork); (void)hipEventDestroy(L.ev_join); }
        if (L.ev_leg) (void)hipEventDestroy(L.ev_leg);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    delete c;
}

// The lanes of a context (and the concurrent culprit search of mina_state_job_batch) are separate streams; they only run side by side
// on separate hardware queues, and the runtime's default is 4.  The HIP runtime reads the variable when it initialises, so this takes
// effect when the library is loaded before the process's first HIP call (the operator's cgo binding); a value set by the user is kept.
// 16: what the reference-shaped boundary wants -- a lone job's three legs + the other slots; measured with the system runtime (ROCm 7.2), 8192
// proofs per mina_verify_state_batch call: 4 - 16 queues 54.6 - 56 ms, 24 queues 68.4 ms.  A process that pipelines device-resident jobs
// itself (bench.py's headline: 4 lanes x 4 streams since round 6; the test-suite) sets 24 -- one queue per stream plus a few; a process should never hold more streams than that.
__attribute__((constructor)) static void mb_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

extern "C" int mina_ctx_synchronize(mina_ctx *c) {
    if (!c) return fail(MINA_ERR_ARG, "null ctx");
    HIPC(hipSetDevice(c->device));
    for (int i = 0; i < c->nlanes; ++i) HIPC(hipStreamSynchronize(c->lanes[i].stream));
    return MINA_OK;
}
extern "C" void *mina_ctx_stream(mina_ctx *c) { return c ? (void *)c->lanes[c->pinned >= 0 ? c->pinned : 0].stream : nullptr; }
// Pin the `_dev` entry points to ONE pipeline lane (lane < 0: back to round-robin).  While pinned, everything they queue is ordered on mina_ctx_stream(): a caller
// that queues its own kernels, copies and collectives on that stream (torch: ExternalStream) is ordered against the library by the stream itself -- no
// hipDeviceSynchronize between the steps of a multi-GPU exchange (mina_bridge_amd/sharded.py ShardedStateJob).
extern "C" int mina_ctx_pin_lane(mina_ctx *c, int lane) {
    if (!c) return fail(MINA_ERR_ARG, "null ctx");
    if (lane >= c->nlanes) return fail(MINA_ERR_ARG, "lane outside the pipeline (mina_ctx_set_pipeline)");
    c->pinned = lane < 0 ? -1 : lane;
    return MINA_OK;
}

extern "C" int mina_ctx_set_pipeline(mina_ctx *c, int lanes) {
    if (!c) return fail(MINA_ERR_ARG, "null ctx");
    if (lanes < 1 || lanes > MB_PIPE_LANES) return fail(MINA_ERR_ARG, "lanes must be in 1..32");
    HIPC(hipSetDevice(c->device));
    for (int i = 0; i < c->nlanes; ++i) HIPC(hipStreamSynchronize(c->lanes[i].stream));
    for (int i = 0; i < lanes; ++i)
        if (!c->lanes[i].stream) HIPC(hipStreamCreateWithFlags(&c->lanes[i].stream, hipStreamNonBlocking));
    c->nlanes = lanes; c->rr = 0; c->pinned = -1; c->use_lane0();
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int mina_to_group(mina_ctx *c, int curve, size_t n, const uint8_t *t, uint8_t *out) {
    if (!c || (n && (!t || !out))) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (n == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = h2d(c, c->L->tmp_a, t, n * 32))) return rc;
    if ((rc = c->L->tmp_b.ensure(n * 64))) return rc;
    const int F = base_field_of(curve);
    DISPATCH_FIELD(F, { to_group_kernel<F_><<<cdiv(n, 128), 128, 0, c->L->stream>>>((uint32_t)n, c->fk[F_], c->L->tmp_a.as<uint32_t>(), c->L->tmp_b.as<uint32_t>()); });
    return d2h_sync(c, out, c->L->tmp_b, n * 64);
}

extern "C" int mina_field_mul(mina_ctx *c, int field, size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out) {
    if (!c || (n && (!a || !b || !out))) return fail(MINA_ERR_ARG, "null argument");
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    if (n == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = h2d(c, c->L->tmp_a, a, n * 32))) return rc;
    if ((rc = h2d(c, c->L->tmp_b, b, n * 32))) return rc;
    if ((rc = c->L->tmp_c.ensure(n * 32))) return rc;
    DISPATCH_FIELD(field, { field_mul_kernel<F_><<<cdiv(n, 256), 256, 0, c->L->stream>>>((uint32_t)n, c->fk[F_], c->L->tmp_a.as<uint32_t>(), c->L->tmp_b.as<uint32_t>(), c->L->tmp_c.as<uint32_t>()); });
    return d2h_sync(c, out, c->L->tmp_c, n * 32);
}
extern "C" int mina_field_inv(mina_ctx *c, int field, size_t n, const uint8_t *a, uint8_t *out) {
    if (!c || (n && (!a || !out))) return fail(MINA_ERR_ARG, "null argument");
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    if (n == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = h2d(c, c->L->tmp_a, a, n * 32))) return rc;
    if ((rc = c->L->tmp_c.ensure(n * 32))) return rc;
    DISPATCH_FIELD(field, { field_inv_kernel<F_><<<cdiv(n, 256), 256, 0, c->L->stream>>>((uint32_t)n, c->fk[F_], c->L->tmp_a.as<uint32_t>(), c->L->tmp_c.as<uint32_t>()); });
    return d2h_sync(c, out, c->L->tmp_c, n * 32);
}
extern "C" int mina_field_sqrt(mina_ctx *c, int field, size_t n, const uint8_t *a, uint8_t *out, uint8_t *ok) {
    if (!c || (n && (!a || !out || !ok))) return fail(MINA_ERR_ARG, "null argument");
    if (bad_field(field)) return fail(MINA_ERR_ARG, "bad field");
    if (n == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = h2d(c, c->L->tmp_a, a, n * 32))) return rc;
    if ((rc = c->L->tmp_c.ensure(n * 32))) return rc;
    if ((rc = c->L->tmp_d.ensure(n))) return rc;
    DISPATCH_FIELD(field, { field_sqrt_kernel<F_><<<cdiv(n, 256), 256, 0, c->L->stream>>>((uint32_t)n, c->fk[F_], c->L->tmp_a.as<uint32_t>(), c->L->tmp_c.as<uint32_t>(), c->L->tmp_d.as<uint8_t>()); });
    HIPC(hipMemcpyAsync(ok, c->L->tmp_d.p, n, hipMemcpyDeviceToHost, c->L->stream));
    return d2h_sync(c, out, c->L->tmp_c, n * 32);
}


// ------------------------------------------------------------------------------------------------
// stage timing
static const char *PROF_NAMES[PS_COUNT] = {"msm_digits", "msm_scan", "msm_scatter", "msm_accumulate", "msm_bucket_sum", "msm_segsum",
                                           "msm_reduce2d", "msm_finish", "bpoly_tables", "bpoly_fold", "bpoly_finish", "pstate_hash", "ipa_transcript", "kimchi_to_batch", "pickles_statement"};
void mb_prof_begin(mina_ctx *c, int stage) {
    ProfState &p = c->prof;
    if (p.used == p.recs.size()) {
        ProfState::Rec r; r.stage = stage;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        p.recs.push_back(r);
    }
    p.recs[p.used].stage = stage;
    (void)hipEventRecord(p.recs[p.used].a, c->L->stream);
}
void mb_prof_end(mina_ctx *c, int stage) {
    ProfState &p = c->prof;
    if (p.used >= p.recs.size() || p.recs[p.used].stage != stage) return;
    (void)hipEventRecord(p.recs[p.used].b, c->L->stream);
    ++p.used;
}
extern "C" int mina_prof_enable(mina_ctx *c, int stage_mask) {
    if (!c) return fail(MINA_ERR_ARG, "null ctx");
    c->prof.mask = stage_mask; c->prof.used = 0;
    return MINA_OK;
}
// Synchronises the stream, writes {"stage": [launches, total_ms], ...} for everything recorded since the last
// read / enable, and resets the recording.
extern "C" int mina_prof_read(mina_ctx *c, char *buf, size_t cap) {
    if (!c || !buf || cap < 8) return fail(MINA_ERR_ARG, "bad argument");
    HIPC(hipSetDevice(c->device));
    for (int i = 0; i < c->nlanes; ++i) HIPC(hipStreamSynchronize(c->lanes[i].stream));
    double tot[PS_COUNT] = {0}; int cnt[PS_COUNT] = {0};
    for (size_t i = 0; i < c->prof.used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->prof.recs[i].a, c->prof.recs[i].b) == hipSuccess) { tot[c->prof.recs[i].stage] += ms; cnt[c->prof.recs[i].stage]++; }
    }
    c->prof.used = 0;
    std::string o = "{";
    bool first = true;
    for (int s = 0; s < PS_COUNT; ++s) {
        if (!cnt[s]) continue;
        char tmp[128]; snprintf(tmp, sizeof tmp, "%s\"%s\": [%d, %.6f]", first ? "" : ", ", PROF_NAMES[s], cnt[s], tot[s]);
        o += tmp; first = false;
    }
    o += "}";
    if (o.size() + 1 > cap) return fail(MINA_ERR_ARG, "buffer too small");
    memcpy(buf, o.c_str(), o.size() + 1);
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------
// self-test hook for the lane-cooperative group law: for each i runs the same chain of XYZZ operations
//   acc = P_i + Q_i; acc += acc (doubling branch); acc += Q_i; acc = 2*acc; acc += (-acc_before) ...
// once with the single-lane routines and once with the 4-lane cooperative ones; outputs both results (affine,
// canonical) and whether the XYZZ coordinates agreed word for word at every step.
template <int F>
__global__ void __launch_bounds__(256)
group_law_selftest_kernel(uint32_t n, FieldK fk, const uint32_t *__restrict__ pw, const uint32_t *__restrict__ qw,
                          uint32_t *__restrict__ out_serial, uint32_t *__restrict__ out_quad, uint32_t *__restrict__ same) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, i = gid >> 2, rho = gid & 3;
    const bool live = i < n; const uint32_t ii = live ? i : 0;
    auto ld = [&](const uint32_t *w) { affine_t a; for (int k = 0; k < 8; ++k) { a.x.v[k] = w[(size_t)ii * 16 + k]; a.y.v[k] = w[(size_t)ii * 16 + 8 + k]; }
                                       a.x = fe_to_mont<F>(a.x, fk.r2); a.y = fe_to_mont<F>(a.y, fk.r2); return a; };
    const affine_t P = ld(pw), Q = ld(qw);
    const xyzz_t xp = xyzz_from_affine<F>(P, fk.one), xq = xyzz_from_affine<F>(Q, fk.one);
    auto eq = [](const xyzz_t &a, const xyzz_t &b) { return fe_eq(a.x, b.x) && fe_eq(a.y, b.y) && fe_eq(a.zz, b.zz) && fe_eq(a.zzz, b.zzz); };
    xyzz_t s = xp, c = xp; bool ok = true;
    xyzz_add<F>(s, xq);            xyzz_add_quad<F>(c, xq);            ok &= eq(s, c);
    { xyzz_t t = s; xyzz_add<F>(s, t); t = c; xyzz_add_quad<F>(c, t); } ok &= eq(s, c);      // P == Q branch
    xyzz_add<F>(s, xq);            xyzz_add_quad<F>(c, xq);            ok &= eq(s, c);
    s = xyzz_dbl<F>(s);            c = xyzz_dbl_quad<F>(c);            ok &= eq(s, c);
    { xyzz_t t = s; xyzz_add<F>(t, xp); xyzz_t u = c; xyzz_add_quad<F>(u, xp);                // general add with non-trivial ZZ on both sides
      xyzz_add<F>(s, t); xyzz_add_quad<F>(c, u); } ok &= eq(s, c);
    { xyzz_t t = s; t.y = fe_neg<F>(t.y); xyzz_t z = s; xyzz_add<F>(z, t); xyzz_t u = c; u.y = fe_neg<F>(u.y); xyzz_t z2 = c; xyzz_add_quad<F>(z2, u);
      ok &= xyzz_is_inf(z) && xyzz_is_inf(z2); }                                              // P == -Q branch
    auto st = [&](const xyzz_t &t, uint32_t *o) {
        if (xyzz_is_inf(t)) { for (int k = 0; k < 16; ++k) o[(size_t)i * 16 + k] = 0; return; }
        fe_t zi = fe_inv<F>(fe_mul<F>(t.zz, t.zzz), fk);
        fe_t x = fe_from_mont<F>(fe_mul<F>(t.x, fe_mul<F>(zi, t.zzz))), y = fe_from_mont<F>(fe_mul<F>(t.y, fe_mul<F>(zi, t.zz)));
        for (int k = 0; k < 8; ++k) { o[(size_t)i * 16 + k] = x.v[k]; o[(size_t)i * 16 + 8 + k] = y.v[k]; } };
    if (live && rho == 0) { st(s, out_serial); st(c, out_quad); same[i] = ok ? 1u : 0u; }
}

extern "C" int mina_selftest_group_law(mina_ctx *c, int curve, size_t n, const uint8_t *p, const uint8_t *q, uint8_t *out_serial,
                                       uint8_t *out_quad, uint8_t *same) {
    if (!c || (n && (!p || !q || !out_serial || !out_quad || !same))) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (n == 0) return MINA_OK;
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = h2d(c, c->L->tmp_a, p, n * 64))) return rc;
    if ((rc = h2d(c, c->L->tmp_b, q, n * 64))) return rc;
    if ((rc = c->L->tmp_c.ensure(n * 64))) return rc;
    if ((rc = c->L->tmp_d.ensure(n * 64 + n * 4))) return rc;
    uint32_t *d_quad = c->L->tmp_d.as<uint32_t>(), *d_same = d_quad + n * 16;
    DISPATCH_FIELD(base_field_of(curve), {
        group_law_selftest_kernel<F_><<<cdiv(n * 4, 256), 256, 0, c->L->stream>>>((uint32_t)n, c->fk[F_], c->L->tmp_a.as<uint32_t>(), c->L->tmp_b.as<uint32_t>(),
                                                                              c->L->tmp_c.as<uint32_t>(), d_quad, d_same);
    });
    std::vector<uint32_t> sm(n);
    HIPC(hipMemcpyAsync(out_quad, d_quad, n * 64, hipMemcpyDeviceToHost, c->L->stream));
    HIPC(hipMemcpyAsync(sm.data(), d_same, n * 4, hipMemcpyDeviceToHost, c->L->stream));
    if ((rc = d2h_sync(c, out_serial, c->L->tmp_c, n * 64))) return rc;
    for (size_t i = 0; i < n; ++i) same[i] = sm[i] ? 1 : 0;
    return MINA_OK;
}

// ------------------------------------------------------------------------------------------------
// plain device-memory helpers for callers of the `_dev` entry points that have no HIP binding of their own
extern "C" int mina_dev_malloc(mina_ctx *c, size_t bytes, void **out) {
    if (!c || !out) return fail(MINA_ERR_ARG, "null argument");
    *out = nullptr;
    HIPC(hipSetDevice(c->device));
    if (hipMalloc(out, bytes ? bytes : 4) != hipSuccess) return fail(MINA_ERR_HIP, "hipMalloc failed");
    return MINA_OK;
}
extern "C" int mina_dev_free(mina_ctx *c, void *p) {
    if (!c) return fail(MINA_ERR_ARG, "null argument");
    HIPC(hipSetDevice(c->device));
    if (p) HIPC(hipFree(p));
    return MINA_OK;
}
// synchronous copies (they wait for the context's lanes first, so results of queued `_dev` calls are visible)
extern "C" int mina_dev_upload(mina_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return fail(MINA_ERR_ARG, "null argument");
    HIPC(hipSetDevice(c->device));
    if (bytes) HIPC(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return MINA_OK;
}
extern "C" int mina_dev_download(mina_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return fail(MINA_ERR_ARG, "null argument");
    int rc = mina_ctx_synchronize(c);
    if (rc) return rc;
    if (bytes) HIPC(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return MINA_OK;
}
