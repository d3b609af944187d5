// api_msm.hip -- K1 host driver: queues the MSM kernel pipeline of msm.cuh on the context stream.
#include "ctx.h"
#include "msm.cuh"
#include <vector>
#include <cstdlib>

// ------------------------------------------------------------------------------------------------
// MSM driver
static MsmShape fixed_shape(const SrsState &s, uint32_t first, uint32_t n) {
    MsmShape sh; sh.n = n; sh.c = s.c; sh.W = s.W; sh.NB = 1u << (s.c - 1); sh.nsets = 1; sh.table_stride = s.depth; sh.base_first = first; sh.nprob = 1; return sh;
}
static MsmShape variable_shape(uint32_t n) {
    MsmShape sh; sh.n = n;
    // window widths whose TOP window is not degenerate for 255-bit scalars: c = 8 (top digit < 64), 13 (W = 20, top digit
    // < 128), 15 (W = 18, the 18th window is always empty); c = 11 / 14 put a quarter of a window's entries in one bucket
    sh.c = n < 2048 ? 8 : (n < (1u << 17) ? 13 : 15);
    sh.W = (256 + sh.c - 1) / sh.c; sh.NB = 1u << (sh.c - 1); sh.nsets = sh.W; sh.table_stride = 0; sh.base_first = 0; sh.nprob = 1; return sh;
}

template <int F>
static int run_msm(mina_ctx *c, const MsmShape &sh, const uint32_t *d_scalars, const affine_t *d_points,
                   uint32_t *d_out_words /* 17 words per problem */, xyzz_t *d_out_xyzz /* one per problem */,
                   const affine_t *d_points29 = nullptr /* the table's 2^261-domain twin (SrsState::table29): the accumulate kernels then run on 29-bit limbs (ec29.cuh) */,
                   const void *d_points29s = nullptr /* ... or its pre-split form (SrsState::table29s, tab29_t records; mina_verify_tuning.msm_fp29 = 2) */) {
    if (d_points29 && !mb_tune().msm_fp29) d_points29 = nullptr;      // cross-check switch: the 8 x 32 law everywhere
    const bool split = d_points29 && d_points29s && mb_tune().msm_fp29 == 2;
    bool red29 = false;                                              // msm_fp29 = 3: the buckets of the multi-MSM form stay on 29-bit limbs and the 2-D reduction runs on them too
    MsmWorkspace &w = c->L->ws;
    const FieldK &fk = c->fk[F];
    if (sh.nprob == 0 || sh.nsets % sh.nprob) return fail(MINA_ERR_ARG, "bad problem count");
    if ((uint64_t)sh.NB * sh.nsets > (1u << 26) || (uint64_t)sh.n * sh.W * sh.nprob > (1u << 28)) return fail(MINA_ERR_ARG, "MSM batch too large for one pipeline");
    const uint32_t nb_total = sh.NB * sh.nsets;
    const size_t entries = (size_t)sh.n * sh.W * sh.nprob;
    const size_t max_tasks = entries / MSM_TASK_LEN + nb_total + 1;
    int rc;
    // sort plan: partitioned LDS sort when the partition table fits (P <= 1024 with <= 2048 buckets per partition),
    // else the atomic counting sort
    SortShape ss; ss.fbits = 8; ss.Gl = cdiv(sh.n, 256); ss.SB = sh.NB * (sh.nsets / sh.nprob);
    while (ss.fbits < 11 && cdiv(ss.SB, 1u << ss.fbits) > 256) ++ss.fbits;
    ss.Pl = cdiv(ss.SB, 1u << ss.fbits);
    const uint64_t gh_words = (uint64_t)ss.Pl * ss.Gl * sh.nprob;
    const bool part_sort = ss.Pl <= 1024 && gh_words <= (1u << 22);
    if (part_sort) {
        if ((rc = w.ghist.ensure((gh_words + 8) * 4))) return rc;
        if ((rc = w.stage.ensure(entries * 8))) return rc;
        if ((rc = w.ekey.ensure(entries * 4))) return rc;
    } else {
        if ((rc = w.ekey.ensure(entries * 4))) return rc;
        if ((rc = w.eval.ensure(entries * 4))) return rc;
        if ((rc = w.eoff.ensure(entries * 4))) return rc;
    }
    if ((rc = w.sorted.ensure(entries * 4))) return rc;
    if ((rc = w.count.ensure((size_t)nb_total * 4))) return rc;
    if ((rc = w.start.ensure(((size_t)nb_total + 1) * 4))) return rc;
    // throughput form of the accumulate (one lane per pair of count-ranked buckets, K1t) when one launch carries enough
    // buckets to fill the chip; a single MSM's 32768 buckets would leave it latency-bound (16 pipelined lanes: 6.2 k
    // proofs/s against 7.6 k with tasks)
    const bool bucket_lanes = part_sort && sh.nprob >= 4;                                 // >= 64 k lanes of ~62 adds each
    if (bucket_lanes) {
        if ((rc = w.order.ensure((size_t)nb_total * 4))) return rc;
        red29 = d_points29 && !split && mb_tune().msm_fp29 == 3;
        if (red29 && ((rc = w.buckets29.ensure((size_t)nb_total * sizeof(xyzz29_t))) || (rc = w.seg_bad.ensure(((size_t)(sh.NB / 128) + 128) * sh.nsets * 4)))) return rc;
    } else {                                                    // task numbering and the 128-B task partials
        if ((rc = w.task_start.ensure(((size_t)nb_total + 1) * 4))) return rc;
        if ((rc = w.rem_pos.ensure((size_t)nb_total * 4))) return rc;
        if ((rc = w.rem_bucket.ensure((size_t)nb_total * 4))) return rc;
        if ((rc = w.partial.ensure(max_tasks * sizeof(xyzz_t)))) return rc;
    }
    if ((rc = w.info.ensure(16))) return rc;
    if (d_points29 && (rc = w.redo.ensure((std::max<size_t>(max_tasks, nb_total) + 1) * 4))) return rc;
    if ((rc = w.heavy.ensure((max_tasks / MSM_HEAVY_TASKS + entries / MSM_HEAVY_ENTRIES + 2) * 4))) return rc;
    if ((rc = w.buckets.ensure((size_t)nb_total * sizeof(xyzz_t)))) return rc;
    if (sh.NB < 128 || sh.NB > 32768) return fail(MINA_ERR_ARG, "unsupported bucket count");
    if ((rc = w.red_r.ensure((size_t)(sh.NB / 128) * sh.nsets * sizeof(xyzz_t)))) return rc;
    if ((rc = w.red_ws.ensure((size_t)128 * sh.nsets * sizeof(xyzz_t)))) return rc;
    if ((rc = w.red2_r.ensure((size_t)24 * sh.nsets * sizeof(xyzz_t)))) return rc;
    if ((rc = w.red2_w.ensure((size_t)24 * sh.nsets * sizeof(xyzz_t)))) return rc;
    if ((rc = w.set_total.ensure((size_t)sh.nsets * sizeof(xyzz_t)))) return rc;

    hipStream_t st = c->L->stream;
    // one bucket set per problem and no affine output wanted: the reduction kernel's result IS the answer (no finish launch)
    const bool fused_finish = sh.nsets == sh.nprob && !d_out_words && d_out_xyzz;
    if (part_sort) {
        { ProfScope ps_(c, PS_DIGITS);
          msm_part_kernel<false><<<ss.Gl * sh.nprob, 1024, 0, st>>>(sh, ss, d_scalars, w.ghist.as<uint32_t>(), nullptr, w.ekey.as<uint32_t>());
          msm_excl_scan_kernel<<<1, 1024, 0, st>>>((uint32_t)gh_words, w.ghist.as<uint32_t>()); }
        { ProfScope ps_(c, PS_SCATTER);
          msm_part_kernel<true><<<ss.Gl * sh.nprob, 1024, 0, st>>>(sh, ss, d_scalars, w.ghist.as<uint32_t>(), w.stage.as<uint2>(), w.ekey.as<uint32_t>());
          msm_part_sort_kernel<<<ss.Pl * sh.nprob, 1024, 0, st>>>(ss, w.ghist.as<uint32_t>(), w.stage.as<uint2>(), w.count.as<uint32_t>(), w.sorted.as<uint32_t>(), w.info.as<uint32_t>()); }
        if (bucket_lanes) {
            ProfScope ps_(c, PS_SCAN);
            msm_order_kernel<<<sh.nprob, 1024, 0, st>>>(ss, sh.nprob, w.ghist.as<uint32_t>(), w.count.as<uint32_t>(), w.start.as<uint32_t>(), w.order.as<uint32_t>(), w.info.as<uint32_t>(), w.heavy.as<uint32_t>());
        } else {
            ProfScope ps_(c, PS_SCAN); msm_scan_kernel<<<1, 1024, 0, st>>>(nb_total, w.count.as<uint32_t>(), w.start.as<uint32_t>(), w.task_start.as<uint32_t>(),
                                                                       w.rem_pos.as<uint32_t>(), w.info.as<uint32_t>());
                                     msm_rem_invert_kernel<<<cdiv(nb_total, 256), 256, 0, st>>>(nb_total, w.rem_pos.as<uint32_t>(), w.rem_bucket.as<uint32_t>());
        }
    } else {
        HIPC(hipMemsetAsync(w.count.p, 0, (size_t)nb_total * 4, st));
        { ProfScope ps_(c, PS_DIGITS); msm_digits_kernel<<<cdiv(entries, 256), 256, 0, st>>>(sh, d_scalars, w.count.as<uint32_t>(), w.ekey.as<uint32_t>(),
                                                           w.eval.as<uint32_t>(), w.eoff.as<uint32_t>()); }
        { ProfScope ps_(c, PS_SCAN); msm_scan_kernel<<<1, 1024, 0, st>>>(nb_total, w.count.as<uint32_t>(), w.start.as<uint32_t>(), w.task_start.as<uint32_t>(),
                                                                       w.rem_pos.as<uint32_t>(), w.info.as<uint32_t>());
                                     msm_rem_invert_kernel<<<cdiv(nb_total, 256), 256, 0, st>>>(nb_total, w.rem_pos.as<uint32_t>(), w.rem_bucket.as<uint32_t>()); }
        { ProfScope ps_(c, PS_SCATTER); msm_scatter_kernel<<<cdiv(entries, 256), 256, 0, st>>>(entries, w.ekey.as<uint32_t>(), w.eval.as<uint32_t>(),
                                                               w.eoff.as<uint32_t>(), w.start.as<uint32_t>(), w.sorted.as<uint32_t>()); }
    }
    if (bucket_lanes) {
        { ProfScope ps_(c, PS_ACCUMULATE);
          // the redo launch is sized by the bucket count (grid-stride over info[3]; lanes beyond it leave at once): a batch of repeated commitments may hand every bucket
          // back, and 16 blocks -- the size until round 6 -- would have summed them on 1024 lanes
          if (d_points29) {
              if (split) msm_accumulate_bucket29_kernel<F, 2><<<cdiv(nb_total / 2, 256), 256, 0, st>>>(nb_total, ss.SB, w.start.as<uint32_t>(), w.order.as<uint32_t>(), w.sorted.as<uint32_t>(), d_points29s, fk.one, fk.m32,
                                                                                                 w.buckets.as<xyzz_t>(), w.info.as<uint32_t>(), w.redo.as<uint32_t>(), nullptr);
              else msm_accumulate_bucket29_kernel<F, 1><<<cdiv(nb_total / 2, 256), 256, 0, st>>>(nb_total, ss.SB, w.start.as<uint32_t>(), w.order.as<uint32_t>(), w.sorted.as<uint32_t>(), d_points29, fk.one, fk.m32,
                                                                                           w.buckets.as<xyzz_t>(), w.info.as<uint32_t>(), w.redo.as<uint32_t>(), red29 ? w.buckets29.as<xyzz29_t>() : nullptr);
              msm_bucket_redo_kernel<F><<<std::min<uint32_t>(1024u, cdiv(nb_total, 64)), 64, 0, st>>>(w.start.as<uint32_t>(), w.info.as<uint32_t>(), w.redo.as<uint32_t>(), w.sorted.as<uint32_t>(), d_points, fk.one, w.buckets.as<xyzz_t>());
          }
          else msm_accumulate_bucket_kernel<F><<<cdiv(nb_total / 2, 256), 256, 0, st>>>(nb_total, ss.SB, w.start.as<uint32_t>(), w.order.as<uint32_t>(), w.sorted.as<uint32_t>(), d_points, fk.one, w.buckets.as<xyzz_t>()); }
        { ProfScope ps_(c, PS_BUCKET_SUM);
          msm_bucket_heavy_entries_kernel<F><<<128, 256, 0, st>>>(w.start.as<uint32_t>(), w.info.as<uint32_t>(), w.heavy.as<uint32_t>(), w.sorted.as<uint32_t>(), d_points, fk.one, w.buckets.as<xyzz_t>());
          // the buckets the two 8 x 32 kernels wrote join the others on 29-bit limbs (none on SRS points with unstructured scalars: a fixed-size launch that finds empty lists)
          if (red29) msm_buckets_to29_kernel<F><<<16, 64, 0, st>>>(w.info.as<uint32_t>(), w.heavy.as<uint32_t>(), w.redo.as<uint32_t>(), w.buckets.as<xyzz_t>(), fk.m32, w.buckets29.as<xyzz29_t>()); }
    } else {
        { ProfScope ps_(c, PS_ACCUMULATE);
          if (d_points29) {
              if (split) msm_accumulate29_kernel<F, 2><<<cdiv(max_tasks, 256), 256, 0, st>>>(nb_total, w.start.as<uint32_t>(), w.task_start.as<uint32_t>(), w.rem_bucket.as<uint32_t>(), w.info.as<uint32_t>(),
                                                                                   w.sorted.as<uint32_t>(), d_points29s, fk.one, fk.m32, w.partial.as<xyzz_t>(), w.info.as<uint32_t>(), w.redo.as<uint32_t>());
              else msm_accumulate29_kernel<F, 1><<<cdiv(max_tasks, 256), 256, 0, st>>>(nb_total, w.start.as<uint32_t>(), w.task_start.as<uint32_t>(), w.rem_bucket.as<uint32_t>(), w.info.as<uint32_t>(),
                                                                             w.sorted.as<uint32_t>(), d_points29, fk.one, fk.m32, w.partial.as<xyzz_t>(), w.info.as<uint32_t>(), w.redo.as<uint32_t>());
              // the tasks handed back (none on SRS points): the 8 x 32 kernel over the redo queue -- a full-size launch whose lanes beyond info[3] leave at once
              msm_accumulate_kernel<F><<<cdiv(max_tasks, 256), 256, 0, st>>>(nb_total, w.start.as<uint32_t>(), w.task_start.as<uint32_t>(), w.rem_bucket.as<uint32_t>(), w.info.as<uint32_t>(),
                                                                     w.sorted.as<uint32_t>(), d_points, fk.one, w.partial.as<xyzz_t>(), w.redo.as<uint32_t>());
          }
          else msm_accumulate_kernel<F><<<cdiv(max_tasks, 256), 256, 0, st>>>(nb_total, w.start.as<uint32_t>(), w.task_start.as<uint32_t>(),
                                                                       w.rem_bucket.as<uint32_t>(), w.info.as<uint32_t>(), w.sorted.as<uint32_t>(), d_points, fk.one, w.partial.as<xyzz_t>()); }
        { ProfScope ps_(c, PS_BUCKET_SUM);
          if (c->nlanes > 1)
              msm_bucket_sum_lane_kernel<F><<<cdiv(nb_total, 256), 256, 0, st>>>(nb_total, w.task_start.as<uint32_t>(), w.rem_pos.as<uint32_t>(), w.info.as<uint32_t>(), w.partial.as<xyzz_t>(),
                                                                      w.buckets.as<xyzz_t>(), w.heavy.as<uint32_t>());
          else
              msm_bucket_sum_kernel<F><<<cdiv((size_t)nb_total * 4, 256), 256, 0, st>>>(nb_total, w.task_start.as<uint32_t>(), w.rem_pos.as<uint32_t>(), w.info.as<uint32_t>(), w.partial.as<xyzz_t>(),
                                                                      w.buckets.as<xyzz_t>(), w.heavy.as<uint32_t>());
                                           msm_bucket_sum_heavy_kernel<F><<<128, 256, 0, st>>>(w.task_start.as<uint32_t>(), w.rem_pos.as<uint32_t>(), w.info.as<uint32_t>(), w.partial.as<xyzz_t>(),
                                                                      w.buckets.as<xyzz_t>(), w.heavy.as<uint32_t>()); }
    }
    {
        // 2-D bucket reduction: C = 128 columns, R = NB / C rows (NB is a power of two in [128, 32768])
        const uint32_t C = 128, log2C = 7, R = sh.NB / C;
        SegSum rows, cols;
        const bool coop = c->nlanes == 1 && !red29;            // one MSM at a time: latency form; pipelined lanes: throughput form (the 29-bit reduction has the throughput form only)
        const uint32_t chunk = coop ? SEG_CHUNK : 16, lpg = coop ? 4 : 1;
        rows.nseg = R * sh.nsets; rows.per_set = R; rows.len = C; rows.seg_stride = C; rows.elem_stride = 1; rows.lanes = C / chunk;
        cols.nseg = C * sh.nsets; cols.per_set = C; cols.len = R; cols.seg_stride = 1; cols.elem_stride = C;
        { uint32_t l = 1; while (l * chunk < R && l < 16) l <<= 1; cols.lanes = l; }   // workers per column, <= 16 (one wave even in the quad form)
        const uint32_t threads_rows = rows.nseg * rows.lanes * lpg, threads_cols = cols.nseg * cols.lanes * lpg;
        const uint32_t blocks = cdiv(threads_rows > threads_cols ? threads_rows : threads_cols, 256);
        { ProfScope ps_(c, PS_REDUCE_A);
          if (red29) {
              const uint32_t nseg = rows.nseg + cols.nseg;
              HIPC(hipMemsetAsync(w.seg_bad.p, 0, (size_t)nseg * 4, st));
              msm_segsum29_kernel<F><<<dim3(blocks, 2), 256, 0, st>>>(sh.NB, rows, cols, w.buckets29.as<xyzz29_t>(), fk.one, w.red_r.as<xyzz_t>(), w.red_ws.as<xyzz_t>(), w.seg_bad.as<uint32_t>());
              msm_segsum29_redo_kernel<F><<<cdiv(nseg, 64), 64, 0, st>>>(sh.NB, rows, cols, w.buckets29.as<xyzz29_t>(), fk.one, w.red_r.as<xyzz_t>(), w.red_ws.as<xyzz_t>(), w.seg_bad.as<uint32_t>());
          }
          else if (coop) msm_segsum_kernel<F, true><<<dim3(blocks, 2), 256, 0, st>>>(sh.NB, rows, cols, w.buckets.as<xyzz_t>(), w.red_r.as<xyzz_t>(), w.red_ws.as<xyzz_t>());
          else msm_segsum_kernel<F, false><<<dim3(blocks, 2), 256, 0, st>>>(sh.NB, rows, cols, w.buckets.as<xyzz_t>(), w.red_r.as<xyzz_t>(), w.red_ws.as<xyzz_t>()); }
        const uint32_t Gr = (R + 15) / 16, Gc = C / 16;
        { ProfScope ps_(c, PS_REDUCE_BC);
          msm_wsum16_kernel<F><<<dim3(Gr + Gc, sh.nsets), 64, 0, st>>>(R, C, Gr, w.red_r.as<xyzz_t>(), w.red_ws.as<xyzz_t>(), w.red2_r.as<xyzz_t>(), w.red2_w.as<xyzz_t>());
          msm_reduce2d_kernel<F><<<sh.nsets, 256, 0, st>>>(Gr, Gc, log2C, w.red2_r.as<xyzz_t>(), w.red2_w.as<xyzz_t>(), w.set_total.as<xyzz_t>(), fused_finish ? d_out_xyzz : nullptr); }
    }
    if (!fused_finish) { ProfScope ps_(c, PS_FINISH); msm_finish_kernel<F><<<sh.nprob, 64, 0, st>>>(sh.nsets / sh.nprob, sh.c, w.set_total.as<xyzz_t>(), fk.one, fk.pm2, d_out_xyzz, d_out_words); }
    HIPC(hipGetLastError());
    return MINA_OK;
}

static void words_to_point_bytes(const uint32_t w[17], uint8_t *out) {
    if (w[16]) { memset(out, 0, 64); return; }
    memcpy(out, w, 64);
}

int mb_msm_fixed(mina_ctx *c, int curve, uint32_t n, const uint32_t *d_scalars, uint32_t *d_out_words, void *d_out_xyzz, uint32_t first) {
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded for this curve");
    if (n == 0 || (uint64_t)first + n > s.depth) return fail(MINA_ERR_ARG, "base range must lie inside the SRS");
    MsmShape sh = fixed_shape(s, first, n);
    int rc = MINA_OK;
    DISPATCH_FIELD(base_field_of(curve), { rc = run_msm<F_>(c, sh, d_scalars, s.table.as<affine_t>(), d_out_words, (xyzz_t *)d_out_xyzz, s.table29.as<affine_t>(), s.table29s.p); });
    return rc;
}

// nprob MSMs over one fixed-base window table (table[w * stride + i] = 2^(cbits w) * base_i): results per problem
int mb_msm_table(mina_ctx *c, int curve, const void *d_table, uint32_t stride, uint32_t cbits, uint32_t W, uint32_t first, uint32_t n,
                 uint32_t nprob, const uint32_t *d_scalars, uint32_t *d_out_words, void *d_out_xyzz) {
    if (n == 0 || nprob == 0 || (uint64_t)first + n > stride) return fail(MINA_ERR_ARG, "bad table MSM shape");
    MsmShape sh; sh.n = n; sh.c = cbits; sh.W = W; sh.NB = 1u << (cbits - 1); sh.nsets = nprob; sh.table_stride = stride; sh.base_first = first; sh.nprob = nprob;
    int rc = MINA_OK;
    const SrsState &s = c->srs[curve];
    const affine_t *t29 = (d_table == s.table.p && s.table29.p) ? (const affine_t *)s.table29.p : nullptr;      // the SRS table has a 2^261-domain twin; other tables do not
    const void *t29s = t29 ? s.table29s.p : nullptr;
    DISPATCH_FIELD(base_field_of(curve), { rc = run_msm<F_>(c, sh, d_scalars, (const affine_t *)d_table, d_out_words, (xyzz_t *)d_out_xyzz, t29, t29s); });
    return rc;
}

int mb_msm_variable(mina_ctx *c, int curve, uint32_t n, const uint32_t *d_scalars, const void *d_points_mont,
                    uint32_t *d_out_words, void *d_out_xyzz) {
    if (n == 0) return fail(MINA_ERR_ARG, "n must be positive");
    MsmShape sh = variable_shape(n);
    int rc = MINA_OK;
    // Round 5: the accumulate kernels run the 29-bit group law on variable bases too.  The points get a 2^261-domain twin first (two products per point against
    // W = 20 mixed adds of ten products each: 1 %); equal / opposite points -- proofs may repeat a commitment -- are found exactly on the lazy limbs and go through
    // the 8 x 32 redo queue like everywhere else.  mina_verify_tuning.msm_fp29 = 0: the 8 x 32 law (cross-check).
    const affine_t *twin = nullptr;
    // (below 32 points -- the part MSMs of a culprit search, single openings -- the twin's launch costs more than the 29-bit law saves: the 8 x 32 law, no redo queue)
    if (mb_tune().msm_fp29 && n >= 32) {
        MsmWorkspace &w = c->L->ws;
        if ((rc = w.points29.ensure((size_t)n * sizeof(affine_t)))) return rc;
        DISPATCH_FIELD(base_field_of(curve), { msm_table29_kernel<F_><<<cdiv(n, 256), 256, 0, c->L->stream>>>(n, (const affine_t *)d_points_mont, c->fk[F_].m32, w.points29.as<affine_t>()); });
        twin = w.points29.as<affine_t>();
    }
    DISPATCH_FIELD(base_field_of(curve), { rc = run_msm<F_>(c, sh, d_scalars, (const affine_t *)d_points_mont, d_out_words, (xyzz_t *)d_out_xyzz, twin); });
    return rc;
}

extern "C" int mina_msm(mina_ctx *c, int curve, size_t n, const uint8_t *bases, const uint8_t *scalars, uint8_t *out) {
    if (!c || !out || (n && (!bases || !scalars))) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (n == 0) { memset(out, 0, 64); return MINA_OK; }
    if (n > (1u << 24)) return fail(MINA_ERR_ARG, "n too large");
    if (!scalars_below_2_255(scalars, n)) return fail(MINA_ERR_ARG, "scalar >= 2^255");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    MsmWorkspace &w = c->L->ws;
    if ((rc = w.scalars.ensure(n * 32))) return rc;
    if ((rc = w.points.ensure(n * sizeof(affine_t)))) return rc;
    if ((rc = c->L->tmp_a.ensure(n * 64))) return rc;
    if ((rc = w.out_words.ensure(17 * 4))) return rc;
    HIPC(hipMemcpyAsync(w.scalars.p, scalars, n * 32, hipMemcpyHostToDevice, c->L->stream));
    HIPC(hipMemcpyAsync(c->L->tmp_a.p, bases, n * 64, hipMemcpyHostToDevice, c->L->stream));
    DISPATCH_FIELD(base_field_of(curve), {
        points_to_mont_kernel<F_><<<cdiv(n, 256), 256, 0, c->L->stream>>>((uint32_t)n, c->L->tmp_a.as<uint32_t>(), c->fk[F_].r2, w.points.as<affine_t>());
    });
    if ((rc = mb_msm_variable(c, curve, (uint32_t)n, w.scalars.as<uint32_t>(), w.points.p, w.out_words.as<uint32_t>(), nullptr))) return rc;
    uint32_t hw[17];
    HIPC(hipMemcpyAsync(hw, w.out_words.p, sizeof hw, hipMemcpyDeviceToHost, c->L->stream));
    HIPC(hipStreamSynchronize(c->L->stream));
    words_to_point_bytes(hw, out);
    return MINA_OK;
}

extern "C" int mina_msm_srs_dev(mina_ctx *c, int curve, size_t n, const void *d_scalars, void *d_out) {
    if (!c || !d_scalars || !d_out) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (n > 0xffffffffu) return fail(MINA_ERR_ARG, "n too large");
    HIPC(hipSetDevice(c->device));
    c->next_lane();
    return mb_msm_fixed(c, curve, (uint32_t)n, (const uint32_t *)d_scalars, (uint32_t *)d_out, nullptr);
}

extern "C" int mina_msm_srs(mina_ctx *c, int curve, size_t n, const uint8_t *scalars, uint8_t *out) {
    if (!c || !out || (n && !scalars)) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (n == 0) { memset(out, 0, 64); return MINA_OK; }
    if (n > 0xffffffffu) return fail(MINA_ERR_ARG, "n too large");
    if (!scalars_below_2_255(scalars, n)) return fail(MINA_ERR_ARG, "scalar >= 2^255");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = c->L->ws.scalars.ensure(n * 32))) return rc;
    if ((rc = c->L->ws.out_words.ensure(17 * 4))) return rc;
    HIPC(hipMemcpyAsync(c->L->ws.scalars.p, scalars, n * 32, hipMemcpyHostToDevice, c->L->stream));
    if ((rc = mb_msm_fixed(c, curve, (uint32_t)n, c->L->ws.scalars.as<uint32_t>(), c->L->ws.out_words.as<uint32_t>(), nullptr))) return rc;
    uint32_t hw[17];
    HIPC(hipMemcpyAsync(hw, c->L->ws.out_words.p, sizeof hw, hipMemcpyDeviceToHost, c->L->stream));
    HIPC(hipStreamSynchronize(c->L->stream));
    words_to_point_bytes(hw, out);
    return MINA_OK;
}

extern "C" int mina_msm_srs_multi(mina_ctx *c, int curve, size_t n, size_t nprob, const uint8_t *scalars, uint8_t *out) {
    if (!c || !out || (n && nprob && !scalars)) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (nprob == 0) return MINA_OK;
    if (n == 0) { memset(out, 0, nprob * 64); return MINA_OK; }
    SrsState &s = c->srs[curve];
    if (s.depth == 0) return fail(MINA_ERR_STATE, "SRS not loaded for this curve");
    if (n > s.depth || nprob > 4096) return fail(MINA_ERR_ARG, "n / nprob out of range");
    if (!scalars_below_2_255(scalars, n * nprob)) return fail(MINA_ERR_ARG, "scalar >= 2^255");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    MsmWorkspace &w = c->L->ws;
    if ((rc = w.scalars.ensure(nprob * n * 32))) return rc;
    if ((rc = w.out_words.ensure(nprob * 17 * 4))) return rc;
    HIPC(hipMemcpyAsync(w.scalars.p, scalars, nprob * n * 32, hipMemcpyHostToDevice, c->L->stream));
    if ((rc = mb_msm_table(c, curve, s.table.p, s.depth, s.c, s.W, 0, (uint32_t)n, (uint32_t)nprob, w.scalars.as<uint32_t>(), w.out_words.as<uint32_t>(), nullptr))) return rc;
    std::vector<uint32_t> hw(nprob * 17);
    HIPC(hipMemcpyAsync(hw.data(), w.out_words.p, hw.size() * 4, hipMemcpyDeviceToHost, c->L->stream));
    HIPC(hipStreamSynchronize(c->L->stream));
    for (size_t m = 0; m < nprob; ++m) words_to_point_bytes(&hw[m * 17], out + m * 64);
    return MINA_OK;
}

extern "C" int mina_msm_srs_range(mina_ctx *c, int curve, uint32_t first, size_t n, const uint8_t *scalars, uint8_t *out) {
    if (!c || !out || (n && !scalars)) return fail(MINA_ERR_ARG, "null argument");
    if (curve != 0 && curve != 1) return fail(MINA_ERR_ARG, "bad curve");
    if (n == 0) { memset(out, 0, 64); return MINA_OK; }
    if (n > 0xffffffffu) return fail(MINA_ERR_ARG, "n too large");
    if (!scalars_below_2_255(scalars, n)) return fail(MINA_ERR_ARG, "scalar >= 2^255");
    HIPC(hipSetDevice(c->device));
    c->use_lane0();
    int rc;
    if ((rc = c->L->ws.scalars.ensure(n * 32))) return rc;
    if ((rc = c->L->ws.out_words.ensure(17 * 4))) return rc;
    HIPC(hipMemcpyAsync(c->L->ws.scalars.p, scalars, n * 32, hipMemcpyHostToDevice, c->L->stream));
    if ((rc = mb_msm_fixed(c, curve, (uint32_t)n, c->L->ws.scalars.as<uint32_t>(), c->L->ws.out_words.as<uint32_t>(), nullptr, first))) return rc;
    uint32_t hw[17];
    HIPC(hipMemcpyAsync(hw, c->L->ws.out_words.p, sizeof hw, hipMemcpyDeviceToHost, c->L->stream));
    HIPC(hipStreamSynchronize(c->L->stream));
    words_to_point_bytes(hw, out);
    return MINA_OK;
}
