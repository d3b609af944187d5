// msm.cuh -- K1: bucket (Pippenger) multi-scalar multiplication on gfx950.
//
// Replaces ark-ec 0.3 `VariableBaseMSM::multi_scalar_mul` (pin core/Cargo.toml:20,49), the routine
// `poly-commitment` `SRS::verify` / `commit_non_hiding` spend their time in (SURVEY.md 8a rows a7,a8,a10).
//
// Design (not the reference's window-per-rayon-task loop):
//   * fixed-base (SRS) path: at SRS-load time the table T[w][i] = 2^(c*w) * G_i (affine) is built in
//     HBM (64 B/point).  Every window then shares ONE set of 2^(c-1) signed-digit buckets: no per-window
//     bucket reduction and no Horner doubling chain at MSM time.
//   * variable-base path: same kernels with W independent bucket sets + a Horner tail.
//   * scalars -> signed c-bit digits -> counting sort by bucket (atomic slot + exclusive scan + scatter),
//     so a bucket's points are contiguous: gathers of 64-B affine points, no atomics on curve points.
//   * accumulate in two balanced levels: level 1 = fixed-length tasks of <= L affine points of one bucket
//     (XYZZ mixed adds, every lane does the same amount of work whatever the digit distribution),
//     level 2 = per-bucket sum of its task partials.
//   * bucket reduction sum_b (b+1) B_b by wave64 suffix-scan / tree reductions on XYZZ values moved with
//     cross-lane shuffles (three levels of 64-way groups), no serial running sum.
#pragma once
#include "ec.cuh"

namespace mb {

static constexpr int MSM_TASK_LEN = 8;          // L: affine points per level-1 task
static constexpr uint32_t MSM_INVALID = 0xffffffffu;

struct MsmShape {
    uint32_t n;        // scalars
    uint32_t c;        // window bits
    uint32_t W;        // windows (W*c >= 256)
    uint32_t NB;       // buckets per bucket set = 2^(c-1)
    uint32_t nsets;    // 1 (shared buckets, fixed-base table) or W (variable-base)
    uint32_t table_stride;  // fixed-base: points per window in the table (>= n); 0 for variable-base
    uint32_t base_first;    // fixed-base: index of the first SRS base used (scalar i multiplies g[base_first + i])
};

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ uint32_t scalar_bits(const uint32_t *s, uint32_t lo, uint32_t c) {
    // bits [lo, lo+c) of a 256-bit little-endian integer (c <= 24)
    uint32_t limb = lo >> 5, sh = lo & 31;
    if (limb >= 8) return 0;
    uint64_t v = s[limb];
    if (limb + 1 < 8) v |= (uint64_t)s[limb + 1] << 32;
    return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}

// raw unsigned digit of window w
__device__ __forceinline__ uint32_t raw_digit(const uint32_t *s, uint32_t w, uint32_t c) { return scalar_bits(s, w * c, c); }

// K1a: signed digits + per-bucket slot via one returning atomic.  One lane per (window, scalar): every atomic
// of the launch is independent, so their latency overlaps (a lane that walked its 16 windows serially paid 16
// dependent round trips).  Digit rule: d = raw + carry_in; d > 2^(c-1) -> d - 2^c, carry out.  carry_in of window w
// is resolved by looking at the lower windows, which almost always stops at w-1.
static __global__ void msm_digits_kernel(MsmShape sh, const uint32_t *__restrict__ scalars /* n x 8 */,
                                         uint32_t *__restrict__ count, uint32_t *__restrict__ ekey,
                                         uint32_t *__restrict__ eval, uint32_t *__restrict__ eoff) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)sh.n * sh.W) return;
    const uint32_t w = (uint32_t)(e / sh.n), i = (uint32_t)(e % sh.n);
    uint32_t s[8];
    const uint4 *sp = reinterpret_cast<const uint4 *>(scalars + (size_t)i * 8);
    uint4 a = sp[0], b = sp[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    const uint32_t half = 1u << (sh.c - 1);
    uint32_t carry = 0;
    for (int j = (int)w - 1; j >= 0; --j) {
        uint32_t r = raw_digit(s, (uint32_t)j, sh.c);
        if (r > half) { carry = 1; break; }
        if (r < half) { carry = 0; break; }
        // r == half: window j carries iff it received a carry itself -> keep looking down
    }
    uint32_t d = raw_digit(s, w, sh.c) + carry;
    uint32_t neg = 0;
    if (d > half) { d = (1u << sh.c) - d; neg = 1; }
    if (d == 0) { ekey[e] = MSM_INVALID; return; }
    const uint32_t bucket = (sh.nsets == 1 ? 0u : w * sh.NB) + (d - 1);
    ekey[e] = bucket;
    eval[e] = (sh.table_stride ? w * sh.table_stride + sh.base_first + i : i) | (neg << 31);
    eoff[e] = atomicAdd(&count[bucket], 1u);
}

// K1b: exclusive scans of count[] and of ceil(count/L) (single block of 1024 lanes; each lane owns a contiguous
// chunk that it reads once with 16-byte loads and keeps in registers when it fits).
static __global__ void __launch_bounds__(1024)
msm_scan_kernel(uint32_t nb_total, const uint32_t *__restrict__ count, uint32_t *__restrict__ start /* nb_total+1 */,
                uint32_t *__restrict__ task_start /* nb_total+1 */) {
    __shared__ uint32_t s_cnt[1024], s_tsk[1024];
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    uint32_t per = (nb_total + nt - 1) / nt;
    per = (per + 3u) & ~3u;                                   // multiple of 4 -> uint4 loads stay aligned
    const uint32_t lo = min(tid * per, nb_total), hi = min(lo + per, nb_total);
    constexpr uint32_t CACHE = 32;
    uint32_t cached[CACHE];
    const bool fits = per <= CACHE;
    uint32_t sc = 0, st = 0;
    if (fits) {
#pragma unroll
        for (uint32_t q = 0; q < CACHE; q += 4) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (lo + q + 3 < hi) v = *reinterpret_cast<const uint4 *>(count + lo + q);
            else { if (lo + q < hi) v.x = count[lo + q]; if (lo + q + 1 < hi) v.y = count[lo + q + 1]; if (lo + q + 2 < hi) v.z = count[lo + q + 2]; }
            cached[q] = v.x; cached[q + 1] = v.y; cached[q + 2] = v.z; cached[q + 3] = v.w;
        }
#pragma unroll
        for (uint32_t q = 0; q < CACHE; ++q) { sc += cached[q]; st += (cached[q] + MSM_TASK_LEN - 1) / MSM_TASK_LEN; }
    } else {
        for (uint32_t b = lo; b < hi; ++b) { uint32_t c = count[b]; sc += c; st += (c + MSM_TASK_LEN - 1) / MSM_TASK_LEN; }
    }
    s_cnt[tid] = sc; s_tsk[tid] = st;
    __syncthreads();
    for (uint32_t d = 1; d < nt; d <<= 1) {
        uint32_t vc = 0, vt = 0;
        if (tid >= d) { vc = s_cnt[tid - d]; vt = s_tsk[tid - d]; }
        __syncthreads();
        s_cnt[tid] += vc; s_tsk[tid] += vt;
        __syncthreads();
    }
    uint32_t pc = s_cnt[tid] - sc, pt = s_tsk[tid] - st;   // exclusive prefix of this lane's chunk
    if (fits) {
#pragma unroll
        for (uint32_t q = 0; q < CACHE; ++q) {
            if (lo + q < hi) { start[lo + q] = pc; task_start[lo + q] = pt; }
            pc += cached[q]; pt += (cached[q] + MSM_TASK_LEN - 1) / MSM_TASK_LEN;
        }
    } else {
        for (uint32_t b = lo; b < hi; ++b) {
            uint32_t c = count[b];
            start[b] = pc; task_start[b] = pt;
            pc += c; pt += (c + MSM_TASK_LEN - 1) / MSM_TASK_LEN;
        }
    }
    if (tid == nt - 1) { start[nb_total] = s_cnt[tid]; task_start[nb_total] = s_tsk[tid]; }
}

// K1c: scatter point references into bucket order
static __global__ void msm_scatter_kernel(size_t total, const uint32_t *__restrict__ ekey, const uint32_t *__restrict__ eval,
                                   const uint32_t *__restrict__ eoff, const uint32_t *__restrict__ start,
                                   uint32_t *__restrict__ sorted) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    uint32_t k = ekey[e];
    if (k == MSM_INVALID) return;
    sorted[start[k] + eoff[e]] = eval[e];
}

__device__ __forceinline__ affine_t load_affine(const affine_t *__restrict__ p) {
    affine_t r;
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    r.x.v[0] = a.x; r.x.v[1] = a.y; r.x.v[2] = a.z; r.x.v[3] = a.w;
    r.x.v[4] = b.x; r.x.v[5] = b.y; r.x.v[6] = b.z; r.x.v[7] = b.w;
    r.y.v[0] = c.x; r.y.v[1] = c.y; r.y.v[2] = c.z; r.y.v[3] = c.w;
    r.y.v[4] = d.x; r.y.v[5] = d.y; r.y.v[6] = d.z; r.y.v[7] = d.w;
    return r;
}

// K1d: level-1 accumulate.  One lane per task = <= L consecutive sorted entries of one bucket.
template <int F>
__global__ void __launch_bounds__(256)
msm_accumulate_kernel(uint32_t nb_total, const uint32_t *__restrict__ start, const uint32_t *__restrict__ task_start,
                      const uint32_t *__restrict__ sorted, const affine_t *__restrict__ points, fe_t one,
                      xyzz_t *__restrict__ partial) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ntasks = task_start[nb_total];
    if (t >= ntasks) return;
    // bucket of task t: largest b with task_start[b] <= t  (task_start is non-decreasing)
    uint32_t lo = 0, hi = nb_total;            // invariant: task_start[lo] <= t < task_start[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (task_start[mid] <= t) lo = mid; else hi = mid;
    }
    const uint32_t b = lo;
    const uint32_t j = t - task_start[b];
    const uint32_t beg = start[b] + j * MSM_TASK_LEN;
    const uint32_t end = min(beg + MSM_TASK_LEN, start[b + 1]);
    xyzz_t acc = xyzz_inf();
    for (uint32_t e = beg; e < end; ++e) {
        uint32_t ref = sorted[e];
        affine_t p = load_affine(points + (ref & 0x7fffffffu));
        if (aff_is_inf(p)) continue;
        if (ref >> 31) p.y = fe_neg<F>(p.y);
        xyzz_add_affine<F>(acc, p.x, p.y, one);
    }
    partial[t] = acc;
}

// K1e: level-2: bucket b = sum of its task partials
template <int F>
__global__ void __launch_bounds__(256)
msm_bucket_sum_kernel(uint32_t nb_total, const uint32_t *__restrict__ task_start, const xyzz_t *__restrict__ partial,
                      xyzz_t *__restrict__ buckets) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb_total) return;
    const uint32_t lo = task_start[b], hi = task_start[b + 1];
    xyzz_t acc = xyzz_inf();
    for (uint32_t t = lo; t < hi; ++t) xyzz_add<F>(acc, partial[t]);
    buckets[b] = acc;
}

// ---------------------------------------------------------------- wave64 XYZZ collectives
__device__ __forceinline__ fe_t shfl_down_fe(const fe_t &a, int d) {
    fe_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = __shfl_down(a.v[i], d, 64);
    return r;
}
__device__ __forceinline__ xyzz_t shfl_down_xyzz(const xyzz_t &a, int d) {
    xyzz_t r; r.x = shfl_down_fe(a.x, d); r.y = shfl_down_fe(a.y, d);
    r.zz = shfl_down_fe(a.zz, d); r.zzz = shfl_down_fe(a.zzz, d); return r;
}
// lane 0 gets the sum over lanes [0, width) of v  (width = power of two <= 64; lanes >= width must hold infinity)
template <int F> __device__ __forceinline__ xyzz_t wave_sum(xyzz_t v, int width = 64) {
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int d = width >> 1; d >= 1; d >>= 1) {
        xyzz_t o = shfl_down_xyzz(v, d);
        if (lane + d < width) xyzz_add<F>(v, o);
    }
    return v;
}
// lane 0 gets  sum_l v_l  (in `sum`) and  sum_l l * v_l  (returned), l < width
template <int F> __device__ __forceinline__ xyzz_t wave_weighted_sum(xyzz_t v, xyzz_t &sum, int width = 64) {
    const int lane = threadIdx.x & 63;
    // suffix scan: s_l = sum_{i >= l} v_i
#pragma unroll 1
    for (int d = 1; d < width; d <<= 1) {
        xyzz_t o = shfl_down_xyzz(v, d);
        if (lane + d < width) xyzz_add<F>(v, o);
    }
    sum = v;                                   // lane 0: total
    if (lane == 0) v = xyzz_inf();             // sum_{l>=1} s_l = sum_l l * v_l
    return wave_sum<F>(v, width);
}

// K1f: level A of the bucket reduction: one wave per 64 consecutive buckets of one set.
//   out_r[g] = sum_l B[64g+l],  out_ws[g] = sum_l l * B[64g+l]
template <int F>
__global__ void __launch_bounds__(64)
msm_reduce_a_kernel(uint32_t nb_total, const xyzz_t *__restrict__ buckets, xyzz_t *__restrict__ out_r,
                    xyzz_t *__restrict__ out_ws) {
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    const uint32_t b = g * 64 + lane;
    xyzz_t v = (b < nb_total) ? buckets[b] : xyzz_inf();
    xyzz_t sum;
    xyzz_t ws = wave_weighted_sum<F>(v, sum);
    if (lane == 0) { out_r[g] = sum; out_ws[g] = ws; }
}

// K1g: level B.  Block (v, set) = super-group v (64 consecutive level-A groups) of one bucket set; two waves:
//   wave 0:  R'_v = sum_u R_{64v+u},  WS'_v = sum_u u * R_{64v+u}        wave 1:  P'_v = sum_u WS_{64v+u}
template <int F>
__global__ void __launch_bounds__(128)
msm_reduce_b_kernel(uint32_t groups, uint32_t nsuper, const xyzz_t *__restrict__ in_r, const xyzz_t *__restrict__ in_ws,
                    xyzz_t *__restrict__ out_r, xyzz_t *__restrict__ out_w, xyzz_t *__restrict__ out_p) {
    const uint32_t v = blockIdx.x, set = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t g = v * 64 + lane;
    const size_t base = (size_t)set * groups;
    if (wave == 0) {
        xyzz_t rv = (g < groups) ? in_r[base + g] : xyzz_inf();
        xyzz_t sum;
        xyzz_t wsum = wave_weighted_sum<F>(rv, sum);
        if (lane == 0) { out_r[(size_t)set * nsuper + v] = sum; out_w[(size_t)set * nsuper + v] = wsum; }
    } else {
        xyzz_t pv = (g < groups) ? in_ws[base + g] : xyzz_inf();
        xyzz_t psum = wave_sum<F>(pv);
        if (lane == 0) out_p[(size_t)set * nsuper + v] = psum;
    }
}

// K1g': level C.  One block per bucket set over its nsuper (<= 64) super-groups; three waves work concurrently:
//   set total = sum_b (b+1) B_b = P + Rall + 64 * ( sum_v WS'_v + 64 * sum_v v * R'_v )
template <int F>
__global__ void __launch_bounds__(192)
msm_reduce_c_kernel(uint32_t nsuper, const xyzz_t *__restrict__ in_r, const xyzz_t *__restrict__ in_w,
                    const xyzz_t *__restrict__ in_p, xyzz_t *__restrict__ set_total) {
    const uint32_t set = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ xyzz_t sh_vw, sh_rall, sh_ws, sh_p;
    int width = 1; while (width < (int)nsuper) width <<= 1;
    const size_t base = (size_t)set * nsuper;
    if (wave == 0) {
        xyzz_t rv = (lane < nsuper) ? in_r[base + lane] : xyzz_inf();
        xyzz_t rall;
        xyzz_t vw = wave_weighted_sum<F>(rv, rall, width);
        if (lane == 0) { sh_vw = vw; sh_rall = rall; }
    } else if (wave == 1) {
        xyzz_t wv = (lane < nsuper) ? in_w[base + lane] : xyzz_inf();
        xyzz_t s = wave_sum<F>(wv, width);
        if (lane == 0) sh_ws = s;
    } else {
        xyzz_t pv = (lane < nsuper) ? in_p[base + lane] : xyzz_inf();
        xyzz_t s = wave_sum<F>(pv, width);
        if (lane == 0) sh_p = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        xyzz_t t = sh_vw;
        for (int i = 0; i < 6; ++i) t = xyzz_dbl<F>(t);   // 64 * sum_v v R'_v
        xyzz_add<F>(t, sh_ws);                            // = sum_g g R_g
        for (int i = 0; i < 6; ++i) t = xyzz_dbl<F>(t);   // * 64
        xyzz_add<F>(t, sh_p);
        xyzz_add<F>(t, sh_rall);
        set_total[set] = t;
    }
}

// K1h: Horner over bucket sets (variable-base), then normalise to affine (Montgomery) + canonical words.
//   out_words[0..16) = x||y canonical little-endian words, out_words[16] = 1 if infinity.
template <int F>
__global__ void msm_finish_kernel(uint32_t nsets, uint32_t c, const xyzz_t *__restrict__ set_total, fe_t one,
                                  fe_t pm2, xyzz_t *__restrict__ out_xyzz, uint32_t *__restrict__ out_words) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    xyzz_t t = set_total[nsets - 1];
    for (int w = (int)nsets - 2; w >= 0; --w) {
        for (uint32_t i = 0; i < c; ++i) t = xyzz_dbl<F>(t);
        xyzz_add<F>(t, set_total[w]);
    }
    if (out_xyzz) *out_xyzz = t;
    if (!out_words) return;
    if (xyzz_is_inf(t)) { for (int i = 0; i < 16; ++i) out_words[i] = 0; out_words[16] = 1; return; }
    fe_t zi = fe_pow<F>(fe_mul<F>(t.zz, t.zzz), pm2, one);       // 1 / (zz * zzz)
    fe_t izz = fe_mul<F>(zi, t.zzz), izzz = fe_mul<F>(zi, t.zz);
    fe_t x = fe_from_mont<F>(fe_mul<F>(t.x, izz));
    fe_t y = fe_from_mont<F>(fe_mul<F>(t.y, izzz));
    for (int i = 0; i < 8; ++i) { out_words[i] = x.v[i]; out_words[8 + i] = y.v[i]; }
    out_words[16] = 0;
}

// canonical affine bytes (x||y words) -> Montgomery affine; (0,0) stays infinity
template <int F>
__global__ void points_to_mont_kernel(uint32_t n, const uint32_t *__restrict__ in_words, fe_t r2, affine_t *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_t x, y;
    for (int k = 0; k < 8; ++k) { x.v[k] = in_words[(size_t)i * 16 + k]; y.v[k] = in_words[(size_t)i * 16 + 8 + k]; }
    affine_t a; a.x = fe_to_mont<F>(x, r2); a.y = fe_to_mont<F>(y, r2);
    out[i] = a;
}
template <int F>
__global__ void points_from_mont_kernel(uint32_t n, const affine_t *__restrict__ in, uint32_t *__restrict__ out_words) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    affine_t a = in[i];
    fe_t x = fe_from_mont<F>(a.x), y = fe_from_mont<F>(a.y);
    for (int k = 0; k < 8; ++k) { out_words[(size_t)i * 16 + k] = x.v[k]; out_words[(size_t)i * 16 + 8 + k] = y.v[k]; }
}

// SRS window table: table[w * stride + i] = 2^(c*w) * G_i  (affine, Montgomery).  One lane per base.
template <int F>
__global__ void __launch_bounds__(256)
msm_build_table_kernel(uint32_t n, uint32_t stride, uint32_t c, uint32_t W, fe_t one, fe_t pm2, affine_t *__restrict__ table) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    affine_t p = table[i];                      // window 0 = the base itself
    for (uint32_t w = 1; w < W; ++w) {
        if (aff_is_inf(p)) { table[(size_t)w * stride + i] = p; continue; }
        xyzz_t t = xyzz_dbl_affine<F>(p.x, p.y);
        for (uint32_t k = 1; k < c; ++k) t = xyzz_dbl<F>(t);
        if (xyzz_is_inf(t)) { p.x = fe_zero(); p.y = fe_zero(); }
        else {
            fe_t zi = fe_pow<F>(fe_mul<F>(t.zz, t.zzz), pm2, one);
            p.x = fe_mul<F>(t.x, fe_mul<F>(zi, t.zzz));
            p.y = fe_mul<F>(t.y, fe_mul<F>(zi, t.zz));
        }
        table[(size_t)w * stride + i] = p;
    }
}

}  // namespace mb
