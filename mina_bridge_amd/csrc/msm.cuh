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
#include "ec29.cuh"

namespace mb {

static constexpr int MSM_TASK_LEN = 8;          // L: affine points per level-1 task
static constexpr uint32_t MSM_INVALID = 0xffffffffu;

struct MsmShape {
    uint32_t n;        // scalars
    uint32_t c;        // window bits
    uint32_t W;        // windows (W*c >= 256)
    uint32_t NB;       // buckets per bucket set = 2^(c-1)
    uint32_t nsets;    // bucket sets in total = nprob * (1 for a fixed-base table | W for variable-base)
    uint32_t table_stride;  // fixed-base: points per window in the table (>= n); 0 for variable-base
    uint32_t base_first;    // fixed-base: index of the first SRS base used (scalar i multiplies g[base_first + i])
    uint32_t nprob;    // independent problems over the SAME bases in one pipeline (scalars laid out [nprob][n]); each
                       // problem owns its bucket sets and gets its own result
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

__device__ __forceinline__ void load_scalar(const uint32_t *__restrict__ p, uint32_t (&s)[8]) {
    const uint4 *sp = reinterpret_cast<const uint4 *>(p);
    const uint4 a = sp[0], b = sp[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
}
// One (problem m, window w, scalar i) entry: signed digit -> (bucket, point reference | sign << 31); false if the digit is 0.
// Digit rule: d = raw + carry_in; d > 2^(c-1) -> d - 2^c, carry out.  carry_in of window w is resolved by looking at
// the lower windows, which almost always stops at w-1.
__device__ __forceinline__ bool msm_entry(const MsmShape &sh, const uint32_t (&s)[8], uint32_t m, uint32_t w, uint32_t i,
                                          uint32_t &bucket, uint32_t &ref) {
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
    if (d == 0) return false;
    bucket = (sh.table_stride ? m : m * sh.W + w) * sh.NB + (d - 1);
    ref = (sh.table_stride ? w * sh.table_stride + sh.base_first + i : i) | (neg << 31);
    return true;
}

// K1a: signed digits + per-bucket slot via one returning atomic.  One lane per (window, scalar): every atomic
// of the launch is independent, so their latency overlaps (a lane that walked its 16 windows serially paid 16
// dependent round trips).  Fallback sort for bucket counts beyond the partitioned sort below (K1p).
static __global__ void msm_digits_kernel(MsmShape sh, const uint32_t *__restrict__ scalars /* n x 8 */,
                                         uint32_t *__restrict__ count, uint32_t *__restrict__ ekey,
                                         uint32_t *__restrict__ eval, uint32_t *__restrict__ eoff) { mb_wave_prio<1>();
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per_prob = (size_t)sh.n * sh.W;
    if (e >= per_prob * sh.nprob) return;
    const uint32_t m = (uint32_t)(e / per_prob);
    const uint32_t ep = (uint32_t)(e - (size_t)m * per_prob);
    const uint32_t w = ep / sh.n, i = ep % sh.n;
    uint32_t s[8];
    load_scalar(scalars + ((size_t)m * sh.n + i) * 8, s);
    uint32_t bucket, ref;
    if (!msm_entry(sh, s, m, w, i, bucket, ref)) { ekey[e] = MSM_INVALID; return; }
    ekey[e] = bucket;
    eval[e] = ref;
    eoff[e] = atomicAdd(&count[bucket], 1u);
}

// K1b: single-block scan.  Produces, per bucket b with cnt entries (f = cnt / L full tasks, r = cnt % L):
//   start[b]       exclusive prefix of cnt                       (start[nb] = total entries)
//   full_start[b]  exclusive prefix of f                         (full_start[nb] = Ft, number of full tasks)
//   rem_pos[b]     index of b's remainder task among all remainder tasks, ordered by DESCENDING r so that the
//                  lanes of a wave run the same number of adds; MSM_INVALID if r == 0
//   info[0] = Ft, info[1] = number of remainder tasks          (rem_bucket = inverse of rem_pos: K1b' below)
//   info[2] = 0: counter of heavy buckets, filled by K1e
// Memory pattern: the bucket array is walked in rows of 4096 (1024 lanes x uint4), so every global load and store is a
// fully coalesced 16-byte-per-lane access (a lane-contiguous chunking made each store instruction touch 64 lines).
// Pass 1 totals the remainder classes (their offsets are needed before ranks can be assigned), pass 2 emits.
template <int NV>
__device__ __forceinline__ void block_exclusive_scan(uint32_t (&v)[NV], uint32_t (&total)[NV], uint32_t (*s_tot)[16],
                                                     uint32_t (*s_pre)[16], uint32_t *s_all) {
    // in: per-lane values; out: v = exclusive prefix over the block (lane order), total = block totals
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    uint32_t own[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) own[k] = v[k];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int k = 0; k < NV; ++k) { uint32_t t = __shfl_up(v[k], d, 64); if ((int)lane >= d) v[k] += t; }
    }
    __syncthreads();                                          // LDS arrays may still be read from the previous call
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < NV; ++k) s_tot[k][wave] = v[k];
    }
    __syncthreads();
    if (wave == 0) {                                          // one wave scans the (<= 16) wave totals
        uint32_t t[NV], own_t[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) { t[k] = (lane < nwaves) ? s_tot[k][lane & 15] : 0u; own_t[k] = t[k]; }
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
#pragma unroll
            for (int k = 0; k < NV; ++k) { uint32_t u = __shfl_up(t[k], d, 64); if ((int)lane >= d) t[k] += u; }
        }
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < NV; ++k) s_pre[k][lane] = t[k] - own_t[k];
        }
        if (lane == 15) {
#pragma unroll
            for (int k = 0; k < NV; ++k) s_all[k] = t[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) { total[k] = s_all[k]; v[k] = s_pre[k][wave] + v[k] - own[k]; }
}

static __global__ void __launch_bounds__(1024)
msm_scan_kernel(uint32_t nb_total, const uint32_t *__restrict__ count, uint32_t *__restrict__ start /* nb_total+1 */,
                uint32_t *__restrict__ full_start /* nb_total+1 */, uint32_t *__restrict__ rem_pos /* nb_total */,
                uint32_t *__restrict__ info /* 2 */) { mb_wave_prio<1>();
    constexpr int NCLS = MSM_TASK_LEN - 1;                     // class kappa = L-1-r  (r = L-1 .. 1)
    constexpr int NV = 2 + NCLS;
    __shared__ uint32_t s_tot[NV][16], s_pre[NV][16], s_all[NV];
    const uint32_t tid = threadIdx.x;
    const uint32_t row_elems = blockDim.x * 4;
    const uint32_t nrows = (nb_total + row_elems - 1) / row_elems;       // nb_total is a multiple of 4 (power of two >= 128)
    // all rows of a <= 32768-bucket problem are fetched up front (8 independent loads in flight: one memory latency
    // instead of one per row and pass); larger problems re-read from L2
    constexpr uint32_t ROWS_CACHED = 8;
    uint4 rows_c[ROWS_CACHED];
#pragma unroll
    for (uint32_t r = 0; r < ROWS_CACHED; ++r) {
        const uint32_t idx = r * row_elems + tid * 4;
        rows_c[r] = (r < nrows && idx < nb_total) ? *reinterpret_cast<const uint4 *>(count + idx) : make_uint4(0, 0, 0, 0);
    }
    auto load4 = [&](uint32_t row) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < ROWS_CACHED) {
#pragma unroll
            for (uint32_t r = 0; r < ROWS_CACHED; ++r) if (r == row) v = rows_c[r];
            return v;
        }
        const uint32_t idx = row * row_elems + tid * 4;
        return (idx < nb_total) ? *reinterpret_cast<const uint4 *>(count + idx) : v;
    };
    // pass 1: class totals
    uint32_t cls_tot[NV];
    {
        uint32_t v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = 0;
        for (uint32_t row = 0; row < nrows; ++row) {
            const uint4 c4 = load4(row);
            const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t r = c[e] % MSM_TASK_LEN;
#pragma unroll
                for (int k = 0; k < NCLS; ++k) v[2 + k] += (r == (uint32_t)(MSM_TASK_LEN - 1 - k)) ? 1u : 0u;
            }
        }
        block_exclusive_scan<NV>(v, cls_tot, s_tot, s_pre, s_all);
    }
    uint32_t carry[NV];                                       // running offsets: entries, full tasks, rank within each class
    carry[0] = 0; carry[1] = 0;
    {
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < NCLS; ++k) { carry[2 + k] = run; run += cls_tot[2 + k]; }
        if (tid == 0) info[1] = run;
    }
    // pass 2: emit
    for (uint32_t row = 0; row < nrows; ++row) {
        const uint4 c4 = load4(row);
        const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
        uint32_t v[NV], tot[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[0] += c[e]; v[1] += c[e] / MSM_TASK_LEN;
            const uint32_t r = c[e] % MSM_TASK_LEN;
#pragma unroll
            for (int k = 0; k < NCLS; ++k) v[2 + k] += (r == (uint32_t)(MSM_TASK_LEN - 1 - k)) ? 1u : 0u;
        }
        block_exclusive_scan<NV>(v, tot, s_tot, s_pre, s_all);
        uint32_t pc = carry[0] + v[0], pf = carry[1] + v[1], rk[NCLS];
#pragma unroll
        for (int k = 0; k < NCLS; ++k) rk[k] = carry[2 + k] + v[2 + k];
        uint32_t o_start[4], o_full[4], o_rem[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o_start[e] = pc; o_full[e] = pf;
            pc += c[e]; pf += c[e] / MSM_TASK_LEN;
            const uint32_t r = c[e] % MSM_TASK_LEN;
            uint32_t pos = MSM_INVALID;
#pragma unroll
            for (int k = 0; k < NCLS; ++k) if (r == (uint32_t)(MSM_TASK_LEN - 1 - k)) { pos = rk[k]; rk[k] += 1; }
            o_rem[e] = pos;
        }
        const uint32_t idx = row * row_elems + tid * 4;
        if (idx < nb_total) {
            *reinterpret_cast<uint4 *>(start + idx) = make_uint4(o_start[0], o_start[1], o_start[2], o_start[3]);
            *reinterpret_cast<uint4 *>(full_start + idx) = make_uint4(o_full[0], o_full[1], o_full[2], o_full[3]);
            *reinterpret_cast<uint4 *>(rem_pos + idx) = make_uint4(o_rem[0], o_rem[1], o_rem[2], o_rem[3]);
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) carry[k] += tot[k];
    }
    if (tid == 0) { start[nb_total] = carry[0]; full_start[nb_total] = carry[1]; info[0] = carry[1]; info[2] = 0; info[3] = 0; }   // info[2]: heavy-bucket counter (K1e); info[3]: redo counter of the 29-bit accumulate kernels
}

// ---------------------------------------------------------------- K1p: partitioned counting sort (no global atomics)
// The atomic sort above costs 1 M device-scope returning atomics and 1 M random 4-byte stores per 2^16 MSM (92 MB of
// fabric traffic for 12 MB of payload); with 16 MSMs in flight it is 28 % of the step time.  K1p sorts in two levels
// with LDS histograms only:
//   level 1: partition = (bucket within its problem) >> fbits (<= 1024 partitions per problem).  Blocks of 1024 lanes own 256 scalars x all windows
//            (each scalar is read once): K1p-a counts per (partition, block), one block scans the P x G table, K1p-c
//            recomputes the digits and writes {ref, fine key} into its slots of the staging array (LDS cursors) --
//            every 64-B line of the staging array is written by ONE block (one XCD's L2 merges the stores);
//   level 2: one block per partition: LDS histogram over its 2^fbits buckets, block scan, bucket counts out (coalesced),
//            references placed at their final positions inside the partition's contiguous range of `sorted`.
struct SortShape {
    uint32_t fbits;    // fine key bits: buckets per partition = 2^fbits (<= 2048)
    uint32_t Pl;       // partitions per problem = ceil(buckets per problem / 2^fbits) (<= 1024)
    uint32_t Gl;       // level-1 blocks per problem = ceil(n / 256)
    uint32_t SB;       // buckets per problem = NB * (bucket sets per problem)
};
// With nprob problems the partition table is block-diagonal (a block only meets its own problem's buckets): it is stored
// as gh[(m * Pl + p) * Gl + g] -- linear in nprob -- and its exclusive scan, taken in that order, is the bucket order.

// K1p-b: in-place exclusive scan of n values by one 1024-lane block (rows of 4096, coalesced uint4); data[n] = total.
// The buffer is padded to a multiple of 4 words past n.
__device__ __forceinline__ void block_excl_scan_inplace(uint32_t n, uint32_t *__restrict__ data, uint32_t (*s_tot)[16], uint32_t (*s_pre)[16],
                                                        uint32_t *s_all) {
    const uint32_t tid = threadIdx.x, row_elems = blockDim.x * 4, nrows = (n + row_elems - 1) / row_elems;
    uint32_t carry = 0;
    for (uint32_t row = 0; row < nrows; ++row) {
        const uint32_t idx = row * row_elems + tid * 4;
        uint4 c4 = make_uint4(0, 0, 0, 0);
        if (idx < n) c4 = *reinterpret_cast<const uint4 *>(data + idx);
        uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) if (idx + e >= n) c[e] = 0;                 // padding words are not data
        uint32_t v[1] = {c[0] + c[1] + c[2] + c[3]}, tot[1];
        block_exclusive_scan<1>(v, tot, s_tot, s_pre, s_all);
        const uint32_t p0 = carry + v[0];
        if (idx < n) *reinterpret_cast<uint4 *>(data + idx) = make_uint4(p0, p0 + c[0], p0 + c[0] + c[1], p0 + c[0] + c[1] + c[2]);
        carry += tot[0];
    }
    __syncthreads();
    if (tid == 0) data[n] = carry;                                              // after every row's padded uint4 store
}

// K1p-a / K1p-c.  SCATTER = false: gh[p * G + blk] = entries of block blk in partition p.
//                 SCATTER = true : gh holds the exclusive scan; entries go to staging[goff[p][blk] + rank].
template <bool SCATTER>
static __global__ void __launch_bounds__(1024)
msm_part_kernel(MsmShape sh, SortShape ss, const uint32_t *__restrict__ scalars, uint32_t *__restrict__ gh, uint2 *__restrict__ staging,
                uint32_t *__restrict__ ekey /* n * W * nprob: (bucket within the problem | sign << 31) or MSM_INVALID, written by the
                                               count pass so that the scatter pass does not cut the digits again */) { mb_wave_prio<1>();
    __shared__ uint32_t cur[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t m = blockIdx.x / ss.Gl, g = blockIdx.x - m * ss.Gl;            // problem, block within the problem
    uint32_t *ghm = gh + (size_t)m * ss.Pl * ss.Gl + g;
    for (uint32_t j = tid; j < ss.Pl; j += 1024) cur[j] = SCATTER ? ghm[(size_t)j * ss.Gl] : 0u;
    __syncthreads();
    const uint32_t i = g * 256 + (tid & 255);                                     // scalar index within the problem
    if (i < sh.n) {
        uint32_t *ek = ekey + (size_t)m * sh.W * sh.n + i;                         // + w * n: coalesced over the 256 scalars of a window
        if (SCATTER) {
            for (uint32_t w = tid >> 8; w < sh.W; w += 4) {
                const uint32_t key = ek[(size_t)w * sh.n];
                if (key == MSM_INVALID) continue;
                const uint32_t lb = key & 0x7fffffffu;
                const uint32_t ref = (sh.table_stride ? w * sh.table_stride + sh.base_first + i : i) | (key & 0x80000000u);
                const uint32_t pos = atomicAdd(&cur[lb >> ss.fbits], 1u);
                staging[pos] = make_uint2(ref, lb & ((1u << ss.fbits) - 1u));
            }
        } else if (sh.c == 16 && sh.W == 16 && sh.table_stride) {
            // The SRS window tables (c = 16): window w's raw digit IS the w-th 16-bit half-word of the scalar, read straight from memory (four lanes share the 32-byte
            // line; it stays in L1).  The generic path below cuts digits of any width out of s[8] with a run-time index -- a chain of selects: 114 instructions per entry
            // against ~25 here (profiles/msm_valu.json: 14.9 M of a C2 call's 291 M wave-instructions were this pass).  Same digits, same carries (msm_entry's rule).
            const uint16_t *__restrict__ hw = reinterpret_cast<const uint16_t *>(scalars + ((size_t)m * sh.n + i) * 8);
            for (uint32_t w = tid >> 8; w < 16; w += 4) {
                uint32_t carry = 0;
                for (int j = (int)w - 1; j >= 0; --j) { const uint32_t r = hw[j]; if (r > 0x8000u) { carry = 1; break; } if (r < 0x8000u) break; }
                uint32_t d = (uint32_t)hw[w] + carry, neg = 0;
                if (d > 0x8000u) { d = 0x10000u - d; neg = 1; }
                if (d == 0) { ek[(size_t)w * sh.n] = MSM_INVALID; continue; }
                const uint32_t lb = m * sh.NB + (d - 1) - m * ss.SB;
                ek[(size_t)w * sh.n] = lb | (neg << 31);
                atomicAdd(&cur[lb >> ss.fbits], 1u);
            }
        } else {
            uint32_t s[8];
            load_scalar(scalars + ((size_t)m * sh.n + i) * 8, s);
            for (uint32_t w = tid >> 8; w < sh.W; w += 4) {
                uint32_t bucket, ref;
                if (!msm_entry(sh, s, m, w, i, bucket, ref)) { ek[(size_t)w * sh.n] = MSM_INVALID; continue; }
                const uint32_t lb = bucket - m * ss.SB;
                ek[(size_t)w * sh.n] = lb | (ref & 0x80000000u);
                atomicAdd(&cur[lb >> ss.fbits], 1u);
            }
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t j = tid; j < ss.Pl; j += 1024) ghm[(size_t)j * ss.Gl] = cur[j];
    }
}

// K1p-b as its own one-block launch.  (Letting the last block of K1p-a do it -- device-scope fence + ticket -- was 5x
// slower: every block's release fence writes back its XCD's whole L2.)
static __global__ void __launch_bounds__(1024) msm_excl_scan_kernel(uint32_t n, uint32_t *__restrict__ data) { mb_wave_prio<1>();
    __shared__ uint32_t s_tot[1][16], s_pre[1][16], s_all[1];
    block_excl_scan_inplace(n, data, s_tot, s_pre, s_all);
}

// K1p-d: level 2, one block per (problem, partition).  One pass over the staging entries: a lane keeps its (up to
// PS_KEEP) entries in registers together with the rank a returning LDS atomic gave them inside their bucket; after the
// block scan they go straight to their final positions.  Partitions beyond PS_KEEP * 1024 entries (skewed digits) rank
// the overflow in a second pass behind the kept entries.
static constexpr int PS_KEEP = 12;
static __global__ void __launch_bounds__(1024)
msm_part_sort_kernel(SortShape ss, const uint32_t *__restrict__ goff, const uint2 *__restrict__ staging,
                     uint32_t *__restrict__ count, uint32_t *__restrict__ sorted, uint32_t *__restrict__ info) { mb_wave_prio<1>();
    if (blockIdx.x == 0 && threadIdx.x == 0) { info[2] = 0; info[3] = 0; }     // heavy-bucket counter of K1t-a (K1b zeroes its own); redo counter of the 29-bit accumulate kernels
    __shared__ uint32_t hist[2048], over[2048];
    __shared__ uint32_t s_tot[1][16], s_pre[1][16], s_all[1];
    const uint32_t tid = threadIdx.x, pg = blockIdx.x, nf = 1u << ss.fbits;
    const uint32_t m = pg / ss.Pl, p = pg - m * ss.Pl;
    const uint32_t nvalid = (ss.SB - p * nf < nf) ? ss.SB - p * nf : nf;          // buckets of this partition (last one may be short)
    const uint32_t begin = goff[(size_t)pg * ss.Gl], end = goff[(size_t)(pg + 1) * ss.Gl];   // goff[nprob * Pl * Gl] = total
    const bool has_over = end - begin > (uint32_t)PS_KEEP * 1024u;
    for (uint32_t j = tid; j < nf; j += 1024) { hist[j] = 0; over[j] = 0; }
    __syncthreads();
    uint2 ent[PS_KEEP]; uint32_t rk[PS_KEEP];
#pragma unroll
    for (int j = 0; j < PS_KEEP; ++j) {
        const uint32_t e = begin + tid + (uint32_t)j * 1024u;
        if (e < end) { ent[j] = staging[e]; rk[j] = atomicAdd(&hist[ent[j].y], 1u); }
    }
    if (has_over)
        for (uint32_t e = begin + tid + (uint32_t)PS_KEEP * 1024u; e < end; e += 1024) atomicAdd(&over[staging[e].y], 1u);
    __syncthreads();
    // exclusive scan of the bucket sizes hist + over: lane t owns buckets 2t, 2t+1 (nf <= 2048)
    const uint32_t k0 = (2 * tid < nf) ? hist[2 * tid] : 0u, k1 = (2 * tid + 1 < nf) ? hist[2 * tid + 1] : 0u;
    const uint32_t h0 = k0 + ((2 * tid < nf) ? over[2 * tid] : 0u), h1 = k1 + ((2 * tid + 1 < nf) ? over[2 * tid + 1] : 0u);
    uint32_t v[1] = {h0 + h1}, tot[1];
    block_exclusive_scan<1>(v, tot, s_tot, s_pre, s_all);
    const uint32_t b0 = m * ss.SB + p * nf + 2 * tid;
    if (2 * tid < nvalid) count[b0] = h0;
    if (2 * tid + 1 < nvalid) count[b0 + 1] = h1;
    __syncthreads();                                                            // all reads of hist / over done
    if (2 * tid < nf) { hist[2 * tid] = v[0]; over[2 * tid] = v[0] + k0; }        // hist: bucket start, over: cursor behind the kept entries
    if (2 * tid + 1 < nf) { hist[2 * tid + 1] = v[0] + h0; over[2 * tid + 1] = v[0] + h0 + k1; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PS_KEEP; ++j) {
        const uint32_t e = begin + tid + (uint32_t)j * 1024u;
        if (e < end) sorted[begin + hist[ent[j].y] + rk[j]] = ent[j].x;
    }
    if (has_over)
        for (uint32_t e = begin + tid + (uint32_t)PS_KEEP * 1024u; e < end; e += 1024) {
            const uint2 x = staging[e];
            sorted[begin + atomicAdd(&over[x.y], 1u)] = x.x;
        }
}

// K1b': rem_bucket = inverse permutation of rem_pos (random 4-byte scatter, spread over the whole chip)
static __global__ void msm_rem_invert_kernel(uint32_t nb_total, const uint32_t *__restrict__ rem_pos, uint32_t *__restrict__ rem_bucket) { mb_wave_prio<1>();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb_total) return;
    const uint32_t p = rem_pos[b];
    if (p != MSM_INVALID) rem_bucket[p] = b;
}

// K1c: scatter point references into bucket order
static __global__ void msm_scatter_kernel(size_t total, const uint32_t *__restrict__ ekey, const uint32_t *__restrict__ eval,
                                   const uint32_t *__restrict__ eoff, const uint32_t *__restrict__ start,
                                   uint32_t *__restrict__ sorted) { mb_wave_prio<1>();
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

// K1d: level-1 accumulate.  One lane per task.  Tasks [0, Ft) are the full tasks (exactly L consecutive sorted
// entries of one bucket, bucket found by binary search in full_start); tasks [Ft, Ft+Rt) are the per-bucket remainders,
// ordered by descending length.  A wave therefore runs lanes of (almost) identical trip count.
template <int F>
__global__ void __launch_bounds__(256)
msm_accumulate_kernel(uint32_t nb_total, const uint32_t *__restrict__ start, const uint32_t *__restrict__ full_start,
                      const uint32_t *__restrict__ rem_bucket, const uint32_t *__restrict__ info,
                      const uint32_t *__restrict__ sorted, const affine_t *__restrict__ points, fe_t one,
                      xyzz_t *__restrict__ partial, const uint32_t *__restrict__ redo = nullptr) { mb_wave_prio<1>();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nfull = info[0], nrem = info[1];
    if (redo) { if (t >= info[3]) return; t = redo[t]; }        // redo mode: lane i takes task redo[i], the tasks the 29-bit kernel handed back; lanes beyond info[3] leave at once
    if (t >= nfull + nrem) return;
    uint32_t b, j;
    if (t < nfull) {
        uint32_t lo = 0, hi = nb_total;        // invariant: full_start[lo] <= t < full_start[hi]
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (full_start[mid] <= t) lo = mid; else hi = mid;
        }
        b = lo; j = t - full_start[b];
    } else {
        b = rem_bucket[t - nfull];
        j = (start[b + 1] - start[b]) / MSM_TASK_LEN;        // after the bucket's full tasks
    }
    const uint32_t beg = start[b] + j * MSM_TASK_LEN;
    const uint32_t cnt = min((uint32_t)MSM_TASK_LEN, start[b + 1] - beg);
    // all (<= L) references first, then software-pipelined gathers: the 64-B point of entry e+1 is in flight
    // while the mixed add of entry e runs (a gather from the 64 MiB table is an L2 miss most of the time).
    uint32_t refs[MSM_TASK_LEN];
#pragma unroll
    for (int e = 0; e < MSM_TASK_LEN; ++e) refs[e] = ((uint32_t)e < cnt) ? sorted[beg + e] : 0u;
    xyzz_t acc = xyzz_inf();
    affine_t nxt = load_affine(points + (refs[0] & 0x7fffffffu));
#pragma unroll 1
    for (uint32_t e = 0; e < cnt; ++e) {
        affine_t p = nxt;
        const uint32_t ref = refs[0];
#pragma unroll
        for (int q = 0; q + 1 < MSM_TASK_LEN; ++q) refs[q] = refs[q + 1];      // rotate (register moves, no indexing)
        if (e + 1 < cnt) nxt = load_affine(points + (refs[0] & 0x7fffffffu));
        if (aff_is_inf(p)) continue;
        if (ref >> 31) p.y = fe_neg<F>(p.y);
        xyzz_add_affine<F>(acc, p.x, p.y, one);
    }
    partial[t] = acc;
}

// K1e: level-2: bucket b = sum of its task partials (its full tasks are contiguous, plus at most one remainder task).
// Four lanes per bucket (cooperative group law): the chain of ~4-8 dependent XYZZ adds is the latency of this stage.
// A bucket with more than MSM_HEAVY_TASKS partials (skewed digits: a short top window, equal or structured scalars)
// is not summed here -- one quad would walk thousands of dependent adds -- but queued for K1e'.
static constexpr uint32_t MSM_HEAVY_TASKS = 24;
template <int F>
__global__ void __launch_bounds__(256)
msm_bucket_sum_kernel(uint32_t nb_total, const uint32_t *__restrict__ full_start, const uint32_t *__restrict__ rem_pos,
                      uint32_t *__restrict__ info, const xyzz_t *__restrict__ partial, xyzz_t *__restrict__ buckets,
                      uint32_t *__restrict__ heavy) { mb_wave_prio<1>();
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = gid >> 2;
    if (b >= nb_total) return;                               // whole quads leave together
    const uint32_t lo = full_start[b], hi = full_start[b + 1], rp = rem_pos[b];
    if (hi - lo > MSM_HEAVY_TASKS) {                         // uniform within the quad
        if ((gid & 3u) == 0) heavy[atomicAdd(&info[2], 1u)] = b;
        return;
    }
    xyzz_t acc = (rp != MSM_INVALID) ? partial[info[0] + rp] : xyzz_inf();
    for (uint32_t t = lo; t < hi; ++t) xyzz_add_quad<F>(acc, partial[t]);
    if ((gid & 3u) == 0) buckets[b] = acc;
}
// Throughput form of K1e (one lane per bucket, plain XYZZ adds): 30 % fewer issue slots than the cooperative form, 3.5x
// its latency.  Used when the context pipelines independent MSMs over several lanes (the chip is then VALU-bound and
// the latency of one MSM's tail is hidden by the others).
template <int F>
__global__ void __launch_bounds__(256)
msm_bucket_sum_lane_kernel(uint32_t nb_total, const uint32_t *__restrict__ full_start, const uint32_t *__restrict__ rem_pos,
                           uint32_t *__restrict__ info, const xyzz_t *__restrict__ partial, xyzz_t *__restrict__ buckets,
                           uint32_t *__restrict__ heavy) { mb_wave_prio<1>();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb_total) return;
    const uint32_t lo = full_start[b], hi = full_start[b + 1], rp = rem_pos[b];
    if (hi - lo > MSM_HEAVY_TASKS) { heavy[atomicAdd(&info[2], 1u)] = b; return; }
    xyzz_t acc = (rp != MSM_INVALID) ? partial[info[0] + rp] : xyzz_inf();
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
// sum over the `width` (power of two <= 16) quads of a wave; every lane of quad 0 ends with the total
template <int F> __device__ __forceinline__ xyzz_t quadwave_sum(xyzz_t v, int width) {
    const int q = (threadIdx.x & 63) >> 2;
#pragma unroll 1
    for (int d = width >> 1; d >= 1; d >>= 1) {
        xyzz_t o = shfl_down_xyzz(v, 4 * d);
        if (q + d < width) xyzz_add_quad<F>(v, o);
    }
    return v;
}

// ---------------------------------------------------------------- K1t: throughput form of K1b/K1d/K1e
// With many MSMs in flight (pipelined lanes, groups of problems) there are enough buckets to give every bucket its OWN
// lane: no tasks, no partials, no level-2 sums -- 31 mixed adds per bucket instead of 27.6 mixed + 3.4 full adds, and
// 38 MB less traffic per MSM.  Lanes of a wave must then run equally long: K1t-a ranks the buckets by entry count
// (descending, 64 classes; ties in any order), so a wave holds 64 buckets of one class and the longest waves start first.
// Buckets beyond MSM_HEAVY_ENTRIES entries are queued for the block-wide kernel K1t-c.
static constexpr uint32_t MSM_HEAVY_ENTRIES = 192;
static constexpr uint32_t MSM_COUNT_CLASSES = 64;

// K1t-a: one block per problem (its SB buckets; ranks are taken within the problem -- the problems of a launch are
// statistically alike, so this balances as well as a global ranking and scales with the group size).
//   start[b] = exclusive prefix of count, based at the problem's first entry (goff of its first partition);
//   order[m * SB + rank] = bucket, ranked by min(count, 63) descending;
//   heavy[] / info[2] = buckets with more than MSM_HEAVY_ENTRIES entries (info[2] is zeroed by K1p-d).
static __global__ void __launch_bounds__(1024)
msm_order_kernel(SortShape ss, uint32_t nprob, const uint32_t *__restrict__ goff, const uint32_t *__restrict__ count,
                 uint32_t *__restrict__ start, uint32_t *__restrict__ order, uint32_t *__restrict__ info, uint32_t *__restrict__ heavy) { mb_wave_prio<1>();
    __shared__ uint32_t s_tot[1][16], s_pre[1][16], s_all[1];
    __shared__ uint32_t cls_n[MSM_COUNT_CLASSES], cls_cur[MSM_COUNT_CLASSES];
    const uint32_t tid = threadIdx.x, m = blockIdx.x, nb = ss.SB, row_elems = blockDim.x * 4, nrows = (nb + row_elems - 1) / row_elems;
    const uint32_t first = m * nb;                               // SB is a multiple of 128
    count += first; start += first; order += first;
    if (tid < MSM_COUNT_CLASSES) cls_n[tid] = 0;
    __syncthreads();
    uint32_t carry = goff[(size_t)m * ss.Pl * ss.Gl];            // entries of the problems before this one
    for (uint32_t row = 0; row < nrows; ++row) {                 // pass 1: prefix of the counts + class histogram
        const uint32_t idx = row * row_elems + tid * 4;
        uint4 c4 = make_uint4(0, 0, 0, 0);
        if (idx < nb) c4 = *reinterpret_cast<const uint4 *>(count + idx);
        const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
        uint32_t v[1] = {c[0] + c[1] + c[2] + c[3]}, tot[1];
        block_exclusive_scan<1>(v, tot, s_tot, s_pre, s_all);
        const uint32_t p0 = carry + v[0];
        if (idx < nb) {
            *reinterpret_cast<uint4 *>(start + idx) = make_uint4(p0, p0 + c[0], p0 + c[0] + c[1], p0 + c[0] + c[1] + c[2]);
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(&cls_n[c[e] < MSM_COUNT_CLASSES - 1 ? c[e] : MSM_COUNT_CLASSES - 1], 1u);
        }
        carry += tot[0];
    }
    __syncthreads();
    if (tid == 0) {
        if (m == nprob - 1) start[nb] = carry;                   // start[nb_total] = total entries
        uint32_t run = 0;                                        // descending classes: 63, 62, ..., 0
        for (int k = MSM_COUNT_CLASSES - 1; k >= 0; --k) { cls_cur[k] = run; run += cls_n[k]; }
    }
    __syncthreads();
    for (uint32_t row = 0; row < nrows; ++row) {                 // pass 2: ranks
        const uint32_t idx = row * row_elems + tid * 4;
        if (idx >= nb) continue;
        const uint4 c4 = *reinterpret_cast<const uint4 *>(count + idx);
        const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t k = c[e] < MSM_COUNT_CLASSES - 1 ? c[e] : MSM_COUNT_CLASSES - 1;
            order[atomicAdd(&cls_cur[k], 1u)] = first + idx + e;
            if (c[e] > MSM_HEAVY_ENTRIES) heavy[atomicAdd(&info[2], 1u)] = first + idx + e;
        }
    }
}

// K1t-b: bucket = sum of its sorted entries.  Lane r takes the buckets of rank r and nb-1-r of its problem (the longest with the
// shortest, ...): every lane then runs ~2x the mean entry count, so the waves of a launch end together instead of
// leaving the SIMDs with one long wave each (isolated 8-MSM launch: 696 -> 58x us).
// At most 2 waves per SIMD: measured +2 % proofs/s with 16 lanes in flight against the default 4 (the same gain as capping the
// residency with dynamic LDS), and nothing lost when a launch runs alone (2 048 waves = 2 per SIMD anyway).
template <int F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
msm_accumulate_bucket_kernel(uint32_t nb_total, uint32_t nb_prob, const uint32_t *__restrict__ start, const uint32_t *__restrict__ order,
                             const uint32_t *__restrict__ sorted, const affine_t *__restrict__ points, fe_t one,
                             xyzz_t *__restrict__ buckets) { mb_wave_prio<1>();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nb_total / 2) return;                               // bucket counts are multiples of 128
    const uint32_t m = r / (nb_prob / 2), lr = r - m * (nb_prob / 2);   // ranks are per problem
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const uint32_t b = order[m * nb_prob + (half ? nb_prob - 1 - lr : lr)];
        const uint32_t beg = start[b], cnt = start[b + 1] - beg;
        if (cnt > MSM_HEAVY_ENTRIES) continue;                   // K1t-c writes it
        xyzz_t acc = xyzz_inf();
        if (cnt) {
            uint32_t ref = sorted[beg], ref_n = cnt > 1 ? sorted[beg + 1] : 0u;
            affine_t nxt = load_affine(points + (ref & 0x7fffffffu));
#pragma unroll 1
            for (uint32_t e = 0; e < cnt; ++e) {                 // the point of entry e+1 and the reference of e+2 are in flight
                affine_t p = nxt;
                const uint32_t cur = ref;
                ref = ref_n;
                if (e + 1 < cnt) nxt = load_affine(points + (ref & 0x7fffffffu));
                if (e + 2 < cnt) ref_n = sorted[beg + e + 2];
                if (aff_is_inf(p)) continue;
                if (cur >> 31) p.y = fe_neg<F>(p.y);
                xyzz_add_affine<F>(acc, p.x, p.y, one);
            }
        }
        buckets[b] = acc;
    }
}

// The pre-split window table (round 5; mina_verify_tuning.msm_fp29 = 2): one 128-byte record per point -- x, y AND p - y as nine 29-bit limbs each, in the 2^261
// domain, plus an infinity flag -- so that the accumulate kernels neither convert 8 x 32 words into limbs (32 instructions per point) nor negate y (18): a negative
// digit LOADS the other y.  Twice the gather traffic of the 64-byte twin (two 64-B lines per point instead of one), 128 MiB per curve instead of 64.
struct alignas(128) tab29_t { uint32_t w[32]; };         // x: w[0..8]; infinity: w[9] != 0; y: w[12..20]; p - y: w[22..30]
static constexpr int TAB29_X = 0, TAB29_INF = 9, TAB29_Y = 12, TAB29_NY = 22;
struct tab29_pt { fe29_t x, y; uint32_t inf; };
__device__ __forceinline__ tab29_pt load_tab29(const tab29_t *__restrict__ tab, uint32_t ref) {
    const uint32_t *__restrict__ w = tab[ref & 0x7fffffffu].w;
    const uint32_t *__restrict__ yw = w + ((ref >> 31) ? TAB29_NY : TAB29_Y);
    tab29_pt p;
#pragma unroll
    for (int i = 0; i < L29; ++i) { p.x.v[i] = w[TAB29_X + i]; p.y.v[i] = yw[i]; }
    p.inf = w[TAB29_INF];
    return p;
}
template <int F>
__global__ void msm_table29s_kernel(size_t n, const affine_t *__restrict__ in, fe_t m32, tab29_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const affine_t p = in[i];
    tab29_t t; for (int k = 0; k < 32; ++k) t.w[k] = 0u;
    if (aff_is_inf(p)) t.w[TAB29_INF] = 1u;               // the SAME predicate as the 8 x 32 kernels and the redo path (x | y == 0)
    else {
        const fe_t y = fe_mul<F>(p.y, m32);
        const fe29_t x29 = fe29_from_words(fe_mul<F>(p.x, m32)), y29 = fe29_from_words(y), ny29 = fe29_from_words(fe_neg<F>(y));
        for (int k = 0; k < L29; ++k) { t.w[TAB29_X + k] = x29.v[k]; t.w[TAB29_Y + k] = y29.v[k]; t.w[TAB29_NY + k] = ny29.v[k]; }
    }
    out[i] = t;
}

// timing probe only (tools/probes: -DMB_GATHER_MASK=0x3ffu makes every gather of the 29-bit accumulate kernels land in the first 64 KiB of the table -- WRONG sums, the
// same instruction stream: what the kernels would run at if the gathers cost nothing; profiles/r06_k1.md)
#ifndef MB_GATHER_MASK
#define MB_GATHER_MASK 0x7fffffffu
#endif
// K1t-b on 29-bit limbs (ec29.cuh): the same lanes, the same order of additions, points gathered from the 2^261-domain twin of the window table (TAB = 1: 64-byte
// records of 8 x 32 words, converted and negated here) or from the pre-split table (TAB = 2: tab29_t); the bucket leaves in the 8 x 32 form.  8-MSM launch alone on
// the GPU: 575 us against 685 us for the 8 x 32 kernel (profiles/r04_k1.md); the residency cap makes no difference here (2 / 3 / 4 / 8 waves per SIMD: 12.5 - 12.6 k
// checks/s): kept at the 8 x 32 kernel's 2.
template <int F, int TAB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
msm_accumulate_bucket29_kernel(uint32_t nb_total, uint32_t nb_prob, const uint32_t *__restrict__ start, const uint32_t *__restrict__ order,
                               const uint32_t *__restrict__ sorted, const void *__restrict__ points29_, fe_t one, fe_t m32,
                               xyzz_t *__restrict__ buckets, uint32_t *__restrict__ info, uint32_t *__restrict__ redo,
                               xyzz29_t *__restrict__ buckets29 /* non-null: the bucket STAYS on 29-bit limbs (the 2-D reduction then runs on them too: msm_segsum29_kernel) */) { mb_wave_prio<1>();
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nb_total / 2) return;
    // lane r takes the buckets of rank r and nb-1-r (the waves of a launch end together).  One bucket per lane in rank order -- twice the waves, 4 per SIMD when a
    // launch is alone -- was measured in round 5: the same rate with 16 lanes in flight (12.67 against 12.76 k checks/s on one box), 15 % slower alone (6.9 against 8.2 k)
    const uint32_t m = r / (nb_prob / 2), lr = r - m * (nb_prob / 2);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const uint32_t b = order[m * nb_prob + (half ? nb_prob - 1 - lr : lr)];
        const uint32_t beg = start[b], cnt = start[b + 1] - beg;
        if (cnt > MSM_HEAVY_ENTRIES) continue;                   // K1t-c writes it
        xyzz29_t acc; bool inf = true, exact = true;
        if (cnt) {
            uint32_t ref = sorted[beg], ref_n = cnt > 1 ? sorted[beg + 1] : 0u;
            if constexpr (TAB == 2) {
                const tab29_t *__restrict__ tab = (const tab29_t *)points29_;
                tab29_pt p = load_tab29(tab, ref);
#pragma unroll 1
                for (uint32_t e = 0; e < cnt && exact; ++e) {
                    ref = ref_n;
                    if (e + 2 < cnt) ref_n = sorted[beg + e + 2];
                    // the next point is gathered INTO p's registers as soon as this add has consumed them (after its first two products)
                    auto next = [&]() { if (e + 1 < cnt) p = load_tab29(tab, ref); };
                    if (p.inf) { next(); continue; }
                    exact = xyzz29_add_affine<F>(acc, inf, p.x, p.y, [&]() { return p.y; }, next, m32);                         // the digit's sign chose y or p - y at the load
                }
            } else {
                const affine_t *__restrict__ points29 = (const affine_t *)points29_;
                affine_t p = load_affine(points29 + (ref & MB_GATHER_MASK));
#pragma unroll 1
                for (uint32_t e = 0; e < cnt && exact; ++e) {
                    const bool is_inf = aff_is_inf(p);               // infinity is (0, 0): the predicate of the 8 x 32 kernels and of the redo path
                    const fe29_t px = fe29_from_words(p.x), py = fe29_from_words(p.y);
                    const uint32_t cur = ref;
                    ref = ref_n;
                    if (e + 1 < cnt) p = load_affine(points29 + (ref & MB_GATHER_MASK));      // the words are dead once converted: the next gather lands in their registers
                    if (e + 2 < cnt) ref_n = sorted[beg + e + 2];
                    if (is_inf) continue;
                    exact = xyzz29_add_affine<F>(acc, inf, px, py, (cur >> 31) != 0, m32);                                      // a negative digit adds (x, p - y) (y != 0 on these curves)
                }
            }
        }
        if (!exact) redo[atomicAdd(&info[3], 1u)] = b;           // two of its points are equal or opposite: msm_bucket_redo_kernel sums this bucket with the 8 x 32 law
        else if (buckets29) { if (inf) { acc.x = fe29_zero(); acc.y = acc.x; acc.zz = acc.x; acc.zzz = acc.x; } buckets29[b] = acc; }      // infinity = zz 0, as in the 8 x 32 form
        else buckets[b] = xyzz29_leave<F>(acc, inf, one);
    }
#endif
}
// the buckets the 8 x 32 kernels wrote (the heavy ones, info[2], and the redone ones, info[3]) into the 29-bit form beside the others: x 2^256 -> x 2^261 is one
// product by mont(32) per coordinate.  Fixed-size launch, grid-stride over both lists.
template <int F>
__global__ void __launch_bounds__(64)
msm_buckets_to29_kernel(const uint32_t *__restrict__ info, const uint32_t *__restrict__ heavy, const uint32_t *__restrict__ redo, const xyzz_t *__restrict__ buckets, fe_t m32,
                        xyzz29_t *__restrict__ buckets29) { mb_wave_prio<1>();
    const uint32_t nh = info[2], nr = info[3];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nh + nr; i += gridDim.x * blockDim.x) {
        const uint32_t b = i < nh ? heavy[i] : redo[i - nh];
        const xyzz_t v = buckets[b];
        xyzz29_t o;
        if (xyzz_is_inf(v)) { o.x = fe29_zero(); o.y = o.x; o.zz = o.x; o.zzz = o.x; }
        else { o.x = fe29_from_words(fe_mul<F>(v.x, m32)); o.y = fe29_from_words(fe_mul<F>(v.y, m32)); o.zz = fe29_from_words(fe_mul<F>(v.zz, m32)); o.zzz = fe29_from_words(fe_mul<F>(v.zzz, m32)); }
        buckets29[b] = o;
    }
}
// the buckets the 29-bit kernel handed back (info[3] of them; none on SRS points): one lane each, the 8 x 32 law with all its cases.  Fixed-size launch.
template <int F>
__global__ void __launch_bounds__(64)
msm_bucket_redo_kernel(const uint32_t *__restrict__ start, const uint32_t *__restrict__ info, const uint32_t *__restrict__ redo,
                       const uint32_t *__restrict__ sorted, const affine_t *__restrict__ points, fe_t one, xyzz_t *__restrict__ buckets) { mb_wave_prio<1>();
    const uint32_t n = info[3];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t b = redo[i], beg = start[b], end = start[b + 1];
        xyzz_t acc = xyzz_inf();
        for (uint32_t e = beg; e < end; ++e) {
            const uint32_t ref = sorted[e];
            affine_t p = load_affine(points + (ref & 0x7fffffffu));
            if (aff_is_inf(p)) continue;
            if (ref >> 31) p.y = fe_neg<F>(p.y);
            xyzz_add_affine<F>(acc, p.x, p.y, one);
        }
        buckets[b] = acc;
    }
}
// K1d on 29-bit limbs: one lane per task of <= 8 entries (the single-MSM form); TAB as above
template <int F, int TAB>
__global__ void __launch_bounds__(256)
msm_accumulate29_kernel(uint32_t nb_total, const uint32_t *__restrict__ start, const uint32_t *__restrict__ full_start,
                        const uint32_t *__restrict__ rem_bucket, const uint32_t *__restrict__ info,
                        const uint32_t *__restrict__ sorted, const void *__restrict__ points29_, fe_t one, fe_t m32,
                        xyzz_t *__restrict__ partial, uint32_t *__restrict__ info_rw, uint32_t *__restrict__ redo) { mb_wave_prio<1>();
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nfull = info[0], nrem = info[1];
    if (t >= nfull + nrem) return;
    uint32_t b, j;
    if (t < nfull) {
        uint32_t lo = 0, hi = nb_total;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (full_start[mid] <= t) lo = mid; else hi = mid; }
        b = lo; j = t - full_start[b];
    } else { b = rem_bucket[t - nfull]; j = (start[b + 1] - start[b]) / MSM_TASK_LEN; }
    const uint32_t beg = start[b] + j * MSM_TASK_LEN;
    const uint32_t cnt = min((uint32_t)MSM_TASK_LEN, start[b + 1] - beg);
    uint32_t refs[MSM_TASK_LEN];
#pragma unroll
    for (int e = 0; e < MSM_TASK_LEN; ++e) refs[e] = ((uint32_t)e < cnt) ? sorted[beg + e] : 0u;
    xyzz29_t acc; bool inf = true, exact = true;
    if constexpr (TAB == 2) {
        const tab29_t *__restrict__ tab = (const tab29_t *)points29_;
        tab29_pt p = load_tab29(tab, refs[0]);
#pragma unroll 1
        for (uint32_t e = 0; e < cnt && exact; ++e) {
#pragma unroll
            for (int q = 0; q + 1 < MSM_TASK_LEN; ++q) refs[q] = refs[q + 1];
            auto next = [&]() { if (e + 1 < cnt) p = load_tab29(tab, refs[0]); };
            if (p.inf) { next(); continue; }
            exact = xyzz29_add_affine<F>(acc, inf, p.x, p.y, [&]() { return p.y; }, next, m32);
        }
    } else {
        const affine_t *__restrict__ points29 = (const affine_t *)points29_;
        affine_t p = load_affine(points29 + (refs[0] & MB_GATHER_MASK));
#pragma unroll 1
        for (uint32_t e = 0; e < cnt && exact; ++e) {
            const bool is_inf = aff_is_inf(p);                   // infinity is (0, 0): the same predicate as the 8 x 32 kernels
            const fe29_t px = fe29_from_words(p.x), py = fe29_from_words(p.y);
            const uint32_t ref = refs[0];
#pragma unroll
            for (int q = 0; q + 1 < MSM_TASK_LEN; ++q) refs[q] = refs[q + 1];
            if (e + 1 < cnt) p = load_affine(points29 + (refs[0] & MB_GATHER_MASK));
            if (is_inf) continue;
            exact = xyzz29_add_affine<F>(acc, inf, px, py, (ref >> 31) != 0, m32);
        }
    }
    if (exact) partial[t] = xyzz29_leave<F>(acc, inf, one);
    else redo[atomicAdd(&info_rw[3], 1u)] = t;                   // msm_accumulate_kernel<F> in redo mode sums this task with the 8 x 32 law
#endif
}
// the 2^261-domain twin of a window table: every coordinate times 32 (one Montgomery product by mont(32)); infinity (0, 0) stays
template <int F>
__global__ void msm_table29_kernel(size_t n, const affine_t *__restrict__ in, fe_t m32, affine_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    affine_t p = in[i]; p.x = fe_mul<F>(p.x, m32); p.y = fe_mul<F>(p.y, m32); out[i] = p;
}

// K1t-c: heavy buckets, one 256-lane block each (grid-stride over the queue): lane t sums entries t, t+256, ..., then a
// wave tree (shuffles) and a 4-wave tree through LDS.
template <int F>
__global__ void __launch_bounds__(256)
msm_bucket_heavy_entries_kernel(const uint32_t *__restrict__ start, const uint32_t *__restrict__ info, const uint32_t *__restrict__ heavy,
                                const uint32_t *__restrict__ sorted, const affine_t *__restrict__ points, fe_t one,
                                xyzz_t *__restrict__ buckets) { mb_wave_prio<1>();
    __shared__ xyzz_t sh[4];
    const uint32_t nheavy = info[2], lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const uint32_t b = heavy[h], beg = start[b], end = start[b + 1];
        xyzz_t acc = xyzz_inf();
        for (uint32_t e = beg + threadIdx.x; e < end; e += 256) {
            const uint32_t ref = sorted[e];
            affine_t p = load_affine(points + (ref & 0x7fffffffu));
            if (aff_is_inf(p)) continue;
            if (ref >> 31) p.y = fe_neg<F>(p.y);
            xyzz_add_affine<F>(acc, p.x, p.y, one);
        }
#pragma unroll 1
        for (int d = 32; d >= 1; d >>= 1) {
            xyzz_t o = shfl_down_xyzz(acc, d);
            if ((int)lane + d < 64) xyzz_add<F>(acc, o);
        }
        __syncthreads();                                         // sh may still be read from the previous bucket
        if (lane == 0) sh[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            xyzz_t v = sh[0];
            xyzz_add<F>(v, sh[1]); xyzz_add<F>(v, sh[2]); xyzz_add<F>(v, sh[3]);
            buckets[b] = v;
        }
    }
}

// K1e': heavy buckets, one 256-lane block (64 quads) each: quad q sums partials lo+q, lo+q+64, ... , then a wave tree
// and a 4-wave tree through LDS.  Grid-stride over the list K1e built; the launch is fixed-size (no host round trip).
template <int F>
__global__ void __launch_bounds__(256)
msm_bucket_sum_heavy_kernel(const uint32_t *__restrict__ full_start, const uint32_t *__restrict__ rem_pos, const uint32_t *__restrict__ info,
                            const xyzz_t *__restrict__ partial, xyzz_t *__restrict__ buckets, const uint32_t *__restrict__ heavy) { mb_wave_prio<1>();
    __shared__ xyzz_t sh[4];
    const uint32_t nheavy = info[2], q = threadIdx.x >> 2, wave = threadIdx.x >> 6;
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const uint32_t b = heavy[h];
        const uint32_t lo = full_start[b], hi = full_start[b + 1], rp = rem_pos[b];
        xyzz_t acc = (q == 0 && rp != MSM_INVALID) ? partial[info[0] + rp] : xyzz_inf();
        for (uint32_t t = lo + q; t < hi; t += 64) xyzz_add_quad<F>(acc, partial[t]);
        acc = quadwave_sum<F>(acc, 16);
        __syncthreads();                                     // sh may still be read from the previous bucket
        if ((threadIdx.x & 63) == 0) sh[wave] = acc;
        __syncthreads();
        if (wave == 0) {
            xyzz_t v = (q < 4) ? sh[q] : xyzz_inf();
            v = quadwave_sum<F>(v, 4);
            if (threadIdx.x == 0) buckets[b] = v;
        }
    }
}

// ---------------------------------------------------------------- bucket reduction  sum_b (b+1) * B_b
// 2-D scheme: view a bucket set as R rows x C columns (b = r*C + c, C = 128).  Then
//     sum_b (b+1) B_b = Tot + C * sum_r r*Row_r + sum_c c*Col_c,    Row_r = sum_c B[r][c], Col_c = sum_r B[r][c], Tot = sum_r Row_r
// K1f computes all row and column sums as PLAIN sums (each lane adds 8 buckets serially, then a short shuffle tree):
// 2 adds per bucket in total and ~70 % of the add slots useful, against 16 % for a per-wave log-depth weighted
// reduction.  K1g then does the two small weighted sums (R <= 256 rows, 128 columns) with wave suffix-scans.
struct SegSum {            // one family of segments (rows or columns)
    uint32_t nseg;         // segments in total (all sets)
    uint32_t per_set;      // segments per bucket set
    uint32_t len;          // elements per segment
    uint32_t seg_stride;   // distance between first elements of consecutive segments of one set
    uint32_t elem_stride;  // distance between consecutive elements of a segment
    uint32_t lanes;        // lanes cooperating on one segment (power of two, <= 64)
};
static constexpr int SEG_CHUNK = 8;

template <int F, bool COOP>
__global__ void __launch_bounds__(256)
msm_segsum_kernel(uint32_t nb_per_set, SegSum rows, SegSum cols, const xyzz_t *__restrict__ buckets,
                  xyzz_t *__restrict__ out_rows, xyzz_t *__restrict__ out_cols) { mb_wave_prio<1>();
    // COOP: `lanes` counts QUADS per segment, four lanes cooperate on every add (lane-cooperative group law; latency form).
    // !COOP: `lanes` counts single lanes per segment, plain XYZZ adds (throughput form: a third fewer issue slots).
    constexpr uint32_t LPG = COOP ? 4 : 1;                    // lanes per worker
    const bool is_col = blockIdx.y != 0;
    const SegSum sg = is_col ? cols : rows;
    xyzz_t *__restrict__ out = is_col ? out_cols : out_rows;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t qid = gid / LPG;                           // worker index
    const uint32_t seg = qid / sg.lanes, sub = qid % sg.lanes;
    const bool live = seg < sg.nseg;
    xyzz_t acc = xyzz_inf();
    if (live) {
        const size_t base = (size_t)(seg / sg.per_set) * nb_per_set + (size_t)(seg % sg.per_set) * sg.seg_stride;
        const uint32_t per_lane = (sg.len + sg.lanes - 1) / sg.lanes;
        const uint32_t e0 = sub * per_lane, e1 = min(e0 + per_lane, sg.len);
        for (uint32_t e = e0; e < e1; ++e) {
            if (COOP) xyzz_add_quad<F>(acc, buckets[base + (size_t)e * sg.elem_stride]);
            else xyzz_add<F>(acc, buckets[base + (size_t)e * sg.elem_stride]);
        }
    }
#pragma unroll 1
    for (uint32_t d = sg.lanes >> 1; d >= 1; d >>= 1) {      // partner worker = LPG*d lanes further; groups never straddle a wave
        xyzz_t o = shfl_down_xyzz(acc, (int)(LPG * d));
        if (sub + d < sg.lanes) { if (COOP) xyzz_add_quad<F>(acc, o); else xyzz_add<F>(acc, o); }
    }
    if (live && sub == 0 && (gid % LPG) == 0) out[seg] = acc;
}

// K1f on 29-bit limbs (round 5; the multi-MSM form with buckets kept on 29-bit limbs): the same segments, workers and order of additions as msm_segsum_kernel<F, false>,
// the adds are xyzz29_add (ec29.cuh) -- 14 lazy / strict products of 135 multiply-accumulates instead of 14 x (96 + 96 carry adds).  A segment that meets the
// exceptional case of the law (two partial sums equal or opposite) is flagged and recomputed by msm_segsum29_redo_kernel with the complete 8 x 32 law; its
// output slot is left alone here.  Sums leave in the 8 x 32 form the later stages read.
__device__ __forceinline__ bool xyzz29_is_inf(const xyzz29_t &a) { uint32_t o = 0; for (int i = 0; i < L29; ++i) o |= a.zz.v[i]; return o == 0u; }
__device__ __forceinline__ xyzz29_t shfl_down_xyzz29(const xyzz29_t &a, int d) {
    xyzz29_t r;
#pragma unroll
    for (int i = 0; i < L29; ++i) { r.x.v[i] = (uint32_t)__shfl_down((int)a.x.v[i], d, 64); r.y.v[i] = (uint32_t)__shfl_down((int)a.y.v[i], d, 64);
                                    r.zz.v[i] = (uint32_t)__shfl_down((int)a.zz.v[i], d, 64); r.zzz.v[i] = (uint32_t)__shfl_down((int)a.zzz.v[i], d, 64); }
    return r;
}
template <int F>
__global__ void __launch_bounds__(256)
msm_segsum29_kernel(uint32_t nb_per_set, SegSum rows, SegSum cols, const xyzz29_t *__restrict__ buckets29, fe_t one, xyzz_t *__restrict__ out_rows, xyzz_t *__restrict__ out_cols,
                    uint32_t *__restrict__ seg_bad /* rows.nseg + cols.nseg words, zeroed by the host */) { mb_wave_prio<1>();
#if defined(__HIP_DEVICE_COMPILE__)
    const bool is_col = blockIdx.y != 0;
    const SegSum sg = is_col ? cols : rows;
    xyzz_t *__restrict__ out = is_col ? out_cols : out_rows;
    uint32_t *__restrict__ bad_out = seg_bad + (is_col ? rows.nseg : 0u);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t seg = gid / sg.lanes, sub = gid % sg.lanes;
    const bool live = seg < sg.nseg;
    xyzz29_t acc; bool inf = true; uint32_t bad = 0;
    acc.x = fe29_zero(); acc.y = acc.x; acc.zz = acc.x; acc.zzz = acc.x;
    if (live) {
        const size_t base = (size_t)(seg / sg.per_set) * nb_per_set + (size_t)(seg % sg.per_set) * sg.seg_stride;
        const uint32_t per_lane = (sg.len + sg.lanes - 1) / sg.lanes;
        const uint32_t e0 = sub * per_lane, e1 = min(e0 + per_lane, sg.len);
#pragma unroll 1
        for (uint32_t e = e0; e < e1 && !bad; ++e) {
            const xyzz29_t v = buckets29[base + (size_t)e * sg.elem_stride];
            if (xyzz29_is_inf(v)) continue;
            if (inf) { acc = v; inf = false; }
            else if (!xyzz29_add<F>(acc, v)) bad = 1;
        }
    }
#pragma unroll 1
    for (uint32_t d = sg.lanes >> 1; d >= 1; d >>= 1) {          // partner worker = d lanes further; groups never straddle a wave
        const xyzz29_t o = shfl_down_xyzz29(acc, (int)d);
        const uint32_t o_inf = (uint32_t)__shfl_down((int)(inf ? 1u : 0u), (int)d, 64), o_bad = (uint32_t)__shfl_down((int)bad, (int)d, 64);
        if (sub + d < sg.lanes) {
            bad |= o_bad;
            if (!bad && !o_inf) { if (inf) { acc = o; inf = false; } else if (!xyzz29_add<F>(acc, o)) bad = 1; }
        }
    }
    if (live && sub == 0) { if (bad) bad_out[seg] = 1u; else out[seg] = xyzz29_leave<F>(acc, inf, one); }
#endif
}
// the segments msm_segsum29_kernel flagged (none on SRS points with honest scalars): one lane per segment, every bucket taken back to the 8 x 32 form and summed with the
// complete law.  Launched over all segments; a lane whose flag is clear leaves at once.
template <int F>
__global__ void __launch_bounds__(64)
msm_segsum29_redo_kernel(uint32_t nb_per_set, SegSum rows, SegSum cols, const xyzz29_t *__restrict__ buckets29, fe_t one, xyzz_t *__restrict__ out_rows, xyzz_t *__restrict__ out_cols,
                         const uint32_t *__restrict__ seg_bad) { mb_wave_prio<1>();
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows.nseg + cols.nseg || !seg_bad[gid]) return;
    const bool is_col = gid >= rows.nseg;
    const SegSum sg = is_col ? cols : rows;
    const uint32_t seg = is_col ? gid - rows.nseg : gid;
    const size_t base = (size_t)(seg / sg.per_set) * nb_per_set + (size_t)(seg % sg.per_set) * sg.seg_stride;
    xyzz_t acc = xyzz_inf();
    for (uint32_t e = 0; e < sg.len; ++e) {
        const xyzz29_t v = buckets29[base + (size_t)e * sg.elem_stride];
        xyzz_add<F>(acc, xyzz29_leave<F>(v, xyzz29_is_inf(v), one));
    }
    (is_col ? out_cols : out_rows)[seg] = acc;
#endif
}

// quad-replicated wave collectives: every value lives on the 4 lanes of a quad (16 values per wave), adds are
// lane-cooperative.  quad 0 gets  sum_q v_q  (in `sum`) and  sum_q q * v_q  (returned), q < width <= 16.
template <int F> __device__ __forceinline__ xyzz_t quadwave_weighted_sum(xyzz_t v, xyzz_t &sum, int width) {
    const int q = (threadIdx.x & 63) >> 2;
#pragma unroll 1
    for (int d = 1; d < width; d <<= 1) {                    // suffix scan over quads
        xyzz_t o = shfl_down_xyzz(v, 4 * d);
        if (q + d < width) xyzz_add_quad<F>(v, o);
    }
    sum = v;
    if (q == 0) v = xyzz_inf();
    return quadwave_sum<F>(v, width);
}

// K1g: weighted sums over groups of 16 consecutive rows / columns.  Block (g, set), one wave:
//   g <  Gr : rows  16g .. 16g+15  ->  (S, W) = (sum_i Row, sum_i i*Row)      g >= Gr : the same for columns
template <int F>
__global__ void __launch_bounds__(64)
msm_wsum16_kernel(uint32_t R, uint32_t C, uint32_t Gr, const xyzz_t *__restrict__ rows, const xyzz_t *__restrict__ cols,
                  xyzz_t *__restrict__ out_s, xyzz_t *__restrict__ out_w) { mb_wave_prio<1>();
    const uint32_t g = blockIdx.x, set = blockIdx.y, q = threadIdx.x >> 2;
    const uint32_t Gc = (C + 15) / 16, G = Gr + Gc;
    xyzz_t v = xyzz_inf();
    if (g < Gr) { const uint32_t r = g * 16 + q; if (r < R) v = rows[(size_t)set * R + r]; }
    else { const uint32_t c = (g - Gr) * 16 + q; if (c < C) v = cols[(size_t)set * C + c]; }
    xyzz_t sum;
    xyzz_t ws = quadwave_weighted_sum<F>(v, sum, 16);
    if (threadIdx.x == 0) { out_s[(size_t)set * G + g] = sum; out_w[(size_t)set * G + g] = ws; }
}

// K1g': one block of 4 waves per bucket set combines the group results:
//   sum_r r*Row_r = 16 * sum_j j*S_j + sum_j W_j   (row groups j < Gr <= 16), likewise for the columns (Gc = 8), then
//   set total = Tot + C * sum_r r*Row_r + sum_c c*Col_c,   Tot = sum_j S_j
template <int F>
__global__ void __launch_bounds__(256)
msm_reduce2d_kernel(uint32_t Gr, uint32_t Gc, uint32_t log2C, const xyzz_t *__restrict__ in_s, const xyzz_t *__restrict__ in_w,
                    xyzz_t *__restrict__ set_total, xyzz_t *__restrict__ out_xyzz /* may be null: also the result of problem `set` */) { mb_wave_prio<1>();
    const uint32_t set = blockIdx.x, wave = threadIdx.x >> 6, q = (threadIdx.x & 63) >> 2, G = Gr + Gc;
    __shared__ xyzz_t sh_tot, sh_roww, sh_ww[2], sh_colw;
    const xyzz_t *s = in_s + (size_t)set * G, *w = in_w + (size_t)set * G;
    auto p2 = [](uint32_t n) { int wd = 1; while (wd < (int)n) wd <<= 1; return wd; };
    xyzz_t sw = xyzz_inf(), tot = xyzz_inf();
    if (wave == 0) {                                          // rows: Tot and sum_j j*S_j
        xyzz_t v = (q < Gr) ? s[q] : xyzz_inf();
        sw = quadwave_weighted_sum<F>(v, tot, p2(Gr));
    } else if (wave == 1) {                                   // rows: sum_j W_j
        xyzz_t v = (q < Gr) ? w[q] : xyzz_inf();
        v = quadwave_sum<F>(v, p2(Gr));
        if (threadIdx.x == 64) sh_ww[0] = v;
    } else if (wave == 2) {                                   // columns: sum_j j*S'_j
        xyzz_t v = (q < Gc) ? s[Gr + q] : xyzz_inf();
        xyzz_t dummy;
        sw = quadwave_weighted_sum<F>(v, dummy, p2(Gc));
    } else {                                                  // columns: sum_j W'_j
        xyzz_t v = (q < Gc) ? w[Gr + q] : xyzz_inf();
        v = quadwave_sum<F>(v, p2(Gc));
        if (threadIdx.x == 192) sh_ww[1] = v;
    }
    __syncthreads();
    if ((wave == 0 || wave == 2) && q == 0) {                 // quad 0 of waves 0 and 2: 16 * SW + WW
        xyzz_t t = sw;
        for (int i = 0; i < 4; ++i) t = xyzz_dbl_quad<F>(t);
        xyzz_add_quad<F>(t, sh_ww[wave >> 1]);
        if (wave == 0) { for (uint32_t i = 0; i < log2C; ++i) t = xyzz_dbl_quad<F>(t); }     // * C
        if ((threadIdx.x & 63) == 0) { if (wave == 0) { sh_roww = t; sh_tot = tot; } else sh_colw = t; }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        xyzz_t t = sh_roww; xyzz_add_quad<F>(t, sh_colw); xyzz_add_quad<F>(t, sh_tot);
        if (threadIdx.x == 0) { set_total[set] = t; if (out_xyzz) out_xyzz[set] = t; }
    }
}

// K1h: per problem (one block each): Horner over its bucket sets (variable-base; a fixed-base problem has one set),
// then normalise to affine (Montgomery) + canonical words.
//   out_words[17 m + 0..16) = x||y canonical little-endian words, out_words[17 m + 16] = 1 if infinity; out_xyzz[m].
template <int F>
__global__ void msm_finish_kernel(uint32_t nsets /* per problem */, uint32_t c, const xyzz_t *__restrict__ set_total, fe_t one,
                                  fe_t pm2, xyzz_t *__restrict__ out_xyzz, uint32_t *__restrict__ out_words) { mb_wave_prio<1>();
    if (threadIdx.x >= 4) return;                              // one quad: lane-cooperative doublings / adds
    set_total += (size_t)blockIdx.x * nsets;
    if (out_xyzz) out_xyzz += blockIdx.x;
    if (out_words) out_words += (size_t)blockIdx.x * 17;
    xyzz_t t = set_total[nsets - 1];
    for (int w = (int)nsets - 2; w >= 0; --w) {
        for (uint32_t i = 0; i < c; ++i) t = xyzz_dbl_quad<F>(t);
        xyzz_add_quad<F>(t, set_total[w]);
    }
    if (threadIdx.x != 0) return;
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
