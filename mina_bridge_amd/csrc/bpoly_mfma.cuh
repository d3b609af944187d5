// bpoly_mfma.cuh -- the batch fold of the IPA challenge polynomials as an int8 MFMA field-GEMM (gfx950).
//
// What it replaces: `bpoly_fold_kernel` (sponge.cuh) for large batches.  The fold
//     S[hi][lo] = sum_b H_b[hi] * L_b[lo]            (H, L: the high / low half tables of proof b, Montgomery form)
// is a dense contraction over the batch -- an (nh x B) by (B x nl) matrix product over the field, B = 8192 proofs per call against
// nh x nl = 2^15 outputs: 2.7e8 field multiply-accumulates, 8 % of the VALU instructions of a whole Proof-of-State step
// (profiles/r02c).  It is the one GEMM-shaped piece of the path, so it goes to the matrix cores:
//   * every table entry is written as 32 balanced base-256 digits d_i in [-128, 127] (exact: sum d_i 256^i = the 255-bit integer),
//     one int8 plane per digit, the batch index contiguous (the MFMA's K dimension);
//   * C'[(hi, a)][(lo, c)] = sum_b Hd[(hi, a)][b] * Ld[(lo, c)][b] is one int8 GEMM with M' = 32 nh, N' = 32 nl, K = B
//     (`v_mfma_i32_32x32x32_i8`; |C'| <= B * 2^14 < 2^31 for B < 2^17);
//   * a 32 x 32 accumulator tile is exactly the digit-by-digit product block of one (hi, lo) pair: its 63 anti-diagonal sums
//     (a + c = k) are the base-256 columns of the 521-bit integer T = sum_b H_b[hi] L_b[lo]; the epilogue adds them into 64-bit bins
//     in LDS and writes 63 sums per output instead of the 1024 products;
//   * a small kernel carries the columns into limbs and reduces:  T / 2^256 mod p = t0 / 2^256 + t1 + t2 2^256  (4 Montgomery products).
// Bit-exact by construction (integers all the way); parity vs the VALU kernel and the CPU oracle: tests/test_gpu_sponge_ipa.py.
#pragma once
#include "sponge.cuh"

namespace mb {

typedef int bp_v4i __attribute__((ext_vector_type(4)));
typedef int bp_v16i __attribute__((ext_vector_type(16)));

static constexpr uint32_t BPM_DIGITS = 32, BPM_COLS = 64;     // 63 anti-diagonals, padded to 64 bins per output
static constexpr uint32_t BPM_KALIGN = 128;                   // the batch dimension of the digit planes is padded to this (one LDS K tile)

// table entry e of proof b (sponge.cuh `bpoly_tables_kernel`) as balanced digits.  Thread mapping: b fastest, so that the 64 lanes of a
// wave write 64 consecutive bytes of a digit plane.  planes: Ld [nl * 32][kpad], Hd [nh * 32][kpad]; columns b >= batch stay zero.
template <int F>
__global__ void __launch_bounds__(256)
bpoly_tables_digits_kernel(BpolyShape sh, uint32_t kpad, FieldK fk, const uint32_t *__restrict__ chals, const uint32_t *__restrict__ weights,
                           int8_t *__restrict__ Ld, int8_t *__restrict__ Hd) { mb_wave_prio<1>();
    const uint32_t nl = 1u << sh.lb, nh = 1u << sh.hb;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)sh.batch * (nl + nh)) return;
    const uint32_t b = (uint32_t)(gid % sh.batch), e = (uint32_t)(gid / sh.batch);
    const uint32_t *cb = chals + (size_t)b * sh.k * 8;
    fe_t acc; uint32_t bits, base;
    if (e < nl) { bits = e; base = 0; acc = fk.one; }
    else {
        bits = e - nl; base = sh.lb;
        if (weights) { fe_t w; for (int i = 0; i < 8; ++i) w.v[i] = weights[(size_t)b * 8 + i]; acc = fe_to_mont<F>(w, fk.r2); }
        else acc = fk.one;
    }
    for (uint32_t q = 0; bits; ++q, bits >>= 1) {
        if (!(bits & 1u)) continue;
        fe_t c; const uint32_t *cp = cb + (size_t)(sh.k - 1 - (base + q)) * 8;
        for (int i = 0; i < 8; ++i) c.v[i] = cp[i];
        acc = fe_mul<F>(acc, fe_to_mont<F>(c, fk.r2));
    }
    int8_t *plane = e < nl ? Ld + (size_t)e * BPM_DIGITS * kpad : Hd + (size_t)(e - nl) * BPM_DIGITS * kpad;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const uint32_t d = ((acc.v[i >> 2] >> (8 * (i & 3))) & 0xffu) + carry;      // 0..256
        carry = d >= 128u ? 1u : 0u;
        plane[(size_t)i * kpad + b] = (int8_t)(d - (carry << 8));                    // the top byte is <= 0x40: no carry out
    }
}

// The same planes, EIGHT table entries per lane (round 5): the entries that differ only in their three low index bits share the product over the high bits, and the
// eight combinations of the low three cost seven products -- (2.5 + 3) conversions + 2.5 + 7 products per 8 entries, 1.9 per entry against ~8 when every entry
// multiplied out all its set bits (2350 -> ~650 instructions per entry: these two launches were 0.54 G of a Proof-of-State step's 30.5 G wave-instructions).
// Same values (canonical Montgomery products are order-independent), same plane layout, b fastest.  Needs lb, hb >= 3.
template <int F>
__device__ __forceinline__ void bpoly_emit_digits(const fe_t &acc, int8_t *__restrict__ plane, uint32_t kpad, uint32_t b) {
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const uint32_t d = ((acc.v[i >> 2] >> (8 * (i & 3))) & 0xffu) + carry;      // 0..256
        carry = d >= 128u ? 1u : 0u;
        plane[(size_t)i * kpad + b] = (int8_t)(d - (carry << 8));                    // the top byte is <= 0x40: no carry out
    }
}
template <int F>
__global__ void __launch_bounds__(256)
bpoly_tables_digits8_kernel(BpolyShape sh, uint32_t kpad, FieldK fk, const uint32_t *__restrict__ chals, const uint32_t *__restrict__ weights,
                            int8_t *__restrict__ Ld, int8_t *__restrict__ Hd) { mb_wave_prio<1>();
    const uint32_t nl8 = 1u << (sh.lb - 3), nh8 = 1u << (sh.hb - 3);
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)sh.batch * (nl8 + nh8)) return;
    const uint32_t b = (uint32_t)(gid % sh.batch), g = (uint32_t)(gid / sh.batch);
    const uint32_t *cb = chals + (size_t)b * sh.k * 8;
    const bool low = g < nl8;
    const uint32_t hi_bits = low ? g : g - nl8, base = low ? 0u : sh.lb, nbits = low ? sh.lb : sh.hb;
    auto chal = [&](uint32_t q) {                                  // the challenge of index bit base + q, Montgomery
        fe_t c; const uint32_t *cp = cb + (size_t)(sh.k - 1 - (base + q)) * 8;
        for (int i = 0; i < 8; ++i) c.v[i] = cp[i];
        return fe_to_mont<F>(c, fk.r2);
    };
    fe_t t0 = fk.one;
    if (!low && weights) { fe_t w; for (int i = 0; i < 8; ++i) w.v[i] = weights[(size_t)b * 8 + i]; t0 = fe_to_mont<F>(w, fk.r2); }
#pragma unroll 1
    for (uint32_t q = 3; q < nbits; ++q) if ((hi_bits >> (q - 3)) & 1u) t0 = fe_mul<F>(t0, chal(q));
    int8_t *plane = (low ? Ld : Hd) + (size_t)hi_bits * 8 * BPM_DIGITS * kpad;      // entry e = hi_bits * 8 + j: plane + j * 32 * kpad
    const size_t estride = (size_t)BPM_DIGITS * kpad;
    const fe_t c0 = chal(0), c1 = chal(1);
    const fe_t t1 = fe_mul<F>(t0, c0), t2 = fe_mul<F>(t0, c1), t3 = fe_mul<F>(t1, c1);
    bpoly_emit_digits<F>(t0, plane, kpad, b); bpoly_emit_digits<F>(t1, plane + estride, kpad, b);
    bpoly_emit_digits<F>(t2, plane + 2 * estride, kpad, b); bpoly_emit_digits<F>(t3, plane + 3 * estride, kpad, b);
    const fe_t c2 = chal(2);
    bpoly_emit_digits<F>(fe_mul<F>(t0, c2), plane + 4 * estride, kpad, b); bpoly_emit_digits<F>(fe_mul<F>(t1, c2), plane + 5 * estride, kpad, b);
    bpoly_emit_digits<F>(fe_mul<F>(t2, c2), plane + 6 * estride, kpad, b); bpoly_emit_digits<F>(fe_mul<F>(t3, c2), plane + 7 * estride, kpad, b);
}

// C' = A B^T over int8 planes with K contiguous; block = 4 waves = 128 x 128 of C' (wave: 64 x 64 = 2 x 2 MFMA tiles = four
// (hi, lo) pairs).  K runs in tiles of 128 bytes staged through LDS, double-buffered: the 256 threads fetch the two 128 x 128-byte tiles
// with row-contiguous 16-byte loads (8 lanes = one 128-byte line of a plane row; the direct-from-L2 form made every load instruction of
// a wave touch 32 - 64 lines and ran at 13 % of the int8 peak), rows padded to 144 bytes in LDS so that the 16-byte fragment reads of 16
// lanes fall into 16 distinct bank groups.  Output: the 63 anti-diagonal sums of every 32 x 32 tile, int64,
// colsum[(mtile * ntiles + ntile) * 64 + k].
static constexpr uint32_t BPM_KT = 128, BPM_ROW = 144;            // K bytes per LDS tile, padded row pitch
__global__ void __launch_bounds__(256)
bpoly_field_gemm_kernel(uint32_t mtiles, uint32_t ntiles, uint32_t kpad, const int8_t *__restrict__ A, const int8_t *__restrict__ B,
                        unsigned long long *__restrict__ colsum) { mb_wave_prio<1>();
    __shared__ __attribute__((aligned(16))) int8_t tiles[2][2][128 * BPM_ROW];   // stage, A | B, 128 rows
    __shared__ unsigned long long bins[4][4][BPM_COLS];           // wave, tile of the wave, column
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t bmn = (mtiles + 3) / 4, bm = blockIdx.x % bmn, bn = blockIdx.x / bmn;
    const uint32_t mt0 = bm * 4 + (wave & 1u) * 2, nt0 = bn * 4 + (wave >> 1) * 2;       // first of this wave's 2 x 2 tiles
    for (uint32_t i = lane; i < 4 * BPM_COLS; i += 64) (&bins[wave][0][0])[i] = 0ull;
    bp_v16i acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    // loader: thread t fetches rows t/8 + 32 i (i < 4) of the block's 128 A rows and 128 B rows, bytes (t % 8) * 16 .. + 16 of the K tile.
    // Rows past the matrix (nh or nl < 4 tiles) read row 0: their products land in tiles that are never written out.
    const uint32_t lrow = tid >> 3, lcol = (tid & 7u) * 16;
    const int8_t *ga[4], *gb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t ra = bm * 128 + lrow + 32 * i, rb = bn * 128 + lrow + 32 * i;
        ga[i] = A + (size_t)(ra < mtiles * 32 ? ra : 0) * kpad + lcol;
        gb[i] = B + (size_t)(rb < ntiles * 32 ? rb : 0) * kpad + lcol;
    }
    bp_v4i ra4[4], rb4[4];
    auto fetch = [&](uint32_t k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra4[i] = *reinterpret_cast<const bp_v4i *>(ga[i] + k0); rb4[i] = *reinterpret_cast<const bp_v4i *>(gb[i] + k0); }
    };
    auto stash = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<bp_v4i *>(&tiles[stage][0][(lrow + 32 * i) * BPM_ROW + lcol]) = ra4[i];
            *reinterpret_cast<bp_v4i *>(&tiles[stage][1][(lrow + 32 * i) * BPM_ROW + lcol]) = rb4[i];
        }
    };
    // fragment of lane l: row (l & 31) of the MFMA tile, 16 consecutive k at (l >> 5) * 16 of each 32-byte k-step
    const uint32_t fa = ((wave & 1u) * 64 + (lane & 31u)) * BPM_ROW + (lane >> 5) * 16, fb = ((wave >> 1) * 64 + (lane & 31u)) * BPM_ROW + (lane >> 5) * 16;
    fetch(0); stash(0);
    __syncthreads();
    int stage = 0;
    for (uint32_t k0 = 0; k0 < kpad; k0 += BPM_KT) {
        const bool more = k0 + BPM_KT < kpad;
        if (more) fetch(k0 + BPM_KT);                              // global loads of the next tile fly during this tile's MFMAs
        const int8_t *ta = tiles[stage][0], *tb = tiles[stage][1];
#pragma unroll
        for (uint32_t ks = 0; ks < BPM_KT; ks += 32) {
            bp_v4i a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { a[i] = *reinterpret_cast<const bp_v4i *>(ta + fa + i * 32 * BPM_ROW + ks); b[i] = *reinterpret_cast<const bp_v4i *>(tb + fb + i * 32 * BPM_ROW + ks); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(stage ^ 1);
        __syncthreads();
        stage ^= 1;
    }
    const bool m_ok[2] = {mt0 < mtiles, mt0 + 1 < mtiles}, n_ok[2] = {nt0 < ntiles, nt0 + 1 < ntiles};
    // C/D layout: col = lane & 31 (the B row: digit c of lo), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (digit a of hi); bin a + c
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31u;
                atomicAdd(&bins[wave][i * 2 + j][row + col], (unsigned long long)(long long)acc[i][j][r]);
            }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (m_ok[i] && n_ok[j]) colsum[((size_t)(mt0 + i) * ntiles + (nt0 + j)) * BPM_COLS + lane] = bins[wave][i * 2 + j][lane];
}

// one thread per output: 63 signed base-256 columns -> 17 limbs -> T / 2^256 mod p, canonical words at out[(hi << lb) + lo]
template <int F>
__global__ void __launch_bounds__(256)
bpoly_colsum_reduce_kernel(uint32_t nh, uint32_t nl, uint32_t lb, FieldK fk, const unsigned long long *__restrict__ colsum, uint32_t *__restrict__ out_words) { mb_wave_prio<1>();
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nh * nl) return;
    const uint32_t hi = gid / nl, lo = gid % nl;
    const long long *cs = reinterpret_cast<const long long *>(colsum) + (size_t)gid * BPM_COLS;      // tile (hi, lo) = tile index gid
    uint32_t limb[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) limb[i] = 0;
    long long carry = 0;
#pragma unroll 1
    for (int k = 0; k < 68; ++k) {                                // 63 columns, then the carry runs out (T < 2^528)
        const long long v = (k < 63 ? cs[k] : 0ll) + carry;
        limb[k >> 2] |= (uint32_t)(v & 0xff) << (8 * (k & 3));
        carry = v >> 8;                                            // arithmetic: the columns are signed, T is not
    }
    fe_t t0, t1, t2 = fe_zero(), one = fe_zero();
    one.v[0] = 1u;
#pragma unroll
    for (int i = 0; i < 8; ++i) { t0.v[i] = limb[i]; t1.v[i] = limb[8 + i]; }
    t2.v[0] = limb[16];
    // T / R = t0 / R + t1 + t2 R  (mod p):  mont(x, 1) = x / R,  mont(x, R^2) = x R
    fe_t r = fe_mul<F>(t0, one);
    r = fe_add<F>(r, fe_mul<F>(fe_mul<F>(t1, fk.r2), one));
    r = fe_add<F>(r, fe_mul<F>(t2, fk.r2));
    r = fe_from_mont<F>(r);
    uint4 *o = reinterpret_cast<uint4 *>(out_words + ((size_t)(hi << lb) + lo) * 8);
    o[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    o[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

}  // namespace mb
