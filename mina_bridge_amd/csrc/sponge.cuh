// sponge.cuh -- K2 (IPA challenge polynomial) and K3 (Poseidon sponge, endo challenges) for gfx950.
//
// Replaces (pins core/Cargo.toml:14,16):
//   poly-commitment `b_poly`, `b_poly_coefficients` and the `sg_rand_base_i * s` fold inside `SRS::verify`;
//   mina-poseidon `ArithmeticSponge` with `PlonkSpongeConstantsKimchi`
//     (width 3, rate 2, 55 full rounds, sbox x^7, no initial ARK; round = sbox -> MDS -> + rc);
//   kimchi `ScalarChallenge::to_field`.
#pragma once
#include "groupmap.cuh"
#include "fp29.cuh"

namespace mb {

// ---------------------------------------------------------------- K2
// b_poly_coefficients(c)[j] = prod_{bit q of j set} c[k-1-q].  Split j = hi * 2^lb + lo:
//   s[j] = H[hi] * L[lo],  L[lo] = prod over low bits, H[hi] = weight * prod over high bits.
// Tables live in HBM: per proof 2^lb + 2^hb entries (k=16: 512 entries = 16 KiB).
struct BpolyShape { uint32_t k, lb, hb, batch; };

template <int F>
__global__ void bpoly_tables_kernel(BpolyShape sh, FieldK fk, const uint32_t *__restrict__ chals /* batch*k*8 canonical */,
                                    const uint32_t *__restrict__ weights /* batch*8 canonical, may be null */,
                                    fe_t *__restrict__ ltab, fe_t *__restrict__ htab) { mb_wave_prio<1>();
    const uint32_t nl = 1u << sh.lb, nh = 1u << sh.hb, per = nl + nh;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)sh.batch * per) return;
    uint32_t b = (uint32_t)(gid / per), e = (uint32_t)(gid % per);
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
    if (e < nl) ltab[(size_t)b * nl + e] = acc; else htab[(size_t)b * nh + (e - nl)] = acc;
}

// One lane per `lo`; each block covers BP_HT consecutive `hi` values and a slice of the batch.
//   partial[slice][j] = sum_{b in slice} H_b[hi] * L_b[lo]      (Montgomery)
static constexpr int BP_HT = 4;
template <int F>
__global__ void __launch_bounds__(256)
bpoly_fold_kernel(BpolyShape sh, uint32_t slices, const fe_t *__restrict__ ltab, const fe_t *__restrict__ htab,
                  fe_t *__restrict__ partial) { mb_wave_prio<1>();
    const uint32_t nl = 1u << sh.lb, nh = 1u << sh.hb;
    const uint32_t lo_blocks = (nl + blockDim.x - 1) / blockDim.x;
    const uint32_t hi_tiles = (nh + BP_HT - 1) / BP_HT;
    uint32_t bid = blockIdx.x;
    const uint32_t lo_blk = bid % lo_blocks; bid /= lo_blocks;
    const uint32_t tile = bid % hi_tiles; const uint32_t slice = bid / hi_tiles;
    const uint32_t lo = lo_blk * blockDim.x + threadIdx.x;
    if (lo >= nl) return;
    const uint32_t b0 = (uint32_t)((uint64_t)sh.batch * slice / slices);
    const uint32_t b1 = (uint32_t)((uint64_t)sh.batch * (slice + 1) / slices);
    fe_t acc[BP_HT];
#pragma unroll
    for (int t = 0; t < BP_HT; ++t) acc[t] = fe_zero();
    uint32_t b = b0;
    for (; b + 3 <= b1; b += 3) {                          // three proofs per step: one reduction per dot product
        const fe_t l0 = ltab[(size_t)b * nl + lo], l1 = ltab[(size_t)(b + 1) * nl + lo], l2 = ltab[(size_t)(b + 2) * nl + lo];
#pragma unroll
        for (int t = 0; t < BP_HT; ++t) {
            uint32_t hi = tile * BP_HT + t;
            if (hi < nh) acc[t] = fe_add<F>(acc[t], fe_dot3<F>(l0, htab[(size_t)b * nh + hi], l1, htab[(size_t)(b + 1) * nh + hi],
                                                               l2, htab[(size_t)(b + 2) * nh + hi]));
        }
    }
    for (; b < b1; ++b) {
        const fe_t l = ltab[(size_t)b * nl + lo];
#pragma unroll
        for (int t = 0; t < BP_HT; ++t) {
            uint32_t hi = tile * BP_HT + t;
            if (hi < nh) acc[t] = fe_add<F>(acc[t], fe_mul<F>(l, htab[(size_t)b * nh + hi]));
        }
    }
#pragma unroll
    for (int t = 0; t < BP_HT; ++t) {
        uint32_t hi = tile * BP_HT + t;
        if (hi < nh) partial[(size_t)slice * ((size_t)1 << sh.k) + ((size_t)hi << sh.lb) + lo] = acc[t];
    }
}

// out[j] (canonical words) = sum over slices of partial[slice][j]
template <int F>
__global__ void bpoly_finish_kernel(uint32_t n, uint32_t slices, const fe_t *__restrict__ partial, uint32_t *__restrict__ out_words) { mb_wave_prio<1>();
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    fe_t acc = partial[j];
    for (uint32_t s = 1; s < slices; ++s) acc = fe_add<F>(acc, partial[(size_t)s * n + j]);
    acc = fe_from_mont<F>(acc);
    uint4 *o = reinterpret_cast<uint4 *>(out_words + (size_t)j * 8);
    o[0] = make_uint4(acc.v[0], acc.v[1], acc.v[2], acc.v[3]);
    o[1] = make_uint4(acc.v[4], acc.v[5], acc.v[6], acc.v[7]);
}

// b_poly(chals, x) = prod_i (1 + chals[i] * x^(2^(k-1-i)))
template <int F>
__global__ void bpoly_eval_kernel(uint32_t k, uint32_t npoints, FieldK fk, const uint32_t *__restrict__ chals,
                                  const uint32_t *__restrict__ xs, uint32_t *__restrict__ out_words) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npoints) return;
    fe_t x; for (int i = 0; i < 8; ++i) x.v[i] = xs[(size_t)p * 8 + i];
    fe_t pw = fe_to_mont<F>(x, fk.r2), r = fk.one;
    for (int i = (int)k - 1; i >= 0; --i) {            // pw = x^(2^(k-1-i))
        fe_t c; for (int q = 0; q < 8; ++q) c.v[q] = chals[(size_t)i * 8 + q];
        r = fe_mul<F>(r, fe_add<F>(fk.one, fe_mul<F>(fe_to_mont<F>(c, fk.r2), pw)));
        pw = fe_sqr<F>(pw);
    }
    r = fe_from_mont<F>(r);
    for (int i = 0; i < 8; ++i) out_words[(size_t)p * 8 + i] = r.v[i];
}

// ---------------------------------------------------------------- K3
struct PoseidonParams { fe_t mds[3][3]; fe_t rc[55][3]; };     // Montgomery, in HBM (read with uniform addresses)
// the same constants for the 3-lane permutation's 29-bit-limb arithmetic (fp29.cuh): Montgomery with R = 2^261, plus the two re-basing
// constants.  Lives in the same device buffer right behind PoseidonParams (api_sponge.hip: mina_poseidon_set_params).
struct PoseidonParams29 { fe29_t mds[3][3]; fe29_t rc[55][3]; fe29_t enter /* 2^266 mod p */, leave /* 2^256 mod p */;
                          fe29_t rc2[55][3];      // the round constants times 2^261 once more (rc 2^522 mod p): added BEFORE the reduction of the 3-lane form's MDS dot product
                          fe29_t zero;            // 0: what the lanes that do not add the round constant read in its place (8- and 16-lane forms)
                          fe29_t absorb;          // 2^522 mod p: (canonical words)(2^522) / 2^261 = x 2^261 -- a field absorbed without leaving the 29-bit form (pstate_hash_kernel<., 3>)
                          uint32_t pad[1]; };     // size: a multiple of 16
static_assert(sizeof(PoseidonParams29) % 16 == 0, "PoseidonParams29 is read with 16-byte loads");
__host__ __device__ static inline const PoseidonParams29 *pparams29_of(const PoseidonParams *pp) { return reinterpret_cast<const PoseidonParams29 *>(pp + 1); }

template <int F>
__device__ __forceinline__ void poseidon_permute(fe_t s[3], const PoseidonParams *__restrict__ pp) {
#pragma unroll 1
    for (int r = 0; r < 55; ++r) {
        fe_t t[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            fe_t x2 = fe_sqr<F>(s[i]);
            fe_t x4 = fe_sqr<F>(x2);
            t[i] = fe_mul<F>(fe_mul<F>(x4, x2), s[i]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            // MDS row as one dot product: a single Montgomery reduction for the three products
            s[i] = fe_add<F>(fe_dot3<F>(pp->mds[i][0], t[0], pp->mds[i][1], t[1], pp->mds[i][2], t[2]), pp->rc[r][i]);
        }
    }
}

template <int F>
__global__ void __launch_bounds__(256)
poseidon_permute_kernel(uint32_t n, FieldK fk, const PoseidonParams *__restrict__ pp, uint32_t *__restrict__ states) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_t s[3];
    for (int j = 0; j < 3; ++j) { fe_t w; for (int q = 0; q < 8; ++q) w.v[q] = states[(size_t)i * 24 + j * 8 + q]; s[j] = fe_to_mont<F>(w, fk.r2); }
    poseidon_permute<F>(s, pp);
    for (int j = 0; j < 3; ++j) { fe_t w = fe_from_mont<F>(s[j]); for (int q = 0; q < 8; ++q) states[(size_t)i * 24 + j * 8 + q] = w.v[q]; }
}

// n independent sponges: absorb len elements, squeeze one (rate 2)
template <int F>
__global__ void __launch_bounds__(256)
poseidon_hash_kernel(uint32_t n, uint32_t len, FieldK fk, const PoseidonParams *__restrict__ pp,
                     const uint32_t *__restrict__ inputs, uint32_t *__restrict__ out_words) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_t s[3] = {fe_zero(), fe_zero(), fe_zero()};
    uint32_t count = 0;
    for (uint32_t e = 0; e < len; ++e) {
        fe_t w; for (int q = 0; q < 8; ++q) w.v[q] = inputs[((size_t)i * len + e) * 8 + q];
        w = fe_to_mont<F>(w, fk.r2);
        if (count == 2) { poseidon_permute<F>(s, pp); count = 0; }
        s[count] = fe_add<F>(s[count], w); ++count;
    }
    poseidon_permute<F>(s, pp);
    fe_t w = fe_from_mont<F>(s[0]);
    for (int q = 0; q < 8; ++q) out_words[(size_t)i * 8 + q] = w.v[q];
}

// ---------------------------------------------------------------- K3, lane-cooperative form
// Four lanes (one DPP quad) per sponge: lane q < 3 owns state element s_q (lane 3 idles).  Per round every lane does
// its own x^7 (4 products), the three results are exchanged with quad_perm DPP moves (no LDS), and lane q computes
// row q of the MDS product (3 products): 7 dependent products per round instead of 21 -- a 3x shorter critical path
// for the sequential sponge work (Fiat-Shamir transcripts, Merkle paths) at batch sizes that cannot fill the chip.
#if defined(__HIPCC__)
template <int F>
__device__ __forceinline__ void poseidon_permute_quad(fe_t &s, const PoseidonParams *__restrict__ pp) {
    const int q = threadIdx.x & 3, qq = q < 3 ? q : 2;
    const fe_t m0 = pp->mds[qq][0], m1 = pp->mds[qq][1], m2 = pp->mds[qq][2];
#pragma unroll 1
    for (int r = 0; r < 55; ++r) {
        // lazy round (fp.cuh "lazy forms"): s < 2p in, every product unreduced, one conditional subtraction of 2p out
        fe_t x2 = fe_mul_nr<F>(s, s);
        fe_t x4 = fe_mul_nr<F>(x2, x2);
        fe_t t = fe_mul_nr<F>(fe_mul_nr<F>(x4, x2), s);
        fe_t t0 = quad_bcast<0>(t), t1 = quad_bcast<1>(t), t2 = quad_bcast<2>(t);
        s = fe_add_csub2p<F>(fe_dot3_nr<F>(m0, t0, m1, t1, m2, t2), pp->rc[r][qq]);
    }
    s = fe_cond_sub_p<F>(s);
}

// Eight lanes per sponge (half a DPP row): the pair of lanes (2e, 2e+1) owns state element e (lanes 6, 7 mirror e = 2).
// x^7 takes 3 dependent products instead of 4 (x2; then x4 on the even lane and x3 on the odd lane, swapped inside the
// pair; then x4 * x3) and an MDS row 1.65 instead of 2.3 (even lane: a two-term dot product, odd lane: the third product,
// as a dot product with a zero term so that both lanes run the same instructions; the halves are swapped and added):
// 4.65 product latencies per round against 6.3 for the quad form, bit-identical state.  Costs 1.5x the issue slots per
// permutation, so only for batches that leave the chip latency-bound (host picks the form).
__device__ __forceinline__ fe_t pair_swap(const fe_t &a) {
    fe_t r = a;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], 0xB1, 0xf, 0xf, true);   // quad_perm:[1,0,3,2]
#endif
    return r;
}
template <int K> __device__ __forceinline__ fe_t oct_bcast(const fe_t &a) {            // lane K of every octet -> its 8 lanes
    fe_t r;
    const int src = (int)((threadIdx.x & 63u) & ~7u) | K;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__shfl((int)a.v[i], src, 64);
    return r;
}
template <int F>
__device__ __forceinline__ void poseidon_permute_oct(fe_t &s, const PoseidonParams *__restrict__ pp) {
    // rounds in the carry-free 29-bit-limb representation (fp29.cuh), as the 3-lane form below: the dependent chain of a product is 135
    // multiply-accumulates instead of 96 multiply + carry-add pairs, and nothing is conditionally subtracted on the way
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t o = threadIdx.x & 7u, e = (o >> 1) < 3 ? (o >> 1) : 2;
    const bool odd = o & 1u;
    const PoseidonParams29 *__restrict__ q = pparams29_of(pp);
    fe29_t zero29;
#pragma unroll
    for (int i = 0; i < L29; ++i) zero29.v[i] = 0u;
    const fe29_t ma = odd ? q->mds[e][2] : q->mds[e][0], mb = odd ? zero29 : q->mds[e][1];
    auto swap29 = [](const fe29_t &a) { fe29_t r;
#pragma unroll
        for (int i = 0; i < L29; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], 0xB1, 0xf, 0xf, true);   // quad_perm:[1,0,3,2]
        return r; };
    auto bcast29 = [](const fe29_t &a, int k) { fe29_t r; const int src = (int)((threadIdx.x & 63u) & ~7u) | k;
#pragma unroll
        for (int i = 0; i < L29; ++i) r.v[i] = (uint32_t)__shfl((int)a.v[i], src, 64);
        return r; };
    fe29_t x = fe29_mul_asm<F>(fe29_from_words(s), q->enter);        // x 2^256 -> x 2^261
    // lazy products (fp29.cuh), as the 3-lane form; the even lane of a pair adds the round constant inside its reduction, the odd lane adds zero.
    // In units of p (tools/fe29_bounds.py; every reduction with signed quotient digits since round 5): x < 4.1, x^2 < 2.2, x^3 / x^4 < 2.1, x^7 < 2.1, each half row < 2.04
    const fe29_t *rcp = odd ? &q->zero : &q->rc2[0][e];
    const size_t rcs = odd ? 0 : 3;                                  // fe29_t elements per round
#pragma unroll 1
    for (int r = 0; r < 55; ++r) {
        const fe29_t x2 = fe29_sqr_sg<F>(x);
        const fe29_t y = fe29_mul_sg<F>(x2, odd ? x : x2);           // even: x^4, odd: x^3
        const fe29_t t = fe29_mul_sg<F>(y, swap29(y));               // x^7 on both lanes of the pair
        const fe29_t t0 = bcast29(t, 0), t1 = bcast29(t, 2), t2 = bcast29(t, 4);
        const fe29_t u = fe29_dot2rc_sg<F>(ma, odd ? t2 : t0, mb, t1, rcp[rcs * r]);   // even: m0 t0 + m1 t1 + rc, odd: m2 t2 (+ 0 * t1)
        x = fe29_add(u, swap29(u));                                  // limbs normalised
    }
    s = fe_cond_sub_p<F>(fe29_to_words(fe29_mul_asm<F>(x, q->leave)));   // x 2^261 -> x 2^256; the strict product: 4.1 p p / 2^261 + p < 1.04 p
#else
    (void)s; (void)pp;
#endif
}
// Three lanes per sponge: 21 sponges per wave64 (lanes 3g, 3g+1, 3g+2 own state elements 0, 1, 2 of sponge g; lane 63 shadows
// group 20).  Same 7 dependent products per round as the quad form, but no idle fourth lane: 63 of 64 lanes work, which is
// what the chip-filling batches want (the quad form caps at 75 % lane use).  The x^7 of the two other lanes arrive by ds_bpermute
// (18 per round; measured on the round alone they cost ~3 %: profiles/r05_clock_power.md).  Bit-identical state.
struct TriPos { uint32_t base, e; };
__device__ __forceinline__ TriPos tri_pos() {
    const uint32_t lane = threadIdx.x & 63u;
    TriPos p; const uint32_t g = lane == 63u ? 20u : lane / 3u; p.base = 3u * g; p.e = lane == 63u ? 0u : lane - 3u * g; return p;
}
__device__ __forceinline__ fe_t tri_bcast(const fe_t &a, uint32_t src_lane) {
    fe_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__shfl((int)a.v[i], (int)src_lane, 64);
    return r;
}
__device__ __forceinline__ fe29_t tri_bcast29(const fe29_t &a, uint32_t src_lane) {
    fe29_t r;
#pragma unroll
    for (int i = 0; i < L29; ++i) r.v[i] = (uint32_t)__shfl((int)a.v[i], (int)src_lane, 64);
    return r;
}
// the 55 rounds of the 3-lane form on a state element that is already in the 29-bit form (x 2^261, below SPONGE29::LANES3_STATE_MILLI_P / 1000 p, limbs normalised) and stays in it
template <int F>
__device__ __forceinline__ void poseidon_rounds_tri(fe29_t &x, const PoseidonParams29 *__restrict__ q, const TriPos &tp) {
#if defined(__HIP_DEVICE_COMPILE__)
    // The row's own term needs no cross-lane move: lane e multiplies ITS x^7 by mds[e][e] and fetches only the two others (18 ds_bpermute per round instead of 27;
    // the column sums of the dot product are the same integers in another order, so the state is bit-identical).  Measured on the round alone (tools/probes/sg_probe
    // --rounds): the 27 moves cost 3.4 - 4.3 % of the round's rate -- two thirds of it waiting and issue, a third the clock they pull down on the power cap.
    const uint32_t en = tp.e == 2u ? 0u : tp.e + 1u, ep = tp.e == 0u ? 2u : tp.e - 1u;
    const fe29_t ms = q->mds[tp.e][tp.e], mn = q->mds[tp.e][en], mp = q->mds[tp.e][ep];
    // Every reduction of a round uses SIGNED quotient digits (fp29.cuh fe29_sqr_sg / fe29_mul_sg / fe29_dot3rc_sg: no instruction per digit, results within (1 p, 2 p] of
    // the exact quotient, limbs normalised; the round constant rides inside the row's reduction).  Values along a round, in units of p, from x < 4.1 (a row plus an absorbed
    // field): x^2 < 2.14, every other power and the row < 2.07 (tools/fe29_bounds.py prove_sponge_rounds: the row's 27 limb products per column reach 0.92 of the signed accumulator).
    fe29_t rc = q->rc2[0][tp.e];
#pragma unroll 1
    for (int r = 0; r < 55; ++r) {
        const fe29_t x2 = fe29_sqr_sg<F>(x);
        const fe29_t x4 = fe29_sqr_sg<F>(x2);
        const fe29_t t = fe29_mul_sg<F>(fe29_mul_sg<F>(x4, x2), x);
        const fe29_t tn = tri_bcast29(t, tp.base + en), tq = tri_bcast29(t, tp.base + ep);
        x = fe29_dot3rc_sg<F>(ms, t, mn, tn, mp, tq, rc);
        rc = q->rc2[r < 54 ? r + 1 : 54][tp.e];                     // the NEXT round's constant, a whole S-box ahead of its use (loaded beside its use it was three exposed load latencies per round)
    }
#else
    (void)x; (void)q; (void)tp;                                      // device-only (the host pass never calls it)
#endif
}
template <int F>
__device__ __forceinline__ void poseidon_permute_tri(fe_t &s, const PoseidonParams *__restrict__ pp) {
    // the rounds run in the carry-free 29-bit-limb representation (fp29.cuh); state in and out as 8 x 32 Montgomery-2^256, canonical
#if defined(__HIP_DEVICE_COMPILE__)
    const TriPos tp = tri_pos();
    const PoseidonParams29 *__restrict__ q = pparams29_of(pp);
    fe29_t x = fe29_mul_asm<F>(fe29_from_words(s), q->enter);        // x 2^256 -> x 2^261
    poseidon_rounds_tri<F>(x, q, tp);
    s = fe_cond_sub_p<F>(fe29_to_words(fe29_mul_asm<F>(x, q->leave)));   // x 2^261 -> x 2^256; the strict product: 4.1 p p / 2^261 + p < 1.04 p: one conditional subtraction
#else
    (void)s; (void)pp;                                               // device-only (the host pass never calls it)
#endif
}
// Sixteen lanes per sponge (4 sponges per wave), for ONE proof or a handful -- the reference's call pattern, where nothing but the length
// of the dependent chain counts.  Quad q of the group holds state element / MDS row q (quad 3 shadows quad 2).  Inside a quad, lanes 0 and 1
// are the x^7 pair of the 8-lane form (lanes 2, 3 shadow them), and lanes 0, 1, 2 multiply the row's three MDS entries by the three x^7 --
// ONE product each, where the 8-lane form's two lanes per row need a two-term dot: a round's chain is 99 + 3 x 135 multiply-accumulates
// instead of 99 + 2 x 135 + 216.  Row sums and the pair exchange are quad_perm DPP moves; the three x^7 travel by ds_bpermute.
template <int F>
__device__ __forceinline__ void poseidon_permute_hex(fe_t &s, const PoseidonParams *__restrict__ pp) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t o = threadIdx.x & 15u, qd = o >> 2, c = o & 3u, e = qd < 3 ? qd : 2u, col = c < 3 ? c : 2u;
    const bool odd = c & 1u;
    const PoseidonParams29 *__restrict__ q = pparams29_of(pp);
    const fe29_t m = q->mds[e][col];
#define MB_QUAD29(NAME, CTRL) auto NAME = [](const fe29_t &a) { fe29_t r; _Pragma("unroll") for (int i = 0; i < L29; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], CTRL, 0xf, 0xf, true); return r; }
    MB_QUAD29(swap29, 0xB1);                                          // quad_perm:[1,0,3,2]
    MB_QUAD29(rot1, 0x09);                                            // quad_perm:[1,2,0,0]: lanes 0..2 read their right neighbour (cyclically), lane 3 as lane 2
    MB_QUAD29(rot2, 0x52);                                            // quad_perm:[2,0,1,1]
#undef MB_QUAD29
    const int src = (int)(((threadIdx.x & 63u) & ~15u) | (col << 2));  // lane 0 of quad `col`: x_col^7
    fe29_t x = fe29_mul_asm<F>(fe29_from_words(s), q->enter);        // x 2^256 -> x 2^261
    // lazy products (fp29.cuh), as the 3-lane form: what counts here is the LENGTH of the dependent chain, and one wave issues an instruction every ~9
    // cycles whatever their dependencies (microbench --dep) -- fewer instructions is the only lever.  Lane 0 of a quad adds the round constant inside its
    // product's reduction (the others add zero); the row's three terms are summed in one carry pass.  In units of p: x < 24.3, x^2 < 12.7, x^3 / x^4 < 10.5,
    // x^7 < 8.9, each term < 8.1
    const fe29_t *rcp = c == 0 ? &q->rc2[0][e] : &q->zero;
    const size_t rcs = c == 0 ? 3 : 0;                               // fe29_t elements per round
#pragma unroll 1
    for (int r = 0; r < 55; ++r) {
        const fe29_t x2 = fe29_sqr_sg<F>(x);
        const fe29_t y = fe29_mul_sg<F>(x2, odd ? x : x2);           // even: x^4, odd: x^3
        const fe29_t t = fe29_mul_sg<F>(y, swap29(y));               // x_e^7 on every lane of quad e
        fe29_t tc;
#pragma unroll
        for (int i = 0; i < L29; ++i) tc.v[i] = (uint32_t)__shfl((int)t.v[i], src, 64);
        const fe29_t pr = fe29_mulrc_sg<F>(m, tc, rcp[rcs * r]);     // mds[e][col] x_col^7 (+ the round constant on lane 0; lane 3 repeats column 2)
        x = fe29_add3(pr, rot1(pr), rot2(pr));                       // the row's three terms, limbs normalised
    }
    s = fe_cond_sub_p<F>(fe29_to_words(fe29_mul_asm<F>(x, q->leave)));   // x 2^261 -> x 2^256; the strict product: 6.1 p p / 2^261 + p < 1.05 p
#else
    (void)s; (void)pp;
#endif
}
// LANES-lane cooperative permutation / element ownership, LANES = 3 (wave-packed triples), 4 (quad), 8 (octet) or 16
template <int F, int LANES> __device__ __forceinline__ void poseidon_permute_coop(fe_t &s, const PoseidonParams *__restrict__ pp) {
    if (LANES == 16) poseidon_permute_hex<F>(s, pp); else if (LANES == 8) poseidon_permute_oct<F>(s, pp); else if (LANES == 3) poseidon_permute_tri<F>(s, pp); else poseidon_permute_quad<F>(s, pp);
}
template <int LANES> __device__ __forceinline__ uint32_t coop_elem() {                 // state element this lane owns
    if (LANES == 3) return tri_pos().e;
    const uint32_t l = threadIdx.x & (LANES - 1), e = LANES == 16 ? (l >> 2) : (LANES == 8 ? (l >> 1) : l);
    return e < 3 ? e : 2;
}
template <int LANES> __device__ __forceinline__ fe_t coop_get(const fe_t &s, int pos) { // element `pos` -> all lanes of the group
    if (LANES == 3) return tri_bcast(s, tri_pos().base + (uint32_t)pos);
    if (LANES == 16) { fe_t r; const int src = (int)(((threadIdx.x & 63u) & ~15u) | ((uint32_t)pos << 2));
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__shfl((int)s.v[i], src, 64);
        return r; }
    if (LANES == 8) return pos == 0 ? oct_bcast<0>(s) : (pos == 1 ? oct_bcast<2>(s) : oct_bcast<4>(s));
    return pos == 0 ? quad_bcast<0>(s) : (pos == 1 ? quad_bcast<1>(s) : quad_bcast<2>(s));
}
// which sponge a thread works on and whether it is the group's writer; blockDim.x is a multiple of 64
template <int LANES> __device__ __forceinline__ uint32_t coop_sponge_index(bool &writer) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (LANES == 3) { const uint32_t lane = threadIdx.x & 63u; writer = lane < 63u && lane % 3u == 0; return (gid >> 6) * 21u + (lane == 63u ? 20u : lane / 3u); }
    writer = (gid % LANES) == 0; return gid / LANES;
}
// threads needed for n sponges
template <int LANES> static inline size_t coop_threads(size_t n) { return LANES == 3 ? ((n + 20) / 21) * 64 : n * LANES; }
// Kernels that run several independent sponge ROLES per item (different absorb lists) put each role in its own blocks of 64, so that a
// wave never holds two roles (divergent roles in one wave would run one after the other: 3x the latency of a one-proof call).
// grid = roles * coop_role_blocks(n); inside: role = blockIdx.x / nblk, item = coop_role_item(blockIdx.x % nblk, writer)
template <int LANES> static inline uint32_t coop_role_blocks(size_t n) { return (uint32_t)((coop_threads<LANES>(n) + 63) / 64); }
template <int LANES> __device__ __forceinline__ uint32_t coop_role_item(uint32_t blk, bool &writer) {      // blockDim.x == 64
    const uint32_t lane = threadIdx.x & 63u;
    if (LANES == 3) { writer = lane < 63u && lane % 3u == 0; return blk * 21u + (lane == 63u ? 20u : lane / 3u); }
    writer = (lane % LANES) == 0; return (blk * 64u + lane) / LANES;
}

// mina-poseidon `ArithmeticSponge` state machine (rate 2) over base field F, Montgomery state, lane-cooperative over
// LANES = 3 (wave-packed triples), 4 or 8 lanes: `s` = the state element this lane owns (coop_elem), the position (squeezed, count)
// is replicated.  Absorbed values and squeezed results are replicated on all lanes of the group.
template <int F, int LANES> struct DevSponge {
    fe_t s; int squeezed; int count; const PoseidonParams *pp;
    __device__ void add_at(int pos, const fe_t &x) { if ((int)coop_elem<LANES>() == pos) s = fe_add<F>(s, x); }
    __device__ fe_t get(int pos) { return coop_get<LANES>(s, pos); }
    __device__ void absorb(const fe_t &x) {
        if (!squeezed) {
            if (count == 2) { poseidon_permute_coop<F, LANES>(s, pp); add_at(0, x); count = 1; }
            else { add_at(count, x); ++count; }
        } else { add_at(0, x); squeezed = 0; count = 1; }
    }
    __device__ fe_t squeeze() {
        if (!squeezed || count == 2) { poseidon_permute_coop<F, LANES>(s, pp); squeezed = 1; count = 1; return get(0); }
        return get(count++);
    }
};

template <int F> __device__ __forceinline__ fe_t load_fe(const uint32_t *p) { fe_t r; for (int i = 0; i < 8; ++i) r.v[i] = p[i]; return r; }
template <int LANES> __device__ __forceinline__ bool coop_writer() {
    if (LANES == 3) { const uint32_t lane = threadIdx.x & 63u; return lane < 63u && lane % 3u == 0; }
    return (threadIdx.x & (LANES - 1)) == 0;
}
template <int LANES> __device__ __forceinline__ uint32_t coop_lane() { return LANES == 3 ? tri_pos().e : (threadIdx.x & (LANES - 1)); }
// the lanes that hold state elements 0, 1, 2 of a cooperative sponge (one lane each)
template <int LANES> __device__ __forceinline__ bool coop_state_owner() {
    const uint32_t ln = coop_lane<LANES>();
    return LANES == 16 ? (ln < 12 && !(ln & 3u)) : (LANES == 8 ? (ln < 6 && !(ln & 1u)) : (LANES == 3 ? (threadIdx.x & 63u) < 63u : ln < 3));
}
template <int LANES> __device__ __forceinline__ void store_fe(uint32_t *p, const fe_t &a) { if (coop_writer<LANES>()) for (int i = 0; i < 8; ++i) p[i] = a.v[i]; }
template <int LANES> __device__ __forceinline__ void store_pt(affine_t *p, const affine_t &a) { if (coop_writer<LANES>()) *p = a; }



template <int FB> __device__ __forceinline__ affine_t load_point_mont(const uint32_t *p, const FieldK &kb) {
    affine_t a; a.x = fe_to_mont<FB>(load_fe<FB>(p), kb.r2); a.y = fe_to_mont<FB>(load_fe<FB>(p + 8), kb.r2); return a;
}
// The same with the checks upstream's deserialiser makes before `SRS::verify` ever sees a point: coordinates canonical and
// the point on y^2 = x^3 + 5 (or the (0,0) encoding of infinity).  A proof carrying anything else must be REJECTED -- off-curve
// points would otherwise enter the combined MSM -- so `ok` feeds the batch verdict.
template <int FB> __device__ __forceinline__ affine_t load_point_checked(const uint32_t *p, const FieldK &kb, bool &ok) {
    const fe_t xw = load_fe<FB>(p), yw = load_fe<FB>(p + 8);
    affine_t a; a.x = fe_to_mont<FB>(xw, kb.r2); a.y = fe_to_mont<FB>(yw, kb.r2);
    bool good = fe_words_canonical<FB>(xw) && fe_words_canonical<FB>(yw);
    if (good && !aff_is_inf(a)) good = fe_eq(fe_sqr<FB>(a.y), fe_add<FB>(fe_mul<FB>(fe_sqr<FB>(a.x), a.x), kb.five));
    ok = ok && good;
    return a;
}


// a16: Merkle-path fold.  One lane group (8 lanes, or a wave-packed triple for chip-filling batches) per path.  node <- H_height(left, right) with the per-height salted initial state
// (mina `hash_with_kimchi(MERKLE_PARAM[height], [l, r])`): state = salt[height]; state[0] += l; state[1] += r; permute;
// node = state[0].  dir 0 = MerkleNode::Left(h): node is the left input, h the right one; dir 1 = MerkleNode::Right(h).
template <int F, int LANES>
__global__ void __launch_bounds__(256)
merkle_fold_coop_kernel(uint32_t n, uint32_t depth, FieldK fk, const PoseidonParams *__restrict__ pp,
                        const fe_t *__restrict__ salts /* depth x 3, Montgomery */, const uint32_t *__restrict__ leaves,
                        const uint32_t *__restrict__ siblings /* n*depth*8 */, const uint8_t *__restrict__ dirs,
                        uint32_t *__restrict__ roots) {
    __builtin_amdgcn_s_setprio(3);                               // a short dependent chain: see salted_hash_kernel (api_account.hip)
    bool writer; const uint32_t path = coop_sponge_index<LANES>(writer), e = coop_elem<LANES>();
    const bool live = path < n;
    const uint32_t pidx = live ? path : 0;                     // dead groups shadow path 0 (whole wave runs the cross-lane moves)
    fe_t node; for (int i = 0; i < 8; ++i) node.v[i] = leaves[(size_t)pidx * 8 + i];
    node = fe_to_mont<F>(node, fk.r2);
    for (uint32_t h = 0; h < depth; ++h) {
        fe_t sib; for (int i = 0; i < 8; ++i) sib.v[i] = siblings[((size_t)pidx * depth + h) * 8 + i];
        sib = fe_to_mont<F>(sib, fk.r2);
        const bool node_is_left = dirs[(size_t)pidx * depth + h] == 0;
        fe_t st = salts[(size_t)h * 3 + e];
        if (e == 0) st = fe_add<F>(st, node_is_left ? node : sib);
        if (e == 1) st = fe_add<F>(st, node_is_left ? sib : node);
        poseidon_permute_coop<F, LANES>(st, pp);
        node = coop_get<LANES>(st, 0);
    }
    if (live && writer) { fe_t o = fe_from_mont<F>(node); for (int i = 0; i < 8; ++i) roots[(size_t)path * 8 + i] = o.v[i]; }
}

// salt[h] = state after absorbing the prefix element of height h into the zero state and permuting
template <int F>
__global__ void merkle_salt_kernel(uint32_t depth, FieldK fk, const PoseidonParams *__restrict__ pp, const uint32_t *__restrict__ prefixes,
                                   fe_t *__restrict__ salts) {
    uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= depth) return;
    fe_t s[3] = {fe_zero(), fe_zero(), fe_zero()};
    fe_t p; for (int i = 0; i < 8; ++i) p.v[i] = prefixes[(size_t)h * 8 + i];
    s[0] = fe_to_mont<F>(p, fk.r2);
    poseidon_permute<F>(s, pp);
    for (int j = 0; j < 3; ++j) salts[(size_t)h * 3 + j] = s[j];
}

// n independent sponges, lane-cooperative (8 lanes each, or wave-packed triples): absorb len elements, squeeze one (same contract as poseidon_hash_kernel)
template <int F, int LANES>
__global__ void __launch_bounds__(256)
poseidon_hash_coop_kernel(uint32_t n, uint32_t len, FieldK fk, const PoseidonParams *__restrict__ pp,
                          const uint32_t *__restrict__ inputs, uint32_t *__restrict__ out_words) {
    bool writer; const uint32_t sp = coop_sponge_index<LANES>(writer), e = coop_elem<LANES>();
    const bool live = sp < n;
    const uint32_t idx = live ? sp : 0;
    fe_t s = fe_zero();
    uint32_t count = 0;
    for (uint32_t el = 0; el < len; ++el) {
        if (count == 2) { poseidon_permute_coop<F, LANES>(s, pp); count = 0; }
        if (e == count) {                                                 // count is 0 or 1: the lane(s) owning that element
            fe_t w; for (int i = 0; i < 8; ++i) w.v[i] = inputs[((size_t)idx * len + el) * 8 + i];
            s = fe_add<F>(s, fe_to_mont<F>(w, fk.r2));
        }
        ++count;
    }
    poseidon_permute_coop<F, LANES>(s, pp);
    s = coop_get<LANES>(s, 0);
    if (live && writer) { fe_t w = fe_from_mont<F>(s); for (int i = 0; i < 8; ++i) out_words[(size_t)sp * 8 + i] = w.v[i]; }
}
#endif

// ScalarChallenge::to_field.  Upstream runs 64 rounds of "double a and b, add +-1 to one of them" in the field;
// a and b stay small integers (2^65 + a signed 64-bit sum), so they are assembled with integer bit masks and
// only the final a * endo + b touches the field (2 conversions + 1 product instead of ~200 field ops).
template <int F> MB_HD fe_t challenge_to_field(uint64_t lo, uint64_t hi, const FieldK &fk) {
    uint64_t pa = 0, na = 0, pb = 0, nb = 0;
    for (int i = 0; i < 64; ++i) {
        const uint64_t w = (i < 32) ? lo : hi;
        const int sh = (2 * i) & 63;
        const uint64_t r0 = (w >> sh) & 1u, r1 = (w >> (sh + 1)) & 1u;
        const uint64_t bit = (uint64_t)1 << i;
        if (r1) { if (r0) pa |= bit; else na |= bit; } else { if (r0) pb |= bit; else nb |= bit; }
    }
    fe_t a = fe_zero(), b = fe_zero();
    { uint64_t l = pa - na; uint32_t h = (pa >= na) ? 2u : 1u; a.v[0] = (uint32_t)l; a.v[1] = (uint32_t)(l >> 32); a.v[2] = h; }
    { uint64_t l = pb - nb; uint32_t h = (pb >= nb) ? 2u : 1u; b.v[0] = (uint32_t)l; b.v[1] = (uint32_t)(l >> 32); b.v[2] = h; }
    return fe_add<F>(fe_mul<F>(fe_to_mont<F>(a, fk.r2), fk.endo), fe_to_mont<F>(b, fk.r2));
}

// Per proof, no weights: out[j] = L[lo] * H[hi] in ONE launch (the batch path above needs tables + fold + finish).
// Block = BP1_HT consecutive `hi` values x the `lo`s of its lanes; grid.y = proof.  Every lane multiplies out its own
// L[lo] (<= 10 products, Montgomery); BP1_HT lanes of the last wave multiply out the block's H[hi] and take them OUT of
// Montgomery form, so that each output is a single product L * h (= l h, canonical) -- 1.5 products per coefficient
// instead of 10 when every lane rebuilt both factors.  Challenges come either as field elements (`chals`) or as the
// 128-bit prechallenges (`prechal`, 4 words each), converted by the first k lanes of every block (ScalarChallenge::to_field).
#if defined(__HIPCC__)
static constexpr uint32_t BP1_HT = 16;         // round 5: 16 (was 4): a lane's <= 10 products for L[lo] are shared by 16 outputs -- (10 + 16) / 16 = 1.6 products per coefficient against 3.5
template <int F>
__global__ void __launch_bounds__(256)
bpoly_single_kernel(BpolyShape sh, FieldK fk, const uint32_t *__restrict__ chals, const uint32_t *__restrict__ prechal,
                    uint32_t *__restrict__ out_words) {
    __shared__ fe_t ch[20];
    __shared__ fe_t hs[BP1_HT];
    if (chals) chals += (size_t)blockIdx.y * sh.k * 8;           // grid.y = proof
    if (prechal) prechal += (size_t)blockIdx.y * sh.k * 4;
    out_words += ((size_t)blockIdx.y << sh.k) * 8;
    const uint32_t nl = 1u << sh.lb, nh = 1u << sh.hb, lo_blocks = (nl + blockDim.x - 1) / blockDim.x;
    const uint32_t ht = nh < BP1_HT ? nh : BP1_HT;
    const uint32_t hi0 = (blockIdx.x / lo_blocks) * ht, lo = (blockIdx.x % lo_blocks) * blockDim.x + threadIdx.x;
    if (threadIdx.x < sh.k) {
        fe_t c;
        if (prechal) {
            const uint32_t *pc = prechal + (size_t)threadIdx.x * 4;
            c = challenge_to_field<F>((uint64_t)pc[0] | ((uint64_t)pc[1] << 32), (uint64_t)pc[2] | ((uint64_t)pc[3] << 32), fk);
        } else {
            for (int i = 0; i < 8; ++i) c.v[i] = chals[(size_t)threadIdx.x * 8 + i];
            c = fe_to_mont<F>(c, fk.r2);
        }
        ch[threadIdx.x] = c;
    }
    __syncthreads();
    const uint32_t hlane = blockDim.x - 1 - threadIdx.x;         // the LAST lanes of the block build the H values
    if (hlane < ht) {
        const uint32_t hi = hi0 + hlane;
        fe_t h = fk.one;
        for (uint32_t q = 0; q < sh.hb; ++q) if ((hi >> q) & 1u) h = fe_mul<F>(h, ch[sh.k - 1 - sh.lb - q]);   // bit q of hi: chals[k-1-lb-q]
        hs[hlane] = fe_from_mont<F>(h);
    }
    fe_t l = fk.one;
    if (lo < nl) for (uint32_t q = 0; q < sh.lb; ++q) if ((lo >> q) & 1u) l = fe_mul<F>(l, ch[sh.k - 1 - q]);  // bit q of lo: chals[k-1-q]
    __syncthreads();
    if (lo >= nl) return;
    for (uint32_t t = 0; t < ht; ++t) {
        const fe_t r = fe_mul<F>(l, hs[t]);                      // (l R)(h) R^-1 = l h, canonical
        uint4 *o = reinterpret_cast<uint4 *>(out_words + ((size_t)(hi0 + t) * nl + lo) * 8);
        o[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
        o[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
    }
}
#endif

#if defined(__HIPCC__)
template <int F>
__global__ void challenge_to_field_kernel(uint32_t n, FieldK fk, const uint32_t *__restrict__ chal /* n*4 words */, uint32_t *__restrict__ out_words) { mb_wave_prio();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t lo = (uint64_t)chal[(size_t)i * 4] | ((uint64_t)chal[(size_t)i * 4 + 1] << 32);
    uint64_t hi = (uint64_t)chal[(size_t)i * 4 + 2] | ((uint64_t)chal[(size_t)i * 4 + 3] << 32);
    fe_t r = fe_from_mont<F>(challenge_to_field<F>(lo, hi, fk));
    for (int q = 0; q < 8; ++q) out_words[(size_t)i * 8 + q] = r.v[q];
}

// vector field ops for the self-test hooks
template <int F>
__global__ void field_mul_kernel(uint32_t n, FieldK fk, const uint32_t *a, const uint32_t *b, uint32_t *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_t x, y; for (int q = 0; q < 8; ++q) { x.v[q] = a[(size_t)i * 8 + q]; y.v[q] = b[(size_t)i * 8 + q]; }
    fe_t r = fe_from_mont<F>(fe_mul<F>(fe_to_mont<F>(x, fk.r2), fe_to_mont<F>(y, fk.r2)));
    for (int q = 0; q < 8; ++q) out[(size_t)i * 8 + q] = r.v[q];
}
template <int F>
__global__ void field_inv_kernel(uint32_t n, FieldK fk, const uint32_t *a, uint32_t *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_t x; for (int q = 0; q < 8; ++q) x.v[q] = a[(size_t)i * 8 + q];
    fe_t r = fe_from_mont<F>(fe_inv<F>(fe_to_mont<F>(x, fk.r2), fk));
    for (int q = 0; q < 8; ++q) out[(size_t)i * 8 + q] = r.v[q];
}
template <int F>
__global__ void field_sqrt_kernel(uint32_t n, FieldK fk, const uint32_t *a, uint32_t *out, uint8_t *ok) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_t x; for (int q = 0; q < 8; ++q) x.v[q] = a[(size_t)i * 8 + q];
    fe_t r; bool sq = fe_sqrt<F>(r, fe_to_mont<F>(x, fk.r2), fk);
    r = sq ? fe_from_mont<F>(r) : fe_zero();
    ok[i] = sq ? 1 : 0;
    for (int q = 0; q < 8; ++q) out[(size_t)i * 8 + q] = r.v[q];
}
#endif

}  // namespace mb
