// groupmap.cuh -- K4: hash-to-curve for the Pasta curves + SRS generation.
//
// Replaces (pins in core/Cargo.toml:14-22, core/Cargo.lock:2825-2827):
//   ark-ff 0.3 `SquareRootField::sqrt` / `legendre` (Tonelli-Shanks with TWO_ADIC_ROOT_OF_UNITY = 5^t),
//   groupmap `BWParameters::{setup,to_group}` (u = 1, f(u) = 6),
//   poly-commitment `SRS::create` (BLAKE2b-512 of the big-endian index -> field -> to_group).
// The outputs are pinned bit-for-bit by the in-tree srs/vesta.srs, srs/pallas.srs.
#pragma once
#include "ec.cuh"

namespace mb {

// Everything a kernel needs to know about one field, computed on the host at context creation.
struct FieldK {
    fe_t one, r2;
    fe_t pm2, pm1d2, tm1d2;      // plain-integer exponents: p-2, (p-1)/2, (t-1)/2 with p-1 = 2^32 t
    fe_t root;                   // 5^t, Montgomery (2-adic root of unity)
    fe_t five;                   // curve b, Montgomery
    fe_t bw_fu, bw_s, bw_c, bw_inv3;   // group map: f(u)=6, sqrt(-3), (sqrt(-3)-1)/2, 1/3
    fe_t endo;                   // endo_r of the curve whose SCALAR field this is: (5^((p-1)/3))^2
    fe_t half;                   // (p-1)/2 plain, for the y-sign flag of the point codec
    fe_t two255, inv2;           // 2^255 and 1/2, Montgomery (shift_scalar of the Fq-sponge's absorb_fr)
    fe_t m32;                    // 32 in Montgomery form = the integer 2^261 mod p: one product by it moves a value from the 2^256 domain to fp29's 2^261 domain (ec29.cuh)
};

template <int F> MB_HD fe_t fe_inv(const fe_t &a, const FieldK &k) { return fe_pow<F>(a, k.pm2, k.one); }

template <int F> MB_HD bool fe_is_square(const fe_t &a, const FieldK &k) {
    if (fe_is_zero(a)) return true;
    return fe_eq(fe_pow<F>(a, k.pm1d2, k.one), k.one);
}

// ark-ff Tonelli-Shanks; returns false for a non-residue.  The particular root returned is part of
// the contract (it fixes the y-sign of every SRS point).
template <int F> MB_HD bool fe_sqrt(fe_t &out, const fe_t &a, const FieldK &k) {
    if (fe_is_zero(a)) { out = a; return true; }
    fe_t w = fe_pow<F>(a, k.tm1d2, k.one);
    fe_t x = fe_mul<F>(w, a);
    fe_t b = fe_mul<F>(x, w);
    // Legendre via b^(2^31): b = a^t
    {
        fe_t l = b;
        for (int i = 0; i < 31; ++i) l = fe_sqr<F>(l);
        if (!fe_eq(l, k.one)) return false;
    }
    fe_t z = k.root;
    int v = 32;
    while (!fe_eq(b, k.one)) {
        int kk = 0; fe_t b2k = b;
        while (!fe_eq(b2k, k.one)) { b2k = fe_sqr<F>(b2k); ++kk; }
        int j = v - kk - 1;
        w = z;
        for (int i = 0; i < j; ++i) w = fe_sqr<F>(w);
        z = fe_sqr<F>(w);
        b = fe_mul<F>(b, z);
        x = fe_mul<F>(x, w);
        v = kk;
    }
    out = x; return true;
}

// groupmap `BWParameters::to_group`
template <int F> MB_HD affine_t bw_to_group(const fe_t &t, const FieldK &k) {
    fe_t t2 = fe_sqr<F>(t);
    fe_t t2pf = fe_add<F>(t2, k.bw_fu);
    fe_t alpha_inv = fe_mul<F>(t2pf, t2);
    fe_t alpha = fe_is_zero(alpha_inv) ? alpha_inv : fe_inv<F>(alpha_inv, k);
    fe_t t4 = fe_sqr<F>(t2);
    fe_t xs[3];
    xs[0] = fe_sub<F>(k.bw_c, fe_mul<F>(fe_mul<F>(t4, alpha), k.bw_s));
    xs[1] = fe_sub<F>(fe_neg<F>(k.one), xs[0]);
    xs[2] = fe_sub<F>(k.one, fe_mul<F>(fe_mul<F>(fe_mul<F>(fe_sqr<F>(t2pf), alpha), t2pf), k.bw_inv3));
    affine_t r; r.x = fe_zero(); r.y = fe_zero();
    for (int i = 0; i < 3; ++i) {
        fe_t y2 = fe_add<F>(fe_mul<F>(fe_sqr<F>(xs[i]), xs[i]), k.five);
        fe_t y;
        if (fe_sqrt<F>(y, y2, k)) { r.x = xs[i]; r.y = y; return r; }
    }
    return r;
}

// ---------------------------------------------------------------- BLAKE2b-512 (RFC 7693), <= 128-byte messages
MB_HD uint64_t b2_rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

MB_HD void blake2b_short(const uint8_t *msg, uint32_t len, uint8_t *out, uint32_t outlen /* 1..64 */) {
    const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                            0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    const uint8_t sigma[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    uint64_t h[8], m[16], v[16];
    for (int i = 0; i < 8; ++i) h[i] = iv[i];
    h[0] ^= 0x01010000ULL ^ (uint64_t)outlen;   // digest length, fanout 1, depth 1
    for (int i = 0; i < 16; ++i) {
        uint64_t w = 0;
        for (int j = 7; j >= 0; --j) { uint32_t idx = 8 * i + j; w = (w << 8) | (idx < len ? msg[idx] : 0); }
        m[i] = w;
    }
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = iv[i]; }
    v[12] ^= (uint64_t)len; v[14] = ~v[14];
    for (int r = 0; r < 12; ++r) {
        const uint8_t *s = sigma[r % 10];
#define MB_B2G(a, b, c, d, x, y)                                                        \
    v[a] = v[a] + v[b] + (x); v[d] = b2_rotr(v[d] ^ v[a], 32); v[c] = v[c] + v[d];      \
    v[b] = b2_rotr(v[b] ^ v[c], 24); v[a] = v[a] + v[b] + (y);                          \
    v[d] = b2_rotr(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = b2_rotr(v[b] ^ v[c], 63);
        MB_B2G(0, 4, 8, 12, m[s[0]], m[s[1]]) MB_B2G(1, 5, 9, 13, m[s[2]], m[s[3]])
        MB_B2G(2, 6, 10, 14, m[s[4]], m[s[5]]) MB_B2G(3, 7, 11, 15, m[s[6]], m[s[7]])
        MB_B2G(0, 5, 10, 15, m[s[8]], m[s[9]]) MB_B2G(1, 6, 11, 12, m[s[10]], m[s[11]])
        MB_B2G(2, 7, 8, 13, m[s[12]], m[s[13]]) MB_B2G(3, 4, 9, 14, m[s[14]], m[s[15]])
#undef MB_B2G
    }
    for (int i = 0; i < 8; ++i) {
        uint64_t w = h[i] ^ v[i] ^ v[i + 8];
        for (int j = 0; j < 8; ++j) if ((uint32_t)(8 * i + j) < outlen) out[8 * i + j] = (uint8_t)(w >> (8 * j));
    }
}
MB_HD void blake2b512_short(const uint8_t *msg, uint32_t len, uint8_t out[64]) { blake2b_short(msg, len, out, 64); }

// first 31 digest bytes, each unpacked LSB-first, read as one big-endian bit string (248 bits < p)
MB_HD fe_t digest_to_plain_fe(const uint8_t d[64]) {
    fe_t r = fe_zero();
    for (int i = 0; i < 31; ++i)
        for (int j = 0; j < 8; ++j) {
            for (int l = 7; l > 0; --l) r.v[l] = (r.v[l] << 1) | (r.v[l - 1] >> 31);
            r.v[0] = (r.v[0] << 1) | ((d[i] >> j) & 1u);
        }
    return r;
}

#if defined(__HIPCC__)
// g[i] for i < depth; thread `depth` produces h ("srs_misc" || u32_be(0)).  out: Montgomery affine.
template <int F>
__global__ void __launch_bounds__(256)
srs_create_kernel(uint32_t depth, FieldK k, affine_t *__restrict__ g_out, affine_t *__restrict__ h_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > depth) return;
    uint8_t msg[12], d[64]; uint32_t len;
    if (i < depth) { msg[0] = (uint8_t)(i >> 24); msg[1] = (uint8_t)(i >> 16); msg[2] = (uint8_t)(i >> 8); msg[3] = (uint8_t)i; len = 4; }
    else { const char *s = "srs_misc"; for (int j = 0; j < 8; ++j) msg[j] = (uint8_t)s[j]; msg[8] = msg[9] = msg[10] = msg[11] = 0; len = 12; }
    blake2b512_short(msg, len, d);
    fe_t t = fe_to_mont<F>(digest_to_plain_fe(d), k.r2);
    affine_t p = bw_to_group<F>(t, k);
    if (i < depth) g_out[i] = p; else *h_out = p;
}

template <int F>
__global__ void to_group_kernel(uint32_t n, FieldK k, const uint32_t *__restrict__ t_words, uint32_t *__restrict__ out_words) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_t t; for (int j = 0; j < 8; ++j) t.v[j] = t_words[(size_t)i * 8 + j];
    affine_t p = bw_to_group<F>(fe_to_mont<F>(t, k.r2), k);
    fe_t x = fe_from_mont<F>(p.x), y = fe_from_mont<F>(p.y);
    for (int j = 0; j < 8; ++j) { out_words[(size_t)i * 16 + j] = x.v[j]; out_words[(size_t)i * 16 + 8 + j] = y.v[j]; }
}

// compressed (ark-serialize 0.3) -> Montgomery affine.  in: 33 bytes per point.  *bad counts points that ark would not
// deserialise: non-canonical x, stray bits in the flag byte, x not on the curve.
template <int F>
__global__ void decompress_kernel(uint32_t n, FieldK k, const uint8_t *__restrict__ in, affine_t *__restrict__ out,
                                  uint32_t *__restrict__ bad) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *b = in + (size_t)i * 33;
    affine_t p; p.x = fe_zero(); p.y = fe_zero();
    fe_t x;
    for (int j = 0; j < 8; ++j) x.v[j] = (uint32_t)b[4 * j] | ((uint32_t)b[4 * j + 1] << 8) | ((uint32_t)b[4 * j + 2] << 16) | ((uint32_t)b[4 * j + 3] << 24);
    // ark reads x (with the two flag bits on top of byte 32) as ONE canonical field element: the other six bits of the
    // flag byte are its bits 256..261 and must be zero, and x itself must be below the modulus -- also for infinity
    if ((b[32] & 0x3f) || !fe_words_canonical<F>(x)) { atomicAdd(bad, 1u); out[i] = p; return; }
    if (b[32] & 0x40) { out[i] = p; return; }
    fe_t xm = fe_to_mont<F>(x, k.r2);
    fe_t y2 = fe_add<F>(fe_mul<F>(fe_sqr<F>(xm), xm), k.five), y;
    if (!fe_sqrt<F>(y, y2, k)) { atomicAdd(bad, 1u); out[i] = p; return; }
    // flag 0x80 <=> y > (p-1)/2 as integers
    fe_t yp = fe_from_mont<F>(y);
    bool hi = false;
    for (int j = 7; j >= 0; --j) { if (yp.v[j] != k.half.v[j]) { hi = yp.v[j] > k.half.v[j]; break; } }
    if (hi != ((b[32] & 0x80) != 0)) y = fe_neg<F>(y);
    p.x = xm; p.y = y; out[i] = p;
}

// Montgomery affine -> 33-byte compressed
template <int F>
__global__ void compress_kernel(uint32_t n, FieldK k, const affine_t *__restrict__ in, uint8_t *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    affine_t p = in[i];
    uint8_t *o = out + (size_t)i * 33;
    if (aff_is_inf(p)) { for (int j = 0; j < 32; ++j) o[j] = 0; o[32] = 0x40; return; }
    fe_t x = fe_from_mont<F>(p.x), y = fe_from_mont<F>(p.y);
    for (int j = 0; j < 8; ++j) { o[4 * j] = (uint8_t)x.v[j]; o[4 * j + 1] = (uint8_t)(x.v[j] >> 8); o[4 * j + 2] = (uint8_t)(x.v[j] >> 16); o[4 * j + 3] = (uint8_t)(x.v[j] >> 24); }
    bool hi = false;
    for (int j = 7; j >= 0; --j) { if (y.v[j] != k.half.v[j]) { hi = y.v[j] > k.half.v[j]; break; } }
    o[32] = hi ? 0x80 : 0x00;
}
#endif

}  // namespace mb
