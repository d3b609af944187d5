// batch_affine_probe.hip -- K1 question of round 4: would affine bucket accumulation with one shared inversion per batch
// (Montgomery's trick: 5M + 1S per add + the inversion's share) beat the XYZZ mixed add (8M + 2S) the MSM kernels use, ON THIS CHIP?
// Standalone: the library's own field / group routines (csrc/fp.cuh, ec.cuh), a 64 MiB table of points gathered at random like the SRS window table.
//   A  xyzz    : lane = one bucket, K sequential mixed adds of gathered points into an XYZZ accumulator            (msm_accumulate_bucket_kernel's inner loop)
//   B  affine  : lane = n independent additions P_j + Q_j (one level of a pairwise tree / one round over n buckets): prefix products of the
//                denominators to global scratch, ONE Fermat inversion per lane, back-substitution, affine results written out
//   B' affine0 : B without the inversion (wrong results): the floor a free inversion would give
// The break-even is about n: the inversion (255 squarings + ~65 products, ~320 products' worth) is paid per LANE, because every lane of a wave
// executes it -- sharing it across lanes costs a cross-lane product scan per level that is as expensive as the additions it saves.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mina_bridge_amd/csrc tools/probes/batch_affine_probe.hip -o gpurun_out/batch_affine_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ec.cuh"
#include "groupmap.cuh"
#include "fp29.cuh"
using namespace mb;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int F = FIELD_FQ;

__device__ __forceinline__ affine_t load_affine(const affine_t *__restrict__ p) {
    affine_t r; const uint4 *q = reinterpret_cast<const uint4 *>(p); uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    r.x.v[0] = a.x; r.x.v[1] = a.y; r.x.v[2] = a.z; r.x.v[3] = a.w; r.x.v[4] = b.x; r.x.v[5] = b.y; r.x.v[6] = b.z; r.x.v[7] = b.w;
    r.y.v[0] = c.x; r.y.v[1] = c.y; r.y.v[2] = c.z; r.y.v[3] = c.w; r.y.v[4] = d.x; r.y.v[5] = d.y; r.y.v[6] = d.z; r.y.v[7] = d.w;
    return r;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
xyzz_kernel(uint32_t lanes, uint32_t K, const uint32_t *__restrict__ refs, const affine_t *__restrict__ table, fe_t one, xyzz_t *__restrict__ out) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= lanes) return;
    xyzz_t acc = xyzz_inf();
    affine_t nxt = load_affine(table + refs[(size_t)l * K]);
#pragma unroll 1
    for (uint32_t e = 0; e < K; ++e) {
        const affine_t p = nxt;
        if (e + 1 < K) nxt = load_affine(table + refs[(size_t)l * K + e + 1]);
        xyzz_add_affine<F>(acc, p.x, p.y, one);
    }
    out[l] = acc;
}

// ---- C  xyzz29 : the same mixed add on 9 limbs of 29 bits (fp29.cuh: no carry instructions, lazy reduction -- every value stays below 16 p = 2^258,
//                  subtractions add 8 p in a redundant limb form and renormalise); table coordinates are x * 2^261 mod p as 8 x 32-bit words
template <int FF, uint32_t MULT> struct KMP {   // MULT * p in normalised 29-bit limbs n_i
    static constexpr uint64_t t0 = (uint64_t)MULT * 1u, n0 = t0 & M29, t1 = (uint64_t)MULT * P29<FF>::L1 + (t0 >> 29), n1 = t1 & M29, t2 = (uint64_t)MULT * P29<FF>::L2 + (t1 >> 29), n2 = t2 & M29,
                              t3 = (uint64_t)MULT * P29<FF>::L3 + (t2 >> 29), n3 = t3 & M29, t4 = (uint64_t)MULT * P29<FF>::L4 + (t3 >> 29), n4 = t4 & M29, n5 = t4 >> 29, n8 = (uint64_t)MULT * P29<FF>::L8;
};
// a + MULT p - b, normalised; needs b < MULT p.  MULT p enters in a redundant limb form -- K_0 = n_0 + 2^30, K_i = n_i + 2^30 - 2 (i = 1..7), K_8 = n_8 - 2: the same
// integer, every limb above any b_i -- so the limb-wise difference never goes negative and ONE carry pass normalises it
template <int FF, uint32_t MULT> __device__ __forceinline__ fe29_t fe29_sub_kp(const fe29_t &a, const fe29_t &b) {
    typedef KMP<FF, MULT> K;
    const uint32_t k[9] = {(uint32_t)K::n0 + (1u << 30), (uint32_t)K::n1 + (1u << 30) - 2, (uint32_t)K::n2 + (1u << 30) - 2, (uint32_t)K::n3 + (1u << 30) - 2, (uint32_t)K::n4 + (1u << 30) - 2,
                           (uint32_t)K::n5 + (1u << 30) - 2, (1u << 30) - 2, (1u << 30) - 2, (uint32_t)K::n8 - 2};
    fe29_t r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L29; ++i) { const uint32_t t = a.v[i] + k[i] - b.v[i] + c; if (i < L29 - 1) { r.v[i] = t & M29; c = t >> 29; } else r.v[i] = t; }
    return r;
}
template <int FF> __device__ __forceinline__ fe29_t fe29_sub8p(const fe29_t &a, const fe29_t &b) { return fe29_sub_kp<FF, 8>(a, b); }
template <int FF> __device__ __forceinline__ fe29_t fe29_sub4p(const fe29_t &a, const fe29_t &b) { return fe29_sub_kp<FF, 4>(a, b); }
struct xyzz29_t { fe29_t x, y, zz, zzz; };
// p == 0 (mod p) for a lazily reduced value below 16 p: a multiple k p = k 2^254 + k c has limbs 5..7 and the low 22 bits of limb 8 zero (k c < 2^129): the cheap
// necessary test; the exact one (limbs 0..4 == k c) only behind it
template <int FF> __device__ __forceinline__ bool fe29_maybe_zero(const fe29_t &a) { return ((a.v[5] | a.v[6] | a.v[7] | (a.v[8] & 0x3fffffu)) == 0u); }
template <int FF> __device__ __forceinline__ void xyzz29_add_affine(xyzz29_t &acc, bool &inf, const fe29_t &qx, const fe29_t &qy, const fe29_t &one29, uint32_t &rare) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (inf) { acc.x = qx; acc.y = qy; acc.zz = one29; acc.zzz = one29; inf = false; return; }
    const fe29_t u2 = fe29_mul_asm<FF>(qx, acc.zz), s2 = fe29_mul_asm<FF>(qy, acc.zzz);
    const fe29_t pd = fe29_sub8p<FF>(u2, acc.x), r = fe29_sub8p<FF>(s2, acc.y);
    if (fe29_maybe_zero<FF>(pd)) { ++rare; }                     // (the library version falls back to the 8 x 32 law here: P == +-Q)
    const fe29_t pp = fe29_sqr_asm<FF>(pd), ppp = fe29_mul_asm<FF>(pd, pp), q = fe29_mul_asm<FF>(acc.x, pp);
    const fe29_t t = fe29_add(ppp, fe29_add(q, q));
    const fe29_t x3 = fe29_sub4p<FF>(fe29_sqr_asm<FF>(r), t);                 // t < 3.4 p; x3 < 6 p: the 8 p of the next addition's subtractions covers it
    fe29_t zero; for (int i = 0; i < L29; ++i) zero.v[i] = 0;
    const fe29_t y3 = fe29_dot2_asm<FF>(r, fe29_sub8p<FF>(q, x3), fe29_sub8p<FF>(zero, acc.y), ppp);
    acc.zz = fe29_mul_asm<FF>(acc.zz, pp); acc.zzz = fe29_mul_asm<FF>(acc.zzz, ppp);
    acc.x = x3; acc.y = y3;
#endif
}
template <int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, WAVES)))
xyzz29_kernel(uint32_t lanes, uint32_t K, const uint32_t *__restrict__ refs, const affine_t *__restrict__ table29, fe29_t one29, fe29_t leave, xyzz_t *__restrict__ out, uint32_t *__restrict__ rare_out) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= lanes) return;
    xyzz29_t acc; bool inf = true; uint32_t rare = 0;
    affine_t nxt = load_affine(table29 + refs[(size_t)l * K]);
#pragma unroll 1
    for (uint32_t e = 0; e < K; ++e) {
        const affine_t p = nxt;
        if (e + 1 < K) nxt = load_affine(table29 + refs[(size_t)l * K + e + 1]);
        xyzz29_add_affine<F>(acc, inf, fe29_from_words(p.x), fe29_from_words(p.y), one29, rare);
    }
    // back to 8 x 32 Montgomery-2^256, canonical: one product by 2^256 mod p each (result < 1.01 p) and a conditional subtraction
#if defined(__HIP_DEVICE_COMPILE__)
    xyzz_t o;
    o.x = fe_cond_sub_p<F>(fe29_to_words(fe29_mul_asm<F>(acc.x, leave))); o.y = fe_cond_sub_p<F>(fe29_to_words(fe29_mul_asm<F>(acc.y, leave)));
    o.zz = fe_cond_sub_p<F>(fe29_to_words(fe29_mul_asm<F>(acc.zz, leave))); o.zzz = fe_cond_sub_p<F>(fe29_to_words(fe29_mul_asm<F>(acc.zzz, leave)));
    out[l] = o;
    if (rare) atomicAdd(rare_out, rare);
#endif
}
// table29[i] = table[i] * 2^5 (Montgomery-2^256 words of x -> words of x * 2^261 mod p): one product by mont(32) per coordinate
__global__ void to_table29_kernel(uint32_t n, const affine_t *__restrict__ in, fe_t m32, affine_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    affine_t p = in[i]; p.x = fe_mul<F>(p.x, m32); p.y = fe_mul<F>(p.y, m32); out[i] = p;
}
__global__ void cmp_xyzz_kernel(uint32_t n, const xyzz_t *__restrict__ a, const xyzz_t *__restrict__ b, uint32_t *bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    bool same = true; for (int w = 0; w < 8; ++w) same = same && a[i].x.v[w] == b[i].x.v[w] && a[i].y.v[w] == b[i].y.v[w] && a[i].zz.v[w] == b[i].zz.v[w] && a[i].zzz.v[w] == b[i].zzz.v[w];
    if (!same) atomicAdd(bad, 1u);
}

template <bool INVERT>
__global__ void __launch_bounds__(256)
affine_kernel(uint32_t lanes, uint32_t n, const uint32_t *__restrict__ refs, const affine_t *__restrict__ table, FieldK fk, fe_t *__restrict__ scratch /* n x lanes */,
              affine_t *__restrict__ out /* lanes x n */) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= lanes) return;
    const uint32_t *r = refs + (size_t)l * 2 * n;
    fe_t pref = fk.one;
#pragma unroll 1
    for (uint32_t j = 0; j < n; ++j) {                           // pass 1: c_j = d_0 ... d_j, d_j = x2 - x1 (x only: 2 x 32 B gathered)
        const fe_t x1 = table[r[2 * j]].x, x2 = table[r[2 * j + 1]].x;
        scratch[(size_t)j * lanes + l] = pref;                   // c_{j-1}
        pref = fe_mul<F>(pref, fe_sub<F>(x2, x1));
    }
    fe_t inv = INVERT ? fe_inv<F>(pref, fk) : pref;
#pragma unroll 1
    for (uint32_t j = n; j-- > 0;) {                             // pass 2: 1 / d_j = inv * c_{j-1};  inv *= d_j
        const affine_t P = load_affine(table + r[2 * j]), Q = load_affine(table + r[2 * j + 1]);
        const fe_t d = fe_sub<F>(Q.x, P.x);
        const fe_t dinv = fe_mul<F>(inv, scratch[(size_t)j * lanes + l]);
        inv = fe_mul<F>(inv, d);
        const fe_t lam = fe_mul<F>(fe_sub<F>(Q.y, P.y), dinv);
        affine_t R;
        R.x = fe_sub<F>(fe_sub<F>(fe_sqr<F>(lam), P.x), Q.x);
        R.y = fe_sub<F>(fe_mul<F>(lam, fe_sub<F>(P.x, R.x)), P.y);
        out[(size_t)l * n + j] = R;
    }
}

// cross-check of B against the XYZZ law on the first `cnt` additions of lane 0..: mismatches counted
__global__ void check_kernel(uint32_t lanes, uint32_t n, uint32_t cnt, const uint32_t *__restrict__ refs, const affine_t *__restrict__ table, FieldK fk, const affine_t *__restrict__ out, uint32_t *bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const uint32_t l = i / n, j = i % n;
    const affine_t P = table[refs[(size_t)l * 2 * n + 2 * j]], Q = table[refs[(size_t)l * 2 * n + 2 * j + 1]];
    xyzz_t a = xyzz_from_affine<F>(P, fk.one);
    xyzz_add_affine<F>(a, Q.x, Q.y, fk.one);
    const fe_t zi = fe_inv<F>(a.zz, fk), zzzi = fe_inv<F>(a.zzz, fk);
    const fe_t x = fe_mul<F>(a.x, zi), y = fe_mul<F>(a.y, zzzi);
    const affine_t R = out[(size_t)l * n + j];
    bool same = true; for (int w = 0; w < 8; ++w) same = same && x.v[w] == R.x.v[w] && y.v[w] == R.y.v[w];
    if (!same) atomicAdd(bad, 1u);
}

template <class K> static double time_kernel(K launch, int reps) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e-3 / reps;
}

FieldK make_fk();   // below: the library builds FieldK in api_core.hip; the probe needs one, pm2, r2

int main() {
    const uint32_t M = 1u << 20;                                  // table points (64 MiB, the size of one curve's window table)
    std::vector<affine_t> tab(M);
    uint64_t s = 0x9e3779b97f4a7c15ull; auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto &p : tab) for (int w = 0; w < 8; ++w) { p.x.v[w] = (uint32_t)rnd(); p.y.v[w] = (uint32_t)rnd(); if (w == 7) { p.x.v[7] &= 0x3fffffffu; p.y.v[7] &= 0x3fffffffu; } }
    const size_t total_adds = (size_t)1 << 23;                    // 8.4 M additions per launch (8 MSMs of 2^16 x 16 windows)
    std::vector<uint32_t> refs(2 * total_adds); for (auto &r : refs) r = (uint32_t)(rnd() % M);
    affine_t *d_tab; uint32_t *d_refs, *d_bad; fe_t *d_scratch; affine_t *d_out; xyzz_t *d_x;
    CHECK(hipMalloc(&d_tab, M * sizeof(affine_t))); CHECK(hipMalloc(&d_refs, refs.size() * 4)); CHECK(hipMalloc(&d_scratch, total_adds * sizeof(fe_t)));
    CHECK(hipMalloc(&d_out, total_adds * sizeof(affine_t))); CHECK(hipMalloc(&d_x, total_adds / 16 * sizeof(xyzz_t))); CHECK(hipMalloc(&d_bad, 4));
    CHECK(hipMemcpy(d_tab, tab.data(), M * sizeof(affine_t), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_refs, refs.data(), refs.size() * 4, hipMemcpyHostToDevice));
    const FieldK fk = make_fk();
    printf("{\"probe\": \"batch_affine_probe\", \"adds_per_launch\": %zu, \"table_MiB\": %zu}\n", total_adds, M * sizeof(affine_t) >> 20);
    for (uint32_t K : {16u, 32u, 64u}) {                          // A: XYZZ, K adds per lane
        const uint32_t lanes = (uint32_t)(total_adds / K);
        const double t = time_kernel([&] { xyzz_kernel<<<(lanes + 255) / 256, 256>>>(lanes, K, d_refs, d_tab, fk.one, d_x); }, 3);
        printf("{\"form\": \"xyzz mixed add (8M + 2S)\", \"adds_per_lane\": %u, \"lanes\": %u, \"ns_per_add_chip\": %.4f, \"G_adds_per_s\": %.2f}\n", K, lanes, t * 1e9 / total_adds, total_adds / t / 1e9);
    }
    {   // C: the same additions on 29-bit limbs; results compared coordinate by coordinate with A's
        affine_t *d_tab29; xyzz_t *d_x29; uint32_t *d_rare;
        CHECK(hipMalloc(&d_tab29, M * sizeof(affine_t))); CHECK(hipMalloc(&d_x29, total_adds / 16 * sizeof(xyzz_t))); CHECK(hipMalloc(&d_rare, 4)); CHECK(hipMemset(d_rare, 0, 4));
        fe_t thirty2 = fe_zero(); thirty2.v[0] = 32; const fe_t m32 = fe_to_mont<F>(thirty2, fk.r2);
        to_table29_kernel<<<(M + 255) / 256, 256>>>(M, d_tab, m32, d_tab29);
        // one29 = 2^261 mod p = mont256(32) as an integer; leave = 2^256 mod p in the 2^261 domain = (2^256 * 2^261) mod p = mont256(2^261 mod p ...): computed on the host with the 8 x 32 routines
        const fe29_t one29 = fe29_from_words(m32);                                        // the integer 2^261 mod p
        const fe_t r512_261 = fe_mul<F>(fe_mul<F>(fk.r2, fk.r2), fk.one);                 // mont mult: r2 * r2 / R = R^3 mod p ... see below
        (void)r512_261;
        // leave: multiplying x 2^261 by L with the 2^261-Montgomery product gives x 2^261 L / 2^261 = x L; we want x 2^256, so L = 2^256 mod p = the integer fk.one
        const fe29_t leave = fe29_from_words(fk.one);
        for (uint32_t K : {16u, 32u, 64u}) {
            const uint32_t lanes = (uint32_t)(total_adds / K);
            xyzz_kernel<<<(lanes + 255) / 256, 256>>>(lanes, K, d_refs, d_tab, fk.one, d_x);
            const double t2 = time_kernel([&] { xyzz29_kernel<2><<<(lanes + 255) / 256, 256>>>(lanes, K, d_refs, d_tab29, one29, leave, d_x29, d_rare); }, 3);
            CHECK(hipMemset(d_bad, 0, 4));
            cmp_xyzz_kernel<<<(lanes + 255) / 256, 256>>>(lanes, d_x, d_x29, d_bad);
            uint32_t bad = 0, rare = 0; CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&rare, d_rare, 4, hipMemcpyDeviceToHost));
            const double t4 = time_kernel([&] { xyzz29_kernel<4><<<(lanes + 255) / 256, 256>>>(lanes, K, d_refs, d_tab29, one29, leave, d_x29, d_rare); }, 3);
            printf("{\"form\": \"xyzz mixed add on 9 x 29-bit limbs (lazy reduction)\", \"adds_per_lane\": %u, \"lanes\": %u, \"ns_per_add_chip_2_waves_per_simd\": %.4f, \"G_adds_per_s_2_waves\": %.2f, "
                   "\"ns_per_add_chip_4_waves_per_simd\": %.4f, \"G_adds_per_s_4_waves\": %.2f, \"buckets_differing_from_the_8x32_law\": %u, \"rare_path_hits\": %u}\n",
                   K, lanes, t2 * 1e9 / total_adds, total_adds / t2 / 1e9, t4 * 1e9 / total_adds, total_adds / t4 / 1e9, bad, rare);
        }
    }
    for (uint32_t n : {16u, 32u, 64u, 128u, 256u, 512u}) {        // B / B': batched affine, n adds per lane per inversion
        const uint32_t lanes = (uint32_t)(total_adds / n);
        const double t = time_kernel([&] { affine_kernel<true><<<(lanes + 255) / 256, 256>>>(lanes, n, d_refs, d_tab, fk, d_scratch, d_out); }, 3);
        CHECK(hipMemset(d_bad, 0, 4));
        check_kernel<<<(4096 + 255) / 256, 256>>>(lanes, n, 4096, d_refs, d_tab, fk, d_out, d_bad);
        uint32_t bad = 0; CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        const double t0 = time_kernel([&] { affine_kernel<false><<<(lanes + 255) / 256, 256>>>(lanes, n, d_refs, d_tab, fk, d_scratch, d_out); }, 3);
        printf("{\"form\": \"batched affine (5M + 1S + inversion / n)\", \"adds_per_lane_per_inversion\": %u, \"lanes\": %u, \"waves_per_simd\": %.2f, \"ns_per_add_chip\": %.4f, \"G_adds_per_s\": %.2f, "
               "\"without_the_inversion_ns_per_add\": %.4f, \"mismatches_vs_xyzz_of_4096\": %u}\n", n, lanes, lanes / 64.0 / 1024.0, t * 1e9 / total_adds, total_adds / t / 1e9, t0 * 1e9 / total_adds, bad);
    }
    return 0;
}

// FieldK of Fq from the modulus (as api_core.hip make_field_consts; only what the probe uses: one, r2, pm2)
FieldK make_fk() {
    FieldK k; fe_t p; for (int i = 0; i < 8; ++i) p.v[i] = modulus_limb<F>(i);
    fe_t a = fe_zero(); a.v[0] = 1;
    for (int i = 0; i < 512; ++i) { a = fe_add<F>(a, a); if (i == 255) k.one = a; }
    k.r2 = a;
    k.pm2 = p; { uint64_t br = 2; for (int i = 0; i < 8 && br; ++i) { uint64_t t = (uint64_t)k.pm2.v[i] - br; k.pm2.v[i] = (uint32_t)t; br = (t >> 32) & 1u; } }
    return k;
}
