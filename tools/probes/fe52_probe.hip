// fe52_probe.hip -- VERDICT r04 "next" #1: can an FP64-FMA Montgomery product on 5 x 52-bit limbs beat the 9 x 29-bit v_mad_u64_u32 schoolbook product of
// fp29.cuh on gfx950?  v_fma_f64 issues at the rate of v_mad_u64_u32 (16 lanes per clock per SIMD), and one FMA pair yields a 104-bit limb product -- but the
// pair needs a subtraction between its halves and both halves must be ADDED into integer columns, where v_mad_u64_u32 accumulates for free.
//
// What is here (arithmetic to match: ark-ff 0.3 `Fp256` Montgomery over the Pasta primes, /root/reference/core/Cargo.toml:19-21; same field elements, R = 2^260):
//   * fe52_mul: the complete product.  Limbs are doubles holding integers < 2^52.  Every limb product a_i * b_j is split exactly by the FMA pair
//         hi_raw = fma_rz(a, b, 2^104)                   = 2^104 + 2^52 * floor(ab / 2^52)     (mantissa field = hi)
//         lo_raw = fma_rz(a, b, (2^104 + 2^52) - hi_raw) = 2^52 + (ab mod 2^52)                (mantissa field = lo)
//     (round-toward-zero, set once per wave in the MODE register), and the RAW BIT PATTERNS are summed as 64-bit integers into the columns -- the exponent
//     fields add up to a constant per column that the column's initial value cancels.  Word-serial Montgomery reduction by the sparse p = 2^254 + t
//     (limbs p0, p1, p2 < 2^22, 0, 2^46): q = lo52(t_i * (-1/p)), three FMA pairs for q * {p0, p1, p2}, integer shifts for q * 2^46.
//   * values: `--values` prints 2 x 4096 (a, b, fe52_mul(a, b)) triples (random 256-bit operands + all-ones / zero / one edge limbs) and tools/probes/fe52_check.py
//     verifies r * 2^260 == a * b (mod p), r < 2^260, with Python integers, for BOTH fields (the same script carries the exact model of the FMA pair).
//   * throughput: dependent chains of fe52_mul against fe29_mul_asm (fp29.cuh) at 1 .. 8 waves per SIMD, products per second of the whole chip.
//   * the instruction count comes from the ISA (tools/probes/fe52_slots.sh: llvm-objdump of this object, VALU instructions between the two s_nop markers).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mina_bridge_amd/csrc -o tools/probes/bin/fe52_probe tools/probes/fe52_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "fp.cuh"
#include "fp29.cuh"

using namespace mb;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct fe52_t { double v[5]; };                          // integers < 2^52 (a lazily reduced value may use the whole 260 bits)
struct fe52_consts { double p[3]; double pinv; };      // p0, p1, p2 (limbs 0..2 of p; limb 3 = 0, limb 4 = 2^46) and -1/p mod 2^52

__device__ __forceinline__ double fma_rz(double a, double b, double c) { double d; asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ double sub_f64(double a, double b) { double d; asm volatile("v_add_f64 %0, %1, -%2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// f64 / f16 round toward zero for the rest of the wave.  As inline asm AFTER the last compiler-generated u64 -> f64 conversion: the backend's mode-register pass brackets
// those conversions with its own s_setreg and restores what it believes is the default (round to nearest) -- a builtin call placed before them is undone.
__device__ __forceinline__ void set_round_toward_zero_f64() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3"); }

constexpr uint64_t EXP_HI = 0x467ull << 52;              // exponent field of 2^104
constexpr uint64_t EXP_LO = 0x433ull << 52;              // exponent field of 2^52
constexpr uint64_t M52 = (1ull << 52) - 1;

// limb product into two integer columns: 2 FMA + 1 subtraction + 2 64-bit additions = 5 issue slots
__device__ __forceinline__ void mac52(uint64_t &col_lo, uint64_t &col_hi, double a, double b) {
    const double C1 = 0x1p104, C3 = 0x1p104 + 0x1p52;
    const double hi = fma_rz(a, b, C1);
    const double lo = fma_rz(a, b, sub_f64(C3, hi));
    col_hi += (uint64_t)__double_as_longlong(hi);
    col_lo += (uint64_t)__double_as_longlong(lo);
}
// the low 52 bits of an integer column as a double
__device__ __forceinline__ double low52_as_double(uint64_t col) {
    return sub_f64(__longlong_as_double((long long)((col & M52) | EXP_LO)), 0x1p52);
}

__device__ __forceinline__ fe52_t fe52_mul(const fe52_t &a, const fe52_t &b, const fe52_consts &k) {
    // columns 0..10; column c receives nlo[c] low halves and nhi[c] high halves: start each at minus their exponent fields (mod 2^64)
    // products: lo of (i, j) -> column i + j, hi -> column i + j + 1.  reduction round i: lo of q_i * p_j -> i + j, hi -> i + j + 1 (j = 0, 1, 2); the low product of q.
    uint64_t t[11];
#pragma unroll
    for (int c = 0; c < 11; ++c) {
        int nlo = 0, nhi = 0;
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) { if (i + j == c) ++nlo; if (i + j + 1 == c) ++nhi; }
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 3; ++j) { if (i + j == c) ++nlo; if (i + j + 1 == c) ++nhi; }
        t[c] = 0 - ((uint64_t)nlo * EXP_LO + (uint64_t)nhi * EXP_HI);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) mac52(t[i + j], t[i + j + 1], a.v[i], b.v[j]);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        // q = low 52 bits of (t[i] mod 2^52) * (-1/p)
        const double C1 = 0x1p104, C3 = 0x1p104 + 0x1p52;
        const double ti = low52_as_double(t[i]);
        const double qh = fma_rz(ti, k.pinv, C1);
        const double ql = fma_rz(ti, k.pinv, sub_f64(C3, qh));     // 2^52 + q: the mantissa field is q
        const double q = sub_f64(ql, 0x1p52);
        mac52(t[i], t[i + 1], q, k.p[0]);
        mac52(t[i + 1], t[i + 2], q, k.p[1]);
        mac52(t[i + 2], t[i + 3], q, k.p[2]);
        // q * 2^46 at limb i + 4: (q mod 2^6) * 2^46 into column i + 4, q >> 6 into column i + 5
        const uint64_t qi = (uint64_t)__double_as_longlong(ql) & M52;
        t[i + 4] += (qi & 63) << 46;
        t[i + 5] += qi >> 6;
        t[i + 1] += t[i] >> 52;                            // column i is now 0 mod 2^52: its carry moves up
    }
    fe52_t r;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        r.v[i] = low52_as_double(t[5 + i]);
        t[6 + i] += t[5 + i] >> 52;
    }
    return r;                                              // < 2p (no conditional subtraction: the 29-bit form has none either)
}

__device__ __forceinline__ fe52_t fe52_from_words(const uint32_t *w) {      // 8 x 32 -> 5 x 52
    fe52_t r;
    uint64_t x[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 5; ++i) {
        const int bit = 52 * i;
        uint64_t v = 0;
        for (int k = 0; k < 3; ++k) { const int wi = bit / 32 + k; if (wi < 8) { const int sh = 32 * k - (bit % 32); v |= sh >= 0 ? ((uint64_t)w[wi] << sh) : ((uint64_t)w[wi] >> -sh); } }
        x[i] = v & M52;
        r.v[i] = (double)x[i];
    }
    return r;
}
__device__ __forceinline__ void fe52_to_words(const fe52_t &a, uint32_t *w) {
    uint64_t x[5]; for (int i = 0; i < 5; ++i) x[i] = (uint64_t)a.v[i];
    for (int i = 0; i < 9; ++i) w[i] = 0;
    for (int i = 0; i < 5; ++i) {
        const int bit = 52 * i, wi = bit / 32, sh = bit % 32;
        const unsigned __int128 v = (unsigned __int128)x[i] << sh;
        w[wi] |= (uint32_t)v; w[wi + 1] |= (uint32_t)(v >> 32); if (wi + 2 < 9) w[wi + 2] |= (uint32_t)(v >> 64);
    }
}

// ---- (1) values for the Python check: n triples (a, b, fe52_mul(a, b)); words in, 9 words out (the result may exceed 2^256 by less than p)
__global__ void fe52_values(uint32_t n, fe52_consts k, const uint32_t *a, const uint32_t *b, uint32_t *r) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe52_t x = fe52_from_words(a + 8 * i), y = fe52_from_words(b + 8 * i);
    set_round_toward_zero_f64();
    fe52_t z = fe52_mul(x, y, k);
    fe52_to_words(z, r + 9 * i);
}

// ---- (2) throughput of dependent chains
constexpr int ITERS = 256;
__global__ void __launch_bounds__(256) fe52_chain(fe52_consts k, uint32_t *out, uint32_t seed) {
    uint32_t w[8]; for (int i = 0; i < 8; ++i) w[i] = seed * (i + 1) + threadIdx.x; w[7] &= 0x3fffffffu;
    fe52_t x = fe52_from_words(w); w[0] ^= 0x5a5a5a5au; fe52_t y = fe52_from_words(w);
    set_round_toward_zero_f64();
    asm volatile("s_nop 7");                               // ISA marker: the loop body between the two markers is one product
    for (int it = 0; it < ITERS; ++it) x = fe52_mul(x, y, k);
    asm volatile("s_nop 6");
    uint32_t o[9]; fe52_to_words(x, o);
    uint32_t acc = 0; for (int i = 0; i < 9; ++i) acc ^= o[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) fe29_chain(uint32_t *out, uint32_t seed) {
#if defined(__HIP_DEVICE_COMPILE__)
    fe29_t x, y; for (int i = 0; i < 9; ++i) { x.v[i] = (seed * (i + 1) + threadIdx.x) & M29; y.v[i] = (seed ^ (0x85ebca6bu * (i + 2))) & M29; }
    for (int it = 0; it < ITERS; ++it) x = fe29_mul_asm<FIELD_FQ>(x, y);
    uint32_t r = 0; for (int i = 0; i < 9; ++i) r ^= x.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
#endif
}
// pure instruction streams at the same occupancies: v_fma_f64, v_add_f64, the 64-bit integer add, v_mad_u64_u32 (operands on distinct register banks)
template <int OP> __global__ void __launch_bounds__(256) stream(uint32_t *out, uint32_t seed) {
    double d[16], e[16]; uint64_t u[16]; uint32_t a[16], b[16];
    for (int i = 0; i < 16; ++i) { d[i] = (double)(seed + i + threadIdx.x); e[i] = (double)(seed ^ (i * 77u)); u[i] = seed * (i + 3); a[i] = seed + i; b[i] = seed ^ (i * 0x9e3779b9u); }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(e[(i + 5) % 16]), "v"(e[(i + 9) % 16]));
            if (OP == 1) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(e[(i + 5) % 16]));
            if (OP == 2) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(u[i]) : "v"(u[(i + 5) % 16]));
            if (OP == 3) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(u[i]) : "v"(a[(i + 1) % 16]), "v"(b[(i + 6) % 16]) : "s20", "s21");
            if (OP == 4) {                                   // the limb product's own mix: fma, sub, fma, add64, add64
                double h, l;
                asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(h) : "v"(e[(i + 5) % 16]), "v"(e[(i + 9) % 16]), "v"(d[i]));
                asm volatile("v_add_f64 %0, %1, -%2" : "=v"(l) : "v"(d[(i + 3) % 16]), "v"(h));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(l) : "v"(e[(i + 5) % 16]), "v"(e[(i + 9) % 16]));
                asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(u[i]) : "v"(h));
                asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(u[(i + 7) % 16]) : "v"(l));
            }
        }
    }
    uint32_t r = 0; for (int i = 0; i < 16; ++i) r ^= (uint32_t)d[i] ^ (uint32_t)u[i] ^ (uint32_t)(u[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <class K> static double time_kernel(K launch, int reps) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3 / reps;
}

static void limbs52_of(const uint64_t w[4], double out[5]) {            // 4 x 64 -> 5 x 52
    for (int i = 0; i < 5; ++i) {
        const int bit = 52 * i, wi = bit / 64, sh = bit % 64;
        unsigned __int128 v = w[wi]; if (wi + 1 < 4) v |= (unsigned __int128)w[wi + 1] << 64;
        out[i] = (double)((uint64_t)(v >> sh) & M52);
    }
}
static uint64_t neg_inv52(uint64_t p0) {                                // -1/p mod 2^52 (Newton)
    uint64_t x = 1; for (int i = 0; i < 6; ++i) x *= 2 - p0 * x;
    return (0 - x) & M52;
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f}\n", prop.gcnArchName, cus, clk / 1e6);
    // Pasta moduli: p (Pallas base), q (Vesta base)
    const uint64_t MOD[2][4] = {{0x992d30ed00000001ull, 0x224698fc094cf91bull, 0, 0x4000000000000000ull}, {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0, 0x4000000000000000ull}};
    const int n = 4096;
    std::vector<uint32_t> ha(8 * n), hb(8 * n), hr(9 * n);
    uint32_t *da, *db, *dr, *out; CHECK(hipMalloc(&da, ha.size() * 4)); CHECK(hipMalloc(&db, hb.size() * 4)); CHECK(hipMalloc(&dr, hr.size() * 4));
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    fe52_consts kc[2];
    for (int f = 0; f < 2; ++f) {
        double pl[5]; limbs52_of(MOD[f], pl);
        kc[f].p[0] = pl[0]; kc[f].p[1] = pl[1]; kc[f].p[2] = pl[2]; kc[f].pinv = (double)neg_inv52(MOD[f][0] & M52);
        if (pl[3] != 0 || pl[4] != (double)(1ull << 46)) { fprintf(stderr, "unexpected limb structure of p\n"); return 1; }
        uint64_t s = 0x6d696e6162726467ull + f;
        auto next = [&]() { s += 0x9e3779b97f4a7c15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
        for (int i = 0; i < n; ++i) for (int w = 0; w < 8; w += 2) {
            uint64_t x = next(), y = next();
            if (i < 8) { x = i & 1 ? ~0ull : 0; y = i & 2 ? ~0ull : (i & 4 ? 1 : 0); }         // edge operands: all-ones limbs, zero, one
            ha[8 * i + w] = (uint32_t)x; ha[8 * i + w + 1] = (uint32_t)(x >> 32); hb[8 * i + w] = (uint32_t)y; hb[8 * i + w + 1] = (uint32_t)(y >> 32);
        }
        CHECK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        fe52_values<<<(n + 255) / 256, 256>>>(n, kc[f], da, db, dr);
        CHECK(hipMemcpy(hr.data(), dr, hr.size() * 4, hipMemcpyDeviceToHost));
        if (argc > 1 && !strcmp(argv[1], "--values")) {
            for (int i = 0; i < n; ++i) {
                printf("{\"field\": %d, \"a\": \"", f); for (int w = 7; w >= 0; --w) printf("%08x", ha[8 * i + w]);
                printf("\", \"b\": \""); for (int w = 7; w >= 0; --w) printf("%08x", hb[8 * i + w]);
                printf("\", \"r\": \""); for (int w = 8; w >= 0; --w) printf("%08x", hr[9 * i + w]);
                printf("\"}\n");
            }
        }
    }
    if (argc > 1 && !strcmp(argv[1], "--values")) return 0;
    const char *names[] = {"v_fma_f64", "v_add_f64", "v_lshl_add_u64", "v_mad_u64_u32", "limb-product mix: fma, sub, fma, add64, add64"};
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = cus * wps;
#define STREAM(OP, PER) { double t = time_kernel([&] { stream<OP><<<blocks, 256>>>(out, 12345u); }, 5); \
        printf("{\"probe\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr\": %.2f, \"ns_per_wave_instr_per_simd\": %.3f}\n", names[OP], wps, t * clk / ((double)ITERS * 16 * PER * wps), t * 1e9 / ((double)ITERS * 16 * PER * wps)); }
        STREAM(0, 1) STREAM(1, 1) STREAM(2, 1) STREAM(3, 1) STREAM(4, 5)
        const double t52 = time_kernel([&] { fe52_chain<<<blocks, 256>>>(kc[1], out, 777u); }, 3), t29 = time_kernel([&] { fe29_chain<<<blocks, 256>>>(out, 777u); }, 3);
        printf("{\"probe\": \"fe52_mul chain (5 x 52, FP64 FMA pairs)\", \"waves_per_simd\": %d, \"ns_per_product_per_wave\": %.1f, \"chip_Gproducts_per_s\": %.1f}\n", wps, t52 * 1e9 / ITERS / wps, (double)blocks * 256 * ITERS / t52 / 1e9);
        printf("{\"probe\": \"fe29_mul_asm chain (9 x 29, v_mad_u64_u32)\", \"waves_per_simd\": %d, \"ns_per_product_per_wave\": %.1f, \"chip_Gproducts_per_s\": %.1f}\n", wps, t29 * 1e9 / ITERS / wps, (double)blocks * 256 * ITERS / t29 / 1e9);
    }
    return 0;
}
