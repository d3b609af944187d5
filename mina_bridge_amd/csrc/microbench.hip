// microbench.hip -- gfx950 VALU instruction-rate probe for 256-bit modular arithmetic.
// Measures wave-instruction issue cost (cycles per wave64 instruction per SIMD) of the candidate
// building blocks: v_mad_u64_u32, v_mul_lo_u32, v_mul_hi_u32, v_mad_u32_u24, v_fma_f64, v_add_co chains,
// plus the library's fe_mul / fe_sqr / fe_add and the XYZZ mixed add.  Output: one JSON line per probe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>
#include "ec.cuh"
#include "fp29.cuh"

using namespace mb;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 256;     // loop trips
constexpr int UNROLL = 16;     // independent chains per trip body (ILP)

template <int OP>
__global__ void __launch_bounds__(256) probe(uint32_t *out, uint32_t seed) {
    uint32_t a[UNROLL], b[UNROLL];
    uint64_t acc[UNROLL];
    double d[UNROLL];
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed * (i + 3) + threadIdx.x; b[i] = seed ^ (0x9e3779b9u * (i + 1)); acc[i] = a[i]; d[i] = (double)a[i]; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) % UNROLL]));
            if (OP == 5) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(a[i]), "+v"(b[i]), "+v"(a[(i + 1) % UNROLL]) : : "vcc");
            if (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 8) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 9) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(acc[i]));
            if (OP == 10) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(acc[i]), "+v"(a[i]) : "v"(a[(i + 1) % UNROLL]), "v"(b[i]) : "vcc");
            // the carry-free form of fp29.cuh: the carry-out goes to a scratch SGPR pair nobody reads (one pair / four pairs in rotation)
            if (OP == 11) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]) : "s20", "s21");
            if (OP == 12) {
                if ((i & 3) == 0) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]) : "s20", "s21");
                if ((i & 3) == 1) asm volatile("v_mad_u64_u32 %0, s[22:23], %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]) : "s22", "s23");
                if ((i & 3) == 2) asm volatile("v_mad_u64_u32 %0, s[24:25], %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]) : "s24", "s25");
                if ((i & 3) == 3) asm volatile("v_mad_u64_u32 %0, s[26:27], %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]) : "s26", "s27");
            }
            if (OP == 13) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[0]) : "v"(a[i]), "v"(b[i]) : "s20", "s21");      // ONE accumulator: the dependent column chain
            if (OP == 14) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[i]));
            if (OP == 15) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(a[i]));
        }
        // OP 16: the instruction mix of one lane-round of the 29-bit-limb Poseidon permutation (fp29.cuh): 765 multiply-accumulates and ~290 simple
        // instructions (64-bit column shifts, limb masks), interleaved as the generated routines interleave them (a carry after every ~2.6 products).
        // Does the mix issue at the multiply-accumulate rate alone (the simple instructions hide in its shadow) or does every instruction take its slot?
        if (OP == 16) {
#pragma unroll
            for (int j = 0; j < 153; ++j) {
                asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[j % UNROLL]) : "v"(a[j % UNROLL]), "v"(b[(j + 5) % UNROLL]) : "s20", "s21");
                if ((j * 58) / 153 != ((j + 1) * 58) / 153) {
                    if (j & 1) asm volatile("v_lshrrev_b64 %0, 29, %1" : "=v"(acc[(j + 7) % UNROLL]) : "v"(acc[(j + 8) % UNROLL]));
                    else asm volatile("v_and_b32 %0, 0x1fffffff, %1" : "=v"(a[(j + 3) % UNROLL]) : "v"(b[(j + 9) % UNROLL]));
                }
            }
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < UNROLL; ++i) r ^= a[i] ^ b[i] ^ (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32) ^ (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// ---- how many simple VALU instructions hide behind a multiply-accumulate?  153 v_mad_u64_u32 per trip (16 independent accumulators) with S simple
// instructions spread evenly between them: 64-bit shifts, masks, 32-bit shifts and add / add-with-carry pairs in rotation (the instructions a
// Karatsuba recombination or a shift-add form of the reduction's power-of-two limb would add).  KIND 1: only add_co / addc_co pairs.
template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int S, int KIND>
__global__ void __launch_bounds__(256) probe_ratio(uint32_t *out, uint32_t seed) {
    uint32_t a[UNROLL], b[UNROLL];
    uint64_t acc[UNROLL];
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed * (i + 3) + threadIdx.x; b[i] = seed ^ (0x9e3779b9u * (i + 1)); acc[i] = a[i]; }
    for (int it = 0; it < ITERS; ++it) {
        static_for<0, 153 + S>([&](auto tc) {              // one flat schedule resolved at compile time: slot t is a multiply-accumulate when the running count of them steps
            constexpr int t = decltype(tc)::value, j = (t * 153) / (153 + S), e = t - j;
            if constexpr (((t + 1) * 153) / (153 + S) != j) {
                asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[j % UNROLL]) : "v"(a[j % UNROLL]), "v"(b[(j + 5) % UNROLL]) : "s20", "s21");
            } else {
                // KIND 2: the simple instructions of one lane-round of the LAZY 3-lane permutation in their proportions -- per 774 multiply-accumulates 80 64-bit
                // shifts, 45 negations, 40 masks, 17 + 5 32-bit shifts: 37 per 153 (16 shr64 = kind 0, 9 sub = kind 5, 8 and = kind 1, 4 shl = kind 4)
                constexpr int e37 = e % 37;
                constexpr int kind2 = e37 >= 32 ? (e37 == 33 ? 5 : 4) : (e37 % 2 == 0 ? 0 : (e37 % 4 == 1 ? 5 : 1));
                // KIND 3: the SIGNED-digit 3-lane round (late round 5): per 774 multiply-accumulates 80 64-bit shifts, 40 masks, 5 and-or (the one digit a product still
                // makes), 17 + 5 32-bit shifts -- 29 per 153 (16 shr64 = kind 0, 9 and = kind 1, 4 shl = kind 4); the quotient digits' 45 negations are gone
                constexpr int k3tab[29] = {0, 1, 0, 4, 0, 1, 0, 1, 0, 4, 0, 1, 0, 1, 0, 4, 0, 1, 0, 1, 0, 4, 0, 1, 0, 1, 0, 0, 0};
                constexpr int kind3 = k3tab[e % 29];
                constexpr int kind = KIND == 3 ? kind3 : KIND == 2 ? kind2 : KIND == 1 ? 2 + (e & 1) : e % 5;
                if constexpr (kind == 5) asm volatile("v_sub_u32 %0, 0, %1" : "=v"(a[(j + 14) % UNROLL]) : "v"(b[(j + 3) % UNROLL]));
                if constexpr (kind == 0) asm volatile("v_lshrrev_b64 %0, 29, %1" : "=v"(acc[(j + 7) % UNROLL]) : "v"(acc[(j + 8) % UNROLL]));
                if constexpr (kind == 1) asm volatile("v_and_b32 %0, 0x1fffffff, %1" : "=v"(a[(j + 3) % UNROLL]) : "v"(b[(j + 9) % UNROLL]));
                if constexpr (kind == 2) asm volatile("v_add_co_u32 %0, vcc, %1, %2" : "=v"(a[(j + 4) % UNROLL]) : "v"(a[(j + 11) % UNROLL]), "v"(b[(j + 2) % UNROLL]) : "vcc");
                if constexpr (kind == 3) asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(b[(j + 6) % UNROLL]) : "v"(a[(j + 12) % UNROLL]), "v"(b[(j + 13) % UNROLL]) : "vcc");
                if constexpr (kind == 4) asm volatile("v_lshlrev_b32 %0, 22, %1" : "=v"(a[(j + 10) % UNROLL]) : "v"(b[(j + 1) % UNROLL]));
            }
        });
    }
    uint32_t r = 0;
    for (int i = 0; i < UNROLL; ++i) r ^= a[i] ^ b[i] ^ (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// ---- the latency regime (one wave per SIMD): how fast does ONE wave issue multiply-accumulates that chain through the same accumulator, and how much do
// independent accumulators help?  153 v_mad_u64_u32 per trip over NACC accumulators in rotation (NACC = 1: every one waits for the one before).
// SIMPLE > 0: a dependent simple instruction (the column's mask) after every SIMPLE-th multiply-accumulate, as the field routines have.
template <int NACC, int SIMPLE>
__global__ void __launch_bounds__(256) probe_dep(uint32_t *out, uint32_t seed) {
    uint32_t a[UNROLL], b[UNROLL];
    uint64_t acc[UNROLL];
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed * (i + 3) + threadIdx.x; b[i] = seed ^ (0x9e3779b9u * (i + 1)); acc[i] = a[i]; }
    for (int it = 0; it < ITERS; ++it) {
        static_for<0, 153>([&](auto tc) {
            constexpr int j = decltype(tc)::value, k = j % NACC;
            asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[k]) : "v"(a[(j + 1) % UNROLL]), "v"(b[(j + 6) % UNROLL]) : "s20", "s21");
            if constexpr (SIMPLE > 0 && j % (SIMPLE > 0 ? SIMPLE : 1) == SIMPLE - 1)
                asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[k]));
        });
    }
    uint32_t r = 0;
    for (int i = 0; i < UNROLL; ++i) r ^= a[i] ^ b[i] ^ (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
__global__ void __launch_bounds__(256) probe_field(uint32_t *out, uint32_t seed) {
    fe_t x, y;
    for (int i = 0; i < 8; ++i) { x.v[i] = seed * (i + 1) + threadIdx.x; y.v[i] = seed ^ (0x85ebca6bu * (i + 2)); }
    x.v[7] &= 0x3fffffffu; y.v[7] &= 0x3fffffffu;
    fe_t one = fe_zero(); one.v[0] = 1;
    xyzz_t acc; acc.x = x; acc.y = y; acc.zz = one; acc.zzz = one;
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) x = fe_mul<FIELD_FQ>(x, y);
        if (OP == 1) x = fe_sqr<FIELD_FQ>(x);
        if (OP == 2) x = fe_add<FIELD_FQ>(x, y);
        if (OP == 3) x = fe_sub<FIELD_FQ>(x, y);
        if (OP == 4) { xyzz_add_affine<FIELD_FQ>(acc, x, y, one); }
        if (OP == 5) { x = fe_mul<FIELD_FQ>(x, y); y = fe_mul<FIELD_FQ>(y, y); }   // two independent products per trip
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= x.v[i] ^ y.v[i] ^ acc.x.v[i] ^ acc.y.v[i] ^ acc.zz.v[i] ^ acc.zzz.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// ---- single-wave LATENCY probes (round 4, VERDICT r03 item 7: one proof per call is a dependent chain of ~300 permutations x 55 rounds; a round's chain is
// 99 + 3 x 135 dependent multiply-accumulates of ONE wave).  Would splitting the 81 limb products of a 29-bit Montgomery product over the 9 lanes of a DPP row
// shorten the chain enough?  (a) the product as it is, dependent chain, one wave per SIMD; (b) the INSTRUCTION SKELETON of a lane-split product -- the same
// count and dependency shape of DPP broadcasts, multiply-accumulates, DPP-shifted 64-bit column adds, relaxed carry passes, the quotient product by -1/p, the
// product by the sparse p, the final carry resolution and the move of the high half back to lanes 0..8 -- values are NOT a correct product: a lower bound on time.
template <int OP>
__global__ void __launch_bounds__(256) probe_latency(uint32_t *out, uint32_t seed) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (OP == 0) {
        fe29_t x, y; for (int i = 0; i < 9; ++i) { x.v[i] = (seed * (i + 1) + threadIdx.x) & M29; y.v[i] = (seed ^ (0x85ebca6bu * (i + 2))) & M29; }
        for (int it = 0; it < ITERS; ++it) x = fe29_mul_asm<FIELD_FQ>(x, y);
        uint32_t r = 0; for (int i = 0; i < 9; ++i) r ^= x.v[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    } else {
        uint32_t x = seed + threadIdx.x, y = seed ^ threadIdx.x;
        const uint32_t pc = 0x1234567u ^ (threadIdx.x & 15u);
#define BCAST(v, n) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x150 + (n), 0xf, 0xf, false)
#define SHR(v, n) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x110 + (n), 0xf, 0xf, true)
        for (int it = 0; it < ITERS; ++it) {
            uint64_t q[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) { uint32_t yj = 0; switch (j) { case 0: yj = BCAST(y, 0); break; case 1: yj = BCAST(y, 1); break; case 2: yj = BCAST(y, 2); break; case 3: yj = BCAST(y, 3); break;
                                            case 4: yj = BCAST(y, 4); break; case 5: yj = BCAST(y, 5); break; case 6: yj = BCAST(y, 6); break; case 7: yj = BCAST(y, 7); break; default: yj = BCAST(y, 8); }
                                          q[j] = (uint64_t)x * yj; }
            uint64_t acc = q[0];
#define ADDSHR(n) { const uint32_t lo = SHR((uint32_t)q[n], n), hi = SHR((uint32_t)(q[n] >> 32), n); acc += ((uint64_t)hi << 32) | lo; }
            ADDSHR(1) ADDSHR(2) ADDSHR(3) ADDSHR(4) ADDSHR(5) ADDSHR(6) ADDSHR(7) ADDSHR(8)
#define RELAX() { const uint64_t hi = acc >> 29; acc = (acc & M29) + (((uint64_t)SHR((uint32_t)(hi >> 32), 1) << 32) | SHR((uint32_t)hi, 1)); }
            RELAX() RELAX()
            // m = T_low * (-1/p) mod 2^261: 9 products by per-lane constants, shifted column adds, relaxed carries
            const uint32_t tl = (uint32_t)acc & M29;
#pragma unroll
            for (int j = 0; j < 9; ++j) q[j] = (uint64_t)tl * (pc + j);
            uint64_t m = q[0];
#undef ADDSHR
#define ADDSHR(n) { const uint32_t lo = SHR((uint32_t)q[n], n), hi = SHR((uint32_t)(q[n] >> 32), n); m += ((uint64_t)hi << 32) | lo; }
            ADDSHR(1) ADDSHR(2) ADDSHR(3) ADDSHR(4) ADDSHR(5) ADDSHR(6) ADDSHR(7) ADDSHR(8)
            { const uint64_t hi = m >> 29; m = (m & M29) + (((uint64_t)SHR((uint32_t)(hi >> 32), 1) << 32) | SHR((uint32_t)hi, 1)); }
            { const uint64_t hi = m >> 29; m = (m & M29) + (((uint64_t)SHR((uint32_t)(hi >> 32), 1) << 32) | SHR((uint32_t)hi, 1)); }
            // T += m * p (6 non-zero limbs of p)
            const uint32_t ml = (uint32_t)m & M29;
#pragma unroll
            for (int j = 0; j < 6; ++j) q[j] = (uint64_t)ml * (pc ^ j);
#undef ADDSHR
#define ADDSHR(n) { const uint32_t lo = SHR((uint32_t)q[n], n), hi = SHR((uint32_t)(q[n] >> 32), n); acc += ((uint64_t)hi << 32) | lo; }
            acc += q[0]; ADDSHR(1) ADDSHR(2) ADDSHR(3) ADDSHR(4) ADDSHR(8)
            // carry out of the low half: generate / propagate resolution over the row (4 steps), then the high half moves down 9 lanes, relaxed carries
            uint32_t g = (uint32_t)(acc >> 29) != 0, pr = ((uint32_t)acc & M29) == M29;
#pragma unroll
            for (int d = 1; d <= 8; d <<= 1) { const uint32_t g2 = SHR(g, 1), p2 = SHR(pr, 1); g |= pr & g2; pr &= p2; }
            const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)((uint32_t)acc + g), 0x100 + 9, 0xf, 0xf, true), ny = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(acc >> 32), 0x100 + 9, 0xf, 0xf, true);
            acc = ((uint64_t)ny << 32) | nx;
            RELAX() RELAX()
            x = (uint32_t)acc & M29;                             // the next product depends on this one
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    }
#endif
}

static double wall_now() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

template <class K>
static double time_kernel(K launch, int reps) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3 / reps;
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;   // Hz
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f}\n", prop.gcnArchName, cus, clk / 1e6);
    uint32_t *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    if (argc > 1 && !strcmp(argv[1], "--dep")) {          // one or two waves per SIMD: cycles per multiply-accumulate of ONE wave against the number of independent accumulators
        for (int wps = 1; wps <= 2; ++wps) {
            const int blocks = cus * wps;
#define DEP(NACC, SIMPLE)                                                                                    \
            {                                                                                                \
                double t = time_kernel([&] { probe_dep<NACC, SIMPLE><<<blocks, 256>>>(out, 12345u); }, 5);   \
                printf("{\"probe\": \"153 v_mad_u64_u32 per trip over %d accumulator(s)%s\", \"waves_per_simd\": %d, \"cycles_per_mac_per_wave\": %.2f}\n", \
                       NACC, SIMPLE ? ", a dependent 64-bit shift after every 8th" : "", wps, t * clk / ((double)ITERS * 153));   \
            }
            DEP(1, 0) DEP(2, 0) DEP(3, 0) DEP(4, 0) DEP(8, 0) DEP(1, 8) DEP(2, 8)
        }
        CHECK(hipFree(out));
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "--sustain")) {     // the same two streams held for SECONDS each (default 6), rate per ~0.5 s window: the probes above are bursts of
        // 0.1 - 0.5 ms, shorter than the power controller's averaging; the bench holds the chip at its 1400 W cap for seconds (profiles/r05_clock_power.md)
        const double secs = argc > 2 ? atof(argv[2]) : 6.0;
        const int wps = 8, blocks = cus * wps;
#define SUSTAIN(S, KIND, LABEL)                                                                              \
        {                                                                                                    \
            double one = time_kernel([&] { probe_ratio<S, KIND><<<blocks, 256>>>(out, 12345u); }, 5);        \
            const int reps = (int)(0.5 / one) + 1;                                                           \
            for (double done = 0; done < secs;) {                                                            \
                double t = time_kernel([&] { probe_ratio<S, KIND><<<blocks, 256>>>(out, 12345u); }, reps);   \
                done += t * reps;                                                                            \
                printf("{\"probe\": \"sustained: %s\", \"waves_per_simd\": %d, \"t_s\": %.2f, \"wall\": %.3f, \"nominal_cycles_per_mac\": %.3f, \"nominal_cycles_per_instr\": %.3f, \"T_mac_per_s\": %.2f}\n", \
                       LABEL, wps, done, wall_now(), t * clk / ((double)ITERS * 153 * wps), t * clk / ((double)ITERS * (153 + S) * wps), (double)ITERS * 153 * blocks * 256 / t / 1e12); \
                fflush(stdout);                                                                              \
            }                                                                                                \
        }
        SUSTAIN(0, 0, "pure v_mad_u64_u32")
        SUSTAIN(29, 3, "the signed-digit 3-lane round's mix (153 + 29)")
        CHECK(hipFree(out));
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "--ratio")) {       // the sweep alone: cycles per multiply-accumulate against simple instructions per multiply-accumulate
        for (int wps = 4; wps <= 8; wps *= 2) {
            const int blocks = cus * wps;
#define RATIO(S, KIND)                                                                                       \
            {                                                                                                \
                double t = time_kernel([&] { probe_ratio<S, KIND><<<blocks, 256>>>(out, 12345u); }, 5);      \
                printf("{\"probe\": \"153 v_mad_u64_u32 + %d simple per trip (%s)\", \"waves_per_simd\": %d, \"simple_per_mac\": %.2f, \"cycles_per_mac\": %.2f, \"cycles_per_instr\": %.2f}\n", \
                       S, KIND == 3 ? "the signed-digit 3-lane round's mix: 16 shr64, 9 and, 4 shl32" : KIND == 2 ? "the lazy 3-lane round's mix: 16 shr64, 9 sub, 8 and, 4 shl32" : KIND ? "add_co / addc_co pairs" : "shr64, and, add_co, addc_co, shl32 in rotation", wps, S / 153.0, t * clk / ((double)ITERS * 153 * wps), t * clk / ((double)ITERS * (153 + S) * wps)); \
            }
            RATIO(0, 0) RATIO(58, 0) RATIO(77, 0) RATIO(115, 0) RATIO(153, 0) RATIO(191, 0) RATIO(230, 0) RATIO(306, 0) RATIO(459, 0)
            RATIO(77, 1) RATIO(153, 1) RATIO(230, 1) RATIO(306, 1)
            RATIO(37, 2)                                                   // the lazy 3-lane round's own mix (until late in round 5)
            RATIO(29, 3)                                                   // the signed-digit round's own mix: what bench.py prices a multiply-accumulate at
        }
        CHECK(hipFree(out));
        return 0;
    }
    for (int waves_per_simd = 2; waves_per_simd <= 8; waves_per_simd *= 2) {    // 2 waves/SIMD already cover VALU latency with UNROLL=16; more must not change the rates
    printf("{\"waves_per_simd\": %d}\n", waves_per_simd);
    const int blocks = cus * waves_per_simd;          // 256-thread blocks: 4 waves -> one per SIMD
    const char *names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_fma_f64", "add_co+addc_co(2 instr)",
                           "v_add_u32", "v_mul_u32_u24", "v_mul_hi_u32_u24", "v_lshlrev_b64", "mad_u64_u32+addc(2 instr)",
                           "v_mad_u64_u32 (carry to a scratch sgpr pair)", "v_mad_u64_u32 (4 scratch pairs in rotation)", "v_mad_u64_u32 (one accumulator: dependent chain)",
                           "v_lshrrev_b64", "v_and_b32 (literal)", "fp29 round mix: 153 v_mad_u64_u32 + 29 v_lshrrev_b64 + 29 v_and_b32 per trip"};
#define RUN(OP)                                                                                              \
    {                                                                                                        \
        double t = time_kernel([&] { probe<OP><<<blocks, 256>>>(out, 12345u); }, 5);                         \
        double wave_instr_per_simd = (double)ITERS * UNROLL * waves_per_simd;                                \
        printf("{\"probe\": \"%s\", \"cycles_per_wave_instr_per_simd\": %.2f, \"time_us\": %.1f}\n", names[OP], \
               t * clk / wave_instr_per_simd, t * 1e6);                                                      \
    }
    if (waves_per_simd == 2) { RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(7) RUN(8) RUN(9) }
    RUN(0) RUN(6) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15)
    {   // the mix: cycles per multiply-accumulate OF THE MIX (153 per trip) and per instruction (211 per trip)
        double t = time_kernel([&] { probe<16><<<blocks, 256>>>(out, 12345u); }, 5);
        const double macs = (double)ITERS * 153 * waves_per_simd, instr = (double)ITERS * 211 * waves_per_simd;
        printf("{\"probe\": \"%s\", \"cycles_per_mac_of_the_mix\": %.2f, \"cycles_per_instr_of_the_mix\": %.2f, \"time_us\": %.1f}\n", names[16], t * clk / macs, t * clk / instr, t * 1e6);
    }
    }
    const char *fnames[] = {"fe_mul (dependent)", "fe_sqr (dependent)", "fe_add", "fe_sub", "xyzz_add_affine", "2x fe_mul (independent)"};
    for (int wps = 1; wps <= 3; ++wps) {
        const int fb = cus * wps;
#define RUNF(OP)                                                                                             \
    {                                                                                                        \
        double t = time_kernel([&] { probe_field<OP><<<fb, 256>>>(out, 777u); }, 3);                         \
        printf("{\"probe\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_op_per_wave\": %.0f, \"ns_per_op_latency\": %.0f, \"chip_Gop_per_s\": %.1f}\n", \
               fnames[OP], wps, t * clk / ITERS / wps, t * 1e9 / ITERS, (double)fb * 256 * ITERS / t / 1e9); \
    }
        RUNF(0) RUNF(1) RUNF(2) RUNF(3) RUNF(4) RUNF(5)
    }
    for (int wps = 1; wps <= 2; ++wps) {
        const int fb = cus * wps;
        const double t0 = time_kernel([&] { probe_latency<0><<<fb, 256>>>(out, 777u); }, 3), t1 = time_kernel([&] { probe_latency<1><<<fb, 256>>>(out, 777u); }, 3);
        printf("{\"probe\": \"fe29_mul_asm (dependent chain, 135 MAC + ~43 simple)\", \"waves_per_simd\": %d, \"cycles_per_product\": %.0f, \"ns_per_product\": %.0f}\n", wps, t0 * clk / ITERS / wps, t0 * 1e9 / ITERS);
        printf("{\"probe\": \"lane-split 29-bit Montgomery product over a DPP row, instruction skeleton (24 MAC + ~120 DPP / simple; a LOWER bound)\", \"waves_per_simd\": %d, \"cycles_per_product\": %.0f, \"ns_per_product\": %.0f}\n", wps, t1 * clk / ITERS / wps, t1 * 1e9 / ITERS);
    }
    CHECK(hipFree(out));
    return 0;
}
