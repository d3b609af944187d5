// sg_probe.hip -- signed quotient digits for the 9 x 29-bit Montgomery products (round 5, late): the lazy forms spend one v_sub per quotient digit (m_k = -col mod 2^32)
// and the strict forms a v_sub and a v_and.  With SIGNED digits -- s_k = (int32) low word of the column, subtracted by v_mad_i64_i32 against the NEGATED prime limbs -- the
// digit is the column's own low register: no instruction.  Digit 8 = (col & M29) - 2^30 (one v_and_or) carries a built-in offset: the result is
// T / R + (1 p, 2 p] -+ 2^-27 p -- positive, and tighter than the lazy forms' + 8 p.  This probe: the two forms side by side on one dependent chain (x <- x^3 / R^2),
// values for the host check (tools/probes/sg_check.py) and rates at 2 / 5 / 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mina_bridge_amd/csrc -o tools/probes/bin/sg_probe tools/probes/sg_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
#include "fp.cuh"
#include "fp29.cuh"
using namespace mb;

#include <atomic>
#include <thread>
// socket power / shader clock from amdgpu's hwmon files, sampled every 10 ms by a host thread WHILE the kernels run (a read after the last kernel has ended sees an idling chip)
struct HwmonSampler {
    char pw[512] = "", fq[512] = "";
    std::atomic<bool> stop{false}; std::atomic<double> sw{0}, sf{0}; std::atomic<int> n{0}; std::thread th;
    static double rd(const char *path) { double v = 0; FILE *f = path[0] ? fopen(path, "r") : nullptr; if (f) { if (fscanf(f, "%lf", &v) != 1) v = 0; fclose(f); } return v; }
    HwmonSampler() {
        char bus[64] = "", cmd[256];                         // the hwmon directory of THE device the kernels run on (a box may expose several cards in sysfs)
        if (hipDeviceGetPCIBusId(bus, sizeof bus, 0) != hipSuccess) bus[0] = 0;
        for (char *c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c += 32;
        snprintf(cmd, sizeof cmd, "ls /sys/bus/pci/devices/%s/hwmon/hwmon*/power1_input 2>/dev/null | head -1", bus);
        FILE *f = popen(cmd, "r"); if (f) { if (fgets(pw, sizeof pw, f)) pw[strcspn(pw, "\n")] = 0; pclose(f); }
        fprintf(stderr, "hwmon: %s (device %s)\n", pw, bus);
        if (pw[0]) { snprintf(fq, sizeof fq, "%s", pw); char *q = strstr(fq, "power1_input"); if (q) strcpy(q, "freq1_input"); }
        th = std::thread([this] { while (!stop) { const double w = rd(pw), z = rd(fq); sw = sw + w / 1e6; sf = sf + z / 1e6; n = n + 1; std::this_thread::sleep_for(std::chrono::milliseconds(10)); } });
    }
    void reset() { sw = 0; sf = 0; n = 0; }
    void read(double &w, double &mhz) { const int k = n; w = k ? sw / k : 0; mhz = k ? sf / k : 0; }
    ~HwmonSampler() { stop = true; th.join(); }
};
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int F, int SG>
__global__ void __launch_bounds__(64) chain_kernel(fe29_t *io, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    fe29_t x = io[t];
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        if (SG) { const fe29_t x2 = fe29_sqr_sg<F>(x); x = fe29_mul_sg<F>(x2, x); }
        else { const fe29_t x2 = fe29_sqr_lz<F>(x); x = fe29_mul_lz<F>(x2, x); }
    }
    io[t] = x;
#endif
}

// The 3-lane round on registers (sponge.cuh poseidon_permute_tri), for the power question of profiles/r05_clock_power.md: MODE 0 = as shipped (27 ds_bpermute per round for the three
// x^7, round constants loaded from a 55-entry table in global memory), MODE 1 = no cross-lane moves (every term uses the lane's own x^7: wrong values, same arithmetic), MODE 2 = cross-lane
// moves but the round constant from a register, MODE 3 = neither, MODE 4 = 18 moves (the lane's own x^7 stays in place).  The difference in power / clock / rate between the modes is what the moves and the loads cost.
template <int F, int MODE>
__global__ void __launch_bounds__(64) round_kernel(fe29_t *io, const fe29_t *__restrict__ rc, int nperm) {
#if defined(__HIP_DEVICE_COMPILE__)
    const size_t t_ = (size_t)blockIdx.x * 64 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, g = lane == 63u ? 20u : lane / 3u, base = 3u * g, e = lane == 63u ? 0u : lane - base;
    fe29_t x = io[t_];
    const fe29_t m0 = rc[165 + e * 3], m1 = rc[166 + e * 3], m2 = rc[167 + e * 3], c0 = rc[e];
#pragma unroll 1
    for (int it = 0; it < nperm; ++it) {
#pragma unroll 1
        for (int r = 0; r < 55; ++r) {
            const fe29_t x2 = fe29_sqr_sg<F>(x);
            const fe29_t x4 = fe29_sqr_sg<F>(x2);
            const fe29_t t = fe29_mul_sg<F>(fe29_mul_sg<F>(x4, x2), x);
            fe29_t t0 = t, t1 = t, t2 = t;
            if (MODE == 0 || MODE == 2) {
#pragma unroll
                for (int i = 0; i < L29; ++i) { t0.v[i] = (uint32_t)__shfl((int)t.v[i], (int)base, 64); t1.v[i] = (uint32_t)__shfl((int)t.v[i], (int)base + 1, 64); t2.v[i] = (uint32_t)__shfl((int)t.v[i], (int)base + 2, 64); }
            }
            if (MODE == 4) {                                     // the own term stays in place: 18 moves (what ships since this probe)
                const int en = e == 2u ? 0 : (int)e + 1, ep = e == 0u ? 2 : (int)e - 1;
#pragma unroll
                for (int i = 0; i < L29; ++i) { t1.v[i] = (uint32_t)__shfl((int)t.v[i], (int)base + en, 64); t2.v[i] = (uint32_t)__shfl((int)t.v[i], (int)base + ep, 64); }
            }
            x = fe29_dot3rc_sg<F>(m0, t0, m1, t1, m2, t2, (MODE == 0 || MODE == 1 || MODE == 4) ? rc[r * 3 + e] : c0);
        }
    }
    io[t_] = x;
#endif
}

int main(int argc, char **argv) {
    const bool values = argc > 1 && !strcmp(argv[1], "--values");
    if (values) {
        const int N = 256;
        std::vector<fe29_t> in(N), a(N), b(N);
        srand(7);
        for (int i = 0; i < N; ++i) for (int k = 0; k < 9; ++k) in[i].v[k] = k < 8 ? (((uint32_t)rand() << 15) ^ rand()) & M29 : (uint32_t)rand() & 0x3fffffu;    // < 2^254
        for (int k = 0; k < 9; ++k) { in[0].v[k] = 0; in[1].v[k] = k == 0; in[2].v[k] = k < 8 ? M29 : 0x3fffffu; in[3].v[k] = k == 8 ? 0x400000u : (k == 0 ? 0u : 0u); }   // 0, 1, all ones, p - 1 (top limb) ...
        fe29_t *d;
        CHECK(hipMalloc(&d, N * sizeof(fe29_t)));
        for (int f = 0; f < 2; ++f) for (int n : {1, 2, 7}) {
            CHECK(hipMemcpy(d, in.data(), N * sizeof(fe29_t), hipMemcpyHostToDevice));
            if (f == 0) chain_kernel<0, 1><<<N / 64, 64>>>(d, n); else chain_kernel<1, 1><<<N / 64, 64>>>(d, n);
            CHECK(hipMemcpy(a.data(), d, N * sizeof(fe29_t), hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(d, in.data(), N * sizeof(fe29_t), hipMemcpyHostToDevice));
            if (f == 0) chain_kernel<0, 0><<<N / 64, 64>>>(d, n); else chain_kernel<1, 0><<<N / 64, 64>>>(d, n);
            CHECK(hipMemcpy(b.data(), d, N * sizeof(fe29_t), hipMemcpyDeviceToHost));
            for (int i = 0; i < N; ++i) {
                printf("%d %d", f, n);
                for (int k = 0; k < 9; ++k) printf(" %u", in[i].v[k]);
                for (int k = 0; k < 9; ++k) printf(" %u", a[i].v[k]);
                for (int k = 0; k < 9; ++k) printf(" %u", b[i].v[k]);
                printf("\n");
            }
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "--rounds")) {      // round_kernel modes 0..3 held for SECONDS each at 5 waves per SIMD, power and clock per window from hwmon
        const double secs = argc > 2 ? atof(argv[2]) : 5.0;
        const int wps = argc > 3 ? atoi(argv[3]) : 5, nperm = 26 * 8;
        const size_t waves = (size_t)1024 * wps, lanes = waves * 64;
        std::vector<fe29_t> in(lanes), tab(176); srand(13);
        auto rnd = [](fe29_t &v) { for (int k = 0; k < 9; ++k) v.v[k] = k < 8 ? (((uint32_t)rand() << 15) ^ rand()) & M29 : (uint32_t)rand() & 0x3fffffu; };
        for (auto &v : in) rnd(v);
        for (auto &v : tab) rnd(v);
        fe29_t *d, *dt; CHECK(hipMalloc(&d, lanes * sizeof(fe29_t))); CHECK(hipMemcpy(d, in.data(), lanes * sizeof(fe29_t), hipMemcpyHostToDevice));
        CHECK(hipMalloc(&dt, tab.size() * sizeof(fe29_t))); CHECK(hipMemcpy(dt, tab.data(), tab.size() * sizeof(fe29_t), hipMemcpyHostToDevice));
        HwmonSampler hw;
        hipEvent_t s0, s1; CHECK(hipEventCreate(&s0)); CHECK(hipEventCreate(&s1));
        const char *names[5] = {"27 ds_bpermute + round constant from memory per round (shipped until this probe)", "no cross-lane moves", "27 cross-lane moves, round constant in a register", "neither", "18 ds_bpermute (own term in place) + round constant from memory: as shipped"};
        for (int rep = 0; rep < 2; ++rep) for (int mode = 0; mode < 5; ++mode) {
            for (double done = 0; done < secs;) {
                hw.reset();
                CHECK(hipEventRecord(s0));
                for (int r = 0; r < 2; ++r) {
                    if (mode == 0) round_kernel<0, 0><<<waves, 64>>>(d, dt, nperm); else if (mode == 1) round_kernel<0, 1><<<waves, 64>>>(d, dt, nperm);
                    else if (mode == 2) round_kernel<0, 2><<<waves, 64>>>(d, dt, nperm); else if (mode == 3) round_kernel<0, 3><<<waves, 64>>>(d, dt, nperm); else round_kernel<0, 4><<<waves, 64>>>(d, dt, nperm);
                }
                CHECK(hipEventRecord(s1));
                CHECK(hipEventSynchronize(s1));
                double w, mhz; hw.read(w, mhz);
                float ms; CHECK(hipEventElapsedTime(&ms, s0, s1)); done += ms * 1e-3;
                printf("{\"probe\": \"3-lane round on registers, %s\", \"mode\": %d, \"rep\": %d, \"waves_per_simd\": %d, \"t_s\": %.2f, \"G_lane_rounds_per_s\": %.3f, \"socket_power_w\": %.0f, \"sclk_mhz\": %.0f}\n",
                       names[mode], mode, rep, wps, done, 2.0 * nperm * 55 * lanes / ms / 1e6, w, mhz);
                fflush(stdout);
            }
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "--sustain")) {     // the signed-digit chain on RANDOM field elements held for SECONDS at 5 waves per SIMD, with the socket power and the
        // shader clock read from amdgpu's hwmon files per window: does real field arithmetic alone (registers only: no LDS, no cross-lane moves, no memory in the loop) pull the clock down
        // the way the verifier's step does (profiles/r05_clock_power.md)?
        const double secs = argc > 2 ? atof(argv[2]) : 6.0;
        const int wps = argc > 3 ? atoi(argv[3]) : 5, nn = 20000;
        const size_t waves = (size_t)1024 * wps, lanes = waves * 64;
        std::vector<fe29_t> in(lanes); srand(11);
        for (size_t i = 0; i < lanes; ++i) for (int k = 0; k < 9; ++k) in[i].v[k] = k < 8 ? (((uint32_t)rand() << 15) ^ rand()) & M29 : (uint32_t)rand() & 0x3fffffu;
        fe29_t *d; CHECK(hipMalloc(&d, lanes * sizeof(fe29_t))); CHECK(hipMemcpy(d, in.data(), lanes * sizeof(fe29_t), hipMemcpyHostToDevice));
        HwmonSampler hw;
        hipEvent_t s0, s1; CHECK(hipEventCreate(&s0)); CHECK(hipEventCreate(&s1));
        for (int sg = 1; sg >= 0; --sg) {
            for (double done = 0; done < secs;) {
                hw.reset();
                CHECK(hipEventRecord(s0));
                for (int r = 0; r < 4; ++r) { if (sg) chain_kernel<0, 1><<<waves, 64>>>(d, nn); else chain_kernel<0, 0><<<waves, 64>>>(d, nn); }
                CHECK(hipEventRecord(s1));
                CHECK(hipEventSynchronize(s1));
                double w, mhz; hw.read(w, mhz);
                float ms; CHECK(hipEventElapsedTime(&ms, s0, s1)); done += ms * 1e-3;
                printf("{\"probe\": \"sustained x <- x^3 chain, %s, random field elements\", \"waves_per_simd\": %d, \"t_s\": %.2f, \"G_products_per_s\": %.2f, \"socket_power_w\": %.0f, \"sclk_mhz\": %.0f}\n",
                       sg ? "signed digits" : "lazy (v_sub per digit)", wps, done, 4 * 2.0 * nn * lanes / ms / 1e6, w, mhz);
                fflush(stdout);
            }
        }
        return 0;
    }
    const int n = 4000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {2, 5, 8}) {
        const size_t waves = (size_t)1024 * wps, lanes = waves * 64;
        fe29_t *d; CHECK(hipMalloc(&d, lanes * sizeof(fe29_t))); CHECK(hipMemset(d, 1, lanes * sizeof(fe29_t)));
        for (int sg = 0; sg < 2; ++sg) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                if (sg) chain_kernel<0, 1><<<waves, 64>>>(d, n); else chain_kernel<0, 0><<<waves, 64>>>(d, n);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
            }
            printf("{\"form\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"G_products_per_s\": %.2f}\n", sg ? "signed digits" : "lazy (v_sub per digit)", wps, best, 2.0 * n * lanes / best / 1e6);
        }
        CHECK(hipFree(d));
    }
    return 0;
}
