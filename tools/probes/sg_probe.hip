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
#include "fp.cuh"
#include "fp29.cuh"
using namespace mb;
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
