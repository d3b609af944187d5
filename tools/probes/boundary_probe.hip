// dev probe: do the kernel boundaries of OTHER streams slow a running kernel down?  A "victim" kernel (one wave per SIMD-ish: 1024 blocks of
// 64 threads) runs a fixed amount of work -- either pure ALU (dependent 64-bit multiply-adds) or L2-resident memory reads (a 1 MiB
// table walked with dependent loads) -- alone, and while S other streams issue back-to-back tiny kernels (one wave, ~2 us each).
// usage: boundary_probe        prints one JSON line per case
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void alu_kernel(unsigned iters, unsigned long long *out) {
    unsigned long long a = threadIdx.x + 1, b = blockIdx.x + 3;
    for (unsigned i = 0; i < iters; ++i) a = a * b + (a >> 7);
    if (a == 42) out[0] = a;
}
__global__ void mem_kernel(unsigned iters, const unsigned *table, unsigned mask, unsigned long long *out) {
    unsigned idx = (blockIdx.x * 64 + threadIdx.x) & mask;
    unsigned long long acc = 0;
    for (unsigned i = 0; i < iters; ++i) { idx = table[idx] & mask; acc += idx; }
    if (acc == 42) out[0] = acc;
}
__global__ void tiny_kernel(unsigned long long *out) { if (threadIdx.x == 999) out[1] = 1; }
__global__ void tiny_store_kernel(unsigned *buf) { buf[blockIdx.x * 64 + threadIdx.x] += 1; }
static double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
int main() {
    unsigned long long *out; hipMalloc(&out, 64);
    const unsigned n = 1u << 18; std::vector<unsigned> h(n); for (unsigned i = 0; i < n; ++i) h[i] = (i * 2654435761u + 12345u);
    unsigned *table; hipMalloc(&table, n * 4); hipMemcpy(table, h.data(), n * 4, hipMemcpyHostToDevice);
    unsigned *sbuf; hipMalloc(&sbuf, 64 * 4 * 64); hipMemset(sbuf, 0, 64 * 4 * 64);
    hipStream_t victim; hipStreamCreateWithFlags(&victim, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 2; ++kind)
        for (int noise_kind = 0; noise_kind < 2; ++noise_kind)
            for (int S : {0, 1, 4, 15}) {
                if (S == 0 && noise_kind == 1) continue;
                std::vector<hipStream_t> st(S);
                for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                auto launch_victim = [&] { if (kind == 0) alu_kernel<<<1024, 64, 0, victim>>>(200000, out); else mem_kernel<<<1024, 64, 0, victim>>>(20000, table, n - 1, out); };
                launch_victim(); hipDeviceSynchronize();
                hipEventRecord(e0, victim); launch_victim(); hipEventRecord(e1, victim);
                const auto t0 = std::chrono::steady_clock::now();
                long tiny = 0;
                while (hipEventQuery(e1) == hipErrorNotReady && ms_since(t0) < 2000.0 && S)
                    for (int i = 0; i < S; ++i) { if (noise_kind == 0) tiny_kernel<<<1, 64, 0, st[i]>>>(out); else tiny_store_kernel<<<4, 64, 0, st[i]>>>(sbuf + i * 256); ++tiny; }
                hipDeviceSynchronize();
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                printf("{\"victim\": \"%s\", \"noise\": \"%s\", \"noise_streams\": %d, \"victim_ms\": %.3f, \"tiny_kernels_issued\": %ld}\n", kind == 0 ? "alu" : "l2_reads",
                       noise_kind == 0 ? "empty kernels" : "kernels that store", S, ms, tiny);
                for (auto &s : st) hipStreamDestroy(s);
            }
    return 0;
}
