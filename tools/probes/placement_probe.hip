// dev probe: where do the waves of small concurrent kernels land?  Every stream runs the same latency-bound kernel (W one-wave blocks of
// dependent 64-bit multiply-adds, like the 8-lane sponge kernels of a 256-proof call: 32 waves).  S streams at once hold S*W waves -- far
// fewer than the 1024 SIMDs -- so if the dispatcher spread them the kernel time would not depend on S.  Each wave records the
// (XCC, SE, CU, SIMD) it ran on (HW_ID / XCC_ID registers) so that the sharing can be seen directly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void chain_kernel(unsigned iters, unsigned *where, unsigned long long *out) {
    unsigned long long a = threadIdx.x + 1, b = blockIdx.x + 3;
    for (unsigned i = 0; i < iters; ++i) a = a * b + (a >> 7);
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[blockIdx.x * 2] = hw; where[blockIdx.x * 2 + 1] = xcc;
    }
    if (a == 42) out[0] = a;
}
int main(int argc, char **argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 32;
    unsigned long long *out; hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int S : {1, 2, 4, 8, 16}) {
        std::vector<hipStream_t> st(S); std::vector<unsigned *> where(S);
        for (int i = 0; i < S; ++i) { hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); hipMalloc(&where[i], W * 8); }
        for (int i = 0; i < S; ++i) chain_kernel<<<W, 64, 0, st[i]>>>(1000, where[i], out);
        hipDeviceSynchronize();
        hipEventRecord(e0, st[0]);
        for (int rep = 0; rep < 4; ++rep) for (int i = 0; i < S; ++i) chain_kernel<<<W, 64, 0, st[i]>>>(100000, where[i], out);
        hipEventRecord(e1, st[0]);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::map<unsigned long long, int> per_simd, per_cu;
        for (int i = 0; i < S; ++i) {
            std::vector<unsigned> h(W * 2); hipMemcpy(h.data(), where[i], W * 8, hipMemcpyDeviceToHost);
            for (int w = 0; w < W; ++w) {
                const unsigned hw = h[w * 2], xcc = h[w * 2 + 1] & 0xf;
                const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                const unsigned long long cuid = ((unsigned long long)xcc << 16) | (se << 8) | (sh << 6) | cu;
                per_cu[cuid]++; per_simd[(cuid << 2) | simd]++;
            }
        }
        int max_simd = 0, max_cu = 0; for (auto &kv : per_simd) if (kv.second > max_simd) max_simd = kv.second; for (auto &kv : per_cu) if (kv.second > max_cu) max_cu = kv.second;
        printf("{\"streams\": %d, \"waves_per_kernel\": %d, \"ms_per_kernel_on_stream0\": %.3f, \"waves\": %d, \"distinct_simds\": %zu, \"distinct_cus\": %zu, \"max_waves_on_one_simd\": %d, \"max_waves_on_one_cu\": %d}\n",
               S, W, ms / 4, S * W, per_simd.size(), per_cu.size(), max_simd, max_cu);
        for (int i = 0; i < S; ++i) { hipStreamDestroy(st[i]); hipFree(where[i]); }
    }
    return 0;
}
