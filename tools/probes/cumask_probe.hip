// dev probe: what does a CU mask of hipExtStreamCreateWithCUMask select on a multi-XCD device?  For a few masks, launch 4096 one-wave blocks on
// the masked stream and count the blocks per XCC and the distinct CUs per XCC they ran on (HW_ID / XCC_ID registers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void where_kernel(unsigned iters, unsigned *where) {
    unsigned long long a = threadIdx.x + 1, b = blockIdx.x + 3;
    for (unsigned i = 0; i < iters; ++i) a = a * b + (a >> 7);
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[blockIdx.x * 2] = hw; where[blockIdx.x * 2 + 1] = (xcc & 0xf) | (a == 42 ? 16 : 0);
    }
}
static void run(const char *name, const unsigned *mask) {
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("%s: create failed\n", name); return; }
    const int W = 4096; unsigned *where; hipMalloc(&where, W * 8);
    where_kernel<<<W, 64, 0, st>>>(20000, where);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(W * 2); hipMemcpy(h.data(), where, W * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_xcc; std::map<unsigned, std::set<unsigned>> cus;
    for (int w = 0; w < W; ++w) { const unsigned hw = h[w * 2], xcc = h[w * 2 + 1] & 0xf; per_xcc[xcc]++; cus[xcc].insert((hw >> 8) & 0xff); }
    printf("%s:", name);
    for (auto &kv : per_xcc) printf("  xcc%u %d blocks on %zu CUs", kv.first, kv.second, cus[kv.first].size());
    printf("\n");
    hipFree(where); hipStreamDestroy(st);
}
int main() {
    unsigned all[8]; for (auto &x : all) x = 0xffffffffu;
    run("all 256 bits", all);
    unsigned lo96[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0}; run("bits 0..95", lo96);
    unsigned lo128[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}; run("bits 0..127", lo128);
    unsigned hi128[8] = {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}; run("bits 128..255", hi128);
    unsigned p8[8]; for (auto &x : p8) x = 0x07070707u; run("b % 8 < 3", p8);
    unsigned ev[8]; for (auto &x : ev) x = 0x55555555u; run("even bits", ev);
    unsigned w0[8] = {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}; run("bits 0..31", w0);
    unsigned lo12[8]; for (auto &x : lo12) x = 0x00000fffu; run("low 12 of every 32", lo12);
    return 0;
}
