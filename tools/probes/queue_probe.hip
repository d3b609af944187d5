// dev probe: how many kernels of DIFFERENT streams does the GPU really run side by side?  Each kernel is one wave spinning for `us`
// microseconds (wall clock); S streams get `per` such kernels each, back to back.  If everything overlapped the wall time would be
// per * us whatever S; the ratio tells the effective concurrency.      usage: queue_probe [us] [per]     (set GPU_MAX_HW_QUEUES first)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin_kernel(unsigned long long ticks, unsigned *sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned x = 0;
    while (wall_clock64() - t0 < ticks) x += 1;
    if (x == 0xffffffffu) *sink = x;
}
int main(int argc, char **argv) {
    const double us = argc > 1 ? atof(argv[1]) : 200.0;
    const int per = argc > 2 ? atoi(argv[2]) : 20;
    int rate_khz = 0; hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const unsigned long long ticks = (unsigned long long)(us * 1e-6 * rate_khz * 1e3);
    unsigned *sink; hipMalloc(&sink, 4);
    for (int S : {1, 2, 4, 8, 16, 24, 32}) {
        std::vector<hipStream_t> st(S);
        for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto &s : st) spin_kernel<<<1, 64, 0, s>>>(ticks, sink);
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < per; ++i) for (auto &s : st) spin_kernel<<<1, 64, 0, s>>>(ticks, sink);
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("{\"streams\": %d, \"kernels_per_stream\": %d, \"kernel_us\": %.0f, \"wall_ms\": %.2f, \"effective_concurrency\": %.2f}\n", S, per, us, ms, S * per * us * 1e-3 / ms);
        for (auto &s : st) hipStreamDestroy(s);
    }
    return 0;
}
