// How long does the dispatcher take to get every workgroup of a grid started? Per block: wall clock at its first
// instruction; printed: spread (max - min) by block size, static LDS size and register footprint.
// build+run: hipcc --offload-arch=gfx950 -O3 -o /tmp/wgd tools/ubench/wg_dispatch.hip && /tmp/wgd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
template <int LDS_KB, int REGS>
__global__ void k(unsigned long long* t, float* sink) {
    __shared__ char lds[LDS_KB * 1024 > 0 ? LDS_KB * 1024 : 4];
    if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64();
    float r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = threadIdx.x * 0.5f + i;
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 1000) {   // 10 us of residency: every block of the grid is resident at once
#pragma unroll
        for (int i = 0; i < REGS; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += r[i];
    if (s == 1234.5f) { sink[0] = s; lds[threadIdx.x] = 1; sink[1] = lds[(threadIdx.x + 1) & 1023]; }
}
template <int LDS_KB, int REGS>
static void run(const char* name, int blocks, int threads, unsigned long long* d, float* sink) {
    std::vector<unsigned long long> h(blocks);
    double best = 1e9, sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL((k<LDS_KB, REGS>), dim3(blocks), dim3(threads), 0, 0, d, sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        auto mm = std::minmax_element(h.begin(), h.end());
        const double us = (*mm.second - *mm.first) / 100.0;
        if (rep > 0) { best = std::min(best, us); sum += us; }
    }
    printf("%-34s blocks %4d x %4d threads: start spread %.2f us (best %.2f)\n", name, blocks, threads, sum / 4, best);
}
int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 4096 * 8); hipMalloc(&sink, 64);
    run<0, 8>("no LDS, few regs", 256, 512, d, sink);
    run<0, 8>("no LDS, few regs", 128, 1024, d, sink);
    run<0, 8>("no LDS, few regs", 256, 1024, d, sink);
    run<60, 8>("60 KB LDS, few regs", 128, 1024, d, sink);
    run<60, 8>("60 KB LDS, few regs", 256, 512, d, sink);
    run<0, 96>("no LDS, ~100 regs", 128, 1024, d, sink);
    run<0, 96>("no LDS, ~100 regs", 256, 512, d, sink);
    run<60, 96>("60 KB LDS, ~100 regs", 128, 1024, d, sink);
    run<60, 96>("60 KB LDS, ~100 regs", 256, 512, d, sink);
    run<60, 96>("60 KB LDS, ~100 regs", 512, 256, d, sink);
    return 0;
}
