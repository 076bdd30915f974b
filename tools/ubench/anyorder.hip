// Does hipExtAnyOrderLaunch drop the AQL barrier bit on gfx950? Kernel A spins for 100 us; kernel B is launched behind it
// on the same stream, once normally and once with the flag; both record s_memrealtime at start and end.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/anyorder.hip -o gpurun_out/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_readcyclecounter() * 0 + wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipStream_t st; hipStreamCreate(&st);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);   // kHz
    const unsigned long long t100 = (unsigned long long)rate / 10;                     // 100 us
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, t100, d);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, t100 / 10, d + 2);
            else hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, nullptr, nullptr, mode == 1 ? 0 : hipExtAnyOrderLaunch, t100 / 10, d + 2);
            hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, t100 / 10, d + 4);
            hipStreamSynchronize(st);
            unsigned long long h[6]; hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
            const double us = 1e3 / rate;
            printf("mode %d (%s): A [0, %.1f] us  B [%.1f, %.1f]  C [%.1f, %.1f]\n", mode, mode == 0 ? "plain" : mode == 1 ? "ext flags=0" : "ext AnyOrder",
                   (h[1] - h[0]) * us, ((double)h[2] - (double)h[0]) * us, ((double)h[3] - (double)h[0]) * us, ((double)h[4] - (double)h[0]) * us, ((double)h[5] - (double)h[0]) * us);
        }
    }
    return 0;
}
