// Micro-benchmark: how fast can one CU pull an L2-resident buffer in, through (a) LDS-DMA (global_load_lds_dwordx4) and
// (b) plain global_load_dwordx4 into registers (+ ds_write_b128), as a function of waves per CU and loads in flight per wave.
// One block per CU (grid = 256), every wave streams its own 64 KiB window of a 16 MiB buffer (L2/MALL resident) repeatedly.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/l2cu tools/ubench/l2_to_cu.hip && /tmp/l2cu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void glds16(const void* gptr, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_u32(const void* p) {
    return (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char*)p);
}

template <int DEPTH>
__global__ __launch_bounds__(1024) void k_ldsdma(const char* __restrict__ src, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* base = src + ((size_t)(blockIdx.x * nw + w) % 256) * 65536;   // 256 windows of 64 KiB
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_addr_u32(smem) + w * DEPTH * 1024);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) glds16(base + ((it * DEPTH + d) & 63) * 1024 + lane * 16, lds + d * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (smem[threadIdx.x * 16] == 123 && iters < 0) sink[0] = 1;
}

template <int DEPTH>
__global__ __launch_bounds__(1024) void k_regs(const char* __restrict__ src, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* base = src + ((size_t)(blockIdx.x * nw + w) % 256) * 65536;
    char* lds = smem + w * DEPTH * 1024 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = *(const u32x4*)(base + ((it * DEPTH + d) & 63) * 1024 + lane * 16);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) *(u32x4*)(lds + d * 1024) = v[d];
    }
    if (smem[threadIdx.x * 16] == 123 && iters < 0) sink[0] = 1;
}

template <typename F>
static float run(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    char* src; unsigned* sink;
    hipMalloc(&src, 16 << 20); hipMemset(src, 1, 16 << 20); hipMalloc(&sink, 4);
    const int iters = 2000;
    printf("%-8s %6s %6s %10s %12s\n", "path", "waves", "depth", "GB/s/CU", "TB/s chip");
#define RUN(KERN, NAME, DEPTH, WAVES)                                                                            \
    {                                                                                                            \
        const size_t lds = (size_t)WAVES * DEPTH * 1024;                                                         \
        hipFuncSetAttribute((const void*)KERN<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
        float ms = run([&] { hipLaunchKernelGGL(KERN<DEPTH>, dim3(256), dim3(64 * WAVES), lds, 0, src, iters, sink); }); \
        double bytes = 256.0 * WAVES * DEPTH * 1024.0 * iters;                                                   \
        printf("%-8s %6d %6d %10.1f %12.2f\n", NAME, WAVES, DEPTH, bytes / 256 / ms / 1e6, bytes / ms / 1e9);    \
    }
    RUN(k_ldsdma, "ldsdma", 2, 4) RUN(k_ldsdma, "ldsdma", 6, 4) RUN(k_ldsdma, "ldsdma", 6, 8) RUN(k_ldsdma, "ldsdma", 6, 12)
    RUN(k_ldsdma, "ldsdma", 8, 16) RUN(k_ldsdma, "ldsdma", 4, 16)
    RUN(k_regs, "regs", 2, 4) RUN(k_regs, "regs", 6, 4) RUN(k_regs, "regs", 6, 8) RUN(k_regs, "regs", 6, 12)
    RUN(k_regs, "regs", 8, 16) RUN(k_regs, "regs", 4, 16)
    return 0;
}
