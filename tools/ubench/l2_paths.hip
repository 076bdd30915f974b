// Micro-benchmark: aggregate L2 -> CU rate with EVERY CU pulling an L2-resident working set, by path:
//   dma   NL loader waves per CU issue global_load_lds_dwordx4 (1 KiB per wave-instruction) into an LDS ring
//   vgpr  NV waves per CU issue global_load_dwordx4 into registers (8 in flight per wave)
//   both  the two at once (the question: are they one ceiling or two?)
// Working set per XCD: a 2 MiB region read by all of the XCD's CUs (L2 hits after the first touch), as the activation panel of a
// prefill GEMM is. Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/l2p tools/ubench/l2_paths.hip && /tmp/l2p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void glds16(const void* gptr, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_u32(const void* p) { return (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char*)p); }

// waves [0, NL): DMA loaders; waves [NL, NL + NV): register loaders. Region of this block's XCD: src + (blockIdx.x & 7) * 2 MiB.
template <int NL, int NV, int DEPTH>
__global__ __launch_bounds__(64 * (NL + NV)) void k_paths(const char* __restrict__ src, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)(blockIdx.x & 7) * (2u << 20);
    // every wave walks the 2 MiB region in 1 KiB pieces with its own stride pattern (different waves / CUs touch different lines at a time)
    const unsigned start = ((blockIdx.x >> 3) * 37u + w * 131u) & 2047u;
    if (w < NL) {
        const unsigned lds = __builtin_amdgcn_readfirstlane(lds_addr_u32(smem) + w * DEPTH * 1024);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) glds16(base + (size_t)((start + it * DEPTH + d) & 2047u) * 1024 + lane * 16, lds + d * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        u32x4 acc = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = *(const u32x4*)(base + (size_t)((start + it * DEPTH + d) & 2047u) * 1024 + lane * 16);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
        if (acc[0] == 0x12345678u && iters < 0) sink[0] = acc[1];
    }
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
    const int iters = 3000;
    printf("%-22s %10s %12s %12s\n", "config", "B/clk/CU@2.1", "TB/s chip", "(dma / vgpr)");
#define RUN(NL, NV, DEPTH)                                                                                                   \
    {                                                                                                                        \
        auto kfn = k_paths<NL, NV, DEPTH>;                                                                                   \
        hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                       \
        const size_t lds = (size_t)(NL > 0 ? NL : 1) * DEPTH * 1024;                                                         \
        float ms = run([&] { hipLaunchKernelGGL(kfn, dim3(256), dim3(64 * (NL + NV)), lds, 0, src, iters, sink); });         \
        double bytes = 256.0 * (NL + NV) * DEPTH * 1024.0 * iters;                                                           \
        printf("dma %2d vgpr %2d depth %2d  %10.1f %12.2f   (%5.2f / %5.2f)\n", NL, NV, DEPTH, bytes / 256 / (ms * 1e-3) / 2.1e9, bytes / ms / 1e9, \
               bytes / ms / 1e9 * NL / (NL + NV), bytes / ms / 1e9 * NV / (NL + NV));                                        \
    }
    RUN(4, 0, 8) RUN(4, 0, 12) RUN(8, 0, 8) RUN(12, 0, 8) RUN(16, 0, 6)
    RUN(0, 4, 8) RUN(0, 8, 8) RUN(0, 12, 8) RUN(0, 16, 8)
    RUN(4, 8, 8) RUN(2, 8, 8) RUN(4, 4, 8) RUN(8, 8, 8)
    return 0;
}
