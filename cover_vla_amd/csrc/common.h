// Shared device helpers for the gfx950 (CDNA4) kernels of libcover_hip.
// Wave = 64 lanes everywhere in this library; nothing here is portable to other targets by design.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits; all bf16 tensors in the C ABI are uint16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define WAVE 64

// hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KiB) once per (kernel, DEVICE): the attribute belongs to the device's copy of the function, so a
// process-wide `static` result would leave the second device of a process without it. One cache object per expansion site = per kernel instantiation.
#include <atomic>
struct LdsAttrCache { std::atomic<unsigned> ok{0}; };   // bit d: device d has the attribute
static inline hipError_t lds_attr_160k_cached(const void* fn, LdsAttrCache& c) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned bit = 1u << (dev & 31);
    if (c.ok.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) c.ok.fetch_or(bit, std::memory_order_release);
    return e;
}
#define LDS_ATTR_160K(kfn) ([&]() -> hipError_t { static LdsAttrCache c_; return lds_attr_160k_cached((const void*)(kfn), c_); }())

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN kept quiet: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32), one instruction per
// PAIR instead of ~6 integer ops per element
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// activation ids shared by bf16 and f32 GEMM epilogues (mirrored in include/cover_hip.h)
enum { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2, ACT_SILU = 3, ACT_RELU = 4 };

__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        case ACT_GELU_TANH: {
            const float k0 = 0.7978845608028654f, k1 = 0.044715f;
            float u = k0 * (x + k1 * x * x * x);
            return 0.5f * x * (1.0f + tanhf(u));
        }
        case ACT_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_RELU: return x > 0.f ? x : 0.f;
        default: return x;
    }
}

// The same activations for values that are ROUNDED TO bf16 right behind them (the bf16 GEMM epilogues): SiLU and tanh-GELU through the hardware
// exp2 and reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp of fp32, far below half a bf16 ulp) instead of an IEEE division / tanhf -- the GLU store
// loop of the 224 x 192 prefill tile spent 6.9 of its 8.5 us epilogue in this arithmetic (cover_gemm_probe). tanh(u) = 1 - 2 / (1 + e^{2u})
// saturates correctly at both ends (e^{2u} -> inf gives 1, -> 0 gives -1); its absolute error near 0 is what 0.5 x (1 + tanh) needs.
// The fp32 kernels (verifier heads, pi0 projections: pinned at 1e-5) keep act_apply.
__device__ __forceinline__ float act_apply_bf16(float x, int act) {
    switch (act) {
        case ACT_GELU_TANH: {
            const float k0 = 0.7978845608028654f, k1 = 0.044715f;
            const float u = k0 * (x + k1 * x * x * x);
            const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));
            return 0.5f * x * (1.0f + t);
        }
        case ACT_SILU: return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
        default: return act_apply(x, act);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (<= 16 waves); `red` is >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16 (hardware adds the lane part)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Same copy issued from inline asm: hipcc does not see a VMEM operation, so it neither counts it nor drains it with an
// automatic `s_waitcnt vmcnt(0)` in front of the next ds_read (it cannot prove the LDS-DMA destination does not alias
// the read). The caller owns the waits: counted `s_waitcnt vmcnt(N)` + s_barrier before the data is read.
// lds_wave_base_u32 = wave-uniform LDS byte address (readfirstlane'd). M0 is saved/restored inside the statement.
__device__ __forceinline__ void glds16_asm(const void* gsrc, uint32_t lds_wave_base_u32) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_wave_base_u32)
        : "memory");
}
// the 4-byte form: LDS destination = M0 + lane * 4
__device__ __forceinline__ void glds4_asm(const void* gsrc, uint32_t lds_wave_base_u32) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_wave_base_u32)
        : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_u32(const void* p) {
    return (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)p);
}
