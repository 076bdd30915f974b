// What the matrix pipes deliver with nothing else in the way: back-to-back v_mfma_f32_16x16x32_bf16 (and 32x32x16) on
// independent accumulators, W waves per SIMD, every CU busy. build+run: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma tools/ubench/mfma_peak.hip && /tmp/mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float s = 0.f;
    if (KIND == 0) {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    } else {
        f32x16 c[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][15];
    }
    if (s == 12345.f) out[0] = s;
}
int main() {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind)
        for (int waves : {4, 8, 16}) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, d, iters);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flop = (kind == 0 ? 8.0 * 16 * 16 * 32 * 2 : 4.0 * 32 * 32 * 16 * 2) * iters * waves * 256;
                if (rep == 2) printf("%s waves/CU=%2d: %.2f ms  %.0f TFLOP/s\n", kind == 0 ? "16x16x32" : "32x32x16", waves, ms, flop / ms / 1e9);
            }
        }
    return 0;
}
