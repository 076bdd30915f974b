// What v_permlane16_swap / v_permlane32_swap return on gfx950 (the builtins' two results for x = lane id), and the four-row reductions built on them against
// the __shfl_xor forms:  hipcc --offload-arch=gfx950 -O3 -o /tmp/pls tools/ubench/permlane_swap.hip && /tmp/pls
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
    const unsigned l = threadIdx.x;
    const u2 a = __builtin_amdgcn_permlane16_swap(l, l, false, false);
    const u2 b = __builtin_amdgcn_permlane32_swap(l, l, false, false);
    o[l] = a[0]; o[64 + l] = a[1]; o[128 + l] = b[0]; o[192 + l] = b[1];
    // distinct operands: vdst = lane, vsrc = 100 + lane
    const u2 c = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
    o[256 + l] = c[0]; o[320 + l] = c[1];
    float x = (float)((l * 37u) % 61u);
    float m1 = fmaxf(x, __shfl_xor(x, 16)); m1 = fmaxf(m1, __shfl_xor(m1, 32));
    o[384 + l] = (unsigned)m1;
    // both results of the builtin feeding ONE instruction (the sum over l and l ^ 16): ROCm 7.0's clang emits v_add_f32 v, v, v here (= 2 x[l]) ...
    const unsigned xi = __builtin_bit_cast(unsigned, x);
    const u2 e = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
    o[448 + l] = (unsigned)(__builtin_bit_cast(float, e[0]) + __builtin_bit_cast(float, e[1]));
    // ... the inline-asm form does what the instruction does
    float fa = x, fb = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(fa), "+v"(fb));
    o[512 + l] = (unsigned)(fa + fb);
    o[576 + l] = (unsigned)(x + __shfl_xor(x, 16));
}
int main() {
    unsigned* d; hipMalloc(&d, 640 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[640]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"permlane16_swap(l,l)[0]", "permlane16_swap(l,l)[1]", "permlane32_swap(l,l)[0]", "permlane32_swap(l,l)[1]", "permlane16_swap(l,100+l)[0]", "permlane16_swap(l,100+l)[1]", "shfl max", "builtin: r[0] + r[1]", "inline asm: a + b", "x + shfl_xor(x, 16)"};
    for (int r = 0; r < 10; ++r) { printf("%-28s", names[r]); for (int i = 0; i < 64; ++i) printf(" %u", h[r * 64 + i]); printf("\n"); }
    return 0;
}
