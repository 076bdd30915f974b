// Optional in-library kernel timing with hipEvents on the launch stream (bench.py's `roofline` object):
// when enabled, the tiled-GEMM / attention launchers bracket each launch with an event pair taken from a pre-created pool;
// the weight-streaming launchers pass the pair to hipExtLaunchKernelGGL instead, which stamps the kernel's own start and
// stop (an event packet on either side of a 20-40 us kernel adds ~3 us to what it measures).
// Classes: 0 = weight-streaming GEMM with >= 16 MB of weights (work = weight bytes), 1 = LDS-tiled GEMM (work = FLOPs),
// 2 = attention (work = 0), 3 = small weight-streaming GEMMs (work = weight bytes).
#include <hip/hip_runtime.h>
#include <vector>
#include "kernels.h"

namespace {
struct Rec { hipEvent_t a, b; int cls; double work; };
std::vector<Rec> g_pool;
size_t g_used = 0;
bool g_on = false;
}  // namespace

bool prof_enabled() { return g_on; }

int prof_open(hipStream_t st, int cls, double work) {
    if (!g_on || g_used >= g_pool.size()) return -1;
    Rec& r = g_pool[g_used];
    r.cls = cls;
    r.work = work;
    if (hipEventRecord(r.a, st) != hipSuccess) return -1;
    return (int)g_used++;
}
int prof_reserve(int cls, double work, hipEvent_t* start, hipEvent_t* stop) {
    if (!g_on || g_used >= g_pool.size()) return -1;
    Rec& r = g_pool[g_used];
    r.cls = cls;
    r.work = work;
    *start = r.a;
    *stop = r.b;
    return (int)g_used++;
}
void prof_close(hipStream_t st, int id) {
    if (id >= 0) (void)hipEventRecord(g_pool[id].b, st);
}

extern "C" int cover_profile_begin(int max_events) {
    if (max_events <= 0) return COVER_EINVAL;
    while ((int)g_pool.size() < max_events) {
        Rec r;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return COVER_EHIP;
        g_pool.push_back(r);
    }
    g_used = 0;
    g_on = true;
    return COVER_OK;
}

// ms[4], count[4], work[4]; synchronises the device
extern "C" int cover_profile_end(double* ms, long long* count, double* work) {
    g_on = false;
    if (hipDeviceSynchronize() != hipSuccess) return COVER_EHIP;
    for (int c = 0; c < 4; ++c) { ms[c] = 0; count[c] = 0; work[c] = 0; }
    for (size_t i = 0; i < g_used; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_pool[i].a, g_pool[i].b) != hipSuccess) continue;
        const int c = g_pool[i].cls;
        if (c < 0 || c > 3) continue;
        ms[c] += t;
        count[c] += 1;
        work[c] += g_pool[i].work;
    }
    g_used = 0;
    return COVER_OK;
}

// ---- diagnostic: pure streaming read (HBM ceiling for a given byte count and launch shape), not part of the ABI ----
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__global__ __launch_bounds__(512) void stream_read_k(const u32x4_t* __restrict__ src, size_t n16, unsigned* sink, int nt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4_t acc = {0, 0, 0, 0};
    // each wave reads 1 KiB per instruction, 8 in flight, grid-strided (consecutive waves read consecutive KiB)
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = nt ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    for (; i < n16; i += stride) acc ^= src[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
extern "C" int cover_debug_stream_read(const void* src, size_t bytes, int blocks, int nt, void* sink, void* stream) {
    hipLaunchKernelGGL(stream_read_k, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const u32x4_t*)src, bytes / 16, (unsigned*)sink, nt);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
