// Optional in-library kernel timing (bench.py's `roofline` / `roofline_mfma` objects). When enabled, every GEMM / attention
// launcher claims a record holding an event pair and passes the pair to hipExtLaunchKernelGGL, which stamps the kernel's own
// start and stop (an event packet on either side of a 20-40 us kernel adds ~3 us to what it measures); launchers that
// bracket several kernels (attention pairs) record the pair around them instead.
// Classes: 0 = weight-streaming GEMM with >= 16 MB of weights (work = weight bytes), 1 = LDS-tiled GEMM of a ViT-sized
// problem (work = FLOPs), 2 = attention (work = 0), 3 = small weight-streaming GEMMs (work = weight bytes),
// 4 = LDS-tiled GEMM of an LLM-sized problem (N*K >= 16 M: prefill; work = FLOPs), 5 = split-K reductions behind a weight-streaming GEMM,
// 6 / 8 / 9 = split-K reductions behind a ViT-sized (class 1) / an LLM-sized (class 4) / an fp8 (class 7) LDS-tiled GEMM (work = 0), 7 = fp8 LDS-tiled GEMM (work = FLOPs).
// Thread safety: records are claimed with one atomic fetch_add (bench.py may queue launches from two host threads); a
// record is written only by the thread that claimed it, and cover_profile_end runs after every launcher has returned.
// Launches replayed from a hipGraph never reach the launchers: they carry neither events nor work (both sides of the
// ratio), so bench.py runs its profiled decision without graph replay.
#include <hip/hip_runtime.h>
#include <atomic>
#include <vector>
#include "kernels.h"

namespace {
struct Rec { hipEvent_t a, b; int cls; double work; };
std::vector<Rec> g_pool;                 // sized by cover_profile_begin only (never while profiling is on)
std::atomic<size_t> g_used{0};
std::atomic<bool> g_on{false};
std::atomic<long long> g_dropped{0};

int claim(int cls, double work) {
    if (!g_on.load(std::memory_order_acquire)) return -1;
    const size_t id = g_used.fetch_add(1, std::memory_order_relaxed);
    if (id >= g_pool.size()) { g_dropped.fetch_add(1, std::memory_order_relaxed); return -1; }
    g_pool[id].cls = cls;
    g_pool[id].work = work;
    return (int)id;
}
}  // namespace

bool prof_enabled() { return g_on.load(std::memory_order_acquire); }

int prof_open(hipStream_t st, int cls, double work) {
    const int id = claim(cls, work);
    if (id < 0) return -1;
    if (hipEventRecord(g_pool[id].a, st) != hipSuccess) { g_pool[id].cls = -1; return -1; }
    return id;
}
int prof_reserve(int cls, double work, hipEvent_t* start, hipEvent_t* stop) {
    const int id = claim(cls, work);
    if (id < 0) return -1;
    *start = g_pool[id].a;
    *stop = g_pool[id].b;
    return id;
}
void prof_close(hipStream_t st, int id) {
    if (id >= 0 && hipEventRecord(g_pool[id].b, st) != hipSuccess) g_pool[id].cls = -1;
}

extern "C" int cover_profile_begin(int max_events) {
    if (max_events <= 0 || g_on.load()) return COVER_EINVAL;
    while ((int)g_pool.size() < max_events) {
        Rec r;
        r.cls = -1; r.work = 0;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return COVER_EHIP;
        g_pool.push_back(r);
    }
    g_used.store(0);
    g_dropped.store(0);
    g_on.store(true, std::memory_order_release);
    return COVER_OK;
}

// ms[n_cls], count[n_cls], work[n_cls] (n_cls <= COVER_PROF_CLASSES); synchronises the device. Returns COVER_EWORKSPACE when
// the pool overflowed (records were dropped: the sums are then incomplete and must not be used).
extern "C" int cover_profile_end_n(double* ms, long long* count, double* work, int n_cls) {
    g_on.store(false, std::memory_order_release);
    if (n_cls <= 0 || n_cls > COVER_PROF_CLASSES) return COVER_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return COVER_EHIP;
    for (int c = 0; c < n_cls; ++c) { ms[c] = 0; count[c] = 0; work[c] = 0; }
    size_t used = g_used.load();
    if (used > g_pool.size()) used = g_pool.size();
    for (size_t i = 0; i < used; ++i) {
        const int c = g_pool[i].cls;
        if (c < 0 || c >= n_cls) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_pool[i].a, g_pool[i].b) != hipSuccess) continue;
        ms[c] += t;
        count[c] += 1;
        work[c] += g_pool[i].work;
    }
    g_used.store(0);
    return g_dropped.load() > 0 ? COVER_EWORKSPACE : COVER_OK;
}
extern "C" int cover_profile_end(double* ms, long long* count, double* work) { return cover_profile_end_n(ms, count, work, 4); }

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
