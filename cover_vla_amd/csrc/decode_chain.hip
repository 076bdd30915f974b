// Persistent decode chain: the weight-streaming projections of a candidate-decode layer (M <= 32 rows) as PHASES OF ONE LAUNCH.
//
//   launch l:   o_proj(l) + residual  ->  [RMSNorm]  gate_up(l) + SiLU-GLU  ->  down(l) + residual  ->  [RMSNorm]  qkv(l + 1)
//   (the fused decode attention of layer l + 1 stays its own launch between two chain launches; the first launch of a pass is
//    sums-of-squares(x) -> [RMSNorm] qkv(0), the last one ends after down)
//
// Why: as separate launches the seven kernels of a layer-step cost ~108 us for 405 MB of weights (3.7 TB/s): every launch pays its
// own ramp (first weights ~2 us after launch, first activation chunk later still), its tail, the boundary, and the two split-K
// reduce + norm launches stream nothing at all. Here every workgroup (one per CU, 256) requests the first WINDOW of the next
// phase's weights (24 KiB per wave, 48 MiB chip-wide = ~8 us of HBM time) BEFORE it waits at the seam, so HBM keeps streaming while
// the grid synchronises, and the seams themselves get shorter:
//   * every phase gives a workgroup WHOLE output columns (n-blocks x the full K): no split-K slabs, no reduction launches;
//   * RMSNorm is split in two: the producer leaves, beside its 16 output columns, the rows' partial sums of squares (one fp32 per
//     row per workgroup); the consumer adds the 256 partials of a row in a fixed order and applies the norm while it stages the
//     activation chunk into LDS (same arithmetic and rounding points as rmsnorm_bf16_k / splitk_reduce_norm);
//   * hand-off = MI355X guide, Guideline 16: payload stored write-through (agent-scope relaxed atomic stores -> `sc1`), every
//     storing wave drains vmcnt, ONE lane arrives on a counter; ONE wave polls ONE word relaxed with s_sleep, ONE agent acquire,
//     block barrier, then plain loads. The barrier is sense-reversing and hierarchical (8 group counters -> top counter ->
//     per-group generation words) and every spin is BOUNDED: on a timeout the workgroup records an error code, stops waiting for
//     the rest of the launch (the results are then garbage, the launch still ends) and the host reports it
//     (cover_decode_chain_status; the model classes check it at their next host sync when the chain is switched on).
//     The barrier words belong to the PASS: they are carved from the decoder's pass workspace (one decoder, one stream) and zeroed
//     by a kernel in front of the pass's first launch, so two decoders on two streams never count on the same words and a pass
//     that gave up leaves nothing behind; only the sticky status word is per device.
// Residency: 256 workgroups of 512 threads with ~112 KiB of LDS each = one per CU on every CU. The launcher asks the occupancy API
// once per device (and refuses devices with fewer CUs), and serialises chain launches that arrive on DIFFERENT streams with an event
// (two such grids in flight at once would share the CUs and both spin at their first barrier until the bound).
//
// A workgroup = 8 waves; wave w owns the 128-deep k-slice w of every 1024-deep activation chunk for ALL of the workgroup's
// n-blocks; weights go HBM -> VGPR (non-temporal, 1 KiB per wave instruction, fragment-major packing) -> MFMA, activations through a
// double-buffered 64 KiB LDS chunk in fragment-major order; the k-slices are summed through LDS at the end of a phase. The phase
// bodies are fully unrolled per (n-blocks per workgroup, K) so that every wait is a counted vmcnt (see gemm_skinny3).
// Shapes: dim = Hq * D = 4096, mlp = 11008 (Llama-2-7B); anything else takes the separate-launch path.
//
// OUTCOME (MI355X, round 4; per-phase timelines and ablations in docs/OPTIMISATION_LOG.md): correct (fused == split phases bit for bit,
// also with co-running tenants; HF pin G3), and at PARITY with the separate kernels, not ahead: 114 vs 109.6 us per layer. The seams
// behave as designed (7.6 / 4.8 us, covered by the 48 MiB window that loads meanwhile); what eats the gain at M = 32 is the ACTIVATION
// panel: with whole output columns per workgroup every CU reads the full M x K panel of every phase (367 MB per layer from L2 beside
// 405 MB of weights from HBM), and a CU pulls ~60 GB/s whatever the source -- the down projection (704 KiB of activations + 352 KiB of
// weights per CU) is bound by that, not by HBM. Split-K (what the separate kernels do) trades that traffic for partial slabs and a
// reduction launch and comes out the same. Opt-in: COVER_DECODE_CHAIN=1.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "kernels.h"
#include "decode_chain.h"

namespace {

enum { EPI_PLAIN = 0, EPI_RESIDUAL_SSQ = 1, EPI_GLU = 2 };
enum { PH_GEMM = 0, PH_SSQ = 1 };

struct ChainPhase {
    int kind;                  // PH_GEMM / PH_SSQ (sums of squares of A's rows only: feeds the first norm of a pass)
    int nbw;                   // n-blocks (16 output columns) per workgroup: 1, 3 or 6
    int ktot;                  // contraction length: 4096 or 11008
    int N;                     // GEMM columns (2 * mlp for the GLU phase)
    const bf16_t* A; int lda;  // activation panel [M][lda]
    const bf16_t* Wp;          // packed weights [N/16][K/32][64 lanes][8]
    int norm_in, norm_style;   // 1: A's rows go through RMSNorm while they are staged (rstd from ssq_in)
    float norm_eps, norm_w_offset;
    const float* norm_w; const float* ssq_in;
    int epi, act;
    bf16_t* C; int ldc;        // EPI_RESIDUAL_SSQ: C is the residual stream, updated in place
    float* ssq_out;            // EPI_RESIDUAL_SSQ / PH_SSQ: [M][256] partial sums of squares, one per workgroup
    const float* bias;         // EPI_PLAIN only (may be NULL)
};
struct ChainArgs {
    ChainPhase ph[4];
    int n_phases, M;
    unsigned* sync;            // BARRIER_WORDS words of the PASS workspace (one decoder, one stream), zeroed by a kernel at the start of every pass
    unsigned* err;             // status word of the device (sticky give-up code; shared by every pass: it carries no barrier state)
};

constexpr int NWG = 256;                       // workgroups = CUs
constexpr int NGRP = 8;                        // barrier groups (key = blockIdx & 7: the XCD a block is observed to land on)
constexpr int LINE = 32;                       // words per 128-byte line
constexpr int BARRIER_WORDS = (2 * NGRP + 1) * LINE;   // group counters, top counter, per-group generation words
constexpr unsigned SPIN_LIMIT = 1u << 21;      // polls before a workgroup gives up (~1 s)
constexpr int KC = 1024;                       // a wave owns one 128-deep slice of every 1024-deep chunk of K
constexpr int LDS_RED = 6 * 16 * 1024;         // k-slice reduction buffer: [8 waves][NBW <= 6][2][4][64 lanes] fp32
constexpr int LDS_XT = 0;                      // wave-private activation scratch, 8 x 4 KiB (aliases the reduction buffer: used before it)
constexpr int LDS_NORMW = LDS_RED, LDS_RSTD = LDS_NORMW + 4096 * 4, LDS_TOTAL = LDS_RSTD + 32 * 4;

#ifdef COVER_DC_DEBUG
}  // namespace
__device__ unsigned long long g_dc_dbg[256 * 4 * 8];   // [workgroup][phase of the launch][slot]: 100 MHz wall clock, thread 0
extern "C" int cover_dc_debug(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dc_dbg), sizeof(g_dc_dbg)); }
namespace {
#define DCT(slot) do { if (threadIdx.x == 0) g_dc_dbg[(blockIdx.x * 4 + (sy.p & 3)) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define DCT(slot) do { } while (0)
#endif
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ unsigned ld_agent(unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st8_agent(void* p, uint32_t lo, uint32_t hi) {   // write-through 8-byte payload store
    __hip_atomic_store((gu64*)p, ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stf_agent(float* p, float v) { st_agent((unsigned*)p, __float_as_uint(v)); }

struct Sync {
    unsigned* w;
    unsigned* e;       // the device's status word
    unsigned g0;       // generation at launch start
    int k;             // barriers passed in this launch
    int p;             // phase index of the launch (debug stamps)
    bool dead;         // a spin ran into its bound: stop waiting
    __device__ __forceinline__ unsigned* cnt(int g) const { return w + g * LINE; }
    __device__ __forceinline__ unsigned* top() const { return w + NGRP * LINE; }
    __device__ __forceinline__ unsigned* gen(int g) const { return w + (NGRP + 1 + g) * LINE; }
    __device__ __forceinline__ unsigned* err() const { return e; }
};

// ONE lane of the workgroup (wave 0, lane 0): arrive. Called after every storing wave has drained its stores and the block barrier.
__device__ __forceinline__ void grid_arrive(Sync& s) {
    const int g = blockIdx.x & (NGRP - 1);
    const unsigned gsize = (unsigned)((gridDim.x - g + NGRP - 1) / NGRP);
    const unsigned target = s.g0 + (unsigned)(s.k + 1);
    const unsigned old = __hip_atomic_fetch_add((gu32*)s.cnt(g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gsize - 1) {                                        // last of the group: reset, then one arrival at the top
        (void)__hip_atomic_exchange((gu32*)s.cnt(g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the reset is done before anybody can be released
        const unsigned ngroups = gridDim.x < NGRP ? gridDim.x : NGRP;
        const unsigned old2 = __hip_atomic_fetch_add((gu32*)s.top(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old2 == ngroups - 1) {
            (void)__hip_atomic_exchange((gu32*)s.top(), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int i = 0; i < NGRP; ++i) st_agent(s.gen(i), target);
        }
    }
}
// The same lane: wait for the release of barrier k + 1 (the caller's wave then issues ONE agent-scope acquire, a block barrier follows).
__device__ __forceinline__ void grid_wait(Sync& s) {
    const int g = blockIdx.x & (NGRP - 1);
    const unsigned target = s.g0 + (unsigned)(s.k + 1);
    if (!s.dead) {
        unsigned spins = 0;
        while (ld_agent(s.gen(g)) != target) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && ld_agent(s.err()) != 0u)) {
                st_agent(s.err(), 0x1000u + (unsigned)s.k);      // give-up code: barrier index of this launch
                s.dead = true;
                break;
            }
        }
    }
}

// RMSNorm of 8 consecutive elements of one row (rmsnorm_bf16_k arithmetic), branch-free: `wl` holds the EFFECTIVE weights the phase
// put into LDS -- w for style 1 (Llama: w * bf16(x * rstd)), off + w otherwise (Gemma: (x * rstd) * (off + w)) -- and `rnd` selects the
// intermediate bf16 rounding of style 1.
__device__ __forceinline__ uint4 norm8(uint4 raw, float rstd, const float* wl, bool rnd) {
    const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
    const float4 w0 = *(const float4*)wl, w1 = *(const float4*)(wl + 4);
    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    float o[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = bf2f((bf16_t)(rw[i] & 0xffffu)) * rstd, b = bf2f((bf16_t)(rw[i] >> 16)) * rstd;
        const float ar = bfround(a), br = bfround(b);
        o[2 * i] = (rnd ? ar : a) * wv[2 * i];
        o[2 * i + 1] = (rnd ? br : b) * wv[2 * i + 1];
    }
    return make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
}

// NBW fixes the phase's role: 1 = o_proj / down (residual + sums of squares, raw input rows), 3 = qkv (plain store, normed input rows),
// 6 = gate_up (GLU epilogue, normed input rows).
//
// Inside a phase NOTHING is shared between the waves of a workgroup until the final k-slice sum: wave w owns k-slice w of every
// 1024-deep chunk for all of the workgroup's n-blocks, so it is also the ONLY consumer of that slice of the activation rows. Each wave
// therefore loads its own MFMA B-operand fragments straight from global memory (row f * 16 + rr, k = kst * 32 + gg * 8: 16 bytes per
// lane, 1 KiB per instruction, L2 hits: every workgroup reads the same panel), applies the RMSNorm to them in registers, and never
// touches LDS or a block barrier in the main loop. (A first version staged 64 KiB chunks through LDS like gemm_skinny3, whose two
// n-groups do share them: the 1-n-block phases then moved 2 bytes of activations through LDS per byte of weights behind a block barrier
// per chunk and ran at half the HBM rate.)
template <int NBW, int KTOT>
__device__ __forceinline__ void gemm_phase(const ChainPhase& ph, int M, char* smem, Sync& sy, bool wait_seam, bool arrive_after) {
    constexpr bool NORM = NBW >= 3;
    constexpr int EPI = NBW <= 2 ? EPI_RESIDUAL_SSQ : (NBW == 3 ? EPI_PLAIN : EPI_GLU);
    constexpr int NCH = (KTOT + KC - 1) / KC;            // 1024-deep chunks; this wave's slice of a chunk = 128 k
    constexpr int NP = NCH * 2;                           // step PAIRS of a wave: pair p = 64 k = (chunk p / 2, half p % 2 of its slice)
    constexpr int DP = NBW <= 2 ? 6 : (NBW == 3 ? 3 : 2); // pairs in flight: DP x (2 NBW weight KiB + 4 activation KiB) per wave
    constexpr int K32 = KTOT / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N16 = ph.N >> 4;
    const int nb_begin = blockIdx.x * NBW;
    const bool active = nb_begin < N16;                   // (the GLU phase fills 230 of the 256 workgroups)
    const int kw0 = w * 128;                              // this wave's k-slice inside every chunk
    u32x4 ws[DP][2][NBW];                                 // weight fragments of a pair: [k-step][n-block], one 1 KiB load each
    uint4 xl[DP][2][2];                                   // activation LINES of a pair: [row half f][8-row tile h]: lane = (row h*8 + l/8, 16 B l%8)
    f32x4 acc[NBW][2];
#pragma unroll
    for (int i = 0; i < NBW; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // first k of pair p (compile-time p); a slice beyond K (ragged last chunk) re-reads valid weights against zero activations
    auto k_of = [&](int p) { return (p >> 1) * KC + kw0 + (p & 1) * 64; };
    auto w_load = [&](u32x4(&dst)[2][NBW], int p) {
        int k = k_of(p);
        k = k + 64 <= KTOT ? k : KTOT - 64;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < NBW; ++i) {
                int nb = nb_begin + i;
                nb = nb < N16 ? nb : N16 - 1;              // clamped: the output of a clamped n-block is never stored
#if defined(COVER_DC_ABL) && (COVER_DC_ABL & 1)
                dst[u][i] = (u32x4){(unsigned)k, (unsigned)nb, 0x3f803f80u, 0x3f803f80u};   // ablation: no weight stream
#else
                dst[u][i] = __builtin_nontemporal_load((const u32x4*)(ph.Wp + ((size_t)nb * K32 + (k >> 5) + u) * 512) + lane);
#endif
            }
    };

    // ---- seam: weights first (they depend on nothing), then the wait ----
    DCT(0);
    if (active) {
#pragma unroll
        for (int p = 0; p < DP; ++p)
            if (p < NP) w_load(ws[p], p);
    }
    DCT(1);
    if constexpr (NORM) {                                  // the norm weights of the 4096-wide panel -> LDS (immutable: no ordering needed)
        float* nwl = (float*)(smem + LDS_NORMW);
        float4 a = *(const float4*)(ph.norm_w + tid * 8), b = *(const float4*)(ph.norm_w + tid * 8 + 4);
        const float off = ph.norm_style == 1 ? 0.0f : ph.norm_w_offset;   // effective weights: see norm8
        a.x += off; a.y += off; a.z += off; a.w += off; b.x += off; b.y += off; b.z += off; b.w += off;
        *(float4*)(nwl + tid * 8) = a;
        *(float4*)(nwl + tid * 8 + 4) = b;
    }
    // ONE lane polls ONE word (bounded, relaxed). Its wave's window loads were issued first and return first (loads return in order): the
    // poll costs that wave the window's latency only when the barrier is already complete, and the seam is then short anyway
    if (wait_seam && tid == 0) grid_wait(sy);
    if (wait_seam && w == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // ONE acquire per workgroup; the block barrier below publishes it
    if (wait_seam) sy.k += 1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only: the weight windows stay in flight
    DCT(2);

    const int rr = lane & 15, gg = lane >> 4;             // MFMA operand coordinates
    const int r8 = lane >> 3, j8 = lane & 7;              // line coordinates: row r8 of an 8-row tile, 16-byte piece j8 of its 128-byte line
    if (!active) {                                         // nothing to compute in this phase: only the seam behind it
        if (arrive_after) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tid == 0) grid_arrive(sy);
        }
        return;
    }

    // ---- activations. Every workgroup reads the whole panel (M x K, L2-resident), so HOW it is read decides the phase: MFMA-operand
    //      shaped loads (16 rows x 64 B per instruction: half lines) come out of the L2s at ~8 TB/s chip-wide -- measured: the 180 MB
    //      that the 256 workgroups of the down projection read took 22.7 us, the 90 MB weight stream beside them 11.8 -- whole 128-byte
    //      lines at several times that. So a lane group reads LINES (8 rows x 128 B = the 64 k of a step pair per instruction), the RMSNorm
    //      is applied to them in registers, and a wave-private 4 KiB LDS scratch turns them into the four operand fragments of the pair
    //      (XOR-swizzled 16-byte slots: writes and reads conflict-free; same wave: no barrier, LDS ops are in order).
    const bf16_t* arow[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int row = f * 16 + h * 8 + r8;
            row = row < M ? row : M - 1;
            arow[f][h] = ph.A + (size_t)row * ph.lda + kw0 + j8 * 8;
        }
    auto x_load = [&](uint4 (&dst)[2][2], int p) {         // p compile-time
        const int koff = (p >> 1) * KC + (p & 1) * 64;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#if defined(COVER_DC_ABL) && (COVER_DC_ABL & 2)
                if (true) { dst[f][h] = make_uint4((unsigned)koff, 0x3f803f80u, 0x3f803f80u, (unsigned)f); } else   // ablation: no activation loads
#endif
                if (KTOT % KC == 0) {                      // every chunk whole
                    dst[f][h] = *(const uint4*)(arow[f][h] + koff);
                } else {                                   // ragged last chunk: slices beyond K contribute zeros
                    const bool in = koff + kw0 + 64 <= KTOT;
                    uint4 v = *(const uint4*)(arow[f][h] + (in ? koff : 0));
                    if (!in) v = make_uint4(0u, 0u, 0u, 0u);
                    dst[f][h] = v;
                }
            }
    };
    const float* nwl = (const float*)(smem + LDS_NORMW);
    char* xt = smem + LDS_XT + w * 4096;                   // [f][h][row r8: 128 B, slot j ^ r8]
    const int wr_off = r8 * 128 + ((j8 ^ r8) << 4);
    int rd_off[2];                                         // fragment (k-step u) of rows rr: tile h = rr / 8, row rr % 8, piece u * 4 + gg
#pragma unroll
    for (int u = 0; u < 2; ++u) rd_off[u] = (rr >> 3) * 1024 + (rr & 7) * 128 + (((u * 4 + gg) ^ (rr & 7)) << 4);
#pragma unroll
    for (int p = 0; p < DP; ++p)
        if (p < NP) x_load(xl[p], p);
    // ---- per-row 1 / rms from the 256 partial sums of squares (fixed order: 4 sequential per lane, then the wave butterfly), computed
    //      UNDER the latency of the activation window requested above ----
    float rstd[2][2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};       // of this lane's line rows f * 16 + h * 8 + r8
    if constexpr (NORM) {
        float* rl = (float*)(smem + LDS_RSTD);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int row = w * 4 + j;
            row = row < M ? row : M - 1;
            const float4 q = *(const float4*)(ph.ssq_in + (size_t)row * NWG + lane * 4);
            float sq = q.x;
            sq += q.y; sq += q.z; sq += q.w;
            sq = wave_sum(sq);
            if (lane == 0) rl[w * 4 + j] = rsqrtf(sq / (float)KTOT + ph.norm_eps);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = f * 16 + h * 8 + r8;
                rstd[f][h] = rl[row < M ? row : M - 1];
            }
    }
    DCT(3);

    // ---- main loop: fully unrolled; a pair's loads were issued DP pairs earlier, its slots are refilled in place right behind its MFMAs;
    //      no block barrier: the eight waves drift freely ----
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int slot = p % DP;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint4 v = xl[slot][f][h];
                if constexpr (NORM) v = norm8(v, rstd[f][h], nwl + (p >> 1) * KC + kw0 + (p & 1) * 64 + j8 * 8, ph.norm_style == 1);
                *(uint4*)(xt + (f * 2 + h) * 1024 + wr_off) = v;
            }
        bf16x8 xf[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int f = 0; f < 2; ++f) xf[u][f] = as_bf16x8(*(const uint4*)(xt + f * 2048 + rd_off[u]));
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < NBW; ++i) {
                const bf16x8 wf = __builtin_bit_cast(bf16x8, ws[slot][u][i]);
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[u][f], acc[i][f], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (p + DP < NP) {
            x_load(xl[slot], p + DP);
            w_load(ws[slot], p + DP);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef COVER_DC_DEBUG
        if (p == 0) { asm volatile("s_nop 0" :: "v"(acc[0][0][0])); DCT(4); }
#endif
    }
#ifdef COVER_DC_DEBUG
    asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[NBW - 1][1][3]));
#endif
    DCT(5);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with its scratch (the reduction buffer aliases it)

    // ---- sum the 8 k-slices through LDS (fixed wave order), epilogue, write-through stores ----
    float* red = (float*)smem;                             // [w][i][f][e][lane]
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[((((w * NBW + i) * 2 + f) * 4 + e) << 6) + lane] = acc[i][f][e];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    DCT(6);
    const int r = lane & 15, g = lane >> 4;
    auto slice_sum = [&](int i, int f, float (&v)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += red[((((ww * NBW + i) * 2 + f) * 4 + e) << 6) + lane];
    };
    if constexpr (EPI == EPI_GLU) {
        {
            constexpr int NU = (NBW / 2) * 2;               // (gate, up) pairs x 2 row halves
            for (int j = w; j < NU; j += 8) {
                const int pi = j >> 1, f = j & 1;
                const int nb = nb_begin + 2 * pi, m = f * 16 + r;
                float gv[4], uv[4];
                slice_sum(2 * pi, f, gv);
                slice_sum(2 * pi + 1, f, uv);
                if (nb + 1 < N16 && m < M) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {           // epi_store4_glu's arithmetic
                        float gg = bfround(gv[e]);
                        const float uu = bfround(uv[e]);
                        gg = bfround(act_apply_bf16(gg, ph.act));
                        o[e] = bfround(gg * uu);
                    }
                    st8_agent(ph.C + (size_t)m * ph.ldc + (nb >> 1) * 16 + 4 * g, pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
                }
            }
        }
    } else {
        constexpr int NU = NBW * 2;
        for (int j = w; j < NU; j += 8) {
            const int i = j >> 1, f = j & 1;
            const int nb = nb_begin + i, m = f * 16 + r, n0 = nb * 16 + 4 * g;
            float v[4];
            slice_sum(i, f, v);
            const bool ok = nb < N16 && m < M;
            if constexpr (EPI == EPI_PLAIN) {
                if (ok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ph.bias ? v[e] + ph.bias[n0 + e] : v[e];
                    st8_agent(ph.C + (size_t)m * ph.ldc + n0, pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                }
            } else {                                        // x = bf16(bf16(o) + x), partial sum of squares of the STORED values
                float q = 0.f;
                if (ok) {
                    bf16_t* xp = ph.C + (size_t)m * ph.ldc + n0;
                    const uint2 xo = *(const uint2*)xp;
                    const float xv[4] = {bf2f((bf16_t)(xo.x & 0xffffu)), bf2f((bf16_t)(xo.x >> 16)), bf2f((bf16_t)(xo.y & 0xffffu)), bf2f((bf16_t)(xo.y >> 16))};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = bfround(bfround(v[e]) + xv[e]);
                        q += v[e] * v[e];
                    }
                    st8_agent(xp, pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                }
                q += __shfl_xor(q, 16);                     // the row's 16 columns: 4 lanes (g = 0..3)
                q += __shfl_xor(q, 32);
                if (ok && g == 0) stf_agent(ph.ssq_out + (size_t)m * NWG + nb, q);
            }
        }
    }
    if (arrive_after) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every storing wave has drained
        DCT(7);
        if (tid == 0) grid_arrive(sy);
    } else {
        DCT(7);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                          // LDS is reused by the next phase
    }
}

// Partial sums of squares of A's rows over this workgroup's 16 columns (the norm in front of the first GEMM of a pass).
__device__ __forceinline__ void ssq_phase(const ChainPhase& ph, int M, Sync& sy, bool arrive_after) {
    const int tid = threadIdx.x;
    const int row = tid >> 4, col = blockIdx.x * 16 + (tid & 15);
    float q = 0.f;
    if (row < M) {
        const float v = bf2f(ph.A[(size_t)row * ph.lda + col]);
        q = v * v;
    }
    q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4); q += __shfl_xor(q, 8);
    if (row < M && (tid & 15) == 0) stf_agent(ph.ssq_out + (size_t)row * NWG + blockIdx.x, q);
    if (arrive_after) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid == 0) grid_arrive(sy);
    }
}

__global__ __launch_bounds__(512) void decode_chain_k(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Sync sy;
    sy.w = a.sync;
    sy.e = a.err;
    sy.k = 0;
    sy.dead = false;
    {   // generation at launch start: it cannot advance before this workgroup has arrived at the first barrier
        unsigned g0 = 0;
        if (threadIdx.x == 0) g0 = ld_agent(sy.gen(blockIdx.x & (NGRP - 1)));
        sy.g0 = __builtin_amdgcn_readfirstlane(g0);
    }
    for (int p = 0; p < a.n_phases; ++p) {
        const ChainPhase& ph = a.ph[p];
        sy.p = p;
        const bool seam = p > 0, more = p + 1 < a.n_phases;
        if (ph.kind == PH_SSQ) { ssq_phase(ph, a.M, sy, more); continue; }
        if (ph.nbw == 1 && ph.ktot == 4096) gemm_phase<1, 4096>(ph, a.M, smem, sy, seam, more);
        else if (ph.nbw == 1) gemm_phase<1, 11008>(ph, a.M, smem, sy, seam, more);
        else if (ph.nbw == 3) gemm_phase<3, 4096>(ph, a.M, smem, sy, seam, more);
        else gemm_phase<6, 4096>(ph, a.M, smem, sy, seam, more);
    }
}

unsigned* g_err[16] = {};          // per device: ONE status word (sticky give-up code), never barrier state
hipEvent_t g_last_ev[16] = {};     // per device: completion of the last chain launch, and the stream it went to (see chain_serialise)
hipStream_t g_last_st[16] = {};
bool g_have_last[16] = {};
hipError_t g_attr[16];
bool g_attr_done[16] = {};
int g_resident[16] = {};           // 0 = not asked yet, 1 = 256 workgroups of decode_chain_k fit the device at once, -1 = they do not
std::mutex g_mu;

__global__ void chain_zero_k(unsigned* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------
// once per DEVICE: the LDS attribute of the kernel, and whether one workgroup per CU on every CU is what the hardware will admit (the
// software grid barrier needs all 256 workgroups resident at once)
static hipError_t chain_device_setup(int dev) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_attr_done[dev]) {
        g_attr[dev] = hipFuncSetAttribute((const void*)decode_chain_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        g_attr_done[dev] = true;
        int per_cu = 0;
        hipDeviceProp_t p;
        if (g_attr[dev] == hipSuccess && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_chain_k, 512, LDS_TOTAL) == hipSuccess &&
            hipGetDeviceProperties(&p, dev) == hipSuccess)
            g_resident[dev] = (per_cu >= 1 && p.multiProcessorCount >= NWG) ? 1 : -1;
        else
            g_resident[dev] = -1;
    }
    return g_attr[dev];
}

bool decode_chain_supported(const cover_dec_desc* d, int rows) {
    // OPT-IN (COVER_DECODE_CHAIN=1; 2 = every phase its own launch). Measured on MI355X (docs/OPTIMISATION_LOG.md, round 4): the chain
    // runs a 7B decode layer in 114 us against 109.6 us for the separate kernels, the headline decision in 34.9-35.3 ms against 34.7:
    // parity, so the separate launches stay the default. Read per call (the tests A/B the two paths inside one process).
    const char* env = getenv("COVER_DECODE_CHAIN");
    if (!env || (env[0] != '1' && env[0] != '2')) return false;
    if (rows < 1 || rows > 32) return false;
    if (d->dim != 4096 || d->Hq * d->D != 4096 || d->mlp != 11008 || (d->Hq + 2 * d->Hkv) * d->D != 12288) return false;
    if (d->act != ACT_SILU && d->act != ACT_GELU_TANH) return false;
    for (int l = 0; l < d->n_layers; ++l)
        if (d->layers_host[l].qkv_w8) return false;         // e4m3 weight stream: separate-launch path
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
    if (chain_device_setup(dev) != hipSuccess) return false;
    return g_resident[dev] == 1;
}
// [2][32][NWG] partial sums of squares, then the BARRIER_WORDS of this pass
size_t decode_chain_ws_bytes() { return (size_t)2 * 32 * NWG * sizeof(float) + (size_t)BARRIER_WORDS * sizeof(unsigned); }
static unsigned* chain_barrier_words(float* ssq) { return (unsigned*)(ssq + 2 * 32 * NWG); }

static hipError_t chain_err_word(int dev, unsigned** out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_err[dev]) {
        unsigned* p = nullptr;
        hipError_t e = hipMalloc((void**)&p, LINE * sizeof(unsigned));
        if (e != hipSuccess) return e;
        e = hipMemset(p, 0, LINE * sizeof(unsigned));
        if (e != hipSuccess) return e;
        g_err[dev] = p;
    }
    *out = g_err[dev];
    return hipSuccess;
}

// Two chain launches must never be in flight at once on one device: each wants every CU (112 KiB of LDS per workgroup), so workgroups of
// two launches from different streams would share the CUs between them and both grids would spin at their first barrier until the bound.
// A launch on another stream than the previous one therefore waits for the previous one's completion event (stream-ordered launches of
// ONE stream already serialise). Capturing streams are left alone: an event wait across a capture boundary is not legal, and a captured
// decode loop replays on the stream that owns it.
static hipError_t chain_serialise_before(int dev, hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return hipSuccess;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_have_last[dev] && g_last_st[dev] != st) return hipStreamWaitEvent(st, g_last_ev[dev], 0);
    return hipSuccess;
}
static void chain_serialise_after(int dev, hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_last_ev[dev] && hipEventCreateWithFlags(&g_last_ev[dev], hipEventDisableTiming) != hipSuccess) return;
    if (hipEventRecord(g_last_ev[dev], st) == hipSuccess) { g_last_st[dev] = st; g_have_last[dev] = true; }
}

// error word of the chain launches on this device since the last call: 0 = every barrier completed; resets the word. (The barrier words
// themselves live in the pass workspace and are zeroed at the start of every pass, so a pass that gave up does not poison the next one.)
int decode_chain_status() {
    int dev = 0;
    unsigned* s = nullptr;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || chain_err_word(dev, &s) != hipSuccess) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    unsigned e = 0;
    if (hipMemcpy(&e, s, sizeof e, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (e != 0) (void)hipMemset(s, 0, sizeof(unsigned));
    return (int)e;
}

static void gemm_ph(ChainPhase& p, int nbw, int ktot, int N, const void* A, int lda, const void* Wp, void* C, int ldc, int epi, int act) {
    memset(&p, 0, sizeof p);
    p.kind = PH_GEMM; p.nbw = nbw; p.ktot = ktot; p.N = N;
    p.A = (const bf16_t*)A; p.lda = lda; p.Wp = (const bf16_t*)Wp;
    p.C = (bf16_t*)C; p.ldc = ldc; p.epi = epi; p.act = act;
}

// stage: 0 = [sums of squares(x) -> norm -> qkv(layer 0)];  1 = [o_proj(l) -> gate_up(l) -> down(l) (-> qkv(l + 1) when next != NULL)]
// split: every phase as its own launch (no in-kernel barrier): the debugging / A-B form, bit-identical to the fused one
hipError_t launch_decode_chain(const cover_dec_desc* d, int stage, const cover_dec_layer* L, const cover_dec_layer* next, void* x, void* qkv,
                               void* attn, void* mlp, float* ssq, int rows, bool split, hipStream_t st) {
    ChainArgs a;
    memset(&a, 0, sizeof a);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    e = chain_device_setup(dev);
    if (e != hipSuccess) return e;
    e = chain_err_word(dev, &a.err);
    if (e != hipSuccess) return e;
    a.sync = chain_barrier_words(ssq);
    e = chain_serialise_before(dev, st);
    if (e != hipSuccess) return e;
    if (stage == 0)   // first launch of a pass: this pass's barrier words start from zero (a kernel, not a memset node: see gemm_bf16.hip "Tail reduction")
        hipLaunchKernelGGL(chain_zero_k, dim3(1), dim3(256), 0, st, a.sync, BARRIER_WORDS);
    a.M = rows;
    const int dim = d->dim, nqkv = (d->Hq + 2 * d->Hkv) * d->D;
    float* ssq_a = ssq;                 // behind o_proj
    float* ssq_b = ssq + 32 * NWG;      // behind down (and of the pass input)
    auto norm_from = [&](ChainPhase& p, const float* w, const float* sq) {
        p.norm_in = 1; p.norm_style = d->norm_style; p.norm_eps = d->norm_eps; p.norm_w_offset = d->norm_w_offset; p.norm_w = w; p.ssq_in = sq;
    };
    int n = 0;
    if (stage == 0) {
        ChainPhase& s = a.ph[n++];
        memset(&s, 0, sizeof s);
        s.kind = PH_SSQ; s.A = (const bf16_t*)x; s.lda = dim; s.ssq_out = ssq_b;
        ChainPhase& q = a.ph[n++];
        gemm_ph(q, 3, 4096, nqkv, x, dim, L->qkv_w, qkv, nqkv, EPI_PLAIN, 0);
        q.bias = L->qkv_b;
        norm_from(q, L->in_norm_w, ssq_b);
    } else {
        ChainPhase& o = a.ph[n++];
        gemm_ph(o, 1, 4096, dim, attn, d->Hq * d->D, L->o_w, x, dim, EPI_RESIDUAL_SSQ, 0);
        o.ssq_out = ssq_a;
        ChainPhase& g = a.ph[n++];
        gemm_ph(g, 6, 4096, 2 * d->mlp, x, dim, L->gate_up_w, mlp, d->mlp, EPI_GLU, d->act);
        norm_from(g, L->post_norm_w, ssq_a);
        ChainPhase& dn = a.ph[n++];
        gemm_ph(dn, 1, 11008, dim, mlp, d->mlp, L->down_w, x, dim, EPI_RESIDUAL_SSQ, 0);
        dn.ssq_out = ssq_b;
        if (next) {
            ChainPhase& q = a.ph[n++];
            gemm_ph(q, 3, 4096, nqkv, x, dim, next->qkv_w, qkv, nqkv, EPI_PLAIN, 0);
            q.bias = next->qkv_b;
            norm_from(q, next->in_norm_w, ssq_b);
        }
    }
    auto launch = [&](const ChainArgs& aa, double bytes) {
        hipEvent_t ea, eb;
        if (prof_enabled() && prof_reserve(0, bytes, &ea, &eb) >= 0)
            hipExtLaunchKernelGGL(decode_chain_k, dim3(NWG), dim3(512), (uint32_t)LDS_TOTAL, st, ea, eb, 0, aa);
        else
            hipLaunchKernelGGL(decode_chain_k, dim3(NWG), dim3(512), LDS_TOTAL, st, aa);
    };
    auto wbytes = [&](const ChainPhase& p) { return p.kind == PH_GEMM ? 2.0 * (double)p.N * (double)p.ktot : 0.0; };
    if (!split) {
        a.n_phases = n;
        double b = 0.0;
        for (int i = 0; i < n; ++i) b += wbytes(a.ph[i]);
        launch(a, b);
    } else {
        for (int i = 0; i < n; ++i) {
            ChainArgs one = a;
            one.ph[0] = a.ph[i];
            one.n_phases = 1;
            launch(one, wbytes(a.ph[i]));
        }
    }
    e = hipGetLastError();
    chain_serialise_after(dev, st);
    return e;
}
