// Fused single-token decode attention for candidate batches (OpenVLA-style decode: every candidate attends
//   [segment 0: one KV block shared by ALL candidates | segment 1: a KV block shared by the candidates of one prompt |
//    segment 2: the candidate's own generated tokens, including the one produced in this pass]).
//
// One launch per layer replaces rope_kv_write + shared-segment attention + per-candidate attention (three dependent
// launches whose cost at this size is pure latency, ~33 us per layer). A block is 16 candidates x one head, 16 waves:
//   phase 1  the block forms its q / k_new / v_new rows -- from the bf16 QKV buffer or directly from the split-K partial
//            sums of the weight-streaming QKV GEMM (+bias, bf16 rounding) -- applies RoPE, hands q / k_new / v_new to the
//            other waves through LDS and appends k_new / v_new to the candidate's own cache segment (nobody waits for
//            those stores: both block barriers are LDS-only);
//   phase 2  32-key tiles, one (rarely two) per wave, waves split into three pools with FIXED roles:
//            pool A  segment 0, tile t -> wave t mod WA; the 16 candidates are the query rows of the tile (a ninth, mostly
//                    empty tile of a 257-key segment goes to the first pool-C wave and a state slot of its own: launcher),
//            pool B  segment 1, the p-th distinct prompt slot of the block -> wave p mod WB, with a column mask,
//            pool C  segment 2: a tile is 4 candidates x 8 own tokens (key slot 8g+e <-> candidate c0+g, token e, so the
//                    V^T operand is ONE 16-byte load per lane) with a block-diagonal mask, candidates 4w..4w+3 -> wave w;
//                    the token written in this pass is taken from LDS, so nobody waits for the cache stores;
//   phase 3  the waves' online-softmax states are merged through LDS in wave order and the rows are stored.
// Pool sizes depend only on len0, and a candidate's non-zero states sit in the same pool-A / pool-C waves and
// in exactly one pool-B wave wherever it is placed, so its result does not depend on its neighbours in the block
// (merging an empty state is exact) -- the prompt-permutation property the sampler tests rely on.
// MFMA scheme (swapped operands, permuted key order, V read transposed) is the one of attention.hip.
// Restrictions: T = 1, Hq == Hkv (MHA), D in {64, 128}, every segment COVER_MASK_LEN.
#include "common.h"
#include "kernels.h"

struct DecAttnDev {
    const bf16_t* qkv; int ld_qkv;
    const float* partial; int n_splits; const float* bias;
    int N, H;
    float scale_log2e;
    const int* positions; const float* cos_t; const float* sin_t; int n_pos, rope_mode;
    // segment 0 (shared, slot fixed), 1 (per prompt), 2 (own, written here); strides in elements
    const bf16_t* k0; const bf16_t* vt0; int k0_t, k0_h, vt0_h, vt0_d, len0;
    const bf16_t* k1; const bf16_t* vt1; long long k1_slot, vt1_slot; int k1_t, k1_h, vt1_h, vt1_d; const int* slot1; const int* len1; int len1_c;
    bf16_t* k2; bf16_t* vt2; long long k2_slot, vt2_slot; int k2_t, k2_h, vt2_h, vt2_d; const int* slot2; int len2, write_t;
    bf16_t* out; long long o_row;
    int WA, WC;   // wave pools: [0, WA) segment 0, [WA, 16 - WC) segment 1, [16 - WC, 16) segment 2
    int tail_tile; // >= 0: this segment-0 tile is taken by the first pool-C wave into the EXTRA state slot (see the launcher)
};

constexpr int DA_NW = 16;

// VS = 2 / 4: the value / output columns of a (16 candidates, head) unit are split over two / four blocks (blockIdx.z): all compute
// the scores from the full K, each loads, multiplies and merges only its D/2 columns of V. The tile phase is bound by the
// bytes one CU can pull in (64 blocks read 11.5 MB at N = 32, H = 32); two CUs at 0.75x the bytes each finish it sooner.
#ifdef COVER_DA_DEBUG
__device__ unsigned long long g_da_dbg[512 * 8];   // per block: start, phase 1 done (LDS hand-off), tile phase done, end
extern "C" int cover_da_debug(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_da_dbg), sizeof(g_da_dbg)); }
#define DAT(slot) do { if (threadIdx.x == 0) g_da_dbg[(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) & 511) * 8 + (slot)] = wall_clock64(); } while (0)
// per WAVE of one workgroup (tile 1, head 7, value block 2): 0 reached the phase-1 barrier, 1 passed it, 2 first tile folded, 3 all tiles folded, 4 role, 5 tiles, 6 kernel start
__device__ unsigned long long g_da_wave[16 * 8];
extern "C" int cover_da_debug_waves(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_da_wave), sizeof(g_da_wave)); }
#define DAW(slot, val) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 1 && blockIdx.y == 7 && blockIdx.z == 2) g_da_wave[(threadIdx.x >> 6) * 8 + (slot)] = (val); } while (0)
#else
#define DAT(slot) do { } while (0)
#define DAW(slot, val) do { } while (0)
#endif
template <int D, int VS>
__global__ __launch_bounds__(1024) void decode_attn_fused_k(DecAttnDev a) {
    DAT(0);
    DAW(6, wall_clock64());
    constexpr int KS = D / 32, DB = D / 16 / VS, HALF = D / 2, DV = D / VS;
    constexpr int OW = DB * 4 * 64;                    // floats of one wave's O state
    constexpr int IPW = (DB * 4 + DA_NW - 1) / DA_NW;  // (db, e) output items merged per wave
    static_assert(DB * 4 % DA_NW == 0 || DB * 4 < DA_NW, "merge items must split evenly over the waves");
    __shared__ __attribute__((aligned(16))) bf16_t qs[16 * D];
    __shared__ __attribute__((aligned(16))) bf16_t kn[16 * D];   // k_new / v_new of the block's candidates (pool C reads them)
    __shared__ __attribute__((aligned(16))) bf16_t vn[16 * D];
    __shared__ float so[(DA_NW + 1) * OW];            // slot DA_NW: the segment-0 tail tile (empty when there is none)
    __shared__ float sm[(DA_NW + 1) * 16];
    __shared__ float sl[(DA_NW + 1) * 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, h = blockIdx.y;
    const int dv0 = blockIdx.z * DV;                  // first value / output column of this block
    const int r = lane & 15, g = lane >> 4;
    const int ncols = 3 * a.H * D;
    const int WB = DA_NW - a.WA - a.WC;

    // ---------------- small index loads first (everything below chains on them) ----------------
    const int cand_r = tile * 16 + r;
    const bool q_ok = cand_r < a.N;
    const int my_slot1 = q_ok ? (a.slot1 ? a.slot1[cand_r] : cand_r) : -1;
    const int my_len1 = q_ok ? (a.len1 ? a.len1[cand_r] : a.len1_c) : 0;
    const bool p1 = tid < 16 * HALF;
    const int pc = tid / HALF, pi = tid - pc * HALF;   // phase-1 item: candidate pc of the block, rotation pair (pi, pi + HALF)
    const int pcand = tile * 16 + pc;
    const bool p1_ok = p1 && pcand < a.N;
    int ppos = 0, pslot2 = pcand;
    if (p1_ok) {
        if (a.rope_mode != 0) ppos = a.positions[pcand];
        if (a.slot2) pslot2 = a.slot2[pcand];
    }
#ifdef COVER_DA_DEBUG   // (timeline builds only: in the product the slab loads below go out BEHIND the index loads without waiting for them --
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   //  only cos / sin chain on the position -- one round trip less in front of phase 1b)
#endif
    DAT(4);
    // ---------------- phase 1a: q / k / v elements (i, i + HALF) of one candidate, summed over the split-K partials ----------------
    float x[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // q1 q2 k1 k2 v1 v2
    float cs = 1.f, sn = 0.f;
    if (p1_ok) {
        unsigned col[6];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            col[2 * j] = (unsigned)((j * a.H + h) * D + pi);
            col[2 * j + 1] = col[2 * j] + HALF;
        }
        // Order of issue = order of return: the index loads above are the oldest, the q / k / v loads below depend on nothing but the
        // candidate number and go out BEHIND them without waiting, and only then the cos / sin entries, which chain on the position:
        // two round trips in front of phase 1b (index -> cos / sin, with the slabs travelling underneath) instead of three
        // (index -> cos / sin -> slabs, what the branches of the former layout made of it).
        float pv[6][4], bs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const unsigned pstride = (unsigned)a.N * (unsigned)ncols;   // (N * ncols * n_splits < 2^31 is checked by the launcher)
        if (a.n_splits <= 0) {
            const unsigned ro = (unsigned)pcand * (unsigned)a.ld_qkv;
#pragma unroll
            for (int j = 0; j < 6; ++j) x[j] = bf2f(a.qkv[ro + col[j]]);
        } else {
            // uniform base pointer per split + 32-bit lane offsets
            const unsigned ro = (unsigned)pcand * (unsigned)ncols;
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {   // (uniform test per slab: a block's 16 waves share one ~60 GB/s load path, and loads of
                if (sidx < a.n_splits) {             //  slabs that are not there -- formerly re-reads of the last one -- cost as much as real ones)
                    const float* ps = a.partial + (size_t)sidx * pstride;
#pragma unroll
                    for (int j = 0; j < 6; ++j) pv[j][sidx] = ps[ro + col[j]];
                } else {
#pragma unroll
                    for (int j = 0; j < 6; ++j) pv[j][sidx] = 0.f;
                }
            }
            if (a.bias) {
#pragma unroll
                for (int j = 0; j < 6; ++j) bs[j] = a.bias[col[j]];
            }
        }
        if (a.rope_mode != 0) {
            ppos = ppos < 0 ? 0 : (ppos >= a.n_pos ? a.n_pos - 1 : ppos);
            const unsigned po = (unsigned)(ppos * HALF + pi);
            cs = a.cos_t[po];
            sn = a.sin_t[po];
        }
        if (a.n_splits > 0) {
            const unsigned ro = (unsigned)pcand * (unsigned)ncols;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float v = 0.f;
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) v += sidx < a.n_splits ? pv[j][sidx] : 0.f;   // v is never -0: adding +0 is exact
                for (int sidx = 4; sidx < a.n_splits; ++sidx) v += a.partial[(size_t)sidx * pstride + ro + col[j]];
                if (a.bias) v += bs[j];
                x[j] = bfround(v);
            }
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DAT(5);
    // ---------------- tile iterator: every role describes its tile by per-lane K-row / V^T-row pointers and a mask ----------------
    // a run leader = the FIRST candidate of the tile that carries its prompt slot (compared against every earlier row, not only
    // the previous one: slots may recur non-contiguously, e.g. 0,1,0,1 -- with a previous-row test each recurrence would open
    // a second run of the same slot and its keys would enter the softmax twice; invalid rows carry slot -1 and never match)
    bool leader = q_ok;
#pragma unroll
    for (int j = 0; j < 15; ++j) {
        const int other = __shfl(my_slot1, (lane & 48) | j);
        leader = leader && !(j < r && other == my_slot1);
    }
    const unsigned leaders = (unsigned)(__ballot(leader && g == 0) & 0xffffull);
    const int role = w < a.WA ? 0 : (w < a.WA + WB ? 1 : 2);
    const int wl = role == 0 ? w : (role == 1 ? w - a.WA : w - a.WA - WB);
    const int cl = 4 * wl;                            // pool C: candidates cl..cl+3 of the block

    // a tile = uniform K / V^T base pointers + 32-bit lane offsets (elements; < 2^31 checked by the launcher)
    const bf16_t* kbase = nullptr;
    const bf16_t* vbase = nullptr;
    unsigned ko0 = 0, ko1 = 0;                        // K rows of the two 16-key halves (this lane's row, + g*8)
    unsigned vo = 0;                                  // V^T row d = r (+16*db via vstep), this lane's 8 keys
    unsigned vstep = 0;
    unsigned vmask = 0;                               // bit e: this lane's e-th score is visible
    int it = -1, bcol = -1, bslot = 0, blen = 0;      // iterator state: tile index inside the current run; pool B run
    // (only in the value-split variant: the unsplit one is at its 128-register budget, and its blocks are not the few-units case)
    bool tail_pending = (VS >= 2 && role == 2 && wl == 0 && a.tail_tile >= 0), is_tail = false;
    auto next_tile = [&]() -> bool {
        if (role == 0) {            // segment 0: tiles wl, wl + WA, ...
            it = it < 0 ? wl : it + a.WA;
            const int t0 = 32 * it;
            if (t0 >= a.len0 || it == a.tail_tile) return false;   // (the tail tile belongs to the first pool-C wave)
            int key0 = t0 + 8 * (r >> 2) + (r & 3), key1 = key0 + 4;
            key0 = key0 < a.len0 ? key0 : a.len0 - 1;
            key1 = key1 < a.len0 ? key1 : a.len0 - 1;
            kbase = a.k0 + h * a.k0_h;
            ko0 = key0 * a.k0_t + g * 8;
            ko1 = key1 * a.k0_t + g * 8;
            vbase = a.vt0 + h * a.vt0_h + t0;
            vo = r * a.vt0_d + g * 8;
            vstep = 16 * a.vt0_d;
            vmask = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) vmask |= (q_ok && t0 + 8 * g + e < a.len0) ? (1u << e) : 0u;
            return true;
        }
        if (role == 1) {            // segment 1: the runs (distinct prompt slots) p = wl, wl + WB, ... of the block, all their tiles
            ++it;
            if (bcol < 0 || 32 * it >= blen) {
                it = 0;
                bool found = false;
                for (int c = bcol + 1; c < 16 && !found; ++c) {
                    if (!((leaders >> c) & 1u)) continue;
                    if (__popc(leaders & ((1u << c) - 1u)) % WB != wl) continue;
                    const int glen = __shfl(my_len1, c);
                    if (glen <= 0) continue;
                    bcol = c; bslot = __builtin_amdgcn_readfirstlane(__shfl(my_slot1, c)); blen = __builtin_amdgcn_readfirstlane(glen);
                    found = true;
                }
                if (!found) return false;
            }
            const int t0 = 32 * it;
            int key0 = t0 + 8 * (r >> 2) + (r & 3), key1 = key0 + 4;
            key0 = key0 < blen ? key0 : blen - 1;
            key1 = key1 < blen ? key1 : blen - 1;
            kbase = a.k1 + (long long)bslot * a.k1_slot + h * a.k1_h;
            ko0 = key0 * a.k1_t + g * 8;
            ko1 = key1 * a.k1_t + g * 8;
            vbase = a.vt1 + (long long)bslot * a.vt1_slot + h * a.vt1_h + t0;
            vo = r * a.vt1_d + g * 8;
            vstep = 16 * a.vt1_d;
            vmask = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) vmask |= (q_ok && my_slot1 == bslot && t0 + 8 * g + e < blen) ? (1u << e) : 0u;
            return true;
        }
        // segment 2: candidates cl..cl+3, token chunk [tb, tb + 8): key slot 8g + e <-> candidate cl + g, token tb + e;
        // K operand row i of half X <-> key slot 8*(i>>2) + 4X + (i&3) = candidate cl + (i>>2), token tb + 4X + (i&3)
        ++it;
        const int tb = 8 * it;
        if (tb >= a.len2) {
            if (VS < 2 || !tail_pending) return false;
            // the segment-0 tail tile (pool A has exactly one wave per FULL tile): described like a pool-A tile
            tail_pending = false;
            is_tail = true;
            const int t0 = 32 * a.tail_tile;
            int key0 = t0 + 8 * (r >> 2) + (r & 3), key1 = key0 + 4;
            key0 = key0 < a.len0 ? key0 : a.len0 - 1;
            key1 = key1 < a.len0 ? key1 : a.len0 - 1;
            kbase = a.k0 + h * a.k0_h;
            ko0 = key0 * a.k0_t + g * 8;
            ko1 = key1 * a.k0_t + g * 8;
            vbase = a.vt0 + h * a.vt0_h + t0;
            vo = r * a.vt0_d + g * 8;
            vstep = 16 * a.vt0_d;
            vmask = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) vmask |= (q_ok && t0 + 8 * g + e < a.len0) ? (1u << e) : 0u;
            return true;
        }
        int cg = tile * 16 + cl + (r >> 2), cv = tile * 16 + cl + g;
        cg = cg < a.N ? cg : a.N - 1;
        cv = cv < a.N ? cv : a.N - 1;
        const int sk = a.slot2 ? a.slot2[cg] : cg, sv = a.slot2 ? a.slot2[cv] : cv;
        int ta = tb + (r & 3), tc = ta + 4;
        ta = ta < a.len2 ? ta : a.len2 - 1;
        tc = tc < a.len2 ? tc : a.len2 - 1;
        kbase = a.k2 + h * a.k2_h;
        ko0 = (unsigned)sk * (unsigned)a.k2_slot + ta * a.k2_t + g * 8;
        ko1 = (unsigned)sk * (unsigned)a.k2_slot + tc * a.k2_t + g * 8;
        vbase = a.vt2 + h * a.vt2_h + tb;
        vo = (unsigned)sv * (unsigned)a.vt2_slot + r * a.vt2_d;
        vstep = 16 * a.vt2_d;
        vmask = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) vmask |= (q_ok && cl + g == r && tb + e < a.len2) ? (1u << e) : 0u;
        return true;
    };
    u32x4 kr0[KS], kr1[KS], vr[DB];
    auto load_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kr0[ks] = *(const u32x4*)(kbase + ko0 + ks * 32);
            kr1[ks] = *(const u32x4*)(kbase + ko1 + ks * 32);
        }
#pragma unroll
        for (int db = 0; db < DB; ++db) vr[db] = *(const u32x4*)(vbase + vo + (db + (dv0 >> 4)) * vstep);
    };
    // ---------------- phase 1b: RoPE, q / k_new / v_new -> LDS, k_new / v_new -> own cache segment ----------------
    if (p1) {
        if (p1_ok) {
            if (a.rope_mode == 2) {
                cs = bfround(cs); sn = bfround(sn);
                const float a1 = bfround(bfround(x[0] * cs) + bfround(-x[1] * sn)), a2 = bfround(bfround(x[1] * cs) + bfround(x[0] * sn));
                const float b1 = bfround(bfround(x[2] * cs) + bfround(-x[3] * sn)), b2 = bfround(bfround(x[3] * cs) + bfround(x[2] * sn));
                x[0] = a1; x[1] = a2; x[2] = b1; x[3] = b2;
            } else if (a.rope_mode != 0) {
#pragma clang fp contract(off)   // same uncontracted arithmetic as rope_kv_write (torch rounds each product)
                const float a1 = x[0] * cs - x[1] * sn, a2 = x[1] * cs + x[0] * sn;
                const float b1 = x[2] * cs - x[3] * sn, b2 = x[3] * cs + x[2] * sn;
                x[0] = a1; x[1] = a2; x[2] = b1; x[3] = b2;
            }
        }
        qs[pc * D + pi] = f2bf(x[0]);
        qs[pc * D + pi + HALF] = f2bf(x[1]);
        kn[pc * D + pi] = f2bf(x[2]);
        kn[pc * D + pi + HALF] = f2bf(x[3]);
        vn[pc * D + pi] = f2bf(x[4]);
        vn[pc * D + pi + HALF] = f2bf(x[5]);
    }
    // the first tile's K and V go in flight only now and land behind the barrier; issuing them before phase 1 measured
    // slower (22.9 -> 24.4 / 26.9 us per layer at N=32, H=32: the tile phase is bound by per-CU load throughput)
    DAT(6);
    bool have = next_tile();
    if (have) load_tile();
    DAW(0, wall_clock64());
    DAW(4, (unsigned long long)role);
    // LDS-only barrier: q / k_new / v_new are exchanged through LDS, so neither the cache stores nor the tile loads
    // still in flight are waited for here
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    DAT(1);
    DAW(1, wall_clock64());
    // k_new / v_new -> the candidate's own cache segment, BEHIND the barrier: the V^T stores touch one 128-byte line per lane (d-major rows),
    // a block issues ~2 000 of them, and in front of the barrier their ISSUE time (per-block timelines: the last wave reached the barrier
    // 2 us after the first) was on every wave's critical path; here it runs under the flight time of the first tiles' loads. Nobody reads
    // these rows in this launch (pool C takes the new token from LDS).
    if (p1_ok) {
        bf16_t* kd = a.k2 + a.write_t * a.k2_t + h * a.k2_h;
        const unsigned kof = (unsigned)pslot2 * (unsigned)a.k2_slot + pi;
        if (blockIdx.z == 0) { kd[kof] = f2bf(x[2]); kd[kof + HALF] = f2bf(x[3]); }
        bf16_t* vd = a.vt2 + h * a.vt2_h + a.write_t;
        const unsigned vof = (unsigned)pslot2 * (unsigned)a.vt2_slot + pi * a.vt2_d;
        if (pi >= dv0 && pi < dv0 + DV) vd[vof] = f2bf(x[4]);
        if (pi + HALF >= dv0 && pi + HALF < dv0 + DV) vd[vof + HALF * a.vt2_d] = f2bf(x[5]);
    }

    // ---------------- phase 2 ----------------
    // Every tile is computed as an independent softmax state (m, l, O) and parked in the wave's LDS slot; a further tile
    // of the same wave (rare: more tiles than waves in the pool) is folded into the slot. No O accumulator lives across
    // tiles, which keeps the kernel inside the 128-register budget of a 16-wave block with K and V of a tile in flight.
    bool first = true;
#ifdef COVER_DA_DEBUG
    int dbg_tiles = 0;
#endif
    int sw = w;                                        // state slot this wave is writing
#pragma clang loop unroll(disable)
    while (have) {
        if (VS >= 2 && is_tail && sw == w) {   // switching to the extra slot: close this wave's own slot first
            if (first) {
#pragma unroll
                for (int i = 0; i < DB * 4; ++i) so[w * OW + i * 64 + lane] = 0.f;
                if (g == 0) {
                    sm[w * 16 + r] = -INFINITY;
                    sl[w * 16 + r] = 0.f;
                }
            }
            sw = DA_NW;
            first = true;
        }
        if (role == 2 && !is_tail) {   // the token written by this pass comes from LDS (its cache stores may still be in flight)
            const int e = a.write_t - 8 * it;
            if (e >= 0 && e < 8) {
                const bf16_t* kp = kn + (cl + (r >> 2)) * D + g * 8;
                const bool mine = (r & 3) == (e & 3);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 nk = *(const u32x4*)(kp + ks * 32);
                    kr0[ks] = (mine && e < 4) ? nk : kr0[ks];
                    kr1[ks] = (mine && e >= 4) ? nk : kr1[ks];
                    asm volatile("" ::: "memory");
                }
                const unsigned keep = (e & 1) ? 0x0000ffffu : 0xffff0000u;
                const int sh = (e & 1) ? 16 : 0, wi = e >> 1;
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const unsigned nv = (unsigned)vn[(cl + g) * D + dv0 + db * 16 + r] << sh;
#pragma unroll
                    for (int q = 0; q < 4; ++q) vr[db][q] = (q == wi) ? ((vr[db][q] & keep) | nv) : vr[db][q];
                    asm volatile("" ::: "memory");
                }
            }
        }
        // S^T (two 16-key halves), masked
        f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 qf = as_bf16x8(*(const uint4*)(qs + r * D + ks * 32 + g * 8));
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kr0[ks]), qf, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kr1[ks]), qf, s1, 0, 0, 0);
        }
        float sc[8], m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = ((vmask >> e) & 1u) ? (e < 4 ? s0[e] : s1[e - 4]) * a.scale_log2e : -INFINITY;
            m = fmaxf(m, sc[e]);
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f, p[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            p[e] = (m == -INFINITY) ? 0.f : exp2f(sc[e] - m);
            l += p[e];
        }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        uint4 pp;
        pp.x = pack_bf2(p[0], p[1]); pp.y = pack_bf2(p[2], p[3]); pp.z = pack_bf2(p[4], p[5]); pp.w = pack_bf2(p[6], p[7]);
        const bf16x8 pf = as_bf16x8(pp);
        float* slot = so + sw * OW + lane;
        float fo = 0.f, ft = 1.f;
        if (!first) {
            const float mo = sm[sw * 16 + r], lo = sl[sw * 16 + r];
            const float mx = fmaxf(mo, m);
            fo = (mo == -INFINITY) ? 0.f : exp2f(mo - mx);
            ft = (m == -INFINITY) ? 0.f : exp2f(m - mx);
            m = mx;
            l = lo * fo + l * ft;
        }
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vr[db]), pf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* sp = slot + (db * 4 + e) * 64;
                *sp = first ? o[e] : *sp * fo + o[e] * ft;
            }
            asm volatile("" ::: "memory");   // keep the LDS reads of the fold next to their use (register budget)
        }
        if (g == 0) {
            sm[sw * 16 + r] = m;
            sl[sw * 16 + r] = l;
        }
#ifdef COVER_DA_DEBUG
        if (first) DAW(2, wall_clock64());
        dbg_tiles++;
#endif
        first = false;
        have = next_tile();
        if (have) load_tile();
    }
#ifdef COVER_DA_DEBUG
    DAW(3, wall_clock64());
    DAW(5, (unsigned long long)dbg_tiles);
#endif
    if (first && sw == w) {   // a wave without a tile contributes the empty state
#pragma unroll
        for (int i = 0; i < DB * 4; ++i) so[w * OW + i * 64 + lane] = 0.f;
        if (g == 0) {
            sm[w * 16 + r] = -INFINITY;
            sl[w * 16 + r] = 0.f;
        }
    }
    if (role == 2 && wl == 0 && sw == w) {   // no tail tile: the extra slot is empty
#pragma unroll
        for (int i = 0; i < DB * 4; ++i) so[DA_NW * OW + i * 64 + lane] = 0.f;
        if (g == 0) {
            sm[DA_NW * 16 + r] = -INFINITY;
            sl[DA_NW * 16 + r] = 0.f;
        }
    }

    // ---------------- phase 3: merge the waves' states in wave order ----------------
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only again: the cache stores drain behind the merge
    DAT(2);
    if (!q_ok) return;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < DA_NW + 1; ++i) mx = fmaxf(mx, sm[i * 16 + r]);
    float lt = 0.f, f[DA_NW + 1];
#pragma unroll
    for (int i = 0; i < DA_NW + 1; ++i) {
        const float mi = sm[i * 16 + r];
        f[i] = (mi == -INFINITY) ? 0.f : exp2f(mi - mx);
        lt += sl[i * 16 + r] * f[i];
    }
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    const int item0 = w * IPW, db = item0 >> 2, e0 = item0 & 3;
    if (item0 >= DB * 4) return;                       // fewer merge items than waves (small D / VS = 2)
    float o[IPW];
#pragma unroll
    for (int e = 0; e < IPW; ++e) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < DA_NW + 1; ++i) acc += so[i * OW + (item0 + e) * 64 + lane] * f[i];
        o[e] = acc * inv;
    }
    bf16_t* op = a.out + (long long)cand_r * a.o_row + (long long)h * D + dv0 + db * 16 + 4 * g + e0;
    if constexpr (IPW == 2) {
        *(uint32_t*)op = pack_bf2(o[0], o[1]);
    } else {
        *op = f2bf(o[0]);
    }
    DAT(3);
}

hipError_t launch_decode_attention_fused(const cover_decode_attn_args* x, hipStream_t st) {
    if (x->N <= 0) return hipSuccess;
    if (x->D != 64 && x->D != 128) return hipErrorInvalidValue;
    for (int i = 0; i < 3; ++i)
        if (x->seg[i].mask_mode != COVER_MASK_LEN) return hipErrorInvalidValue;
    if (x->seg[0].len <= 0 || x->seg[2].len <= 0 || x->write_t < 0 || x->write_t >= x->seg[2].len) return hipErrorInvalidValue;
    if (x->rope_mode != 0 && (!x->cos_table || !x->sin_table || !x->positions)) return hipErrorInvalidValue;
    DecAttnDev a;
    a.qkv = (const bf16_t*)x->qkv; a.ld_qkv = x->ld_qkv;
    a.partial = x->partial; a.n_splits = x->n_splits; a.bias = x->bias;
    a.N = x->N; a.H = x->H;
    a.scale_log2e = x->scale * 1.4426950408889634f;
    a.positions = x->positions; a.cos_t = x->cos_table; a.sin_t = x->sin_table; a.n_pos = x->n_pos; a.rope_mode = x->rope_mode;
    const cover_kv_segment &s0 = x->seg[0], &s1 = x->seg[1], &s2 = x->seg[2];
    const long long slot0 = 0;  // shared segment: one slot, given by the base pointer
    a.k0 = (const bf16_t*)s0.k + slot0 * s0.k_slot_stride; a.vt0 = (const bf16_t*)s0.vt + slot0 * s0.vt_slot_stride;
    const long long strides[] = {s0.k_t_stride, s0.k_h_stride, s0.vt_h_stride, s0.vt_d_stride, s1.k_t_stride, s1.k_h_stride, s1.vt_h_stride,
                                 s1.vt_d_stride, s2.k_t_stride, s2.k_h_stride, s2.vt_h_stride, s2.vt_d_stride};
    for (long long v : strides)
        if (v < 0 || v > (1ll << 26)) return hipErrorInvalidValue;   // in-slot offsets are formed in 32-bit arithmetic
    if ((long long)x->N * 3 * x->H * x->D * (x->n_splits > 0 ? x->n_splits : 1) >= (1ll << 31)) return hipErrorInvalidValue;
    if (!s2.slot_of_batch && ((long long)x->N * s2.k_slot_stride >= (1ll << 31) || (long long)x->N * s2.vt_slot_stride >= (1ll << 31)))
        return hipErrorInvalidValue;   // with an explicit slot table the caller keeps slot * slot_stride below 2^31 elements
    if ((s2.vt_d_stride & 7) || (s0.vt_d_stride & 7) || (s1.vt_d_stride & 7)) return hipErrorInvalidValue;   // 16-byte V^T row reads
    a.k0_t = (int)s0.k_t_stride; a.k0_h = (int)s0.k_h_stride; a.vt0_h = (int)s0.vt_h_stride; a.vt0_d = (int)s0.vt_d_stride; a.len0 = s0.len;
    a.k1 = (const bf16_t*)s1.k; a.vt1 = (const bf16_t*)s1.vt;
    a.k1_slot = s1.k_slot_stride; a.k1_t = (int)s1.k_t_stride; a.k1_h = (int)s1.k_h_stride;
    a.vt1_slot = s1.vt_slot_stride; a.vt1_h = (int)s1.vt_h_stride; a.vt1_d = (int)s1.vt_d_stride;
    a.slot1 = s1.slot_of_batch; a.len1 = s1.len_of_batch; a.len1_c = s1.len;
    a.k2 = (bf16_t*)s2.k; a.vt2 = (bf16_t*)s2.vt;
    a.k2_slot = s2.k_slot_stride; a.k2_t = (int)s2.k_t_stride; a.k2_h = (int)s2.k_h_stride;
    a.vt2_slot = s2.vt_slot_stride; a.vt2_h = (int)s2.vt_h_stride; a.vt2_d = (int)s2.vt_d_stride;
    a.slot2 = s2.slot_of_batch; a.len2 = s2.len; a.write_t = x->write_t;
    a.out = (bf16_t*)x->out; a.o_row = x->out_row_stride;
    // wave pools (see the header comment): 4 waves for segment 2 (4 candidates each), one wave per segment-0 tile up to 9,
    // the rest (>= 3) for the distinct prompts of a block
    const int nA = (s0.len + 31) / 32;
    a.WC = 4;
    a.WA = nA < 9 ? nA : 9;
    a.tail_tile = -1;
    // Nine segment-0 tiles (257 keys = BOS + 256 patches: eight full tiles and ONE key) would cost pool A a ninth wave and leave
    // pool B three waves for the four prompts of a block -- one of them then runs two tiles back to back, each a full
    // global round trip, and the whole block waits for it. The ninth tile goes to the first pool-C wave instead (its own
    // tile is 8 short keys), into a state slot of its own so that the merge order stays independent of where a candidate sits.
    // few (candidate tile, head) units: split the value columns over two blocks each (see the kernel's VS comment)
    static const char* vs_env = getenv("COVER_DA_VSPLIT");
    const int units = ((x->N + 15) / 16) * x->H;
    // units <= 64 (N = 32 at 32 heads): four blocks per unit put one block on every CU; the score part is computed four times
    // over, but a block then pulls 10 instead of 12 KiB per wave-tile and merges a quarter of the columns (decode pass 3.655 ->
    // 3.58 ms). D = 64 heads have too few columns to split in four.
    int VS = vs_env ? (atoi(vs_env) == 4 ? 4 : atoi(vs_env) == 2 ? 2 : 1) : (units <= 64 ? 4 : units <= 128 ? 2 : 1);
    if (x->D != 128 && VS == 4) VS = 2;
    static const char* tail_env = getenv("COVER_DA_TAIL");   // experiment knob: 0 keeps nine pool-A waves
    if (nA == 9 && VS >= 2 && !(tail_env && tail_env[0] == '0')) { a.WA = 8; a.tail_tile = 8; }
    dim3 grid((x->N + 15) / 16, x->H, VS), block(64 * DA_NW);
    const int pid = prof_enabled() ? prof_open(st, 2, 0.0) : -1;
    if (x->D == 128) {
        if (VS == 4) hipLaunchKernelGGL((decode_attn_fused_k<128, 4>), grid, block, 0, st, a);
        else if (VS == 2) hipLaunchKernelGGL((decode_attn_fused_k<128, 2>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((decode_attn_fused_k<128, 1>), grid, block, 0, st, a);
    } else {
        if (VS == 2) hipLaunchKernelGGL((decode_attn_fused_k<64, 2>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((decode_attn_fused_k<64, 1>), grid, block, 0, st, a);
    }
    prof_close(st, pid);
    return hipGetLastError();
}
