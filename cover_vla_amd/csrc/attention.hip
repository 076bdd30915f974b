// Flash-style attention for gfx950, one wave per 16 query rows, no LDS in the main loop.
//
// Both products run on v_mfma_f32_16x16x32_bf16 with SWAPPED operands so that the query index is lane&15 in
// every accumulator (softmax row state is per lane, replicated over the 4 lane groups):
//   S^T[key][q] = K-fragment (rows = keys, contraction over d)  x  Q-fragment
//   O^T[d][q]   = V^T-fragment (rows = d, contraction over keys) x  P-fragment
// The key order inside a 32-key tile is permuted (key_of(X,i) = t0 + 8*(i>>2) + 4*X + (i&3)) so that the eight
// probabilities a lane holds after QK^T are exactly the eight contraction elements the PV MFMA expects from
// that lane: P never leaves registers, no shuffles, no LDS. V is read from a TRANSPOSED cache [d][t] so its
// operand is one 16-byte load per lane as well.
//
// Masks come from lengths (COVER_MASK_LEN / CAUSAL / VISLEN), never from a materialised [T,T] tensor
// (make_att_2d_masks, modeling_pi0.py:98-128). Masked keys get probability exactly 0, as in the reference
// where exp(-2.38e38 - max) underflows to 0 (paligemma_with_expert.py:418-423).
//
// Query rows of a tile enumerate (token, q-head-within-kv-group) pairs so that GQA/MQA heads sharing one
// kv head share K/V loads (pi0: 8 q heads x 1 kv head; decode: 5 suffix tokens x 8 heads = 40 rows = 3 waves).
// KSPLIT mode (few query rows, e.g. single-token decode): the 4 waves of a block split the key tiles and merge
// their (m, l, O) partial states through LDS.
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "kernels.h"

struct SegDev {
    const bf16_t* k;
    const bf16_t* vt;
    long long k_slot, k_t, k_h, vt_slot, vt_h, vt_d;
    const int* slot_of_batch;
    const int* len_of_batch;
    const int* vis_len;
    int len, mode, causal_off;
};
struct AttnDev {
    const bf16_t* q;
    bf16_t* out;
    long long q_b, q_t, q_h, o_b, o_t, o_h;
    int B, Tq, Hq, Hkv, G, R;
    float scale_log2e;
    int n_seg;
    SegDev seg[3];
    const float* si_o; const float* si_ml;
    float* so_o; float* so_ml;
    uint8_t* o8; uint8_t* o8mx; int o8_rows;   // MX block-scaled e4m3 output INSTEAD of `out` (cover_attn_args.out8): same strides, in bytes
    bool shared;                               // host side: launch the workgroup-shared-keys form (attn_shared_k)
};

#ifdef COVER_AT_DEBUG
__device__ unsigned long long g_at_dbg[512 * 8];   // per block (thread 0): start, Q loaded, first tile's K landed, tiles done, merged, end (100 MHz)
extern "C" int cover_at_debug(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_at_dbg), sizeof(g_at_dbg)); }
#define ATT(slot) do { if (threadIdx.x == 0) g_at_dbg[(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) & 511) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define ATT(slot) do { } while (0)
#endif
// NWS: waves per block in key-split mode (4 or 8), a compile-time constant there so that the merge below is straight-line code
// MXO: the output rows are written as e4m3 with one E8M0 scale per 32 columns (cover_quantize_act_fp8_mx's arithmetic on the bf16 values that would have
// been stored): key-split mode with two ADJACENT d blocks per wave (DB == 2 NWS), so that a wave holds whole 32-column blocks after the merge
template <int D, bool KSPLIT, int NWS = 4, bool MXO = false>
__device__ __forceinline__ void attn_body(const AttnDev& a, int bx, int kvh, int b) {
    ATT(0);
    constexpr int KS = D / 32;  // k-steps of QK^T
    constexpr int DB = D / 16;  // 16-row d blocks of O^T
    static_assert(!MXO || (KSPLIT && DB == 2 * NWS), "block-scaled output: two adjacent d blocks per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = KSPLIT ? NWS : (int)(blockDim.x >> 6);
    const int tile = KSPLIT ? bx : bx * nw + w;
    const int r = lane & 15, g = lane >> 4;

    const int qi = tile * 16 + r;
    const bool q_ok = qi < a.R;
    const int qc = q_ok ? qi : 0;
    const int t = qc / a.G, gh = qc - t * a.G;
    const int h = kvh * a.G + gh;
    if (!KSPLIT && tile * 16 >= a.R) return;  // whole wave idle (no barriers in this mode)

    // Q fragments
    bf16x8 qf[KS];
    {
        const bf16_t* qp = a.q + (long long)b * a.q_b + (long long)t * a.q_t + (long long)h * a.q_h + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = as_bf16x8(*(const uint4*)(qp + ks * 32));
    }

#ifdef COVER_AT_DEBUG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT(1);
#endif
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) oacc[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long srow = ((long long)b * a.Tq + t) * a.Hq + h;  // state row of this lane's query
    if (a.si_o && q_ok && (!KSPLIT || w == 0)) {  // resume from a previous call's state (one wave owns it)
        m_run = a.si_ml[srow * 2];
        l_run = a.si_ml[srow * 2 + 1];
        const float* so = a.si_o + srow * D + 4 * g;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const float4 v = *(const float4*)(so + db * 16);
            oacc[db] = (f32x4){v.x * l_run, v.y * l_run, v.z * l_run, v.w * l_run};
        }
    }

    // largest token index in this tile (wave-uniform) for causal early exit
    int t_hi = (tile * 16 + 15) / a.G;
    t_hi = t_hi < a.Tq ? t_hi : a.Tq - 1;

    int tile_counter = 0;
    for (int si = 0; si < a.n_seg; ++si) {
        const SegDev& sg = a.seg[si];
        const int slot = sg.slot_of_batch ? sg.slot_of_batch[b] : b;
        const int len = sg.len_of_batch ? sg.len_of_batch[b] : sg.len;
        int kend = len;
        if (sg.mode == COVER_MASK_CAUSAL) kend = min(kend, t_hi + sg.causal_off + 1);
        int vis = len;  // per-lane visible key bound (exclusive)
        if (sg.mode == COVER_MASK_CAUSAL) vis = min(len, t + sg.causal_off + 1);
        else if (sg.mode == COVER_MASK_VISLEN) vis = min(len, sg.vis_len[t]);
        const bf16_t* kb = sg.k + (long long)slot * sg.k_slot + (long long)kvh * sg.k_h + g * 8;
        const bf16_t* vb = sg.vt + (long long)slot * sg.vt_slot + (long long)kvh * sg.vt_h + (long long)r * sg.vt_d + g * 8;

        for (int t0 = 0; t0 < kend; t0 += 32, ++tile_counter) {
            if (KSPLIT && (tile_counter & (nw - 1)) != w) continue;
            // ---- issue every load of this tile up front (K fragments, and V^T fragments when they fit in registers):
            // single-token decode is a chain of dependent global round trips, so K and V must travel together ----
            int key0 = t0 + 8 * (r >> 2) + (r & 3);
            int key1 = key0 + 4;
            key0 = key0 < len ? key0 : len - 1;
            key1 = key1 < len ? key1 : len - 1;
            const bf16_t* k0p = kb + (long long)key0 * sg.k_t;
            const bf16_t* k1p = kb + (long long)key1 * sg.k_t;
            const bf16_t* vp = vb + t0;
            uint4 kr0[KS], kr1[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kr0[ks] = *(const uint4*)(k0p + ks * 32);
                kr1[ks] = *(const uint4*)(k1p + ks * 32);
            }
            // D = 256: 64 registers of V^T fragments beside 64 of K, 64 of O and 32 of Q only fit when the block is the 4-wave key-split
            // one (launch bounds 256 below: one wave per SIMD). Without them a tile is TWO dependent round trips (K, softmax, then V):
            // six per wave over the ~330 keys of a pi0 denoise step instead of three.
            constexpr bool V_EARLY = (D <= 128) || KSPLIT;
            uint4 vr[V_EARLY ? DB : 1];
            if (V_EARLY) {
#pragma unroll
                for (int db = 0; db < DB; ++db) vr[db] = *(const uint4*)(vp + (long long)(db * 16) * sg.vt_d);
            }
#ifdef COVER_AT_DEBUG
            if (tile_counter == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ATT(2); }
#endif
            // ---- S^T = K . Q^T for two 16-key blocks ----
            f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(kr0[ks]), qf[ks], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(kr1[ks]), qf[ks], s1, 0, 0, 0);
            }
            // lane (q = r, g) holds keys t0 + 8g + e, e = 0..7 (s0 -> e 0..3, s1 -> e 4..7)
            float sc[8];
            float tmax = -INFINITY;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (e < 4 ? s0[e] : s1[e - 4]) * a.scale_log2e;
                const int key = t0 + 8 * g + e;
                sc[e] = (key < vis) ? v : -INFINITY;
                tmax = fmaxf(tmax, sc[e]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = fmaxf(m_run, tmax);
            float alpha = 1.f, psum = 0.f;
            float p[8];
            if (m_new == -INFINITY) {
#pragma unroll
                for (int e = 0; e < 8; ++e) p[e] = 0.f;
            } else {
                alpha = exp2f(m_run - m_new);  // m_run = -inf -> 0
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    p[e] = exp2f(sc[e] - m_new);  // masked -> exp2(-inf) = 0
                    psum += p[e];
                }
            }
            psum += __shfl_xor(psum, 16);
            psum += __shfl_xor(psum, 32);
            l_run = l_run * alpha + psum;
            m_run = m_new;
            uint4 pp;
            pp.x = pack_bf2(p[0], p[1]);
            pp.y = pack_bf2(p[2], p[3]);
            pp.z = pack_bf2(p[4], p[5]);
            pp.w = pack_bf2(p[6], p[7]);
            const bf16x8 pf = as_bf16x8(pp);
            // ---- O^T += V^T . P ----
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const bf16x8 vf = as_bf16x8(V_EARLY ? vr[V_EARLY ? db : 0] : *(const uint4*)(vp + (long long)(db * 16) * sg.vt_d));
                f32x4 o = oacc[db];
                o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
                oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o, 0, 0, 0);
            }
        }
    }

    ATT(3);
    // In key-split mode every wave merges and stores ITS share of the d blocks (db = w, w + NWS, ...). The merge used to run on
    // wave 0 alone behind `i < nw` tests on a run-time nw: the compiler kept it as a chain of dependent LDS reads, 11.8 us of a
    // 23.5 us launch at D = 256 (3.8 of 7.7 at D = 64; per-block timelines, tools/dbg/at_timeline.py). Same sums in the same order.
    int db_first = 0, db_step = 1;
    if (KSPLIT) {
        // LDS layout per wave: [DB*4 floats per lane][64 lanes] + m[16] + l[16]
        float* so = (float*)smem;
        constexpr int OW = DB * 4 * 64;
        float* sm = so + NWS * OW;
        float* sl = sm + NWS * 16;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int e = 0; e < 4; ++e) so[w * OW + (db * 4 + e) * 64 + lane] = oacc[db][e];
        if (g == 0) {
            sm[w * 16 + r] = m_run;
            sl[w * 16 + r] = l_run;
        }
        __syncthreads();
        float mi[NWS], li[NWS], f[NWS];
#pragma unroll
        for (int i = 0; i < NWS; ++i) {
            mi[i] = sm[i * 16 + r];
            li[i] = sl[i * 16 + r];
        }
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NWS; ++i) mx = fmaxf(mx, mi[i]);
        float lt = 0.f;
#pragma unroll
        for (int i = 0; i < NWS; ++i) {
            f[i] = (mi[i] == -INFINITY) ? 0.f : exp2f(mi[i] - mx);
            lt += li[i] * f[i];
        }
#pragma unroll
        for (int dbi = 0; dbi < (DB + NWS - 1) / NWS; ++dbi) {
            const int db = MXO ? 2 * w + dbi : w + dbi * NWS;
            if (db < DB) {
                float x[NWS][4];
#pragma unroll
                for (int i = 0; i < NWS; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[i][e] = so[i * OW + (db * 4 + e) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < NWS; ++i) acc += x[i][e] * f[i];
                    oacc[dbi][e] = acc;              // this wave's dbi-th block (db = w + dbi * NWS; MXO: 2 w + dbi) now lives in slot dbi
                }
            }
        }
        l_run = lt;
        m_run = mx;
        db_first = MXO ? 2 * w : w;
        db_step = MXO ? 1 : NWS;
    }
    ATT(4);

    if (!q_ok) return;
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    if (a.so_o) {  // hand the state to the next call instead of writing the final output
        if (g == 0 && (!KSPLIT || w == 0)) {
            a.so_ml[srow * 2] = m_run;
            a.so_ml[srow * 2 + 1] = l_run;
        }
        float* so = a.so_o + srow * D + 4 * g;
#pragma unroll
        for (int k = 0; k < DB; ++k) {
            const int db = db_first + k * db_step;
            if (db < DB) *(float4*)(so + db * 16) = make_float4(oacc[k][0] * inv, oacc[k][1] * inv, oacc[k][2] * inv, oacc[k][3] * inv);
        }
        return;
    }
    if constexpr (MXO) {
        // slots 0, 1 = d blocks 2w, 2w + 1 = columns 32 w .. 32 w + 31 of head h: lane (r, g) holds 2 x 4 of the block's values of query row r, the other 24
        // sit in the lanes r + 16 g'. (MHA only: a query row is a token.)
        float v[8];
        float mx = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[k * 4 + e] = bf2f(f2bf(oacc[k][e] * inv));
                mx = fmaxf(mx, fabsf(v[k * 4 + e]));
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sc = 1.0f;
        if (mx > 0.f) {   // smallest power of two >= 2^-126 with mx / sc <= 448 (e4m3_pow2_scale of gemm_common.h)
            int ex;
            const float fr = frexpf(mx / 448.0f, &ex);
            sc = ldexpf(1.0f, fr == 0.5f ? ex - 1 : ex);
        }
        sc = fmaxf(sc, 1.1754943508222875e-38f);
        const float is = 1.0f / sc;
        const long long ro = (long long)b * a.o_b + (long long)t * a.o_t;
        const int col = h * (int)a.o_h + 32 * w;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[k * 4] * is, v[k * 4 + 1] * is, 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[k * 4 + 2] * is, v[k * 4 + 3] * is, pk, true);
            *(uint32_t*)(a.o8 + ro + col + k * 16 + 4 * g) = (uint32_t)pk;
        }
        if (g == 0) a.o8mx[((size_t)(col >> 7) * a.o8_rows + (size_t)(ro / a.o_t)) * 4 + ((col >> 5) & 3)] = (uint8_t)((__builtin_bit_cast(uint32_t, sc) >> 23) & 0xffu);
        return;
    }
    bf16_t* op = a.out + (long long)b * a.o_b + (long long)t * a.o_t + (long long)h * a.o_h + 4 * g;
#pragma unroll
    for (int k = 0; k < DB; ++k) {
        const int db = db_first + k * db_step;
        if (db < DB) {
            uint2 v;
            v.x = pack_bf2(oacc[k][0] * inv, oacc[k][1] * inv);
            v.y = pack_bf2(oacc[k][2] * inv, oacc[k][3] * inv);
            *(uint2*)(op + db * 16) = v;
        }
    }
    ATT(5);
}

// ---------------------------------------------------------------------------------------------------
// Many query rows over SHARED keys (round 6; the attention pass of the large-N candidate decode: 64 samples of a prompt attend the same
// [shared prefix | prompt text] keys). attn_body gives every 16-row query tile its own wave(s) and every wave its own copy of the K / V^T
// fragments from global memory: at 8 prompts x 64 samples x 32 heads that is 1 024 workgroups pulling 144 KB each through the L2 -- 147 MB per
// launch for 4.6 MB of distinct keys; the per-block timeline (tools/dbg/at_timeline.py, SHAPE=c5) shows 6 us between a block's Q and its first K tile
// and 23 us per launch. Here a workgroup = (64 query rows, head, batch entry): four waves with one 16-row tile each, the K / V^T tile of 32 keys staged
// ONCE per workgroup in LDS (LDS-DMA ring of four stages, three tiles ahead, one barrier per tile; swizzled rows: conflict-free 16-byte reads)
// -- a quarter of the L2 traffic, one workgroup per (prompt, head) = one round of 256 on the chip. Per lane the arithmetic is attn_body's, in the same
// order (non-key-split mode): the results are bit-identical to attn_kernel<128, false>. MHA, D = 128, length masks only. Config 5 on one box, alternating
// (profiles/r06_attn_shared_keys.txt): 411.5 -> 403.2 / 405.3 ms per decision (1 792 launches, ~4 us each); what is left of a launch is one wave per SIMD
// walking ten dependent tiles (LDS reads -> MFMA -> two cross-row reductions -> exp2 -> MFMA) with nobody to overlap.
// ---------------------------------------------------------------------------------------------------
template <bool MXO>
__global__ __launch_bounds__(256) void attn_shared_k(AttnDev a) {
    constexpr int D = 128, KS = D / 32, DB = D / 16;
    constexpr int KT_BYTES = 32 * 256, VT_BYTES = D * 64, ST_BYTES = KT_BYTES + VT_BYTES, NST = 4, DIST = 3;
    __shared__ __attribute__((aligned(16))) char smem[NST * ST_BYTES];   // 64 KiB
    ATT(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int t = blockIdx.x * 64 + w * 16 + r;
    const bool q_ok = t < a.Tq;
    const int tc = q_ok ? t : a.Tq - 1;

    bf16x8 qf[KS];
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 oacc[DB];
    const long long srow = ((long long)b * a.Tq + tc) * a.Hq + h;
    // segment table (workgroup-uniform; scalars and compare-selects: a run-time index into an array would put it in scratch)
    int len0 = 0, len1 = 0, len2 = 0;
    const bf16_t *kp0 = nullptr, *kp1 = nullptr, *kp2 = nullptr, *vp0 = nullptr, *vp1 = nullptr, *vp2 = nullptr;
    long long kt0 = 0, kt1 = 0, kt2 = 0, vd0 = 0, vd1 = 0, vd2 = 0;
#define COVER_SEG_SETUP(I, LEN, KP, VP, KT, VD)                                          \
    if (I < a.n_seg) {                                                                   \
        const int slot_ = a.seg[I].slot_of_batch ? a.seg[I].slot_of_batch[b] : b;        \
        LEN = a.seg[I].len_of_batch ? a.seg[I].len_of_batch[b] : a.seg[I].len;           \
        KP = a.seg[I].k + (long long)slot_ * a.seg[I].k_slot + (long long)h * a.seg[I].k_h;    \
        VP = a.seg[I].vt + (long long)slot_ * a.seg[I].vt_slot + (long long)h * a.seg[I].vt_h; \
        KT = a.seg[I].k_t;                                                               \
        VD = a.seg[I].vt_d;                                                              \
    }
    COVER_SEG_SETUP(0, len0, kp0, vp0, kt0, vd0)
    COVER_SEG_SETUP(1, len1, kp1, vp1, kt1, vd1)
    COVER_SEG_SETUP(2, len2, kp2, vp2, kt2, vd2)
#undef COVER_SEG_SETUP
    // The K / V^T tiles travel by LDS-DMA (global_load_lds_dwordx4: no staging registers, no LDS write instructions) into a ring of NST stages, DIST tiles
    // ahead of the arithmetic: a tile is 0.5 us of MFMA + softmax against ~2 us of load latency, so one tile of look-ahead leaves the chain of ~10 tiles
    // latency-bound (first version, register-staged double buffer: 30.9 us launch to launch against 34.5 of the per-tile kernel, decisions -1.7 ms; this
    // one 29.9 against 35.0, decisions -7.3 ms). The DMA writes a wave's
    // 64 x 16 bytes linearly, so the swizzles are applied on the GLOBAL side: LDS position (key row k, pos) holds the row's chunk pos ^ swz(k), position
    // (d row, pos) of the V^T tile holds chunk pos ^ ((d >> 2) & 3) -- conflict-free 16-byte reads at a 256-byte / 64-byte row pitch. Every thread issues
    // exactly four pieces per tile (past the end: the last tile again), so one counted vmcnt per tile is exact.
    const uint32_t lds_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(smem));
    auto issue = [&](const bf16_t* kp, const bf16_t* vp, long long kt, long long vd, int len, int t0, int stage) __attribute__((always_inline)) {
        const uint32_t base = lds_u32 + stage * ST_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = 8 * w + 4 * j + (lane >> 4);
            const int c = (lane & 15) ^ ((k & 3) | ((k >> 3) << 2));
            int key = t0 + k;
            key = key < len ? key : len - 1;
            glds16_asm(kp + (long long)key * kt + c * 8, base + (8 * w + 4 * j) * 256);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int d = (2 * w + j) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((d >> 2) & 3);
            glds16_asm(vp + (long long)d * vd + t0 + c * 8, base + KT_BYTES + (2 * w + j) * 1024);
        }
    };
    // issue cursor: the tile the next DMA batch fetches (workgroup-uniform; explicit per-segment code, see above)
    int iseg = len0 > 0 ? 0 : (len1 > 0 ? 1 : (len2 > 0 ? 2 : 3)), it0 = 0;
    const bf16_t *ikp = kp0, *ivp = vp0;
    long long ikt = kt0, ivd = vd0;
    int ilen = len0;
    if (iseg == 1) { ikp = kp1; ivp = vp1; ikt = kt1; ivd = vd1; ilen = len1; }
    if (iseg == 2) { ikp = kp2; ivp = vp2; ikt = kt2; ivd = vd2; ilen = len2; }
    int istage = 0;
    auto issue_next = [&]() __attribute__((always_inline)) {
        issue(ikp, ivp, ikt, ivd, ilen, it0, istage);   // (past the last tile the cursor stays on it: a harmless re-load into a stage nobody reads any more)
        istage = istage == NST - 1 ? 0 : istage + 1;
        if (iseg < 3 && it0 + 32 < ilen) {
            it0 += 32;
        } else if (iseg == 0 && len1 > 0) {
            iseg = 1; it0 = 0; ikp = kp1; ivp = vp1; ikt = kt1; ivd = vd1; ilen = len1;
        } else if (iseg <= 1 && len2 > 0) {
            iseg = 2; it0 = 0; ikp = kp2; ivp = vp2; ikt = kt2; ivd = vd2; ilen = len2;
        } else {
            iseg = 3;   // done: keep fetching the last tile
        }
    };
    int cstage = 0;
#ifdef COVER_AT_DEBUG   // where a tile's time goes (thread 0 of every workgroup; 100 MHz ticks summed over the tiles): wait | barrier | DMA issue | S^T | softmax | PV
    unsigned long long seg_t[6] = {0, 0, 0, 0, 0, 0}, tp = 0;
#define TSEG(i) do { const unsigned long long tn_ = wall_clock64(); seg_t[i] += tn_ - tp; tp = tn_; } while (0)
#else
#define TSEG(i) do { } while (0)
#endif
    auto tile = [&](int vis, int t0) __attribute__((always_inline)) {
#ifdef COVER_AT_DEBUG
        tp = wall_clock64();
#endif
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DIST - 1)) : "memory");   // this thread's pieces of the tile have landed ...
        TSEG(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // ... everybody's have, and everybody is done with the previous tile's stage
        TSEG(1);
        issue_next();                                                           // tile + DIST, into the stage of tile - 1
        TSEG(2);
        const char* ks_ = smem + cstage * ST_BYTES;
        const char* vs_ = ks_ + KT_BYTES;
        const int k0 = 8 * (r >> 2) + (r & 3), k1 = k0 + 4;
        f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int ksi = 0; ksi < KS; ++ksi) {
            const int pos = ((ksi * 4 + g) ^ r) << 4;   // swz(k0) = swz(k1) = r
            const uint4 kr0 = *(const uint4*)(ks_ + k0 * 256 + pos);
            const uint4 kr1 = *(const uint4*)(ks_ + k1 * 256 + pos);
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(kr0), qf[ksi], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(kr1), qf[ksi], s1, 0, 0, 0);
        }
#ifdef COVER_AT_DEBUG
        asm volatile("" : "+v"(s0), "+v"(s1));
        TSEG(3);
#endif
        float sc[8];
        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (e < 4 ? s0[e] : s1[e - 4]) * a.scale_log2e;
            const int key = t0 + 8 * g + e;
            sc[e] = (key < vis) ? v : -INFINITY;
            tmax = fmaxf(tmax, sc[e]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        float alpha = 1.f, psum = 0.f;
        float p[8];
        if (m_new == -INFINITY) {
#pragma unroll
            for (int e = 0; e < 8; ++e) p[e] = 0.f;
        } else {
            alpha = exp2f(m_run - m_new);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                p[e] = exp2f(sc[e] - m_new);
                psum += p[e];
            }
        }
        psum += __shfl_xor(psum, 16);
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        uint4 pp;
        pp.x = pack_bf2(p[0], p[1]);
        pp.y = pack_bf2(p[2], p[3]);
        pp.z = pack_bf2(p[4], p[5]);
        pp.w = pack_bf2(p[6], p[7]);
        const bf16x8 pf = as_bf16x8(pp);
#ifdef COVER_AT_DEBUG
        asm volatile("" : "+v"(pp.x), "+v"(pp.y), "+v"(pp.z), "+v"(pp.w));
        TSEG(4);
#endif
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const bf16x8 vf = as_bf16x8(*(const uint4*)(vs_ + (db * 16 + r) * 64 + ((g ^ (r >> 2)) << 4)));
            f32x4 o = oacc[db];
            o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
            oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o, 0, 0, 0);
        }
#ifdef COVER_AT_DEBUG
#pragma unroll
        for (int db = 0; db < DB; ++db) asm volatile("" : "+v"(oacc[db]));
        TSEG(5);
#endif
        cstage = cstage == NST - 1 ? 0 : cstage + 1;
    };
    if (iseg < 3) {
#pragma unroll
        for (int i = 0; i < DIST; ++i) issue_next();   // tiles 0 .. DIST - 1
    }
    ATT(1);   // (timeline builds: segment table read, first DMA batch issued)
    // Q fragments and the resumed state AFTER the first DMA batch: the compiler's wait for them then covers pieces that are needed at the first tile anyway
    {
        const bf16_t* qp = a.q + (long long)b * a.q_b + (long long)tc * a.q_t + (long long)h * a.q_h + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = as_bf16x8(*(const uint4*)(qp + ks * 32));
    }
#pragma unroll
    for (int db = 0; db < DB; ++db) oacc[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.si_o && q_ok) {   // resume from a previous call's state
        m_run = a.si_ml[srow * 2];
        l_run = a.si_ml[srow * 2 + 1];
        const float* so = a.si_o + srow * D + 4 * g;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const float4 v = *(const float4*)(so + db * 16);
            oacc[db] = (f32x4){v.x * l_run, v.y * l_run, v.z * l_run, v.w * l_run};
        }
    }
    // every ordinary load is CONSUMED here, once: the compiler's wait-count pass does not see the DMA pieces and would otherwise re-wait for "its four
    // youngest loads" (the Q fragments) with vmcnt(0) inside every tile -- draining the ring
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
#pragma unroll
    for (int db = 0; db < DB; ++db) asm volatile("" : "+v"(oacc[db]));
    asm volatile("" : "+v"(m_run), "+v"(l_run));
    ATT(2);   // (Q fragments and the resumed state landed)
    if (iseg < 3) {
        for (int t0 = 0; t0 < len0; t0 += 32) tile(len0, t0);
        for (int t0 = 0; t0 < len1; t0 += 32) tile(len1, t0);
        for (int t0 = 0; t0 < len2; t0 += 32) tile(len2, t0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the surplus re-loads of the tail)
    }
    ATT(3);   // (tiles done)
#ifdef COVER_AT_DEBUG
    if (threadIdx.x == 0) {   // slots 6, 7 of the block's record: (wait | barrier | issue) and (S^T | softmax | PV) ticks, 16 bits each
        const unsigned bi = (((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) & 511) * 8;
        g_at_dbg[bi + 6] = (seg_t[0] & 0xffff) | ((seg_t[1] & 0xffff) << 16) | ((seg_t[2] & 0xffff) << 32);
        g_at_dbg[bi + 7] = (seg_t[3] & 0xffff) | ((seg_t[4] & 0xffff) << 16) | ((seg_t[5] & 0xffff) << 32);
    }
#endif
    if (!q_ok) return;
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    if constexpr (MXO) {
        const long long ro = (long long)b * a.o_b + (long long)t * a.o_t;
#pragma unroll
        for (int j = 0; j < DB / 2; ++j) {   // d blocks 2j, 2j + 1 = columns 32 j .. 32 j + 31 of head h
            float v[8];
            float mx = 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[k * 4 + e] = bf2f(f2bf(oacc[2 * j + k][e] * inv));
                    mx = fmaxf(mx, fabsf(v[k * 4 + e]));
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float scl = 1.0f;
            if (mx > 0.f) {
                int ex;
                const float fr = frexpf(mx / 448.0f, &ex);
                scl = ldexpf(1.0f, fr == 0.5f ? ex - 1 : ex);
            }
            scl = fmaxf(scl, 1.1754943508222875e-38f);
            const float is = 1.0f / scl;
            const int col = h * (int)a.o_h + 32 * j;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[k * 4] * is, v[k * 4 + 1] * is, 0, false);
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[k * 4 + 2] * is, v[k * 4 + 3] * is, pk, true);
                *(uint32_t*)(a.o8 + ro + col + k * 16 + 4 * g) = (uint32_t)pk;
            }
            if (g == 0) a.o8mx[((size_t)(col >> 7) * a.o8_rows + (size_t)(ro / a.o_t)) * 4 + ((col >> 5) & 3)] = (uint8_t)((__builtin_bit_cast(uint32_t, scl) >> 23) & 0xffu);
        }
        ATT(5);
        return;
    }
    if (a.so_o) {
        if (g == 0) {
            a.so_ml[srow * 2] = m_run;
            a.so_ml[srow * 2 + 1] = l_run;
        }
        float* so = a.so_o + srow * D + 4 * g;
#pragma unroll
        for (int db = 0; db < DB; ++db) *(float4*)(so + db * 16) = make_float4(oacc[db][0] * inv, oacc[db][1] * inv, oacc[db][2] * inv, oacc[db][3] * inv);
        return;
    }
    bf16_t* op = a.out + (long long)b * a.o_b + (long long)t * a.o_t + (long long)h * a.o_h + 4 * g;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        uint2 v;
        v.x = pack_bf2(oacc[db][0] * inv, oacc[db][1] * inv);
        v.y = pack_bf2(oacc[db][2] * inv, oacc[db][3] * inv);
        *(uint2*)(op + db * 16) = v;
    }
    ATT(5);
}
// when launch_d takes the workgroup-shared form: MHA at D = 128, length masks, enough query rows per (batch entry, head) to share the keys, enough
// workgroups to fill the chip (COVER_ATTN_SHARED=0: never)
static bool attn_shared_ok(const cover_attn_args* x) {
    static const char* env = getenv("COVER_ATTN_SHARED");
    if (env && env[0] == '0') return false;
    if (x->D != 128 || x->Hq != x->Hkv || x->Tq < 48 || x->n_seg < 1 || x->n_seg > 3) return false;
    for (int i = 0; i < x->n_seg; ++i)
        if (x->seg[i].mask_mode != COVER_MASK_LEN) return false;
    return (long long)((x->Tq + 63) / 64) * x->Hq * x->B >= 128;
}

template <int D, bool KSPLIT, int NWS = 4, bool MXO = false>
__global__ __launch_bounds__((KSPLIT && D > 128) ? 64 * NWS : 512) void attn_kernel(AttnDev a) {
    attn_body<D, KSPLIT, NWS, MXO>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}
// Two independent attention problems (the two row groups of a prefill pass: shared-prefix rows and the prompts' text rows)
// in ONE launch: both are far too small to fill the chip, so back to back they cost two latency floors. Key-split mode
// only; blockIdx.x enumerates (tile, batch) of problem 0, then of problem 1.
template <int D>
__global__ __launch_bounds__(512) void attn_kernel_dual(AttnDev a0, AttnDev a1, int tiles0, int n0, int tiles1) {
    int idx = blockIdx.x;
    if (idx < n0) {
        attn_body<D, true>(a0, idx % tiles0, blockIdx.y, idx / tiles0);
    } else {
        idx -= n0;
        attn_body<D, true>(a1, idx % tiles1, blockIdx.y, idx / tiles1);
    }
}

template <int D>
static hipError_t launch_dual_d(const AttnDev& a0, const AttnDev& a1, hipStream_t st) {
    const int t0 = (a0.R + 15) / 16, t1 = (a1.R + 15) / 16;
    const size_t lds = (size_t)(4 * (D / 16) * 4 * 64 + 2 * 4 * 16) * sizeof(float);
    dim3 grid(t0 * a0.B + t1 * a1.B, a0.Hkv), block(256);
    hipLaunchKernelGGL((attn_kernel_dual<D>), grid, block, lds, st, a0, a1, t0, t0 * a0.B, t1);
    return hipGetLastError();
}

template <int D>
static hipError_t launch_d(const AttnDev& a, hipStream_t st) {
    const int tiles = (a.R + 15) / 16;
    const long long qtiles = (long long)tiles * a.Hkv * a.B;
    static const char* e_max = getenv("COVER_ATTN_KSPLIT_MAX");
    static const char* e_nw8 = getenv("COVER_ATTN_NW8_MAX");
    long long ks_max = e_max ? atoll(e_max) : 1023;
    const long long nw8_max = e_nw8 ? atoll(e_nw8) : 0;
    // a pass RESUMED from a state (the chained decode pass of large-N candidate decode: 8 prompts x 64 samples x 32 heads = 1024 query
    // tiles over ~280 keys) is a chain of ~10 dependent key tiles per wave: split the keys over the block's waves there too
    // (config 5: 24.7 -> see profiles/ us per layer)
    if (a.si_o != nullptr && !e_max && D <= 128) ks_max = 4095;
    if (a.shared) {
        dim3 grid((a.Tq + 63) / 64, a.Hq, a.B);
        if (a.o8) hipLaunchKernelGGL((attn_shared_k<true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_shared_k<false>), grid, dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if (a.o8) {   // block-scaled output: the key-split kernel with four waves at D = 128 only (attention_mx_ok below says when)
        if constexpr (D == 128) {
            if (qtiles > ks_max) return hipErrorInvalidValue;
            const size_t lds = (size_t)(4 * (D / 16) * 4 * 64 + 2 * 4 * 16) * sizeof(float);
            hipLaunchKernelGGL((attn_kernel<D, true, 4, true>), dim3(tiles, a.Hkv, a.B), dim3(256), lds, st, a);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    }
    if (qtiles <= ks_max) {
        // too few query tiles to fill the chip (single-token decode, ViT-sized sequences): split the key tiles over the
        // 4 (or, when even 4 waves per tile leave most CUs idle and D allows the LDS merge buffer, 8) waves of a block
        const int nw = (qtiles <= nw8_max && D <= 128) ? 8 : 4;
        const size_t lds = (size_t)(nw * (D / 16) * 4 * 64 + 2 * nw * 16) * sizeof(float);
        dim3 grid(tiles, a.Hkv, a.B), block(64 * nw);
        if (nw == 8) {
            if constexpr (D <= 128) hipLaunchKernelGGL((attn_kernel<D, true, 8>), grid, block, lds, st, a);
        } else {
            hipLaunchKernelGGL((attn_kernel<D, true, 4>), grid, block, lds, st, a);
        }
    } else {
        const int nw = tiles >= 4 ? 4 : tiles;
        dim3 grid((tiles + nw - 1) / nw, a.Hkv, a.B), block(64 * nw);
        hipLaunchKernelGGL((attn_kernel<D, false>), grid, block, 0, st, a);
    }
    return hipGetLastError();
}

// Can this problem write its output block-scaled (cover_attn_args.out8)? MHA at D = 128 (a head = four 32-column blocks = one 128-deep k-tile of the consuming
// GEMM), whole rows of Hq * D bytes, final output (no state_out), and few enough query tiles for the key-split kernel (as launch_d decides).
bool attention_mx_ok(const cover_attn_args* x) {
    if (x->D != 128 || x->Hq != x->Hkv || x->state_out_o != nullptr || x->B <= 0 || x->Tq <= 0) return false;
    if (x->o_h_stride != 128 || x->o_t_stride != (long long)x->Hq * 128 || (x->o_b_stride % x->o_t_stride) != 0) return false;
    static const char* e_max = getenv("COVER_ATTN_KSPLIT_MAX");
    const long long ks_max = (x->state_in_o != nullptr && !e_max) ? 4095 : (e_max ? atoll(e_max) : 1023);
    const long long qtiles = (long long)((x->Tq + 15) / 16) * x->Hkv * x->B;
    return qtiles <= ks_max || attn_shared_ok(x);
}

static hipError_t build_attn_dev(const cover_attn_args* x, AttnDev& a) {
    if (x->n_seg < 1 || x->n_seg > 3 || x->Hq % x->Hkv != 0) return hipErrorInvalidValue;
    a.q = (const bf16_t*)x->q;
    a.out = (bf16_t*)x->out;
    a.q_b = x->q_b_stride; a.q_t = x->q_t_stride; a.q_h = x->q_h_stride;
    a.o_b = x->o_b_stride; a.o_t = x->o_t_stride; a.o_h = x->o_h_stride;
    a.B = x->B; a.Tq = x->Tq; a.Hq = x->Hq; a.Hkv = x->Hkv;
    a.G = x->Hq / x->Hkv;
    a.R = x->Tq * a.G;
    a.scale_log2e = x->scale * 1.4426950408889634f;
    a.n_seg = x->n_seg;
    a.si_o = x->state_in_o; a.si_ml = x->state_in_ml; a.so_o = x->state_out_o; a.so_ml = x->state_out_ml;
    if ((a.si_o == nullptr) != (a.si_ml == nullptr) || (a.so_o == nullptr) != (a.so_ml == nullptr)) return hipErrorInvalidValue;
    a.o8 = (uint8_t*)x->out8; a.o8mx = (uint8_t*)x->out8_mx; a.o8_rows = x->out8_rows;
    if (!a.o8 || !a.o8mx) { a.o8 = nullptr; a.o8mx = nullptr; }
    if (a.o8 && !attention_mx_ok(x)) return hipErrorInvalidValue;
    a.shared = attn_shared_ok(x);
    for (int i = 0; i < x->n_seg; ++i) {
        const cover_kv_segment& s = x->seg[i];
        SegDev& d = a.seg[i];
        d.k = (const bf16_t*)s.k; d.vt = (const bf16_t*)s.vt;
        d.k_slot = s.k_slot_stride; d.k_t = s.k_t_stride; d.k_h = s.k_h_stride;
        d.vt_slot = s.vt_slot_stride; d.vt_h = s.vt_h_stride; d.vt_d = s.vt_d_stride;
        d.slot_of_batch = s.slot_of_batch; d.len_of_batch = s.len_of_batch; d.vis_len = s.vis_len;
        d.len = s.len; d.mode = s.mask_mode; d.causal_off = s.causal_offset;
        if (d.mode == COVER_MASK_VISLEN && d.vis_len == nullptr) return hipErrorInvalidValue;
    }
    return hipSuccess;
}

hipError_t launch_attention_bf16(const cover_attn_args* x, hipStream_t st) {
    AttnDev a;
    hipError_t e = build_attn_dev(x, a);
    if (e != hipSuccess) return e;
    if (a.B <= 0 || a.R <= 0) return hipSuccess;
    const int pid = prof_enabled() ? prof_open(st, 2, 0.0) : -1;
    switch (x->D) {
        case 64: e = launch_d<64>(a, st); break;
        case 96: e = launch_d<96>(a, st); break;
        case 128: e = launch_d<128>(a, st); break;
        case 256: e = launch_d<256>(a, st); break;
        default: e = hipErrorInvalidValue;
    }
    prof_close(st, pid);
    return e;
}

// Both problems in one launch when both would run in key-split mode with 4 waves (else: two launches).
hipError_t launch_attention_bf16_pair(const cover_attn_args* x0, const cover_attn_args* x1, hipStream_t st) {
    AttnDev a0, a1;
    hipError_t e = build_attn_dev(x0, a0);
    if (e == hipSuccess) e = build_attn_dev(x1, a1);
    if (e != hipSuccess) return e;
    static const char* e_max = getenv("COVER_ATTN_KSPLIT_MAX");
    static const char* e_pair = getenv("COVER_ATTN_PAIR");
    const long long ks_max = e_max ? atoll(e_max) : 1023;
    const long long q0 = (long long)((a0.R + 15) / 16) * a0.Hkv * a0.B, q1 = (long long)((a1.R + 15) / 16) * a1.Hkv * a1.B;
    const bool pair = x0->D == x1->D && a0.Hkv == a1.Hkv && a0.B > 0 && a1.B > 0 && a0.R > 0 && a1.R > 0 && q0 <= ks_max && q1 <= ks_max &&
                      (x0->D == 64 || x0->D == 96 || x0->D == 128) && !(e_pair && e_pair[0] == '0');
    if (!pair) {
        e = launch_attention_bf16(x0, st);
        return e == hipSuccess ? launch_attention_bf16(x1, st) : e;
    }
    const int pid = prof_enabled() ? prof_open(st, 2, 0.0) : -1;
    switch (x0->D) {
        case 64: e = launch_dual_d<64>(a0, a1, st); break;
        case 96: e = launch_dual_d<96>(a0, a1, st); break;
        default: e = launch_dual_d<128>(a0, a1, st); break;
    }
    prof_close(st, pid);
    return e;
}
