// Candidate decode at LARGE N (BASELINE config 5: N = 512 candidates x 56 action tokens): RoPE + KV append + attention over the
// candidate's OWN generated tokens, one wave per (candidate, head), on the VALU -- the part of the decode attention that has no
// reuse at all (every candidate reads its own K/V once) and is therefore pure HBM streaming. The shared image prefix and the
// prompt's text keys, which ARE shared (by all / by the samples of one prompt), stay on the MFMA flash kernel (attention.hip) that
// is chained behind this one through the (o, m, l) state.
//
// Why not the MFMA kernels: at N = 512 the fused decode attention costs 109-362 us per layer and the three-launch path
// 58 (RoPE / V^T scatter) + 2 x 64 us (profiles/r02_g_config5_kernel_stats.txt): a 16-row MFMA tile holds ONE query per (candidate,
// head), and the transposed V cache makes every appended token 4096 scattered 2-byte stores per candidate.
//
// Own-token cache layout (the regions the legacy layout uses, re-interpreted): K and V both HEAD-MAJOR, NOT transposed,
//     [slot = candidate][h][t][d]        bf16, or e4m3 with one power-of-two fp32 scale per (slot, h, t) row
// so that (a) the append of a token is one contiguous row per head, (b) the keys of one (candidate, head) are one contiguous
// block: a wave streams them with 16-byte loads, lanes_per_key = D * esize / 16 lanes per key. fp8 ("fp8 KV" of config 5; the
// reference has no fp8 path): s_row = smallest 2^e with amax / 2^e <= 448, q = RNE_e4m3(x / s_row); scales of
// K / V live in the second half of the (bf16-sized) regions: scale index (slot * H + h) * t_cap + t at byte offset region_elems.
//
// Arithmetic: RoPE exactly as rope_kv_write (HF rotate_half in bf16 arithmetic, or the pi0 fp32 form); scores q.k in fp32 over
// the bf16 / de-quantised values, scaled after the product (eager_attention_forward order), softmax in fp32 with the probabilities
// rounded to bf16 before PV and the row sum taken over the unrounded ones (as attention.hip); the state handed on is
// (O / l in fp32, m in scaled-log2 units, l) -- attention.hip's chaining convention (cover_attn_args.state_in_*).
#include "common.h"
#include "kernels.h"

struct OwnAttnDev {
    bf16_t* qkv; int ld_qkv;                 // bf16 [N][3*H*D]; q is rotated IN PLACE (the chained MFMA pass reads it)
    int N, H;
    float scale_log2e;
    const int* positions; const float* cos_t; const float* sin_t; int n_pos, rope_mode;
    void* k; void* v;                        // own-token regions (bf16 or e4m3 elements)
    float* k_scale; float* v_scale;          // fp8 only
    long long slot_stride;                   // elements between candidates' blocks = H * t_cap * D
    int t_cap;
    const int* slot_of_batch;                // [N] or NULL (slot = n)
    int write_t;                             // position of the token appended here; keys 0..write_t are attended
    float* state_o; float* state_ml;         // fp32 [N][H][D], [N][H][2]
};

// One wave per (candidate n, head h); 4 waves per block = 4 consecutive heads of one candidate.
// NIT = 16-byte loads per lane for the whole K (or V) block of this (n, h) = ceil(keys / keys_per_load), compile-time so that
// every load of a pass is issued unconditionally up front (rows beyond the last key re-read it and are masked).
template <int D, bool F8, int NIT>
__global__ __launch_bounds__(256) void decode_own_attn_k(OwnAttnDev a) {
    constexpr int ES = F8 ? 1 : 2;                 // bytes per cached element
    constexpr int DPL = 16 / ES;                   // d's per lane per load
    constexpr int LPK = D / DPL;                   // lanes per key
    constexpr int KPL = 64 / LPK;                  // keys per wave-load
    constexpr int HALF = D / 2, PPL = HALF / 64 > 0 ? HALF / 64 : 1;   // rotation pairs per lane (D = 128: 1, D = 64: half the lanes idle)
    static_assert(D == 64 || D == 128, "head dims of the Llama-style candidate decode");
    __shared__ float xs[4][3][D];                   // per wave: roped q, roped k_new, v_new as fp32 (exactly bf16 / de-quantised values)
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x;
    const int h = blockIdx.y * 4 + w;
    if (h >= a.H) return;                           // (no block barrier below: waves are independent)
    const int slot = a.slot_of_batch ? a.slot_of_batch[n] : n;
    int pos = 0;
    if (a.rope_mode != 0) {
        pos = a.positions[n];
        pos = pos < 0 ? 0 : (pos >= a.n_pos ? a.n_pos - 1 : pos);
    }
    // ---- the cached K rows (and, while they fit the registers, V rows) are requested FIRST: they depend on nothing but the slot, and
    //      the q / k / v + RoPE work below then runs under their latency (the chain was: indices -> qkv -> LDS -> K -> scores -> V) ----
    constexpr bool VEARLY = NIT <= 8;
    const int j = lane % LPK, kq = lane / LPK;            // this lane's d range [j*DPL, (j+1)*DPL) of key (KPL * it + kq)
    const int nkeys = a.write_t;                          // cached keys (the appended one comes from LDS)
    const size_t blk = (size_t)slot * a.slot_stride + (size_t)h * a.t_cap * D;      // elements to this (slot, h) block
    const size_t sblk = ((size_t)slot * a.H + h) * a.t_cap;
    uint4 kr[NIT], vr[NIT];
    float ksc[NIT], vsc[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int t = it * KPL + kq;
        t = t < nkeys ? t : (nkeys > 0 ? nkeys - 1 : 0);
        kr[it] = *(const uint4*)((const char*)a.k + (blk + (size_t)t * D) * ES + j * 16);
        if constexpr (F8) ksc[it] = a.k_scale[sblk + t];
    }
    if constexpr (VEARLY) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int t = it * KPL + kq;
            t = t < nkeys ? t : (nkeys > 0 ? nkeys - 1 : 0);
            vr[it] = *(const uint4*)((const char*)a.v + (blk + (size_t)t * D) * ES + j * 16);
            if constexpr (F8) vsc[it] = a.v_scale[sblk + t];
        }
    }
    // ---- q / k_new / v_new of (n, h): rotation pair (i, i + HALF) per lane ----
    bf16_t* row = a.qkv + (size_t)n * a.ld_qkv;
    const int i = lane;                             // D = 64: lanes >= 32 idle in this part
    const bool pair_ok = i < HALF;
    float q1 = 0.f, q2 = 0.f, k1 = 0.f, k2 = 0.f, v1 = 0.f, v2 = 0.f;
    if (pair_ok) {
        const bf16_t* qp = row + (size_t)h * D;
        const bf16_t* kp = row + (size_t)(a.H + h) * D;
        const bf16_t* vp = row + (size_t)(2 * a.H + h) * D;
        q1 = bf2f(qp[i]); q2 = bf2f(qp[i + HALF]);
        k1 = bf2f(kp[i]); k2 = bf2f(kp[i + HALF]);
        v1 = bf2f(vp[i]); v2 = bf2f(vp[i + HALF]);
        if (a.rope_mode != 0) {
            float c = a.cos_t[(size_t)pos * HALF + i], s = a.sin_t[(size_t)pos * HALF + i];
            if (a.rope_mode == 2) {   // HF rotate_half in bf16 arithmetic (rope_kv_write's expressions)
                c = bfround(c); s = bfround(s);
                const float o1 = bfround(bfround(q1 * c) + bfround(-q2 * s)), o2 = bfround(bfround(q2 * c) + bfround(q1 * s));
                const float p1 = bfround(bfround(k1 * c) + bfround(-k2 * s)), p2 = bfround(bfround(k2 * c) + bfround(k1 * s));
                q1 = o1; q2 = o2; k1 = p1; k2 = p2;
            } else {                  // apply_rope (paligemma_with_expert.py:34-57): fp32, one rounding
#pragma clang fp contract(off)
                const float o1 = bfround(q1 * c - q2 * s), o2 = bfround(q2 * c + q1 * s);
                const float p1 = bfround(k1 * c - k2 * s), p2 = bfround(k2 * c + k1 * s);
                q1 = o1; q2 = o2; k1 = p1; k2 = p2;
            }
            bf16_t* qw = row + (size_t)h * D;       // rotated q back in place for the chained pass over the shared segments
            qw[i] = f2bf(q1); qw[i + HALF] = f2bf(q2);
        }
    }
    // ---- append k_new / v_new to the cache (row (slot, h, write_t)); fp8: per-row power-of-two scale ----
    const size_t rbase = (size_t)slot * a.slot_stride + ((size_t)h * a.t_cap + a.write_t) * D;
    if constexpr (F8) {
        float ka = wave_max(fmaxf(fabsf(k1), fabsf(k2))), va = wave_max(fmaxf(fabsf(v1), fabsf(v2)));
        auto pow2_scale = [](float mx) {
            float s = 1.0f;
            if (mx > 0.f) {
                int e;
                const float f = frexpf(mx / 448.0f, &e);
                s = ldexpf(1.0f, f == 0.5f ? e - 1 : e);
            }
            return s;
        };
        const float ks = pow2_scale(ka), vs = pow2_scale(va);
        const float kinv = 1.0f / ks, vinv = 1.0f / vs;
        // quantise, and keep the DE-QUANTISED values for this pass (the new key is attended from registers / LDS)
        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(k1 * kinv, k2 * kinv, 0, false);
        const int pv = __builtin_amdgcn_cvt_pk_fp8_f32(v1 * vinv, v2 * vinv, 0, false);
        const f32x2_t kb = __builtin_amdgcn_cvt_pk_f32_fp8((uint32_t)pk, false), vb = __builtin_amdgcn_cvt_pk_f32_fp8((uint32_t)pv, false);
        k1 = kb[0] * ks; k2 = kb[1] * ks; v1 = vb[0] * vs; v2 = vb[1] * vs;
        if (pair_ok) {
            uint8_t* kd = (uint8_t*)a.k + rbase;
            uint8_t* vd = (uint8_t*)a.v + rbase;
            kd[i] = (uint8_t)(pk & 0xff); kd[i + HALF] = (uint8_t)((pk >> 8) & 0xff);
            vd[i] = (uint8_t)(pv & 0xff); vd[i + HALF] = (uint8_t)((pv >> 8) & 0xff);
        }
        if (lane == 0) {
            const size_t si = ((size_t)slot * a.H + h) * a.t_cap + a.write_t;
            a.k_scale[si] = ks;
            a.v_scale[si] = vs;
        }
    } else {
        if (pair_ok) {
            bf16_t* kd = (bf16_t*)a.k + rbase;
            bf16_t* vd = (bf16_t*)a.v + rbase;
            kd[i] = f2bf(k1); kd[i + HALF] = f2bf(k2);
            vd[i] = f2bf(v1); vd[i + HALF] = f2bf(v2);
        }
    }
    // ---- hand q / k_new / v_new to the streaming layout through LDS (wave-private: no block barrier) ----
    if (pair_ok) {
        xs[w][0][i] = q1; xs[w][0][i + HALF] = q2;
        xs[w][1][i] = k1; xs[w][1][i + HALF] = k2;
        xs[w][2][i] = v1; xs[w][2][i + HALF] = v2;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: LDS writes of the wave are visible to its later reads in order
    float qv[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) qv[e] = xs[w][0][j * DPL + e];
    auto unpack = [&](const uint4& raw, float (&x)[DPL]) {
        const uint32_t wv[4] = {raw.x, raw.y, raw.z, raw.w};
        if constexpr (F8) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(wv[u], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(wv[u], true);
                x[4 * u] = lo[0]; x[4 * u + 1] = lo[1]; x[4 * u + 2] = hi[0]; x[4 * u + 3] = hi[1];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x[2 * u] = bf2f((bf16_t)(wv[u] & 0xffffu));
                x[2 * u + 1] = bf2f((bf16_t)(wv[u] >> 16));
            }
        }
    };
    // ---- pass 1: scores of the cached keys + the new key ----
    float sc[NIT + 1];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        float x[DPL];
        unpack(kr[it], x);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < DPL; ++e) s += qv[e] * x[e];
#pragma unroll
        for (int o = 1; o < LPK; o <<= 1) s += __shfl_xor(s, o);
        if constexpr (F8) s *= ksc[it];
        sc[it] = (it * KPL + kq < nkeys) ? s * a.scale_log2e : -INFINITY;
    }
    {   // the appended key: attended by lane group 0 only (the other groups carry -inf for it)
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < DPL; ++e) s += qv[e] * xs[w][1][j * DPL + e];
#pragma unroll
        for (int o = 1; o < LPK; o <<= 1) s += __shfl_xor(s, o);
        sc[NIT] = kq == 0 ? s * a.scale_log2e : -INFINITY;
    }
    // ---- long segments: V loads in flight while the softmax statistics are formed (K's registers are free by now) ----
    if constexpr (!VEARLY) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int t = it * KPL + kq;
            t = t < nkeys ? t : (nkeys > 0 ? nkeys - 1 : 0);
            vr[it] = *(const uint4*)((const char*)a.v + (blk + (size_t)t * D) * ES + j * 16);
            if constexpr (F8) vsc[it] = a.v_scale[sblk + t];
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int it = 0; it <= NIT; ++it) m = fmaxf(m, sc[it]);
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));   // over the key groups (every lane of a group holds its scores)
    float l = 0.f;
    float p[NIT + 1];
#pragma unroll
    for (int it = 0; it <= NIT; ++it) {
        const float pe = exp2f(sc[it] - m);      // the new key is always visible: m is finite
        l += pe;
        p[it] = bfround(pe);                     // probabilities enter PV rounded to bf16 (eager attention: P cast to the value dtype)
    }
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) l += __shfl_xor(l, o);
    // ---- pass 2: O = P . V over this lane's d range, then summed over the key groups ----
    float acc[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) acc[e] = p[NIT] * xs[w][2][j * DPL + e];   // (p[NIT] is 0 outside lane group 0)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        // a masked slot (index clamped to a row that may never have been written: write_t == 0 reads row 0 before the append lands)
        // contributes exactly nothing -- its bits and its scale are replaced, not multiplied by a zero probability (0 * NaN)
        const bool vis = it * KPL + kq < nkeys;
        float x[DPL];
        unpack(vis ? vr[it] : make_uint4(0u, 0u, 0u, 0u), x);
        const float pw = vis ? (F8 ? p[it] * vsc[it] : p[it]) : 0.f;   // the row scale is a power of two: exact
#pragma unroll
        for (int e = 0; e < DPL; ++e) acc[e] += pw * x[e];
    }
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) {
#pragma unroll
        for (int e = 0; e < DPL; ++e) acc[e] += __shfl_xor(acc[e], o);
    }
    // ---- state: normalised output, running max (scaled-log2 units), running sum ----
    const size_t srow = (size_t)n * a.H + h;
    if (kq == 0) {
        const float inv = 1.0f / l;
        float* so = a.state_o + srow * D + j * DPL;
#pragma unroll
        for (int e = 0; e < DPL; e += 4) *(float4*)(so + e) = make_float4(acc[e] * inv, acc[e + 1] * inv, acc[e + 2] * inv, acc[e + 3] * inv);
        if (j == 0) {
            a.state_ml[srow * 2] = m;
            a.state_ml[srow * 2 + 1] = l;
        }
    }
}

template <int D, bool F8>
static hipError_t launch_own_d(const OwnAttnDev& a, hipStream_t st) {
    constexpr int KPL = 64 / (D / (F8 ? 16 : 8));
    const int nit = (a.write_t + KPL - 1) / KPL;         // loads per lane per pass for the cached keys
    dim3 grid(a.N, (a.H + 3) / 4), block(256);
#define OWN(NIT_) hipLaunchKernelGGL((decode_own_attn_k<D, F8, NIT_>), grid, block, 0, st, a)
    if (nit <= 1) OWN(1);
    else if (nit <= 2) OWN(2);
    else if (nit <= 4) OWN(4);
    else if (nit <= 6) OWN(6);
    else if (nit <= 8) OWN(8);
    else if (nit <= 12) OWN(12);
    else if (nit <= 16) OWN(16);
    else return hipErrorInvalidValue;                    // more own tokens than 16 loads cover (bf16, D = 128: 64 keys)
#undef OWN
    return hipGetLastError();
}

hipError_t launch_decode_own_attention(const cover_own_attn_args* x, hipStream_t st) {
    if (!x || !x->qkv || !x->k || !x->v || !x->state_o || !x->state_ml || x->N <= 0 || x->H <= 0) return hipErrorInvalidValue;
    if (x->write_t < 0 || x->write_t >= x->t_cap || (x->ld_qkv & 1)) return hipErrorInvalidValue;
    if (x->rope_mode != 0 && (!x->positions || !x->cos_table || !x->sin_table)) return hipErrorInvalidValue;
    if (x->fp8 && (!x->k_scale || !x->v_scale)) return hipErrorInvalidValue;
    OwnAttnDev a;
    a.qkv = (bf16_t*)x->qkv; a.ld_qkv = x->ld_qkv; a.N = x->N; a.H = x->H;
    a.scale_log2e = x->scale * 1.4426950408889634f;
    a.positions = x->positions; a.cos_t = x->cos_table; a.sin_t = x->sin_table; a.n_pos = x->n_pos; a.rope_mode = x->rope_mode;
    a.k = x->k; a.v = x->v; a.k_scale = x->k_scale; a.v_scale = x->v_scale;
    a.slot_stride = x->slot_stride; a.t_cap = x->t_cap; a.slot_of_batch = x->slot_of_batch; a.write_t = x->write_t;
    a.state_o = x->state_o; a.state_ml = x->state_ml;
    const int pid = prof_enabled() ? prof_open(st, 2, 0.0) : -1;
    hipError_t e;
    if (x->D == 128) e = x->fp8 ? launch_own_d<128, true>(a, st) : launch_own_d<128, false>(a, st);
    else if (x->D == 64) e = x->fp8 ? launch_own_d<64, true>(a, st) : launch_own_d<64, false>(a, st);
    else e = hipErrorInvalidValue;
    prof_close(st, pid);
    return e;
}
