// HBM-bound row kernels: norms, RoPE + KV placement, gathers, patch extraction, casts.
// All bf16 traffic is 16 bytes per lane; reductions are wave shuffles + one LDS hop per block.
#include "common.h"
#include "kernels.h"

// ---------------------------------------------------------------------------------------------------
// LayerNorm / RMSNorm: one 256-thread block per row, row cached in registers (dim <= 8192, dim % 8 == 0)
// ---------------------------------------------------------------------------------------------------
#define NORM_MAX_CHUNKS 4

template <bool IN_F32>
__device__ __forceinline__ void load_row8(const void* x, size_t off, float (&v)[8]) {
    if (IN_F32) {
        const float4 a = *(const float4*)((const float*)x + off);
        const float4 b = *(const float4*)((const float*)x + off + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const uint4 u = *(const uint4*)((const bf16_t*)x + off);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
}
__device__ __forceinline__ void store_row8(bf16_t* y, size_t off, const float (&v)[8]) {
    uint4 u;
    u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
    *(uint4*)(y + off) = u;
}

__global__ __launch_bounds__(256) void layernorm_bf16_k(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        const float* __restrict__ b, bf16_t* __restrict__ y, int ldy,
                                                        int dim, float eps) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    const int nch = dim >> 3;
    float v[NORM_MAX_CHUNKS][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NORM_MAX_CHUNKS; ++c) {
        const int ch = threadIdx.x + c * 256;
        if (ch < nch) {
            load_row8<false>(x, (size_t)row * ldx + ch * 8, v[c]);
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[c][i];
        }
    }
    const float mean = block_sum(s, red) / dim;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NORM_MAX_CHUNKS; ++c) {
        const int ch = threadIdx.x + c * 256;
        if (ch < nch) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = v[c][i] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(block_sum(q, red) / dim + eps);
#pragma unroll
    for (int c = 0; c < NORM_MAX_CHUNKS; ++c) {
        const int ch = threadIdx.x + c * 256;
        if (ch < nch) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int col = ch * 8 + i;
                o[i] = (v[c][i] - mean) * rstd * w[col] + (b ? b[col] : 0.f);
            }
            store_row8(y, (size_t)row * ldy + ch * 8, o);
        }
    }
}

__device__ __forceinline__ float e4m3_pow2_scale_rows(float mx) {   // smallest power of two s with mx / s <= 448; 1 for an all-zero row
    float s = 1.0f;
    if (mx > 0.f) {
        int e;
        const float f = frexpf(mx / 448.0f, &e);
        s = ldexpf(1.0f, f == 0.5f ? e - 1 : e);
    }
    return s;
}
// style 0 (Gemma): y = bf16(x * rstd * (w_offset + w)) ; style 1 (Llama): y = bf16(w * bf16(x * rstd))
template <bool IN_F32, bool Q8 = false>
__global__ __launch_bounds__(256) void rmsnorm_bf16_k(const void* __restrict__ x, int ldx, const float* __restrict__ w,
                                                      float w_offset, int style, bf16_t* __restrict__ y, int ldy, int dim,
                                                      float eps, uint8_t* __restrict__ q8 = nullptr, int ld8 = 0, float* __restrict__ q8s = nullptr) {
    __shared__ float red[32];
    const int row = blockIdx.x;
    const int nch = dim >> 3;
    float v[NORM_MAX_CHUNKS][8];
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NORM_MAX_CHUNKS; ++c) {
        const int ch = threadIdx.x + c * 256;
        if (ch < nch) {
            load_row8<IN_F32>(x, (size_t)row * ldx + ch * 8, v[c]);
#pragma unroll
            for (int i = 0; i < 8; ++i) q += v[c][i] * v[c][i];
        }
    }
    const float rstd = rsqrtf(block_sum(q, red) / dim + eps);
#pragma unroll
    for (int c = 0; c < NORM_MAX_CHUNKS; ++c) {
        const int ch = threadIdx.x + c * 256;
        if (ch < nch) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int col = ch * 8 + i;
                const float ww = w ? w[col] : 0.f;
                if (style == 1) o[i] = ww * bfround(v[c][i] * rstd);
                else o[i] = v[c][i] * rstd * (w_offset + ww);
            }
            store_row8(y, (size_t)row * ldy + ch * 8, o);
            if (Q8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[c][i] = bfround(o[i]);   // the stored bf16 row is what gets quantised
            }
        }
    }
    if constexpr (Q8) {   // e4m3 twin of the y row in the MX MFMA operand order (cover_quantize_act_fp8's arithmetic)
        float mx = 0.f;
#pragma unroll
        for (int c = 0; c < NORM_MAX_CHUNKS; ++c)
            if ((int)(threadIdx.x + c * 256) < nch)
#pragma unroll
                for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(v[c][i]));
        mx = block_max(mx, red + 16);
        const float sc = e4m3_pow2_scale_rows(mx), inv = 1.0f / sc;
        if (threadIdx.x == 0) q8s[row] = sc;
        uint8_t* qrow = q8 + (size_t)row * ld8;
#pragma unroll
        for (int c = 0; c < NORM_MAX_CHUNKS; ++c) {
            const int ch = threadIdx.x + c * 256;
            if (ch < nch) {
                const int k = ch * 8;
                int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * inv, v[c][1] * inv, 0, false);
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * inv, v[c][3] * inv, lo, true);
                int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] * inv, v[c][5] * inv, 0, false);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] * inv, v[c][7] * inv, hi, true);
                *(uint2*)(qrow + (k >> 6) * 64 + ((k >> 3) & 3) * 16 + ((k >> 5) & 1) * 8) = make_uint2((uint32_t)lo, (uint32_t)hi);
            }
        }
    }
}

hipError_t launch_layernorm_bf16(const bf16_t* x, int ldx, const float* w, const float* b, bf16_t* y, int ldy, int rows,
                                 int dim, float eps, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    if (dim % 8 || dim > NORM_MAX_CHUNKS * 256 * 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_bf16_k, dim3(rows), dim3(256), 0, st, x, ldx, w, b, y, ldy, dim, eps);
    return hipGetLastError();
}
hipError_t launch_rmsnorm(const void* x, int x_f32, int ldx, const float* w, float w_offset, int style, bf16_t* y, int ldy,
                          int rows, int dim, float eps, hipStream_t st, uint8_t* q8, int ld8, float* q8s) {
    if (rows <= 0) return hipSuccess;
    if (dim % 8 || dim > NORM_MAX_CHUNKS * 256 * 8) return hipErrorInvalidValue;
    if (q8 && q8s) {
        if ((dim & 127) || ld8 < dim || (ld8 & 15)) return hipErrorInvalidValue;
        if (x_f32)
            hipLaunchKernelGGL((rmsnorm_bf16_k<true, true>), dim3(rows), dim3(256), 0, st, x, ldx, w, w_offset, style, y, ldy, dim, eps, q8, ld8, q8s);
        else
            hipLaunchKernelGGL((rmsnorm_bf16_k<false, true>), dim3(rows), dim3(256), 0, st, x, ldx, w, w_offset, style, y, ldy, dim, eps, q8, ld8, q8s);
        return hipGetLastError();
    }
    if (x_f32)
        hipLaunchKernelGGL((rmsnorm_bf16_k<true, false>), dim3(rows), dim3(256), 0, st, x, ldx, w, w_offset, style, y, ldy, dim, eps, (uint8_t*)nullptr, 0, (float*)nullptr);
    else
        hipLaunchKernelGGL((rmsnorm_bf16_k<false, false>), dim3(rows), dim3(256), 0, st, x, ldx, w, w_offset, style, y, ldy, dim, eps, (uint8_t*)nullptr, 0, (float*)nullptr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// RoPE + KV placement: one wave per (row, head) of the fused qkv buffer
// ---------------------------------------------------------------------------------------------------
// V placement for multi-token groups (prefill): lanes = 64 CONSECUTIVE TOKENS of one (batch element, kv head), a wave = one
// 8-wide d chunk of them. Each lane reads its token's 16 bytes; each of the eight stores then writes 64 consecutive tokens of
// one V^T row = one 128-byte line, where the (row, head)-per-wave form writes one 2-byte element into each of 64 lines.
__device__ __forceinline__ bool rope_v_tokens(const cover_rope_args& a) { return a.T >= 16 && a.n_splits <= 0 && (a.D & 7) == 0 && (a.ld_qkv & 7) == 0 && (((uintptr_t)a.qkv) & 15) == 0; }
__device__ __forceinline__ int rope_v_waves(const cover_rope_args& a) { return a.B * a.Hkv * (a.D >> 3) * ((a.T + 63) >> 6); }
__device__ __forceinline__ void rope_v_body(const cover_rope_args& a, int wid) {
    const int lane = threadIdx.x & 63;
    const int tg = (a.T + 63) >> 6, dch = a.D >> 3;
    const int g = wid % tg, c = (wid / tg) % dch, h = (wid / (tg * dch)) % a.Hkv, b = wid / (tg * dch * a.Hkv);
    const int t = g * 64 + lane;
    if (b >= a.B || t >= a.T) return;
    const int row = b * a.T + t;
    const bf16_t* src = (const bf16_t*)a.qkv + (size_t)row * a.ld_qkv + (size_t)(a.Hq + a.Hkv + h) * a.D + 8 * c;
    const uint4 u = *(const uint4*)src;
    const int slot = a.slot_of_batch ? a.slot_of_batch[b] : b;
    const int tt = a.t_offset + (a.t_offset_of_batch ? a.t_offset_of_batch[b] : 0) + t;
    bf16_t* dst = (bf16_t*)a.vt_cache + (size_t)slot * a.vt_slot_stride + (size_t)h * a.vt_h_stride + (size_t)(8 * c) * a.vt_d_stride + tt;
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        dst[(size_t)(2 * e) * a.vt_d_stride] = (bf16_t)(w[e] & 0xffffu);
        dst[(size_t)(2 * e + 1) * a.vt_d_stride] = (bf16_t)(w[e] >> 16);
    }
}

// q / k heads of multi-token groups, vectorised: a lane owns 8 consecutive elements of BOTH halves of one (row, head) -- two 16-byte
// loads, the 2 x 32 bytes of its cos / sin entries, two 16-byte stores -- so a wave rotates 64 / (D / 16) heads instead of one
// (prefill of a 7B decoder, 448 rows x 64 q / k heads: 28 672 one-head waves of 2-byte accesses were 5 rounds of resident waves, each a
// chain of three dependent loads). Same per-element arithmetic as the scalar body below.
__host__ __device__ __forceinline__ bool rope_qk_vec(const cover_rope_args& a) {
    return a.T >= 16 && a.n_splits <= 0 && (a.ld_qkv & 7) == 0 && (((uintptr_t)a.qkv) & 15) == 0 && (a.D == 64 || a.D == 128 || a.D == 256) &&
           a.rope_mode != 0 && (((uintptr_t)a.cos_table | (uintptr_t)a.sin_table) & 15) == 0 &&
           (!a.k_cache || ((((uintptr_t)a.k_cache) & 15) == 0 && ((a.k_slot_stride | a.k_t_stride | a.k_h_stride) & 7) == 0));
}
__device__ __forceinline__ void rope_qk_vec_body(const cover_rope_args& a, int wid) {
    const int lane = threadIdx.x & 63;
    const int half = a.D >> 1, lpi = half >> 3, ipw = 64 / lpi;
    const int nqk = a.Hq + a.Hkv;
    const long long item = (long long)wid * ipw + lane / lpi;
    if (item >= (long long)a.B * a.T * nqk) return;
    const int c = lane % lpi;
    const int row = (int)(item / nqk), hh = (int)(item - (long long)row * nqk);
    const int b = row / a.T, t = row - b * a.T;
    bf16_t* src = (bf16_t*)a.qkv + (size_t)row * a.ld_qkv + (size_t)hh * a.D;
    bf16_t* dst = src;
    if (hh >= a.Hq && a.k_cache) {
        const int slot = a.slot_of_batch ? a.slot_of_batch[b] : b;
        const int tt = a.t_offset + (a.t_offset_of_batch ? a.t_offset_of_batch[b] : 0) + t;
        dst = (bf16_t*)a.k_cache + (size_t)slot * a.k_slot_stride + (size_t)tt * a.k_t_stride + (size_t)(hh - a.Hq) * a.k_h_stride;
    }
    int pos = a.positions ? a.positions[row] : t;
    pos = pos < 0 ? 0 : (pos >= a.n_pos ? a.n_pos - 1 : pos);
    const float* ct = a.cos_table + (size_t)pos * half + c * 8;
    const float* sn = a.sin_table + (size_t)pos * half + c * 8;
    const uint4 u1 = *(const uint4*)(src + c * 8), u2 = *(const uint4*)(src + half + c * 8);
    const float4 c0 = *(const float4*)ct, c1 = *(const float4*)(ct + 4), s0 = *(const float4*)sn, s1 = *(const float4*)(sn + 4);
    const uint32_t w1[4] = {u1.x, u1.y, u1.z, u1.w}, w2[4] = {u2.x, u2.y, u2.z, u2.w};
    const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    float o1[8], o2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x1 = __uint_as_float((e & 1) ? (w1[e >> 1] & 0xffff0000u) : (w1[e >> 1] << 16));
        const float x2 = __uint_as_float((e & 1) ? (w2[e >> 1] & 0xffff0000u) : (w2[e >> 1] << 16));
        float cc = cv[e], ss = sv[e];
        if (a.rope_mode == 2) {  // HF rotate_half in bf16 arithmetic
            cc = bfround(cc); ss = bfround(ss);
            o1[e] = bfround(bfround(x1 * cc) + bfround(-x2 * ss));
            o2[e] = bfround(bfround(x2 * cc) + bfround(x1 * ss));
        } else {  // apply_rope (paligemma_with_expert.py:34-57): fp32, one rounding
#pragma clang fp contract(off)   // torch rounds each product: no FMA here
            o1[e] = x1 * cc - x2 * ss;
            o2[e] = x2 * cc + x1 * ss;
        }
    }
    uint4 r1, r2;
    r1.x = pack_bf2(o1[0], o1[1]); r1.y = pack_bf2(o1[2], o1[3]); r1.z = pack_bf2(o1[4], o1[5]); r1.w = pack_bf2(o1[6], o1[7]);
    r2.x = pack_bf2(o2[0], o2[1]); r2.y = pack_bf2(o2[2], o2[3]); r2.z = pack_bf2(o2[4], o2[5]); r2.w = pack_bf2(o2[6], o2[7]);
    *(uint4*)(dst + c * 8) = r1;
    *(uint4*)(dst + half + c * 8) = r2;
}

__device__ __forceinline__ void rope_kv_body(const cover_rope_args& a, int wid) {
    const int lane = threadIdx.x & 63;
    const int nh = a.Hq + 2 * a.Hkv;
    const int rows = a.B * a.T;
    if (rope_v_tokens(a)) {   // q / k heads one wave per (row, head) as below (or vectorised); the V heads by token groups
        const int nqk = a.Hq + a.Hkv;
        const bool vec = rope_qk_vec(a);
        const int ipw = vec ? 64 / (a.D >> 4) : 1;
        // no rotation and no K cache (the ViT towers' V transposition): the q / k heads stay where they are, no waves for them
        const int qk_waves = (a.rope_mode == 0 && !a.k_cache) ? 0 : (rows * nqk + ipw - 1) / ipw;
        if (wid >= qk_waves) {
            if (wid < qk_waves + rope_v_waves(a)) rope_v_body(a, wid - qk_waves);
            return;
        }
        if (vec) { rope_qk_vec_body(a, wid); return; }
        const int row = wid / nqk, hh = wid - row * nqk;
        wid = row * nh + hh;
    }
    if (wid >= rows * nh) return;
    const int row = wid / nh, hh = wid - row * nh;
    const int b = row / a.T, t = row - b * a.T;
    bf16_t* src = (bf16_t*)a.qkv + (size_t)row * a.ld_qkv + (size_t)hh * a.D;
    const int half = a.D >> 1;
    const int slot = a.slot_of_batch ? a.slot_of_batch[b] : b;
    const int tt = a.t_offset + (a.t_offset_of_batch ? a.t_offset_of_batch[b] : 0) + t;
    // element d of this (row, head): either the stored bf16 projection or, on the weight-streaming path, the split-K
    // partial sums folded here (bf16-rounded like the projection output would have been)
    const int ncols = nh * a.D;
    const size_t pbase = (size_t)row * ncols + (size_t)hh * a.D;
    const size_t pstride = (size_t)rows * ncols;
    auto ld = [&](int d) -> float {
        if (a.n_splits <= 0) return bf2f(src[d]);
        float v = 0.f;
        for (int s = 0; s < a.n_splits; ++s) v += a.partial[s * pstride + pbase + d];
        if (a.bias) v += a.bias[hh * a.D + d];
        return bfround(v);
    };

    if (hh >= a.Hq + a.Hkv) {  // V head -> transposed cache
        const int h = hh - a.Hq - a.Hkv;
        bf16_t* dst = (bf16_t*)a.vt_cache + (size_t)slot * a.vt_slot_stride + (size_t)h * a.vt_h_stride + tt;
        for (int d = lane; d < a.D; d += 64) dst[(size_t)d * a.vt_d_stride] = f2bf(ld(d));
        return;
    }
    const bool is_k = hh >= a.Hq;
    bf16_t* dst = src;  // q (and k without a cache) rotate in place
    if (is_k && a.k_cache)
        dst = (bf16_t*)a.k_cache + (size_t)slot * a.k_slot_stride + (size_t)tt * a.k_t_stride +
              (size_t)(hh - a.Hq) * a.k_h_stride;
    if (a.rope_mode == 0) {
        if (dst != src || a.n_splits > 0)
            for (int d = lane; d < a.D; d += 64) dst[d] = f2bf(ld(d));
        return;
    }
    int pos = a.positions ? a.positions[row] : t;
    pos = pos < 0 ? 0 : (pos >= a.n_pos ? a.n_pos - 1 : pos);
    const float* ct = a.cos_table + (size_t)pos * half;
    const float* stb = a.sin_table + (size_t)pos * half;
    for (int i = lane; i < half; i += 64) {
        const float x1 = ld(i), x2 = ld(i + half);
        float c = ct[i], s = stb[i], o1, o2;
        if (a.rope_mode == 2) {  // HF rotate_half in bf16 arithmetic
            c = bfround(c); s = bfround(s);
            o1 = bfround(bfround(x1 * c) + bfround(-x2 * s));
            o2 = bfround(bfround(x2 * c) + bfround(x1 * s));
        } else {  // apply_rope (paligemma_with_expert.py:34-57): fp32, one rounding
#pragma clang fp contract(off)   // torch rounds each product: no FMA here
            o1 = x1 * c - x2 * s;
            o2 = x2 * c + x1 * s;
        }
        dst[i] = f2bf(o1);
        dst[i + half] = f2bf(o2);
    }
}
__global__ __launch_bounds__(256) void rope_kv_write_k(cover_rope_args a) {
    rope_kv_body(a, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
}
// two row groups of one pass in one launch (waves [0, waves0) belong to the first)
__global__ __launch_bounds__(256) void rope_kv_write2_k(cover_rope_args a0, cover_rope_args a1, int waves0) {
    const int wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid < waves0) rope_kv_body(a0, wid);
    else rope_kv_body(a1, wid - waves0);
}
static bool rope_args_ok(const cover_rope_args* a) {
    return a->vt_cache && !(a->D & 1) && (a->rope_mode == 0 || (a->cos_table && a->sin_table));
}
static long long rope_waves_host(const cover_rope_args* a) {
    const bool vt = a->T >= 16 && a->n_splits <= 0 && (a->D & 7) == 0 && (a->ld_qkv & 7) == 0 && (((uintptr_t)a->qkv) & 15) == 0;
    if (!vt) return (long long)a->B * a->T * (a->Hq + 2 * a->Hkv);
    const long long ipw = rope_qk_vec(*a) ? 64 / (a->D >> 4) : 1;
    const long long qk_waves = (a->rope_mode == 0 && !a->k_cache) ? 0 : ((long long)a->B * a->T * (a->Hq + a->Hkv) + ipw - 1) / ipw;
    return qk_waves + (long long)a->B * a->Hkv * (a->D >> 3) * ((a->T + 63) >> 6);
}
hipError_t launch_rope_kv_write_pair(const cover_rope_args* a0, const cover_rope_args* a1, hipStream_t st) {
    const long long w0 = rope_waves_host(a0), w1 = rope_waves_host(a1);
    if (w0 <= 0) return launch_rope_kv_write(a1, st);
    if (w1 <= 0) return launch_rope_kv_write(a0, st);
    if (!rope_args_ok(a0) || !rope_args_ok(a1)) return hipErrorInvalidValue;
    const int blocks = (int)((w0 + w1 + 3) / 4);
    hipLaunchKernelGGL(rope_kv_write2_k, dim3(blocks), dim3(256), 0, st, *a0, *a1, (int)w0);
    return hipGetLastError();
}
hipError_t launch_rope_kv_write(const cover_rope_args* a, hipStream_t st) {
    const long long waves = rope_waves_host(a);
    if (waves <= 0) return hipSuccess;
    if (!a->vt_cache || (a->D & 1)) return hipErrorInvalidValue;
    if (a->rope_mode != 0 && (!a->cos_table || !a->sin_table)) return hipErrorInvalidValue;
    const int blocks = (int)((waves + 3) / 4);
    hipLaunchKernelGGL(rope_kv_write_k, dim3(blocks), dim3(256), 0, st, *a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// gathers / copies / casts
// ---------------------------------------------------------------------------------------------------
__global__ void embed_gather_k(const bf16_t* __restrict__ table, int dim, const int64_t* __restrict__ ids, float scale,
                               bf16_t* __restrict__ out, int ldo) {
    const int i = blockIdx.x;
    const bf16_t* src = table + (size_t)ids[i] * dim;
    for (int c = threadIdx.x; c < (dim >> 3); c += blockDim.x) {
        float v[8];
        load_row8<false>(src, (size_t)c * 8, v);
        if (scale != 1.0f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= scale;
        }
        store_row8(out, (size_t)i * ldo + c * 8, v);
    }
}
hipError_t launch_embed_gather(const bf16_t* table, int dim, const int64_t* ids, int n, float scale, bf16_t* out, int ldo,
                               hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (dim % 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embed_gather_k, dim3(n), dim3(128), 0, st, table, dim, ids, scale, out, ldo);
    return hipGetLastError();
}

__global__ void patchify_k(cover_patchify_args a) {
    const int gw = a.W / a.patch, gh = a.H / a.patch;
    const int np = gw * gh;
    const int row = blockIdx.x;  // img * np + p
    const int img = row / np, p = row - img * np;
    const int py0 = (p / gw) * a.patch, px0 = (p % gw) * a.patch;
    const int kk = 3 * a.patch * a.patch;
    bf16_t* o = (bf16_t*)a.out + (size_t)row * a.ld_out;
    for (int k = threadIdx.x; k < a.ld_out; k += blockDim.x) {
        float v = 0.f;
        if (k < kk) {
            const int c = k / (a.patch * a.patch), rem = k - c * a.patch * a.patch;
            const int y = py0 + rem / a.patch, x = px0 + rem % a.patch;
            float pix;
            if (a.in_u8_hwc) pix = (float)((const uint8_t*)a.img)[(size_t)img * a.img_stride + ((size_t)y * a.W + x) * 3 + c];
            else pix = ((const float*)a.img)[(size_t)img * a.img_stride + ((size_t)c * a.H + y) * a.W + x];
            v = pix * a.mul[c] + a.add[c];
        }
        o[k] = f2bf(v);
    }
}
hipError_t launch_patchify(const cover_patchify_args* a, hipStream_t st) {
    const int np = (a->W / a->patch) * (a->H / a->patch);
    if (a->n_img <= 0) return hipSuccess;
    hipLaunchKernelGGL(patchify_k, dim3(a->n_img * np), dim3(256), 0, st, *a);
    return hipGetLastError();
}

__global__ void copy_rows_k(const bf16_t* __restrict__ src, int lds_, bf16_t* __restrict__ dst, int ldd, int cols,
                            const int* __restrict__ sidx, const int* __restrict__ didx) {
    const int i = blockIdx.x;
    const bf16_t* s = src + (size_t)(sidx ? sidx[i] : i) * lds_;
    bf16_t* d = dst + (size_t)(didx ? didx[i] : i) * ldd;
    if ((cols & 7) == 0 && ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {
        for (int c = threadIdx.x; c < (cols >> 3); c += blockDim.x) ((uint4*)d)[c] = ((const uint4*)s)[c];
    } else {
        for (int c = threadIdx.x; c < cols; c += blockDim.x) d[c] = s[c];
    }
}
hipError_t launch_copy_rows_bf16(const bf16_t* src, int lds_, bf16_t* dst, int ldd, int rows, int cols,
                                 const int* sidx, const int* didx, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(copy_rows_k, dim3(rows), dim3(128), 0, st, src, lds_, dst, ldd, cols, sidx, didx);
    return hipGetLastError();
}

__global__ void add_rows_k(bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ add, int ld_add, int cols,
                           int add_rows) {
    const int r = blockIdx.x;
    bf16_t* xr = x + (size_t)r * ldx;
    const bf16_t* ar = add + (size_t)(r % add_rows) * ld_add;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) xr[c] = f2bf(bf2f(xr[c]) + bf2f(ar[c]));
}
hipError_t launch_add_bias_rows_bf16(bf16_t* x, int ldx, const bf16_t* add, int ld_add, int rows, int cols, int add_rows,
                                     hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(add_rows_k, dim3(rows), dim3(256), 0, st, x, ldx, add, ld_add, cols, add_rows);
    return hipGetLastError();
}

__global__ void scale_bf16_k(bf16_t* __restrict__ x, int ldx, int cols, float pre_div, float post_mul) {
    bf16_t* xr = x + (size_t)blockIdx.x * ldx;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        float v = bf2f(xr[c]);
        if (pre_div != 1.0f) v = bfround(v / pre_div);
        xr[c] = f2bf(v * post_mul);
    }
}
hipError_t launch_scale_bf16(bf16_t* x, int ldx, int rows, int cols, float pre_div, float post_mul, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(scale_bf16_k, dim3(rows), dim3(256), 0, st, x, ldx, cols, pre_div, post_mul);
    return hipGetLastError();
}

__global__ void cast_f2b_k(const float* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy, int cols) {
    for (int c = threadIdx.x; c < cols; c += blockDim.x)
        y[(size_t)blockIdx.x * ldy + c] = f2bf(x[(size_t)blockIdx.x * ldx + c]);
}
__global__ void cast_b2f_k(const bf16_t* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int cols) {
    for (int c = threadIdx.x; c < cols; c += blockDim.x)
        y[(size_t)blockIdx.x * ldy + c] = bf2f(x[(size_t)blockIdx.x * ldx + c]);
}
hipError_t launch_cast_f32_to_bf16(const float* x, int ldx, bf16_t* y, int ldy, int rows, int cols, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(cast_f2b_k, dim3(rows), dim3(256), 0, st, x, ldx, y, ldy, cols);
    return hipGetLastError();
}
hipError_t launch_cast_bf16_to_f32(const bf16_t* x, int ldx, float* y, int ldy, int rows, int cols, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(cast_b2f_k, dim3(rows), dim3(256), 0, st, x, ldx, y, ldy, cols);
    return hipGetLastError();
}
