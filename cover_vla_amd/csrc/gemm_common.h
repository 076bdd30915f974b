// Shared device code of the GEMM kernels (gemm_bf16.hip, gemm_fp8.hip): the epilogue description, the per-group epilogue
// arithmetic with its bf16 rounding points, and the direct / LDS-staged tile epilogues.
#pragma once
#include <type_traits>
#include "common.h"
#include "kernels.h"

#define BK 64
#ifndef PCTL
#define PCTL(slot) do { } while (0)
#endif

struct EpiDev {
    const float* bias;
    const void* residual;
    const float* lscale;
    int ldr, res_f32, act, glu, out_f32;
    float out_scale;
    const float* norm_w;
    const float* norm_b;
    bf16_t* norm_out;
    int ld_norm_out, norm_style;
    float norm_w_offset, norm_eps;
    const uint8_t* w8;      // optional e4m3 twin of the weight (cover_pack_weight_fp8) + its packed-order per-channel scales:
    const float* w8s;       // read by the weight-streaming kernels (M <= 32) instead of the bf16 image, same results
    const uint8_t* a8;      // optional e4m3 twin of the activation rows + row scales (host side only: selects the fp8 tiled kernel)
    const float* a8s;
    int lda8;
    uint8_t* nq8;           // optional e4m3 twin of norm_out (+ row scales), written by the norm that writes norm_out
    float* nq8s;
    int ldnq8;
    // MX block scales (config 5, down_proj): w8_kl = 1: w8 is the k-linear e4m3 image (cover_pack_weight_fp8_klinear); a8mx: e8m0 block scales of a8
    // ([Kp / 128][M][4] bytes, one per 32 consecutive k; a8 is then PLAIN row-major e4m3 and a8s is not read); o8 / o8mx: a GLU GEMM writes its output
    // rows in that form (pitch ldo8 bytes) INSTEAD of bf16 C -- the next GEMM's operand without a quantiser launch
    int w8_kl;
    const uint8_t* a8mx;
    uint8_t* o8;
    uint8_t* o8mx;
    int ldo8;
};

// e4m3 quantisation of 8 consecutive bf16-valued elements k..k+7 of an activation row into the MX MFMA operand order
// (cover_quantize_act_fp8): position 64 (k / 64) + 16 ((k / 8) % 4) + 8 ((k / 32) % 2)
__device__ __forceinline__ void store_q8_chunk(uint8_t* row, int k, const float (&v)[8], float inv) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
    *(uint2*)(row + (k >> 6) * 64 + ((k >> 3) & 3) * 16 + ((k >> 5) & 1) * 8) = make_uint2((uint32_t)lo, (uint32_t)hi);
}
// smallest power of two s with mx / s <= 448 (the e4m3 maximum); 1 for an all-zero row
__device__ __forceinline__ float e4m3_pow2_scale(float mx) {
    float s = 1.0f;
    if (mx > 0.f) {
        int e;
        const float f = frexpf(mx / 448.0f, &e);
        s = ldexpf(1.0f, f == 0.5f ? e - 1 : e);
    }
    return s;
}
// MX block quantisation (cover_quantize_act_fp8_mx's arithmetic) of 8 consecutive bf16-valued elements held by this thread; the four lanes 4q .. 4q + 3
// of the wave hold the four chunks of ONE 32-wide block and must all be active. s = smallest power of two >= 2^-126 with amax_block / s <= 448
// (2^0 for an all-zero block); returns its e8m0 byte (127 + log2 s) and the 8 e4m3 bytes RNE(v / s).
__device__ __forceinline__ uint32_t mx_quant_chunk(const float (&v)[8], uint2& q) {
    float mx = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(v[e]));
    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0xB1, 0xf, 0xf, true)));   // quad_perm [1,0,3,2]
    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x4E, 0xf, 0xf, true)));   // quad_perm [2,3,0,1]
    const float s = fmaxf(e4m3_pow2_scale(mx), 1.1754943508222875e-38f);
    const float inv = 1.0f / s;   // exact
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
    q = make_uint2((uint32_t)lo, (uint32_t)hi);
    return (__builtin_bit_cast(uint32_t, s) >> 23) & 0xffu;
}

// val[4] are 4 consecutive columns n0..n0+3 of row m: bias / activation / layer-scale / residual / scale, in place.
__device__ __forceinline__ void epi_value4(const EpiDev& e, int m, int n0, int N, float v[4]) {
    const bool full = (n0 + 3 < N);
    if (full) {
        // Whole group of four columns: every operand load is unconditional and issued before the first use. With the
        // per-element `n0 + i < N` guards below, each bias / layer-scale / residual element became its own branch + load +
        // s_waitcnt vmcnt(0): up to 12 dependent L2 round trips per group in the epilogue of every GEMM and reduction.
        float b[4] = {0.f, 0.f, 0.f, 0.f}, ls[4] = {1.f, 1.f, 1.f, 1.f}, r[4] = {0.f, 0.f, 0.f, 0.f};
        if (e.bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = e.bias[n0 + i];
        }
        if (e.lscale) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ls[i] = e.lscale[n0 + i];
        }
        if (e.residual) {
            if (e.res_f32) {
                const float* rp = (const float*)e.residual + (size_t)m * e.ldr + n0;
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = rp[i];
            } else {
                const bf16_t* rp = (const bf16_t*)e.residual + (size_t)m * e.ldr + n0;
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = bf2f(rp[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // the arithmetic and its rounding points are those of the generic path below
            float x = v[i];
            if (e.bias) x += b[i];
            x = bfround(x);
            if (e.act != ACT_NONE) x = bfround(act_apply_bf16(x, e.act));
            if (e.lscale) x = bfround(x * ls[i]);
            if (e.residual) x = x + r[i];
            if (e.out_scale != 1.0f) x *= e.out_scale;
            v[i] = x;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float x = v[i];
        if (e.bias && (full || n0 + i < N)) x += e.bias[n0 + i];
        x = bfround(x);                       // nn.Linear output rounding point
        if (e.act != ACT_NONE) x = bfround(act_apply_bf16(x, e.act));
        v[i] = x;
    }
    if (e.lscale) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (full || n0 + i < N) v[i] = bfround(v[i] * e.lscale[n0 + i]);
    }
    if (e.residual) {
        if (e.res_f32) {
            const float* r = (const float*)e.residual + (size_t)m * e.ldr + n0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (full || n0 + i < N) v[i] = v[i] + r[i];
        } else {
            const bf16_t* r = (const bf16_t*)e.residual + (size_t)m * e.ldr + n0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (full || n0 + i < N) v[i] = v[i] + bf2f(r[i]);
        }
    }
    if (e.out_scale != 1.0f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= e.out_scale;
    }
}
__device__ __forceinline__ bool epi_is_plain(const EpiDev& e) {
    return !e.bias && !e.residual && !e.lscale && e.act == ACT_NONE && e.out_scale == 1.0f;
}
// the store half of epi_store4 (values already through epi_value4)
__device__ __forceinline__ void epi_put4(const EpiDev& e, void* C, int ldc, int m, int n0, int N, const float v[4]) {
    if (n0 >= N) return;
    const bool full = (n0 + 3 < N);
    if (e.out_f32) {
        float* o = (float*)C + (size_t)m * ldc + n0;
        if (full && ((((uintptr_t)o) & 15) == 0)) {
            *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n0 + i < N) o[i] = v[i];
        }
    } else {
        bf16_t* o = (bf16_t*)C + (size_t)m * ldc + n0;
        if (full && ((((uintptr_t)o) & 7) == 0)) {
            uint2 p;
            p.x = pack_bf2(v[0], v[1]);
            p.y = pack_bf2(v[2], v[3]);
            *(uint2*)o = p;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n0 + i < N) o[i] = f2bf(v[i]);
        }
    }
}

__device__ __forceinline__ void epi_store4(const EpiDev& e, void* C, int ldc, int m, int n0, int N, float v[4]) {
    if (n0 >= N) return;
    const bool full = (n0 + 3 < N);
    epi_value4(e, m, n0, N, v);
    if (e.out_f32) {
        float* o = (float*)C + (size_t)m * ldc + n0;
        if (full && ((((uintptr_t)o) & 15) == 0)) {
            *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int i = 0; i < 4; ++i)
                if (n0 + i < N) o[i] = v[i];
        }
    } else {
        bf16_t* o = (bf16_t*)C + (size_t)m * ldc + n0;
        if (full && ((((uintptr_t)o) & 7) == 0)) {
            uint2 p;
            p.x = pack_bf2(v[0], v[1]);
            p.y = pack_bf2(v[2], v[3]);
            *(uint2*)o = p;
        } else {
            for (int i = 0; i < 4; ++i)
                if (n0 + i < N) o[i] = f2bf(v[i]);
        }
    }
}

// GLU epilogue: g[4] = gate columns, u[4] = matching up columns; output column j0 (N_out = N/2 columns).
__device__ __forceinline__ void epi_store4_glu(const EpiDev& e, void* C, int ldc, int m, int j0, int Nout, float g[4],
                                               float u[4]) {
    if (j0 >= Nout) return;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float gg = bfround(g[i]), uu = bfround(u[i]);
        gg = bfround(act_apply_bf16(gg, e.act));
        v[i] = bfround(gg * uu);
    }
    if (e.out_f32) {
        float* o = (float*)C + (size_t)m * ldc + j0;
        for (int i = 0; i < 4; ++i)
            if (j0 + i < Nout) o[i] = v[i];
    } else {
        bf16_t* o = (bf16_t*)C + (size_t)m * ldc + j0;
        if (j0 + 3 < Nout && ((((uintptr_t)o) & 7) == 0)) {
            uint2 p;
            p.x = pack_bf2(v[0], v[1]);
            p.y = pack_bf2(v[2], v[3]);
            *(uint2*)o = p;
        } else {
            for (int i = 0; i < 4; ++i)
                if (j0 + i < Nout) o[i] = f2bf(v[i]);
        }
    }
}

// Epilogue of one wave's (WM*16) x (WN*16) tile at (mw, nw): split-K partial slab, GLU, or bias/act/residual store.
// The m-fragment index is a template parameter: left as a loop the compiler keeps it rolled for WM = 4 and the
// accumulators end up in scratch (dynamic indexing).
template <int WM, int WN, int F>
__device__ __forceinline__ void tiled_epilogue_row(f32x4 (&acc)[WN][WM], const EpiDev& epi, void* C, int ldc, int M, int N, int mw,
                                                   int nw, int r, int g, float* __restrict__ partial) {
    const int m = mw + F * 16 + r;
    if (m < M) {
        if (partial) {  // split-K: raw fp32 partial sums, epilogue applied by splitk_reduce
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                const int n = nw + b * 16 + 4 * g;
                if (n < N) {
                    float* o = partial + ((size_t)blockIdx.y * M + m) * N + n;
                    if (n + 3 < N && ((((uintptr_t)o) & 15) == 0)) {
                        *(float4*)o = make_float4(acc[b][F][0], acc[b][F][1], acc[b][F][2], acc[b][F][3]);
                    } else {
                        const float v[4] = {acc[b][F][0], acc[b][F][1], acc[b][F][2], acc[b][F][3]};
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (n + i < N) o[i] = v[i];
                    }
                }
            }
        } else if (epi.glu) {
#pragma unroll
            for (int b = 0; b < WN; b += 2) {
                const int nblk = (nw >> 4) + b;  // even block = gate, odd = up
                float gv[4] = {acc[b][F][0], acc[b][F][1], acc[b][F][2], acc[b][F][3]};
                float uv[4] = {acc[b + 1][F][0], acc[b + 1][F][1], acc[b + 1][F][2], acc[b + 1][F][3]};
                epi_store4_glu(epi, C, ldc, m, (nblk >> 1) * 16 + 4 * g, N >> 1, gv, uv);
            }
        } else {
            // values of the whole row first, stores after: the operand loads (bias, layer scale, residual) of all WN groups
            // then share one round trip -- behind a store they cannot be hoisted (C may alias the residual)
            float vv[WN][4];
#pragma unroll
            for (int b = 0; b < WN; ++b) {
#pragma unroll
                for (int i = 0; i < 4; ++i) vv[b][i] = acc[b][F][i];
                const int n = nw + b * 16 + 4 * g;
                if (n < N) epi_value4(epi, m, n, N, vv[b]);
            }
#pragma unroll
            for (int b = 0; b < WN; ++b) epi_put4(epi, C, ldc, m, nw + b * 16 + 4 * g, N, vv[b]);
        }
    }
    if constexpr (F + 1 < WM) tiled_epilogue_row<WM, WN, F + 1>(acc, epi, C, ldc, M, N, mw, nw, r, g, partial);
}
template <int WM, int WN>
__device__ __forceinline__ void tiled_epilogue(f32x4 (&acc)[WN][WM], const EpiDev& epi, void* C, int ldc, int M, int N, int mw,
                                               int nw, int r, int g, float* __restrict__ partial) {
    tiled_epilogue_row<WM, WN, 0>(acc, epi, C, ldc, M, N, mw, nw, r, g, partial);
}

// ---- LDS-staged epilogue --------------------------------------------------------------------------------------------
// The direct epilogue stores 8 bytes per lane: one wave instruction = 16 rows x 32 contiguous bytes, a partial line per
// row, and with a power-of-two row pitch (N = 12288, 4096: 24 / 8 KiB) all 16 rows of an instruction sit in the SAME L2
// channel. Per-block timelines of the 224 x 128 kernel at M = 448 (tools/dbg/pc_timeline.py): main loop 46.6 us, epilogue 21.6 us
// -- a third of the kernel waiting for its own stores. Here the finished values (bias / activation / GLU / scale applied in
// registers, as before) go to the block's LDS tile -- the pipeline stages are dead by now -- and leave as 16-byte
// stores, a row of the tile contiguous: BN = 128 bf16 columns = 256 B = 16 lanes, four full rows per wave instruction.
// Raw fp32 split-K partials take the same path when the tile fits. Falls back to the direct form for ragged tiles.
// mode 0: raw fp32 accumulators (split-K partials, or the generic epilogue whose arithmetic runs in the store loop);
// mode 1: plain / GLU -- the accumulators rounded to bf16 (the nn.Linear output rounding point), nothing else.
template <int WM, int WN, int F>
__device__ __forceinline__ void staged_fill_row(f32x4 (&acc)[WN][WM], char* st, int pitch, int m0, int n0, int mw, int nw, int r, int g,
                                                int mode, int r0, int rpp) {
    // rows [r0, r0 + rpp) of the tile are staged in this pass (a 16-row fragment lies inside one pass: rpp % 16 == 0)
    const int lr = mw + F * 16 - m0 - r0;
    char* row = st + (size_t)(lr + r) * pitch;
    if (lr < 0 || lr >= rpp) {
    } else if (mode == 0) {
#pragma unroll
        for (int b = 0; b < WN; ++b)
            *(float4*)(row + (nw - n0 + b * 16 + 4 * g) * 4) = make_float4(acc[b][F][0], acc[b][F][1], acc[b][F][2], acc[b][F][3]);
    } else {
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            uint2 p;
            p.x = pack_bf2(acc[b][F][0], acc[b][F][1]);
            p.y = pack_bf2(acc[b][F][2], acc[b][F][3]);
            *(uint2*)(row + (nw - n0 + b * 16 + 4 * g) * 2) = p;
        }
    }
    if constexpr (F + 1 < WM) staged_fill_row<WM, WN, F + 1>(acc, st, pitch, m0, n0, mw, nw, r, g, mode, r0, rpp);
}

// BM x BN tile of the block at (m0, n0); the calling threads are `nthr` consecutive threads with index `t` (every one of them
// owns accumulators). `st` = the block's dynamic LDS (at least st_bytes large), free to overwrite.
template <int WM, int WN, int BM, int BN, bool MXO = false>   // MXO: the kernel can write the block-scaled e4m3 GLU output (epi.o8; gemm_fp8.hip)
__device__ __forceinline__ void tiled_epilogue_staged(f32x4 (&acc)[WN][WM], const EpiDev& epi, void* C, int ldc, int M, int N, int m0,
                                                      int n0, int mw, int nw, int r, int g, float* __restrict__ partial, char* st,
                                                      int st_bytes, int t, int nthr, bool owner = true) {
    // owner: this wave holds a tile's sums (false for the second wave of a k-split pair, gemm_v3.hip: it only helps with the store loop)
    const bool raw = partial != nullptr;
    const bool glu = !raw && epi.glu;
    const bool plain = !raw && !glu && epi_is_plain(epi);
    const bool generic = !raw && !glu && !plain;
    const int mode = (raw || generic) ? 0 : 1;                        // what the LDS tile holds: fp32 accumulators / bf16 values
    const int esz_out = (raw || epi.out_f32) ? 4 : 2;
    const int ocols = glu ? BN / 2 : BN;                               // output columns of the tile
    const int Nout = glu ? (N >> 1) : N;
    const int oc0 = glu ? (n0 >> 1) : n0;
    const int row_bytes = ocols * esz_out;                             // of the OUTPUT tile
    const int pitch = BN * (mode == 0 ? 4 : 2) + 16;                   // LDS row; +16 B: consecutive rows start 4 banks apart
    char* base = raw ? (char*)(partial + (size_t)blockIdx.y * M * N) : (char*)C;
    const size_t ld_bytes = (size_t)(raw ? N : ldc) * esz_out;
    const int cvalid = min(ocols, Nout - oc0);                         // ragged last column tile: whole 16-byte chunks only
    // rows per staging pass: the whole tile when it fits the (dead) pipeline stages, else half of it (a 224 x 192 tile of fp32 slab values is
    // 172 KiB: its direct, un-staged epilogue -- 16 rows x 16 bytes per store instruction -- took 14.7 us of a 53 us launch at M = 2 232)
    const int rpp = ((size_t)BM * pitch <= (size_t)st_bytes || (BM % 32) != 0) ? BM : BM / 2;
    const int npass = BM / rpp;
    const bool ok = (size_t)rpp * pitch <= (size_t)st_bytes && cvalid > 0 && ((cvalid * esz_out) & 15) == 0 && (ld_bytes & 15) == 0 &&
                    ((((uintptr_t)base) + (size_t)oc0 * esz_out) & 15) == 0 && !((glu || plain) && epi.out_f32);
    if (!ok) {   // uniform over the block
        if (epi.o8 && !raw) __builtin_trap();   // the block-scaled output exists in the staged GLU store loop only (the launcher checks its conditions)
        if (owner) tiled_epilogue<WM, WN>(acc, epi, C, ldc, M, N, mw, nw, r, g, partial);
        return;
    }
    const int cpr = (cvalid * esz_out) >> 4;                           // 16-byte chunks per output row
    const int total = rpp * cpr;
    const unsigned magic = 0xFFFFFFFFu / (unsigned)cpr + 1u;            // c / cpr == umulhi(c, magic) for every c < 2^32 / cpr (c < 16 K here) ...
    auto div_cpr = [&](int c) -> int { return cpr == 1 ? c : (int)__umulhi((unsigned)c, magic); };   // ... except cpr == 1, whose magic is 2^32
    for (int pass = 0; pass < npass; ++pass) {
    const int r0 = pass * rpp;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every wave is done reading the pipeline stages / the previous pass's tile
    PCTL(4);
    if (owner) staged_fill_row<WM, WN, 0>(acc, st, pitch, m0, n0, mw, nw, r, g, mode, r0, rpp);
    PCTL(5);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    PCTL(6);
    if (generic && !epi.out_f32) {
        // Generic epilogue, bf16 output, with the operand loads HOISTED: per-block timelines of the 64 x 64 tile at ViT sizes showed 4.0 of
        // the 4.5 us epilogue in this store loop -- bias / layer-scale / residual were read element by element BEHIND the LDS reads, two
        // dependent iterations per thread. Here the operands of two chunks (8 columns each) are requested up front as 16-byte loads, then
        // the staged sums are read and the SAME arithmetic (epi_value4's rounding points) runs on registers.
        const bool al = (!epi.bias || (((uintptr_t)epi.bias) & 15) == 0) && (!epi.lscale || (((uintptr_t)epi.lscale) & 15) == 0) && (n0 & 7) == 0 &&
                        (!epi.residual || ((((uintptr_t)epi.residual) & 15) == 0 && ((epi.ldr * (epi.res_f32 ? 4 : 2)) & 15) == 0));
        if (al && epi.residual && !epi.res_f32 && !epi.bias && !epi.lscale && epi.act == ACT_NONE && epi.out_scale == 1.0f) {
            // Residual-only epilogue (decoder o_proj / down without a split: x += A W^T in place), EIGHT chunks per thread in flight. The general
            // loop below keeps two: with the output aliasing the residual every iteration's loads wait behind the previous iteration's stores
            // (one vmcnt), so a 224-row tile was 5-10 dependent memory round trips -- 13.3 us of store loop in the pi0 prefix o_proj at
            // M = 2 232 on 256 threads, 7.3 us on 512 (profiles/r06_v3_schedule_ab.txt). Same arithmetic: bf16(sum) + residual, rounded once more.
            constexpr int NV = 8;
            for (int c0 = t; c0 < total; c0 += NV * nthr) {
                uint4 rb[NV];
                int rowi[NV], chi[NV], mi[NV];
                bool okv[NV];
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int c = c0 + u * nthr;
                    rowi[u] = div_cpr(c < total ? c : 0); chi[u] = (c < total ? c : 0) - rowi[u] * cpr; mi[u] = m0 + r0 + rowi[u];
                    okv[u] = c < total && mi[u] < M;
                    const int mm = okv[u] ? mi[u] : (M - 1);
                    rb[u] = *(const uint4*)((const bf16_t*)epi.residual + (size_t)mm * epi.ldr + n0 + chi[u] * 8);
                }
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    if (!okv[u]) continue;
                    const char* lrow = st + (size_t)rowi[u] * pitch;
                    const float4 a4 = *(const float4*)(lrow + chi[u] * 32), b4 = *(const float4*)(lrow + chi[u] * 32 + 16);
                    const float v[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
                    const uint32_t rw[4] = {rb[u].x, rb[u].y, rb[u].z, rb[u].w};
                    uint32_t ow[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        ow[i] = pack_bf2(bfround(v[2 * i]) + bf2f((bf16_t)(rw[i] & 0xffffu)), bfround(v[2 * i + 1]) + bf2f((bf16_t)(rw[i] >> 16)));
                    *(uint4*)(base + (size_t)mi[u] * ld_bytes + (size_t)oc0 * esz_out + chi[u] * 16) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                }
            }
            continue;
        }
        if (al) {
            constexpr int NU = 2;   // chunks per thread in flight (four were measured on the one-wave-per-SIMD kernels: the 13 us store loop of the pi0 prefix
            // o_proj -- 224 x 96 tiles with an in-place residual at M = 2 232 -- did not move, so it is not this loop's round trips; four spill the 224 x 192 tile)
            for (int c0 = t; c0 < total; c0 += NU * nthr) {
                int rowi[NU], chi[NU], mi[NU];
                bool ok[NU];
                float4 bb[NU][2] = {}, ll[NU][2] = {}, rf[NU][2] = {};
                uint4 rb[NU] = {};
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int c = c0 + u * nthr;
                    rowi[u] = div_cpr(c); chi[u] = c - rowi[u] * cpr; mi[u] = m0 + r0 + rowi[u];
                    ok[u] = c < total && mi[u] < M;
                    const int mm = ok[u] ? mi[u] : (M - 1), cc = ok[u] ? chi[u] : 0;        // clamped: loads stay unconditional
                    const int n = n0 + cc * 8;
                    if (epi.bias) { bb[u][0] = *(const float4*)(epi.bias + n); bb[u][1] = *(const float4*)(epi.bias + n + 4); }
                    if (epi.lscale) { ll[u][0] = *(const float4*)(epi.lscale + n); ll[u][1] = *(const float4*)(epi.lscale + n + 4); }
                    if (epi.residual) {
                        if (epi.res_f32) {
                            const float* rp = (const float*)epi.residual + (size_t)mm * epi.ldr + n;
                            rf[u][0] = *(const float4*)rp; rf[u][1] = *(const float4*)(rp + 4);
                        } else {
                            rb[u] = *(const uint4*)((const bf16_t*)epi.residual + (size_t)mm * epi.ldr + n);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    if (!ok[u]) continue;
                    const char* lrow = st + (size_t)rowi[u] * pitch;
                    const float4 a4 = *(const float4*)(lrow + chi[u] * 32), b4 = *(const float4*)(lrow + chi[u] * 32 + 16);
                    float v[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
                    const float bv[8] = {bb[u][0].x, bb[u][0].y, bb[u][0].z, bb[u][0].w, bb[u][1].x, bb[u][1].y, bb[u][1].z, bb[u][1].w};
                    const float lv[8] = {ll[u][0].x, ll[u][0].y, ll[u][0].z, ll[u][0].w, ll[u][1].x, ll[u][1].y, ll[u][1].z, ll[u][1].w};
                    float rv[8];
                    if (epi.residual) {
                        if (epi.res_f32) {
                            rv[0] = rf[u][0].x; rv[1] = rf[u][0].y; rv[2] = rf[u][0].z; rv[3] = rf[u][0].w;
                            rv[4] = rf[u][1].x; rv[5] = rf[u][1].y; rv[6] = rf[u][1].z; rv[7] = rf[u][1].w;
                        } else {
                            const uint32_t rw[4] = {rb[u].x, rb[u].y, rb[u].z, rb[u].w};
#pragma unroll
                            for (int i = 0; i < 4; ++i) { rv[2 * i] = bf2f((bf16_t)(rw[i] & 0xffffu)); rv[2 * i + 1] = bf2f((bf16_t)(rw[i] >> 16)); }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {   // epi_value4's arithmetic and rounding points
                        float x = v[i];
                        if (epi.bias) x += bv[i];
                        x = bfround(x);
                        if (epi.act != ACT_NONE) x = bfround(act_apply_bf16(x, epi.act));
                        if (epi.lscale) x = bfround(x * lv[i]);
                        if (epi.residual) x = x + rv[i];
                        if (epi.out_scale != 1.0f) x *= epi.out_scale;
                        v[i] = x;
                    }
                    char* dst = base + (size_t)mi[u] * ld_bytes + (size_t)oc0 * esz_out + chi[u] * 16;
                    *(uint4*)dst = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
                }
            }
            continue;
        }
    }
    if (glu && (epi.act == ACT_SILU || epi.act == ACT_GELU_TANH)) {
        // GLU store loop with the activation selected ONCE (the per-element switch of act_apply inside an 8-wide body was a scalar branch per
        // element): output columns 8 ch .. 8 ch + 7 = pair ch / 2, half ch % 2: gate at tile column 32 (ch / 2) + 8 (ch % 2), up 16 further.
        // epi_store4_glu's arithmetic on the already bf16-rounded gate / up values.
        auto body = [&](auto ACT, auto MXO_) {
            constexpr int act = decltype(ACT)::value;
            constexpr bool mxo = decltype(MXO_)::value;   // e4m3 + MX block scales instead of bf16 (epi.o8): see cover_gemm_epi.out8
            for (int c = t; c < total; c += nthr) {
                const int row = div_cpr(c), ch = c - row * cpr;
                const int m = m0 + r0 + row;
                if (m >= M) continue;
                const char* gp = st + (size_t)row * pitch + ((ch >> 1) * 32 + (ch & 1) * 8) * 2;
                const uint4 gq = *(const uint4*)gp, uq = *(const uint4*)(gp + 32);
                const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w}, uw[4] = {uq.x, uq.y, uq.z, uq.w};
                uint32_t ow[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float g0 = bfround(act_apply_bf16(bf2f((bf16_t)(gw[i] & 0xffffu)), act)), g1 = bfround(act_apply_bf16(bf2f((bf16_t)(gw[i] >> 16)), act));
                    ow[i] = pack_bf2(g0 * bf2f((bf16_t)(uw[i] & 0xffffu)), g1 * bf2f((bf16_t)(uw[i] >> 16)));
                }
                if constexpr (!mxo) {
                    *(uint4*)(base + (size_t)m * ld_bytes + (size_t)oc0 * esz_out + ch * 16) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                } else {
                    // the bf16 values that would have been stored, block-quantised: chunk ch of the row sits in lane % 4 == ch % 4 (cpr % 4 == 0, nthr % 4 == 0),
                    // so a quad holds one 32-wide block; rows and the loop bound are uniform over a quad
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { v[2 * i] = bf2f((bf16_t)(ow[i] & 0xffffu)); v[2 * i + 1] = bf2f((bf16_t)(ow[i] >> 16)); }
                    uint2 q;
                    const uint32_t sb = mx_quant_chunk(v, q);
                    const int col = oc0 + ch * 8;
                    *(uint2*)(epi.o8 + (size_t)m * epi.ldo8 + col) = q;
                    if ((ch & 3) == 0) epi.o8mx[((size_t)(col >> 7) * M + m) * 4 + ((col >> 5) & 3)] = (uint8_t)sb;
                }
            }
        };
        if (epi.o8) {
            if constexpr (MXO) {
                if (epi.act == ACT_SILU) body(std::integral_constant<int, ACT_SILU>{}, std::true_type{});
                else body(std::integral_constant<int, ACT_GELU_TANH>{}, std::true_type{});
            } else {
                __builtin_trap();   // (the launcher sends such a GEMM to a kernel instantiated with MXO)
            }
        } else {
            if (epi.act == ACT_SILU) body(std::integral_constant<int, ACT_SILU>{}, std::false_type{});
            else body(std::integral_constant<int, ACT_GELU_TANH>{}, std::false_type{});
        }
        continue;
    }
    for (int c = t; c < total; c += nthr) {
        const int row = div_cpr(c), ch = c - row * cpr;
        const int m = m0 + r0 + row;
        if (m >= M) continue;
        const char* lrow = st + (size_t)row * pitch;
        char* dst = base + (size_t)m * ld_bytes + (size_t)oc0 * esz_out + ch * 16;
        if (raw || plain) {
            *(uint4*)dst = *(const uint4*)(lrow + ch * 16);
        } else if (glu) {   // output columns 8 ch .. 8 ch + 7 = pair ch / 2, half ch % 2: gate at tile column 32 (ch / 2) + 8 (ch % 2), up 16 further
            const char* gp = lrow + ((ch >> 1) * 32 + (ch & 1) * 8) * 2;
            const uint4 gq = *(const uint4*)gp, uq = *(const uint4*)(gp + 32);
            const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w}, uw[4] = {uq.x, uq.y, uq.z, uq.w};
            uint32_t ow[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // epi_store4_glu's arithmetic on the already bf16-rounded gate / up values
                const float g0 = bfround(act_apply_bf16(bf2f((bf16_t)(gw[i] & 0xffffu)), epi.act)), g1 = bfround(act_apply_bf16(bf2f((bf16_t)(gw[i] >> 16)), epi.act));
                ow[i] = pack_bf2(g0 * bf2f((bf16_t)(uw[i] & 0xffffu)), g1 * bf2f((bf16_t)(uw[i] >> 16)));
            }
            *(uint4*)dst = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        } else {            // generic epilogue on the staged fp32 sums, ONE copy of the arithmetic (a loop, not 14-28 inlined copies:
                            // inlined per fragment it made the epilogue 65 000 lines of ISA and a block spent 18 us fetching it); the
                            // bias / layer-scale / residual operands are read row-contiguous here
            if (epi.out_f32) {       // 4 fp32 outputs per chunk
                const float4 a4 = *(const float4*)(lrow + ch * 16);
                float v[4] = {a4.x, a4.y, a4.z, a4.w};
                epi_value4(epi, m, n0 + ch * 4, N, v);
                *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
            } else {                 // 8 bf16 outputs per chunk
                const float4 a4 = *(const float4*)(lrow + ch * 32), b4 = *(const float4*)(lrow + ch * 32 + 16);
                float v[4] = {a4.x, a4.y, a4.z, a4.w}, w[4] = {b4.x, b4.y, b4.z, b4.w};
                epi_value4(epi, m, n0 + ch * 8, N, v);
                epi_value4(epi, m, n0 + ch * 8 + 4, N, w);
                *(uint4*)dst = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(w[0], w[1]), pack_bf2(w[2], w[3]));
            }
        }
    }
    }   // passes
}

// LDS-DMA of 16 B per lane with a scalar base: LDS destination = M0 (wave-uniform) + lane * 16, global source = sbase + voff (per lane)
__device__ __forceinline__ void glds16_s(uint32_t voff, const void* sbase, uint32_t lds_wave_base_u32) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_wave_base_u32)
        : "memory");
}

// the 4-byte form: LDS destination = M0 + lane * 4
__device__ __forceinline__ void glds4_s(uint32_t voff, const void* sbase, uint32_t lds_wave_base_u32) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_wave_base_u32)
        : "memory");
}

// gemm_fp8.hip: the LDS-tiled GEMM on the MX-scaled fp8 matrix instruction, for the tile configurations listed there
bool gemm_fp8_tiled_supported(int pick);
hipError_t launch_gemm_fp8_tiled(int pick, const uint8_t* A8, int lda8, const float* a_scale, const uint8_t* W8, const float* w_scale, void* C, int ldc,
                                 int M, int N, int Kp, const EpiDev& epi, int tiles_m, int tiles_n, int kt_per, int S, float* partial, size_t lds,
                                 int prof_cls, double prof_work, hipStream_t st);

// gemm_v3.hip: the self-loading 8-wave tiled bf16 GEMM (picks 23..26 of launch_gemm_bf16's tile table)
hipError_t launch_gemm_v3(int pick, const bf16_t* A, int lda, const bf16_t* Wp, void* C, int ldc, int M, int N, int Kp, const EpiDev& epi, int tiles_m,
                          int tiles_n, int kt_per, int S, float* partial, int prof_cls, double prof_work, hipStream_t st);
