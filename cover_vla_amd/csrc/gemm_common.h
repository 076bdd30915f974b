// Shared device code of the GEMM kernels (gemm_bf16.hip, gemm_fp8.hip): the epilogue description, the per-group epilogue
// arithmetic with its bf16 rounding points, and the direct / LDS-staged tile epilogues.
#pragma once
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
};

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
            if (e.act != ACT_NONE) x = bfround(act_apply(x, e.act));
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
        if (e.act != ACT_NONE) x = bfround(act_apply(x, e.act));
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
        gg = bfround(act_apply(gg, e.act));
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
                                                int mode) {
    char* row = st + (size_t)(mw + F * 16 + r - m0) * pitch;
    if (mode == 0) {
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
    if constexpr (F + 1 < WM) staged_fill_row<WM, WN, F + 1>(acc, st, pitch, m0, n0, mw, nw, r, g, mode);
}

// BM x BN tile of the block at (m0, n0); the calling threads are `nthr` consecutive threads with index `t` (every one of them
// owns accumulators). `st` = the block's dynamic LDS (at least st_bytes large), free to overwrite.
template <int WM, int WN, int BM, int BN>
__device__ __forceinline__ void tiled_epilogue_staged(f32x4 (&acc)[WN][WM], const EpiDev& epi, void* C, int ldc, int M, int N, int m0,
                                                      int n0, int mw, int nw, int r, int g, float* __restrict__ partial, char* st,
                                                      int st_bytes, int t, int nthr) {
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
    const bool ok = (size_t)BM * pitch <= (size_t)st_bytes && cvalid > 0 && ((cvalid * esz_out) & 15) == 0 && (ld_bytes & 15) == 0 &&
                    ((((uintptr_t)base) + (size_t)oc0 * esz_out) & 15) == 0 && !((glu || plain) && epi.out_f32);
    if (!ok) {   // uniform over the block
        tiled_epilogue<WM, WN>(acc, epi, C, ldc, M, N, mw, nw, r, g, partial);
        return;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every wave is done reading the pipeline stages
    PCTL(4);
    staged_fill_row<WM, WN, 0>(acc, st, pitch, m0, n0, mw, nw, r, g, mode);
    PCTL(5);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    PCTL(6);
    const int cpr = (cvalid * esz_out) >> 4;                           // 16-byte chunks per output row
    const int total = BM * cpr;
    for (int c = t; c < total; c += nthr) {
        const int row = c / cpr, ch = c - row * cpr;
        const int m = m0 + row;
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
                const float g0 = bfround(act_apply(bf2f((bf16_t)(gw[i] & 0xffffu)), epi.act)), g1 = bfround(act_apply(bf2f((bf16_t)(gw[i] >> 16)), epi.act));
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
}

// gemm_fp8.hip: the LDS-tiled GEMM on the MX-scaled fp8 matrix instruction, for the tile configurations listed there
bool gemm_fp8_tiled_supported(int pick);
hipError_t launch_gemm_fp8_tiled(int pick, const uint8_t* A8, int lda8, const float* a_scale, const uint8_t* W8, const float* w_scale, void* C, int ldc,
                                 int M, int N, int Kp, const EpiDev& epi, int tiles_m, int tiles_n, int kt_per, int S, float* partial, size_t lds,
                                 int prof_cls, double prof_work, hipStream_t st);
