// fp8 (OCP e4m3) GEMM on the CDNA4 block-scaled matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4 (BASELINE config 5:
// "fp8 weights + fp8 KV (CDNA4 fp8 MFMA)"; the reference has no fp8 path -- SURVEY.md 7 step 9 -- so the bar is agreement rate /
// score RMSE against the bf16 pipeline, plus kernel-level parity against an fp32 restatement on the SAME quantised operands).
//
//   C[M,N] = epi( a_scale[m] * w_scale[n] * sum_k A8[m,k] * W8[n,k] )        fp32 accumulate, 128 k per instruction
//
// Non-scaled fp8 MFMA runs at the bf16 rate on gfx950; only the MX-scaled form reaches the 2x rate (MI355X_MICROARCH.md,
// matrix-core table). It is used here with every E8M0 block scale = 2^0: the quantisation scales are per ROW of A (dynamic,
// cover_quantize_act_fp8) and per OUTPUT CHANNEL of W (static, cover_quantize_rows_fp8), both powers of two, and are applied to the
// fp32 sums in the epilogue -- they are constant along k, so this is exact.
//
// Operand images. W8 is the e4m3 image the weight-streaming kernels already read (cover_pack_weight_fp8):
//   Wq[n/16][k/64][lane = n%16 + 16*g][16 B] with the lane's bytes = k = 64c + {32h + 8g + e : h = 0,1; e = 0..7}  (c = k/64)
// i.e. one 1-KiB block per (16 n, 64 k). The 32-byte MFMA operand of lane (n%16, g) for a 128-deep step is its 16 bytes of block
// c = 0 followed by its 16 bytes of block c = 1. The contraction pairs byte j of lane (.., g) of one operand with byte j of lane
// (.., g) of the other, so ANY k order works as long as both operands use the same one: cover_quantize_act_fp8 writes the
// activation rows in exactly that order (inside every 64-block, position g*16 + h*8 + e holds k = 32h + 8g + e). A row of a
// 128-deep k-tile is then 128 B = 8 chunks of 16 B, lane (m%16, g) needs chunks g (c = 0) and 4 + g (c = 1): the same LDS image,
// XOR chunk swizzle and read pattern as the bf16 kernel (gemm_tiled_pc), with half the bytes per FLOP.
//
// Structure = gemm_tiled_pc: NL loader waves own every LDS-DMA piece (global_load_lds, counted vmcnt ring, NST stages), CGM x CGN
// MFMA waves of (WM*16) x (WN*16), one barrier per k-tile. The MFMA waves refill their fragment registers IN PLACE with the next
// tile's data as soon as a fragment's last MFMA of the current tile has issued (see phase0 / mid_phases / last_phase below): the
// LDS latency hides behind the MFMAs without a second fragment set (which would not fit the register budget of a 12-wave block).
// (First version: all fragment reads right after the barrier, then the MFMAs behind counted waits -- qkv at M = 512 1478 TFLOP/s.)
#include <stdlib.h>
#include <type_traits>
#include <hip/hip_ext.h>
#include "gemm_common.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;

// inline-asm LDS reads / counted waits with compile-time immediates, and the template recursions that replace loops over them
// (an asm operand cannot name a lambda capture, and an "n" operand must be a constant expression)
template <int OFF>
__device__ __forceinline__ void ds_read128(u32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {   // lgkmcnt is a 4-bit field: a larger count is clamped (a stricter wait is always safe)
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N > 15 ? 15 : N));
}
template <int F, int WM>
__device__ __forceinline__ void read_x_frags(u32x4 (&xlo)[WM], u32x4 (&xhi)[WM], uint32_t a0, uint32_t a1) {
    if constexpr (F < WM) {
        ds_read128<F * 2048>(xlo[F], a0);
        ds_read128<F * 2048>(xhi[F], a1);
        read_x_frags<F + 1, WM>(xlo, xhi, a0, a1);
    }
}
template <int B, int WN>
__device__ __forceinline__ void read_w_frags(u32x4 (&wlo)[WN], u32x4 (&whi)[WN], uint32_t ba) {
    if constexpr (B < WN) {
        ds_read128<B * 2048>(wlo[B], ba);
        ds_read128<B * 2048 + 1024>(whi[B], ba);
        read_w_frags<B + 1, WN>(wlo, whi, ba);
    }
}
template <int B, int BEND, int WN>
__device__ __forceinline__ void read_w_frags_range(u32x4 (&wlo)[WN], u32x4 (&whi)[WN], uint32_t ba) {
    if constexpr (B < BEND) {
        ds_read128<B * 2048>(wlo[B], ba);
        ds_read128<B * 2048 + 1024>(whi[B], ba);
        read_w_frags_range<B + 1, BEND, WN>(wlo, whi, ba);
    }
}
// sx: the E8M0 block scale of the activation operand in byte 0 (every lane: the scale of ITS 32 operand bytes); 2^0 without MX block scales
__device__ __forceinline__ f32x4 mfma_f8(const u32x4& wl, const u32x4& wh, const u32x4& xl, const u32x4& xh, f32x4 c, int sx = 0x7f7f7f7f) {
    const i32x8 wa = {(int)wl[0], (int)wl[1], (int)wl[2], (int)wl[3], (int)wh[0], (int)wh[1], (int)wh[2], (int)wh[3]};
    const i32x8 xa = {(int)xl[0], (int)xl[1], (int)xl[2], (int)xl[3], (int)xh[0], (int)xh[1], (int)xh[2], (int)xh[3]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xa, c, 0, 0, 0, 0x7f7f7f7f, 0, sx);   // weight block scales 2^0 (per-channel scales in the epilogue)
}
// activation block scales of the current k-tile, per m-fragment: none (per-row scales, applied in the epilogue) ...
struct NoScale {
    __device__ __forceinline__ void prepare() const {}
    __device__ __forceinline__ int operator()(int) const { return 0x7f7f7f7f; }
};
// ... or MX: raw[f] = the dword [k-tile][row] of four E8M0 bytes (k-blocks g = 0..3 of the tile) that this lane's ds_read_b32 of fragment f fetched for the
// NEXT tile; prepare() -- called behind a wait that covers those reads -- moves the lane's own byte (g = lane / 16) down for the MFMAs of that tile
template <int WM>
struct MxScale {
    uint32_t raw[WM];
    int s[WM];
    int sh;
    __device__ __forceinline__ void prepare() {
#pragma unroll
        for (int f = 0; f < WM; ++f) {
            asm volatile("" : "+v"(raw[f]));
            s[f] = (int)((raw[f] >> sh) & 0xffu);
        }
    }
    __device__ __forceinline__ int operator()(int f) const { return s[f]; }
};
template <int F, int WM>
__device__ __forceinline__ void read_mx_scales(MxScale<WM>& sc, uint32_t addr) {
    if constexpr (F < WM) {
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(sc.raw[F]) : "v"(addr), "n"(F * 64));
        read_mx_scales<F + 1, WM>(sc, addr);
    }
}

// ---- software-pipelined consumer: fragments are refilled IN PLACE, as soon as their last MFMA of the tile has issued -------------
// Steady state of tile kt (registers hold every fragment of tile kt; the reads of x[0..WM) and w[WN-1] are the youngest in flight):
//   phase 0        MFMA(w[0], x[f]) behind counted waits (x[f] and everything older landed)
//   lgkmcnt(0), s_barrier          tile kt+1 published; every fragment of tile kt is in registers, so its stage may be refilled
//   read w'[0]                     (next tile, into w[0]'s registers: dead after phase 0)
//   phase b = 1 .. WN-2            MFMA(w[b], x[*]); read w'[b]
//   phase WN-1     MFMA(w[WN-1], x[f]); read x'[f] after each; read w'[WN-1] at the end
// so the LDS latency of tile kt+1's fragments hides behind tile kt's MFMAs and no second fragment set is needed (a full second
// set does not fit the 168 registers of a 12-wave block). Issue order of the next tile's reads: w'[0..WN-2], x'[0..WM), w'[WN-1]
// (two reads each: the k-halves c = 0, 1), which is what the waits of the next phase 0 count.
template <int F, int WM, int WN, typename SC>
__device__ __forceinline__ void phase0(f32x4 (&acc)[WN][WM], u32x4 (&xlo)[WM], u32x4 (&xhi)[WM], u32x4 (&wlo)[WN], u32x4 (&whi)[WN], SC& sc) {
    if constexpr (F < WM) {
        wait_lgkm<2 * (WM - 1 - F) + 2>();                       // younger reads allowed in flight: x[F+1..WM) and w[WN-1]
        asm volatile("" : "+v"(wlo[0]), "+v"(whi[0]), "+v"(xlo[F]), "+v"(xhi[F]));
        if constexpr (F == 0) sc.prepare();                      // (the tile's block-scale reads are older than every fragment read)
        __builtin_amdgcn_sched_barrier(0);
        acc[0][F] = mfma_f8(wlo[0], whi[0], xlo[F], xhi[F], acc[0][F], sc(F));
        phase0<F + 1, WM, WN>(acc, xlo, xhi, wlo, whi, sc);
    }
}
template <int B, int WM, int WN, bool REFILL, typename SC>
__device__ __forceinline__ void mid_phases(f32x4 (&acc)[WN][WM], u32x4 (&xlo)[WM], u32x4 (&xhi)[WM], u32x4 (&wlo)[WN], u32x4 (&whi)[WN], uint32_t ba, SC& sc) {
    if constexpr (B < WN - 1) {
#pragma unroll
        for (int f = 0; f < WM; ++f) acc[B][f] = mfma_f8(wlo[B], whi[B], xlo[f], xhi[f], acc[B][f], sc(f));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (REFILL) {
            ds_read128<B * 2048>(wlo[B], ba);
            ds_read128<B * 2048 + 1024>(whi[B], ba);
            __builtin_amdgcn_sched_barrier(0);
        }
        mid_phases<B + 1, WM, WN, REFILL>(acc, xlo, xhi, wlo, whi, ba, sc);
    }
}
template <int F, int WM, int WN, bool REFILL, typename SC>
__device__ __forceinline__ void last_phase(f32x4 (&acc)[WN][WM], u32x4 (&xlo)[WM], u32x4 (&xhi)[WM], u32x4 (&wlo)[WN], u32x4 (&whi)[WN], uint32_t a0, uint32_t a1,
                                           uint32_t ba, SC& sc) {
    if constexpr (F < WM) {
        acc[WN - 1][F] = mfma_f8(wlo[WN - 1], whi[WN - 1], xlo[F], xhi[F], acc[WN - 1][F], sc(F));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (REFILL) {
            ds_read128<F * 2048>(xlo[F], a0);
            ds_read128<F * 2048>(xhi[F], a1);
            __builtin_amdgcn_sched_barrier(0);
        }
        last_phase<F + 1, WM, WN, REFILL>(acc, xlo, xhi, wlo, whi, a0, a1, ba, sc);
    } else if constexpr (REFILL) {
        ds_read128<(WN - 1) * 2048>(wlo[WN - 1], ba);
        ds_read128<(WN - 1) * 2048 + 1024>(whi[WN - 1], ba);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// MXA: MX block-scaled activations (cover_gemm_epi.a8_mx; see gemm_tiled_v3_f8 below for the operand order): tiles of at most 64 rows, the tile's 64 scale
// dwords of a k-tile are one more piece of EVERY loader wave (the same 256 bytes to the same place: every wave's counted vmcnt stays uniform)
template <int WM, int WN, int NST, int NL, int CGM, int CGN, bool MXA = false>
__global__ __launch_bounds__(64 * (CGM * CGN + NL)) void gemm_tiled_pc_f8(const uint8_t* __restrict__ A8, int lda8, const uint8_t* __restrict__ W8,
                                                                          void* C, int ldc, int M, int N, int Kp, EpiDev epi, int tiles_m, int tiles_n,
                                                                          int kt_per, float* __restrict__ partial, const float* __restrict__ a_scale,
                                                                          const float* __restrict__ w_scale) {
    constexpr int NCW = CGM * CGN;
    constexpr int BM_ = CGM * WM * 16, BN_ = CGN * WN * 16;
    constexpr int A_BYTES = BM_ * 128, B_BYTES = BN_ * 128;   // one 128-deep k-tile: 128 B per row, as a 64-deep bf16 tile
    constexpr int AT = A_BYTES / 1024, BT = B_BYTES / 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                   // [NST][A_BYTES]
    char* Bs = smem + NST * A_BYTES;   // [NST][B_BYTES]
    char* Ss = smem + NST * (A_BYTES + B_BYTES);   // MXA: [NST][64 rows x 4 B]
    static_assert(!MXA || BM_ <= 64, "one block-scale dword per row, one piece per tile");
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: the row tiles that share a weight tile run on one XCD (one L2)
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int tn = bid / tiles_m, tm = bid % tiles_m;
    const int m0 = tm * BM_, n0 = tn * BN_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K64 = Kp >> 6;
    const int N16 = (N + 15) >> 4;
    const int nk_total = Kp >> 7;
    const int kt0 = blockIdx.y * kt_per;
    const int nk = min(kt_per, nk_total - kt0);

    if (w >= NCW) {   // ---------------- loader waves ----------------
        constexpr int PT = (AT + BT) / NL;
        constexpr int PTX = PT + (MXA ? 1 : 0);   // pieces a wave issues per tile (with the block-scale piece)
        static_assert((AT + BT) % NL == 0 && AT % NL == 0, "pieces must split evenly over the loader waves");
        static_assert((NST - 2) * PTX <= 63, "counted vmcnt must fit its 6-bit field");
        const int l = w - NCW;
        const uint8_t* src[PT];
        uint32_t dst[PT];
        size_t step[PT];
        const uint32_t as_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(As));
        const uint32_t bs_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(Bs));
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int j = l + i * NL;
            if (j < AT) {   // A: LDS chunk position p = j*64 + lane: row = p>>3, c = p&7 holds global chunk c ^ (row&7)
                const int row = j * 8 + (lane >> 3), c = lane & 7;
                int gr = m0 + row;
                gr = gr < M ? gr : M - 1;
                src[i] = A8 + (size_t)gr * lda8 + (size_t)kt0 * 128 + ((c ^ (row & 7)) << 4);
                dst[i] = as_u32 + j * 1024;
                step[i] = 128;
            } else {
                const int jb = j - AT;
                const int nbi = jb >> 1, kbi = jb & 1;
                int nb = (n0 >> 4) + nbi;
                nb = nb < N16 ? nb : N16 - 1;
                src[i] = W8 + ((size_t)nb * K64 + (size_t)kt0 * 2 + kbi) * 1024 + lane * 16;
                dst[i] = bs_u32 + jb * 1024;
                step[i] = 2048;
            }
        }
        const uint8_t* ssrc = nullptr;
        const uint32_t ss_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(Ss));
        if constexpr (MXA) {
            int gr = m0 + lane;
            gr = gr < M ? gr : M - 1;
            ssrc = epi.a8mx + ((size_t)kt0 * M + gr) * 4;
        }
        auto issue = [&](int buf, int kt) {
#pragma unroll
            for (int i = 0; i < PT; ++i)
                glds16_asm(src[i] + kt * step[i], dst[i] + buf * ((l + i * NL) < AT ? A_BYTES : B_BYTES));
            if constexpr (MXA) glds4_asm(ssrc + (size_t)kt * M * 4, ss_u32 + buf * 256);
        };
#pragma unroll
        for (int s = 0; s < NST - 1; ++s)
            if (s < nk) issue(s, s);
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int younger = min(nk - 1 - kt, NST - 2);   // tiles issued after kt that may stay in flight
            if (NST >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PTX) : "memory");
            else if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PTX) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + NST - 1 < nk) issue(cur == 0 ? NST - 1 : cur - 1, kt + NST - 1);   // stage (kt-1) % NST: every consumer is past tile kt-1
            cur = cur == NST - 1 ? 0 : cur + 1;
        }
        return;
    }
    // ---------------- MFMA waves ----------------
    const int wm = w / CGN, wn = w % CGN;
    const int r = lane & 15, g = lane >> 4;
    f32x4 acc[WN][WM];  // [n-block b][m-frag f]
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
        for (int f = 0; f < WM; ++f) acc[b][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const uint32_t a_addr0 = lds_addr_u32(As) + ((wm * (WM * 16) + r) * 8 + ((0 * 4 + g) ^ (r & 7))) * 16;
    const uint32_t a_addr1 = lds_addr_u32(As) + ((wm * (WM * 16) + r) * 8 + ((1 * 4 + g) ^ (r & 7))) * 16;
    const uint32_t b_addr = lds_addr_u32(Bs) + (wn * WN * 2 * 64 + lane) * 16;
    static_assert(WN >= 2, "the pipelined consumer refills w[0] while w[WN-1] is still needed");
    u32x4 xlo[WM], xhi[WM], wlo[WN], whi[WN];
    std::conditional_t<MXA, MxScale<WM>, NoScale> sc;   // NoScale: per-row activation scales, applied to the sums below
    const uint32_t s_addr = lds_addr_u32(Ss) + (wm * (WM * 16) + r) * 4;
    // prologue: tile 0 published; all of its fragments requested in the steady-state order w[0..WN-2], x[0..WM), w[WN-1]
    asm volatile("s_barrier" ::: "memory");
    if constexpr (MXA) { sc.sh = 8 * g; read_mx_scales<0, WM>(sc, s_addr); }
    {
        const uint32_t ba = b_addr;
        read_w_frags_range<0, WN - 1, WN>(wlo, whi, ba);
        read_x_frags<0, WM>(xlo, xhi, a_addr0, a_addr1);
        ds_read128<(WN - 1) * 2048>(wlo[WN - 1], ba);
        ds_read128<(WN - 1) * 2048 + 1024>(whi[WN - 1], ba);
    }
    int nxt = NST > 1 ? 1 : 0;                                    // stage of tile kt + 1
    for (int kt = 0; kt + 1 < nk; ++kt) {
        phase0<0, WM, WN>(acc, xlo, xhi, wlo, whi, sc);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // w[WN-1] landed too: stage kt is free; tile kt+1 is published
        asm volatile("" : "+v"(wlo[WN - 1]), "+v"(whi[WN - 1]));
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t a0 = a_addr0 + nxt * A_BYTES, a1 = a_addr1 + nxt * A_BYTES, ba = b_addr + nxt * B_BYTES;
        if constexpr (MXA) read_mx_scales<0, WM>(sc, s_addr + nxt * 256);   // tile kt + 1's block scales (this tile's were shifted down in phase 0)
        ds_read128<0>(wlo[0], ba);
        ds_read128<1024>(whi[0], ba);
        __builtin_amdgcn_sched_barrier(0);
        mid_phases<1, WM, WN, true>(acc, xlo, xhi, wlo, whi, ba, sc);
        last_phase<0, WM, WN, true>(acc, xlo, xhi, wlo, whi, a0, a1, ba, sc);
        nxt = nxt == NST - 1 ? 0 : nxt + 1;
    }
    {   // last tile: no refills
        phase0<0, WM, WN>(acc, xlo, xhi, wlo, whi, sc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(wlo[WN - 1]), "+v"(whi[WN - 1]));
        __builtin_amdgcn_sched_barrier(0);
        mid_phases<1, WM, WN, false>(acc, xlo, xhi, wlo, whi, 0u, sc);
        last_phase<0, WM, WN, false>(acc, xlo, xhi, wlo, whi, 0u, 0u, 0u, sc);
    }
    // quantisation scales (constant along k): row scale of A x channel scale of W, on the fp32 sums
    {
        const int mw = m0 + wm * (WM * 16), nw = n0 + wn * (WN * 16);
        float as[WM];
#pragma unroll
        for (int f = 0; f < WM; ++f) {
            int m = mw + f * 16 + r;
            m = m < M ? m : M - 1;
            as[f] = MXA ? 1.0f : a_scale[m];
        }
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            int nb = (nw >> 4) + b;
            nb = nb < N16 ? nb : N16 - 1;
            const float4 ws = *(const float4*)(w_scale + (size_t)nb * 16 + 4 * g);
#pragma unroll
            for (int f = 0; f < WM; ++f) {
                acc[b][f][0] *= as[f] * ws.x; acc[b][f][1] *= as[f] * ws.y; acc[b][f][2] *= as[f] * ws.z; acc[b][f][3] *= as[f] * ws.w;
            }
        }
    }
    tiled_epilogue_staged<WM, WN, BM_, BN_>(acc, epi, C, ldc, M, N, m0, n0, m0 + wm * (WM * 16), n0 + wn * (WN * 16), r, g, partial, smem,
                                            NST * (A_BYTES + B_BYTES), tid, 64 * NCW);
}

// ---------------------------------------------------------------------------------------------------
// Self-loading form of the kernel above (round 5; the fp8 twin of gemm_v3.hip): 8 waves, NO loader waves -- 256 registers per lane instead of the
// 168 of a 12-wave block -- every wave issues LDS-DMA pieces itself, spread between its MFMAs at compile-time slots: the first half of the waves
// owns the activation pieces, the second half the weight pieces (one DMA role per wave: a wave's vmcnt retires in order, and each role keeps
// its own ring depth NSTA / NSTB). The consumer side is the in-place-refill pipeline of gemm_tiled_pc_f8 unchanged (phase 0 | barrier |
// middle phases | last phase); the barrier behind phase 0 of tile kt publishes tile kt + 1 and releases the stage of tile kt, which the
// pieces issued during the middle and last phases refill with tile kt + NST. Every wave waits for ITS OWN pieces of tile kt + 1 with a counted
// vmcnt in front of that barrier; the surplus issues at the end of the K range re-load the last tile (clamped), so every count stays exact.
// ---------------------------------------------------------------------------------------------------
template <int V> using IC8 = std::integral_constant<int, V>;
// DMA slot s of a tile = behind the MFMAs of middle phase s + 1 (s < WN - 2), then behind MFMA s - (WN - 2) of the last phase
template <int B, int WM, int WN, bool REFILL, typename DMA, typename SC>
__device__ __forceinline__ void mid_phases_dma(f32x4 (&acc)[WN][WM], u32x4 (&xlo)[WM], u32x4 (&xhi)[WM], u32x4 (&wlo)[WN], u32x4 (&whi)[WN], uint32_t ba, DMA& dma,
                                               SC& sc) {
    if constexpr (B < WN - 1) {
#pragma unroll
        for (int f = 0; f < WM; ++f) acc[B][f] = mfma_f8(wlo[B], whi[B], xlo[f], xhi[f], acc[B][f], sc(f));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (REFILL) {
            ds_read128<B * 2048>(wlo[B], ba);
            ds_read128<B * 2048 + 1024>(whi[B], ba);
            dma(IC8<B - 1>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        mid_phases_dma<B + 1, WM, WN, REFILL>(acc, xlo, xhi, wlo, whi, ba, dma, sc);
    }
}
template <int F, int WM, int WN, bool REFILL, typename DMA, typename SC>
__device__ __forceinline__ void last_phase_dma(f32x4 (&acc)[WN][WM], u32x4 (&xlo)[WM], u32x4 (&xhi)[WM], u32x4 (&wlo)[WN], u32x4 (&whi)[WN], uint32_t a0, uint32_t a1,
                                               uint32_t ba, DMA& dma, SC& sc) {
    if constexpr (F < WM) {
        acc[WN - 1][F] = mfma_f8(wlo[WN - 1], whi[WN - 1], xlo[F], xhi[F], acc[WN - 1][F], sc(F));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (REFILL) {
            ds_read128<F * 2048>(xlo[F], a0);
            ds_read128<F * 2048>(xhi[F], a1);
            dma(IC8<WN - 2 + F>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        last_phase_dma<F + 1, WM, WN, REFILL>(acc, xlo, xhi, wlo, whi, a0, a1, ba, dma, sc);
    } else if constexpr (REFILL) {
        ds_read128<(WN - 1) * 2048>(wlo[WN - 1], ba);
        ds_read128<(WN - 1) * 2048 + 1024>(whi[WN - 1], ba);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// MX = 1: MX block-scaled activations (cover_gemm_epi.a8_mx): A8 rows are plain row-major e4m3, W8 is the k-linear image. The matrix instruction's own k
// order (probed: tools/dbg/mx_probe2.py, profiles/r06_mx_block_scales.txt) is: operand bytes 0..15 of lane (row, g) = k 16 g .. 16 g + 15 of the 128-deep
// step, bytes 16..31 = k 64 + 16 g ..; block b = k 32 b .. 32 b + 31 takes its E8M0 scale from lane group b. So the LDS read pattern is the one of the
// per-row-scale kernel (chunks g and 4 + g of the row's 128 B) and lane (row, g) hands over byte g of the dword a8mx[k-tile][row]. Every activation-role wave moves one more piece per tile: 64 of the tile's (up to 256) dwords, global_load_lds_dword, into a ring of
// NSTA x 1 KiB behind the weight ring; every wave reads its WM dwords of tile kt + 1 right behind the barrier that publishes it (older than every
// fragment read of that tile, so the counted lgkmcnt waits of the phases cover them) and shifts its own byte down in phase 0 of that tile.
// MX = 2: the kernel can WRITE that form from a GLU epilogue (cover_gemm_epi.out8); its own operands carry per-row scales as with MX = 0.
template <int WM, int WN, int CGM, int CGN, int NSTA, int NSTB, int MX = 0>
__global__ __launch_bounds__(64 * CGM * CGN) void gemm_tiled_v3_f8(const uint8_t* __restrict__ A8, int lda8, const uint8_t* __restrict__ W8, void* C, int ldc, int M,
                                                                   int N, int Kp, EpiDev epi, int tiles_m, int tiles_n, int kt_per, float* __restrict__ partial,
                                                                   const float* __restrict__ a_scale, const float* __restrict__ w_scale) {
    constexpr bool MXA = MX == 1;
    constexpr int NW = CGM * CGN, NH = NW / 2;
    constexpr int BM_ = CGM * WM * 16, BN_ = CGN * WN * 16;
    constexpr int A_BYTES = BM_ * 128, B_BYTES = BN_ * 128;   // one 128-deep k-tile: 128 B per row
    constexpr int AT = A_BYTES / 1024, BT = B_BYTES / 1024;
    constexpr int PTA = AT / NH, PTB = BT / NH;
    constexpr int NSLOT = WN - 2 + WM;                        // DMA slots per tile (see mid_phases_dma / last_phase_dma)
    static_assert(NW == 8, "two waves per SIMD");
    static_assert(AT % NH == 0 && BT % NH == 0, "pieces must split evenly over the waves of a role");
    static_assert(WN >= 2, "the pipelined consumer refills w[0] while w[WN-1] is still needed");
    static_assert(NSTA >= 2 && NSTB >= 2 && (NSTA - 1) * PTA <= 63 && (NSTB - 1) * PTB <= 63, "ring depths / counted vmcnt field");
    static_assert(NSTA * A_BYTES + NSTB * B_BYTES + (MXA ? NSTA * 1024 : 0) <= 160 * 1024, "LDS");
    static_assert(!MXA || BM_ <= 64 * NH, "one block-scale dword per row, 64 rows per activation-role wave");
    static_assert(!MXA || (NSTA - 1) * (PTA + 1) <= 63, "counted vmcnt field (with the block-scale piece)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                    // [NSTA][A_BYTES]
    char* Bs = smem + NSTA * A_BYTES;   // [NSTB][B_BYTES]
    char* Ss = smem + NSTA * A_BYTES + NSTB * B_BYTES;   // MX: [NSTA][256 rows x 4 B]
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int tn = bid / tiles_m, tm = bid % tiles_m;
    const int m0 = tm * BM_, n0 = tn * BN_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K64 = Kp >> 6;
    const int N16 = (N + 15) >> 4;
    const int nk_total = Kp >> 7;
    const int kt0 = blockIdx.y * kt_per;
    const int nk = min(kt_per, nk_total - kt0);
    const uint32_t as_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(As));
    const uint32_t bs_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(Bs));
    const int wm = w / CGN, wn = w % CGN;
    const int r = lane & 15, g = lane >> 4;
    f32x4 acc[WN][WM];  // [n-block b][m-frag f]
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
        for (int f = 0; f < WM; ++f) acc[b][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const uint32_t a_addr0 = lds_addr_u32(As) + ((wm * (WM * 16) + r) * 8 + ((0 * 4 + g) ^ (r & 7))) * 16;
    const uint32_t a_addr1 = lds_addr_u32(As) + ((wm * (WM * 16) + r) * 8 + ((1 * 4 + g) ^ (r & 7))) * 16;
    const uint32_t b_addr = lds_addr_u32(Bs) + (wn * WN * 2 * 64 + lane) * 16;
    const uint32_t s_addr = lds_addr_u32(Ss) + (wm * (WM * 16) + r) * 4;
    const uint32_t ss_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(Ss));
    using Scale = std::conditional_t<MXA, MxScale<WM>, NoScale>;

    auto run = [&](auto ROLE) {
        constexpr int role = decltype(ROLE)::value;
        constexpr int PTD = role ? PTB : PTA;                       // 16-byte DMA pieces per tile
        constexpr int PT = PTD + ((MXA && role == 0) ? 1 : 0);      // ... + the block-scale piece (piece PTD)
        constexpr int NST = role ? NSTB : NSTA, SB = role ? B_BYTES : A_BYTES, KSH = role ? 11 : 7;
        const int wl = w % NH;
        uint32_t voff[PT];
        const char* sbase[PT];
        uint32_t dst0[PT];
        if constexpr (MXA && role == 0) {   // rows 64 wl .. 64 wl + 63 of the tile (clamped: rows beyond the tile / beyond M are never read back)
            int gr = m0 + wl * 64 + lane;
            gr = gr < M ? gr : M - 1;
            voff[PTD] = (uint32_t)gr * 4u;
            sbase[PTD] = (const char*)epi.a8mx + (size_t)kt0 * M * 4;
            dst0[PTD] = ss_u32 + wl * 256;
        }
        const size_t mx_step = (size_t)M * 4;   // bytes from one k-tile's block scales to the next
#pragma unroll
        for (int i = 0; i < PTD; ++i) {
            const int j = wl + NH * i;
            if constexpr (role == 0) {   // A: LDS chunk position p = j*64 + lane: row = p>>3, c = p&7 holds global chunk c ^ (row&7)
                const int row = j * 8 + (lane >> 3), c = lane & 7;
                int gr = m0 + row;
                gr = gr < M ? gr : M - 1;
                voff[i] = (uint32_t)((size_t)gr * lda8 + ((c ^ (row & 7)) << 4));
                sbase[i] = (const char*)(A8 + (size_t)kt0 * 128);
                dst0[i] = as_u32 + j * 1024;
            } else {
                const int nbi = j >> 1, kbi = j & 1;
                int nb = (n0 >> 4) + nbi;
                nb = nb < N16 ? nb : N16 - 1;
                voff[i] = lane * 16;
                sbase[i] = (const char*)(W8 + ((size_t)nb * K64 + (size_t)kt0 * 2 + kbi) * 1024);
                dst0[i] = bs_u32 + j * 1024;
            }
        }
        auto issue = [&](int i, int stage, int t) {
            if (MXA && role == 0 && i == PTD) glds4_s(voff[i], sbase[i] + (size_t)(uint32_t)t * mx_step, dst0[i] + stage * 1024);
            else glds16_s(voff[i], sbase[i] + ((size_t)(uint32_t)t << KSH), dst0[i] + stage * SB);
        };
        // ---- prologue: this wave's pieces of tiles 0 .. NST-1 (every stage), then tile 0 landed -> barrier -> all of its fragments requested
#pragma unroll
        for (int s2 = 0; s2 < NST; ++s2)
#pragma unroll
            for (int p = 0; p < PT; ++p) issue(p, s2, min(s2, nk - 1));
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * PT) : "memory");
        asm volatile("s_barrier" ::: "memory");
        u32x4 xlo[WM], xhi[WM], wlo[WN], whi[WN];
        Scale sc;
        if constexpr (MXA) { sc.sh = 8 * g; read_mx_scales<0, WM>(sc, s_addr); }
        {
            const uint32_t ba = b_addr;
            read_w_frags_range<0, WN - 1, WN>(wlo, whi, ba);
            read_x_frags<0, WM>(xlo, xhi, a_addr0, a_addr1);
            ds_read128<(WN - 1) * 2048>(wlo[WN - 1], ba);
            ds_read128<(WN - 1) * 2048 + 1024>(whi[WN - 1], ba);
        }
        int na = NSTA > 1 ? 1 : 0, nb_ = NSTB > 1 ? 1 : 0;    // read stages of tile kt + 1 in the activation / weight ring
        int cr = 0;                                           // this role's ring: stage of tile kt (released by the barrier behind phase 0)
        for (int kt = 0; kt + 1 < nk; ++kt) {
            phase0<0, WM, WN>(acc, xlo, xhi, wlo, whi, sc);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PT) : "memory");   // own pieces of tile kt + 1 landed; NST - 2 younger tiles in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // w[WN-1] landed too: stage kt is free; tile kt+1 is published
            asm volatile("" : "+v"(wlo[WN - 1]), "+v"(whi[WN - 1]));
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t a0 = a_addr0 + na * A_BYTES, a1 = a_addr1 + na * A_BYTES, ba = b_addr + nb_ * B_BYTES;
            if constexpr (MXA) read_mx_scales<0, WM>(sc, s_addr + na * 1024);   // tile kt + 1's block scales (this tile's were shifted down in phase 0)
            ds_read128<0>(wlo[0], ba);
            ds_read128<1024>(whi[0], ba);
            __builtin_amdgcn_sched_barrier(0);
            const int dt = min(kt + NST, nk - 1), ds = cr;
            auto dma = [&](auto SLOT) {
                constexpr int slot = decltype(SLOT)::value;
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    if ((p * NSLOT) / PT == slot) issue(p, ds, dt);
            };
            mid_phases_dma<1, WM, WN, true>(acc, xlo, xhi, wlo, whi, ba, dma, sc);
            last_phase_dma<0, WM, WN, true>(acc, xlo, xhi, wlo, whi, a0, a1, ba, dma, sc);
            na = na == NSTA - 1 ? 0 : na + 1;
            nb_ = nb_ == NSTB - 1 ? 0 : nb_ + 1;
            cr = cr == NST - 1 ? 0 : cr + 1;
        }
        {   // last tile: no refills, no DMA
            phase0<0, WM, WN>(acc, xlo, xhi, wlo, whi, sc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(wlo[WN - 1]), "+v"(whi[WN - 1]));
            __builtin_amdgcn_sched_barrier(0);
            mid_phases<1, WM, WN, false>(acc, xlo, xhi, wlo, whi, 0u, sc);
            last_phase<0, WM, WN, false>(acc, xlo, xhi, wlo, whi, 0u, 0u, 0u, sc);
        }
    };
    if (w < NH) run(IC8<0>{});
    else run(IC8<1>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus pieces of the clamped tail must not land in the epilogue's staging area
    // quantisation scales (constant along k): row scale of A x channel scale of W, on the fp32 sums
    {
        const int mw = m0 + wm * (WM * 16), nw = n0 + wn * (WN * 16);
        float as[WM];
#pragma unroll
        for (int f = 0; f < WM; ++f) {
            int m = mw + f * 16 + r;
            m = m < M ? m : M - 1;
            as[f] = MXA ? 1.0f : a_scale[m];   // (block scales went into the matrix instruction)
        }
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            int nb = (nw >> 4) + b;
            nb = nb < N16 ? nb : N16 - 1;
            const float4 ws = *(const float4*)(w_scale + (size_t)nb * 16 + 4 * g);
#pragma unroll
            for (int f = 0; f < WM; ++f) {
                acc[b][f][0] *= as[f] * ws.x; acc[b][f][1] *= as[f] * ws.y; acc[b][f][2] *= as[f] * ws.z; acc[b][f][3] *= as[f] * ws.w;
            }
        }
    }
    tiled_epilogue_staged<WM, WN, BM_, BN_, MX == 2>(acc, epi, C, ldc, M, N, m0, n0, m0 + wm * (WM * 16), n0 + wn * (WN * 16), r, g, partial, smem,
                                                     NSTA * A_BYTES + NSTB * B_BYTES, tid, 64 * NW);
}

// ---------------------------------------------------------------------------------------------------
// Dynamic per-row e4m3 quantisation of activation rows, written in the k order the MFMA operands want (see the header):
//   s_m = smallest power of two with max_k |x[m,k]| / s_m <= 448,  q = RNE_e4m3(x / s_m)   (division exact)
//   out[m][64 c + 16 g + 8 h + e] = q[m][64 c + 32 h + 8 g + e],  zero beyond K up to Kp (multiple of 128).
// One 256-thread block per row, 8 elements (16 B in, 8 B out) per thread per step.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quantize_act_fp8_k(const bf16_t* __restrict__ X, int ldx, int K, int Kp, uint8_t* __restrict__ out, int ld8,
                                                          float* __restrict__ scales) {
    __shared__ float red[16];
    const int m = blockIdx.x;
    const bf16_t* x = X + (size_t)m * ldx;
    const int nch = Kp >> 3;
    float mx = 0.f;
    for (int c = threadIdx.x; c < nch; c += 256) {
        const int k = c * 8;
        if (k + 8 <= K) {
            const uint4 v = *(const uint4*)(x + k);
            const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) mx = fmaxf(mx, fmaxf(fabsf(bf2f((bf16_t)(wv[i] & 0xffffu))), fabsf(bf2f((bf16_t)(wv[i] >> 16)))));
        } else {
            for (int e = 0; e < 8; ++e)
                if (k + e < K) mx = fmaxf(mx, fabsf(bf2f(x[k + e])));
        }
    }
    mx = block_max(mx, red);
    float s = 1.0f;
    if (mx > 0.f) {   // smallest power of two s with mx / s <= 448 (as quantize_rows_fp8_k)
        int e;
        const float f = frexpf(mx / 448.0f, &e);
        s = ldexpf(1.0f, f == 0.5f ? e - 1 : e);
    }
    if (threadIdx.x == 0) scales[m] = s;
    const float inv = 1.0f / s;
    uint8_t* o = out + (size_t)m * ld8;
    for (int c = threadIdx.x; c < nch; c += 256) {
        const int k = c * 8;
        float v[8];
        if (k + 8 <= K) {
            const uint4 q = *(const uint4*)(x + k);
            const uint32_t wv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] = bf2f((bf16_t)(wv[i] & 0xffffu)) * inv;
                v[2 * i + 1] = bf2f((bf16_t)(wv[i] >> 16)) * inv;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (k + e < K) ? bf2f(x[k + e]) * inv : 0.f;
        }
        int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
        int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
        // k = 64 cc + 32 h + 8 g  ->  position 64 cc + 16 g + 8 h
        const int cc = k >> 6, h = (k >> 5) & 1, gq = (k >> 3) & 3;
        *(uint2*)(o + cc * 64 + gq * 16 + h * 8) = make_uint2((uint32_t)lo, (uint32_t)hi);
    }
}

// ---------------------------------------------------------------------------------------------------
// The MX form (cover_quantize_act_fp8_mx): plain row-major e4m3, one power-of-two scale per 32 consecutive k of a row, E8M0 bytes in
// mx[k / 128][m][(k / 32) % 4]. One 256-thread block per row, 8 elements per thread per step; the four lanes of a quad hold one block.
// (The GLU epilogue of gemm_tiled_v3_f8<.., MX = 2> writes the same bytes without this launch: tiled_epilogue_staged / mx_quant_chunk.)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quantize_act_fp8_mx_k(const bf16_t* __restrict__ X, int ldx, int M, int K, int Kp, uint8_t* __restrict__ out, int ld8,
                                                             uint8_t* __restrict__ mx) {
    const int m = blockIdx.x;
    const bf16_t* x = X + (size_t)m * ldx;
    uint8_t* o = out + (size_t)m * ld8;
    const int nch = Kp >> 3;   // a multiple of 16: whole quads
    for (int c = threadIdx.x; c < nch; c += 256) {
        const int k = c * 8;
        float v[8];
        if (k + 8 <= K) {
            const uint4 q = *(const uint4*)(x + k);
            const uint32_t wv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[2 * i] = bf2f((bf16_t)(wv[i] & 0xffffu)); v[2 * i + 1] = bf2f((bf16_t)(wv[i] >> 16)); }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (k + e < K) ? bf2f(x[k + e]) : 0.f;
        }
        uint2 q8;
        const uint32_t sb = mx_quant_chunk(v, q8);
        *(uint2*)(o + k) = q8;
        if ((c & 3) == 0) mx[((size_t)(k >> 7) * M + m) * 4 + ((k >> 5) & 3)] = (uint8_t)sb;
    }
}
hipError_t launch_quantize_act_fp8_mx(const bf16_t* X, int ldx, int M, int K, uint8_t* out, int ld8, uint8_t* mx, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    const int Kp = (K + 127) / 128 * 128;
    if (ld8 < Kp || (ld8 & 15) || (ldx & 7)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(quantize_act_fp8_mx_k, dim3(M), dim3(256), 0, st, X, ldx, M, K, Kp, out, ld8, mx);
    return hipGetLastError();
}

hipError_t launch_quantize_act_fp8(const bf16_t* X, int ldx, int M, int K, uint8_t* out, int ld8, float* scales, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    const int Kp = (K + 127) / 128 * 128;
    if (ld8 < Kp || (ld8 & 15) || (ldx & 7)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(quantize_act_fp8_k, dim3(M), dim3(256), 0, st, X, ldx, K, Kp, out, ld8, scales);
    return hipGetLastError();
}

// pick: the tile configuration index of launch_gemm_bf16's table (10: 64x128, 12: 256x128, 13: 128x256, 15: 224x128, 17: 224x96, 18: 128x192; the 224x192 tile needs 84 accumulator + 80 fragment registers and spills at 3 waves per SIMD)
bool gemm_fp8_tiled_supported(int pick) { return pick == 10 || pick == 12 || pick == 13 || pick == 15 || pick == 17 || pick == 18; }

hipError_t launch_gemm_fp8_tiled(int pick, const uint8_t* A8, int lda8, const float* a_scale, const uint8_t* W8, const float* w_scale, void* C, int ldc,
                                 int M, int N, int Kp, const EpiDev& epi, int tiles_m, int tiles_n, int kt_per, int S, float* partial, size_t lds,
                                 int prof_cls, double prof_work, hipStream_t st) {
    hipError_t e = hipSuccess;
    dim3 grid(tiles_m * tiles_n, S);
#define LAUNCH_F8(WM_, WN_, NST_, NL_, CGM_, CGN_) LAUNCH_F8X(WM_, WN_, NST_, NL_, CGM_, CGN_, false, lds)
#define LAUNCH_F8X(WM_, WN_, NST_, NL_, CGM_, CGN_, MX_, lds)                                                                \
    do {                                                                                                                     \
        auto kfn = gemm_tiled_pc_f8<WM_, WN_, NST_, NL_, CGM_, CGN_, MX_>;                                                   \
        if (lds > 64 * 1024) {                                                                                               \
            e = LDS_ATTR_160K(kfn);                                                                                                        \
        }                                                                                                                    \
        if (e == hipSuccess) {                                                                                               \
            dim3 block(64 * (CGM_ * CGN_ + NL_));                                                                            \
            hipEvent_t ea, eb;                                                                                               \
            if (prof_enabled() && prof_reserve(prof_cls, prof_work, &ea, &eb) >= 0)                                          \
                hipExtLaunchKernelGGL(kfn, grid, block, (uint32_t)lds, st, ea, eb, 0, A8, lda8, W8, C, ldc, M, N, Kp, epi, tiles_m, tiles_n, kt_per, partial, a_scale, w_scale); \
            else                                                                                                             \
                hipLaunchKernelGGL(kfn, grid, block, lds, st, A8, lda8, W8, C, ldc, M, N, Kp, epi, tiles_m, tiles_n, kt_per, partial, a_scale, w_scale); \
        }                                                                                                                    \
    } while (0)
#define LAUNCH_F8V3(WM_, WN_, CGM_, CGN_, NA_, NB_) \
    do { if (epi.a8mx) LAUNCH_F8V3X(WM_, WN_, CGM_, CGN_, NA_, NB_, 1); else if (epi.o8) LAUNCH_F8V3X(WM_, WN_, CGM_, CGN_, NA_, NB_, 2); else LAUNCH_F8V3X(WM_, WN_, CGM_, CGN_, NA_, NB_, 0); } while (0)
#define LAUNCH_F8V3X(WM_, WN_, CGM_, CGN_, NA_, NB_, MX_)                                                                    \
    do {                                                                                                                     \
        auto kfn = gemm_tiled_v3_f8<WM_, WN_, CGM_, CGN_, NA_, NB_, MX_>;                                                    \
        const size_t lds3 = ((size_t)NA_ * CGM_ * WM_ * 16 + (size_t)NB_ * CGN_ * WN_ * 16) * 128 + (MX_ == 1 ? (size_t)NA_ * 1024 : 0); \
        e = LDS_ATTR_160K(kfn);                                                                                                            \
        if (e == hipSuccess) {                                                                                               \
            dim3 block(64 * CGM_ * CGN_);                                                                                    \
            hipEvent_t ea, eb;                                                                                               \
            if (prof_enabled() && prof_reserve(prof_cls, prof_work, &ea, &eb) >= 0)                                          \
                hipExtLaunchKernelGGL(kfn, grid, block, (uint32_t)lds3, st, ea, eb, 0, A8, lda8, W8, C, ldc, M, N, Kp, epi, tiles_m, tiles_n, kt_per, partial, a_scale, w_scale); \
            else                                                                                                             \
                hipLaunchKernelGGL(kfn, grid, block, lds3, st, A8, lda8, W8, C, ldc, M, N, Kp, epi, tiles_m, tiles_n, kt_per, partial, a_scale, w_scale); \
        }                                                                                                                    \
    } while (0)
    // the self-loading form (gemm_tiled_v3_f8) for the tiles whose eight MFMA waves split into two DMA roles; COVER_V3_F8=0 keeps the loader-wave form
    static const char* v3f8_env = getenv("COVER_V3_F8");
    const bool v3f8 = !(v3f8_env && v3f8_env[0] == '0') && (size_t)M * lda8 + 4096 < ((size_t)1 << 31);
    // MX block scales (operand or output) exist on the self-loading kernels only: launch_gemm_bf16 keeps such a GEMM on their tiles
    if ((epi.a8mx || epi.o8) && !(v3f8 && (pick == 12 || pick == 13 || pick == 15 || pick == 18)) && !(pick == 10 && epi.a8mx && !epi.o8)) return hipErrorInvalidValue;
    if (epi.a8mx && !epi.w8_kl) return hipErrorInvalidValue;
    if (v3f8 && (pick == 12 || pick == 13 || pick == 15 || pick == 18)) {
        switch (pick) {
            case 12: LAUNCH_F8V3(4, 4, 4, 2, 3, 3); break;
            case 13: LAUNCH_F8V3(4, 4, 2, 4, 3, 3); break;
            case 15: LAUNCH_F8V3(7, 2, 2, 4, 3, 3); break;
            default: LAUNCH_F8V3(4, 3, 2, 4, 3, 3); break;
        }
        if (e == hipSuccess) e = hipGetLastError();
        return e;
    }
    switch (pick) {
        case 10:
            if (epi.a8mx) { const size_t ldsx = lds + 4 * 256; LAUNCH_F8X(2, 4, 4, 4, 2, 2, true, ldsx); }   // (+ the block-scale ring)
            else LAUNCH_F8(2, 4, 4, 4, 2, 2);
            break;
        case 12: LAUNCH_F8(4, 4, 3, 4, 4, 2); break;
        case 13: LAUNCH_F8(4, 4, 3, 4, 2, 4); break;
        case 15: LAUNCH_F8(7, 2, 3, 4, 2, 4); break;
        case 17: LAUNCH_F8(7, 2, 4, 4, 2, 3); break;
        case 18: LAUNCH_F8(4, 3, 3, 4, 2, 4); break;
        default: return hipErrorInvalidValue;
    }
#undef LAUNCH_F8
#undef LAUNCH_F8X
    if (e == hipSuccess) e = hipGetLastError();
    return e;
}
