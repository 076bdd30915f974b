// Self-loading 8-wave tiled bf16 GEMM for gfx950 ("v3"): C[M,N] = epi(A[M,K] . W[N,K]^T), fp32 accumulate on v_mfma_f32_16x16x32_bf16.
//
// Same operand layouts, LDS images and epilogue as gemm_tiled_pc (gemm_bf16.hip): packed fragment-major weights, activation rows
// staged with the XOR chunk swizzle on the SOURCE address of the LDS-DMA, accumulators acc[n-block][m-fragment]. What differs is who
// does what, and when:
//   * no loader waves. gemm_tiled_pc's 4 loader waves make the block 12 waves = 3 per SIMD = 168 registers per lane, which is what the
//     84 accumulator + 80 fragment registers of a 112 x 48 wave tile just fit -- with nothing left to schedule with. Here the block is
//     8 waves (2 x 4 wave tiles, two per SIMD, 256 registers) and every wave issues its share of the tile's LDS-DMA pieces itself.
//   * nothing is issued as a burst. In gemm_tiled_pc every MFMA wave pushes the WM + WN fragment reads of the next 32-deep step at the
//     LDS right behind the barrier, all eight waves at the same moment (80 KiB = 320 LDS cycles per step), and a wave cannot issue its
//     first MFMA before its last read has been accepted; the in-kernel counters of round 4 had the matrix pipes idle 30 % of a k-tile
//     with or without DMA. Here the WM + WN reads of the next step and the wave's DMA pieces are spread BETWEEN the MFMAs of the
//     current step at compile-time positions (read j behind MFMA j NM / NR, piece p behind MFMA (2 p + 1) NM / PT of the two-step
//     phase), so the LDS and the vector-memory address path see an even request stream and the matrix pipe always has the next MFMA.
//   * one s_barrier per 64-deep k-tile, in the middle of the tile (between its two steps) as before: a wave reaching barrier kt + 1
//     has every fragment of tile kt in registers, so that barrier releases the stage of tile kt for tile kt + NST.
// Ordering of LDS-DMA data for the readers: every wave waits for ITS OWN pieces of tile kt + 1 with a counted vmcnt (the pieces of the
// younger tiles stay in flight) before barrier kt + 1; the first fragment read of tile kt + 1 is issued behind that barrier.
#include <stdlib.h>
#include <type_traits>
#include <hip/hip_ext.h>
#include "common.h"
#include "kernels.h"
// the epilogue's own stamps (gemm_common.h PCTL(4..6): stages released / LDS tile filled / tile published) go to the probe words as well
__device__ unsigned long long g_v3_probe[16];
#define V3P(slot, val) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_v3_probe[slot] = (val); } while (0)
#define PCTL(slot) V3P(8 + (slot), wall_clock64())
#include "gemm_common.h"

// In-kernel probe of the LAST launch (always on: six stores by one thread of workgroup (0, 0)): wall-clock stamps (100 MHz) at kernel start,
// loop start, loop end and kernel end, and the shader cycle counter at loop start / end -- the clock the loop really ran at (the matrix
// peak the rooflines are priced against assumes 2.4 GHz; under MFMA + LDS + HBM load the chip sustains 1.6-1.9) and the split of a
// launch into prologue / k-loop / epilogue. cover_gemm_probe() reads it.
int gemm_v3_probe(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_v3_probe), sizeof(g_v3_probe)) == hipSuccess ? 0 : -1; }

template <int V> using IC = std::integral_constant<int, V>;
// Experiment knobs (compile time; tools/ab_build.sh builds the variants):
//   COVER_V3_RDSPAN_NUM / _DEN  the NR fragment reads of the next step sit behind the first NM * NUM / DEN MFMAs of a step (1 / 1 = spread over all of them:
//                               the round-5 schedule, where the last read went out behind MFMA 18 of 21 and the step's lgkmcnt(0) waited for it)
//   COVER_V3_PRIO               1: the weight-role waves (second half) run at s_setprio 1 inside the loop, 2: the activation-role waves
//   COVER_V3_DEPHASE_NUM / _DEN the weight-role waves pass the mid-tile barrier NM * NUM / DEN MFMAs into the tile's second step, so that the two waves of a
//                               SIMD do not reach their `s_waitcnt lgkmcnt(0)` at the same moment
#ifndef COVER_V3_RDSPAN_NUM   // default 4 / 7 (12 of 21 MFMAs): pi0 prefix layer 552 -> 542 us over three same-box repetitions, M = 448 flat (profiles/r06_v3_schedule_ab.txt)
#define COVER_V3_RDSPAN_NUM 4
#define COVER_V3_RDSPAN_DEN 7
#endif
#ifndef COVER_V3_PRIO
#define COVER_V3_PRIO 0
#endif
#ifndef COVER_V3_DEPHASE_NUM
#define COVER_V3_DEPHASE_NUM 0
#define COVER_V3_DEPHASE_DEN 1
#endif
// position (MFMA index inside the two-step phase of 2 NM MFMAs that follows a barrier) behind which a wave issues its piece p of PT:
// evenly spread in general; with a TWO-stage ring the refill of the stage the barrier has just released is the tile the NEXT barrier
// needs, so its pieces go out at once
constexpr int v3_pos(int p, int PT, int NM, int NST) { return NST == 2 ? p : ((2 * p + 1) * NM) / PT; }
constexpr int v3_half_pieces(int PT, int NM, int NST) {   // pieces issued in the first step of the phase
    int c = 0;
    for (int p = 0; p < PT; ++p)
        if (v3_pos(p, PT, NM, NST) < NM) ++c;
    return c;
}

template <int WM, int WN, int CGM, int CGN, int NSTA, int NSTB>
__global__ __launch_bounds__(64 * CGM * CGN) void gemm_tiled_v3(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ Wp, void* C, int ldc, int M, int N,
                                                     int Kp, EpiDev epi, int tiles_m, int tiles_n, int kt_per, float* __restrict__ partial) {
    PCTL(0);
    V3P(0, wall_clock64());
    constexpr int NW = CGM * CGN, NH = NW / 2;                            // waves: a CGM x CGN grid of wave tiles (2 x 4: two waves per SIMD, 2 x 2: one)
    constexpr int BM_ = CGM * WM * 16, BN_ = CGN * WN * 16;
    constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = BN_ * BK * 2;
    constexpr int AT = A_BYTES / 1024, BT = B_BYTES / 1024;               // 1-KiB LDS-DMA pieces per k-tile
    constexpr int PTA = AT / NH, PTB = BT / NH;                           // pieces per wave: the first half of the waves owns the activation pieces, the second the weight pieces
    constexpr int NM = WM * WN, NR = WM + WN;                             // MFMAs / fragment reads per 32-deep step
    static_assert(NW == 4 || NW == 8, "one or two waves per SIMD");
    static_assert(AT % NH == 0 && BT % NH == 0, "pieces must split evenly over the waves of a role");
    static_assert(NSTA >= 2 && NSTB >= 2, "a ring needs two stages");
    static_assert((NSTA - 2) * PTA <= 63 && (NSTB - 2) * PTB <= 63, "counted vmcnt must fit its 6-bit field");
    static_assert(PTA <= 2 * NM && PTB <= 2 * NM, "at most one piece behind an MFMA");
    static_assert(NSTA * A_BYTES + NSTB * B_BYTES <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                   // [NSTA][A_BYTES]
    char* Bs = smem + NSTA * A_BYTES;  // [NSTB][B_BYTES]
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: the row tiles of one column of tiles (one weight panel) land on one L2
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int tn = bid / tiles_m, tm = bid % tiles_m;
    const int m0 = tm * BM_, n0 = tn * BN_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K32 = Kp >> 5;
    const int N16 = (N + 15) >> 4;
    const int nk_total = Kp / BK;
    const int kt0 = blockIdx.y * kt_per;
    const int nk = min(kt_per, nk_total - kt0);
    const uint32_t as_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(As));
    const uint32_t bs_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(Bs));

    // ---- MFMA side (every wave) ----
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
    // fragment read j of a step, in the order of first use by the MFMA loop below: w0, x0 .. x(WM-1), w1 .. w(WN-1)
    auto read_nth = [&](uint32_t aa, uint32_t ba, int j, u32x4(&xf)[WM], u32x4(&wf)[WN]) {
        if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(wf[0]) : "v"(ba));
        else if (j <= WM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[j - 1]) : "v"(aa), "n"((j - 1) * 2048));
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[j - WM]) : "v"(ba), "n"((j - WM) * 2048));
    };
    auto landed = [&](u32x4(&xf)[WM], u32x4(&wf)[WN]) {   // ties the fragments to the wait that precedes this call
#pragma unroll
        for (int f = 0; f < WM; ++f) asm volatile("" : "+v"(xf[f]));
#pragma unroll
        for (int b = 0; b < WN; ++b) asm volatile("" : "+v"(wf[b]));
    };

    // ---- the whole pipeline, once per DMA role (ROLE 0: this wave owns activation pieces, 1: weight pieces). Each role runs its own ring
    // depth: a wave's vmcnt retires in issue order, so one wave cannot keep a deep weight ring (HBM latency) in flight behind a shallow
    // activation ring (L2 latency) -- two kinds of wave can.
    auto run = [&](auto ROLE) {
        constexpr int role = decltype(ROLE)::value;
        constexpr int PT = role ? PTB : PTA, NST = role ? NSTB : NSTA, SB = role ? B_BYTES : A_BYTES, KSH = role ? 11 : 7;
        constexpr int HALF = v3_half_pieces(PT, NM, NST);
        const int wl = w % NH;
        uint32_t voff[PT];          // per lane: byte offset from the piece's scalar base
        const char* sbase[PT];      // wave-uniform: source of the piece in the slice's first k-tile
        uint32_t dst0[PT];          // wave-uniform: LDS byte address of the piece in stage 0
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int j = wl + NH * i;
            if constexpr (role == 0) {   // A: LDS chunk position p = j*64 + lane: row = p>>3, c = p&7 holds global chunk c ^ (row&7)
                const int row = j * 8 + (lane >> 3), c = lane & 7;
                int gr = m0 + row;
                gr = gr < M ? gr : M - 1;
                voff[i] = (uint32_t)(((size_t)gr * lda + ((c ^ (row & 7)) << 3)) * 2);
                sbase[i] = (const char*)(A + (size_t)kt0 * BK);
                dst0[i] = as_u32 + j * 1024;
            } else {
                const int nbi = j >> 1, kbi = j & 1;
                int nb = (n0 >> 4) + nbi;
                nb = nb < N16 ? nb : N16 - 1;
                voff[i] = lane * 16;
                sbase[i] = (const char*)(Wp + ((size_t)nb * K32 + (size_t)kt0 * 2 + kbi) * 512);
                dst0[i] = bs_u32 + j * 1024;
            }
        }
        // issue piece i of tile t into stage `stage` of this role's ring. t is clamped to the last tile by the caller: the surplus issues at
        // the end of the K range re-load the last tile into a stage nobody reads any more, which keeps every vmcnt count of the steady state exact.
        auto issue = [&](int i, int stage, int t) { glds16_s(voff[i], sbase[i] + ((size_t)(uint32_t)t << KSH), dst0[i] + stage * SB); };
        // one 32-deep step: NM MFMAs on (xf, wf) with -- behind them -- the reads of step RKS of stages (rsa, rsb) into (xn, wn_) and this wave's DMA
        // pieces of phase half DH (0: first step behind a barrier, 1: second) of tile dt into stage ds
        // MFMAs [I0, I1) of the step only (the de-phased waves split the step that holds the barrier); the reads sit behind the first SPAN of them
        auto group = [&](const u32x4(&xf)[WM], const u32x4(&wf)[WN], u32x4(&xn)[WM], u32x4(&wn_)[WN], int rsa, int rsb, auto RKS, auto DH, int ds, auto RD,
                         auto DMA, int dt, auto I0_, auto I1_) {
            constexpr int rks = decltype(RKS)::value, dh = decltype(DH)::value;
            constexpr bool rd = decltype(RD)::value != 0, dma = decltype(DMA)::value != 0;
            constexpr int I0 = decltype(I0_)::value, I1 = decltype(I1_)::value;
            constexpr int SPAN0 = (NM * COVER_V3_RDSPAN_NUM) / COVER_V3_RDSPAN_DEN;
            constexpr int SPAN = SPAN0 < 1 ? 1 : (SPAN0 > I1 - I0 ? I1 - I0 : SPAN0);
            const uint32_t aa = (rks ? a_addr1 : a_addr0) + rsa * A_BYTES;
            const uint32_t ba = b_addr + rsb * B_BYTES + rks * 1024;
#pragma unroll
            for (int b = 0; b < WN; ++b)
#pragma unroll
                for (int f = 0; f < WM; ++f) {
                    const int i = b * WM + f;
                    if (i < I0 || i >= I1) continue;
                    acc[b][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[b]), __builtin_bit_cast(bf16x8, xf[f]), acc[b][f], 0, 0, 0);
                    if (rd) {
#pragma unroll
                        for (int j = 0; j < NR; ++j)
                            if (I0 + (j * SPAN) / NR == i) read_nth(aa, ba, j, xn, wn_);
                    }
                    if (dma) {
#pragma unroll
                        for (int p = 0; p < PT; ++p) {
                            const int pos = v3_pos(p, PT, NM, NST) - dh * NM;                 // position inside this step, < 0 or >= NM: the other step's piece
                            if (pos >= 0 && pos < NM && (pos == i || (i == I0 && pos < I0))) issue(p, ds, dt);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
        // ---- prologue: tiles 0 .. NST-2 and the first-step pieces of tile NST-1 ----
#pragma unroll
        for (int s = 0; s < NST - 1; ++s)
#pragma unroll
            for (int p = 0; p < PT; ++p) issue(p, s, min(s, nk - 1));
#pragma unroll
        for (int p = 0; p < PT; ++p)
            if (v3_pos(p, PT, NM, NST) < NM) issue(p, NST - 1, min(NST - 1, nk - 1));
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PT + HALF) : "memory");   // tile 0: everything but the younger tiles and the half tile
        __builtin_amdgcn_s_barrier();
        PCTL(1);
        V3P(1, wall_clock64());
        V3P(4, __builtin_readcyclecounter());
        u32x4 xa[WM], wa[WN], xb[WM], wb[WN];
#pragma unroll
        for (int j = 0; j < NR; ++j) read_nth(a_addr0, b_addr, j, xa, wa);
        asm volatile("s_waitcnt lgkmcnt(0)");
        landed(xa, wa);
        // ---- main loop (stage indices are wave-uniform run-time values: they only enter scalar address arithmetic) ----
        // tile kt:  first its step 0 | reads of its step 1 | second-step pieces of tile kt + NST - 1 -> the stage of tile kt - 1
        //           own pieces of tile kt + 1 landed, fragments of tile kt complete -> barrier kt + 1
        //           then its step 1 | reads of step 0 of tile kt + 1 | first-step pieces of tile kt + NST -> the stage of tile kt
        int ca = 0, cb = 0, cr = 0, pr = NST - 1;      // read stages of the activation / weight ring; this role's DMA ring (current / previous tile)
        constexpr int DPH = role == 1 ? (NM * COVER_V3_DEPHASE_NUM) / COVER_V3_DEPHASE_DEN : 0;   // MFMAs of the second step in front of the barrier
        static_assert(DPH >= 0 && DPH < NM, "the barrier stays inside the step");
        if (COVER_V3_PRIO == 1 + (1 - role)) __builtin_amdgcn_s_setprio(1);
        for (int kt = 0; kt < nk - 1; ++kt) {
            const int na = ca == NSTA - 1 ? 0 : ca + 1, nb_ = cb == NSTB - 1 ? 0 : cb + 1, nr = cr == NST - 1 ? 0 : cr + 1;
            group(xa, wa, xb, wb, ca, cb, IC<1>{}, IC<1>{}, pr, IC<1>{}, IC<1>{}, min(kt + NST - 1, nk - 1), IC<0>{}, IC<NM>{});
            asm volatile("s_waitcnt lgkmcnt(0)");
            if constexpr (DPH > 0) {
                landed(xb, wb);
                group(xb, wb, xa, wa, na, nb_, IC<0>{}, IC<0>{}, cr, IC<0>{}, IC<0>{}, 0, IC<0>{}, IC<DPH>{});
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PT) : "memory");
            __builtin_amdgcn_s_barrier();
            landed(xb, wb);
            group(xb, wb, xa, wa, na, nb_, IC<0>{}, IC<0>{}, cr, IC<1>{}, IC<1>{}, min(kt + NST, nk - 1), IC<DPH>{}, IC<NM>{});
            asm volatile("s_waitcnt lgkmcnt(0)");
            landed(xa, wa);
            ca = na; cb = nb_; pr = cr; cr = nr;
        }
        if (COVER_V3_PRIO) __builtin_amdgcn_s_setprio(0);
        // last tile of the slice: no barrier, no DMA, no reads of a next tile
        group(xa, wa, xb, wb, ca, cb, IC<1>{}, IC<1>{}, pr, IC<1>{}, IC<0>{}, 0, IC<0>{}, IC<NM>{});
        asm volatile("s_waitcnt lgkmcnt(0)");
        landed(xb, wb);
        group(xb, wb, xa, wa, ca, cb, IC<0>{}, IC<0>{}, cr, IC<0>{}, IC<0>{}, 0, IC<0>{}, IC<NM>{});
    };
    if (w < NH) run(IC<0>{});
    else run(IC<1>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus pieces of the clamped tail must not land in the epilogue's staging area
    PCTL(2);
    V3P(5, __builtin_readcyclecounter());
    V3P(2, wall_clock64());
    V3P(6, (unsigned long long)nk);
    tiled_epilogue_staged<WM, WN, BM_, BN_>(acc, epi, C, ldc, M, N, m0, n0, m0 + wm * (WM * 16), n0 + wn * (WN * 16), r, g, partial, smem,
                                            NSTA * A_BYTES + NSTB * B_BYTES, tid, 64 * NW);
    PCTL(3);
    V3P(3, wall_clock64());
}

// ---------------------------------------------------------------------------------------------------
// K-split wave pairs ("v3k", round 6). The tiles above put CGM x CGN wave tiles on the block's waves: a 224 x 96 tile (qkv at M = 448: 256 of
// them = one per CU) has only 2 x 2 wave tiles of 112 x 48 = ONE wave per SIMD, and that wave cannot cover its own fragment reads with its own
// MFMAs (1 199 cycles per k-tile against 672 of MFMA issue); a 224 x 128 tile on eight waves is 112 x 32 per wave = 0.64 fragment reads per
// MFMA. Here the two waves of a SIMD (w and w + 4) own the SAME wave tile and split every 64-deep k-tile between them: wave-pair member wk
// runs the 32-deep step wk of every tile. Two waves per SIMD whatever the tile (one's LDS wait hides under the other's MFMAs), wave tiles
// twice as large at the same register budget per k (112 x 64: 0.39 reads per MFMA), the same LDS images, DMA roles and counted waits as above.
// One 32-deep step per wave per k-tile makes the loop simpler than the two-step one above:
//   body kt:  NM MFMAs on the fragments of tile kt | the NR fragment reads of tile kt + 1 behind the first of them | this wave's DMA pieces
//             of tile kt + NST into the stage of tile kt (every wave has read tile kt before the barrier that opened this body)
//             own pieces of tile kt + 2 landed (counted vmcnt: NST - 2 younger tiles stay in flight), fragments complete -> barrier
// At the end the second member hands its sums to the first through the (dead) ring, which runs the epilogue's fill; all eight waves share
// the store loop. Sums: (steps 0 of all tiles) + (steps 1 of all tiles) in fp32 -- another order than the kernels above, same rounding points.
// ---------------------------------------------------------------------------------------------------
template <int WM, int WN, int CGM, int CGN, int NST>
__global__ __launch_bounds__(128 * CGM * CGN) void gemm_tiled_v3k(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ Wp, void* C, int ldc, int M, int N,
                                                               int Kp, EpiDev epi, int tiles_m, int tiles_n, int kt_per, float* __restrict__ partial) {
    PCTL(0);
    V3P(0, wall_clock64());
    constexpr int NT = CGM * CGN, NW = 2 * NT, NH = NT;                   // wave tiles; waves (pair member wk = w / NT); waves per DMA role
    constexpr int BM_ = CGM * WM * 16, BN_ = CGN * WN * 16;
    constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = BN_ * BK * 2;
    constexpr int AT = A_BYTES / 1024, BT = B_BYTES / 1024;
    constexpr int PTA = AT / NH, PTB = BT / NH;
    constexpr int NM = WM * WN, NR = WM + WN;
    static_assert(NW == 8, "two waves per SIMD");
    static_assert(AT % NH == 0 && BT % NH == 0, "pieces must split evenly over the waves of a role");
    static_assert(NST >= 3, "the ring holds the tile being read, the tile landing and at least one in flight");
    static_assert((NST - 2) * PTA <= 63 && (NST - 2) * PTB <= 63, "counted vmcnt must fit its 6-bit field");
    static_assert(PTA <= NM && PTB <= NM, "at most one piece behind an MFMA");
    static_assert(NST * (A_BYTES + B_BYTES) <= 160 * 1024, "LDS");
    static_assert(NT * NM * 1024 <= NST * (A_BYTES + B_BYTES), "the pair hand-off fits the ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                   // [NST][A_BYTES]
    char* Bs = smem + NST * A_BYTES;   // [NST][B_BYTES]
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap (as above)
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int tn = bid / tiles_m, tm = bid % tiles_m;
    const int m0 = tm * BM_, n0 = tn * BN_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K32 = Kp >> 5;
    const int N16 = (N + 15) >> 4;
    const int nk_total = Kp / BK;
    const int kt0 = blockIdx.y * kt_per;
    const int nk = min(kt_per, nk_total - kt0);
    const uint32_t as_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(As));
    const uint32_t bs_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(Bs));
    const int wk = w / NT, ww = w % NT;
    const int wm = ww / CGN, wn = ww % CGN;
    const int r = lane & 15, g = lane >> 4;
    f32x4 acc[WN][WM];
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
        for (int f = 0; f < WM; ++f) acc[b][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment addresses of this wave's step (wk) in stage 0
    const uint32_t a_addr = lds_addr_u32(As) + ((wm * (WM * 16) + r) * 8 + ((wk * 4 + g) ^ (r & 7))) * 16;
    const uint32_t b_addr = lds_addr_u32(Bs) + (wn * WN * 2 * 64 + lane) * 16 + wk * 1024;
    auto read_nth = [&](uint32_t aa, uint32_t ba, int j, u32x4(&xf)[WM], u32x4(&wf)[WN]) {   // first-use order: w0, x0 .. x(WM-1), w1 .. w(WN-1)
        if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(wf[0]) : "v"(ba));
        else if (j <= WM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[j - 1]) : "v"(aa), "n"((j - 1) * 2048));
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[j - WM]) : "v"(ba), "n"((j - WM) * 2048));
    };
    auto landed = [&](u32x4(&xf)[WM], u32x4(&wf)[WN]) {
#pragma unroll
        for (int f = 0; f < WM; ++f) asm volatile("" : "+v"(xf[f]));
#pragma unroll
        for (int b = 0; b < WN; ++b) asm volatile("" : "+v"(wf[b]));
    };
    auto run = [&](auto ROLE) {
        constexpr int role = decltype(ROLE)::value;
        constexpr int PT = role ? PTB : PTA, SB = role ? B_BYTES : A_BYTES, KSH = role ? 11 : 7;
        constexpr int RSPAN = (NM * 2) / 3 < NR ? NM : (NM * 2) / 3;      // the reads sit behind the first two thirds of the MFMAs: landed when the last one issues
        const int wl = w % NH;
        uint32_t voff[PT];
        const char* sbase[PT];
        uint32_t dst0[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int j = wl + NH * i;
            if constexpr (role == 0) {
                const int row = j * 8 + (lane >> 3), c = lane & 7;
                int gr = m0 + row;
                gr = gr < M ? gr : M - 1;
                voff[i] = (uint32_t)(((size_t)gr * lda + ((c ^ (row & 7)) << 3)) * 2);
                sbase[i] = (const char*)(A + (size_t)kt0 * BK);
                dst0[i] = as_u32 + j * 1024;
            } else {
                const int nbi = j >> 1, kbi = j & 1;
                int nb = (n0 >> 4) + nbi;
                nb = nb < N16 ? nb : N16 - 1;
                voff[i] = lane * 16;
                sbase[i] = (const char*)(Wp + ((size_t)nb * K32 + (size_t)kt0 * 2 + kbi) * 512);
                dst0[i] = bs_u32 + j * 1024;
            }
        }
        auto issue = [&](int i, int stage, int t) { glds16_s(voff[i], sbase[i] + ((size_t)(uint32_t)t << KSH), dst0[i] + stage * SB); };
        // NM MFMAs on (xf, wf); behind them the reads of stage rs into (xn, wn_) and this wave's pieces of tile dt into stage ds.
        // (A 112 x 64 wave tile -- 224 x 128 on four pairs, with the activation fragments single-buffered -- was built too: 112 accumulator
        // registers leave the loop 14 spills per k-tile and the epilogue 600; measured 47.7 / 76.4 us against 34.2 / 56.7 for o_proj / down at
        // M = 448 on the eight-wave 224 x 128 kernel. Not kept: profiles/r06_v3_schedule_ab.txt.)
        auto group = [&](const u32x4(&xf)[WM], const u32x4(&wf)[WN], u32x4(&xn)[WM], u32x4(&wn_)[WN], int rs, int ds, int dt, auto RD) {
            constexpr bool rd = decltype(RD)::value != 0;
            const uint32_t aa = a_addr + rs * A_BYTES, ba = b_addr + rs * B_BYTES;
#pragma unroll
            for (int b = 0; b < WN; ++b)
#pragma unroll
                for (int f = 0; f < WM; ++f) {
                    const int i = b * WM + f;
                    acc[b][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[b]), __builtin_bit_cast(bf16x8, xf[f]), acc[b][f], 0, 0, 0);
                    if (rd) {
#pragma unroll
                        for (int j = 0; j < NR; ++j)
                            if ((j * RSPAN) / NR == i) read_nth(aa, ba, j, xn, wn_);
#pragma unroll
                        for (int p = 0; p < PT; ++p)
                            if (((2 * p + 1) * NM) / (2 * PT) == i) issue(p, ds, dt);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
        // ---- prologue: every stage of the ring is requested; tiles 0 and 1 have landed before anybody reads ----
#pragma unroll
        for (int s = 0; s < NST; ++s)
#pragma unroll
            for (int p = 0; p < PT; ++p) issue(p, s, min(s, nk - 1));
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PT) : "memory");
        __builtin_amdgcn_s_barrier();
        PCTL(1);
        V3P(1, wall_clock64());
        V3P(4, __builtin_readcyclecounter());
        u32x4 xa[WM], wa[WN], xb[WM], wb[WN];
#pragma unroll
        for (int j = 0; j < NR; ++j) read_nth(a_addr, b_addr, j, xa, wa);
        asm volatile("s_waitcnt lgkmcnt(0)");
        __builtin_amdgcn_s_barrier();          // every wave has read tile 0: body 0 may refill its stage
        landed(xa, wa);
        int rs = 1, ds = 0;                    // stage of tile kt + 1 (read) / of tile kt (refilled with tile kt + NST)
        auto body = [&](const u32x4(&xf)[WM], const u32x4(&wf)[WN], u32x4(&xn)[WM], u32x4(&wn_)[WN], int kt) {
            group(xf, wf, xn, wn_, rs, ds, min(kt + NST, nk - 1), IC<1>{});
            asm volatile("s_waitcnt lgkmcnt(0)");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PT) : "memory");
            __builtin_amdgcn_s_barrier();
            landed(xn, wn_);
            ds = rs;
            rs = rs == NST - 1 ? 0 : rs + 1;
        };
        int kt = 0;
        for (; kt + 2 <= nk - 1; kt += 2) {
            body(xa, wa, xb, wb, kt);
            body(xb, wb, xa, wa, kt + 1);
        }
        if (kt < nk - 1) {
            body(xa, wa, xb, wb, kt);
            group(xb, wb, xa, wa, 0, 0, 0, IC<0>{});
        } else {
            group(xa, wa, xb, wb, 0, 0, 0, IC<0>{});
        }
    };
    if (w < NH) run(IC<0>{});
    else run(IC<1>{});
    PCTL(2);
    V3P(5, __builtin_readcyclecounter());
    V3P(2, wall_clock64());
    V3P(6, (unsigned long long)nk);
    // ---- the pair's sums: member 1 -> ring -> member 0 ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the surplus pieces of the clamped tail have landed; nobody reads the ring any more ...
    __builtin_amdgcn_s_barrier();                                   // ... in any wave
    {
        const uint32_t red = lds_addr_u32(smem) + (uint32_t)((ww * NM) * 64 + lane) * 16;
        if (wk == 1) {
#pragma unroll
            for (int b = 0; b < WN; ++b)
#pragma unroll
                for (int f = 0; f < WM; ++f)
                    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(red), "v"(acc[b][f]), "n"((b * WM + f) * 1024) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wk == 0) {   // one n-block (WM reads in flight) per round trip
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                f32x4 o[WM];
#pragma unroll
                for (int f = 0; f < WM; ++f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(o[f]) : "v"(red), "n"((b * WM + f) * 1024) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int f = 0; f < WM; ++f) {
                    asm volatile("" : "+v"(o[f]));
                    acc[b][f] += o[f];
                }
            }
        }
    }
    tiled_epilogue_staged<WM, WN, BM_, BN_>(acc, epi, C, ldc, M, N, m0, n0, m0 + wm * (WM * 16), n0 + wn * (WN * 16), r, g, partial, smem,
                                            NST * (A_BYTES + B_BYTES), tid, 64 * NW, wk == 0);
    PCTL(3);
    V3P(3, wall_clock64());
}

// pick -> instantiation (the tile table of launch_gemm_bf16 continues with these indices)
hipError_t launch_gemm_v3(int pick, const bf16_t* A, int lda, const bf16_t* Wp, void* C, int ldc, int M, int N, int Kp, const EpiDev& epi, int tiles_m,
                          int tiles_n, int kt_per, int S, float* partial, int prof_cls, double prof_work, hipStream_t st) {
    hipError_t e = hipSuccess;
    if ((size_t)M * lda * 2 + 4096 >= ((size_t)1 << 31)) return hipErrorInvalidValue;   // 32-bit lane offsets of the activation pieces
    dim3 grid(tiles_m * tiles_n, S);
#define LAUNCH_V3(WM_, WN_, CGM_, CGN_, NA_, NB_)                                                                            \
    do {                                                                                                                     \
        auto kfn = gemm_tiled_v3<WM_, WN_, CGM_, CGN_, NA_, NB_>;                                                            \
        const size_t lds = ((size_t)NA_ * CGM_ * WM_ * 16 + (size_t)NB_ * CGN_ * WN_ * 16) * BK * 2;                         \
        dim3 block(64 * CGM_ * CGN_);                                                                                        \
        e = LDS_ATTR_160K(kfn);                                                                                                            \
        if (e == hipSuccess) {                                                                                               \
            hipEvent_t ea, eb;                                                                                               \
            if (prof_enabled() && prof_reserve(prof_cls, prof_work, &ea, &eb) >= 0)                                          \
                hipExtLaunchKernelGGL(kfn, grid, block, (uint32_t)lds, st, ea, eb, 0, A, lda, Wp, C, ldc, M, N, Kp, epi, tiles_m, tiles_n, kt_per, partial); \
            else                                                                                                             \
                hipLaunchKernelGGL(kfn, grid, block, lds, st, A, lda, Wp, C, ldc, M, N, Kp, epi, tiles_m, tiles_n, kt_per, partial); \
        }                                                                                                                    \
    } while (0)
#define LAUNCH_V3K(WM_, WN_, CGM_, CGN_, NST_)                                                                               \
    do {                                                                                                                     \
        auto kfn = gemm_tiled_v3k<WM_, WN_, CGM_, CGN_, NST_>;                                                               \
        const size_t lds = ((size_t)NST_ * (CGM_ * WM_ * 16 + CGN_ * WN_ * 16)) * BK * 2;                                   \
        dim3 block(128 * CGM_ * CGN_);                                                                                       \
        e = LDS_ATTR_160K(kfn);                                                                                              \
        if (e == hipSuccess) {                                                                                               \
            hipEvent_t ea, eb;                                                                                               \
            if (prof_enabled() && prof_reserve(prof_cls, prof_work, &ea, &eb) >= 0)                                          \
                hipExtLaunchKernelGGL(kfn, grid, block, (uint32_t)lds, st, ea, eb, 0, A, lda, Wp, C, ldc, M, N, Kp, epi, tiles_m, tiles_n, kt_per, partial); \
            else                                                                                                             \
                hipLaunchKernelGGL(kfn, grid, block, lds, st, A, lda, Wp, C, ldc, M, N, Kp, epi, tiles_m, tiles_n, kt_per, partial); \
        }                                                                                                                    \
    } while (0)
    switch (pick) {
        case 23: LAUNCH_V3(7, 3, 2, 4, 3, 3); break;          // 224x192, 8 waves of 112x48
        case 24: LAUNCH_V3(7, 2, 2, 4, 3, 3); break;          // 224x128, 8 waves of 112x32
        case 25: LAUNCH_V3(8, 2, 2, 4, 3, 3); break;          // 256x128, 8 waves of 128x32
        case 26: LAUNCH_V3(4, 4, 2, 4, 3, 3); break;          // 128x256, 8 waves of 64x64
        case 27: LAUNCH_V3(7, 3, 2, 2, 4, 4); break;          // 224x96,  4 waves of 112x48 (one per SIMD): the A/B partner of pick 30
        // (round 5 also instantiated other ring depths -- (2,3), (3,4) -- and 4-wave 112x128 / 224x128 tiles: measured flat / slower, removed in round 6)
        case 30: LAUNCH_V3K(7, 3, 2, 2, 4); break;                                                              // 224x96,  4 wave PAIRS of 112x48 splitting k (two waves per SIMD)
        // (224x64 on 4 wave pairs of 112x32 with TWO K slices for o_proj / down at M = 448: 34.4 / 67.3 us against 33.0 / 55.4 on 224x128 x 4 slices --
        //  905-1 107 cycles per k-tile for half the FLOPs of a 1 155-1 400-cycle tile; not kept: profiles/r06_v3_schedule_ab.txt)
        default: return hipErrorInvalidValue;
    }
#undef LAUNCH_V3
#undef LAUNCH_V3K
    if (e == hipSuccess) e = hipGetLastError();
    return e;
}
