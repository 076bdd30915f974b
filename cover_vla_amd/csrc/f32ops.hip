// fp32 kernels for the parts of the path the reference keeps in float32: the verifier heads
// (bridge_verifier/ensemble_eval/model.py:7-112, efficient_ensemble_merged.py:194-247) and the pi0 suffix
// projections / Euler update (modeling_pi0.py:569-629,748-751,713-714).
// The GEMM runs on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, k-ordered), everything else is VALU.
#include "common.h"
#include <stdlib.h>
#include "kernels.h"

// ---------------------------------------------------------------------------------------------------
// C[m,n] = residual + alpha * act(sum_k A[m,k] B[n,k] + bias[n])     (generic strides, optional batch)
// ---------------------------------------------------------------------------------------------------
// Tile = TM x TN (64x64 or 32x32), BK = 32, 4 waves (2x2), each wave (TM/32)x(TN/32) MFMA fragments. The next k-tile is
// fetched into registers while the current one is multiplied (these GEMMs have tiny grids, so nothing else hides the
// global-load latency).
template <int TM, int TN, int FK>
__global__ __launch_bounds__(256) void gemm_f32_k(cover_gemm_f32_args a) {
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int EA = TM * FK / 256, EB = TN * FK / 256;  // elements per thread per tile
    __shared__ float As[TM][FK + 1];
    __shared__ float Bs[TN][FK + 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int bz = blockIdx.z;
    const float* A = a.A + (size_t)bz * a.a_batch_stride;
    const float* B = a.B + (size_t)bz * a.b_batch_stride;
    float* C = a.C + (size_t)bz * a.c_batch_stride;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // thread -> (row, k-run) maps: unit-stride axis contiguous across threads
    const bool a_rowmajor = !(a.a_row_stride == 1 && a.a_k_stride != 1);
    const bool b_rowmajor = !(a.b_row_stride == 1 && a.b_k_stride != 1);
    const int ar = a_rowmajor ? tid / (FK / EA) : tid % TM, ak = a_rowmajor ? (tid % (FK / EA)) * EA : (tid / TM) * EA;
    const int br = b_rowmajor ? tid / (FK / EB) : tid % TN, bk = b_rowmajor ? (tid % (FK / EB)) * EB : (tid / TN) * EB;
    float ra[EA], rb[EB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int k = k0 + ak + e, rr = m0 + ar;
            ra[e] = (rr < a.M && k < a.K) ? A[(size_t)rr * a.a_row_stride + (size_t)k * a.a_k_stride] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int k = k0 + bk + e, rr = n0 + br;
            rb[e] = (rr < a.N && k < a.K) ? B[(size_t)rr * a.b_row_stride + (size_t)k * a.b_k_stride] : 0.f;
        }
    };
    const int r = lane & 15, g = lane >> 4;
    fetch(0);
    for (int k0 = 0; k0 < a.K; k0 += FK) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EA; ++e) As[ar][ak + e] = ra[e];
#pragma unroll
        for (int e = 0; e < EB; ++e) Bs[br][bk + e] = rb[e];
        __syncthreads();
        if (k0 + FK < a.K) fetch(k0 + FK);
#pragma unroll
        for (int k4 = 0; k4 < FK / 4; ++k4) {
            float af[FM], bf[FN];
#pragma unroll
            for (int f = 0; f < FM; ++f) af[f] = As[wm * (TM / 2) + f * 16 + r][k4 * 4 + g];
#pragma unroll
            for (int f = 0; f < FN; ++f) bf[f] = Bs[wn * (TN / 2) + f * 16 + r][k4 * 4 + g];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // D[row = 4g + e][col = r]
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm * (TM / 2) + i * 16 + 4 * g + e;
                const int n = n0 + wn * (TN / 2) + j * 16 + r;
                if (m < a.M && n < a.N) {
                    float v = acc[i][j][e];
                    if (a.bias) v += a.bias[(size_t)bz * a.bias_batch_stride + n];
                    v = act_apply(v, a.act);
                    v *= a.alpha;
                    if (a.residual) v += a.residual[(size_t)bz * a.c_batch_stride + (size_t)m * a.ld_residual + n];
                    C[(size_t)m * a.c_row_stride + n] = v;
                }
            }
}
// Small-grid variant for k-contiguous operands (every nn.Linear of the verifier heads): no LDS staging and no barrier in
// the k loop. A block owns a (16*FM) x 32 output tile, its 4 waves split K in interleaved 16-wide steps and keep UNR steps
// of 16-byte loads in flight each; lane (r, g) loads k = 16s + 4g .. 4g+3 of its row, and the j-th of four MFMAs per step
// takes component j from both operands (the 16x16x4 MFMA sums over the four lane groups, so any k <-> (g, j) bijection
// that is the same for A and B is valid). The four partial tiles meet in LDS and leave through the usual epilogue.
// UNR = k-steps a wave keeps in flight per round trip: the kernel is a chain of dependent load rounds (a 2048-deep contraction is 32 steps
// per wave: 8 rounds at UNR = 4, 4 at UNR = 8); the order in which a wave adds its steps does not depend on UNR (same sums, bit for bit).
template <int FM, int UNR = 4>
__global__ __launch_bounds__(256) void gemm_f32_direct_k(cover_gemm_f32_args a) {
    __shared__ float red[4][FM * 2][4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * (16 * FM), n0 = blockIdx.x * 32;
    const int bz = blockIdx.z;
    const float* A = a.A + (size_t)bz * a.a_batch_stride;
    const float* B = a.B + (size_t)bz * a.b_batch_stride;
    float* C = a.C + (size_t)bz * a.c_batch_stride;
    const float* ap[FM];
    const float* bp[2];
#pragma unroll
    for (int f = 0; f < FM; ++f) {
        int m = m0 + f * 16 + r;
        m = m < a.M ? m : a.M - 1;
        ap[f] = A + (size_t)m * a.a_row_stride + 4 * g;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int n = n0 + j * 16 + r;
        n = n < a.N ? n : a.N - 1;
        bp[j] = B + (size_t)n * a.b_row_stride + 4 * g;
    }
    f32x4 acc[FM][2];
#pragma unroll
    for (int f = 0; f < FM; ++f)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nsteps = a.K >> 4;
    for (int s0 = w; s0 < nsteps; s0 += 4 * UNR) {
        float4 av[UNR][FM], bv[UNR][2];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int s = s0 + 4 * u;
            const int ko = (s < nsteps ? s : s0) * 16;   // out-of-range steps re-read a valid one and are skipped below
#pragma unroll
            for (int f = 0; f < FM; ++f) av[u][f] = *(const float4*)(ap[f] + ko);
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[u][j] = *(const float4*)(bp[j] + ko);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (s0 + 4 * u < nsteps) {
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][f].x, bv[u][j].x, acc[f][j], 0, 0, 0);
                        acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][f].y, bv[u][j].y, acc[f][j], 0, 0, 0);
                        acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][f].z, bv[u][j].z, acc[f][j], 0, 0, 0);
                        acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][f].w, bv[u][j].w, acc[f][j], 0, 0, 0);
                    }
            }
        }
    }
#pragma unroll
    for (int f = 0; f < FM; ++f)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[w][f * 2 + j][e][lane] = acc[f][j][e];
    __syncthreads();
    // D[row = 4g + e][col = r]; thread (e = tid >> 6, lane) finishes element e of every fragment
    const int e = tid >> 6;
#pragma unroll
    for (int q = 0; q < FM * 2; ++q) {
        float v = red[0][q][e][lane] + red[1][q][e][lane];
        v += red[2][q][e][lane];
        v += red[3][q][e][lane];
        const int m = m0 + (q >> 1) * 16 + 4 * g + e;
        const int n = n0 + (q & 1) * 16 + r;
        if (m < a.M && n < a.N) {
            if (a.bias) v += a.bias[(size_t)bz * a.bias_batch_stride + n];
            v = act_apply(v, a.act);
            v *= a.alpha;
            if (a.residual) v += a.residual[(size_t)bz * a.c_batch_stride + (size_t)m * a.ld_residual + n];
            C[(size_t)m * a.c_row_stride + n] = v;
        }
    }
}

hipError_t launch_gemm_f32(const cover_gemm_f32_args* a, hipStream_t st) {
    if (a->M <= 0 || a->N <= 0) return hipSuccess;
    const int nb = a->batch > 0 ? a->batch : 1;
    const long long blocks64 = (long long)((a->N + 63) / 64) * ((a->M + 63) / 64) * nb;
    const bool k_contig = a->a_k_stride == 1 && a->b_k_stride == 1 && (a->K & 15) == 0 && a->K >= 64 && (a->a_row_stride & 3) == 0 &&
                          (a->b_row_stride & 3) == 0 && (a->a_batch_stride & 3) == 0 && (a->b_batch_stride & 3) == 0 &&
                          (((uintptr_t)a->A | (uintptr_t)a->B) & 15) == 0;
    // (the direct MFMA kernel reads its operands straight from global: fine while the grid is small or the operands stay L2-resident;
    //  COVER_F32_DIRECT_MAX overrides the block-count bound, read per call: A/B runs)
    const char* dmax_env = getenv("COVER_F32_DIRECT_MAX");
    const long long direct_max = dmax_env ? atoll(dmax_env) : 16384;   // (1024 until round 4: config 5's 5120-row trajectory GEMMs 235 -> 149 us, 292 -> 191 us)
    if (k_contig && blocks64 < direct_max) {
        // experiment knob COVER_F32_UNR=8: eight steps in flight for K >= 512. Measured (round 4): the verifier tail got SLOWER, 0.563 vs 0.525 ms
        // -- the deeper window costs occupancy (126 registers: 3 waves per SIMD instead of 6) and these grids live off occupancy. Default 4.
        static const char* unr_env = getenv("COVER_F32_UNR");
        const bool deep = a->K >= 512 && unr_env && unr_env[0] == '8';
        if (a->M <= 16) {
            dim3 grid((a->N + 31) / 32, (a->M + 15) / 16, nb);
            if (deep) hipLaunchKernelGGL((gemm_f32_direct_k<1, 8>), grid, dim3(256), 0, st, *a);
            else hipLaunchKernelGGL((gemm_f32_direct_k<1, 4>), grid, dim3(256), 0, st, *a);
        } else {
            dim3 grid((a->N + 31) / 32, (a->M + 31) / 32, nb);
            if (deep) hipLaunchKernelGGL((gemm_f32_direct_k<2, 8>), grid, dim3(256), 0, st, *a);
            else hipLaunchKernelGGL((gemm_f32_direct_k<2, 4>), grid, dim3(256), 0, st, *a);
        }
        return hipGetLastError();
    }
    if (blocks64 >= 256) {
        dim3 grid((a->N + 63) / 64, (a->M + 63) / 64, nb);
        hipLaunchKernelGGL((gemm_f32_k<64, 64, 32>), grid, dim3(256), 0, st, *a);
    } else if (a->K >= 256) {
        // small grids are bound by one global-load latency per k-step: a 128-deep step puts 4x the loads in flight
        dim3 grid((a->N + 31) / 32, (a->M + 31) / 32, nb);
        hipLaunchKernelGGL((gemm_f32_k<32, 32, 128>), grid, dim3(256), 0, st, *a);
    } else {
        dim3 grid((a->N + 31) / 32, (a->M + 31) / 32, nb);
        hipLaunchKernelGGL((gemm_f32_k<32, 32, 32>), grid, dim3(256), 0, st, *a);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// row kernels (fp32)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_f32_k(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ y, int ldy,
                                                       int dim, float eps, int rows_per_group, long long wb_group_stride) {
    __shared__ float red[16];
    {   // stacked row groups with their own affine parameters
        const long long go = (long long)(blockIdx.x / rows_per_group) * wb_group_stride;
        if (w) w += go;
        if (b) b += go;
    }
    const float* xr = x + (size_t)blockIdx.x * ldx;
    float s = 0.f;
    for (int c = threadIdx.x; c < dim; c += 256) s += xr[c];
    const float mean = block_sum(s, red) / dim;
    float q = 0.f;
    for (int c = threadIdx.x; c < dim; c += 256) {
        const float d = xr[c] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(block_sum(q, red) / dim + eps);
    float* yr = y + (size_t)blockIdx.x * ldy;
    for (int c = threadIdx.x; c < dim; c += 256) yr[c] = (xr[c] - mean) * rstd * (w ? w[c] : 1.f) + (b ? b[c] : 0.f);
}
hipError_t launch_layernorm_f32(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int rows,
                                int dim, float eps, hipStream_t st, int rows_per_group, long long wb_group_stride) {
    if (rows <= 0) return hipSuccess;
    if (rows_per_group <= 0) rows_per_group = rows;
    hipLaunchKernelGGL(layernorm_f32_k, dim3(rows), dim3(256), 0, st, x, ldx, w, b, y, ldy, dim, eps, rows_per_group, wb_group_stride);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void softmax_rows_f32_k(float* __restrict__ x, int ldx, int cols, float scale) {
    __shared__ float red[16];
    float* xr = x + (size_t)blockIdx.x * ldx;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, xr[c] * scale);
    m = block_max(m, red);
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float e = expf(xr[c] * scale - m);
        xr[c] = e;
        s += e;
    }
    const float inv = 1.f / block_sum(s, red);
    for (int c = threadIdx.x; c < cols; c += 256) xr[c] *= inv;
}
hipError_t launch_softmax_rows_f32(float* x, int ldx, int rows, int cols, float scale, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(softmax_rows_f32_k, dim3(rows), dim3(256), 0, st, x, ldx, cols, scale);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void l2norm_rows_f32_k(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                         int ldy, int cols) {
    __shared__ float red[16];
    const float* xr = x + (size_t)blockIdx.x * ldx;
    float q = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) q += xr[c] * xr[c];
    const float nrm = sqrtf(block_sum(q, red));
    float* yr = y + (size_t)blockIdx.x * ldy;
    for (int c = threadIdx.x; c < cols; c += 256) yr[c] = xr[c] / nrm;
}
hipError_t launch_l2norm_rows_f32(const float* x, int ldx, float* y, int ldy, int rows, int cols, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(l2norm_rows_f32_k, dim3(rows), dim3(256), 0, st, x, ldx, y, ldy, cols);
    return hipGetLastError();
}

// Cross entropy of each row against the label on its diagonal (label of row r = r), and the rank of that label's logit in its
// row (number of strictly larger logits, plus equal logits at a lower column): the retrieval statistics of the verifier's
// contrastive evaluation (finetune_trajectory_bridge_ddp.py:446-469, :1081-1090).
__global__ __launch_bounds__(256) void xent_diag_f32_k(const float* __restrict__ x, int ldx, int cols, float* __restrict__ loss,
                                                       int* __restrict__ rank) {
    __shared__ float red[16];
    const int r = blockIdx.x;
    const float* xr = x + (size_t)r * ldx;
    const float d = xr[r];
    float m = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, xr[c]);
    m = block_max(m, red);
    float s = 0.f, above = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float v = xr[c];
        s += expf(v - m);
        above += (v > d || (v == d && c < r)) ? 1.f : 0.f;
    }
    s = block_sum(s, red);
    above = block_sum(above, red);
    if (threadIdx.x == 0) {
        loss[r] = (m + logf(s)) - d;
        rank[r] = (int)above;
    }
}
hipError_t launch_xent_diag_f32(const float* x, int ldx, int rows, int cols, float* loss, int* rank, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(xent_diag_f32_k, dim3(rows), dim3(256), 0, st, x, ldx, cols, loss, rank);
    return hipGetLastError();
}

__global__ void add_f32_k(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                          float* __restrict__ y, int ldy, int cols, int b_rows) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += blockDim.x)
        y[(size_t)r * ldy + c] = a[(size_t)r * lda + c] + b[(size_t)(r % b_rows) * ldb + c];
}
hipError_t launch_add_f32(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int rows, int cols,
                          int b_rows, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(add_f32_k, dim3(rows), dim3(256), 0, st, a, lda, b, ldb, y, ldy, cols, b_rows);
    return hipGetLastError();
}

__global__ void act_f32_k(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int cols, int act) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) y[(size_t)r * ldy + c] = act_apply(x[(size_t)r * ldx + c], act);
}
hipError_t launch_act_f32(const float* x, int ldx, float* y, int ldy, int rows, int cols, int act, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(act_f32_k, dim3(rows), dim3(256), 0, st, x, ldx, y, ldy, cols, act);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// small multi-head attention: one block per (batch, head); Tq*Tk <= 4096, Dh <= 128
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mha_f32_k(cover_mha_f32_args a) {
    extern __shared__ float sc[];  // [Tq][Tk]
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const float* q = a.q + (size_t)b * a.q_b_stride + (size_t)h * a.Dh;
    const float* k = a.k + (size_t)b * a.k_b_stride + (size_t)h * a.Dh;
    const float* v = a.v + (size_t)b * a.v_b_stride + (size_t)h * a.Dh;
    const uint8_t* pad = a.key_pad ? a.key_pad + (size_t)b * a.Tk : nullptr;
    const int n = a.Tq * a.Tk;
    const bool vec4 = (a.Dh & 3) == 0 && ((a.q_t_stride | a.k_t_stride | a.q_b_stride | a.k_b_stride) & 3) == 0 &&
                      (((uintptr_t)a.q | (uintptr_t)a.k) & 15) == 0;
    for (int idx = threadIdx.x; idx < n; idx += 256) {
        const int i = idx / a.Tk, j = idx - i * a.Tk;
        const float* qi = q + (size_t)i * a.q_t_stride;
        const float* kj = k + (size_t)j * a.k_t_stride;
        float s = 0.f;
        if (vec4) {   // 16-byte loads, eight of them in flight; the sum runs in the same d order as the scalar loop
#pragma unroll 4
            for (int d = 0; d < a.Dh; d += 4) {
                const float4 qv = *(const float4*)(qi + d), kv = *(const float4*)(kj + d);
                s += (qv.x * a.scale) * kv.x;
                s += (qv.y * a.scale) * kv.y;
                s += (qv.z * a.scale) * kv.z;
                s += (qv.w * a.scale) * kv.w;
            }
        } else {
            for (int d = 0; d < a.Dh; ++d) s += (qi[d] * a.scale) * kj[d];
        }
        sc[idx] = (pad && pad[j]) ? -INFINITY : s;
    }
    __syncthreads();
    // softmax per query row: one wave per row
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = w; i < a.Tq; i += 4) {
        float m = -INFINITY;
        for (int j = lane; j < a.Tk; j += 64) m = fmaxf(m, sc[i * a.Tk + j]);
        m = wave_max(m);
        float s = 0.f;
        for (int j = lane; j < a.Tk; j += 64) {
            const float e = (m == -INFINITY) ? 0.f : expf(sc[i * a.Tk + j] - m);
            sc[i * a.Tk + j] = e;
            s += e;
        }
        s = wave_sum(s);
        const float inv = 1.f / s;  // all-masked row -> NaN, as torch
        for (int j = lane; j < a.Tk; j += 64) sc[i * a.Tk + j] *= inv;
    }
    __syncthreads();
    float* o = a.out + (size_t)b * a.o_b_stride + (size_t)h * a.Dh;
    for (int idx = threadIdx.x; idx < a.Tq * a.Dh; idx += 256) {
        const int i = idx / a.Dh, d = idx - i * a.Dh;
        float acc = 0.f;
        int j = 0;
        for (; j + 8 <= a.Tk; j += 8) {   // eight value loads in flight (a rolled loop waits for each one), summed in key order
            float vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = v[(size_t)(j + u) * a.v_t_stride + d];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += sc[i * a.Tk + j + u] * vv[u];
        }
        for (; j < a.Tk; ++j) acc += sc[i * a.Tk + j] * v[(size_t)j * a.v_t_stride + d];
        o[(size_t)i * a.o_t_stride + d] = acc;
    }
}
hipError_t launch_mha_f32(const cover_mha_f32_args* a, hipStream_t st) {
    if (a->B <= 0) return hipSuccess;
    if ((long long)a->Tq * a->Tk > 8192) return hipErrorInvalidValue;
    const size_t lds = (size_t)a->Tq * a->Tk * sizeof(float);
    hipLaunchKernelGGL(mha_f32_k, dim3(a->B * a->H), dim3(256), lds, st, *a);
    return hipGetLastError();
}

__global__ void masked_mean_f32_k(const float* __restrict__ x, const uint8_t* __restrict__ pad, float* __restrict__ y,
                                  int T, int D) {
    const int b = blockIdx.x;
    float cnt = 0.f;
    for (int t = 0; t < T; ++t) cnt += (pad && pad[(size_t)b * T + t]) ? 0.f : 1.f;
    cnt = fmaxf(cnt, 1e-9f);
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) {
            const float mk = (pad && pad[(size_t)b * T + t]) ? 0.f : 1.f;
            s += x[((size_t)b * T + t) * D + d] * mk;
        }
        y[(size_t)b * D + d] = s / cnt;
    }
}
hipError_t launch_masked_mean_f32(const float* x, const uint8_t* pad, float* y, int B, int T, int D, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(masked_mean_f32_k, dim3(B), dim3(256), 0, st, x, pad, y, T, D);
    return hipGetLastError();
}

// create_sinusoidal_pos_embedding (modeling_pi0.py:71-89) in float64, cast to bf16 (embed_suffix :593-596)
__global__ void sincos_time_embed_k(const float* __restrict__ time, int dim, double min_period, double max_period,
                                    bf16_t* __restrict__ out, int ldo) {
    const int b = blockIdx.x, half = dim >> 1;
    const double tt = (double)time[b];
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        const double fraction = half > 1 ? (double)i / (double)(half - 1) : 0.0;
        const double period = min_period * pow(max_period / min_period, fraction);
        const double arg = 1.0 / period * 2.0 * 3.141592653589793 * tt;
        out[(size_t)b * ldo + i] = f2bf((float)sin(arg));
        out[(size_t)b * ldo + half + i] = f2bf((float)cos(arg));
    }
}
hipError_t launch_sincos_time_embed(const float* time, int B, int dim, double min_period, double max_period, bf16_t* out,
                                    int ldo, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(sincos_time_embed_k, dim3(B), dim3(256), 0, st, time, dim, min_period, max_period,
                       out, ldo);
    return hipGetLastError();
}
