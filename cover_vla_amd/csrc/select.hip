// Selection kernels: action-token decode head and the verifier's fuse + score + grouped arg-max (K20).
// Index results must be bit-exact against the oracle, so every reduction that decides an index runs in a
// fixed order with "first maximum wins" ties (torch.max / np.argmax semantics).
#include "common.h"
#include "kernels.h"

// (value, index) arg-max with smallest-index tie break
__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}
__device__ __forceinline__ void block_argmax(float& v, int& i, float* sv, int* si) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(i, o);
        argmax_combine(v, i, ov, oi);
    }
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        sv[w] = v;
        si[w] = i;
    }
    __syncthreads();
    v = sv[0];
    i = si[0];
    for (int k = 1; k < nw; ++k) argmax_combine(v, i, sv[k], si[k]);
}

// one block per row
__global__ __launch_bounds__(256) void token_select_k(cover_token_select_args a) {
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ float probs[4096];
    const int row = blockIdx.x;
    const float* lg = a.logits + (size_t)row * a.ld;
    float v = -INFINITY;
    int idx = 0x7fffffff;
    for (int c = a.lo + threadIdx.x; c < a.hi; c += 256) argmax_combine(v, idx, lg[c], c);
    block_argmax(v, idx, sv, si);
    if (a.uniform == nullptr) {
        if (threadIdx.x == 0) {
            a.token_out[row] = idx;
            if (a.logit_out) a.logit_out[row] = v;
        }
        return;
    }
    // inverse-CDF sampling over [lo, hi) (width <= 4096): p_i = exp((l_i - max) / T), sequential fp32 cumsum in
    // index order, pick the first i with cumsum_i > u * total (the oracle does the same arithmetic in numpy).
    const int n = a.hi - a.lo;
    const float inv_t = 1.0f / a.temperature;
    for (int c = threadIdx.x; c < n; c += 256) probs[c] = expf((lg[a.lo + c] - v) * inv_t);
    __syncthreads();
    // sequential fp32 sums in index order (the oracle's arithmetic), but without a data-dependent exit inside the chain:
    // thread 0 leaves the running sums in LDS (loads pipeline, only the adds are serial), then everyone searches them
    __shared__ float csum[4096];
    __shared__ float tgt;
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int c = 0; c < n; ++c) total += probs[c];
        tgt = a.uniform[row] * total;
        float cs = 0.f;
        for (int c = 0; c < n; ++c) {
            cs += probs[c];
            csum[c] = cs;
        }
    }
    __syncthreads();
    int pick = n - 1;   // first index whose running sum exceeds the target (n - 1 if none does)
    for (int c = threadIdx.x; c < n; c += 256)
        if (csum[c] > tgt) { pick = c < pick ? c : pick; break; }
    {   // block-wide minimum of the per-thread first hits (max of the negated index; indices < 4096 are exact in fp32)
        float nv = -(float)pick;
        int pi = pick;
        block_argmax(nv, pi, sv, si);
        pick = (int)(-nv);
    }
    if (threadIdx.x == 0) {
        a.token_out[row] = a.lo + pick;
        if (a.logit_out) a.logit_out[row] = lg[a.lo + pick];
    }
}
// Greedy pick over a WIDE range (the pi0-FAST head: 257 152 logits per row, a handful of rows): one 1024-thread block per row,
// 16-byte loads, four of them in flight per lane. With the scalar loop above a row costs ~310 us (one dword per lane per
// dependent iteration); here ~15. Same value / smallest-index tie rule: the combine is associative and commutative.
__global__ __launch_bounds__(1024) void token_argmax_wide_k(cover_token_select_args a) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int row = blockIdx.x;
    const float* lg = a.logits + (size_t)row * a.ld;
    float v = -INFINITY;
    int idx = 0x7fffffff;
    const int n4 = (a.hi - a.lo) >> 2;
    const float4* p = (const float4*)(lg + a.lo);
    auto take = [&](const float4& x, int c) {
        const int i0 = a.lo + 4 * c;
        argmax_combine(v, idx, x.x, i0);
        argmax_combine(v, idx, x.y, i0 + 1);
        argmax_combine(v, idx, x.z, i0 + 2);
        argmax_combine(v, idx, x.w, i0 + 3);
    };
    int c = threadIdx.x;
    for (; c + 3 * 1024 < n4; c += 4 * 1024) {
        const float4 x0 = p[c], x1 = p[c + 1024], x2 = p[c + 2048], x3 = p[c + 3072];
        take(x0, c); take(x1, c + 1024); take(x2, c + 2048); take(x3, c + 3072);
    }
    for (; c < n4; c += 1024) take(p[c], c);
    for (int t = a.lo + 4 * n4 + threadIdx.x; t < a.hi; t += 1024) argmax_combine(v, idx, lg[t], t);
    block_argmax(v, idx, sv, si);
    if (threadIdx.x == 0) {
        a.token_out[row] = idx;
        if (a.logit_out) a.logit_out[row] = v;
    }
}
hipError_t launch_token_select(const cover_token_select_args* a, hipStream_t st) {
    if (a->rows <= 0) return hipSuccess;
    if (a->hi <= a->lo) return hipErrorInvalidValue;
    if (a->uniform && (a->hi - a->lo > 4096 || !(a->temperature > 0.f))) return hipErrorInvalidValue;
    const bool wide = a->uniform == nullptr && a->hi - a->lo >= 8192 && (a->lo & 3) == 0 && (a->ld & 3) == 0 &&
                      (((uintptr_t)a->logits) & 15) == 0;
    if (wide) hipLaunchKernelGGL(token_argmax_wide_k, dim3(a->rows), dim3(1024), 0, st, *a);
    else hipLaunchKernelGGL(token_select_k, dim3(a->rows), dim3(256), 0, st, *a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// K20 (efficient_ensemble_merged.py:404-448): mean over members -> renormalise -> IT . ACT^T -> [G, S] ->
// arg-max of group means -> arg-max within the group. Single block (N <= 4096 candidates).
// ---------------------------------------------------------------------------------------------------
__device__ void group_argmax_dev(const float* scores, int N, int gs, int* result, float* best, float* gmean_s,
                                 float* sv, int* si) {
    const int G = N / gs;
    for (int gi = threadIdx.x; gi < G; gi += blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < gs; ++j) s += scores[gi * gs + j];
        gmean_s[gi] = s / (float)gs;
    }
    __syncthreads();
    float v = -INFINITY;
    int idx = 0x7fffffff;
    for (int gi = threadIdx.x; gi < G; gi += blockDim.x) argmax_combine(v, idx, gmean_s[gi], gi);
    block_argmax(v, idx, sv, si);
    const int bg = idx;
    const float bgm = v;
    float v2 = -INFINITY;
    int i2 = 0x7fffffff;
    for (int j = threadIdx.x; j < gs; j += blockDim.x) argmax_combine(v2, i2, scores[bg * gs + j], j);
    block_argmax(v2, i2, sv, si);
    if (threadIdx.x == 0) {
        result[0] = bg * gs + i2;
        result[1] = bg;
        result[2] = i2;
        result[3] = 0;
        best[0] = v2;
        best[1] = bgm;
    }
}

// One 1024-thread block per 16 candidates (one wave per candidate); every block recomputes the fused image-text embedding (dim x members
// floats: nothing) with the arithmetic of the former single-block kernel, so scores and embeddings are what that kernel produced, bit for
// bit, for every N -- that kernel took 36 us at N = 32 on the tail of every decision and 480 us at N = 512 (16 waves x 32 candidates in
// series). The grouped arg-max is the launch behind it (group_argmax_k).
__global__ __launch_bounds__(1024) void score_rows_k(cover_score_select_args a, float* fit_ws, float* fact_ws) {
    __shared__ float red[16];
    __shared__ float sfit[4096];
    const int dim = a.dim;
    {
        float q = 0.f;
        for (int d = threadIdx.x; d < dim; d += blockDim.x) {
            float s = 0.f;
            for (int m = 0; m < a.n_members; ++m) s += a.it[(size_t)m * dim + d];
            s /= (float)a.n_members;
            sfit[d] = s;
            q += s * s;
        }
        const float nrm = sqrtf(block_sum(q, red));
        for (int d = threadIdx.x; d < dim; d += blockDim.x) {
            const float f = sfit[d] / nrm;
            sfit[d] = f;
            if (blockIdx.x == 0) fit_ws[d] = f;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = blockIdx.x * 16 + w;
    if (n >= a.N) return;
    float q = 0.f;
    for (int d = lane; d < dim; d += 64) {
        float s = 0.f;
        for (int m = 0; m < a.n_members; ++m) s += a.act[((size_t)m * a.N + n) * dim + d];
        s /= (float)a.n_members;
        fact_ws[(size_t)n * dim + d] = s;
        q += s * s;
    }
    const float nrm = sqrtf(wave_sum(q));
    float dot = 0.f;
    for (int d = lane; d < dim; d += 64) {
        const float f = fact_ws[(size_t)n * dim + d] / nrm;
        fact_ws[(size_t)n * dim + d] = f;
        dot += sfit[d] * f;
    }
    dot = wave_sum(dot);
    if (lane == 0) a.scores_out[n] = dot;
}
__global__ __launch_bounds__(256) void group_argmax_k(const float* scores, int N, int gs, int* result, float* best) {
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ float gmean[4096];
    group_argmax_dev(scores, N, gs, result, best, gmean, sv, si);
}
hipError_t launch_group_argmax(const float* scores, int N, int gs, int* result, float* best, hipStream_t st) {
    if (N <= 0 || gs <= 0 || N % gs != 0 || N / gs > 4096) return hipErrorInvalidValue;
    hipLaunchKernelGGL(group_argmax_k, dim3(1), dim3(256), 0, st, scores, N, gs, result, best);
    return hipGetLastError();
}
hipError_t launch_score_select(const cover_score_select_args* a, hipStream_t st) {
    if (a->N <= 0 || a->group_size <= 0 || a->N % a->group_size != 0 || a->N / a->group_size > 4096 || a->dim <= 0 || a->dim > 4096)
        return hipErrorInvalidValue;
    if (!a->fused_it_out || !a->fused_act_out) return hipErrorInvalidValue;
    hipLaunchKernelGGL(score_rows_k, dim3((a->N + 15) / 16), dim3(1024), 0, st, *a, a->fused_it_out, a->fused_act_out);
    hipLaunchKernelGGL(group_argmax_k, dim3(1), dim3(256), 0, st, (const float*)a->scores_out, a->N, a->group_size, a->result_out, a->best_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// tokens -> verifier histories on the device (one thread per (candidate, history row))
// ---------------------------------------------------------------------------------------------------
__global__ void tokens_to_histories_k(const int64_t* __restrict__ tokens, int ld_tokens, int N, int tok_vocab,
                                      const float* __restrict__ centers, int n_centers, const float* __restrict__ past,
                                      int n_past, int n_use, float pad_value, float* __restrict__ hist, uint8_t* __restrict__ pad) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * 10) return;
    const int n = idx / 10, t = idx - n * 10;
    const int n_pad = 10 - n_past - n_use;
    float* o = hist + (size_t)idx * 7;
    if (t < n_pad) {
        for (int d = 0; d < 7; ++d) o[d] = pad_value;
        pad[idx] = 1;
    } else if (t < n_pad + n_past) {
        const float* p = past + (size_t)(t - n_pad) * 7;
        for (int d = 0; d < 7; ++d) o[d] = p[d];
        pad[idx] = 0;
    } else {
        const int step = t - n_pad - n_past;          // the step-th 7-token action of the candidate's chunk (action-chunk horizon > 1)
        for (int d = 0; d < 7; ++d) {
            long long b = (long long)tok_vocab - tokens[(size_t)n * ld_tokens + step * 7 + d] - 1;
            b = b < 0 ? 0 : (b > n_centers - 1 ? n_centers - 1 : b);
            float a = centers[b];
            if (d == 6) a = a < 0.5f ? 0.f : 1.f;
            o[d] = a;
        }
        pad[idx] = 0;
    }
}
hipError_t launch_tokens_to_histories(const int64_t* tokens, int ld_tokens, int N, int tok_vocab, const float* centers,
                                      int n_centers, const float* past, int n_past, float pad_value, float* hist, uint8_t* pad,
                                      hipStream_t st, int n_use) {
    if (N <= 0) return hipSuccess;
    if (n_past < 0 || n_use < 1 || n_past + n_use > 10 || ld_tokens < 7 * n_use) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tokens_to_histories_k, dim3((N * 10 + 255) / 256), dim3(256), 0, st, tokens, ld_tokens, N, tok_vocab,
                       centers, n_centers, past, n_past, n_use, pad_value, hist, pad);
    return hipGetLastError();
}

// flow-matching chunks -> verifier histories on the device (one thread per (candidate, history row))
__global__ void actions_to_histories_k(const float* __restrict__ actions, long long n_stride, long long t_stride, int N, int n_use,
                                       const float* __restrict__ lo_hi, const float* __restrict__ past, int n_past,
                                       float pad_value, float* __restrict__ hist, uint8_t* __restrict__ pad) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * 10) return;
    const int n = idx / 10, t = idx - n * 10;
    const int n_pad = 10 - n_past - n_use;
    float* o = hist + (size_t)idx * 7;
    if (t < n_pad) {
        for (int d = 0; d < 7; ++d) o[d] = pad_value;
        pad[idx] = 1;
    } else if (t < n_pad + n_past) {
        const float* p = past + (size_t)(t - n_pad) * 7;
        for (int d = 0; d < 7; ++d) o[d] = p[d];
        pad[idx] = 0;
    } else {
        const float* a = actions + (size_t)n * n_stride + (size_t)(t - n_pad - n_past) * t_stride;
        for (int d = 0; d < 6; ++d) {
            float v = a[d];
            if (lo_hi) v = (v - (-1.0f)) / (1.0f - (-1.0f)) * (lo_hi[6 + d] - lo_hi[d]) + lo_hi[d];   // denormalize_bound
            o[d] = v;
        }
        o[6] = a[6] < 0.5f ? 0.f : 1.f;
        pad[idx] = 0;
    }
}
hipError_t launch_actions_to_histories(const float* actions, long long n_stride, long long t_stride, int N, int n_use,
                                       const float* lo_hi, const float* past, int n_past, float pad_value, float* hist,
                                       uint8_t* pad, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    if (n_past < 0 || n_use < 1 || n_past + n_use > 10) return hipErrorInvalidValue;
    hipLaunchKernelGGL(actions_to_histories_k, dim3((N * 10 + 255) / 256), dim3(256), 0, st, actions, n_stride, t_stride, N, n_use,
                       lo_hi, past, n_past, pad_value, hist, pad);
    return hipGetLastError();
}
