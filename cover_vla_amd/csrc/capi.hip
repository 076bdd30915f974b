// extern "C" surface of libcover_hip (include/cover_hip.h) + the composite tower / decoder forwards.
// Host code only; kernels live in the sibling .hip files.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include "kernels.h"

static thread_local std::string g_err;
static int fail(int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}
#define HIPCHK(call, what)                                  \
    do {                                                    \
        hipError_t _e = (call);                             \
        if (_e != hipSuccess) return fail(_e == hipErrorInvalidValue ? COVER_EINVAL : COVER_EHIP, what, _e); \
    } while (0)
#define ST(s) ((hipStream_t)(s))

extern "C" {

int cover_abi_version(void) { return COVER_ABI_VERSION; }
const char* cover_last_error(void) { return g_err.c_str(); }

int cover_device_info(int dev, int* n_cu, size_t* total_mem, char* name, int name_len) {
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, dev), "hipGetDeviceProperties");
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (total_mem) *total_mem = p.totalGlobalMem;
    if (name && name_len > 0) {
        strncpy(name, p.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    return COVER_OK;
}

int cover_packed_k(int K) { return (K + 127) / 128 * 128; }
size_t cover_packed_weight_bytes(int N, int K) {
    return (size_t)((N + 15) / 16) * 16 * (size_t)cover_packed_k(K) * 2;
}
int cover_pack_weight_bf16(const void* W, int ldw, int N, int K, void* Wp, int glu, void* stream) {
    if (!W || !Wp || N <= 0 || K <= 0) return fail(COVER_EINVAL, "cover_pack_weight_bf16: bad arguments");
    if (glu && ((N / 2) % 16 != 0 || (N & 1))) return fail(COVER_EINVAL, "cover_pack_weight_bf16: glu needs N/2 % 16 == 0");
    HIPCHK(launch_pack_weight_bf16((const bf16_t*)W, ldw, N, K, (bf16_t*)Wp, cover_packed_k(K), glu, ST(stream)),
           "pack_weight");
    return COVER_OK;
}
size_t cover_packed_weight_fp8_bytes(int N, int K) { return (size_t)((N + 15) / 16) * 16 * (size_t)cover_packed_k(K); }
int cover_quantize_rows_fp8(const void* W, int ldw, int N, int K, float* scales, void* Wdq, void* stream) {
    if (!W || !scales || !Wdq || N <= 0 || K <= 0) return fail(COVER_EINVAL, "cover_quantize_rows_fp8: bad arguments");
    HIPCHK(launch_quantize_rows_fp8((const bf16_t*)W, ldw, N, K, scales, (bf16_t*)Wdq, ST(stream)), "quantize_rows_fp8");
    return COVER_OK;
}
int cover_pack_weight_fp8(const void* Wdq, int ldw, const float* scales, int N, int K, void* Wq, float* scales_packed, int glu,
                          void* stream) {
    if (!Wdq || !scales || !Wq || !scales_packed || N <= 0 || K <= 0) return fail(COVER_EINVAL, "cover_pack_weight_fp8: bad arguments");
    if (glu && ((N / 2) % 16 != 0 || (N & 1))) return fail(COVER_EINVAL, "cover_pack_weight_fp8: glu needs N/2 % 16 == 0");
    HIPCHK(launch_pack_weight_fp8((const bf16_t*)Wdq, ldw, scales, N, K, (uint8_t*)Wq, scales_packed, cover_packed_k(K), glu, ST(stream)),
           "pack_weight_fp8");
    return COVER_OK;
}
int cover_pack_weight_fp8_klinear(const void* Wdq, int ldw, const float* scales, int N, int K, void* Wq, float* scales_packed, void* stream) {
    if (!Wdq || !scales || !Wq || !scales_packed || N <= 0 || K <= 0) return fail(COVER_EINVAL, "cover_pack_weight_fp8_klinear: bad arguments");
    HIPCHK(launch_pack_weight_fp8((const bf16_t*)Wdq, ldw, scales, N, K, (uint8_t*)Wq, scales_packed, cover_packed_k(K), 0, ST(stream), 1),
           "pack_weight_fp8 (k-linear)");
    return COVER_OK;
}
int cover_quantize_act_fp8_mx(const void* X, int ldx, int M, int K, void* out8, int ld8, void* mx, void* stream) {
    if (!X || !out8 || !mx || M < 0 || K <= 0) return fail(COVER_EINVAL, "cover_quantize_act_fp8_mx: bad arguments");
    HIPCHK(launch_quantize_act_fp8_mx((const bf16_t*)X, ldx, M, K, (uint8_t*)out8, ld8, (uint8_t*)mx, ST(stream)),
           "quantize_act_fp8_mx (ld8 >= padded K, ld8 % 16 == 0, ldx % 8 == 0)");
    return COVER_OK;
}
int cover_quantize_act_fp8(const void* X, int ldx, int M, int K, void* out8, int ld8, float* scales, void* stream) {
    if (!X || !out8 || !scales || M < 0 || K <= 0) return fail(COVER_EINVAL, "cover_quantize_act_fp8: bad arguments");
    HIPCHK(launch_quantize_act_fp8((const bf16_t*)X, ldx, M, K, (uint8_t*)out8, ld8, scales, ST(stream)),
           "quantize_act_fp8 (ld8 >= padded K, ld8 % 16 == 0, ldx % 8 == 0)");
    return COVER_OK;
}
size_t cover_gemm_workspace_bytes(int M, int N, int K) { return gemm_workspace_bytes(M, N, K); }
int cover_gemm_probe(unsigned long long* out) {
    if (!out) return fail(COVER_EINVAL, "cover_gemm_probe: null pointer");
    if (gemm_v3_probe(out) != 0) return fail(COVER_EHIP, "cover_gemm_probe: could not read the probe words");
    return COVER_OK;
}
int cover_gemm_plan_counts(long long* counts, int n, int reset) {
    if (n < 0 || (n > 0 && !counts)) return fail(COVER_EINVAL, "cover_gemm_plan_counts: bad arguments");
    gemm_plan_counts(counts, n, reset);
    return COVER_GEMM_PLANS;
}
int cover_gemm_bf16(const void* A, int lda, const void* Wp, void* C, int ldc, int M, int N, int K,
                    const cover_gemm_epi* epi, void* ws, size_t ws_bytes, int variant, void* stream) {
    if (!A || !Wp || !C) return fail(COVER_EINVAL, "cover_gemm_bf16: null pointer");
    if (lda % 8) return fail(COVER_EINVAL, "cover_gemm_bf16: lda must be a multiple of 8 elements");
    if (lda < cover_packed_k(K)) return fail(COVER_EINVAL, "cover_gemm_bf16: lda < padded K");
    if (epi && epi->glu && (N % 32)) return fail(COVER_EINVAL, "cover_gemm_bf16: glu needs N % 32 == 0");
    HIPCHK(launch_gemm_bf16((const bf16_t*)A, lda, (const bf16_t*)Wp, C, ldc, M, N, K, epi, (float*)ws, ws_bytes, variant,
                            ST(stream)),
           "gemm_bf16");
    return COVER_OK;
}

int cover_attention_bf16(const cover_attn_args* a, void* stream) {
    if (!a) return fail(COVER_EINVAL, "cover_attention_bf16: null args");
    HIPCHK(launch_attention_bf16(a, ST(stream)), "attention_bf16 (D must be 64/96/128/256, 1..3 segments)");
    return COVER_OK;
}

int cover_decode_attention_fused(const cover_decode_attn_args* a, void* stream) {
    if (!a || !a->out || (a->n_splits <= 0 && !a->qkv) || (a->n_splits > 0 && !a->partial))
        return fail(COVER_EINVAL, "cover_decode_attention_fused: null pointer");
    HIPCHK(launch_decode_attention_fused(a, ST(stream)), "decode_attention_fused (D in {64,128}, COVER_MASK_LEN segments, 0 <= write_t < seg[2].len)");
    return COVER_OK;
}

int cover_decode_own_attention(const cover_own_attn_args* a, void* stream) {
    if (!a) return fail(COVER_EINVAL, "cover_decode_own_attention: null args");
    HIPCHK(launch_decode_own_attention(a, ST(stream)), "decode_own_attention (D in {64,128}, 0 <= write_t < t_cap, at most 16 loads per lane per pass)");
    return COVER_OK;
}

int cover_layernorm_bf16(const void* x, int ldx, const float* w, const float* b, void* y, int ldy, int rows, int dim,
                         float eps, void* stream) {
    HIPCHK(launch_layernorm_bf16((const bf16_t*)x, ldx, w, b, (bf16_t*)y, ldy, rows, dim, eps, ST(stream)), "layernorm_bf16");
    return COVER_OK;
}
int cover_rmsnorm_bf16(const void* x, int x_f32, int ldx, const float* w, float w_offset, int style, void* y, int ldy,
                       int rows, int dim, float eps, void* stream) {
    HIPCHK(launch_rmsnorm(x, x_f32, ldx, w, w_offset, style, (bf16_t*)y, ldy, rows, dim, eps, ST(stream)), "rmsnorm_bf16");
    return COVER_OK;
}
int cover_rope_kv_write(const cover_rope_args* a, void* stream) {
    if (!a) return fail(COVER_EINVAL, "cover_rope_kv_write: null args");
    HIPCHK(launch_rope_kv_write(a, ST(stream)), "rope_kv_write");
    return COVER_OK;
}
int cover_embed_gather(const void* table, int dim, const int64_t* ids, int n, float scale, void* out, int ldo,
                       void* stream) {
    HIPCHK(launch_embed_gather((const bf16_t*)table, dim, ids, n, scale, (bf16_t*)out, ldo, ST(stream)), "embed_gather");
    return COVER_OK;
}
int cover_patchify(const cover_patchify_args* a, void* stream) {
    if (!a || a->patch <= 0 || a->H % a->patch || a->W % a->patch || a->ld_out < 3 * a->patch * a->patch)
        return fail(COVER_EINVAL, "cover_patchify: bad geometry");
    HIPCHK(launch_patchify(a, ST(stream)), "patchify");
    return COVER_OK;
}
int cover_copy_rows_bf16(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, const int* sidx,
                         const int* didx, void* stream) {
    HIPCHK(launch_copy_rows_bf16((const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, rows, cols, sidx, didx, ST(stream)),
           "copy_rows");
    return COVER_OK;
}
int cover_add_rows_bf16(void* x, int ldx, const void* add, int ld_add, int rows, int cols, int add_rows, void* stream) {
    if (add_rows <= 0) return fail(COVER_EINVAL, "cover_add_rows_bf16: add_rows <= 0");
    HIPCHK(launch_add_bias_rows_bf16((bf16_t*)x, ldx, (const bf16_t*)add, ld_add, rows, cols, add_rows, ST(stream)), "add_rows");
    return COVER_OK;
}
int cover_scale_bf16(void* x, int ldx, int rows, int cols, float pre_div, float post_mul, void* stream) {
    HIPCHK(launch_scale_bf16((bf16_t*)x, ldx, rows, cols, pre_div, post_mul, ST(stream)), "scale_bf16");
    return COVER_OK;
}
int cover_cast_f32_to_bf16(const float* x, int ldx, void* y, int ldy, int rows, int cols, void* stream) {
    HIPCHK(launch_cast_f32_to_bf16(x, ldx, (bf16_t*)y, ldy, rows, cols, ST(stream)), "cast_f32_to_bf16");
    return COVER_OK;
}
int cover_cast_bf16_to_f32(const void* x, int ldx, float* y, int ldy, int rows, int cols, void* stream) {
    HIPCHK(launch_cast_bf16_to_f32((const bf16_t*)x, ldx, y, ldy, rows, cols, ST(stream)), "cast_bf16_to_f32");
    return COVER_OK;
}

int cover_gemm_f32(const cover_gemm_f32_args* a, void* stream) {
    if (!a || !a->A || !a->B || !a->C) return fail(COVER_EINVAL, "cover_gemm_f32: null pointer");
    HIPCHK(launch_gemm_f32(a, ST(stream)), "gemm_f32");
    return COVER_OK;
}
int cover_layernorm_f32(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int rows, int dim,
                        float eps, void* stream) {
    HIPCHK(launch_layernorm_f32(x, ldx, w, b, y, ldy, rows, dim, eps, ST(stream)), "layernorm_f32");
    return COVER_OK;
}
int cover_layernorm_f32_grouped(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int rows, int dim,
                                float eps, int rows_per_group, long long wb_group_stride, void* stream) {
    if (rows_per_group <= 0) return fail(COVER_EINVAL, "cover_layernorm_f32_grouped: rows_per_group");
    HIPCHK(launch_layernorm_f32(x, ldx, w, b, y, ldy, rows, dim, eps, ST(stream), rows_per_group, wb_group_stride), "layernorm_f32_grouped");
    return COVER_OK;
}
int cover_softmax_rows_f32(float* x, int ldx, int rows, int cols, float scale, void* stream) {
    HIPCHK(launch_softmax_rows_f32(x, ldx, rows, cols, scale, ST(stream)), "softmax_rows_f32");
    return COVER_OK;
}
int cover_l2norm_rows_f32(const float* x, int ldx, float* y, int ldy, int rows, int cols, void* stream) {
    HIPCHK(launch_l2norm_rows_f32(x, ldx, y, ldy, rows, cols, ST(stream)), "l2norm_rows_f32");
    return COVER_OK;
}
int cover_add_f32(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int rows, int cols, int b_rows,
                  void* stream) {
    if (b_rows <= 0) return fail(COVER_EINVAL, "cover_add_f32: b_rows <= 0");
    HIPCHK(launch_add_f32(a, lda, b, ldb, y, ldy, rows, cols, b_rows, ST(stream)), "add_f32");
    return COVER_OK;
}
int cover_xent_diag_f32(const float* logits, int ld, int rows, int cols, float* loss, int* rank, void* stream) {
    if (!logits || !loss || !rank) return fail(COVER_EINVAL, "cover_xent_diag_f32: null pointer");
    if (rows > cols) return fail(COVER_EINVAL, "cover_xent_diag_f32: row r is labelled r, so rows <= cols");
    HIPCHK(launch_xent_diag_f32(logits, ld, rows, cols, loss, rank, ST(stream)), "xent_diag_f32");
    return COVER_OK;
}
int cover_act_f32(const float* x, int ldx, float* y, int ldy, int rows, int cols, int act, void* stream) {
    if (!x || !y) return fail(COVER_EINVAL, "cover_act_f32: null pointer");
    HIPCHK(launch_act_f32(x, ldx, y, ldy, rows, cols, act, ST(stream)), "act_f32");
    return COVER_OK;
}
int cover_mha_f32(const cover_mha_f32_args* a, void* stream) {
    if (!a) return fail(COVER_EINVAL, "cover_mha_f32: null args");
    HIPCHK(launch_mha_f32(a, ST(stream)), "mha_f32 (Tq*Tk <= 8192)");
    return COVER_OK;
}
int cover_masked_mean_f32(const float* x, const uint8_t* pad, float* y, int B, int T, int D, void* stream) {
    HIPCHK(launch_masked_mean_f32(x, pad, y, B, T, D, ST(stream)), "masked_mean_f32");
    return COVER_OK;
}
int cover_sincos_time_embed(const float* time, int B, int dim, double min_period, double max_period, void* out, int ldo,
                            void* stream) {
    if (dim % 2) return fail(COVER_EINVAL, "cover_sincos_time_embed: dimension must be divisible by 2");
    HIPCHK(launch_sincos_time_embed(time, B, dim, min_period, max_period, (bf16_t*)out, ldo, ST(stream)), "sincos_time_embed");
    return COVER_OK;
}

int cover_token_select(const cover_token_select_args* a, void* stream) {
    if (!a) return fail(COVER_EINVAL, "cover_token_select: null args");
    HIPCHK(launch_token_select(a, ST(stream)), "token_select (sampling width <= 4096, temperature > 0)");
    return COVER_OK;
}
int cover_score_select(const cover_score_select_args* a, void* stream) {
    if (!a) return fail(COVER_EINVAL, "cover_score_select: null args");
    HIPCHK(launch_score_select(a, ST(stream)), "score_select (N % group_size == 0, fused_*_out required)");
    return COVER_OK;
}
int cover_tokens_to_histories(const int64_t* tokens, int ld_tokens, int N, int tok_vocab, const float* centers, int n_centers,
                              const float* past, int n_past, float pad_value, float* hist_out, uint8_t* pad_out, void* stream) {
    if (!tokens || !centers || !hist_out || !pad_out || (n_past > 0 && !past)) return fail(COVER_EINVAL, "cover_tokens_to_histories: null pointer");
    HIPCHK(launch_tokens_to_histories(tokens, ld_tokens, N, tok_vocab, centers, n_centers, past, n_past, pad_value, hist_out, pad_out, ST(stream)),
           "tokens_to_histories (0 <= n_past <= 9)");
    return COVER_OK;
}
int cover_tokens_to_histories_steps(const int64_t* tokens, int ld_tokens, int N, int tok_vocab, const float* centers, int n_centers,
                                    const float* past, int n_past, int n_use, float pad_value, float* hist_out, uint8_t* pad_out, void* stream) {
    if (!tokens || !centers || !hist_out || !pad_out || (n_past > 0 && !past)) return fail(COVER_EINVAL, "cover_tokens_to_histories_steps: null pointer");
    HIPCHK(launch_tokens_to_histories(tokens, ld_tokens, N, tok_vocab, centers, n_centers, past, n_past, pad_value, hist_out, pad_out, ST(stream), n_use),
           "tokens_to_histories_steps (n_use >= 1, n_past + n_use <= 10, ld_tokens >= 7 n_use)");
    return COVER_OK;
}
int cover_actions_to_histories(const float* actions, long long n_stride, long long t_stride, int N, int n_use, const float* lo_hi,
                               const float* past, int n_past, float pad_value, float* hist_out, uint8_t* pad_out, void* stream) {
    if (!actions || !hist_out || !pad_out || (n_past > 0 && !past)) return fail(COVER_EINVAL, "cover_actions_to_histories: null pointer");
    HIPCHK(launch_actions_to_histories(actions, n_stride, t_stride, N, n_use, lo_hi, past, n_past, pad_value, hist_out, pad_out, ST(stream)),
           "actions_to_histories (n_use >= 1, n_past + n_use <= 10)");
    return COVER_OK;
}
int cover_group_argmax(const float* scores, int N, int group_size, int* result_out, float* best_out, void* stream) {
    HIPCHK(launch_group_argmax(scores, N, group_size, result_out, best_out, ST(stream)), "group_argmax");
    return COVER_OK;
}

int cover_resample_axis(const void* in, int in_is_f32, void* out, int out_is_f32, int Hin, int Win, int C, int Hout, int Wout,
                        int axis, const int* bounds, const void* coefs, int ksize, int fixed_point, void* stream) {
    if (!in || !out || !bounds || !coefs || ksize <= 0 || (axis != 0 && axis != 1)) return fail(COVER_EINVAL, "cover_resample_axis: bad arguments");
    if ((axis == 0 && Win != Wout) || (axis == 1 && Hin != Hout)) return fail(COVER_EINVAL, "cover_resample_axis: the other axis must keep its size");
    HIPCHK(launch_resample_axis(in, in_is_f32, out, out_is_f32, Hin, Win, C, Hout, Wout, axis, bounds, coefs, ksize, fixed_point, ST(stream)),
           "resample_axis (fixed point: uint8 -> uint8 only)");
    return COVER_OK;
}
int cover_u8_hwc_to_f32_chw_norm(const uint8_t* in, float* out, int H, int W, const float* mean3, const float* std3, void* stream) {
    if (!in || !out || !mean3 || !std3) return fail(COVER_EINVAL, "cover_u8_hwc_to_f32_chw_norm: null pointer");
    HIPCHK(launch_u8_to_chw_norm(in, out, H, W, 3, mean3, std3, ST(stream)), "u8_hwc_to_f32_chw_norm");
    return COVER_OK;
}
int cover_u8_hwc_to_f32_chw_scale_norm(const uint8_t* in, float* out, int H, int W, float scale, const float* mean3, const float* std3, void* stream) {
    if (!in || !out || !mean3 || !std3) return fail(COVER_EINVAL, "cover_u8_hwc_to_f32_chw_scale_norm: null pointer");
    HIPCHK(launch_u8_to_chw_scale_norm(in, out, H, W, 3, scale, mean3, std3, ST(stream)), "u8_hwc_to_f32_chw_scale_norm");
    return COVER_OK;
}
int cover_resize_bilinear_pad_f32(const float* in, float* out, int NC, int Hin, int Win, int Hr, int Wr, int Hout, int Wout, int pad_top,
                                  int pad_left, float pad_value, void* stream) {
    if (!in || !out) return fail(COVER_EINVAL, "cover_resize_bilinear_pad_f32: null pointer");
    HIPCHK(launch_bilinear_pad(in, out, NC, Hin, Win, Hr, Wr, Hout, Wout, pad_top, pad_left, pad_value, ST(stream)), "resize_bilinear_pad_f32");
    return COVER_OK;
}

// ---------------------------------------------------------------------------------------------------
// composite forwards
// ---------------------------------------------------------------------------------------------------
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
struct Carver {
    char* base;
    size_t cap, off;
    void* take(size_t bytes) {
        off = align_up(off, 256);
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
};

static size_t vit_ws(const cover_vit_desc* d, int n_seq, int T, Carver* c, void** h, void** qkv, void** attn, void** mlp,
                     void** vt, void** sk, size_t* sk_bytes, int* tcap) {
    const size_t R = (size_t)n_seq * T;
    const int HD = d->heads * d->head_dim_p;
    *tcap = (T + 31) / 32 * 32;
    Carver tmp{nullptr, 0, 0};
    Carver& cc = c ? *c : tmp;
    void* p;
    p = cc.take(R * d->dim * 2); if (h) *h = p;
    p = cc.take(R * 3 * HD * 2); if (qkv) *qkv = p;
    p = cc.take(R * HD * 2); if (attn) *attn = p;
    p = cc.take(R * d->mlp_p * 2); if (mlp) *mlp = p;
    p = cc.take((size_t)n_seq * HD * (*tcap) * 2); if (vt) *vt = p;
    size_t skb = 0;
    {
        const int ns[4] = {3 * HD, d->dim, d->mlp_p, d->dim};
        const int ks[4] = {d->dim, HD, d->dim, d->mlp_p};
        for (int i = 0; i < 4; ++i) {
            size_t b = gemm_workspace_bytes((int)R, ns[i], ks[i]);
            skb = b > skb ? b : skb;
        }
    }
    p = cc.take(skb); if (sk) *sk = p;
    if (sk_bytes) *sk_bytes = skb;
    return cc.off + 256;
}
size_t cover_vit_workspace_bytes(const cover_vit_desc* d, int n_seq, int T) {
    int tcap;
    return vit_ws(d, n_seq, T, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &tcap);
}

__global__ void zero_u32_k(unsigned* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
int cover_vit_forward(const cover_vit_desc* d, void* x, int n_seq, int T, void* attn_out, cover_workspace ws, int variant,
                      void* stream) {
    if (!d || !x || !d->layers_host) return fail(COVER_EINVAL, "cover_vit_forward: null pointer");
    if (d->last_attn_only && !attn_out) return fail(COVER_EINVAL, "cover_vit_forward: last_attn_only needs attn_out");
    hipStream_t st = ST(stream);
    Carver c{(char*)ws.ptr, ws.bytes, 0};
    void *h, *qkv, *attn, *mlp, *vt, *sk;
    size_t skb;
    int tcap;
    const size_t need = vit_ws(d, n_seq, T, &c, &h, &qkv, &attn, &mlp, &vt, &sk, &skb, &tcap);
    if (!ws.ptr || ws.bytes < need) return fail(COVER_EWORKSPACE, "cover_vit_forward: workspace too small");
    const int R = n_seq * T, dim = d->dim, H = d->heads, Dp = d->head_dim_p, HD = H * Dp;
    // the transposed-V scratch must be finite where keys >= T are read under a zero probability
    // (a kernel, not hipMemsetAsync: this forward is replayed inside hipGraphs, and a memset NODE was seen to run unordered with the kernel
    //  nodes around it on ROCm 7.2 -- see the tail-reduction note in cover_decoder_forward)
    if (tcap != T) {
        const size_t words = (size_t)n_seq * HD * tcap / 2;   // bf16 elements / 2 (HD is even)
        hipLaunchKernelGGL(zero_u32_k, dim3((unsigned)((words + 1023) / 1024 < 2048 ? (words + 1023) / 1024 : 2048)), dim3(256), 0, st, (unsigned*)vt, words);
        HIPCHK(hipGetLastError(), "zero vt");
    }

    // ln1 of layer 0 is its own launch; every later LayerNorm rides on the GEMM that produces its input (folded into the
    // split-K reduction whenever that GEMM splits K, which the dim-wide outputs of ViT-sized problems do)
    HIPCHK(launch_layernorm_bf16((const bf16_t*)x, dim, d->layers_host[0].ln1_w, d->layers_host[0].ln1_b, (bf16_t*)h, dim, R, dim, d->ln_eps, st), "vit ln1");
    for (int l = 0; l < d->n_layers; ++l) {
        const cover_vit_layer& L = d->layers_host[l];
        cover_gemm_epi e;
        memset(&e, 0, sizeof e);
        e.out_scale = 1.0f;
        e.bias = L.qkv_b;
        HIPCHK(launch_gemm_bf16((const bf16_t*)h, dim, (const bf16_t*)L.qkv_w, qkv, 3 * HD, R, 3 * HD, dim, &e, (float*)sk, skb, variant, st), "vit qkv");
        cover_rope_args ra;
        memset(&ra, 0, sizeof ra);
        ra.qkv = qkv; ra.ld_qkv = 3 * HD; ra.B = n_seq; ra.T = T; ra.Hq = H; ra.Hkv = H; ra.D = Dp; ra.rope_mode = 0;
        ra.vt_cache = vt; ra.vt_slot_stride = (long long)HD * tcap; ra.vt_h_stride = (long long)Dp * tcap; ra.vt_d_stride = tcap;
        HIPCHK(launch_rope_kv_write(&ra, st), "vit v-transpose");
        cover_attn_args aa;
        memset(&aa, 0, sizeof aa);
        aa.q = qkv; aa.q_b_stride = (long long)T * 3 * HD; aa.q_t_stride = 3 * HD; aa.q_h_stride = Dp;
        aa.out = attn; aa.o_b_stride = (long long)T * HD; aa.o_t_stride = HD; aa.o_h_stride = Dp;
        aa.B = n_seq; aa.Tq = T; aa.Hq = H; aa.Hkv = H; aa.D = Dp; aa.scale = d->attn_scale; aa.n_seg = 1;
        aa.seg[0].k = (const bf16_t*)qkv + HD; aa.seg[0].k_slot_stride = (long long)T * 3 * HD; aa.seg[0].k_t_stride = 3 * HD; aa.seg[0].k_h_stride = Dp;
        aa.seg[0].vt = vt; aa.seg[0].vt_slot_stride = (long long)HD * tcap; aa.seg[0].vt_h_stride = (long long)Dp * tcap; aa.seg[0].vt_d_stride = tcap;
        aa.seg[0].len = T; aa.seg[0].mask_mode = COVER_MASK_LEN;
        HIPCHK(launch_attention_bf16(&aa, st), "vit attention");
        const bool last_attn = d->last_attn_only && l == d->n_layers - 1;
        memset(&e, 0, sizeof e);
        e.out_scale = 1.0f;
        e.bias = L.proj_b;
        if (last_attn) {
            HIPCHK(launch_gemm_bf16((const bf16_t*)attn, HD, (const bf16_t*)L.proj_w, attn_out, dim, R, dim, HD, &e, (float*)sk, skb, variant, st), "vit proj (tap)");
            break;
        }
        e.residual = x; e.ld_residual = dim; e.layer_scale = L.ls1;
        e.norm_w = L.ln2_w; e.norm_b = L.ln2_b; e.norm_out = h; e.ld_norm_out = dim; e.norm_style = 2; e.norm_eps = d->ln_eps;
        HIPCHK(launch_gemm_bf16((const bf16_t*)attn, HD, (const bf16_t*)L.proj_w, x, dim, R, dim, HD, &e, (float*)sk, skb, variant, st), "vit proj (+ln2)");
        memset(&e, 0, sizeof e);
        e.out_scale = 1.0f;
        e.bias = L.fc1_b; e.act = d->act;
        HIPCHK(launch_gemm_bf16((const bf16_t*)h, dim, (const bf16_t*)L.fc1_w, mlp, d->mlp_p, R, d->mlp_p, dim, &e, (float*)sk, skb, variant, st), "vit fc1");
        memset(&e, 0, sizeof e);
        e.out_scale = 1.0f;
        e.bias = L.fc2_b; e.residual = x; e.ld_residual = dim; e.layer_scale = L.ls2;
        if (l + 1 < d->n_layers) {
            const cover_vit_layer& Ln = d->layers_host[l + 1];
            e.norm_w = Ln.ln1_w; e.norm_b = Ln.ln1_b; e.norm_out = h; e.ld_norm_out = dim; e.norm_style = 2; e.norm_eps = d->ln_eps;
        }
        HIPCHK(launch_gemm_bf16((const bf16_t*)mlp, d->mlp_p, (const bf16_t*)L.fc2_w, x, dim, R, dim, d->mlp_p, &e, (float*)sk, skb, variant, st), "vit fc2 (+next ln1)");
    }
    return COVER_OK;
}

static size_t dec_ws(const cover_dec_desc* d, int rows, Carver* c, void** h, void** qkv, void** attn, void** mlp, void** sk,
                     size_t* sk_bytes, void** st_o = nullptr, void** st_ml = nullptr, void** q8 = nullptr, void** q8s = nullptr, void** q8mx = nullptr) {
    Carver tmp{nullptr, 0, 0};
    Carver& cc = c ? *c : tmp;
    const int nqkv = (d->Hq + 2 * d->Hkv) * d->D;
    void* p;
    p = cc.take((size_t)rows * d->dim * 2); if (h) *h = p;
    p = cc.take((size_t)rows * nqkv * 2); if (qkv) *qkv = p;
    p = cc.take((size_t)rows * d->Hq * d->D * 2); if (attn) *attn = p;
    p = cc.take((size_t)rows * d->mlp * 2); if (mlp) *mlp = p;
    p = cc.take((size_t)rows * d->Hq * d->D * 4); if (st_o) *st_o = p;   // attention state (seg0_shared decode)
    p = cc.take((size_t)rows * d->Hq * 2 * 4); if (st_ml) *st_ml = p;
    {   // e4m3 twin of the current GEMM input (rows > 64 with fp8 weights: the MX-scaled fp8 tiled GEMM) + its row scales
        int kmax = d->dim > d->mlp ? d->dim : d->mlp;
        kmax = kmax > d->Hq * d->D ? kmax : d->Hq * d->D;
        p = cc.take((size_t)rows * ((kmax + 127) / 128 * 128)); if (q8) *q8 = p;
        p = cc.take((size_t)rows * 4); if (q8s) *q8s = p;
        p = cc.take((size_t)rows * ((kmax + 127) / 128 * 4)); if (q8mx) *q8mx = p;   // MX block scales of the down_proj input ([k-tile][row][4])
    }
    size_t skb = 0;
    {
        const int ns[4] = {nqkv, d->dim, 2 * d->mlp, d->dim};
        const int ks[4] = {d->dim, d->Hq * d->D, d->dim, d->mlp};
        for (int i = 0; i < 4; ++i) {
            size_t b = gemm_workspace_bytes(rows, ns[i], ks[i]);
            skb = b > skb ? b : skb;
        }
    }
    p = cc.take(skb); if (sk) *sk = p;
    if (sk_bytes) *sk_bytes = skb;
    return cc.off + 256;
}
size_t cover_decoder_workspace_bytes(const cover_dec_desc* d, int rows) {
    return dec_ws(d, rows, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int cover_decoder_forward(const cover_dec_desc* d, const cover_dec_pass* p, void* x, cover_workspace ws, int variant,
                          void* stream) {
    if (!d || !p || !x || !d->layers_host) return fail(COVER_EINVAL, "cover_decoder_forward: null pointer");
    if (p->n_groups < 1 || p->n_groups > 2) return fail(COVER_EINVAL, "cover_decoder_forward: 1 or 2 groups");
    hipStream_t st = ST(stream);
    int rows = 0, row0[2] = {0, 0};
    for (int g = 0; g < p->n_groups; ++g) {
        const cover_dec_group& G = p->groups[g];
        if (G.n_seg < 1 || G.n_seg > 3 || G.write_seg < 0 || G.write_seg >= G.n_seg)
            return fail(COVER_EINVAL, "cover_decoder_forward: bad segment description");
        row0[g] = rows;
        rows += G.B * G.T;
    }
    Carver c{(char*)ws.ptr, ws.bytes, 0};
    void *h, *qkv, *attn, *mlp, *sk, *st_o, *st_ml, *q8, *q8s, *q8mx;
    size_t skb;
    const size_t need = dec_ws(d, rows, &c, &h, &qkv, &attn, &mlp, &sk, &skb, &st_o, &st_ml, &q8, &q8s, &q8mx);
    if (!ws.ptr || ws.bytes < need) return fail(COVER_EWORKSPACE, "cover_decoder_forward: workspace too small");
    const int dim = d->dim, Hq = d->Hq, Hkv = d->Hkv, D = d->D, nqkv = (Hq + 2 * Hkv) * D, HD = Hq * D;

    // fp8 profile with more rows than the weight-streaming kernels take (config 5: M = 512 decode rows, and its prefill): the
    // projections run on the MX-scaled fp8 matrix instruction -- the input rows of every GEMM are quantised to e4m3 (per-row
    // power-of-two scale) right before it, into one shared buffer
    const char* f8_env = getenv("COVER_FP8_MFMA");   // read per call (bench.py's agreement run toggles it inside one process)
    const bool f8 = rows > 64 && d->layers_host[0].qkv_w8 && d->layers_host[0].o_w8 && d->layers_host[0].gate_up_w8 && d->layers_host[0].down_w8 &&
                    !(f8_env && f8_env[0] == '0');
    // the norms that produce h also emit its e4m3 twin (RMSNorm widths that are multiples of 128): two quantise launches per layer less
    const bool f8q = f8 && (dim % 128) == 0;
    auto use_q8 = [&](int K, cover_gemm_epi& e) { e.a8 = q8; e.a8_scale = (const float*)q8s; e.ld_a8 = (K + 127) / 128 * 128; };
    auto norm_q8 = [&](cover_gemm_epi& e) { if (f8q) { e.norm_out8 = q8; e.norm_out8_scale = (float*)q8s; e.ld_norm_out8 = dim; } };
    auto quant = [&](const void* src, int K, cover_gemm_epi& e) -> hipError_t {
        const int kp = (K + 127) / 128 * 128;
        e.a8 = q8; e.a8_scale = (const float*)q8s; e.ld_a8 = kp;
        return launch_quantize_act_fp8((const bf16_t*)src, K, rows, K, (uint8_t*)q8, kp, (float*)q8s, st);
    };
    // MX block scales for the down_proj input (its weight twin is the k-linear image): the GLU epilogue of gate_up writes e4m3 rows + one E8M0 scale per 32
    // columns INTO THE mlp BUFFER (no bf16 GLU output, no quantiser launch); COVER_FP8_MX_FUSE=0 keeps the bf16 output and runs cover_quantize_act_fp8_mx
    // on it (same bytes, one launch more: the A/B of the fusion). Sizes below the fp8 tiles' (tests on small geometries) stay on the bf16 kernels.
    const int mlp_p = (d->mlp + 127) / 128 * 128;
    const bool down_kl = d->layers_host[0].down_klinear != 0;
    const bool mx = f8 && down_kl && rows >= 400 && dim >= 2048 && d->mlp >= 2048 && (d->mlp % 32) == 0 &&
                    (d->act == COVER_ACT_SILU || d->act == COVER_ACT_GELU_TANH);
    const char* mxf_env = getenv("COVER_FP8_MX_FUSE");
    const bool mx_fused = mx && !(mxf_env && mxf_env[0] == '0');
    // the same for the o_proj input: the attention kernel of the large-N decode pass writes the block-scaled rows INTO THE attn BUFFER (a head = one
    // 128-deep k-tile of o_proj); every other pass (prefill: two groups) keeps the bf16 rows and quantises them with one launch
    const bool o_kl = d->layers_host[0].o_klinear != 0;
    const bool mxo = f8 && o_kl && rows >= 400 && dim >= 2048 && D == 128 && Hq == Hkv;
    bool mxo_fused = false;   // set per layer by the attention launch below
    // h = in_norm_0(x); afterwards every norm is folded into the epilogue of the GEMM that produces its input
    {
        const bool f32in = p->x_f32 != nullptr;
        HIPCHK(launch_rmsnorm(f32in ? (const void*)p->x_f32 : (const void*)x, f32in ? 1 : 0, dim, d->layers_host[0].in_norm_w,
                              d->norm_w_offset, d->norm_style, (bf16_t*)h, dim, rows, dim, d->norm_eps, st,
                              f8q ? (uint8_t*)q8 : nullptr, f8q ? (dim + 127) / 128 * 128 : 0, f8q ? (float*)q8s : nullptr), "dec in_norm");
    }
    for (int l = 0; l < d->n_layers; ++l) {
        const cover_dec_layer& L = d->layers_host[l];
        mxo_fused = false;
        const bool first_f32 = (l == 0 && p->x_f32 != nullptr);
        cover_gemm_epi e;
        memset(&e, 0, sizeof e);
        e.out_scale = 1.0f;
        e.bias = L.qkv_b;
        e.w8 = L.qkv_w8; e.w8_scale = L.qkv_s;
        if (f8q) use_q8(dim, e);                               // h8 came with the norm that produced h
        else if (f8) HIPCHK(quant(h, dim, e), "dec quantise (qkv input)");
        // weight-streaming path: leave the split-K partials for rope_kv_write to fold (one launch and one pass less)
        int qkv_splits = 0;
        if (rows <= 64 && p->n_groups == 1 && (variant == 0 || variant == 3) && p->groups[0].own_kv_mode == 0)
            HIPCHK(launch_gemm_skinny_partial((const bf16_t*)h, dim, (const bf16_t*)L.qkv_w, (float*)sk, skb, rows, nqkv, dim, &qkv_splits, st,
                                              L.qkv_w8, L.qkv_s), "dec qkv (partials)");
        else {
            // few-token groups on the LDS tiles (the pi0 action expert: 200 rows, T = 5): a split-K launch leaves its slabs for
            // rope_kv_write to fold as well (same sums, same order, same rounding as the reduction launch it replaces)
            const char* fold_env = getenv("COVER_QKV_FOLD");   // A/B knob (read per call: the tests toggle it): 0 keeps the reduction launch
            const bool fold = rows > 64 && p->n_groups == 1 && p->groups[0].own_kv_mode == 0 && p->groups[0].T < 16 && !f8 &&
                              !(p->groups[0].seg0_shared && p->groups[0].T == 1) && !(fold_env && fold_env[0] == '0');
            HIPCHK(launch_gemm_bf16((const bf16_t*)h, dim, (const bf16_t*)L.qkv_w, qkv, nqkv, rows, nqkv, dim, &e, (float*)sk, skb, variant, st,
                                    fold ? &qkv_splits : nullptr), "dec qkv");
        }
        cover_rope_args ras[2];
        cover_attn_args aas[2];
        bool pending[2] = {false, false};
        // both groups in one RoPE and one attention launch: plain multi-token groups only (no shared-prefix phase split)
        const bool pair_ok = p->n_groups == 2 && qkv_splits == 0 && !(p->groups[0].seg0_shared && p->groups[0].T == 1) &&
                             !(p->groups[1].seg0_shared && p->groups[1].T == 1);
        for (int g = 0; g < p->n_groups; ++g) {
            const cover_dec_group& G = p->groups[g];
            if (G.B * G.T == 0) continue;
            bf16_t* gq = (bf16_t*)qkv + (size_t)row0[g] * nqkv;
            const cover_kv_segment& W = G.segs[G.write_seg];
            // The fused launch is one block per (16 candidates, head) and re-reads the shared segment per block; with many units
            // AND long own-token segments (config 5: N = 512, up to 56 own keys, several serial tiles per pool-C wave) the
            // three-launch path -- shared segment as ONE flash pass over all candidates, then the per-candidate segments seeded
            // with its state -- is faster (N = 512: fused 109 / 163 / 362 us per layer at 1 / 16 / 56 own keys, crossover ~20;
            // decision 1045 -> 950 ms). At <= 8 own keys the fused launch wins at every N measured (N = 128 / 256 / 512).
            if (G.own_kv_mode > 0) {
                // large-N candidate decode: own-token pass on the VALU (RoPE + append + attention over the candidate's own keys, state
                // out), then ONE MFMA pass over [shared prefix | the prompt's text] with the samples of a prompt as the query rows of a
                // batch entry, resumed from that state
                if (!(G.T == 1 && G.n_seg == 3 && G.write_seg == 2 && G.seg0_shared && Hq == Hkv && G.seg1_group > 0 && G.B % G.seg1_group == 0 &&
                      qkv_splits == 0 && G.write_t_offset_of_batch == nullptr && G.segs[0].mask_mode == COVER_MASK_LEN && G.segs[1].mask_mode == COVER_MASK_LEN))
                    return fail(COVER_EINVAL, "cover_decoder_forward: own_kv_mode needs T == 1, three segments (shared | per prompt | own), MHA, a regular seg1_group");
                const long long cap = W.k_slot_stride / ((long long)Hkv * D);
                cover_own_attn_args oa;
                memset(&oa, 0, sizeof oa);
                oa.qkv = gq; oa.ld_qkv = nqkv; oa.N = G.B; oa.H = Hq; oa.D = D; oa.scale = d->attn_scale;
                oa.positions = G.positions; oa.cos_table = d->cos_table; oa.sin_table = d->sin_table; oa.n_pos = d->n_pos; oa.rope_mode = d->rope_mode;
                oa.fp8 = G.own_kv_mode == 2; oa.t_cap = (int)cap; oa.slot_stride = W.k_slot_stride;
                char* kb = (char*)L.k_cache + 2 * G.seg_k_offset[2];
                char* vb = (char*)L.vt_cache + 2 * G.seg_vt_offset[2];
                oa.k = kb; oa.v = vb;
                if (oa.fp8) { oa.k_scale = (float*)(kb + G.own_region_elems); oa.v_scale = (float*)(vb + G.own_region_elems); }
                oa.slot_of_batch = G.write_slot_of_batch ? G.write_slot_of_batch : G.segs[2].slot_of_batch;
                oa.write_t = G.write_t_offset;
                oa.state_o = (float*)st_o + (size_t)row0[g] * Hq * D;
                oa.state_ml = (float*)st_ml + (size_t)row0[g] * Hq * 2;
                HIPCHK(launch_decode_own_attention(&oa, st), "dec own-token attention");
                cover_attn_args sa;
                memset(&sa, 0, sizeof sa);
                const int S = G.seg1_group;
                sa.q = gq; sa.q_b_stride = (long long)S * nqkv; sa.q_t_stride = nqkv; sa.q_h_stride = D;
                sa.out = (bf16_t*)attn + (size_t)row0[g] * HD; sa.o_b_stride = (long long)S * HD; sa.o_t_stride = HD; sa.o_h_stride = D;
                sa.B = G.B / S; sa.Tq = S; sa.Hq = Hq; sa.Hkv = Hkv; sa.D = D; sa.scale = d->attn_scale; sa.n_seg = 2;
                for (int s = 0; s < 2; ++s) {
                    sa.seg[s] = G.segs[s];
                    sa.seg[s].k = (const bf16_t*)L.k_cache + G.seg_k_offset[s];
                    sa.seg[s].vt = (const bf16_t*)L.vt_cache + G.seg_vt_offset[s];
                }
                sa.seg[1].slot_of_batch = G.seg1_slot_of_group; sa.seg[1].len_of_batch = G.seg1_len_of_group;
                sa.state_in_o = oa.state_o; sa.state_in_ml = oa.state_ml;
                if (mxo && p->n_groups == 1 && !(mxf_env && mxf_env[0] == '0')) {
                    sa.out8 = attn; sa.out8_mx = q8mx; sa.out8_rows = rows;
                    if (attention_mx_ok(&sa)) mxo_fused = true;
                    else { sa.out8 = nullptr; sa.out8_mx = nullptr; }
                }
                HIPCHK(launch_attention_bf16(&sa, st), "dec attention (shared prefix + prompt text, resumed from the own-token state)");
                continue;
            }
            static const char* da_max_env = getenv("COVER_DA_FUSED_MAX_N");   // experiment knob: force the three-launch path above N
            const int da_max = da_max_env ? atoi(da_max_env) : (1 << 30);
            const bool da_long = ((G.B + 15) / 16) * Hq >= 768 && G.segs[2].len > 16;
            if (G.T == 1 && G.seg0_shared && G.n_seg == 3 && G.write_seg == 2 && Hq == Hkv && (D == 64 || D == 128) && G.B <= da_max && !da_long &&
                G.segs[0].mask_mode == COVER_MASK_LEN && G.segs[1].mask_mode == COVER_MASK_LEN && G.segs[2].mask_mode == COVER_MASK_LEN &&
                G.write_t_offset_of_batch == nullptr && G.segs[2].len_of_batch == nullptr && G.segs[0].len_of_batch == nullptr) {
                // single-token candidate decode: RoPE + KV append + 3-segment attention in ONE launch
                cover_decode_attn_args da;
                memset(&da, 0, sizeof da);
                da.qkv = gq; da.ld_qkv = nqkv;
                if (qkv_splits > 0) { da.n_splits = qkv_splits; da.partial = (const float*)sk; da.bias = L.qkv_b; }
                da.N = G.B; da.H = Hq; da.D = D; da.scale = d->attn_scale;
                da.positions = G.positions; da.cos_table = d->cos_table; da.sin_table = d->sin_table; da.n_pos = d->n_pos;
                da.rope_mode = d->rope_mode;
                for (int s = 0; s < 3; ++s) {
                    da.seg[s] = G.segs[s];
                    da.seg[s].k = (const bf16_t*)L.k_cache + G.seg_k_offset[s];
                    da.seg[s].vt = (const bf16_t*)L.vt_cache + G.seg_vt_offset[s];
                }
                da.seg[2].slot_of_batch = G.write_slot_of_batch ? G.write_slot_of_batch : G.segs[2].slot_of_batch;
                da.write_t = G.write_t_offset;
                da.out = (bf16_t*)attn + (size_t)row0[g] * HD; da.out_row_stride = HD;
                HIPCHK(launch_decode_attention_fused(&da, st), "dec fused decode attention");
                continue;
            }
            cover_rope_args& ra = ras[g];
            memset(&ra, 0, sizeof ra);
            ra.qkv = gq; ra.ld_qkv = nqkv; ra.B = G.B; ra.T = G.T; ra.Hq = Hq; ra.Hkv = Hkv; ra.D = D;
            ra.positions = G.positions; ra.cos_table = d->cos_table; ra.sin_table = d->sin_table; ra.n_pos = d->n_pos;
            ra.rope_mode = d->rope_mode;
            ra.k_cache = (bf16_t*)L.k_cache + G.seg_k_offset[G.write_seg];
            ra.k_slot_stride = W.k_slot_stride; ra.k_t_stride = W.k_t_stride; ra.k_h_stride = W.k_h_stride;
            ra.vt_cache = (bf16_t*)L.vt_cache + G.seg_vt_offset[G.write_seg];
            ra.vt_slot_stride = W.vt_slot_stride; ra.vt_h_stride = W.vt_h_stride; ra.vt_d_stride = W.vt_d_stride;
            ra.slot_of_batch = G.write_slot_of_batch; ra.t_offset_of_batch = G.write_t_offset_of_batch; ra.t_offset = G.write_t_offset;
            if (qkv_splits > 0) {
                ra.n_splits = qkv_splits; ra.partial = (const float*)sk; ra.bias = L.qkv_b;
            }
            cover_attn_args& aa = aas[g];
            memset(&aa, 0, sizeof aa);
            aa.q = gq; aa.q_b_stride = (long long)G.T * nqkv; aa.q_t_stride = nqkv; aa.q_h_stride = D;
            aa.out = (bf16_t*)attn + (size_t)row0[g] * HD; aa.o_b_stride = (long long)G.T * HD; aa.o_t_stride = HD; aa.o_h_stride = D;
            aa.B = G.B; aa.Tq = G.T; aa.Hq = Hq; aa.Hkv = Hkv; aa.D = D; aa.scale = d->attn_scale; aa.n_seg = G.n_seg;
            for (int s = 0; s < G.n_seg; ++s) {
                aa.seg[s] = G.segs[s];
                aa.seg[s].k = (const bf16_t*)L.k_cache + G.seg_k_offset[s];
                aa.seg[s].vt = (const bf16_t*)L.vt_cache + G.seg_vt_offset[s];
            }
            pending[g] = true;
            if (pair_ok) continue;   // both groups: launched together below
            HIPCHK(launch_rope_kv_write(&ra, st), "dec rope/kv");
            if (G.seg0_shared && G.T == 1 && G.n_seg >= 2 && G.segs[0].mask_mode == COVER_MASK_LEN) {
                // phase A: the shared segment, candidates as the query rows of one "sequence"
                cover_attn_args pa = aa;
                pa.B = 1; pa.Tq = G.B; pa.q_b_stride = 0; pa.q_t_stride = nqkv;
                pa.n_seg = 1;
                pa.out = nullptr;
                pa.state_out_o = (float*)st_o + (size_t)row0[g] * Hq * D;
                pa.state_out_ml = (float*)st_ml + (size_t)row0[g] * Hq * 2;
                HIPCHK(launch_attention_bf16(&pa, st), "dec attention (shared segment)");
                // phase B: each candidate's own segments, seeded with the phase-A state
                for (int s = 1; s < G.n_seg; ++s) aa.seg[s - 1] = aa.seg[s];
                aa.n_seg = G.n_seg - 1;
                aa.state_in_o = pa.state_out_o;
                aa.state_in_ml = pa.state_out_ml;
            }
            HIPCHK(launch_attention_bf16(&aa, st), "dec attention");
        }
        if (pair_ok && pending[0] && pending[1]) {
            // two row groups of one pass (prefill: shared-prefix rows + the prompts' text rows): RoPE / KV placement of both,
            // then attention of both, one launch each -- the second group attends keys the first one writes
            HIPCHK(launch_rope_kv_write_pair(&ras[0], &ras[1], st), "dec rope/kv (both groups)");
            HIPCHK(launch_attention_bf16_pair(&aas[0], &aas[1], st), "dec attention (both groups)");
        } else if (pair_ok) {
            for (int g = 0; g < 2; ++g)
                if (pending[g]) {
                    HIPCHK(launch_rope_kv_write(&ras[g], st), "dec rope/kv");
                    HIPCHK(launch_attention_bf16(&aas[g], st), "dec attention");
                }
        }
        memset(&e, 0, sizeof e);
        e.out_scale = 1.0f;
        e.residual = first_f32 ? (const void*)p->x_f32 : (const void*)x; e.residual_f32 = first_f32 ? 1 : 0; e.ld_residual = dim;
        e.norm_w = L.post_norm_w; e.norm_out = h; e.ld_norm_out = dim; e.norm_style = d->norm_style;
        e.norm_w_offset = d->norm_w_offset; e.norm_eps = d->norm_eps;
        e.w8 = L.o_w8; e.w8_scale = L.o_s; e.w8_klinear = L.o_klinear;
        if (mxo) {
            if (!mxo_fused) HIPCHK(launch_quantize_act_fp8_mx((const bf16_t*)attn, HD, rows, HD, (uint8_t*)q8, HD, (uint8_t*)q8mx, st), "dec quantise (o_proj input, MX)");
            e.a8 = mxo_fused ? attn : q8; e.ld_a8 = HD; e.a8_mx = q8mx;
        } else if (f8 && !L.o_klinear) {
            HIPCHK(quant(attn, HD, e), "dec quantise (o_proj input)");
        }   // (a k-linear twin below the MX sizes: bf16 operands on the bf16 image)
        norm_q8(e);
        HIPCHK(launch_gemm_bf16((const bf16_t*)attn, HD, (const bf16_t*)L.o_w, x, dim, rows, dim, HD, &e, (float*)sk, skb, variant, st), "dec o_proj (+post_norm)");
        memset(&e, 0, sizeof e);
        e.out_scale = 1.0f;
        e.act = d->act; e.glu = 1;
        e.w8 = L.gate_up_w8; e.w8_scale = L.gate_up_s;
        if (f8q) use_q8(dim, e);
        else if (f8) HIPCHK(quant(h, dim, e), "dec quantise (gate_up input)");
        if (mx_fused) { e.out8 = mlp; e.out8_mx = q8mx; e.ld_out8 = mlp_p; }
        HIPCHK(launch_gemm_bf16((const bf16_t*)h, dim, (const bf16_t*)L.gate_up_w, mlp, d->mlp, rows, 2 * d->mlp, dim, &e, (float*)sk, skb, variant, st), "dec gate_up");
        memset(&e, 0, sizeof e);
        e.out_scale = 1.0f;
        e.residual = x; e.ld_residual = dim;
        if (l + 1 < d->n_layers) {  // next layer's input norm rides on this GEMM
            e.norm_w = d->layers_host[l + 1].in_norm_w; e.norm_out = h; e.ld_norm_out = dim; e.norm_style = d->norm_style;
            e.norm_w_offset = d->norm_w_offset; e.norm_eps = d->norm_eps;
        }
        e.w8 = L.down_w8; e.w8_scale = L.down_s; e.w8_klinear = L.down_klinear;
        if (mx) {
            if (!mx_fused) HIPCHK(launch_quantize_act_fp8_mx((const bf16_t*)mlp, d->mlp, rows, d->mlp, (uint8_t*)q8, mlp_p, (uint8_t*)q8mx, st), "dec quantise (down input, MX)");
            e.a8 = mx_fused ? mlp : q8; e.ld_a8 = mlp_p; e.a8_mx = q8mx;
        } else if (f8 && !L.down_klinear) {
            HIPCHK(quant(mlp, d->mlp, e), "dec quantise (down input)");
        }   // (a k-linear twin below the MX sizes: bf16 operands on the bf16 image)
        if (l + 1 < d->n_layers) norm_q8(e);
        HIPCHK(launch_gemm_bf16((const bf16_t*)mlp, d->mlp, (const bf16_t*)L.down_w, x, dim, rows, dim, d->mlp, &e, (float*)sk, skb, variant, st), "dec down (+next in_norm)");
    }
    if (p->final_norm)
        HIPCHK(launch_rmsnorm(x, 0, dim, d->final_norm_w, d->norm_w_offset, d->norm_style, (bf16_t*)x, dim, rows, dim, d->norm_eps, st), "dec final_norm");
    return COVER_OK;
}

// ---------------------------------------------------------------------------------------------------
// graphs, timers
// ---------------------------------------------------------------------------------------------------
int cover_graph_begin(void* stream) {
    HIPCHK(hipStreamBeginCapture(ST(stream), hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
    return COVER_OK;
}
int cover_graph_end(void* stream, void** out) {
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamEndCapture(ST(stream), &g), "hipStreamEndCapture");
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    HIPCHK(e, "hipGraphInstantiate");
    *out = (void*)ge;
    return COVER_OK;
}
int cover_graph_launch(void* ge, void* stream) {
    HIPCHK(hipGraphLaunch((hipGraphExec_t)ge, ST(stream)), "hipGraphLaunch");
    return COVER_OK;
}
int cover_graph_destroy(void* ge) {
    HIPCHK(hipGraphExecDestroy((hipGraphExec_t)ge), "hipGraphExecDestroy");
    return COVER_OK;
}

struct Timer { hipEvent_t a, b; };
int cover_timer_create(void** out) {
    Timer* t = new Timer;
    hipError_t e = hipEventCreate(&t->a);
    if (e == hipSuccess) e = hipEventCreate(&t->b);
    if (e != hipSuccess) { delete t; return fail(COVER_EHIP, "hipEventCreate", e); }
    *out = t;
    return COVER_OK;
}
int cover_timer_start(void* timer, void* stream) {
    HIPCHK(hipEventRecord(((Timer*)timer)->a, ST(stream)), "hipEventRecord");
    return COVER_OK;
}
int cover_timer_stop(void* timer, void* stream, float* ms) {
    Timer* t = (Timer*)timer;
    HIPCHK(hipEventRecord(t->b, ST(stream)), "hipEventRecord");
    HIPCHK(hipEventSynchronize(t->b), "hipEventSynchronize");
    HIPCHK(hipEventElapsedTime(ms, t->a, t->b), "hipEventElapsedTime");
    return COVER_OK;
}
int cover_timer_destroy(void* timer) {
    Timer* t = (Timer*)timer;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
    return COVER_OK;
}
int cover_stream_sync(void* stream) {
    HIPCHK(hipStreamSynchronize(ST(stream)), "hipStreamSynchronize");
    return COVER_OK;
}

size_t cover_sizeof(const char* n) {
#define SZ(T) if (!strcmp(n, #T)) return sizeof(T)
    SZ(cover_gemm_epi); SZ(cover_kv_segment); SZ(cover_attn_args); SZ(cover_rope_args); SZ(cover_patchify_args);
    SZ(cover_gemm_f32_args); SZ(cover_mha_f32_args); SZ(cover_token_select_args); SZ(cover_score_select_args);
    SZ(cover_workspace); SZ(cover_vit_layer); SZ(cover_vit_desc); SZ(cover_dec_layer); SZ(cover_dec_desc);
    SZ(cover_dec_group); SZ(cover_dec_pass); SZ(cover_decode_attn_args); SZ(cover_own_attn_args);
#undef SZ
    return 0;
}

}  // extern "C"
