// Internal launch API of libcover_hip (host-side declarations). Every launcher is asynchronous on `st`
// and returns a hipError_t; nothing here allocates. The extern "C" surface is include/cover_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cover_hip.h"

typedef uint16_t bf16_t;

// ---- gemm_bf16.hip -------------------------------------------------------------------------------
// C[M,N] = epi(A[M,K] x W[N,K]^T); W pre-packed by cover_pack_weight_bf16 (fragment-major).
hipError_t launch_gemm_bf16(const bf16_t* A, int lda, const bf16_t* Wp, void* C, int ldc, int M, int N, int K,
                            const cover_gemm_epi* epi, float* splitk_ws, size_t splitk_ws_bytes, int variant,
                            hipStream_t st, int* splits_out = nullptr);
// splits_out (optional): a split-K launch of the LDS-tiled kernels with a bias-only epilogue leaves its S raw fp32 slabs [S][M][N] in ws
// and returns S here instead of folding them (the decoder folds them in rope_kv_write: one launch less per layer); 0 = C is written.
hipError_t launch_gemm_skinny_partial(const bf16_t* A, int lda, const bf16_t* Wp, float* ws, size_t ws_bytes, int M, int N,
                                      int K, int* S_out, hipStream_t st, const void* w8 = nullptr, const float* w8s = nullptr);
#define COVER_GEMM_PLANS 32
void gemm_plan_counts(long long* out, int n, int reset);   // per-plan launch counters (tests): see gemm_bf16.hip
int gemm_v3_probe(unsigned long long* out);                  // 16 words: in-kernel probe of the last gemm_v3.hip launch (g_v3_probe)
hipError_t launch_quantize_rows_fp8(const bf16_t* W, int ldw, int N, int K, float* scales, bf16_t* Wdq, hipStream_t st);
hipError_t launch_pack_weight_fp8(const bf16_t* Wdq, int ldw, const float* scales, int N, int K, uint8_t* Wq, float* scales_packed,
                                  int Kpad, int glu, hipStream_t st, int kl = 0);
hipError_t launch_quantize_act_fp8(const bf16_t* X, int ldx, int M, int K, uint8_t* out, int ld8, float* scales, hipStream_t st);
hipError_t launch_quantize_act_fp8_mx(const bf16_t* X, int ldx, int M, int K, uint8_t* out, int ld8, uint8_t* mx, hipStream_t st);
hipError_t launch_pack_weight_bf16(const bf16_t* W, int ldw, int N, int K, bf16_t* Wp, int Kpad, int glu_interleave,
                                   hipStream_t st);

// ---- attention.hip -------------------------------------------------------------------------------
hipError_t launch_attention_bf16(const cover_attn_args* a, hipStream_t st);
bool attention_mx_ok(const cover_attn_args* x);   // can this problem write cover_attn_args.out8?
hipError_t launch_attention_bf16_pair(const cover_attn_args* a0, const cover_attn_args* a1, hipStream_t st);
hipError_t launch_decode_attention_fused(const cover_decode_attn_args* a, hipStream_t st);
hipError_t launch_decode_own_attention(const cover_own_attn_args* a, hipStream_t st);   // decode_own.hip

// ---- rowops.hip ----------------------------------------------------------------------------------
hipError_t launch_layernorm_bf16(const bf16_t* x, int ldx, const float* w, const float* b, bf16_t* y, int ldy, int rows,
                                 int dim, float eps, hipStream_t st);
// q8 / q8s (optional): e4m3 twin of the stored y rows + row scales, as launch_quantize_act_fp8 would produce them (dim % 128 == 0)
hipError_t launch_rmsnorm(const void* x, int x_f32, int ldx, const float* w, float w_offset, int style, bf16_t* y, int ldy,
                          int rows, int dim, float eps, hipStream_t st, uint8_t* q8 = nullptr, int ld8 = 0, float* q8s = nullptr);
hipError_t launch_rope_kv_write(const cover_rope_args* a, hipStream_t st);
hipError_t launch_rope_kv_write_pair(const cover_rope_args* a0, const cover_rope_args* a1, hipStream_t st);
hipError_t launch_embed_gather(const bf16_t* table, int dim, const int64_t* ids, int n, float scale, bf16_t* out,
                               int ldo, hipStream_t st);
hipError_t launch_patchify(const cover_patchify_args* a, hipStream_t st);
hipError_t launch_copy_rows_bf16(const bf16_t* src, int lds_, bf16_t* dst, int ldd, int rows, int cols,
                                 const int* src_row_idx, const int* dst_row_idx, hipStream_t st);
hipError_t launch_add_bias_rows_bf16(bf16_t* x, int ldx, const bf16_t* add, int ld_add, int rows, int cols, int add_rows,
                                     hipStream_t st);
hipError_t launch_scale_bf16(bf16_t* x, int ldx, int rows, int cols, float pre_div, float post_mul, hipStream_t st);
hipError_t launch_cast_f32_to_bf16(const float* x, int ldx, bf16_t* y, int ldy, int rows, int cols, hipStream_t st);
hipError_t launch_cast_bf16_to_f32(const bf16_t* x, int ldx, float* y, int ldy, int rows, int cols, hipStream_t st);

// ---- f32ops.hip ----------------------------------------------------------------------------------
hipError_t launch_gemm_f32(const cover_gemm_f32_args* a, hipStream_t st);
hipError_t launch_layernorm_f32(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int rows,
                                int dim, float eps, hipStream_t st, int rows_per_group = 0, long long wb_group_stride = 0);
hipError_t launch_softmax_rows_f32(float* x, int ldx, int rows, int cols, float scale, hipStream_t st);
hipError_t launch_l2norm_rows_f32(const float* x, int ldx, float* y, int ldy, int rows, int cols, hipStream_t st);
hipError_t launch_add_f32(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int rows, int cols,
                          int b_rows, hipStream_t st);
hipError_t launch_xent_diag_f32(const float* x, int ldx, int rows, int cols, float* loss, int* rank, hipStream_t st);
hipError_t launch_act_f32(const float* x, int ldx, float* y, int ldy, int rows, int cols, int act, hipStream_t st);
hipError_t launch_mha_f32(const cover_mha_f32_args* a, hipStream_t st);
hipError_t launch_masked_mean_f32(const float* x, const uint8_t* pad, float* y, int B, int T, int D, hipStream_t st);
hipError_t launch_sincos_time_embed(const float* time, int B, int dim, double min_period, double max_period, bf16_t* out,
                                    int ldo, hipStream_t st);

// ---- select.hip ----------------------------------------------------------------------------------
hipError_t launch_token_select(const cover_token_select_args* a, hipStream_t st);
hipError_t launch_score_select(const cover_score_select_args* a, hipStream_t st);
hipError_t launch_tokens_to_histories(const int64_t* tokens, int ld_tokens, int N, int tok_vocab, const float* centers,
                                      int n_centers, const float* past, int n_past, float pad_value, float* hist, uint8_t* pad,
                                      hipStream_t st, int n_use = 1);
hipError_t launch_actions_to_histories(const float* actions, long long n_stride, long long t_stride, int N, int n_use,
                                       const float* lo_hi, const float* past, int n_past, float pad_value, float* hist,
                                       uint8_t* pad, hipStream_t st);
hipError_t launch_group_argmax(const float* scores, int N, int gs, int* result, float* best, hipStream_t st);
size_t gemm_workspace_bytes(int M, int N, int K);

// ---- image.hip -----------------------------------------------------------------------------------
hipError_t launch_resample_axis(const void* in, int in_kind, void* out, int out_kind, int Hin, int Win, int C, int Hout, int Wout,
                                int axis, const int* bounds, const void* coefs, int ksize, int fixed, hipStream_t st);
hipError_t launch_u8_to_chw_scale_norm(const uint8_t* in, float* out, int H, int W, int C, float scale, const float* mean, const float* stdv, hipStream_t st);
hipError_t launch_u8_to_chw_norm(const uint8_t* in, float* out, int H, int W, int C, const float* mean, const float* stdv, hipStream_t st);
hipError_t launch_bilinear_pad(const float* in, float* out, int NC, int Hin, int Win, int Hr, int Wr, int Hout, int Wout, int pad_top,
                               int pad_left, float pad_value, hipStream_t st);

// ---- prof.hip: optional per-launch hipEvent timing (classes: 0 skinny GEMM, 1 tiled GEMM, 2 attention) ----------
bool prof_enabled();
int prof_open(hipStream_t st, int cls, double work);
void prof_close(hipStream_t st, int id);
// reserve a record whose two events the caller hands to hipExtLaunchKernelGGL (kernel start / stop timestamps, no event
// packets around the launch); returns -1 when profiling is off
int prof_reserve(int cls, double work, hipEvent_t* start, hipEvent_t* stop);
