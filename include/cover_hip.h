/*
 * libcover_hip — C ABI of the MI355X (gfx950) kernels behind CoVer's candidate-sampling-and-verification
 * hot path.
 *
 * The reference (cover-vla/cover-vla) has NO FFI / plugin interface: its boundary is two Python classes
 * (PI0Policy.select_action, lerobot_custom/lerobot/common/policies/pi0/modeling_pi0.py:263-307, and
 * EfficientEnsembleMerged.compute_max_similarity_scores_batch,
 * bridge_verifier/ensemble_eval/efficient_ensemble_merged.py:309-454) whose arithmetic is eager PyTorch.
 * This header is therefore the set of entry points a ctypes binding UNDER those two classes needs
 * (INTEGRATION.md shows the binding); every entry cites the reference arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a raw DEVICE pointer unless the name ends in _host; the caller owns all buffers
 *   - bf16 tensors are uint16 storage ("bf16" in comments); fp32 tensors are float
 *   - leading dimensions / strides are in ELEMENTS
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); none synchronises or allocates
 *   - return value: 0 = ok, negative = cover_status; cover_last_error() gives the message (thread-local)
 *   - not thread-safe per stream; one process per GPU
 */
#ifndef COVER_HIP_H
#define COVER_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COVER_ABI_VERSION 1

enum cover_status { COVER_OK = 0, COVER_EINVAL = -1, COVER_EHIP = -2, COVER_EWORKSPACE = -3, COVER_EUNSUPPORTED = -4 };
enum cover_act { COVER_ACT_NONE = 0, COVER_ACT_GELU_TANH = 1, COVER_ACT_GELU_ERF = 2, COVER_ACT_SILU = 3, COVER_ACT_RELU = 4 };

int cover_abi_version(void);
const char* cover_last_error(void);
/* number of devices visible / properties of device `dev` (CU count, total memory): plumbing for bench + tests */
int cover_device_info(int dev, int* n_cu, size_t* total_mem, char* name, int name_len);

/* ------------------------------------------------------------------------------------------------
 * bf16 GEMM on MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate.
 * Replaces every nn.Linear on the path: q/k/v/o projections and gated MLPs
 * (paligemma_with_expert.py:273-276,327-341), SigLIP/DINOv2/SigLIP2 ViT linears (un-vendored HF/timm
 * modules called at paligemma_with_expert.py:229-230 and finetune_trajectory_bridge_ddp.py:314-316),
 * multimodal projector, lm_head.
 *
 * Weights are packed once (cover_pack_weight_bf16) into the MFMA fragment-major layout
 *   Wp[n/16][k/32][lane = (n%16) + 16*((k%32)/8)][8]        (1 KiB per 16x32 block, K padded to Kpad)
 * so that both the LDS-tiled kernel and the weight-streaming (M <= 64) kernel read 1 KiB fully
 * coalesced per wave instruction.
 * ------------------------------------------------------------------------------------------------ */
typedef struct cover_gemm_epi {
    const float* bias;        /* [N] fp32 or NULL */
    const void* residual;     /* bf16 [M, ld_residual] or NULL: out = residual + (layer_scale *) val */
    const float* layer_scale; /* [N] fp32 or NULL (DINOv2 LayerScale) */
    int ld_residual;
    int residual_f32;         /* 1: residual is fp32 [M, ld_residual] (pi0 expert layer 0: the un-rounded suffix
                                 embedding is the residual, paligemma_with_expert.py:319-332) */
    int act;                  /* cover_act, applied after bias */
    int glu;                  /* 1: weight rows interleaved gate/up in 16-row blocks (pack with glu_interleave=1):
                                 out[m, j] = act(gate[m, j]) * up[m, j], output has N/2 columns */
    int out_f32;              /* 0: bf16 output, 1: fp32 output */
    float out_scale;          /* final multiply (1.0f = none) */
    /* Optional fused RMSNorm / LayerNorm of the (bf16) output rows into a second tensor: norm_out = norm(C) exactly as
     * cover_rmsnorm_bf16 / cover_layernorm_bf16 compute it from the stored bf16 C. Whenever the GEMM runs split-K it is
     * folded into the reduction (one launch less per GEMM); otherwise the library runs the norm kernel after the GEMM. */
    const float* norm_w;      /* [N] or NULL = no fused norm */
    void* norm_out;           /* bf16 [M, ld_norm_out] */
    int ld_norm_out;
    int norm_style;           /* cover_rmsnorm_bf16 style (0 Gemma, 1 Llama), or 2 = LayerNorm as cover_layernorm_bf16 computes it */
    float norm_w_offset;
    float norm_eps;
    const float* norm_b;      /* [N] LayerNorm bias (norm_style 2) or NULL */
    /* Optional e4m3 twin of the SAME (quantised) weight, cover_pack_weight_fp8, with its packed-order per-channel scales: the
     * weight-streaming kernels (M <= 32, HBM-bound) read it instead of the bf16 image -- half the bytes, bit-identical results
     * (power-of-two scales). NULL = bf16 only. */
    const void* w8;
    const float* w8_scale;
    /* Optional e4m3 twin of the ACTIVATION rows (cover_quantize_act_fp8: per-row power-of-two scales, k order of the MFMA operand)
     * with its row scales. When BOTH twins are given and the problem runs on the LDS-tiled kernels (M > 64), the GEMM runs on the
     * MX-scaled fp8 matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4 (2x the bf16 rate): C = epi(a8_scale[m] * w8_scale[n] *
     * sum_k a8[m,k] * w8[n,k]); A / Wp are then not read. NULL = bf16 activations. */
    const void* a8;
    const float* a8_scale;
    int ld_a8;               /* row pitch of a8 in BYTES (>= cover_packed_k(K), multiple of 16) */
    int ld_norm_out8;        /* row pitch of norm_out8 in BYTES */
    /* Optional e4m3 twin of norm_out (RMSNorm styles only): norm_out8 / norm_out8_scale receive exactly what cover_quantize_act_fp8
     * would make of the stored bf16 norm_out rows -- the next GEMM's fp8 operand without another launch. */
    void* norm_out8;
    float* norm_out8_scale;
    /* MX block-scaled activations (config 5, the GLU output that feeds down_proj; no reference arithmetic -- SURVEY.md 7 step 9).
     * w8_klinear = 1: w8 is the k-linear image of cover_pack_weight_fp8_klinear, read by every kernel that takes w8.
     * a8_mx: e8m0 block scales of a8 as cover_quantize_act_fp8_mx lays them out ([cover_packed_k(K) / 128][M][4] bytes, byte = 127 + log2 of the
     *   power-of-two scale of 32 consecutive k of a row); a8 is then plain row-major e4m3 (pitch ld_a8), a8_scale is not read, and w8 must be the
     *   k-linear image. C = epi(w8_scale[n] * sum_blocks 2^(mx - 127) * sum_k a8 * w8) on v_mfma_scale_f32_16x16x128_f8f6f4's own block scales.
     * out8 / out8_mx / ld_out8 (glu = 1, act SiLU / tanh-GELU, N / 2 a multiple of 32, M > 64 on the fp8 tiles): the GEMM writes its output rows
     *   in exactly that form -- what cover_quantize_act_fp8_mx makes of the bf16 rows it would have stored -- INSTEAD of C. Only the N / 2 real
     *   columns and their scale bytes are written: when N / 2 is not a multiple of 128, zero the pad columns of out8 and set the pad scale bytes to 127
     *   once (the consuming GEMM reads whole 128-deep k-tiles; an E8M0 byte 255 is a NaN). */
    const void* a8_mx;
    void* out8;
    void* out8_mx;
    int ld_out8;
    int w8_klinear;
} cover_gemm_epi;

/* bytes needed for the packed form of an [N, K] weight (K padded to a multiple of 128, N to 16) */
size_t cover_packed_weight_bytes(int N, int K);
int cover_packed_k(int K);
/* W: bf16 [N, ldw] row-major (PyTorch nn.Linear layout) -> Wp. glu_interleave=1 expects W = [gate(N/2 rows); up(N/2 rows)]
 * and emits 16-row blocks gate0,up0,gate1,up1,... */
int cover_pack_weight_bf16(const void* W, int ldw, int N, int K, void* Wp, int glu_interleave, void* stream);

/* e4m3 weights (BASELINE.json config 5; the reference has no fp8 path -- SURVEY.md 7 step 9). Per-output-channel
 * POWER-OF-TWO scales: s_n = smallest 2^e with max_k |W[n,k]| / s_n <= 448, q = RNE_e4m3(W / s_n).
 * cover_quantize_rows_fp8: W bf16 [N, ldw] -> scales[N] fp32 and Wdq bf16 [N, ldw] = s_n * q (exactly representable: pack it with
 *   cover_pack_weight_bf16 for the MFMA-bound kernels).
 * cover_pack_weight_fp8: Wdq + scales -> the e4m3 image (cover_packed_weight_fp8_bytes) for the HBM-bound weight-streaming
 *   kernels and the scales in packed channel order [ceil(N/16)*16] (glu_interleave as cover_pack_weight_bf16). */
size_t cover_packed_weight_fp8_bytes(int N, int K);
int cover_quantize_rows_fp8(const void* W, int ldw, int N, int K, float* scales, void* Wdq, void* stream);
int cover_pack_weight_fp8(const void* Wdq, int ldw, const float* scales, int N, int K, void* Wq, float* scales_packed,
                          int glu_interleave, void* stream);

/* Dynamic per-row e4m3 quantisation of bf16 activation rows for the fp8 tiled GEMM (config 5; no reference arithmetic -- SURVEY.md 7
 * step 9): scales[m] = smallest 2^e with max_k |X[m,k]| / 2^e <= 448, out8[m] = RNE_e4m3(X[m] / scales[m]) in the k order of the
 * MX MFMA operand (inside every 64-wide block, byte 16 g + 8 h + e holds k = 32 h + 8 g + e), zero padded to cover_packed_k(K).
 * ld8 in BYTES. */
int cover_quantize_act_fp8(const void* X, int ldx, int M, int K, void* out8, int ld8, float* scales, void* stream);
/* The MX form of the same (config 5's down_proj input): out8[m] = RNE_e4m3(X[m] / s) in PLAIN row-major order (zero padded to cover_packed_k(K),
 * ld8 in BYTES), one power-of-two scale per 32 consecutive k of a row -- s = smallest 2^e (e >= -126) with amax_block / 2^e <= 448, 2^0 for an
 * all-zero block -- stored as e8m0 bytes (127 + e) in mx[cover_packed_k(K) / 128][M][4] (k-tile major: what one workgroup of the consuming GEMM
 * reads per 128-deep k-tile is one contiguous run of dwords). The weight of the consuming GEMM must be packed with cover_pack_weight_fp8_klinear. */
int cover_quantize_act_fp8_mx(const void* X, int ldx, int M, int K, void* out8, int ld8, void* mx, void* stream);
/* cover_pack_weight_fp8 with the k-linear operand order: block [n/16][k/64] of the image holds, for lane (n % 16, g), the 16 CONSECUTIVE bytes
 * k = 64 (k/64) + 16 g + 0..15 (the default image: two 8-wide runs 32 apart) -- the k order of v_mfma_scale_f32_16x16x128_f8f6f4 itself, in which one
 * MX block scale covers 32 neighbouring k. No glu interleave (down / o projections). */
int cover_pack_weight_fp8_klinear(const void* Wdq, int ldw, const float* scales, int N, int K, void* Wq, float* scales_packed, void* stream);

/* variant: 0 = auto, 1 = LDS-tiled with async global->LDS (global_load_lds), 2 = LDS-tiled register-staged,
 *          3 = weight-streaming (requires M <= 64; the library picks the second-generation split-K kernel or, for
 *              M <= 32 when its plan fills the chip, the third-generation full-K chunk-loop kernel),
 *          5 = first-generation weight streaming (kept for A/B measurements), 6 = third generation forced (tests).
 *          K = TRUE reduction length (A has >= cover_packed_k(K) readable, zero-padded columns when K is not a
 *          multiple of 128).
 * splitk_ws: fp32 scratch (cover_gemm_workspace_bytes) used by the weight-streaming variants and by split-K tiles. */
size_t cover_gemm_workspace_bytes(int M, int N, int K);
int cover_gemm_bf16(const void* A, int lda, const void* Wp, void* C, int ldc, int M, int N, int K,
                    const cover_gemm_epi* epi, void* splitk_ws, size_t splitk_ws_bytes, int variant, void* stream);
/* Which kernel plan every GEMM launch of this process took since the last reset (test / audit hook: a parity test can assert
 * that a shape really ran on the tile it means to cover). counts[i], i = index into the tile table of launch_gemm_bf16 (gemm_bf16.hip):
 *   [0..8]  gemm_tiled: 0 128x128, 1 64x128, 2 64x64 (3 stages), 3 128x128 (4 stages), 4 256x128, 5 128x256, 6 128x128 (8 waves), 7 64x128 (3 stages),
 *           8 128x128 (3 stages)        [10] gemm_tiled_pc 64x128, four loader waves, 4 stages (bf16; 9, 11..18 exist as fp8 kernels only and count in [21])
 *   [19] second-generation weight streaming (gemm_skinny2)   [20] third generation (gemm_skinny3)   [21] fp8 MFMA tiles (any)   [22] first generation
 *   self-loading tiles (gemm_v3.hip), 8 waves: [23] 224x192, [24] 224x128, [25] 256x128, [26] 128x256; 4 waves (one per SIMD): [27] 224x96,
 *   k-split wave pairs (gemm_tiled_v3k): [30] 224x96 ([28], [29], [31]: 112x128 / 224x128 on four waves and 224x128 on wave pairs -- built, measured
 *   slower, not instantiated).
 * Copies min(n, 32) counters, returns 32. */
int cover_gemm_plan_counts(long long* counts, int n, int reset);
/* In-kernel probe of the most recent launch of the self-loading tiled GEMM (gemm_v3.hip; plan counters 23..30), written by one thread of
 * its first workgroup: out[0..3] = 100 MHz wall-clock stamps at kernel start / k-loop start / k-loop end / kernel end, out[4..5] = shader
 * cycle counter at k-loop start / end, out[6] = k-tiles of the loop. (out[5] - out[4]) / (out[2] - out[1]) / 10 ns = the clock the loop ran
 * at; out[12..14] = stamps inside the LDS-staged epilogue (pipeline stages released / LDS tile filled / tile published, the store loop runs
 * from out[14] to out[3]). Synchronous read of 16 device words (synchronise the stream first). Measurement hook: nothing on the product
 * path reads it. */
int cover_gemm_probe(unsigned long long* out);

/* ------------------------------------------------------------------------------------------------
 * Flash-style attention on MFMA, fp32 softmax, over up to 3 KV segments per query row.
 * Replaces eager_attention_forward (paligemma_with_expert.py:376-434: fp32 QK^T, scale after the matmul,
 * masked positions get zero probability, probabilities rounded to bf16 before PV) and the HF/timm ViT
 * attention inside the towers. Masks are generated from lengths, never materialised
 * (make_att_2d_masks, modeling_pi0.py:98-128).
 * K is read row-major [slot][t][h][d]; V is read TRANSPOSED [slot][h][d][t] (written by cover_rope_kv_write),
 * so both MFMA operands are 16-byte contiguous per lane with no LDS transpose.
 * ------------------------------------------------------------------------------------------------ */
enum cover_mask_mode { COVER_MASK_LEN = 0, COVER_MASK_CAUSAL = 1, COVER_MASK_VISLEN = 2 };
typedef struct cover_kv_segment {
    const void* k;  /* bf16 */
    const void* vt; /* bf16, t contiguous */
    long long k_slot_stride, k_t_stride, k_h_stride;
    long long vt_slot_stride, vt_h_stride, vt_d_stride;
    const int* slot_of_batch; /* [B] or NULL (slot = b) */
    const int* len_of_batch;  /* [B] or NULL (use len) */
    const int* vis_len;       /* [Tq] for COVER_MASK_VISLEN: keys [0, vis_len[t]) visible to query token t */
    int len;
    int mask_mode;
    int causal_offset;        /* COVER_MASK_CAUSAL: key j visible iff j <= t + causal_offset */
    int _pad;
} cover_kv_segment;

typedef struct cover_attn_args {
    const void* q; /* bf16 [B][Tq][Hq][D] via strides */
    void* out;     /* bf16 [B][Tq][Hq][D] via strides */
    long long q_b_stride, q_t_stride, q_h_stride;
    long long o_b_stride, o_t_stride, o_h_stride;
    int B, Tq, Hq, Hkv, D;
    float scale;
    int n_seg;
    int _pad;
    cover_kv_segment seg[3];
    /* Optional chaining of calls over different KV segments (flash-decoding style): state = per (b, t, h) the
     * normalised output so far (fp32 [B][Tq][Hq][D]) and its softmax statistics (fp32 [B][Tq][Hq][2] = running max in
     * scaled-log2 units, running sum). state_in seeds the online softmax; state_out receives the result INSTEAD of `out`
     * (which may then be NULL). Used by the decoder to attend a KV segment shared by every candidate once per 16
     * candidates instead of once per candidate. */
    const float* state_in_o; const float* state_in_ml;
    float* state_out_o; float* state_out_ml;
    /* Optional MX block-scaled output INSTEAD of `out` (config 5: the o_proj operand without a quantiser launch; no reference arithmetic):
     * out8 = e4m3 rows addressed with the o_*_stride values in BYTES, out8_mx = E8M0 scales [Hq * D / 128][out8_rows][4] as cover_quantize_act_fp8_mx
     * lays them out (row = byte offset of (b, t) / o_t_stride), exactly what that function makes of the bf16 rows `out` would have received.
     * MHA, D = 128, o_h_stride = 128, o_t_stride = Hq * 128, no state_out, at most 1023 query tiles (4095 with state_in): otherwise COVER_EINVAL. */
    void* out8; void* out8_mx;
    int out8_rows; int _pad2;
} cover_attn_args;
int cover_attention_bf16(const cover_attn_args* args, void* stream);

/* Fused single-token decode attention (OpenVLA-style candidate decode): RoPE + KV append + attention over
 * [seg[0]: ONE slot (slot 0) shared by all N candidates | seg[1]: per-prompt slot | seg[2]: the candidate's own tokens,
 * including the one written here at position write_t] in ONE launch. q/k/v come from the bf16 qkv buffer, or (n_splits > 0)
 * straight from the split-K partial sums of the weight-streaming QKV GEMM. Requirements: one new token per candidate,
 * Hq == Hkv == H, D in {64,128}, COVER_MASK_LEN segments. Same results as cover_rope_kv_write + cover_attention_bf16. */
typedef struct cover_decode_attn_args {
    const void* qkv; int ld_qkv;           /* bf16 [N][3*H*D] (used when n_splits == 0) */
    int n_splits;
    const float* partial; const float* bias; /* fp32 [n_splits][N][3*H*D], bias [3*H*D] or NULL */
    int N, H, D;
    float scale;
    const int* positions; const float* cos_table; const float* sin_table; int n_pos; int rope_mode;
    cover_kv_segment seg[3];               /* k/vt pointers are the segment bases */
    int write_t; int _pad;                 /* position of the new token inside seg[2] (slot = seg[2].slot_of_batch[n] or n) */
    void* out; long long out_row_stride;   /* bf16 [N][H*D] */
} cover_decode_attn_args;
int cover_decode_attention_fused(const cover_decode_attn_args* args, void* stream);

/* Candidate decode at large N (BASELINE config 5): RoPE + KV append + attention over every candidate's OWN generated tokens, one wave
 * per (candidate, head) on the VALU (no reuse there: pure HBM streaming), leaving the (o, m, l) softmax state that
 * cover_attention_bf16 resumes (state_in_*) over the shared image prefix and the prompt's text keys. Own-token cache layout:
 * K and V both [slot][h][t][d] (head-major, NOT transposed), bf16 or -- fp8 = 1, the "fp8 KV" of config 5 -- e4m3 with one
 * power-of-two fp32 scale per (slot, h, t) row at k_scale / v_scale[(slot * H + h) * t_cap + t]. q is rotated in place in qkv.
 * Same arithmetic as cover_rope_kv_write + cover_attention_bf16 on the de-quantised values (scores scaled after the product, fp32
 * softmax, probabilities rounded to bf16 before PV). No reference arithmetic for the fp8 cache.
 * Requirements: one new token per candidate, Hq == Hkv == H, D in {64, 128}, write_t < t_cap <= 64 (bf16, D = 128). */
typedef struct cover_own_attn_args {
    void* qkv; int ld_qkv;                  /* bf16 [N][3*H*D] */
    int N, H, D;
    float scale;
    const int* positions; const float* cos_table; const float* sin_table; int n_pos; int rope_mode;
    void* k; void* v;                       /* own-token regions */
    float* k_scale; float* v_scale;         /* fp8 only */
    int fp8; int t_cap;
    long long slot_stride;                  /* ELEMENTS between two slots = H * t_cap * D */
    const int* slot_of_batch;               /* [N] or NULL (slot = n) */
    int write_t; int _pad;                  /* position of the appended token; keys 0..write_t are attended */
    float* state_o; float* state_ml;        /* fp32 [N][H][D] (normalised), [N][H][2] (max in scaled-log2 units, sum) */
} cover_own_attn_args;
int cover_decode_own_attention(const cover_own_attn_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row kernels (HBM-bound, wave-shuffle reductions, 16-byte vector access)
 * ------------------------------------------------------------------------------------------------ */
/* nn.LayerNorm in fp32 math, bf16 in/out (ViT blocks) */
int cover_layernorm_bf16(const void* x, int ldx, const float* w, const float* b, void* y, int ldy, int rows, int dim,
                         float eps, void* stream);
/* RMSNorm: y = x * rsqrt(mean(x^2)+eps) * (w_offset + w); Gemma uses w_offset=1 (HF GemmaRMSNorm, called at
 * paligemma_with_expert.py:268,335,355), Llama w_offset=0. */
/* style 0: y = bf16(x*rstd*(w_offset+w)) (Gemma); style 1: y = bf16(w * bf16(x*rstd)) (HF LlamaRMSNorm).
 * x_f32 = 1: x is fp32 (pi0 suffix embeddings enter the expert's first norm un-rounded). */
int cover_rmsnorm_bf16(const void* x, int x_f32, int ldx, const float* w, float w_offset, int style, void* y, int ldy,
                       int rows, int dim, float eps, void* stream);

/* RoPE + KV placement. Replaces apply_rope (paligemma_with_expert.py:34-57) and the dict/concat KV cache
 * (:288-308) with in-place appends into a static cache. qkv rows = [B*T][ (Hq+2*Hkv)*D ].
 * cos/sin: fp32 [n_pos][D/2] tables built on the host with the reference's own expressions.
 * rope_mode 0: no rotation (ViT: only V^T is produced); 1: fp32 rotate, one rounding (pi0);
 *           2: bf16 cos/sin and bf16 intermediate roundings (HF Llama rotate_half arithmetic). */
typedef struct cover_rope_args {
    void* qkv; int ld_qkv;       /* bf16; q is rotated IN PLACE */
    int B, T, Hq, Hkv, D;
    const int* positions;        /* [B*T] position ids or NULL */
    const float* cos_table; const float* sin_table; int n_pos;
    int rope_mode;
    void* k_cache;               /* bf16 [slot][t][Hkv][D] or NULL (leave K in the qkv buffer, rotated in place) */
    long long k_slot_stride, k_t_stride, k_h_stride;
    void* vt_cache;              /* bf16 [slot][Hkv][D][t_cap] (required) */
    long long vt_slot_stride, vt_h_stride, vt_d_stride;
    const int* slot_of_batch;    /* [B] or NULL */
    const int* t_offset_of_batch;/* [B] or NULL */
    int t_offset;
    int n_splits;                /* > 0: the q/k/v values are NOT read from qkv but formed on the fly as
                                    bf16(sum_s partial[s][row][col] + bias[col]) from the split-K partials of the QKV
                                    GEMM (fuses the reduction into this kernel on the weight-streaming path); q is still
                                    written to qkv (rotated), k/v go to the caches */
    const float* partial;        /* fp32 [n_splits][B*T][ld_qkv-compatible N = (Hq+2Hkv)*D] */
    const float* bias;           /* [N] or NULL */
} cover_rope_args;
int cover_rope_kv_write(const cover_rope_args* args, void* stream);

/* token embedding gather (K4: modeling_pi0.py:549-553): out[i] = bf16(table[ids[i]] * scale) */
int cover_embed_gather(const void* table, int dim, const int64_t* ids, int n, float scale, void* out, int ldo,
                       void* stream);

/* image -> patch rows for the patch-embedding GEMM (im2col-free conv, K1): out[p][c*ps*ps+py*ps+px] =
 * bf16(pix*mul[c]+add[c]), zero padded to ld_out columns. in_u8_hwc=1: uint8 [H][W][3]; 0: fp32 [3][H][W] */
typedef struct cover_patchify_args {
    const void* img; int in_u8_hwc; int H, W, patch; int n_img; long long img_stride;
    float mul[3]; float add[3];
    void* out; int ld_out;
} cover_patchify_args;
int cover_patchify(const cover_patchify_args* args, void* stream);

/* row gather/scatter copy: dst[dst_row_idx[i]] = src[src_row_idx[i]] (NULL idx = identity) */
int cover_copy_rows_bf16(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, const int* src_row_idx,
                         const int* dst_row_idx, void* stream);
/* x[r] += add[r % add_rows] (position embeddings) with bf16 rounding */
int cover_add_rows_bf16(void* x, int ldx, const void* add, int ld_add, int rows, int cols, int add_rows, void* stream);
/* x = bf16(bf16(x / pre_div) * post_mul): the image-token double rounding of modeling_pi0.py:534-538 (Appendix A.3) */
int cover_scale_bf16(void* x, int ldx, int rows, int cols, float pre_div, float post_mul, void* stream);
int cover_cast_f32_to_bf16(const float* x, int ldx, void* y, int ldy, int rows, int cols, void* stream);
int cover_cast_bf16_to_f32(const void* x, int ldx, float* y, int ldy, int rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp32 kernels: verifier heads (model.py:7-112, efficient_ensemble_merged.py:194-247) and the pi0 suffix
 * projections (modeling_pi0.py:569-629,748-751), which the reference keeps in fp32.
 * ------------------------------------------------------------------------------------------------ */
typedef struct cover_gemm_f32_args {
    const float* A; long long a_row_stride, a_k_stride;   /* A[m,k] */
    const float* B; long long b_row_stride, b_k_stride;   /* B[n,k]  (nn.Linear weight: row stride K, k stride 1) */
    float* C; long long c_row_stride;                     /* C[m,n] */
    const float* bias;      /* [N] or NULL */
    const float* residual;  /* [M, c_row_stride-compatible ld_residual] or NULL: C = residual + val */
    long long ld_residual;
    int M, N, K;
    int act;
    float alpha;            /* C = residual + alpha * act(A.B^T + bias) */
    int batch; long long a_batch_stride, b_batch_stride, c_batch_stride;
    long long bias_batch_stride; /* bias of batch z = bias + z * bias_batch_stride (0: one bias for every batch) */
} cover_gemm_f32_args;
int cover_gemm_f32(const cover_gemm_f32_args* args, void* stream);
int cover_layernorm_f32(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int rows, int dim,
                        float eps, void* stream);
/* the same over stacked row groups with their own affine parameters (ensemble members batched into one launch):
 * row r uses w + (r / rows_per_group) * wb_group_stride (and b likewise) */
int cover_layernorm_f32_grouped(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int rows, int dim,
                                float eps, int rows_per_group, long long wb_group_stride, void* stream);
/* in place: x[r] = softmax(x[r] * scale) */
int cover_softmax_rows_f32(float* x, int ldx, int rows, int cols, float scale, void* stream);
/* y[r] = x[r] / ||x[r]||_2  (finetune_trajectory_bridge_ddp.py:329-330,352-354; efficient_ensemble_merged.py:223,245) */
int cover_l2norm_rows_f32(const float* x, int ldx, float* y, int ldy, int rows, int cols, void* stream);
/* y = a + b[r % b_rows] */
int cover_add_f32(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int rows, int cols, int b_rows,
                  void* stream);
/* small multi-head attention in fp32 (nn.MultiheadAttention core, after the in-projections):
 * q [B][Tq][H*Dh], k/v [B][Tk][H*Dh] via strides, optional key padding mask (1 = ignore key) [B][Tk] */
typedef struct cover_mha_f32_args {
    const float* q; long long q_b_stride, q_t_stride;
    const float* k; long long k_b_stride, k_t_stride;
    const float* v; long long v_b_stride, v_t_stride;
    float* out; long long o_b_stride, o_t_stride;
    const uint8_t* key_pad; /* [B][Tk] or NULL */
    int B, Tq, Tk, H, Dh;
    float scale;
} cover_mha_f32_args;
/* y = act(x) elementwise (COVER_ACT_*): the ReLU between LayerNorm and the second Linear of the verifier's MLP action encoder
 * (`complex_action_encoder`, bridge_verifier/ensemble_eval/efficient_ensemble_merged.py:148-184, 241-243). */
/* Contrastive evaluation statistics: loss[r] = logsumexp(logits[r, :]) - logits[r, r] (F.cross_entropy against labels arange(B),
 * reduction left to the caller) and rank[r] = how many entries of row r rank ahead of its diagonal (top-k hit <=> rank < k).
 * Replaces the per-batch F.cross_entropy + torch.topk of bridge_verifier/ensemble_eval/finetune_trajectory_bridge_ddp.py:446-469,
 * :1081-1090 (validation forward). rows <= cols. */
int cover_xent_diag_f32(const float* logits, int ld, int rows, int cols, float* loss, int* rank, void* stream);
int cover_act_f32(const float* x, int ldx, float* y, int ldy, int rows, int cols, int act, void* stream);
int cover_mha_f32(const cover_mha_f32_args* args, void* stream);
/* masked mean over T (efficient_ensemble_merged.py:236-240): y[b] = sum_t x[b,t]*(1-pad) / max(sum(1-pad), 1e-9) */
int cover_masked_mean_f32(const float* x, const uint8_t* pad, float* y, int B, int T, int D, void* stream);
/* create_sinusoidal_pos_embedding (modeling_pi0.py:71-89) evaluated in float64 on the device, cast to bf16 */
int cover_sincos_time_embed(const float* time, int B, int dim, double min_period, double max_period, void* out, int ldo,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * Selection kernels
 * ------------------------------------------------------------------------------------------------ */
/* Action-token decode head (OpenVLA profile; the 256-bin arithmetic the reference carries at
 * INT-ACT/src/experiments/policies/policy_wrapper.py:259-266). logits fp32 [rows][ld]; greedy: first arg-max over
 * [lo, hi); sampling: softmax((logits - max)/temperature) over [lo, hi), inverse CDF in index order with the
 * host-supplied uniform u[row]. */
typedef struct cover_token_select_args {
    const float* logits; long long ld; int rows; int lo, hi;
    const float* uniform;   /* [rows] in [0,1) or NULL for greedy */
    float temperature;
    int64_t* token_out;     /* [rows] */
    float* logit_out;       /* [rows] selected logit (optional, may be NULL) */
} cover_token_select_args;
int cover_token_select(const cover_token_select_args* args, void* stream);

/* K20: fuse + score + grouped arg-max (efficient_ensemble_merged.py:404-448). it: [n_members][512] image-text
 * embeddings (unit rows), act: [n_members][N][512]; scores_out [N]; result_out int32 [4] =
 * {global_idx, group_idx, idx_in_group, 0}; best_out float [2] = {max_score, best_group_mean}. First index wins ties
 * (torch.max semantics). */
typedef struct cover_score_select_args {
    const float* it; const float* act;
    int n_members, N, dim, group_size;
    float* scores_out; int* result_out; float* best_out;
    float* fused_it_out;   /* [dim] optional */
    float* fused_act_out;  /* [N][dim] optional */
} cover_score_select_args;
int cover_score_select(const cover_score_select_args* args, void* stream);
/* De-tokenise action tokens and assemble the verifier's candidate histories on the device (no host round trip between
 * sampler and verifier). For candidate n: hist[n] = [pad rows (= pad_value) | past[0..n_past) | new row], 10 rows total
 * (efficient_ensemble_merged.py:378-390 front padding), new row d = centers[clip(tok_vocab - token[n][d] - 1, 0, n_centers-1)]
 * (256-bin arithmetic of policy_wrapper.py:259-266) with the gripper (last dim) mapped to 0/1 at 0.5
 * (BridgeSimplerAdapter.postprocess_gripper_verifier, simpler.py:222-226). centers: fp32 table computed on the host in
 * float64. pad_out[n][t] = 1 for padding rows. */
int cover_tokens_to_histories(const int64_t* tokens, int ld_tokens, int N, int tok_vocab, const float* centers, int n_centers,
                              const float* past, int n_past, float pad_value, float* hist_out, uint8_t* pad_out, void* stream);

/* The same for a flow-matching policy's action chunks (pi0): candidate n contributes the first n_use actions of its chunk
 * (actions fp32 [N][chunk][>=7], strides in elements) after the verifier post-processing of the reference adapter
 * (INT-ACT .../simpler.py:96-121, base.py:20-31): dims 0-5 un-normalised (a + 1) / 2 * (hi - lo) + lo with the dataset's
 * p01 / p99 (lo_hi = fp32 [12] = lo[6] | hi[6], NULL = pass through), gripper 0 if a < 0.5 else 1.
 * hist[n] = [pad rows | past[0..n_past) | n_use chunk rows], n_past + n_use <= 10. fp32 arithmetic (the reference does this
 * in float64 on the host and converts: results agree to fp32 rounding). */
/* the same for an action-chunk horizon > 1: the first n_use 7-token actions of every candidate become its n_use newest history rows
 * (the driver scores n_action_steps future steps, run_simpler_eval_with_openpi.py:338-341) */
int cover_tokens_to_histories_steps(const int64_t* tokens, int ld_tokens, int N, int tok_vocab, const float* centers, int n_centers,
                                    const float* past, int n_past, int n_use, float pad_value, float* hist_out, uint8_t* pad_out,
                                    void* stream);
int cover_actions_to_histories(const float* actions, long long n_stride, long long t_stride, int N, int n_use, const float* lo_hi,
                               const float* past, int n_past, float pad_value, float* hist_out, uint8_t* pad_out, void* stream);

/* grouped arg-max over already-computed (e.g. all-gathered) scores: same selection rule as above */
int cover_group_argmax(const float* scores, int N, int group_size, int* result_out, float* best_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Composite forwards: the per-layer Python loops of the reference (18-layer loop in
 * paligemma_with_expert.py:258-349, HF/timm encoder loops) run here in C++ so that a whole tower / prefill /
 * decode step is one call (and one hipGraph when captured). Weight tables are plain HOST arrays of device
 * pointers filled by the Python host.
 * ------------------------------------------------------------------------------------------------ */
typedef struct cover_workspace { void* ptr; size_t bytes; } cover_workspace;

typedef struct cover_vit_layer {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    const void* qkv_w; const float* qkv_b;   /* packed [3*H*Dp, dim] */
    const void* proj_w; const float* proj_b; /* packed [dim, H*Dp] */
    const void* fc1_w; const float* fc1_b;   /* packed [mlp_p, dim] */
    const void* fc2_w; const float* fc2_b;   /* packed [dim, mlp_p] */
    const float *ls1, *ls2;                  /* LayerScale or NULL */
} cover_vit_layer;
typedef struct cover_vit_desc {
    int dim, heads, head_dim_p, mlp_p, n_layers, act; /* head_dim_p / mlp_p = zero-padded sizes (72->96, 4304->4352) */
    float ln_eps, attn_scale;                         /* attn_scale = TRUE head_dim ** -0.5 */
    const cover_vit_layer* layers_host;               /* HOST array [n_layers] */
    int last_attn_only; /* 1: the LAST listed block stops after its attention out-projection and that (pre-residual)
                           tensor is written to attn_out: the forward-hook feature of
                           finetune_trajectory_bridge_ddp.py:272-274 */
    int _pad;
} cover_vit_desc;
/* x: bf16 [n_seq*T, dim] token embeddings, updated in place through the blocks; attn_out bf16 [n_seq*T, dim]
 * (only with last_attn_only). */
size_t cover_vit_workspace_bytes(const cover_vit_desc* d, int n_seq, int T);
int cover_vit_forward(const cover_vit_desc* d, void* x, int n_seq, int T, void* attn_out, cover_workspace ws,
                      int gemm_variant, void* stream);

typedef struct cover_dec_layer {
    const float* in_norm_w; const float* post_norm_w;
    const void* qkv_w; const float* qkv_b;   /* packed [(Hq+2Hkv)*D, dim] */
    const void* o_w;                          /* packed [dim, Hq*D] */
    const void* gate_up_w;                    /* packed, glu-interleaved [2*mlp, dim] */
    const void* down_w;                       /* packed [dim, mlp] */
    void* k_cache; void* vt_cache;            /* this layer's cache bases (bf16) */
    /* optional e4m3 twins (cover_pack_weight_fp8) + packed-order scales of the four projections; NULL = bf16 only */
    const void* qkv_w8; const float* qkv_s; const void* o_w8; const float* o_s;
    const void* gate_up_w8; const float* gate_up_s; const void* down_w8; const float* down_s;
    int down_klinear;                         /* 1: down_w8 is the k-linear image (cover_pack_weight_fp8_klinear): MX block-scaled down_proj input */
    int o_klinear;                            /* the same for o_w8: MX block-scaled attention output */
} cover_dec_layer;
typedef struct cover_dec_desc {
    int dim, Hq, Hkv, D, mlp, n_layers, act;  /* act: COVER_ACT_GELU_TANH (Gemma) / COVER_ACT_SILU (Llama) */
    int norm_style;                           /* cover_rmsnorm_bf16 style */
    float norm_eps, norm_w_offset, attn_scale;
    int rope_mode; int n_pos; int _pad;
    const float* cos_table; const float* sin_table;
    const float* final_norm_w;
    const cover_dec_layer* layers_host;       /* HOST array [n_layers] */
} cover_dec_desc;
/* A pass pushes up to two GROUPS of rows through all layers with ONE read of the weights. Group g holds
 * B x T new tokens laid out [b][t]; rows of group 1 follow group 0 in x. Per layer, each group's K/V are
 * appended to segment `write_seg` of the layer cache (slot = write_slot_of_batch[b] or b,
 * t = write_t_offset (+ write_t_offset_of_batch[b]) + t) and its queries attend its `segs` in order.
 * The k/vt pointers inside `segs` are IGNORED: per layer they are layer.k_cache + seg_k_offset[i] /
 * layer.vt_cache + seg_vt_offset[i] (element offsets), so one description serves every layer.
 * pi0 denoise steps attend their own suffix K/V without keeping it (paligemma_with_expert.py:305-308): give
 * them a scratch segment that the next step overwrites. */
typedef struct cover_dec_group {
    int B, T;
    const int* positions;  /* [B*T] */
    int n_seg; int write_seg;
    cover_kv_segment segs[3];
    long long seg_k_offset[3], seg_vt_offset[3];
    const int* write_slot_of_batch; const int* write_t_offset_of_batch;
    int write_t_offset;
    int seg0_shared;   /* 1: segs[0] is the SAME slot and length for every batch row and T == 1 (decode): it is attended
                          with the batch rows as query rows of one tile (K/V read once per 16 candidates) and chained into
                          the per-row segments through the attention state */
    /* Large-N candidate decode (T == 1, n_seg == 3, write_seg == 2, seg0_shared): own_kv_mode 1 / 2 keeps the write segment in the
     * head-major layout of cover_decode_own_attention (1: bf16, 2: e4m3 + row scales stored behind the data in the region's
     * second half: own_region_elems = slots * cap * Hkv * D of that region) and runs   own-token pass (VALU)  ->  ONE MFMA pass over
     * [segs[0] | segs[1]] with the candidates of a prompt as the query rows of a batch entry. That needs the regular structure
     * the sampler has: rows [i * seg1_group, (i + 1) * seg1_group) share segs[1]'s slot seg1_slot_of_group[i] (NULL: i) and length
     * seg1_len_of_group[i] (NULL: segs[1].len). The mode is a property of the cache, not of a step: use it for every pass. */
    int own_kv_mode; int seg1_group;
    const int* seg1_slot_of_group; const int* seg1_len_of_group;
    long long own_region_elems;
    /* 1: nobody reads the write segment after this pass (pi0 denoise steps: the suffix K/V are per-step temporaries,
     * paligemma_with_expert.py:305-308) -- the pass may then leave it unwritten: with few-token groups on the split-K path the qkv slabs
     * can be folded and rotated INSIDE the attention launch (one launch per layer less; opt-in, COVER_ROPE_ATTN_FUSE=1: measured at parity) */
    int write_scratch;
} cover_dec_group;
typedef struct cover_dec_pass {
    int n_groups; int final_norm;      /* final_norm: apply final_norm_w to x at the end */
    const float* x_f32;                /* optional fp32 [rows, dim] layer-0 input (norm input AND first residual) */
    cover_dec_group groups[2];
} cover_dec_pass;
size_t cover_decoder_workspace_bytes(const cover_dec_desc* d, int rows);
/* x: bf16 [rows, dim] input embeddings, overwritten with the output hidden states */
int cover_decoder_forward(const cover_dec_desc* d, const cover_dec_pass* p, void* x, cover_workspace ws,
                          int gemm_variant, void* stream);

/* hipGraph capture helpers: everything launched on `stream` between begin/end becomes one replayable graph */
int cover_graph_begin(void* stream);
int cover_graph_end(void* stream, void** graph_exec_out);
int cover_graph_launch(void* graph_exec, void* stream);
int cover_graph_destroy(void* graph_exec);

/* Timing on the stream the kernels run on (torch.cuda.Event only sees torch's current stream): opaque hipEvent
 * pairs. cover_timer_stop synchronises the stop event and returns elapsed milliseconds. */
int cover_timer_create(void** timer_out);
int cover_timer_start(void* timer, void* stream);
int cover_timer_stop(void* timer, void* stream, float* ms_out);
int cover_timer_destroy(void* timer);
int cover_stream_sync(void* stream);

/* ---- image resampling on the device (the pre-processing either side of the hot path; cover_vla_amd/imaging.py builds the
 * span tables with the restated filter-bank code) --------------------------------------------------------------------------
 * cover_resample_axis: one separable pass along H (axis 0) or W (axis 1) of an HWC image:
 *   out[o] = sum_{j < bounds[2o+1]} coefs[o*ksize + j] * in[bounds[2o] + j].
 *   fixed_point = 1: Pillow's 8-bit path (int32 coefficients with 22 fractional bits, accumulator seeded with 1 << 21,
 *   clip8(ss >> 22); uint8 -> uint8) = torchvision Resize on a PIL image = open_clip's SigLIP2 preprocess, replaces the
 *   `self.preprocess(image)` call of bridge_verifier/ensemble_eval/efficient_ensemble_merged.py:338.
 *   fixed_point = 0: TensorFlow ScaleAndTranslate float path (fp32 coefficients, sequential accumulation; a uint8 output is
 *   the truncating cast) = tf.image.resize(bilinear, antialias=True) of process_raw_image_to_jpg,
 *   CoVer_VLA/inference/experiments/robot/simpler/eval_utils.py:273-283.
 * cover_u8_hwc_to_f32_chw_norm: ToTensor + Normalize, ((x / 255) - mean) / std, HWC uint8 -> CHW fp32.
 * cover_resize_bilinear_pad_f32: F.interpolate(bilinear, align_corners=False) of an fp32 [NC, Hin, Win] stack to Hr x Wr,
 *   placed at (pad_top, pad_left) of an Hout x Wout canvas filled with pad_value = resize_with_pad,
 *   lerobot_custom/lerobot/common/policies/pi0/modeling_pi0.py:131-150. */
/* fixed_point = 2: OpenCV's 8-bit fixed-point resize arithmetic (imgproc resize.cpp; cv2.resize(..., INTER_LANCZOS4) of the policy-side
 *   adapter, INT-ACT/src/experiments/env_adapters/simpler.py:48-52): int32 coefficients holding shorts with 11 fractional bits;
 *   first pass uint8 -> int32 exact sums (out kind 2), second pass int32 (in kind 2) -> uint8 = saturate((sum + (1 << 21)) >> 22).
 *   The in / out kinds are 0 = uint8, 1 = fp32, 2 = int32 (this mode only).
 * cover_u8_hwc_to_f32_chw_scale_norm: ((float)x * scale - mean) / std, HWC uint8 -> CHW fp32: process_images of
 *   INT-ACT/src/utils/pipeline.py:55-67 (rescale by multiplication, then normalise). */
int cover_u8_hwc_to_f32_chw_scale_norm(const uint8_t* in, float* out, int H, int W, float scale, const float* mean3, const float* std3, void* stream);
int cover_resample_axis(const void* in, int in_is_f32, void* out, int out_is_f32, int Hin, int Win, int C, int Hout, int Wout,
                        int axis, const int* bounds, const void* coefs, int ksize, int fixed_point, void* stream);
int cover_u8_hwc_to_f32_chw_norm(const uint8_t* in, float* out, int H, int W, const float* mean3, const float* std3, void* stream);
int cover_resize_bilinear_pad_f32(const float* in, float* out, int NC, int Hin, int Win, int Hr, int Wr, int Hout, int Wout,
                                  int pad_top, int pad_left, float pad_value, void* stream);

/* Per-launch kernel timing for bench.py's roofline objects: between begin/end every GEMM / attention launch issued by
 * this library carries a hipEvent pair stamped with the kernel's own start / stop on its stream. Classes:
 *   0 weight-streaming GEMM with >= 16 MB of weights (work = weight bytes)   1 LDS-tiled GEMM, ViT-sized (work = FLOPs)
 *   2 attention (work = 0)   3 small weight-streaming GEMMs (work = weight bytes)
 *   4 LDS-tiled GEMM, LLM-sized (N*K >= 16 M; work = FLOPs)   5 split-K reductions behind a weight-streaming GEMM (work = 0)
 *   7 LDS-tiled GEMM on the MX-scaled fp8 matrix instruction (work = FLOPs; priced against the 5 PFLOP/s fp8 peak)
 *   6 / 8 / 9 split-K reductions behind a class-1 / class-4 / class-7 GEMM (work = 0): every reduction is charged to the GEMM class it completes
 * Thread-safe (records are claimed atomically). Launches replayed from a hipGraph are not seen (neither time nor work).
 * end_n() synchronises the device and fills ms[n], count[n], work[n] (n <= COVER_PROF_CLASSES); it returns
 * COVER_EWORKSPACE when more launches were issued than max_events (sums incomplete). end() = end_n(.., 4). */
#define COVER_PROF_CLASSES 10
int cover_profile_begin(int max_events);
int cover_profile_end(double* ms, long long* count, double* work);
int cover_profile_end_n(double* ms, long long* count, double* work, int n_classes);

/* sizeof() of every struct above by name ("cover_attn_args", ...): lets a foreign-language binding check its
 * mirrored layouts at load time. Returns 0 for unknown names. */
size_t cover_sizeof(const char* struct_name);

#ifdef __cplusplus
}
#endif
#endif
