"""ctypes binding of libcover_hip.so (C ABI declared in include/cover_hip.h).

The product path fails loudly when the shared library is missing: there is no CPU fallback anywhere in this
package. Structures mirror the header one-to-one; `check_abi()` compares every mirrored sizeof with the
library's own (cover_sizeof) so a drifted layout is an import-time error, not silent corruption.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64; it MUST be the HIP runtime this process uses (device pointers and streams
# are shared with torch), so torch is imported before libcover_hip.so is dlopen'ed: the loader then resolves the
# library's libamdhip64.so.7 dependency to the copy torch already mapped instead of loading a second runtime.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COVER_LIB_PATH") or os.path.join(_HERE, "libcover_hip.so")   # override: A/B runs of two builds

c_p = C.c_void_p
c_ll = C.c_longlong
c_i = C.c_int
c_f = C.c_float


class CoverError(RuntimeError):
    pass


class GemmEpi(C.Structure):
    _fields_ = [("bias", c_p), ("residual", c_p), ("layer_scale", c_p), ("ld_residual", c_i), ("residual_f32", c_i),
                ("act", c_i), ("glu", c_i), ("out_f32", c_i), ("out_scale", c_f),
                ("norm_w", c_p), ("norm_out", c_p), ("ld_norm_out", c_i), ("norm_style", c_i), ("norm_w_offset", c_f),
                ("norm_eps", c_f), ("norm_b", c_p), ("w8", c_p), ("w8_scale", c_p), ("a8", c_p), ("a8_scale", c_p), ("ld_a8", c_i), ("ld_norm_out8", c_i),
                ("norm_out8", c_p), ("norm_out8_scale", c_p),
                ("a8_mx", c_p), ("out8", c_p), ("out8_mx", c_p), ("ld_out8", c_i), ("w8_klinear", c_i)]


class KvSegment(C.Structure):
    _fields_ = [("k", c_p), ("vt", c_p),
                ("k_slot_stride", c_ll), ("k_t_stride", c_ll), ("k_h_stride", c_ll),
                ("vt_slot_stride", c_ll), ("vt_h_stride", c_ll), ("vt_d_stride", c_ll),
                ("slot_of_batch", c_p), ("len_of_batch", c_p), ("vis_len", c_p),
                ("len", c_i), ("mask_mode", c_i), ("causal_offset", c_i), ("_pad", c_i)]


class AttnArgs(C.Structure):
    _fields_ = [("q", c_p), ("out", c_p),
                ("q_b_stride", c_ll), ("q_t_stride", c_ll), ("q_h_stride", c_ll),
                ("o_b_stride", c_ll), ("o_t_stride", c_ll), ("o_h_stride", c_ll),
                ("B", c_i), ("Tq", c_i), ("Hq", c_i), ("Hkv", c_i), ("D", c_i), ("scale", c_f),
                ("n_seg", c_i), ("_pad", c_i), ("seg", KvSegment * 3),
                ("state_in_o", c_p), ("state_in_ml", c_p), ("state_out_o", c_p), ("state_out_ml", c_p),
                ("out8", c_p), ("out8_mx", c_p), ("out8_rows", c_i), ("_pad2", c_i)]


class DecodeAttnArgs(C.Structure):
    _fields_ = [("qkv", c_p), ("ld_qkv", c_i), ("n_splits", c_i), ("partial", c_p), ("bias", c_p),
                ("N", c_i), ("H", c_i), ("D", c_i), ("scale", c_f),
                ("positions", c_p), ("cos_table", c_p), ("sin_table", c_p), ("n_pos", c_i), ("rope_mode", c_i),
                ("seg", KvSegment * 3), ("write_t", c_i), ("_pad", c_i), ("out", c_p), ("out_row_stride", c_ll)]


class OwnAttnArgs(C.Structure):
    _fields_ = [("qkv", c_p), ("ld_qkv", c_i), ("N", c_i), ("H", c_i), ("D", c_i), ("scale", c_f),
                ("positions", c_p), ("cos_table", c_p), ("sin_table", c_p), ("n_pos", c_i), ("rope_mode", c_i),
                ("k", c_p), ("v", c_p), ("k_scale", c_p), ("v_scale", c_p), ("fp8", c_i), ("t_cap", c_i),
                ("slot_stride", c_ll), ("slot_of_batch", c_p), ("write_t", c_i), ("_pad", c_i),
                ("state_o", c_p), ("state_ml", c_p)]


class RopeArgs(C.Structure):
    _fields_ = [("qkv", c_p), ("ld_qkv", c_i),
                ("B", c_i), ("T", c_i), ("Hq", c_i), ("Hkv", c_i), ("D", c_i),
                ("positions", c_p), ("cos_table", c_p), ("sin_table", c_p), ("n_pos", c_i),
                ("rope_mode", c_i),
                ("k_cache", c_p), ("k_slot_stride", c_ll), ("k_t_stride", c_ll), ("k_h_stride", c_ll),
                ("vt_cache", c_p), ("vt_slot_stride", c_ll), ("vt_h_stride", c_ll), ("vt_d_stride", c_ll),
                ("slot_of_batch", c_p), ("t_offset_of_batch", c_p), ("t_offset", c_i), ("n_splits", c_i),
                ("partial", c_p), ("bias", c_p)]


class PatchifyArgs(C.Structure):
    _fields_ = [("img", c_p), ("in_u8_hwc", c_i), ("H", c_i), ("W", c_i), ("patch", c_i), ("n_img", c_i),
                ("img_stride", c_ll), ("mul", c_f * 3), ("add", c_f * 3), ("out", c_p), ("ld_out", c_i)]


class GemmF32Args(C.Structure):
    _fields_ = [("A", c_p), ("a_row_stride", c_ll), ("a_k_stride", c_ll),
                ("B", c_p), ("b_row_stride", c_ll), ("b_k_stride", c_ll),
                ("C", c_p), ("c_row_stride", c_ll),
                ("bias", c_p), ("residual", c_p), ("ld_residual", c_ll),
                ("M", c_i), ("N", c_i), ("K", c_i), ("act", c_i), ("alpha", c_f),
                ("batch", c_i), ("a_batch_stride", c_ll), ("b_batch_stride", c_ll), ("c_batch_stride", c_ll),
                ("bias_batch_stride", c_ll)]


class MhaF32Args(C.Structure):
    _fields_ = [("q", c_p), ("q_b_stride", c_ll), ("q_t_stride", c_ll),
                ("k", c_p), ("k_b_stride", c_ll), ("k_t_stride", c_ll),
                ("v", c_p), ("v_b_stride", c_ll), ("v_t_stride", c_ll),
                ("out", c_p), ("o_b_stride", c_ll), ("o_t_stride", c_ll),
                ("key_pad", c_p), ("B", c_i), ("Tq", c_i), ("Tk", c_i), ("H", c_i), ("Dh", c_i), ("scale", c_f)]


class TokenSelectArgs(C.Structure):
    _fields_ = [("logits", c_p), ("ld", c_ll), ("rows", c_i), ("lo", c_i), ("hi", c_i),
                ("uniform", c_p), ("temperature", c_f), ("token_out", c_p), ("logit_out", c_p)]


class ScoreSelectArgs(C.Structure):
    _fields_ = [("it", c_p), ("act", c_p), ("n_members", c_i), ("N", c_i), ("dim", c_i), ("group_size", c_i),
                ("scores_out", c_p), ("result_out", c_p), ("best_out", c_p), ("fused_it_out", c_p),
                ("fused_act_out", c_p)]


class Workspace(C.Structure):
    _fields_ = [("ptr", c_p), ("bytes", C.c_size_t)]


class VitLayer(C.Structure):
    _fields_ = [("ln1_w", c_p), ("ln1_b", c_p), ("ln2_w", c_p), ("ln2_b", c_p),
                ("qkv_w", c_p), ("qkv_b", c_p), ("proj_w", c_p), ("proj_b", c_p),
                ("fc1_w", c_p), ("fc1_b", c_p), ("fc2_w", c_p), ("fc2_b", c_p), ("ls1", c_p), ("ls2", c_p)]


class VitDesc(C.Structure):
    _fields_ = [("dim", c_i), ("heads", c_i), ("head_dim_p", c_i), ("mlp_p", c_i), ("n_layers", c_i), ("act", c_i),
                ("ln_eps", c_f), ("attn_scale", c_f), ("layers_host", C.POINTER(VitLayer)),
                ("last_attn_only", c_i), ("_pad", c_i)]


class DecLayer(C.Structure):
    _fields_ = [("in_norm_w", c_p), ("post_norm_w", c_p), ("qkv_w", c_p), ("qkv_b", c_p), ("o_w", c_p),
                ("gate_up_w", c_p), ("down_w", c_p), ("k_cache", c_p), ("vt_cache", c_p),
                ("qkv_w8", c_p), ("qkv_s", c_p), ("o_w8", c_p), ("o_s", c_p), ("gate_up_w8", c_p), ("gate_up_s", c_p),
                ("down_w8", c_p), ("down_s", c_p), ("down_klinear", c_i), ("o_klinear", c_i)]


class DecDesc(C.Structure):
    _fields_ = [("dim", c_i), ("Hq", c_i), ("Hkv", c_i), ("D", c_i), ("mlp", c_i), ("n_layers", c_i), ("act", c_i),
                ("norm_style", c_i), ("norm_eps", c_f), ("norm_w_offset", c_f), ("attn_scale", c_f),
                ("rope_mode", c_i), ("n_pos", c_i), ("_pad", c_i),
                ("cos_table", c_p), ("sin_table", c_p), ("final_norm_w", c_p),
                ("layers_host", C.POINTER(DecLayer))]


class DecGroup(C.Structure):
    _fields_ = [("B", c_i), ("T", c_i), ("positions", c_p), ("n_seg", c_i), ("write_seg", c_i),
                ("segs", KvSegment * 3), ("seg_k_offset", c_ll * 3), ("seg_vt_offset", c_ll * 3),
                ("write_slot_of_batch", c_p), ("write_t_offset_of_batch", c_p),
                ("write_t_offset", c_i), ("seg0_shared", c_i),
                ("own_kv_mode", c_i), ("seg1_group", c_i), ("seg1_slot_of_group", c_p), ("seg1_len_of_group", c_p),
                ("own_region_elems", c_ll), ("write_scratch", c_i)]


class DecPass(C.Structure):
    _fields_ = [("n_groups", c_i), ("final_norm", c_i), ("x_f32", c_p), ("groups", DecGroup * 2)]


_STRUCTS = {
    "cover_gemm_epi": GemmEpi, "cover_kv_segment": KvSegment, "cover_attn_args": AttnArgs,
    "cover_rope_args": RopeArgs, "cover_patchify_args": PatchifyArgs, "cover_gemm_f32_args": GemmF32Args,
    "cover_mha_f32_args": MhaF32Args, "cover_token_select_args": TokenSelectArgs,
    "cover_score_select_args": ScoreSelectArgs, "cover_workspace": Workspace, "cover_vit_layer": VitLayer,
    "cover_vit_desc": VitDesc, "cover_dec_layer": DecLayer, "cover_dec_desc": DecDesc, "cover_dec_group": DecGroup,
    "cover_dec_pass": DecPass, "cover_decode_attn_args": DecodeAttnArgs, "cover_own_attn_args": OwnAttnArgs,
}

# every symbol include/cover_hip.h declares: (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    "cover_abi_version": (c_i, []),
    "cover_last_error": (C.c_char_p, []),
    "cover_device_info": (c_i, [c_i, _P(c_i), _P(C.c_size_t), C.c_char_p, c_i]),
    "cover_packed_weight_bytes": (C.c_size_t, [c_i, c_i]),
    "cover_packed_k": (c_i, [c_i]),
    "cover_pack_weight_bf16": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "cover_packed_weight_fp8_bytes": (C.c_size_t, [c_i, c_i]),
    "cover_quantize_rows_fp8": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "cover_pack_weight_fp8": (c_i, [c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_i, c_p]),
    "cover_quantize_act_fp8": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p]),
    "cover_quantize_act_fp8_mx": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p]),
    "cover_pack_weight_fp8_klinear": (c_i, [c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p]),
    "cover_gemm_workspace_bytes": (C.c_size_t, [c_i, c_i, c_i]),
    "cover_gemm_bf16": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, _P(GemmEpi), c_p, C.c_size_t, c_i, c_p]),
    "cover_gemm_plan_counts": (c_i, [C.POINTER(C.c_longlong), c_i, c_i]),
    "cover_gemm_probe": (c_i, [C.POINTER(C.c_ulonglong)]),
    "cover_attention_bf16": (c_i, [_P(AttnArgs), c_p]),
    "cover_decode_attention_fused": (c_i, [_P(DecodeAttnArgs), c_p]),
    "cover_decode_own_attention": (c_i, [_P(OwnAttnArgs), c_p]),
    "cover_layernorm_bf16": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p]),
    "cover_rmsnorm_bf16": (c_i, [c_p, c_i, c_i, c_p, c_f, c_i, c_p, c_i, c_i, c_i, c_f, c_p]),
    "cover_rope_kv_write": (c_i, [_P(RopeArgs), c_p]),
    "cover_embed_gather": (c_i, [c_p, c_i, c_p, c_i, c_f, c_p, c_i, c_p]),
    "cover_patchify": (c_i, [_P(PatchifyArgs), c_p]),
    "cover_copy_rows_bf16": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "cover_add_rows_bf16": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p]),
    "cover_scale_bf16": (c_i, [c_p, c_i, c_i, c_i, c_f, c_f, c_p]),
    "cover_cast_f32_to_bf16": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_p]),
    "cover_cast_bf16_to_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_p]),
    "cover_gemm_f32": (c_i, [_P(GemmF32Args), c_p]),
    "cover_layernorm_f32": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p]),
    "cover_layernorm_f32_grouped": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_ll, c_p]),
    "cover_softmax_rows_f32": (c_i, [c_p, c_i, c_i, c_i, c_f, c_p]),
    "cover_l2norm_rows_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_p]),
    "cover_add_f32": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p]),
    "cover_xent_diag_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "cover_act_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p]),
    "cover_mha_f32": (c_i, [_P(MhaF32Args), c_p]),
    "cover_masked_mean_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "cover_sincos_time_embed": (c_i, [c_p, c_i, c_i, C.c_double, C.c_double, c_p, c_i, c_p]),
    "cover_token_select": (c_i, [_P(TokenSelectArgs), c_p]),
    "cover_score_select": (c_i, [_P(ScoreSelectArgs), c_p]),
    "cover_group_argmax": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p]),
    "cover_tokens_to_histories": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_f, c_p, c_p, c_p]),
    "cover_tokens_to_histories_steps": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_i, c_f, c_p, c_p, c_p]),
    "cover_actions_to_histories": (c_i, [c_p, c_ll, c_ll, c_i, c_i, c_p, c_p, c_i, c_f, c_p, c_p, c_p]),
    "cover_resample_axis": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p]),
    "cover_u8_hwc_to_f32_chw_norm": (c_i, [c_p, c_p, c_i, c_i, _P(c_f), _P(c_f), c_p]),
    "cover_u8_hwc_to_f32_chw_scale_norm": (c_i, [c_p, c_p, c_i, c_i, c_f, _P(c_f), _P(c_f), c_p]),
    "cover_resize_bilinear_pad_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
    "cover_vit_workspace_bytes": (C.c_size_t, [_P(VitDesc), c_i, c_i]),
    "cover_vit_forward": (c_i, [_P(VitDesc), c_p, c_i, c_i, c_p, Workspace, c_i, c_p]),
    "cover_decoder_workspace_bytes": (C.c_size_t, [_P(DecDesc), c_i]),
    "cover_decoder_forward": (c_i, [_P(DecDesc), _P(DecPass), c_p, Workspace, c_i, c_p]),
    "cover_graph_begin": (c_i, [c_p]),
    "cover_graph_end": (c_i, [c_p, _P(c_p)]),
    "cover_graph_launch": (c_i, [c_p, c_p]),
    "cover_graph_destroy": (c_i, [c_p]),
    "cover_timer_create": (c_i, [_P(c_p)]),
    "cover_timer_start": (c_i, [c_p, c_p]),
    "cover_timer_stop": (c_i, [c_p, c_p, _P(c_f)]),
    "cover_timer_destroy": (c_i, [c_p]),
    "cover_stream_sync": (c_i, [c_p]),
    "cover_profile_begin": (c_i, [c_i]),
    "cover_profile_end": (c_i, [_P(C.c_double), _P(C.c_longlong), _P(C.c_double)]),
    "cover_profile_end_n": (c_i, [_P(C.c_double), _P(C.c_longlong), _P(C.c_double), c_i]),
    "cover_sizeof": (C.c_size_t, [C.c_char_p]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libcover_hip.so (once). Raises CoverError if it has not been built: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CoverError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). cover_vla_amd has no CPU fallback.")
    h = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(h, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = h
    check_abi()
    return _lib


def check_abi() -> None:
    h = _lib
    v = h.cover_abi_version()
    if v != 1:
        raise CoverError(f"libcover_hip ABI version {v}, binding expects 1")
    for cname, st in _STRUCTS.items():
        n = h.cover_sizeof(cname.encode())
        if n != C.sizeof(st):
            raise CoverError(f"struct {cname}: library sizeof {n} != binding sizeof {C.sizeof(st)}")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = _lib.cover_last_error().decode(errors="replace") if _lib is not None else ""
        raise CoverError(f"{what or 'libcover_hip call'} failed ({rc}): {msg}")
