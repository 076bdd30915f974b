"""Tensor-level wrappers over the C ABI (include/cover_hip.h).

PyTorch-ROCm is plumbing here: it owns device memory and the stream; every arithmetic call below goes through
libcover_hip.so. Nothing in this module computes on the CPU or through torch kernels.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib as L

ACT = {"none": 0, "gelu_tanh": 1, "gelu_erf": 2, "silu": 3, "relu": 4}
MASK_LEN, MASK_CAUSAL, MASK_VISLEN = 0, 1, 2


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.CoverError("cover_vla_amd ops need device tensors: there is no CPU path")


# ------------------------------------------------------------------------------------------------ GEMM
@dataclass
class PackedLinear:
    """nn.Linear weight in the MFMA fragment-major layout (+ fp32 bias)."""
    wp: torch.Tensor            # packed bf16 storage
    N: int                      # rows of the packed matrix (2*mlp for GLU)
    K: int
    bias: Optional[torch.Tensor] = None   # fp32 [N]
    glu: bool = False
    w8: Optional[torch.Tensor] = None     # e4m3 twin of the same quantised weight (uint8 storage), cover_pack_weight_fp8
    w8s: Optional[torch.Tensor] = None    # fp32 per-channel power-of-two scales in packed channel order
    use_w8: bool = True                   # tests: False reads the bf16 image in the weight-streaming kernels too
    klinear: bool = False                 # w8 is the k-linear image (cover_pack_weight_fp8_klinear): the operand order of MX block-scaled activations

    @property
    def n_out(self) -> int:
        return self.N // 2 if self.glu else self.N

    @property
    def kp(self) -> int:
        return (self.K + 127) // 128 * 128


def pack_linear(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, glu: bool = False, fp8: bool = False,
                klinear: bool = False) -> PackedLinear:
    """weight: [N, K] (nn.Linear layout) on the device; for glu=True weight = cat([gate, up], 0).
    fp8=True: the weight is QUANTISED to e4m3 with per-output-channel power-of-two scales (cover_quantize_rows_fp8) and kept
    twice -- as the e4m3 image the HBM-bound weight-streaming kernels read (half the bytes) and as the bf16 image of the same
    de-quantised values for the MFMA-bound tiled kernels; both give bit-identical results. klinear=True (fp8, not glu): the e4m3 image in
    the k-linear operand order, which the MX block-scaled activations of quantize_act_fp8_mx / a GLU GEMM's out8 pair with (config 5's down_proj)."""
    assert not (klinear and (glu or not fp8))
    _chk_dev(weight)
    w = weight.to(torch.bfloat16).contiguous()
    N, K = w.shape
    h = L.lib()
    w8 = w8s = None
    if fp8:
        scales = torch.empty(N, dtype=torch.float32, device=w.device)
        wdq = torch.empty_like(w)
        L.check(h.cover_quantize_rows_fp8(w.data_ptr(), K, N, K, scales.data_ptr(), wdq.data_ptr(), _stream()), "quantize_rows_fp8")
        w8 = torch.empty(h.cover_packed_weight_fp8_bytes(N, K), dtype=torch.uint8, device=w.device)
        w8s = torch.empty((N + 15) // 16 * 16, dtype=torch.float32, device=w.device)
        if klinear:
            L.check(h.cover_pack_weight_fp8_klinear(wdq.data_ptr(), K, scales.data_ptr(), N, K, w8.data_ptr(), w8s.data_ptr(), _stream()),
                    "pack_weight_fp8_klinear")
        else:
            L.check(h.cover_pack_weight_fp8(wdq.data_ptr(), K, scales.data_ptr(), N, K, w8.data_ptr(), w8s.data_ptr(), 1 if glu else 0,
                                            _stream()), "pack_weight_fp8")
        w = wdq
    nbytes = h.cover_packed_weight_bytes(N, K)
    wp = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
    L.check(h.cover_pack_weight_bf16(w.data_ptr(), K, N, K, wp.data_ptr(), 1 if glu else 0, _stream()), "pack_weight")
    # the bias of a bf16 nn.Linear is a bf16 parameter in the reference (paligemma.to(bf16), HF bf16 checkpoints): round it
    # ONCE at load; the epilogue then adds exactly that value in fp32
    b = None if bias is None else bias.detach().to(torch.bfloat16).to(torch.float32).contiguous().to(w.device)
    return PackedLinear(wp, N, K, b, glu, w8, w8s, klinear=klinear)


def quantize_act_fp8(x: torch.Tensor, K: Optional[int] = None):
    """bf16 [M, >= K] rows -> (e4m3 rows uint8 [M, padded K] in the MX MFMA operand's k order, fp32 [M] power-of-two row scales)."""
    _chk_dev(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    M = x.shape[0]
    K = x.shape[1] if K is None else K
    kp = (K + 127) // 128 * 128
    q = torch.empty(M, kp, dtype=torch.uint8, device=x.device)
    sc = torch.empty(M, dtype=torch.float32, device=x.device)
    L.check(L.lib().cover_quantize_act_fp8(x.data_ptr(), x.stride(0), M, K, q.data_ptr(), kp, sc.data_ptr(), _stream()), "quantize_act_fp8")
    return q, sc


def quantize_act_fp8_mx(x: torch.Tensor, K: Optional[int] = None):
    """bf16 [M, >= K] rows -> (e4m3 rows uint8 [M, padded K], PLAIN row-major; E8M0 block scales uint8 [padded K / 128, M, 4]: one power-of-two
    scale per 32 consecutive k of a row) -- cover_quantize_act_fp8_mx, the operand form of a k-linear fp8 weight."""
    _chk_dev(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    M = x.shape[0]
    K = x.shape[1] if K is None else K
    kp = (K + 127) // 128 * 128
    q = torch.empty(M, kp, dtype=torch.uint8, device=x.device)
    mx = torch.empty(kp // 128, M, 4, dtype=torch.uint8, device=x.device)
    L.check(L.lib().cover_quantize_act_fp8_mx(x.data_ptr(), x.stride(0), M, K, q.data_ptr(), kp, mx.data_ptr(), _stream()), "quantize_act_fp8_mx")
    return q, mx


def gemm_workspace(M: int, N: int, K: int, device) -> Optional[torch.Tensor]:
    n = L.lib().cover_gemm_workspace_bytes(M, N, K)
    return torch.empty(max(n, 4) // 4, dtype=torch.float32, device=device) if n else None


def gemm(a: torch.Tensor, lin: PackedLinear, *, act: str = "none", residual: Optional[torch.Tensor] = None,
         layer_scale: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, out_f32: bool = False,
         out_scale: float = 1.0, variant: int = 0, ws: Optional[torch.Tensor] = None, norm_w: Optional[torch.Tensor] = None,
         norm_out: Optional[torch.Tensor] = None, norm_style: int = 0, norm_w_offset: float = 0.0, norm_eps: float = 1e-6,
         norm_b: Optional[torch.Tensor] = None, a8: Optional[tuple] = None, out8: Optional[tuple] = None) -> torch.Tensor:
    """out[M, n_out] = epi(a[M, K] @ W^T). `a` is bf16 with row stride >= padded K (zero padded).
    a8 = (q uint8 [M, >= padded K], scales fp32 [M]) from quantize_act_fp8: with an fp8 weight twin and M > 64 the GEMM runs on the
    MX-scaled fp8 matrix instruction (config 5) on those operands. With a k-linear weight twin a8 = (q, mx uint8 [kp / 128, M, 4]) from
    quantize_act_fp8_mx (or another GEMM's out8). out8 = (q uint8 [M, >= n_out], mx uint8 [ceil(n_out / 128), M, 4]): a GLU GEMM on the fp8
    tiles writes its output rows block-quantised into them INSTEAD of `out` (which is returned untouched)."""
    _chk_dev(a, residual, out)
    assert a.dtype == torch.bfloat16 and a.dim() == 2 and a.stride(1) == 1
    M = a.shape[0]
    h = L.lib()
    if out is None:
        out = torch.empty(M, lin.n_out, dtype=torch.float32 if out_f32 else torch.bfloat16, device=a.device)
    e = L.GemmEpi()
    e.bias = _ptr(lin.bias)
    e.residual = _ptr(residual)
    e.ld_residual = residual.stride(0) if residual is not None else 0
    e.residual_f32 = 1 if (residual is not None and residual.dtype == torch.float32) else 0
    e.layer_scale = _ptr(layer_scale)
    e.act = ACT[act]
    e.glu = 1 if lin.glu else 0
    e.out_f32 = 1 if out.dtype == torch.float32 else 0
    e.out_scale = out_scale
    if lin.w8 is not None and lin.use_w8:
        e.w8, e.w8_scale = lin.w8.data_ptr(), lin.w8s.data_ptr()
        e.w8_klinear = 1 if lin.klinear else 0
        if a8 is not None:
            q, qs = a8
            _chk_dev(q, qs)
            assert q.dtype == torch.uint8 and q.shape[0] == M and q.stride(1) == 1
            if lin.klinear:
                assert qs.dtype == torch.uint8 and qs.is_contiguous() and tuple(qs.shape) == (lin.kp // 128, M, 4)
                e.a8, e.a8_mx, e.ld_a8 = q.data_ptr(), qs.data_ptr(), q.stride(0)
            else:
                assert qs.dtype == torch.float32
                e.a8, e.a8_scale, e.ld_a8 = q.data_ptr(), qs.data_ptr(), q.stride(0)
    if out8 is not None:
        q, qs = out8
        _chk_dev(q, qs)
        assert q.dtype == torch.uint8 and q.shape[0] == M and q.stride(1) == 1 and qs.dtype == torch.uint8 and qs.is_contiguous()
        assert tuple(qs.shape) == ((lin.n_out + 127) // 128, M, 4)
        e.out8, e.out8_mx, e.ld_out8 = q.data_ptr(), qs.data_ptr(), q.stride(0)
    if norm_w is not None:
        e.norm_w, e.norm_out, e.ld_norm_out = norm_w.data_ptr(), norm_out.data_ptr(), norm_out.stride(0)
        e.norm_style, e.norm_w_offset, e.norm_eps = norm_style, norm_w_offset, norm_eps
        e.norm_b = _ptr(norm_b)
    if ws is None:
        ws = gemm_workspace(M, lin.N, lin.K, a.device)   # None when this shape needs no split-K scratch
    L.check(h.cover_gemm_bf16(a.data_ptr(), a.stride(0), lin.wp.data_ptr(), out.data_ptr(), out.stride(0), M, lin.N,
                              lin.K, C.byref(e), _ptr(ws), ws.numel() * 4 if ws is not None else 0, variant,
                              _stream()), "gemm_bf16")
    return out


def gemm_probe() -> dict:
    """In-kernel probe of the most recent self-loading tiled GEMM launch (cover_gemm_probe): microseconds of prologue / k-loop / epilogue of
    its first workgroup, k-tiles, shader cycles per k-tile and the clock (GHz) the loop ran at. Synchronises the device."""
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    L.check(L.lib().cover_gemm_probe(buf), "gemm_probe")
    w0, w1, w2, w3, c0, c1, nk = (int(buf[i]) for i in range(7))
    e4, e5, e6 = (int(buf[i]) for i in (12, 13, 14))
    loop_us = (w2 - w1) * 0.01
    staged = w2 <= e4 <= e5 <= e6 <= w3          # (the direct epilogue of a ragged tile leaves older stamps there)
    return dict(prologue_us=(w1 - w0) * 0.01, loop_us=loop_us, epilogue_us=(w3 - w2) * 0.01, k_tiles=nk,
                cycles_per_k_tile=(c1 - c0) / max(nk, 1), clock_ghz=(c1 - c0) / max(loop_us, 1e-9) / 1e3,
                epilogue_split_us=[(e4 - w2) * 0.01, (e5 - e4) * 0.01, (e6 - e5) * 0.01, (w3 - e6) * 0.01] if staged else None)


def gemm_plan_counts(reset: bool = False) -> list:
    """Launch counters per GEMM kernel plan since the last reset (cover_gemm_plan_counts; the map is in include/cover_hip.h): [0..8] gemm_tiled
    picks, [10] the 64x128 loader-wave tile, [19] / [20] / [22] weight-streaming generations 2 / 3 / 1, [21] fp8 MFMA tiles, self-loading tiles
    (gemm_v3.hip) 8 waves [23] 224x192, [24] 224x128, [25] 256x128, [26] 128x256, 4 waves [27] 224x96, k-split
    wave pairs [30] 224x96."""
    n = 32
    buf = (C.c_longlong * n)()
    L.lib().cover_gemm_plan_counts(buf, n, 1 if reset else 0)
    return list(buf)


# ------------------------------------------------------------------------------------------------ attention
@dataclass
class Segment:
    k: torch.Tensor                 # any bf16 tensor; strides given explicitly (elements)
    vt: torch.Tensor
    k_strides: Sequence[int]        # (slot, t, h)
    vt_strides: Sequence[int]       # (slot, h, d)
    length: int = 0
    len_of_batch: Optional[torch.Tensor] = None   # int32 [B]
    slot_of_batch: Optional[torch.Tensor] = None  # int32 [B]
    mask: int = MASK_LEN
    causal_offset: int = 0
    vis_len: Optional[torch.Tensor] = None        # int32 [Tq]
    k_offset: int = 0               # element offsets added to the base pointers
    vt_offset: int = 0


def fill_segment(s: L.KvSegment, seg: Segment) -> None:
    s.k = seg.k.data_ptr() + 2 * seg.k_offset
    s.vt = seg.vt.data_ptr() + 2 * seg.vt_offset
    s.k_slot_stride, s.k_t_stride, s.k_h_stride = seg.k_strides
    s.vt_slot_stride, s.vt_h_stride, s.vt_d_stride = seg.vt_strides
    s.slot_of_batch = _ptr(seg.slot_of_batch)
    s.len_of_batch = _ptr(seg.len_of_batch)
    s.vis_len = _ptr(seg.vis_len)
    s.len = seg.length
    s.mask_mode = seg.mask
    s.causal_offset = seg.causal_offset


def attention(q: torch.Tensor, q_strides: Sequence[int], out: Optional[torch.Tensor], o_strides: Sequence[int], B: int, Tq: int,
              Hq: int, Hkv: int, D: int, scale: float, segments: Sequence[Segment], state_in=None, state_out=None, out8=None):
    """state_in / state_out: optional (o fp32 [B,Tq,Hq,D], ml fp32 [B,Tq,Hq,2]) pairs chaining calls over KV segments.
    out8 = (q uint8 [rows, Hq * D], mx uint8 [Hq * D / 128, rows, 4]): the output rows block-quantised (quantize_act_fp8_mx's form) INSTEAD of `out`
    (MHA, D = 128, few query tiles: cover_attn_args.out8)."""
    _chk_dev(q, out)
    a = L.AttnArgs()
    a.q, a.out = q.data_ptr(), _ptr(out)
    if out8 is not None:
        _chk_dev(out8[0], out8[1])
        assert out8[0].dtype == torch.uint8 and out8[1].dtype == torch.uint8 and out8[1].is_contiguous() and out8[0].is_contiguous()
        a.out8, a.out8_mx, a.out8_rows = out8[0].data_ptr(), out8[1].data_ptr(), out8[0].shape[0]
    if state_in is not None:
        a.state_in_o, a.state_in_ml = state_in[0].data_ptr(), state_in[1].data_ptr()
    if state_out is not None:
        a.state_out_o, a.state_out_ml = state_out[0].data_ptr(), state_out[1].data_ptr()
    a.q_b_stride, a.q_t_stride, a.q_h_stride = q_strides
    a.o_b_stride, a.o_t_stride, a.o_h_stride = o_strides
    a.B, a.Tq, a.Hq, a.Hkv, a.D, a.scale = B, Tq, Hq, Hkv, D, scale
    a.n_seg = len(segments)
    for i, sg in enumerate(segments):
        fill_segment(a.seg[i], sg)
    L.check(L.lib().cover_attention_bf16(C.byref(a), _stream()), "attention_bf16")
    return out


def decode_attention_fused(qkv, N, H, D, scale, segments, write_t, out, *, positions=None, cos=None, sin=None, rope_mode=0,
                           partial=None, bias=None):
    """RoPE + KV append + [shared | per-prompt | own] attention for one new token per candidate, one launch."""
    _chk_dev(qkv, out)
    a = L.DecodeAttnArgs()
    a.qkv, a.ld_qkv = qkv.data_ptr(), qkv.stride(0)
    if partial is not None:
        a.n_splits, a.partial, a.bias = partial.shape[0], partial.data_ptr(), _ptr(bias)
    a.N, a.H, a.D, a.scale = N, H, D, scale
    a.positions, a.cos_table, a.sin_table = _ptr(positions), _ptr(cos), _ptr(sin)
    a.n_pos = cos.shape[0] if cos is not None else 0
    a.rope_mode = rope_mode
    for i, sg in enumerate(segments):
        fill_segment(a.seg[i], sg)
    a.write_t = write_t
    a.out, a.out_row_stride = out.data_ptr(), out.stride(0)
    L.check(L.lib().cover_decode_attention_fused(C.byref(a), _stream()), "decode_attention_fused")
    return out


def decode_own_attention(qkv, N, H, D, scale, k_own, v_own, t_cap, write_t, state, *, positions=None, cos=None, sin=None, rope_mode=0,
                         k_scale=None, v_scale=None, slot_of_batch=None):
    """Large-N candidate decode, own-token pass (cover_decode_own_attention): RoPE + append into the head-major own cache
    (k_own / v_own [slots][H][t_cap][D], bf16 or uint8 e4m3 + fp32 row scales [slots][H][t_cap]) + attention over keys 0..write_t;
    state = (o fp32 [N,H,D], ml fp32 [N,H,2]) for ops.attention(..., state_in=state)."""
    _chk_dev(qkv, k_own, v_own, state[0], state[1])
    a = L.OwnAttnArgs()
    a.qkv, a.ld_qkv = qkv.data_ptr(), qkv.stride(0)
    a.N, a.H, a.D, a.scale = N, H, D, scale
    a.positions, a.cos_table, a.sin_table = _ptr(positions), _ptr(cos), _ptr(sin)
    a.n_pos = cos.shape[0] if cos is not None else 0
    a.rope_mode = rope_mode
    a.k, a.v = k_own.data_ptr(), v_own.data_ptr()
    a.fp8 = 1 if k_own.dtype == torch.uint8 else 0
    a.k_scale, a.v_scale = _ptr(k_scale), _ptr(v_scale)
    a.t_cap, a.slot_stride = t_cap, H * t_cap * D
    a.slot_of_batch = _ptr(slot_of_batch)
    a.write_t = write_t
    a.state_o, a.state_ml = state[0].data_ptr(), state[1].data_ptr()
    L.check(L.lib().cover_decode_own_attention(C.byref(a), _stream()), "decode_own_attention")
    return state


# ------------------------------------------------------------------------------------------------ row kernels
def layernorm(x, w, b, eps, out=None):
    _chk_dev(x, w)
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().cover_layernorm_bf16(x.data_ptr(), x.stride(0), w.data_ptr(), _ptr(b), out.data_ptr(), out.stride(0),
                                         x.shape[0], x.shape[1], eps, _stream()), "layernorm_bf16")
    return out


def rmsnorm(x, w, eps, w_offset=0.0, style=0, out=None):
    _chk_dev(x, w)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().cover_rmsnorm_bf16(x.data_ptr(), 1 if x.dtype == torch.float32 else 0, x.stride(0), _ptr(w), w_offset,
                                       style, out.data_ptr(), out.stride(0), x.shape[0], x.shape[1], eps, _stream()),
            "rmsnorm_bf16")
    return out


def rope_kv_write(qkv, B, T, Hq, Hkv, D, *, positions=None, cos=None, sin=None, rope_mode=0, k_cache=None,
                  k_strides=(0, 0, 0), k_offset=0, vt_cache=None, vt_strides=(0, 0, 0), vt_offset=0, slot_of_batch=None,
                  t_offset_of_batch=None, t_offset=0):
    _chk_dev(qkv, vt_cache)
    a = L.RopeArgs()
    a.qkv, a.ld_qkv = qkv.data_ptr(), qkv.stride(0)
    a.B, a.T, a.Hq, a.Hkv, a.D = B, T, Hq, Hkv, D
    a.positions = _ptr(positions)
    a.cos_table, a.sin_table = _ptr(cos), _ptr(sin)
    a.n_pos = cos.shape[0] if cos is not None else 0
    a.rope_mode = rope_mode
    a.k_cache = None if k_cache is None else k_cache.data_ptr() + 2 * k_offset
    a.k_slot_stride, a.k_t_stride, a.k_h_stride = k_strides
    a.vt_cache = vt_cache.data_ptr() + 2 * vt_offset
    a.vt_slot_stride, a.vt_h_stride, a.vt_d_stride = vt_strides
    a.slot_of_batch = _ptr(slot_of_batch)
    a.t_offset_of_batch = _ptr(t_offset_of_batch)
    a.t_offset = t_offset
    L.check(L.lib().cover_rope_kv_write(C.byref(a), _stream()), "rope_kv_write")


def embed_gather(table, ids, scale=1.0, out=None):
    _chk_dev(table, ids)
    n, dim = ids.numel(), table.shape[1]
    if out is None:
        out = torch.empty(n, dim, dtype=torch.bfloat16, device=table.device)
    L.check(L.lib().cover_embed_gather(table.data_ptr(), dim, ids.data_ptr(), n, scale, out.data_ptr(), out.stride(0),
                                       _stream()), "embed_gather")
    return out


def patchify(img, patch, mul, add, ld_out, out=None):
    """img: uint8 [n,H,W,3] or fp32 [n,3,H,W] -> bf16 [n*nP, ld_out] rows (k = c*p*p + py*p + px)."""
    _chk_dev(img)
    a = L.PatchifyArgs()
    if img.dtype == torch.uint8:
        n, H, W, _ = img.shape
        a.in_u8_hwc = 1
    else:
        assert img.dtype == torch.float32
        n, _, H, W = img.shape
        a.in_u8_hwc = 0
    img = img.contiguous()
    a.img, a.H, a.W, a.patch, a.n_img, a.img_stride = img.data_ptr(), H, W, patch, n, 3 * H * W
    for i in range(3):
        a.mul[i], a.add[i] = float(mul[i]), float(add[i])
    rows = n * (H // patch) * (W // patch)
    if out is None:
        out = torch.empty(rows, ld_out, dtype=torch.bfloat16, device=img.device)
    a.out, a.ld_out = out.data_ptr(), out.stride(0)
    L.check(L.lib().cover_patchify(C.byref(a), _stream()), "patchify")
    return out


def copy_rows(src, dst, rows, cols, src_idx=None, dst_idx=None):
    _chk_dev(src, dst)
    L.check(L.lib().cover_copy_rows_bf16(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), rows, cols,
                                         _ptr(src_idx), _ptr(dst_idx), _stream()), "copy_rows")


def add_rows(x, add):
    _chk_dev(x, add)
    L.check(L.lib().cover_add_rows_bf16(x.data_ptr(), x.stride(0), add.data_ptr(), add.stride(0), x.shape[0], x.shape[1],
                                        add.shape[0], _stream()), "add_rows")
    return x


def scale_bf16(x, pre_div, post_mul):
    _chk_dev(x)
    L.check(L.lib().cover_scale_bf16(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], pre_div, post_mul, _stream()),
            "scale_bf16")
    return x


def cast_f32_to_bf16(x, out=None):
    _chk_dev(x)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().cover_cast_f32_to_bf16(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0],
                                           x.shape[1], _stream()), "cast")
    return out


def cast_bf16_to_f32(x, out=None):
    _chk_dev(x)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    L.check(L.lib().cover_cast_bf16_to_f32(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0],
                                           x.shape[1], _stream()), "cast")
    return out


# ------------------------------------------------------------------------------------------------ fp32 kernels
def gemm_f32(a, b, *, bias=None, act="none", alpha=1.0, residual=None, out=None, b_is_kn=False, batch=1,
             a_bs=0, b_bs=0, c_bs=0, M=None, N=None, K=None, bias_bs=0):
    """C[m,n] = residual + alpha*act(sum_k a[m,k] b[n,k] + bias[n]). b_is_kn: b given as [K, N] (k-major)."""
    _chk_dev(a, b)
    g = L.GemmF32Args()
    M = a.shape[-2] if M is None else M
    K = a.shape[-1] if K is None else K
    if N is None:
        N = b.shape[-1] if b_is_kn else b.shape[-2]
    g.A, g.a_row_stride, g.a_k_stride = a.data_ptr(), a.stride(-2), a.stride(-1)
    g.B = b.data_ptr()
    if b_is_kn:
        g.b_row_stride, g.b_k_stride = b.stride(-1), b.stride(-2)
    else:
        g.b_row_stride, g.b_k_stride = b.stride(-2), b.stride(-1)
    if out is None:
        shape = (batch, M, N) if batch > 1 else (M, N)
        out = torch.empty(shape, dtype=torch.float32, device=a.device)
    g.C, g.c_row_stride = out.data_ptr(), out.stride(-2)
    g.bias, g.residual = _ptr(bias), _ptr(residual)
    g.ld_residual = residual.stride(-2) if residual is not None else 0
    g.M, g.N, g.K, g.act, g.alpha = M, N, K, ACT[act], alpha
    g.batch, g.a_batch_stride, g.b_batch_stride, g.c_batch_stride = batch, a_bs, b_bs, c_bs
    g.bias_batch_stride = bias_bs
    L.check(L.lib().cover_gemm_f32(C.byref(g), _stream()), "gemm_f32")
    return out


def gemm_f32_raw(a_ptr, a_rs, a_ks, b_ptr, b_rs, b_ks, c_ptr, c_rs, M, N, K, *, bias=None, residual_ptr=None, ld_res=0,
                 act="none", alpha=1.0, batch=1, a_bs=0, b_bs=0, c_bs=0):
    """Pointer-level form of gemm_f32 (element strides) for strided / batched views that torch cannot express."""
    g = L.GemmF32Args()
    g.A, g.a_row_stride, g.a_k_stride = a_ptr, a_rs, a_ks
    g.B, g.b_row_stride, g.b_k_stride = b_ptr, b_rs, b_ks
    g.C, g.c_row_stride = c_ptr, c_rs
    g.bias, g.residual, g.ld_residual = _ptr(bias), residual_ptr, ld_res
    g.M, g.N, g.K, g.act, g.alpha = M, N, K, ACT[act], alpha
    g.batch, g.a_batch_stride, g.b_batch_stride, g.c_batch_stride = batch, a_bs, b_bs, c_bs
    L.check(L.lib().cover_gemm_f32(C.byref(g), _stream()), "gemm_f32")


def layernorm_f32(x, w, b, eps=1e-5, out=None):
    _chk_dev(x)
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().cover_layernorm_f32(x.data_ptr(), x.stride(0), _ptr(w), _ptr(b), out.data_ptr(), out.stride(0),
                                        x.shape[0], x.shape[1], eps, _stream()), "layernorm_f32")
    return out


def layernorm_f32_grouped(x, w, b, rows_per_group, eps=1e-5, out=None):
    """x fp32 [G * rows_per_group, dim]; w, b fp32 [G, dim]: group g's rows use w[g], b[g] (ensemble members in one launch)."""
    _chk_dev(x, w)
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().cover_layernorm_f32_grouped(x.data_ptr(), x.stride(0), w.data_ptr(), _ptr(b), out.data_ptr(), out.stride(0),
                                                x.shape[0], x.shape[1], eps, rows_per_group, w.stride(0), _stream()),
            "layernorm_f32_grouped")
    return out


def softmax_rows_f32(x, scale=1.0):
    _chk_dev(x)
    L.check(L.lib().cover_softmax_rows_f32(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], scale, _stream()), "softmax")
    return x


def l2norm_rows_f32(x, out=None):
    _chk_dev(x)
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().cover_l2norm_rows_f32(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0],
                                          x.shape[1], _stream()), "l2norm")
    return out


def add_f32(a, b, out=None):
    _chk_dev(a, b)
    out = torch.empty_like(a) if out is None else out
    L.check(L.lib().cover_add_f32(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0),
                                  a.shape[0], a.shape[1], b.shape[0], _stream()), "add_f32")
    return out


def xent_diag_f32(logits):
    """logits fp32 [B, C] (B <= C) -> (loss fp32 [B] = logsumexp(row) - row[r], rank int32 [B] = entries ranked ahead of row[r])."""
    _chk_dev(logits)
    B, Cn = logits.shape
    loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    rank = torch.empty(B, dtype=torch.int32, device=logits.device)
    L.check(L.lib().cover_xent_diag_f32(logits.data_ptr(), logits.stride(0), B, Cn, loss.data_ptr(), rank.data_ptr(), _stream()),
            "xent_diag_f32")
    return loss, rank


def act_f32(x, act, out=None):
    _chk_dev(x)
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().cover_act_f32(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1], ACT[act], _stream()),
            "act_f32")
    return out


def mha_f32(q, k, v, B, Tq, Tk, H, Dh, q_strides, k_strides, v_strides, key_pad=None, out=None, o_strides=None):
    """strides = (batch, token) in elements; heads are contiguous blocks of Dh inside a token row."""
    _chk_dev(q, k, v)
    if out is None:
        out = torch.empty(B, Tq, H * Dh, dtype=torch.float32, device=q.device)
        o_strides = (Tq * H * Dh, H * Dh)
    a = L.MhaF32Args()
    a.q, a.q_b_stride, a.q_t_stride = q.data_ptr(), q_strides[0], q_strides[1]
    a.k, a.k_b_stride, a.k_t_stride = k.data_ptr(), k_strides[0], k_strides[1]
    a.v, a.v_b_stride, a.v_t_stride = v.data_ptr(), v_strides[0], v_strides[1]
    a.out, a.o_b_stride, a.o_t_stride = out.data_ptr(), o_strides[0], o_strides[1]
    a.key_pad = _ptr(key_pad)
    a.B, a.Tq, a.Tk, a.H, a.Dh, a.scale = B, Tq, Tk, H, Dh, Dh ** -0.5
    L.check(L.lib().cover_mha_f32(C.byref(a), _stream()), "mha_f32")
    return out


def masked_mean_f32(x, pad, B, T, D):
    _chk_dev(x)
    out = torch.empty(B, D, dtype=torch.float32, device=x.device)
    L.check(L.lib().cover_masked_mean_f32(x.data_ptr(), _ptr(pad), out.data_ptr(), B, T, D, _stream()), "masked_mean")
    return out


def sincos_time_embed(time, dim, min_period, max_period, out=None):
    _chk_dev(time)
    B = time.shape[0]
    if out is None:
        out = torch.empty(B, dim, dtype=torch.bfloat16, device=time.device)
    L.check(L.lib().cover_sincos_time_embed(time.data_ptr(), B, dim, min_period, max_period, out.data_ptr(), out.stride(0),
                                            _stream()), "sincos_time_embed")
    return out


# ------------------------------------------------------------------------------------------------ selection
def token_select(logits, lo, hi, uniform=None, temperature=1.0, out_tok=None, out_logit=None):
    """out_tok int64 [rows] / out_logit fp32 [rows] (contiguous, e.g. a row of a [steps, rows] buffer): written in place of fresh tensors."""
    _chk_dev(logits, uniform, out_tok, out_logit)
    rows = logits.shape[0]
    tok = torch.empty(rows, dtype=torch.int64, device=logits.device) if out_tok is None else out_tok
    lg = torch.empty(rows, dtype=torch.float32, device=logits.device) if out_logit is None else out_logit
    assert tok.dtype == torch.int64 and lg.dtype == torch.float32 and tok.is_contiguous() and lg.is_contiguous() and tok.numel() == rows == lg.numel()
    assert uniform is None or (uniform.is_contiguous() and uniform.dtype == torch.float32 and uniform.numel() == rows)
    a = L.TokenSelectArgs()
    a.logits, a.ld, a.rows, a.lo, a.hi = logits.data_ptr(), logits.stride(0), rows, lo, hi
    a.uniform, a.temperature = _ptr(uniform), temperature
    a.token_out, a.logit_out = tok.data_ptr(), lg.data_ptr()
    L.check(L.lib().cover_token_select(C.byref(a), _stream()), "token_select")
    return tok, lg


def score_select(it, act, group_size):
    """it [members, dim], act [members, N, dim] fp32 -> (scores [N], result int32[4], best f32[2], fused_it, fused_act)."""
    _chk_dev(it, act)
    m, N, dim = act.shape
    dev = it.device
    scores = torch.empty(N, dtype=torch.float32, device=dev)
    result = torch.empty(4, dtype=torch.int32, device=dev)
    best = torch.empty(2, dtype=torch.float32, device=dev)
    fit = torch.empty(dim, dtype=torch.float32, device=dev)
    fact = torch.empty(N, dim, dtype=torch.float32, device=dev)
    a = L.ScoreSelectArgs()
    a.it, a.act = it.contiguous().data_ptr(), act.contiguous().data_ptr()
    a.n_members, a.N, a.dim, a.group_size = m, N, dim, group_size
    a.scores_out, a.result_out, a.best_out = scores.data_ptr(), result.data_ptr(), best.data_ptr()
    a.fused_it_out, a.fused_act_out = fit.data_ptr(), fact.data_ptr()
    L.check(L.lib().cover_score_select(C.byref(a), _stream()), "score_select")
    return scores, result, best, fit, fact


def tokens_to_histories(tokens, tok_vocab, centers, past, pad_value=-5.0, n_use=1):
    """tokens int64 [N, >= 7 n_use] (device), centers fp32 [n_centers], past fp32 [n_past, 7] -> (hist fp32 [N,10,7], pad uint8
    [N,10]); n_use = how many 7-token actions of a candidate's chunk become history rows (action-chunk horizon > 1)."""
    _chk_dev(tokens, centers)
    N = tokens.shape[0]
    hist = torch.empty(N, 10, 7, dtype=torch.float32, device=tokens.device)
    pad = torch.empty(N, 10, dtype=torch.uint8, device=tokens.device)
    n_past = 0 if past is None else past.shape[0]
    L.check(L.lib().cover_tokens_to_histories_steps(tokens.data_ptr(), tokens.stride(0), N, tok_vocab, centers.data_ptr(),
                                                    centers.numel(), _ptr(past), n_past, n_use, pad_value, hist.data_ptr(),
                                                    pad.data_ptr(), _stream()), "tokens_to_histories")
    return hist, pad


def actions_to_histories(actions, n_use, past, lo_hi=None, pad_value=-5.0):
    """actions fp32 [N, chunk, >=7] (device, normalised policy output), past fp32 [n_past, 7], lo_hi fp32 [12] (p01[:6] | p99[:6])
    or None -> (hist fp32 [N,10,7], pad uint8 [N,10]): the verifier histories of a flow-matching policy's chunks."""
    _chk_dev(actions)
    assert actions.dtype == torch.float32 and actions.stride(2) == 1
    N = actions.shape[0]
    hist = torch.empty(N, 10, 7, dtype=torch.float32, device=actions.device)
    pad = torch.empty(N, 10, dtype=torch.uint8, device=actions.device)
    n_past = 0 if past is None else past.shape[0]
    L.check(L.lib().cover_actions_to_histories(actions.data_ptr(), actions.stride(0), actions.stride(1), N, n_use, _ptr(lo_hi),
                                               _ptr(past), n_past, pad_value, hist.data_ptr(), pad.data_ptr(), _stream()),
            "actions_to_histories")
    return hist, pad


def group_argmax(scores, group_size):
    _chk_dev(scores)
    result = torch.empty(4, dtype=torch.int32, device=scores.device)
    best = torch.empty(2, dtype=torch.float32, device=scores.device)
    L.check(L.lib().cover_group_argmax(scores.data_ptr(), scores.numel(), group_size, result.data_ptr(), best.data_ptr(),
                                       _stream()), "group_argmax")
    return result, best


# ------------------------------------------------------------------------------------------------ graphs / timers
class Graph:
    """hipGraph captured from whatever runs on the current stream inside the `with` block."""

    def __init__(self):
        self.handle = C.c_void_p()
        self.stream = None

    def __enter__(self):
        self.stream = _stream()
        L.check(L.lib().cover_graph_begin(self.stream), "graph_begin")
        return self

    def __exit__(self, et, ev, tb):
        rc = L.lib().cover_graph_end(self.stream, C.byref(self.handle))
        if et is None:
            L.check(rc, "graph_end")
        return False

    def launch(self):
        L.check(L.lib().cover_graph_launch(self.handle, _stream()), "graph_launch")


class PooledGraph:
    """A launch sequence that ALLOCATES its intermediates (ordinary op wrappers) as one replayable hipGraph.

    call 1 runs `fn()` eagerly inside a private torch memory pool (this sizes every intermediate and leaves their blocks in the pool),
    call 2 captures it inside the same pool (every allocation is served from the pool's cache: no hipMalloc under capture), later calls
    replay. The pool belongs to this object alone, so the blocks the captured kernels write to are never handed to anybody else between
    replays. `fn` takes no arguments: it reads static input tensors the caller refreshes before every call, and returns tensors that stay
    valid (static) from the capture on. Streams that `fn` forks to and joins from become parallel branches of the graph."""

    def __init__(self, fn, device):
        self.fn, self.dev = fn, device
        self.pool = torch.cuda.MemPool()
        self.graph, self.out, self.calls = None, None, 0
        self._cap = None

    def __call__(self):
        self.calls += 1
        if self.graph is not None:
            self.graph.launch()
            return self.out
        if self.calls == 1:
            with torch.cuda.use_mem_pool(self.pool, device=self.dev):
                out = self.fn()
            # (this call's outputs live in the pool until the caller drops them; the capture below then allocates its own)
            return out
        cur = torch.cuda.current_stream()
        if self._cap is None:
            self._cap = torch.cuda.Stream(device=self.dev)
        self._cap.wait_stream(cur)
        with torch.cuda.use_mem_pool(self.pool, device=self.dev):
            with torch.cuda.stream(self._cap):
                with Graph() as g:
                    self.out = self.fn()
        cur.wait_stream(self._cap)
        self.graph = g
        self.graph.launch()          # the capture itself executed nothing
        return self.out


class Timer:
    """hipEvent pair recorded on the current stream."""

    def __init__(self):
        self.h = C.c_void_p()
        L.check(L.lib().cover_timer_create(C.byref(self.h)), "timer_create")

    def start(self):
        L.check(L.lib().cover_timer_start(self.h, _stream()), "timer_start")

    def stop(self) -> float:
        ms = C.c_float()
        L.check(L.lib().cover_timer_stop(self.h, _stream(), C.byref(ms)), "timer_stop")
        return ms.value
