"""Device-resident model pieces assembled from libcover_hip composites: ViT towers and Gemma/Llama decoders.

Host-side only: weight packing (once, at load), descriptor tables for the C ABI, KV-cache geometry. All arithmetic
is in the HIP library. Awkward sizes are zero-padded AT PACK TIME so that no kernel needs a special case:
SigLIP-So400m head_dim 72 -> 96 and MLP 4304 -> 4352, patch K 588 -> 640 (zero weight rows/cols keep the maths exact).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import ops

BF = torch.bfloat16


def _round_up(v, m):
    return (v + m - 1) // m * m


def _pad_head_dim(dh: int) -> int:
    for c in (64, 96, 128, 256):
        if dh <= c:
            return c
    raise L.CoverError(f"head_dim {dh} > 256 unsupported")


def _f32(t, dev):
    return t.detach().to(torch.float32).contiguous().to(dev)


def _bf_f32(t, dev):
    """A parameter the reference holds in bf16 (norm weights / biases / layer scales of a model cast with .to(bfloat16):
    paligemma_with_expert.py:216-227, HF bf16 checkpoints), kept as fp32 storage of the bf16-ROUNDED value: the kernels read
    fp32 vectors, the arithmetic sees exactly the reference's parameter."""
    return t.detach().to(BF).to(torch.float32).contiguous().to(dev)


# ------------------------------------------------------------------------------------------------ ViT tower
class VitTower:
    """Pre-LN ViT (SigLIP / SigLIP2 / DINOv2) from a neutral state dict (cover_vla_amd.synth.vit_state key layout).

    embed(): conv patch embedding as patchify + MFMA GEMM (K1) + position rows; forward(): blocks via
    cover_vit_forward. Replaces the un-vendored HF/timm towers the reference calls at
    paligemma_with_expert.py:229-230 and finetune_trajectory_bridge_ddp.py:314-316.
    """

    def __init__(self, sd: Dict[str, torch.Tensor], *, dim, layers, heads, mlp, patch, act, eps=1e-6, layerscale=False,
                 prefix_tokens=0, device="cuda:0", n_layers_used: Optional[int] = None):
        dev = torch.device(device)
        self.dev, self.dim, self.heads, self.patch = dev, dim, heads, patch
        self.dh = dim // heads
        self.dp = _pad_head_dim(self.dh)
        self.mlp_p = _round_up(mlp, 128)
        self.eps, self.act = eps, act
        self.prefix_tokens = prefix_tokens
        self.layers_total = layers
        nl = layers if n_layers_used is None else n_layers_used
        self.kpatch = 3 * patch * patch
        self.kpatch_p = _round_up(self.kpatch, 128)
        self.patch_lin = ops.pack_linear(sd["patch.weight"].to(dev), sd["patch.bias"])
        self.pos = sd["pos"].to(BF).contiguous().to(dev)
        self.prefix = sd["prefix"].to(BF).contiguous().to(dev) if prefix_tokens else None
        self.post_ln = (_bf_f32(sd["post_ln.weight"], dev), _bf_f32(sd["post_ln.bias"], dev)) if "post_ln.weight" in sd else None
        H, Dh, Dp = heads, self.dh, self.dp
        self._keep = []
        arr = (L.VitLayer * nl)()
        for i in range(nl):
            p = f"blocks.{i}."
            # fused qkv with per-head zero padding Dh -> Dp
            wqkv = torch.zeros(3, H, Dp, dim)
            bqkv = torch.zeros(3, H, Dp)
            for j, n in enumerate(("q", "k", "v")):
                wqkv[j, :, :Dh] = sd[p + n + ".weight"].float().view(H, Dh, dim)
                bqkv[j, :, :Dh] = sd[p + n + ".bias"].float().view(H, Dh)
            qkv = ops.pack_linear(wqkv.view(3 * H * Dp, dim).to(dev), bqkv.view(-1))
            wo = torch.zeros(dim, H, Dp)
            wo[:, :, :Dh] = sd[p + "o.weight"].float().view(dim, H, Dh)
            proj = ops.pack_linear(wo.view(dim, H * Dp).to(dev), sd[p + "o.bias"])
            w1 = torch.zeros(self.mlp_p, dim)
            w1[:mlp] = sd[p + "fc1.weight"].float()
            b1 = torch.zeros(self.mlp_p)
            b1[:mlp] = sd[p + "fc1.bias"].float()
            fc1 = ops.pack_linear(w1.to(dev), b1)
            w2 = torch.zeros(dim, self.mlp_p)
            w2[:, :mlp] = sd[p + "fc2.weight"].float()
            fc2 = ops.pack_linear(w2.to(dev), sd[p + "fc2.bias"])
            ln = [_bf_f32(sd[p + k], dev) for k in ("ln1.weight", "ln1.bias", "ln2.weight", "ln2.bias")]
            ls = [_bf_f32(sd[p + k], dev) for k in ("ls1", "ls2")] if layerscale else [None, None]
            self._keep += [qkv, proj, fc1, fc2, ln, ls]
            a = arr[i]
            a.ln1_w, a.ln1_b, a.ln2_w, a.ln2_b = (t.data_ptr() for t in ln)
            a.qkv_w, a.qkv_b = qkv.wp.data_ptr(), qkv.bias.data_ptr()
            a.proj_w, a.proj_b = proj.wp.data_ptr(), proj.bias.data_ptr()
            a.fc1_w, a.fc1_b = fc1.wp.data_ptr(), fc1.bias.data_ptr()
            a.fc2_w, a.fc2_b = fc2.wp.data_ptr(), fc2.bias.data_ptr()
            a.ls1 = ls[0].data_ptr() if ls[0] is not None else None
            a.ls2 = ls[1].data_ptr() if ls[1] is not None else None
        self._arr = arr
        self.n_layers = nl
        self._ws = None
        self.ws_gen = 0          # bumped whenever the workspace is re-allocated: a hipGraph captured over the old pointer must be re-captured

    def _desc(self, n_layers, last_attn_only):
        d = L.VitDesc()
        d.dim, d.heads, d.head_dim_p, d.mlp_p, d.n_layers = self.dim, self.heads, self.dp, self.mlp_p, n_layers
        d.act = ops.ACT[self.act]
        d.ln_eps, d.attn_scale = self.eps, self.dh ** -0.5
        d.layers_host = C.cast(self._arr, C.POINTER(L.VitLayer))
        d.last_attn_only = 1 if last_attn_only else 0
        return d

    def static_bufs(self, n: int, H: int, W: int) -> dict:
        """Persistent buffers for embed() of n images of H x W: nothing is allocated per call, so the tower can be recorded
        into a hipGraph (fixed addresses) and its host launch cost paid once."""
        P = (H // self.patch) * (W // self.patch)
        T = P + self.prefix_tokens
        dev = self.dev
        b = dict(rows=torch.empty(n * P, self.kpatch_p, dtype=BF, device=dev), x=torch.empty(n, T, self.dim, dtype=BF, device=dev),
                 ws=ops.gemm_workspace(n * P, self.patch_lin.N, self.patch_lin.K, dev))
        if self.prefix_tokens:
            b["tok"] = torch.empty(n * P, self.dim, dtype=BF, device=dev)
            b["idx_dst"] = (torch.arange(n, device=dev)[:, None] * T + self.prefix_tokens +
                            torch.arange(P, device=dev)[None]).reshape(-1).to(torch.int32)
            b["pidx"] = (torch.arange(n, device=dev)[:, None] * T + torch.arange(self.prefix_tokens, device=dev)[None]
                         ).reshape(-1).to(torch.int32)
            b["psrc"] = torch.arange(self.prefix_tokens, device=dev).repeat(n).to(torch.int32)
        return b

    def embed(self, pixels: torch.Tensor, mul=(1.0, 1.0, 1.0), add=(0.0, 0.0, 0.0), bufs: Optional[dict] = None) -> torch.Tensor:
        """pixels: fp32 [n,3,H,W] (normalised) or uint8 [n,H,W,3] (+ per-channel mul/add) -> bf16 [n, prefix+P, dim].
        bufs: static_bufs(...) of the same geometry (no allocation; pixels must be contiguous)."""
        n = pixels.shape[0]
        if bufs is None:
            H, W = (pixels.shape[1], pixels.shape[2]) if pixels.dtype == torch.uint8 else (pixels.shape[2], pixels.shape[3])
            bufs = self.static_bufs(n, H, W)
        rows = ops.patchify(pixels, self.patch, mul, add, self.kpatch_p, out=bufs["rows"])
        P = rows.shape[0] // n
        T = P + self.prefix_tokens
        x = bufs["x"]
        if self.prefix_tokens:
            tok = ops.gemm(rows, self.patch_lin, out=bufs["tok"], ws=bufs["ws"])
            xv = x.view(n * T, self.dim)
            ops.copy_rows(tok, xv, n * P, self.dim, None, bufs["idx_dst"])
            ops.copy_rows(self.prefix, xv, n * self.prefix_tokens, self.dim, bufs["psrc"], bufs["pidx"])
        else:
            ops.gemm(rows, self.patch_lin, out=x.view(n * T, self.dim), ws=bufs["ws"])
        ops.add_rows(x.view(n * T, self.dim), self.pos[:T])
        return x

    def forward(self, x: torch.Tensor, n_layers: Optional[int] = None, last_attn_only=False, post_ln=False,
                gemm_variant=0) -> torch.Tensor:
        """x bf16 [n, T, dim] (modified in place). Returns the hidden states, or the last listed block's attention
        module output when last_attn_only (the verifier's forward-hook feature)."""
        n, T, _ = x.shape
        nl = self.n_layers if n_layers is None else n_layers
        d = self._desc(nl, last_attn_only)
        h = L.lib()
        need = h.cover_vit_workspace_bytes(C.byref(d), n, T)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
            self.ws_gen += 1
        attn_out = torch.empty_like(x) if last_attn_only else None
        ws = L.Workspace(self._ws.data_ptr(), self._ws.numel())
        L.check(h.cover_vit_forward(C.byref(d), x.data_ptr(), n, T, attn_out.data_ptr() if last_attn_only else None, ws,
                                    gemm_variant, torch.cuda.current_stream().cuda_stream), "vit_forward")
        if last_attn_only:
            return attn_out
        if post_ln:
            ops.layernorm(x.view(n * T, self.dim), self.post_ln[0], self.post_ln[1], self.eps, out=x.view(n * T, self.dim))
        return x


# ------------------------------------------------------------------------------------------------ decoder
class KvGeometry:
    """Per-layer cache = one allocation holding up to 3 regions (segments). Region r has n_slots[r] slots of
    capacity cap[r] tokens (multiple of 32). K region layout [slot][t][Hkv][D]; V^T region layout [slot][Hkv][D][cap]."""

    def __init__(self, Hkv, D, slots: List[int], caps: List[int]):
        self.Hkv, self.D = Hkv, D
        self.slots = slots
        self.caps = [_round_up(c, 32) for c in caps]
        self.k_off, self.vt_off = [], []
        off = 0
        for s, c in zip(self.slots, self.caps):
            self.k_off.append(off)
            self.vt_off.append(off)
            off += s * c * Hkv * D
        self.elems = off

    def k_strides(self, r):
        return (self.caps[r] * self.Hkv * self.D, self.Hkv * self.D, self.D)

    def vt_strides(self, r):
        return (self.Hkv * self.D * self.caps[r], self.D * self.caps[r], self.caps[r])


class Decoder:
    """Gemma / Llama decoder stack from an HF-named state dict (synth.decoder_state layout).

    Replaces the layer loop of paligemma_with_expert.py:258-360 (and HF LlamaModel for the OpenVLA profile)."""

    def __init__(self, sd, *, dim, layers, Hq, Hkv, D, mlp, act, norm, eps, rope, n_pos=1024, device="cuda:0",
                 cache: Optional[KvGeometry] = None, share_cache_with: Optional["Decoder"] = None, final_norm_bf16=True,
                 fp8_weights=False):
        """final_norm_bf16: the stack's final norm weight is a bf16 parameter in the reference (PaliGemma's language model, HF
        bf16 Llama) -- False for the pi0 action expert, whose final norm is outside the name filter of
        to_bfloat16_like_physical_intelligence (paligemma_with_expert.py:219-227) and stays fp32. Layer norms are bf16 in all."""
        dev = torch.device(device)
        self.dev, self.dim, self.n_layers, self.Hq, self.Hkv, self.D, self.mlp = dev, dim, layers, Hq, Hkv, D, mlp
        self.act, self.norm, self.eps = act, norm, eps
        self._keep = []
        arr = (L.DecLayer * layers)()
        if share_cache_with is not None:
            self.geom = share_cache_with.geom
            self.k_cache, self.vt_cache = share_cache_with.k_cache, share_cache_with.vt_cache
        else:
            self.geom = cache
            # zero-initialised: V^T rows beyond a segment's length are read under a zero probability and must be finite
            self.k_cache = [torch.zeros(cache.elems, dtype=BF, device=dev) for _ in range(layers)]
            self.vt_cache = [torch.zeros(cache.elems, dtype=BF, device=dev) for _ in range(layers)]
        # fp8 profile: down_proj's e4m3 twin in the k-linear operand order, so that its input can carry MX block scales written by the GLU epilogue
        # of gate_up (cover_decoder_forward; COVER_FP8_MX=0 at load keeps the per-row-scale path with its quantiser launch)
        # (mlp a multiple of 128: the GLU epilogue writes whole 32-column blocks of the REAL columns only, a padded last k-tile would be read unwritten)
        mx_down = bool(fp8_weights) and os.environ.get("COVER_FP8_MX", "1") != "0" and mlp % 128 == 0
        mx_o = bool(fp8_weights) and os.environ.get("COVER_FP8_MX", "1") not in ("0", "down") and D == 128 and Hq == Hkv   # (the attention output likewise)
        for i in range(layers):
            p = f"layers.{i}."
            wqkv = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in ("q", "k", "v")], 0)
            qkv = ops.pack_linear(wqkv.to(dev), fp8=fp8_weights)
            o = ops.pack_linear(sd[p + "self_attn.o_proj.weight"].to(dev), fp8=fp8_weights, klinear=mx_o)
            gu = ops.pack_linear(torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], 0).to(dev), glu=True,
                                 fp8=fp8_weights)
            down = ops.pack_linear(sd[p + "mlp.down_proj.weight"].to(dev), fp8=fp8_weights, klinear=mx_down)
            n1 = _bf_f32(sd[p + "input_layernorm.weight"], dev)
            n2 = _bf_f32(sd[p + "post_attention_layernorm.weight"], dev)
            self._keep += [qkv, o, gu, down, n1, n2]
            a = arr[i]
            a.in_norm_w, a.post_norm_w = n1.data_ptr(), n2.data_ptr()
            a.qkv_w, a.qkv_b = qkv.wp.data_ptr(), None
            a.o_w, a.gate_up_w, a.down_w = o.wp.data_ptr(), gu.wp.data_ptr(), down.wp.data_ptr()
            a.k_cache, a.vt_cache = self.k_cache[i].data_ptr(), self.vt_cache[i].data_ptr()
            if fp8_weights:
                a.qkv_w8, a.qkv_s, a.o_w8, a.o_s = qkv.w8.data_ptr(), qkv.w8s.data_ptr(), o.w8.data_ptr(), o.w8s.data_ptr()
                a.gate_up_w8, a.gate_up_s = gu.w8.data_ptr(), gu.w8s.data_ptr()
                a.down_w8, a.down_s = down.w8.data_ptr(), down.w8s.data_ptr()
                a.down_klinear = 1 if mx_down else 0
                a.o_klinear = 1 if mx_o else 0
        self._arr = arr
        self.fp8_weights = fp8_weights
        self.final_norm = (_bf_f32 if final_norm_bf16 else _f32)(sd["norm.weight"], dev)
        cos, sin = rope_tables(rope, n_pos, D)
        self.cos, self.sin = cos.contiguous().to(dev), sin.contiguous().to(dev)
        d = L.DecDesc()
        d.dim, d.Hq, d.Hkv, d.D, d.mlp, d.n_layers = dim, Hq, Hkv, D, mlp, layers
        d.act = ops.ACT[act]
        d.norm_style = 1 if norm == "llama" else 0
        d.norm_eps, d.norm_w_offset, d.attn_scale = eps, (0.0 if norm == "llama" else 1.0), D ** -0.5
        d.rope_mode, d.n_pos = (2 if rope == "hf" else 1), n_pos
        d.cos_table, d.sin_table = self.cos.data_ptr(), self.sin.data_ptr()
        d.final_norm_w = self.final_norm.data_ptr()
        d.layers_host = C.cast(self._arr, C.POINTER(L.DecLayer))
        self.desc = d
        self._ws = None
        self.ws_gen = 0          # bumped whenever the workspace is re-allocated (see reserve)

    def reserve(self, rows: int) -> None:
        """Sizes the pass workspace for up to `rows` input rows once. A captured hipGraph holds the workspace's device pointer; a later
        pass with more rows would otherwise re-allocate it under the graphs captured for smaller passes (their replay would then write
        through a freed pointer). Owners that capture graphs call this with their largest pass and compare ws_gen before a replay."""
        need = L.lib().cover_decoder_workspace_bytes(C.byref(self.desc), rows)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
            self.ws_gen += 1

    def group(self, B, T, positions, segs, write_seg, write_slot=None, write_t_off_of_batch=None, write_t_off=0,
              seg0_shared=False, own_kv=None, seg1_group=0, seg1_slot_of_group=None, seg1_len_of_group=None, write_scratch=False):
        """segs: list of dicts {region, length, len_of_batch, slot_of_batch, mask, causal_offset, vis_len}.
        own_kv "bf16" / "fp8" (large-N candidate decode, cover_dec_group.own_kv_mode): the write segment lives in the head-major
        layout of cover_decode_own_attention; rows [i * seg1_group, (i + 1) * seg1_group) share segs[1]'s slot / length.
        write_scratch: nobody reads the write segment after this pass (pi0 denoise steps) -- the library may leave it unwritten."""
        g = L.DecGroup()
        g.B, g.T = B, T
        g.positions = positions.data_ptr()
        g.n_seg, g.write_seg = len(segs), write_seg
        keep = [positions, write_slot, write_t_off_of_batch]
        for i, s in enumerate(segs):
            r = s["region"]
            sg = g.segs[i]
            sg.k_slot_stride, sg.k_t_stride, sg.k_h_stride = self.geom.k_strides(r)
            sg.vt_slot_stride, sg.vt_h_stride, sg.vt_d_stride = self.geom.vt_strides(r)
            for name in ("slot_of_batch", "len_of_batch", "vis_len"):
                t = s.get(name)
                setattr(sg, name, t.data_ptr() if t is not None else None)
                keep.append(t)
            sg.len = s.get("length", 0)
            sg.mask_mode = s.get("mask", ops.MASK_LEN)
            sg.causal_offset = s.get("causal_offset", 0)
            g.seg_k_offset[i] = self.geom.k_off[r]
            g.seg_vt_offset[i] = self.geom.vt_off[r]
        g.write_slot_of_batch = write_slot.data_ptr() if write_slot is not None else None
        g.write_t_offset_of_batch = write_t_off_of_batch.data_ptr() if write_t_off_of_batch is not None else None
        g.write_t_offset = write_t_off
        g.seg0_shared = 1 if seg0_shared else 0
        g.write_scratch = 1 if write_scratch else 0
        if own_kv is not None:
            r = segs[write_seg]["region"]
            g.own_kv_mode = {"bf16": 1, "fp8": 2}[own_kv]
            g.seg1_group = seg1_group
            g.seg1_slot_of_group = seg1_slot_of_group.data_ptr() if seg1_slot_of_group is not None else None
            g.seg1_len_of_group = seg1_len_of_group.data_ptr() if seg1_len_of_group is not None else None
            g.own_region_elems = self.geom.slots[r] * self.geom.caps[r] * self.geom.Hkv * self.geom.D
            keep += [seg1_slot_of_group, seg1_len_of_group]
        g._keep = keep
        return g

    def workspace(self, rows: int) -> torch.Tensor:
        """A pass workspace of its own for up to `rows` rows: independent passes of one decoder that run CONCURRENTLY on different streams
        (row-group chains of the pi0 denoise loop) must not share the decoder's."""
        return torch.empty(L.lib().cover_decoder_workspace_bytes(C.byref(self.desc), rows), dtype=torch.uint8, device=self.dev)

    def forward(self, x: torch.Tensor, groups, final_norm=False, x_f32: Optional[torch.Tensor] = None, gemm_variant=0, ws: Optional[torch.Tensor] = None):
        """x bf16 [rows, dim] (overwritten with the output hidden states). ws: a workspace from workspace() instead of the decoder's own."""
        p = L.DecPass()
        p.n_groups, p.final_norm = len(groups), 1 if final_norm else 0
        p.x_f32 = x_f32.data_ptr() if x_f32 is not None else None
        for i, g in enumerate(groups):
            p.groups[i] = g
        rows = x.shape[0]
        h = L.lib()
        need = h.cover_decoder_workspace_bytes(C.byref(self.desc), rows)
        if ws is not None:
            if ws.numel() < need:
                raise ValueError("Decoder.forward: the caller's workspace is too small for this pass")
            ws = L.Workspace(ws.data_ptr(), ws.numel())
        else:
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
                self.ws_gen += 1
            ws = L.Workspace(self._ws.data_ptr(), self._ws.numel())
        L.check(h.cover_decoder_forward(C.byref(self.desc), C.byref(p), x.data_ptr(), ws, gemm_variant,
                                        torch.cuda.current_stream().cuda_stream), "decoder_forward")
        return x


def rope_tables(kind: str, n_pos: int, D: int):
    """cos/sin fp32 [n_pos, D/2] built with the reference's own expressions (host, once)."""
    if kind == "pi0":  # apply_rope, paligemma_with_expert.py:43-49
        d_half = D // 2
        freq_exponents = (2.0 / D) * torch.arange(d_half, dtype=torch.float32)
        timescale = 10_000 ** freq_exponents
        radians = torch.arange(n_pos)[:, None].to(torch.float32) / timescale[None, :].to(torch.float32)
        return torch.cos(radians), torch.sin(radians)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))  # HF LlamaRotaryEmbedding
    freqs = torch.arange(n_pos).float()[:, None] * inv_freq[None, :]
    return freqs.cos(), freqs.sin()
