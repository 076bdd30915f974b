"""Transformer building blocks of the oracle, in eager-PyTorch bf16 semantics (every op rounds where eager
PyTorch on bf16 tensors rounds). Shared by the pi0 (P1) and OpenVLA-7B (P2) restatements.

Follows
  paligemma_with_expert.py:34-57     apply_rope (fp32, half-split, one rounding)
  paligemma_with_expert.py:258-360   decoder layer loop (RMSNorm -> qkv -> RoPE -> attention -> o_proj + residual ->
                                     RMSNorm -> gated MLP + residual), final norm
  paligemma_with_expert.py:376-434   eager_attention_forward (fp32 QK^T, scale after, big_neg mask, fp32 softmax,
                                     probabilities -> bf16, PV in bf16)
  HF transformers (un-vendored; 4.48.3 pinned by the reference): GemmaRMSNorm x*rsqrt(mean x^2+eps)*(1+w) in fp32;
  LlamaRMSNorm w*bf16(x*rsqrt(..)); GemmaMLP / LlamaMLP down(act(gate x) * up x); Siglip/timm pre-LN ViT block.
Weights: neutral state dicts produced by cover_vla_amd/synth.py (HF key names for decoders).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

BF = torch.bfloat16
BIG_NEG = -2.3819763e38


def lin(x, w, b=None):
    return F.linear(x, w, b)


def act_fn(name):
    return {"gelu_tanh": lambda x: F.gelu(x, approximate="tanh"), "gelu_erf": F.gelu, "silu": F.silu}[name]


# ------------------------------------------------------------------------------------------------ norms / rope
def gemma_rmsnorm(x, w, eps=1e-6):
    xf = x.float()
    out = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    out = out * (1.0 + w.float())
    return out.type_as(x)


def llama_rmsnorm(x, w, eps=1e-5):
    dt = x.dtype
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(dt)


def rope_tables_pi0(n_pos, D, max_wavelength=10_000):
    """cos/sin [n_pos, D/2] with the reference's expressions (paligemma_with_expert.py:43-49)."""
    d_half = D // 2
    freq_exponents = (2.0 / D) * torch.arange(d_half, dtype=torch.float32)
    timescale = max_wavelength ** freq_exponents
    radians = torch.arange(n_pos)[:, None].to(torch.float32) / timescale[None, :].to(torch.float32)
    return torch.cos(radians), torch.sin(radians)


def rope_tables_hf(n_pos, D, base=10000.0):
    """HF LlamaRotaryEmbedding: inv_freq = 1/base^(2i/D), freqs = pos * inv_freq (fp32)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))
    freqs = torch.arange(n_pos).float()[:, None] * inv_freq[None, :]
    return freqs.cos(), freqs.sin()


def apply_rope_pi0(x, positions, cos_t, sin_t):
    """x [B,L,H,D] -> fp32 rotate -> x.dtype (one rounding)."""
    dt = x.dtype
    x = x.float()
    half = x.shape[-1] // 2
    cos = cos_t[positions][:, :, None, :]
    sin = sin_t[positions][:, :, None, :]
    x1, x2 = x[..., :half], x[..., half:]
    return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], -1).to(dt)


def apply_rope_hf(x, positions, cos_t, sin_t):
    """HF apply_rotary_pos_emb in the tensor dtype: cos/sin cast to bf16, q*cos + rotate_half(q)*sin with bf16 ops."""
    dt = x.dtype
    cos = torch.cat([cos_t, cos_t], -1)[positions][:, :, None, :].to(dt)
    sin = torch.cat([sin_t, sin_t], -1)[positions][:, :, None, :].to(dt)
    half = x.shape[-1] // 2
    rot = torch.cat([-x[..., half:], x[..., :half]], -1)
    return (x * cos) + (rot * sin)


# ------------------------------------------------------------------------------------------------ attention
def eager_attention(q, k, v, mask, scale, scores_bf16=False):
    """q [B,Tq,Hq,D], k/v [B,Tk,Hkv,D] (bf16), mask bool [B,Tq,Tk] -> [B,Tq,Hq*D] bf16.
    paligemma_with_expert.py:376-434 semantics. scores_bf16: HF transformers' eager_attention_forward of the Llama family instead
    (models/llama/modeling_llama.py: QK^T and the scaling in the TENSOR dtype -- bf16 scores --, additive mask, softmax in fp32, cast
    back, PV in bf16): used to pin the restatement to HF's bf16 outputs; the default keeps the reference's fp32 scores."""
    B, Tq, Hq, D = q.shape
    G = Hq // k.shape[2]
    k = k.repeat_interleave(G, dim=2)
    v = v.repeat_interleave(G, dim=2)
    if scores_bf16:
        att = torch.matmul(q.transpose(1, 2), k.transpose(1, 2).transpose(2, 3)) * scale
        att = att + torch.where(mask[:, None, :, :], 0.0, torch.finfo(att.dtype).min).to(att.dtype)
        probs = torch.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
        out = torch.matmul(probs, v.permute(0, 2, 1, 3))
        return out.permute(0, 2, 1, 3).reshape(B, Tq, Hq * D)
    qf = q.float().transpose(1, 2)
    kf = k.float().transpose(1, 2)
    att = torch.matmul(qf, kf.transpose(2, 3))
    att = att * scale
    att = torch.where(mask[:, None, :, :], att, torch.tensor(BIG_NEG))
    probs = torch.softmax(att, dim=-1).to(v.dtype)
    out = torch.matmul(probs, v.permute(0, 2, 1, 3))
    return out.permute(0, 2, 1, 3).reshape(B, Tq, Hq * D)


# ------------------------------------------------------------------------------------------------ decoder
class DecoderCfg:
    def __init__(self, dim, layers, Hq, Hkv, D, mlp, act, norm, eps, rope, scores_bf16=False):
        self.dim, self.layers, self.Hq, self.Hkv, self.D, self.mlp = dim, layers, Hq, Hkv, D, mlp
        self.act, self.norm, self.eps, self.rope = act, norm, eps, rope  # norm: "gemma"|"llama"; rope: "pi0"|"hf"
        self.scores_bf16 = scores_bf16   # HF Llama eager attention (bf16 scores) instead of the reference's fp32 scores

    def rms(self, x, w):
        return gemma_rmsnorm(x, w, self.eps) if self.norm == "gemma" else llama_rmsnorm(x, w, self.eps)

    def tables(self, n_pos):
        return rope_tables_pi0(n_pos, self.D) if self.rope == "pi0" else rope_tables_hf(n_pos, self.D)

    def apply_rope(self, x, pos, tabs):
        return apply_rope_pi0(x, pos, *tabs) if self.rope == "pi0" else apply_rope_hf(x, pos, *tabs)


def fake_quant_rows_e4m3(x):
    """What cover_quantize_act_fp8 + the fp8 MFMA see of a bf16 activation tensor [..., K]: per-row power-of-two scale (smallest
    2^e with amax / 2^e <= 448), RNE to OCP e4m3, de-quantised (exactly representable in bf16). Config 5 only (no reference
    arithmetic: SURVEY.md 7 step 9)."""
    xf = x.float()
    amax = xf.abs().amax(dim=-1, keepdim=True)
    s = torch.where(amax > 0, torch.pow(2.0, torch.ceil(torch.log2(amax.double() / 448.0))).float(), torch.ones_like(amax))
    return ((xf / s).to(torch.float8_e4m3fn).float() * s).to(x.dtype)


def fake_quant_blocks_e4m3(x):
    """What cover_quantize_act_fp8_mx (or the GLU epilogue that writes the same bytes) + the block-scaled fp8 MFMA see of a bf16 activation tensor
    [..., K], K a multiple of 32: one power-of-two scale per 32 consecutive elements (smallest 2^e, e >= -126, with amax_block / 2^e <= 448), RNE to OCP
    e4m3, de-quantised. Config 5's down_proj input (no reference arithmetic: SURVEY.md 7 step 9)."""
    xf = x.float()
    blk = xf.reshape(*xf.shape[:-1], xf.shape[-1] // 32, 32)
    amax = blk.abs().amax(dim=-1, keepdim=True)
    e = torch.where(amax > 0, torch.ceil(torch.log2(amax.double() / 448.0)), torch.zeros_like(amax, dtype=torch.float64)).clamp(min=-126)
    s = torch.pow(2.0, e).float()
    return ((blk / s).to(torch.float8_e4m3fn).float() * s).reshape(xf.shape).to(x.dtype)


def decoder_forward(cfg: DecoderCfg, sd, x, positions, mask, past=None, keep_kv=True, final_norm=True, n_pos=4096, act_fp8=False,
                    kv_fp8=False, act_mx_down=False, act_mx_o=False):
    """x [B,T,dim] (bf16, or fp32 for the pi0 suffix at layer 0). past: list of (K,V) [B,Tp,Hkv,D] per layer (post-RoPE) or
    None. mask bool [B,T,Tp+T]. Returns (hidden [B,T,dim], new list of (K,V) including this pass's tokens if keep_kv).
    act_fp8: the input rows of the four projections are e4m3-quantised per row (the fp8 MFMA profile, config 5).
    kv_fp8: this pass's K (after RoPE) and V rows are e4m3-quantised per (token, head) row before they enter the cache (the fp8
    own-token KV cache of config 5). act_mx_down / act_mx_o (with act_fp8): the down_proj / o_proj input carries MX block scales instead of one scale per row."""
    fq = fake_quant_rows_e4m3 if act_fp8 else (lambda t: t)
    fq_down = fake_quant_blocks_e4m3 if (act_fp8 and act_mx_down) else fq
    fq_o = fake_quant_blocks_e4m3 if (act_fp8 and act_mx_o) else fq
    tabs = cfg.tables(n_pos)
    B, T, _ = x.shape
    new_kv = []
    act = act_fn(cfg.act)
    for l in range(cfg.layers):
        p = f"layers.{l}."
        h = fq(cfg.rms(x, sd[p + "input_layernorm.weight"]).to(BF))
        q = lin(h, sd[p + "self_attn.q_proj.weight"]).view(B, T, cfg.Hq, cfg.D)
        k = lin(h, sd[p + "self_attn.k_proj.weight"]).view(B, T, cfg.Hkv, cfg.D)
        v = lin(h, sd[p + "self_attn.v_proj.weight"]).view(B, T, cfg.Hkv, cfg.D)
        q = cfg.apply_rope(q, positions, tabs)
        k = cfg.apply_rope(k, positions, tabs)
        if kv_fp8:
            k, v = fake_quant_rows_e4m3(k), fake_quant_rows_e4m3(v)
        if past is not None:
            kk = torch.cat([past[l][0], k], 1)
            vv = torch.cat([past[l][1], v], 1)
        else:
            kk, vv = k, v
        if keep_kv:
            new_kv.append((kk, vv))
        a = fq_o(eager_attention(q, kk, vv, mask, cfg.D ** -0.5, scores_bf16=cfg.scores_bf16).to(BF))
        o = lin(a, sd[p + "self_attn.o_proj.weight"])
        o += x  # in-place add into the bf16 o_proj output (paligemma_with_expert.py:332): fp32 x is rounded here
        res = o.clone()
        h = fq(cfg.rms(o, sd[p + "post_attention_layernorm.weight"]))
        h = lin(fq_down(act(lin(h, sd[p + "mlp.gate_proj.weight"])) * lin(h, sd[p + "mlp.up_proj.weight"])), sd[p + "mlp.down_proj.weight"])
        h += res
        x = h
    if final_norm:
        x = cfg.rms(x, sd["norm.weight"])
    return x, new_kv


# ------------------------------------------------------------------------------------------------ ViT
class VitCfg:
    def __init__(self, dim, layers, heads, mlp, patch, act, eps=1e-6, layerscale=False, prefix_tokens=0):
        self.dim, self.layers, self.heads, self.mlp, self.patch = dim, layers, heads, mlp, patch
        self.act, self.eps, self.layerscale, self.prefix_tokens = act, eps, layerscale, prefix_tokens


def vit_embed(cfg: VitCfg, sd, pixels):
    """pixels fp32 [B,3,H,W] (already normalised) -> bf16 tokens [B, prefix+P, dim] (conv patch embed + bias + pos)."""
    B = pixels.shape[0]
    w = sd["patch.weight"].view(cfg.dim, 3, cfg.patch, cfg.patch)
    x = F.conv2d(pixels.to(w.dtype), w, sd["patch.bias"], stride=cfg.patch)   # bf16 towers: pixels cast to bf16 (eager semantics)
    x = x.flatten(2).transpose(1, 2)
    if cfg.prefix_tokens:
        x = torch.cat([sd["prefix"][None].expand(B, -1, -1), x], 1)
    return x + sd["pos"][None, : x.shape[1]]


def vit_block(cfg: VitCfg, sd, p, x, attn_only=False):
    B, T, C = x.shape
    H = cfg.heads
    Dh = C // H
    h = F.layer_norm(x, (C,), sd[p + "ln1.weight"], sd[p + "ln1.bias"], cfg.eps)
    q = lin(h, sd[p + "q.weight"], sd[p + "q.bias"]).view(B, T, H, Dh)
    k = lin(h, sd[p + "k.weight"], sd[p + "k.bias"]).view(B, T, H, Dh)
    v = lin(h, sd[p + "v.weight"], sd[p + "v.bias"]).view(B, T, H, Dh)
    mask = torch.ones(B, T, T, dtype=torch.bool)
    a = eager_attention(q, k, v, mask, Dh ** -0.5).to(x.dtype)
    a = lin(a, sd[p + "o.weight"], sd[p + "o.bias"])
    if attn_only:
        return a
    if cfg.layerscale:
        a = a * sd[p + "ls1"]
    x = x + a
    h = F.layer_norm(x, (C,), sd[p + "ln2.weight"], sd[p + "ln2.bias"], cfg.eps)
    h = lin(act_fn(cfg.act)(lin(h, sd[p + "fc1.weight"], sd[p + "fc1.bias"])), sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    if cfg.layerscale:
        h = h * sd[p + "ls2"]
    return x + h


def vit_encode(cfg: VitCfg, sd, x, n_blocks=None, last_attn_only=False, post_ln=False):
    """x bf16 [B,T,dim] tokens -> after n_blocks blocks (default all). last_attn_only: the last of them returns its
    attention-module output (incl. out-proj, pre-residual): the hook feature of finetune_trajectory_bridge_ddp.py:272-274."""
    n = cfg.layers if n_blocks is None else n_blocks
    for i in range(n):
        x = vit_block(cfg, sd, f"blocks.{i}.", x, attn_only=(last_attn_only and i == n - 1))
    if post_ln:
        x = F.layer_norm(x, (cfg.dim,), sd["post_ln.weight"], sd["post_ln.bias"], cfg.eps)
    return x


def to_bf16(sd):
    return {k: (v.to(BF) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in sd.items()}
