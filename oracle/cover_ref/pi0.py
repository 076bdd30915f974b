"""CPU restatement of the pi0 sampler at the tensor boundary PI0FlowMatching.sample_actions
(lerobot_custom/lerobot/common/policies/pi0/modeling_pi0.py:672-715).

Follows
  modeling_pi0.py:71-89     create_sinusoidal_pos_embedding (float64)
  modeling_pi0.py:98-128    make_att_2d_masks
  modeling_pi0.py:517-567   embed_prefix (image tokens x bf16(sqrt(D)) after HF-4.48.3's / sqrt(D); token embeddings x sqrt(D))
  modeling_pi0.py:569-629   embed_suffix (fp32 projections, bf16 state token and time embedding, SiLU MLP)
  modeling_pi0.py:717-752   denoise_step ; :697-715 Euler loop (dt = -1/num_steps, while time >= -dt/2)
  paligemma_with_expert.py:236-434 via cover_ref.blocks.decoder_forward
Semantics of record for the un-vendored HF pieces = transformers 4.48.3 (requirements.txt:25), see SURVEY.md §8c.
Neutral state dict: vision.* (synth.vit_state), projector.{weight,bias}, lm.* / expert.* (synth.decoder_state),
state_proj / action_in_proj / action_out_proj / action_time_mlp_in / action_time_mlp_out .{weight,bias} (fp32).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import blocks as Bk

BF = torch.bfloat16


class Pi0Cfg:
    def __init__(self, vit: Bk.VitCfg, lm: Bk.DecoderCfg, expert: Bk.DecoderCfg, proj_width=1024, max_state_dim=32,
                 max_action_dim=32, chunk_size=4, num_steps=10, n_img_tokens=256):
        self.vit, self.lm, self.expert = vit, lm, expert
        self.proj_width, self.max_state_dim, self.max_action_dim = proj_width, max_state_dim, max_action_dim
        self.chunk_size, self.num_steps, self.n_img_tokens = chunk_size, num_steps, n_img_tokens


def sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def create_sinusoidal_pos_embedding(time, dimension, min_period, max_period):
    fraction = torch.linspace(0.0, 1.0, dimension // 2, dtype=torch.float64)
    period = min_period * (max_period / min_period) ** fraction
    scaling_factor = 1.0 / period * 2 * math.pi
    sin_input = scaling_factor[None, :] * time[:, None]
    return torch.cat([torch.sin(sin_input), torch.cos(sin_input)], dim=1)


def make_att_2d_masks(pad_masks, att_masks):
    cumsum = torch.cumsum(att_masks, dim=1)
    att_2d = cumsum[:, None, :] <= cumsum[:, :, None]
    pad_2d = pad_masks[:, None, :] * pad_masks[:, :, None]
    return att_2d & pad_2d


def embed_image(cfg: Pi0Cfg, sd, pixels):
    """HF-4.48.3 PaliGemma.get_image_features: SigLIP tower (post-LN) -> projector -> / sqrt(hidden)."""
    vs = sub(sd, "vision.")
    x = Bk.vit_embed(cfg.vit, vs, pixels)
    x = Bk.vit_encode(cfg.vit, vs, x, post_ln=True)
    x = F.linear(x, sd["projector.weight"], sd["projector.bias"])
    return x / (cfg.lm.dim ** 0.5)


def embed_prefix(cfg, sd, images, img_masks, lang_tokens, lang_masks):
    embs, pads, att = [], [], []
    for img, m in zip(images, img_masks):
        e = embed_image(cfg, sd, img).to(BF)
        e = e * torch.tensor(e.shape[-1] ** 0.5, dtype=e.dtype)
        embs.append(e)
        pads.append(m[:, None].expand(e.shape[0], e.shape[1]))
        att += [0] * e.shape[1]
    le = F.embedding(lang_tokens, sd["lm.embed_tokens.weight"])
    le = le * math.sqrt(le.shape[-1])
    embs.append(le)
    pads.append(lang_masks)
    att += [0] * le.shape[1]
    embs = torch.cat(embs, 1)
    pads = torch.cat(pads, 1)
    att = torch.tensor(att, dtype=torch.bool)[None].expand(embs.shape[0], -1)
    return embs, pads, att


def embed_suffix(cfg, sd, state, noisy_actions, timestep):
    state_emb = F.linear(state, sd["state_proj.weight"], sd["state_proj.bias"]).to(BF)
    B = state_emb.shape[0]
    time_emb = create_sinusoidal_pos_embedding(timestep, cfg.proj_width, 4e-3, 4.0).type(BF)
    action_emb = F.linear(noisy_actions, sd["action_in_proj.weight"], sd["action_in_proj.bias"])
    time_emb = time_emb[:, None, :].expand_as(action_emb)
    at = torch.cat([action_emb, time_emb], dim=2)  # fp32 + bf16 -> fp32
    at = F.linear(at, sd["action_time_mlp_in.weight"], sd["action_time_mlp_in.bias"])
    at = F.silu(at)
    at = F.linear(at, sd["action_time_mlp_out.weight"], sd["action_time_mlp_out.bias"])
    embs = torch.cat([state_emb[:, None, :], at], dim=1)  # bf16 + fp32 -> fp32
    pads = torch.ones(B, 1 + at.shape[1], dtype=torch.bool)
    att = torch.tensor([1, 1] + [0] * (cfg.chunk_size - 1), dtype=embs.dtype)[None].expand(B, -1)
    return embs, pads, att


def sample_actions(cfg: Pi0Cfg, sd, images, img_masks, lang_tokens, lang_masks, state, noise, trace=None,
                   prefix_embs=None):
    """Returns x_t [B, chunk, max_action_dim] fp32. `trace` (dict) optionally receives intermediates.
    prefix_embs: optional override of the embedded prefix (isolates the decoder from the vision tower in tests)."""
    B = state.shape[0]
    pe, ppad, patt = embed_prefix(cfg, sd, images, img_masks, lang_tokens, lang_masks)
    if prefix_embs is not None:
        pe = prefix_embs
    pmask = make_att_2d_masks(ppad, patt)
    ppos = torch.cumsum(ppad, dim=1) - 1
    _, kv = Bk.decoder_forward(cfg.lm, sub(sd, "lm."), pe, ppos.clamp(min=0), pmask, past=None, keep_kv=True, final_norm=False)
    if trace is not None:
        trace["prefix_embs"] = pe
        trace["kv"] = kv
    dt = torch.tensor(-1.0 / cfg.num_steps, dtype=torch.float32)
    x_t = noise.clone()
    time = torch.tensor(1.0, dtype=torch.float32)
    vs = []
    while time >= -dt / 2:
        se, spad, satt = embed_suffix(cfg, sd, state, x_t, time.expand(B))
        S, P = spad.shape[1], ppad.shape[1]
        prefix_2d = ppad[:, None, :].expand(B, S, P)
        suffix_2d = make_att_2d_masks(spad, satt)
        full = torch.cat([prefix_2d, suffix_2d], dim=2)
        pos = torch.sum(ppad, dim=-1)[:, None] + torch.cumsum(spad, dim=1) - 1
        out, _ = Bk.decoder_forward(cfg.expert, sub(sd, "expert."), se, pos, full, past=kv, keep_kv=False, final_norm=True)
        out = out[:, -cfg.chunk_size:].to(torch.float32)
        v_t = F.linear(out, sd["action_out_proj.weight"], sd["action_out_proj.bias"])
        vs.append(v_t)
        x_t = x_t + dt * v_t
        time = time + dt
    if trace is not None:
        trace["v_t"] = vs
    return x_t


def cast_like_reference(sd):
    """to_bfloat16_like_physical_intelligence (paligemma_with_expert.py:216-227): PaliGemma (vision tower, projector,
    language model incl. embeddings and norms) and the expert's layers in bf16; pi0's own projections stay fp32.
    NB the expert's final norm weight is NOT under 'gemma_expert.model.layers' and stays fp32."""
    out = {}
    for k, v in sd.items():
        if k.startswith(("vision.", "projector.", "lm.")) or k.startswith("expert.layers."):
            out[k] = v.to(BF)
        else:
            out[k] = v.float()
    return out
