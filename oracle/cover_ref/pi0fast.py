"""CPU restatement of the pi0-FAST token path (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline may import this package).

Follows lerobot_custom/lerobot/common/policies/pi0fast/modeling_pi0fast.py:
  :888-946  embed_inputs        image tokens (PaliGemma get_image_features: SigLIP tower -> projector -> / sqrt(hidden), HF 4.48.3)
                                 then the token embeddings; pad / block masks concatenated in that order
  :236-330  block_causal_update_causal_mask   prefix (token type 0: image + prompt tokens) bidirectional, suffix causal, padded
                                 keys masked
  :861-884  generate_actions    greedy `generate` on the left-padded batch; :352-354 positions are 1-indexed
  GemmaModel.forward (4.48.3)   inputs_embeds * sqrt(hidden) in the embedding dtype, HF rotary embedding, tied lm_head
  :735-792  decode_actions_with_fast   token ids -> DCT coefficients (BPE decoder = the un-vendored FAST processor, a callable
                                 here) -> relaxed pad / truncate -> idct(coeff / scale, axis=0, norm="ortho")
The padding side does not enter the arithmetic (positions come from the cumulative pad mask, padded keys are masked): rows are
kept right-padded here and the prefix of each row is compacted, which is what the HIP path does too.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import blocks as Bk
from .pi0 import sub


def embed_image(vit: "Bk.VitCfg", sd, pixels, hidden):
    vs = sub(sd, "vision.")
    x = Bk.vit_embed(vit, vs, pixels)
    x = Bk.vit_encode(vit, vs, x, post_ln=True)
    x = F.linear(x, sd["projector.weight"], sd["projector.bias"])
    return x / (hidden ** 0.5)


def embed_inputs(vit, lm: "Bk.DecoderCfg", sd, image, tokens, pad):
    """image [B,3,H,W]; tokens int64 [B,L] right padded, pad [B,L] 0/1 -> (embs [B, n_img + L, dim] in the weight dtype, pad mask)."""
    w = sd["lm.embed_tokens.weight"]
    img = embed_image(vit, sd, image.to(w.dtype), lm.dim).to(w.dtype)
    te = F.embedding(tokens, w)
    embs = torch.cat([img, te], dim=1)
    pm = torch.cat([torch.ones(img.shape[0], img.shape[1], dtype=pad.dtype), pad], dim=1)
    return embs, pm


def forward_logits(lm, sd, embs, pad_mask, token_type):
    """One PaliGemma forward as PI0FAST drives it: embs [B,T,dim], pad_mask / token_type [B,T] -> logits fp32 [B,T,V]."""
    cs = torch.cumsum(token_type.to(torch.int64), dim=1)
    mask = (cs[:, None, :] <= cs[:, :, None]) & pad_mask[:, None, :].bool()
    pos = torch.cumsum(pad_mask.to(torch.int64), dim=1).clamp(min=1)          # 1-indexed; padded rows are never read
    x = embs * torch.tensor(lm.dim ** 0.5, dtype=embs.dtype)
    h, _ = Bk.decoder_forward(lm, sub(sd, "lm."), x, pos, mask, past=None, keep_kv=False, final_norm=True, n_pos=int(pos.max()) + 2)
    return F.linear(h.to(sd["lm.embed_tokens.weight"].dtype), sd["lm.embed_tokens.weight"]).float()


def generate(vit, lm, sd, image, tokens, pad, n_new, eos=1, pad_id=0, force=None):
    """Greedy generation, no cache (every step re-runs the sequence): returns (tokens int64 [B,n_new], logits fp32 [n_new,B,V])."""
    B, L = tokens.shape
    lens = pad.sum(1)
    gen = torch.zeros(B, 0, dtype=torch.long)
    done = torch.zeros(B, dtype=torch.bool)
    out = []
    pe, pm = embed_inputs(vit, lm, sd, image, tokens, pad)
    n_img = pe.shape[1] - L
    for step in range(n_new):
        T = n_img + L + step
        embs = torch.zeros(B, T, pe.shape[2], dtype=pe.dtype)
        pmask = torch.zeros(B, T, dtype=torch.long)
        ttype = torch.zeros(B, T, dtype=torch.long)
        last = torch.zeros(B, dtype=torch.long)
        ge = F.embedding(gen, sd["lm.embed_tokens.weight"])
        for b in range(B):                       # compact: [image | valid prompt tokens | generated | padding]
            n = n_img + int(lens[b])
            embs[b, :n] = torch.cat([pe[b, :n_img], pe[b, n_img:n_img + int(lens[b])]], 0)
            embs[b, n:n + step] = ge[b]
            pmask[b, :n + step] = 1
            ttype[b, n:n + step] = 1
            last[b] = n + step - 1
        lg = forward_logits(lm, sd, embs, pmask, ttype)[torch.arange(B), last]
        nxt = lg.argmax(-1) if force is None else force[:, step]
        nxt = torch.where(done, torch.full_like(nxt, pad_id), nxt)
        out.append(lg)
        gen = torch.cat([gen, nxt[:, None]], dim=1)
        done = done | (nxt == eos)
    return gen, torch.stack(out)


def decode_actions_with_fast(token_lists, bpe_decode, min_token, scale, time_horizon, action_dim, relaxed=True):
    """:735-792. bpe_decode(list[int]) -> str is the FAST processor's BPE decoder (not vendored: injected)."""
    from scipy.fft import idct
    outs = []
    for toks in token_lists:
        coeff = np.array(list(map(ord, bpe_decode(toks)))) + min_token
        if relaxed:
            want = time_horizon * action_dim
            diff = want - coeff.shape[0]
            if diff < 0:
                coeff = coeff[:want]
            elif diff > 0:
                coeff = np.pad(coeff, (0, diff), mode="constant", constant_values=0)
        coeff = coeff.reshape(-1, action_dim)
        outs.append(idct(coeff / scale, axis=0, norm="ortho"))
    return np.stack(outs)
