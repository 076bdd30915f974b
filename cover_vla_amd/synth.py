"""Deterministic synthetic checkpoints in the reference's own on-disk key layouts.

There is no network and no real checkpoint offline, so parity tests and bench.py run on seeded random weights
(SURVEY.md §8d). The dictionaries produced here use the SAME keys the reference loads:
  * verifier: merged checkpoint {"ensemble_components": [ {text_aware_visual_extraction, vision_poolings,
    text_pooling, input_projection, single_step_action_encoder, trajectory_encoder, action_padding_value} ]}
    (bridge_verifier/ensemble_eval/efficient_ensemble_merged.py:37-53,94-184)
  * pi0: model.safetensors keys without the leading "model." (modeling_pi0.py:486-494; HF PaliGemma / Gemma names)
so that the golden-vector generator can `load_state_dict` them straight into the reference's modules.
All tensors are CPU fp32 unless stated; `nontrivial=True` also randomises biases and norm weights so every
epilogue path is numerically exercised.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

Tensor = torch.Tensor


class _G:
    """Seeded tensor source. device="cpu" (tests, goldens: bit-reproducible everywhere) or a cuda device with
    wdtype=bf16 (bench: a 7B-parameter synthetic checkpoint is drawn directly in HBM in about a second)."""

    def __init__(self, seed: int, nontrivial: bool = True, std: float = 0.02, device="cpu", wdtype=torch.float32):
        self.device = torch.device(device)
        self.g = torch.Generator(device=self.device).manual_seed(seed)
        self.nontrivial = nontrivial
        self.std = std
        self.wdtype = wdtype

    def w(self, *shape, std=None) -> Tensor:
        t = torch.randn(*shape, generator=self.g, device=self.device, dtype=self.wdtype)
        return t * (self.std if std is None else std)

    def b(self, n) -> Tensor:
        return torch.randn(n, generator=self.g, device=self.device) * 0.02 if self.nontrivial else torch.zeros(n, device=self.device)

    def ln_w(self, n) -> Tensor:
        return 1.0 + torch.randn(n, generator=self.g, device=self.device) * 0.1 if self.nontrivial else torch.ones(n, device=self.device)

    def rms_w(self, n, base=0.0) -> Tensor:
        if self.nontrivial:
            return base + torch.randn(n, generator=self.g, device=self.device) * 0.1
        return torch.full((n,), base, device=self.device)


def sincos_position_embedding(seq_len: int, dim: int) -> Tensor:
    """Same expression as bridge_verifier/ensemble_eval/model.py:40-47 (a registered buffer of the checkpoint)."""
    pos = torch.arange(seq_len).float()
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
    s = torch.einsum("i,j->ij", pos, inv_freq)
    return torch.cat((s.sin(), s.cos()), dim=-1)


# ------------------------------------------------------------------------------------------------ verifier
def _pooling_sd(g: _G, input_dim: int, dim: int, layers: int, std: float) -> Dict[str, Tensor]:
    sd = {"query": torch.randn(1, 1, dim, generator=g.g), "layer_norm.weight": g.ln_w(dim), "layer_norm.bias": g.b(dim)}
    for i in range(layers):
        p = f"blocks.{i}."
        sd[p + "attention.q_proj_weight"] = g.w(dim, dim, std=std)
        sd[p + "attention.k_proj_weight"] = g.w(dim, input_dim, std=std)
        sd[p + "attention.v_proj_weight"] = g.w(dim, input_dim, std=std)
        sd[p + "attention.in_proj_bias"] = g.b(3 * dim)
        sd[p + "attention.out_proj.weight"] = g.w(dim, dim, std=std)
        sd[p + "attention.out_proj.bias"] = g.b(dim)
        sd[p + "mlp.fc1.weight"] = g.w(dim, dim, std=std)
        sd[p + "mlp.fc1.bias"] = g.b(dim)
        sd[p + "mlp.fc2.weight"] = g.w(dim, dim, std=std)
        sd[p + "mlp.fc2.bias"] = g.b(dim)
        sd[p + "q_layer_norm.weight"] = g.ln_w(dim)
        sd[p + "q_layer_norm.bias"] = g.b(dim)
        sd[p + "layer_norm.weight"] = g.ln_w(dim)
        sd[p + "layer_norm.bias"] = g.b(dim)
    return sd


def _traj_sd(g: _G, d: int, ff: int, layers: int, std: float) -> Dict[str, Tensor]:
    sd = {}
    for i in range(layers):
        p = f"layers.{i}."
        sd[p + "self_attn.in_proj_weight"] = g.w(3 * d, d, std=std)
        sd[p + "self_attn.in_proj_bias"] = g.b(3 * d)
        sd[p + "self_attn.out_proj.weight"] = g.w(d, d, std=std)
        sd[p + "self_attn.out_proj.bias"] = g.b(d)
        sd[p + "linear1.weight"] = g.w(ff, d, std=std)
        sd[p + "linear1.bias"] = g.b(ff)
        sd[p + "linear2.weight"] = g.w(d, ff, std=std)
        sd[p + "linear2.bias"] = g.b(d)
        sd[p + "norm1.weight"] = g.ln_w(d)
        sd[p + "norm1.bias"] = g.b(d)
        sd[p + "norm2.weight"] = g.ln_w(d)
        sd[p + "norm2.bias"] = g.b(d)
    return sd


def verifier_checkpoint(n_members: int = 3, seed: int = 1234, num_patches: int = 576, vision_dim: int = 1024,
                        text_dim: int = 1024, dim: int = 512, action_dim: int = 7, pooling_layers: int = 4,
                        traj_layers: int = 4, nontrivial: bool = True, std: float = 0.05, use_transformer: bool = True,
                        history_length: int = 10) -> dict:
    """Merged verifier checkpoint (weights-only flavour: only `ensemble_components`). use_transformer=False gives the MLP
    action-encoder variant (`complex_action_encoder` = Sequential(Linear, LayerNorm, ReLU, Dropout, Linear) -> keys 0.*, 1.*,
    4.*; efficient_ensemble_merged.py:148-184) together with the top-level metadata keys of the full-format checkpoint
    (:41-47), which is the only way the reference learns `use_transformer`."""
    comps = []
    for m in range(n_members):
        g = _G(seed + 1000 * m, nontrivial)
        if not use_transformer:
            comps.append({
                "text_aware_visual_extraction": {"temperature": torch.tensor(0.07),
                                                 "pos_emb": sincos_position_embedding(num_patches, vision_dim)},
                "vision_poolings": _pooling_sd(g, vision_dim, dim, pooling_layers, std),
                "text_pooling": _pooling_sd(g, text_dim, dim, pooling_layers, std),
                "input_projection": {"weight": g.w(dim, 2 * dim, std=std), "bias": g.b(dim)},
                "single_step_action_encoder": None, "trajectory_encoder": None,
                "complex_action_encoder": {"0.weight": g.w(dim, history_length * action_dim, std=0.2), "0.bias": g.b(dim),
                                           "1.weight": g.ln_w(dim), "1.bias": g.b(dim),
                                           "4.weight": g.w(dim, dim, std=std), "4.bias": g.b(dim)},
                "action_padding_value": -5.0,
            })
            continue
        comps.append({
            "text_aware_visual_extraction": {"temperature": torch.tensor(0.07),
                                             "pos_emb": sincos_position_embedding(num_patches, vision_dim)},
            "vision_poolings": _pooling_sd(g, vision_dim, dim, pooling_layers, std),
            "text_pooling": _pooling_sd(g, text_dim, dim, pooling_layers, std),
            "input_projection": {"weight": g.w(dim, 2 * dim, std=std), "bias": g.b(dim)},
            "single_step_action_encoder": {"weight": g.w(dim, action_dim, std=0.5), "bias": g.b(dim)},
            "trajectory_encoder": _traj_sd(g, dim, 2 * dim, traj_layers, std),
            "action_padding_value": -5.0,
        })
    if not use_transformer:
        return {"ensemble_components": comps, "backbone": "hf-hub:timm/ViT-L-16-SigLIP2-384", "use_transformer": False,
                "history_length": history_length, "action_dim": action_dim, "num_models": n_members}
    return {"ensemble_components": comps}


def verifier_inputs(n_candidates: int, seed: int = 7, num_patches: int = 576, num_tokens: int = 64, dim: int = 1024,
                    min_hist: int = 4, hist_len: int = 10):
    """Unit-norm patch / text features (what extract_features returns) + candidate action histories in the
    verifier format (dims 0-5 small reals, gripper in {0,1}), ragged lengths in [min_hist, hist_len]."""
    g = torch.Generator().manual_seed(seed)
    pf = torch.nn.functional.normalize(torch.randn(1, num_patches, dim, generator=g), dim=-1)
    tf = torch.nn.functional.normalize(torch.randn(1, num_tokens, dim, generator=g), dim=-1)
    hists = []
    for i in range(n_candidates):
        n = min_hist + (i * 3) % (hist_len - min_hist + 1)
        h = torch.randn(n, 7, generator=g) * 0.02
        h[:, 6] = (torch.rand(n, generator=g) > 0.5).float()
        hists.append(h.double().numpy())
    return pf, tf, hists


def verifier_batch_inputs(batch: int, seed: int = 7, num_patches: int = 576, num_tokens: int = 64, dim: int = 1024, hist_len: int = 10,
                          pad_value: float = -5.0):
    """A validation batch of `batch` DISTINCT (image, text, history) triples at the feature boundary: unit-norm patch / text
    features [B, P, D] / [B, T, D] and histories [B, H, 7], short ones left-padded with the padding value (BridgeDataset)."""
    g = torch.Generator().manual_seed(seed)
    pf = torch.nn.functional.normalize(torch.randn(batch, num_patches, dim, generator=g), dim=-1)
    tf = torch.nn.functional.normalize(torch.randn(batch, num_tokens, dim, generator=g), dim=-1)
    hist = torch.randn(batch, hist_len, 7, generator=g) * 0.02
    hist[:, :, 6] = (torch.rand(batch, hist_len, generator=g) > 0.5).float()
    for b in range(batch):
        hist[b, : (b * 2) % 5] = pad_value
    return pf, tf, hist


# ------------------------------------------------------------------------------------------------ transformers
def vit_state(g: _G, *, dim: int, layers: int, heads: int, mlp: int, patch: int, n_pos: int, layerscale: bool = False,
              prefix_tokens: int = 0, post_ln: bool = True) -> Dict[str, Tensor]:
    """Generic pre-LN ViT encoder in neutral names (used for SigLIP, SigLIP2, DINOv2 towers):
    patch.weight [dim, 3*p*p] (conv kernel flattened c,py,px), patch.bias, pos [n_pos, dim], optional
    prefix [prefix_tokens, dim] (CLS/register tokens), blocks.i.{ln1,ln2}.{weight,bias}, blocks.i.{q,k,v,o}.{weight,bias},
    blocks.i.{fc1,fc2}.{weight,bias}, optional blocks.i.{ls1,ls2}, post_ln.{weight,bias}."""
    sd = {"patch.weight": g.w(dim, 3 * patch * patch), "patch.bias": g.b(dim), "pos": g.w(n_pos, dim)}
    if prefix_tokens:
        sd["prefix"] = g.w(prefix_tokens, dim)
    for i in range(layers):
        p = f"blocks.{i}."
        for n in ("ln1", "ln2"):
            sd[p + n + ".weight"] = g.ln_w(dim)
            sd[p + n + ".bias"] = g.b(dim)
        for n in ("q", "k", "v", "o"):
            sd[p + n + ".weight"] = g.w(dim, dim)
            sd[p + n + ".bias"] = g.b(dim)
        sd[p + "fc1.weight"] = g.w(mlp, dim)
        sd[p + "fc1.bias"] = g.b(mlp)
        sd[p + "fc2.weight"] = g.w(dim, mlp)
        sd[p + "fc2.bias"] = g.b(dim)
        if layerscale:
            sd[p + "ls1"] = g.ln_w(dim)
            sd[p + "ls2"] = g.ln_w(dim)
    if post_ln:
        sd["post_ln.weight"] = g.ln_w(dim)
        sd["post_ln.bias"] = g.b(dim)
    return sd


def decoder_state(g: _G, *, dim: int, layers: int, Hq: int, Hkv: int, D: int, mlp: int, rms_base: float,
                  vocab: int = 0) -> Dict[str, Tensor]:
    """Gemma/Llama-style decoder in HF names: layers.i.{input_layernorm,post_attention_layernorm}.weight,
    layers.i.self_attn.{q,k,v,o}_proj.weight, layers.i.mlp.{gate,up,down}_proj.weight, norm.weight,
    optional embed_tokens.weight."""
    sd = {}
    for i in range(layers):
        p = f"layers.{i}."
        sd[p + "input_layernorm.weight"] = g.rms_w(dim, rms_base)
        sd[p + "post_attention_layernorm.weight"] = g.rms_w(dim, rms_base)
        sd[p + "self_attn.q_proj.weight"] = g.w(Hq * D, dim)
        sd[p + "self_attn.k_proj.weight"] = g.w(Hkv * D, dim)
        sd[p + "self_attn.v_proj.weight"] = g.w(Hkv * D, dim)
        sd[p + "self_attn.o_proj.weight"] = g.w(dim, Hq * D)
        sd[p + "mlp.gate_proj.weight"] = g.w(mlp, dim)
        sd[p + "mlp.up_proj.weight"] = g.w(mlp, dim)
        sd[p + "mlp.down_proj.weight"] = g.w(dim, mlp)
    sd["norm.weight"] = g.rms_w(dim, rms_base)
    if vocab:
        sd["embed_tokens.weight"] = g.w(vocab, dim)
    return sd


# ------------------------------------------------------------------------------------------------ pi0 (P1)
def pi0_state(c: dict, seed: int = 1234, nontrivial: bool = True, std: float = 0.1, device="cpu",
              wdtype=torch.float32) -> Dict[str, Tensor]:
    """Neutral pi0 state dict (fp32 master copy) for a config dict with keys lm_dim, lm_mlp, ex_dim, ex_mlp, layers,
    Hq, Hkv, D, vocab, vit_dim, vit_mlp, vit_layers, vit_heads, patch, image. Key layout: see oracle/cover_ref/pi0.py.
    device / wdtype: draw the checkpoint directly in HBM in bf16 (bench: PI0_FULL is 3.3 G parameters); pi0's own five
    projections are always returned in fp32 (the reference keeps them fp32, modeling_pi0.py:488-494)."""
    g = _G(seed, nontrivial, std, device, wdtype)
    n_patches = (c["image"] // c["patch"]) ** 2
    sd = {}
    for k, v in vit_state(g, dim=c["vit_dim"], layers=c["vit_layers"], heads=c["vit_heads"], mlp=c["vit_mlp"],
                          patch=c["patch"], n_pos=n_patches).items():
        sd["vision." + k] = v
    sd["projector.weight"] = g.w(c["lm_dim"], c["vit_dim"])
    sd["projector.bias"] = g.b(c["lm_dim"])
    for k, v in decoder_state(g, dim=c["lm_dim"], layers=c["layers"], Hq=c["Hq"], Hkv=c["Hkv"], D=c["D"], mlp=c["lm_mlp"],
                              rms_base=0.0, vocab=c["vocab"]).items():
        sd["lm." + k] = v
    for k, v in decoder_state(g, dim=c["ex_dim"], layers=c["layers"], Hq=c["Hq"], Hkv=c["Hkv"], D=c["D"], mlp=c["ex_mlp"],
                              rms_base=0.0).items():
        sd["expert." + k] = v
    pw = c["ex_dim"]
    for n, (o, i) in {"state_proj": (pw, 32), "action_in_proj": (pw, 32), "action_out_proj": (32, pw),
                      "action_time_mlp_in": (pw, 2 * pw), "action_time_mlp_out": (pw, pw)}.items():
        sd[n + ".weight"] = g.w(o, i).float()
        sd[n + ".bias"] = g.b(o)
    return sd


PI0_FULL = dict(lm_dim=2048, lm_mlp=16384, ex_dim=1024, ex_mlp=4096, layers=18, Hq=8, Hkv=1, D=256, vocab=257152,
                vit_dim=1152, vit_mlp=4304, vit_layers=27, vit_heads=16, patch=14, image=224, chunk=4)


# ------------------------------------------------------------------------------------------------ OpenVLA-7B (P2)
OPENVLA_7B = dict(dino_dim=1024, dino_layers=24, dino_heads=16, dino_mlp=4096, dino_prefix=5,
                  sig_dim=1152, sig_layers=27, sig_heads=16, sig_mlp=4304, patch=14, image=224,
                  llm_dim=4096, llm_layers=32, Hq=32, Hkv=32, D=128, llm_mlp=11008, vocab=32064, tok_vocab=32000, n_bins=256)
OPENVLA_SMALL = dict(dino_dim=128, dino_layers=3, dino_heads=4, dino_mlp=256, dino_prefix=5,
                     sig_dim=128, sig_layers=3, sig_heads=2, sig_mlp=200, patch=14, image=56,
                     llm_dim=256, llm_layers=2, Hq=4, Hkv=4, D=64, llm_mlp=512, vocab=1088, tok_vocab=1024, n_bins=256)


def openvla_state(c: dict, seed: int = 1234, nontrivial: bool = True, std: float = 0.02, device="cpu",
                  wdtype=torch.float32, peaked: bool = False) -> Dict[str, Tensor]:
    """Prismatic / OpenVLA-7B shaped checkpoint (no such model exists in the reference):
    dino.* and siglip.* (vit_state layout), projector.fc{1,2,3}.{weight,bias}, llm.* (decoder_state layout, Llama),
    lm_head.weight.

    peaked=True: the SAME draws (same seeds, same order), re-scaled so that the checkpoint behaves like a trained one where it matters
    for parity checks -- decisions with a margin. An i.i.d. N(0, std) decoder is chaotic (every layer's update is as large as the
    stream: two bf16 evaluation paths of the 7B shapes end ~8-10 % apart) and its logits are flat (256 Gaussian bins: top-1 / top-2
    margins of a fraction of that noise), so "bit-exact arg-max" can hardly ever be decided on it. Three changes, all standard:
      * depth-scaled residual projections: o_proj / down_proj x (2 L)^-1/2 / 4 (the GPT-2 / Llama initialisation rule, and a further 1/4:
        the 64 sub-layer updates of the 7B stack then add up to a third of the embedding's magnitude instead of all of it);
      * token embeddings at unit scale (x 1 / std), as large as the sum of the layer updates, so the stream has a clean component;
      * a peaked action head: the n_bins action rows of lm_head are multiplied by log-normal gains exp(2 z) (normalised to unit RMS,
        seed + 7): a few bins carry most of the probability mass, as after training.
    Measured at the 7B shapes on the GPU (tools/dbg/r05/peaked_sweep.py, greedy M = 1 vs M = 8 decode rows, i.e. two tilings of the same bf16
    arithmetic, 8 prompts x 7 steps): steps whose top-1 / top-2 margin exceeds twice the logit difference 24 of 56 on the flat checkpoint,
    44 with (2 L)^-1/2 alone, 55 of 56 with the extra 1/4 at sigma 1.5 (30 of them by more than 10 x; 31 different token rows among the 32
    sampled candidates, greedy decoding picks action tokens 98 % of the time). Against the e4m3 pipeline (bench.py --dtype fp8, teacher-forced
    per step, rows of 32 whose bf16 margin exceeds twice the fp8 logit error): sigma 1.5 16-27 rows per step, sigma 2.0 (the default) 16-29,
    sigma 2.5 21-29 with the top-1 agreement starting to drop (profiles/r05_peaked_checkpoint_sweep.txt)."""
    g = _G(seed, nontrivial, std, device, wdtype)
    n_patches = (c["image"] // c["patch"]) ** 2
    sd = {}
    for k, v in vit_state(g, dim=c["dino_dim"], layers=c["dino_layers"], heads=c["dino_heads"], mlp=c["dino_mlp"],
                          patch=c["patch"], n_pos=n_patches + c["dino_prefix"], layerscale=True,
                          prefix_tokens=c["dino_prefix"], post_ln=False).items():
        sd["dino." + k] = v
    for k, v in vit_state(g, dim=c["sig_dim"], layers=c["sig_layers"], heads=c["sig_heads"], mlp=c["sig_mlp"],
                          patch=c["patch"], n_pos=n_patches, post_ln=False).items():
        sd["siglip." + k] = v
    fused = c["dino_dim"] + c["sig_dim"]
    for n, (o, i) in {"fc1": (4 * fused, fused), "fc2": (c["llm_dim"], 4 * fused), "fc3": (c["llm_dim"], c["llm_dim"])}.items():
        sd[f"projector.{n}.weight"] = g.w(o, i)
        sd[f"projector.{n}.bias"] = g.b(o)
    for k, v in decoder_state(g, dim=c["llm_dim"], layers=c["llm_layers"], Hq=c["Hq"], Hkv=c["Hkv"], D=c["D"],
                              mlp=c["llm_mlp"], rms_base=1.0, vocab=c["vocab"]).items():
        sd["llm." + k] = v
    sd["lm_head.weight"] = g.w(c["vocab"], c["llm_dim"])
    if peaked:
        import os
        L = c["llm_layers"]
        rs = (2.0 * L) ** -0.5 * float(os.environ.get("COVER_SYNTH_RES", "0.25"))         # (experiment knobs: tools/dbg/r05/peaked_sweep.py)
        for l in range(L):
            sd[f"llm.layers.{l}.self_attn.o_proj.weight"] *= rs
            sd[f"llm.layers.{l}.mlp.down_proj.weight"] *= rs
        sd["llm.embed_tokens.weight"] *= (float(os.environ.get("COVER_SYNTH_EMBED", "1.0")) / std)
        z = torch.randn(c["n_bins"], generator=torch.Generator().manual_seed(seed + 7))
        gain = torch.exp(float(os.environ.get("COVER_SYNTH_SIGMA", "2.0")) * z)
        gain = (gain / gain.pow(2).mean().sqrt()).to(sd["lm_head.weight"].device, sd["lm_head.weight"].dtype)
        lo, hi = c["tok_vocab"] - c["n_bins"], c["tok_vocab"]
        sd["lm_head.weight"][lo:hi] *= gain[:, None]
    return sd


SIGLIP2_L = dict(dim=1024, layers=24, heads=16, mlp=4096, patch=16, image=384, context_length=64, vocab=256000)
SIGLIP2_SMALL = dict(dim=128, layers=2, heads=2, mlp=256, patch=16, image=64, context_length=16, vocab=128)


def siglip2_state(c: dict, seed: int = 4321, nontrivial: bool = True, std: float = 0.02, device="cpu",
                  wdtype=torch.float32) -> Dict[str, Tensor]:
    """Verifier backbone (SigLIP2 ViT-L/16-384 shapes): image.* / text.* towers, text.tok_emb, text.proj.{weight,bias}."""
    g = _G(seed, nontrivial, std, device, wdtype)
    n_patches = (c["image"] // c["patch"]) ** 2
    sd = {}
    for k, v in vit_state(g, dim=c["dim"], layers=c["layers"], heads=c["heads"], mlp=c["mlp"], patch=c["patch"],
                          n_pos=n_patches, post_ln=False).items():
        sd["image." + k] = v
    for k, v in vit_state(g, dim=c["dim"], layers=c["layers"], heads=c["heads"], mlp=c["mlp"], patch=c["patch"],
                          n_pos=c["context_length"], post_ln=True).items():
        if not k.startswith("patch."):
            sd["text." + k] = v
    sd["text.tok_emb"] = g.w(c["vocab"], c["dim"])
    sd["text.proj.weight"] = g.w(c["dim"], c["dim"])
    sd["text.proj.bias"] = g.b(c["dim"])
    return sd


# ------------------------------------------------------------------------------------------------ test double: tokenizer
class CharTokenizer:
    """Character-level stand-in for the HF tokenizers the pi0-FAST policy is built around (neither `google/paligemma-3b-pt-224` nor
    `physical-intelligence/fast` can be downloaded here): the call surface PI0FAST.create_input_tokens / extract_actions use
    (modeling_pi0fast.py:570-640, 794-859) -- `__call__`, `pad`, `batch_decode`, `encode`, `vocab_size`, `eos_token_id`,
    `pad_token_id` -- over ids = 3 + ord(char). TEST INFRASTRUCTURE: the same object drives the reference's own functions in
    oracle/gen_golden_pi0fast.py and cover_vla_amd.pi0fast.PI0FASTPolicy in the tests."""
    pad_token_id, eos_token_id, bos_token_id = 0, 1, 2

    def __init__(self, vocab_size: int = 512, padding_side: str = "right"):
        self.vocab_size, self.padding_side = vocab_size, padding_side

    def _ids(self, text, add_special_tokens):
        ids = [3 + ord(c) for c in text]
        return ([self.bos_token_id] + ids) if add_special_tokens else ids

    def _pad(self, seqs, masks=None):
        n = max((len(s) for s in seqs), default=0)
        ids = torch.full((len(seqs), n), self.pad_token_id, dtype=torch.long)
        mask = torch.zeros(len(seqs), n, dtype=torch.long)
        for i, s in enumerate(seqs):
            m = [1] * len(s) if masks is None else masks[i]
            if self.padding_side == "left":
                ids[i, n - len(s):] = torch.tensor(s, dtype=torch.long)
                mask[i, n - len(s):] = torch.tensor(m, dtype=torch.long)
            else:
                ids[i, :len(s)] = torch.tensor(s, dtype=torch.long)
                mask[i, :len(s)] = torch.tensor(m, dtype=torch.long)
        return {"input_ids": ids, "attention_mask": mask}

    def __call__(self, texts, add_special_tokens=True, return_tensors="pt", padding="longest", truncation=False):
        single = isinstance(texts, str)
        return self._pad([self._ids(t, add_special_tokens) for t in ([texts] if single else texts)])

    def pad(self, batch, padding="longest", max_length=None, return_tensors="pt"):
        return self._pad([list(s) for s in batch["input_ids"]], [list(m) for m in batch["attention_mask"]])

    def encode(self, text, return_tensors="pt", padding=False):
        return torch.tensor([self._ids(text, True)], dtype=torch.long)

    def batch_decode(self, tokens, skip_special_tokens=True):
        out = []
        for row in (tokens.tolist() if torch.is_tensor(tokens) else tokens):
            out.append("".join(chr(t - 3) for t in row if t >= 3 or not skip_special_tokens))
        return out
