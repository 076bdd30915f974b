"""pi0 goldens: the REFERENCE's PI0FlowMatching.sample_actions (imported from /root/reference in this container
only) on tiny seeded configs. See gen_golden.py for the rules; harness-side shims only:
  * sys.modules shells for the lerobot packages (their heavy __init__s pull datasets/jsonlines) pointing at the real dirs
  * a no-op draccus stub, dataclass stubs for lerobot.common.optim.{optimizers,schedulers}
  * transformers-5.x attribute aliases the reference (written for 4.48.3) expects on PaliGemmaForConditionalGeneration
  * the two 4.48.3 -> 5.x behaviour changes are neutralised so the goldens carry 4.48.3 semantics:
      get_image_features divided by sqrt(hidden) in 4.48.3 (not in 5.x)  -> divide here, in the tensor dtype
      Gemma embed_tokens was an UNSCALED nn.Embedding in 4.48.3 (scaled in 5.x) -> plain lookup here
    (the 4.48.3 behaviour is from public knowledge of that release and cannot be re-verified offline: flagged in
    tests/golden/README.md)
"""
from __future__ import annotations

import dataclasses
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
LR = os.path.join(REF, "lerobot_custom", "lerobot")


def _shell(name, path=None, **attrs):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference_pi0():
    class _Choice:
        @classmethod
        def register_subclass(cls, name):
            def deco(c):
                return c
            return deco

        @classmethod
        def get_choice_name(cls, c):
            return "pi0"

    _shell("draccus", ChoiceRegistry=_Choice, wrap=lambda *a, **k: (lambda f: f), parse=None, dump=None,
           config_type=lambda *a, **k: (lambda f: f), set_config_type=lambda *a, **k: None)
    _shell("lerobot", LR)
    _shell("lerobot.common", os.path.join(LR, "common"))
    _shell("lerobot.common.policies", os.path.join(LR, "common", "policies"))
    _shell("lerobot.common.policies.pi0", os.path.join(LR, "common", "policies", "pi0"))
    _shell("lerobot.common.utils", os.path.join(LR, "common", "utils"))
    _shell("lerobot.configs", os.path.join(LR, "configs"))
    _shell("lerobot.common.optim")

    @dataclasses.dataclass
    class AdamWConfig:
        lr: float = 1e-3
        betas: tuple = (0.9, 0.95)
        eps: float = 1e-8
        weight_decay: float = 1e-10

    @dataclasses.dataclass
    class CosineDecayWithWarmupSchedulerConfig:
        peak_lr: float = 0
        decay_lr: float = 0
        num_warmup_steps: int = 0
        num_decay_steps: int = 0

    _shell("lerobot.common.optim.optimizers", AdamWConfig=AdamWConfig, OptimizerConfig=object)
    _shell("lerobot.common.optim.schedulers", CosineDecayWithWarmupSchedulerConfig=CosineDecayWithWarmupSchedulerConfig,
           LRSchedulerConfig=object)
    import importlib
    pwe = importlib.import_module("lerobot.common.policies.pi0.paligemma_with_expert")
    mp = importlib.import_module("lerobot.common.policies.pi0.modeling_pi0")
    return pwe, mp


def build_reference_model(pwe, mp, tiny):
    """PI0FlowMatching with tiny widths: mutate the DEFAULT sub-configs (passing dicts trips a latent bug at
    paligemma_with_expert.py:117)."""
    from transformers import PaliGemmaForConditionalGeneration
    # 5.x -> 4.48.3 attribute aliases used at paligemma_with_expert.py:198,233,245
    if not hasattr(PaliGemmaForConditionalGeneration, "_cover_alias"):
        PaliGemmaForConditionalGeneration._cover_alias = True
        PaliGemmaForConditionalGeneration.vision_tower = property(lambda self: self.model.vision_tower)
        PaliGemmaForConditionalGeneration.language_model = property(
            lambda self: types.SimpleNamespace(model=self.model.language_model))

    orig_init = pwe.PaliGemmaWithExpertConfig.__init__

    def patched_init(self, *a, **k):
        orig_init(self, *a, **k)
        pc, ec = self.paligemma_config, self.gemma_expert_config
        t, v = pc.text_config, pc.vision_config
        t.hidden_size, t.intermediate_size, t.num_hidden_layers = tiny["lm_dim"], tiny["lm_mlp"], tiny["layers"]
        t.num_attention_heads, t.num_key_value_heads, t.head_dim = tiny["Hq"], tiny["Hkv"], tiny["D"]
        t.vocab_size = tiny["vocab"]
        pc.hidden_size = tiny["lm_dim"]
        pc.projection_dim = tiny["lm_dim"]
        pc.vocab_size = tiny["vocab"]
        pc.image_token_index = tiny["vocab"] - 1
        pc.image_token_id = tiny["vocab"] - 1
        v.hidden_size, v.intermediate_size, v.num_hidden_layers = tiny["vit_dim"], tiny["vit_mlp"], tiny["vit_layers"]
        v.num_attention_heads, v.patch_size, v.image_size = tiny["vit_heads"], tiny["patch"], tiny["image"]
        v.projection_dim = tiny["lm_dim"]
        ec.hidden_size, ec.intermediate_size, ec.num_hidden_layers = tiny["ex_dim"], tiny["ex_mlp"], tiny["layers"]
        ec.num_attention_heads, ec.num_key_value_heads, ec.head_dim = tiny["Hq"], tiny["Hkv"], tiny["D"]
        ec.vocab_size = tiny["vocab"]

    pwe.PaliGemmaWithExpertConfig.__init__ = patched_init
    try:
        cfg = types.SimpleNamespace(
            freeze_vision_encoder=True, train_expert_only=False, paligemma_pretrained_path=None,
            attention_implementation="eager", max_state_dim=32, max_action_dim=32, proj_width=tiny["ex_dim"],
            train_state_proj=True, chunk_size=tiny["chunk"], num_steps=10, use_cache=True)
        model = mp.PI0FlowMatching(cfg)
    finally:
        pwe.PaliGemmaWithExpertConfig.__init__ = orig_init
    model.eval()
    pg = model.paligemma_with_expert

    def embed_image_4483(image):
        feats = pg.paligemma.model.get_image_features(image).pooler_output
        return feats / (pg.config.paligemma_config.text_config.hidden_size ** 0.5)

    def embed_tokens_4483(tokens):
        return torch.nn.functional.embedding(tokens, pg.paligemma.model.language_model.embed_tokens.weight)

    pg.embed_image = embed_image_4483
    pg.embed_language_tokens = embed_tokens_4483
    return model


def neutral_to_reference(model, sd):
    """Copy a neutral (cover_vla_amd.synth) state dict into the reference module's parameters."""
    pg = model.paligemma_with_expert
    vt = pg.paligemma.model.vision_tower
    vt = getattr(vt, "vision_model", vt)
    lm = pg.paligemma.model.language_model
    ex = pg.gemma_expert.model
    with torch.no_grad():
        def put(p, t):
            assert p.shape == t.shape, (p.shape, t.shape)
            p.copy_(t.to(p.dtype))
        ps = vt.embeddings.patch_embedding
        put(ps.weight, sd["vision.patch.weight"].view(ps.weight.shape))
        put(ps.bias, sd["vision.patch.bias"])
        put(vt.embeddings.position_embedding.weight, sd["vision.pos"])
        for i, L in enumerate(vt.encoder.layers):
            p = f"vision.blocks.{i}."
            put(L.layer_norm1.weight, sd[p + "ln1.weight"]); put(L.layer_norm1.bias, sd[p + "ln1.bias"])
            put(L.layer_norm2.weight, sd[p + "ln2.weight"]); put(L.layer_norm2.bias, sd[p + "ln2.bias"])
            for a, b in (("q_proj", "q"), ("k_proj", "k"), ("v_proj", "v"), ("out_proj", "o")):
                put(getattr(L.self_attn, a).weight, sd[p + b + ".weight"]); put(getattr(L.self_attn, a).bias, sd[p + b + ".bias"])
            put(L.mlp.fc1.weight, sd[p + "fc1.weight"]); put(L.mlp.fc1.bias, sd[p + "fc1.bias"])
            put(L.mlp.fc2.weight, sd[p + "fc2.weight"]); put(L.mlp.fc2.bias, sd[p + "fc2.bias"])
        put(vt.post_layernorm.weight, sd["vision.post_ln.weight"]); put(vt.post_layernorm.bias, sd["vision.post_ln.bias"])
        mmp = pg.paligemma.model.multi_modal_projector.linear
        put(mmp.weight, sd["projector.weight"]); put(mmp.bias, sd["projector.bias"])
        put(lm.embed_tokens.weight, sd["lm.embed_tokens.weight"])
        for mod, pre in ((lm, "lm."), (ex, "expert.")):
            for i, L in enumerate(mod.layers):
                p = f"{pre}layers.{i}."
                put(L.input_layernorm.weight, sd[p + "input_layernorm.weight"])
                put(L.post_attention_layernorm.weight, sd[p + "post_attention_layernorm.weight"])
                for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
                    put(getattr(L.self_attn, n).weight, sd[p + f"self_attn.{n}.weight"])
                for n in ("gate_proj", "up_proj", "down_proj"):
                    put(getattr(L.mlp, n).weight, sd[p + f"mlp.{n}.weight"])
            put(mod.norm.weight, sd[pre + "norm.weight"])
        for n in ("state_proj", "action_in_proj", "action_out_proj", "action_time_mlp_in", "action_time_mlp_out"):
            put(getattr(model, n).weight, sd[n + ".weight"]); put(getattr(model, n).bias, sd[n + ".bias"])


# small but HIP-friendly: every GEMM K is a multiple of 128, head_dim 64 for the decoders; the tower exercises the
# zero-padding paths (head_dim 32 -> 64, MLP 200 -> 256, patch K 588 -> 640)
TINY = dict(lm_dim=256, lm_mlp=512, ex_dim=128, ex_mlp=256, layers=2, Hq=4, Hkv=1, D=64, vocab=96,
            vit_dim=128, vit_mlp=200, vit_layers=2, vit_heads=4, patch=14, image=56, chunk=4)


def pi0_inputs(tiny, B, L, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(1, 3, tiny["image"], tiny["image"], generator=g) * 2 - 1
    images = [img.repeat(B, 1, 1, 1)]
    img_masks = [torch.ones(B, dtype=torch.bool)]
    n_prompts = max(1, B // 2)
    lens = [3 + (i * 5) % (L - 3) for i in range(n_prompts)]
    toks = torch.zeros(B, L, dtype=torch.long)
    masks = torch.zeros(B, L, dtype=torch.bool)
    for b in range(B):
        pi = b % n_prompts
        gg = torch.Generator().manual_seed(seed * 100 + pi)
        toks[b, :lens[pi]] = torch.randint(1, tiny["vocab"] - 1, (lens[pi],), generator=gg)
        masks[b, :lens[pi]] = True
    state = torch.zeros(B, 32)
    state[:, :7] = (torch.rand(1, 7, generator=g) * 2 - 1)
    noise = torch.randn(B, tiny["chunk"], 32, generator=g)
    return images, img_masks, toks, masks, state, noise


def gen_pi0(save):
    import warnings
    warnings.filterwarnings("ignore")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cover_vla_amd import synth
    pwe, mp = import_reference_pi0()
    for name, B, L, seed in [("pi0_tiny_b6", 6, 12, 21), ("pi0_tiny_b1", 1, 8, 22), ("pi0_tiny_b40", 40, 16, 23)]:
        tiny = dict(TINY)
        model = build_reference_model(pwe, mp, tiny)
        sd = synth.pi0_state(tiny, seed=seed)
        neutral_to_reference(model, sd)
        images, img_masks, toks, masks, state, noise = pi0_inputs(tiny, B, L, seed)
        with torch.no_grad():
            # intermediates through the reference's own methods
            pe, ppad, patt = model.embed_prefix(images, img_masks, toks, masks)
            x = model.sample_actions(images, img_masks, toks, masks, state, noise=noise.clone())
            se, _, _ = model.embed_suffix(state, noise, torch.ones(B))
        save(name, B=B, L=L, seed=seed, actions=x, prefix_embs=pe.float(), suffix_embs_t1=se.float(),
             **{"tiny_" + k: v for k, v in tiny.items()})
