"""P2 pin: outputs of HF transformers modules (this container's transformers, fp32, eager attention) on tiny random
configs, stored together with the weights in the oracle's neutral layout. The OpenVLA-7B profile has no reference code
(SURVEY.md §0), so the oracle's Llama / SigLIP / DINOv2 blocks are pinned to these public implementations instead."""
from __future__ import annotations

import torch


def gen_hf(save):
    import transformers
    from transformers import (Dinov2WithRegistersConfig, Dinov2WithRegistersModel, LlamaConfig, LlamaForCausalLM,
                              SiglipVisionConfig, SiglipVisionModel)
    torch.manual_seed(0)
    # ---------------- Llama
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                      vocab_size=96, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=128, head_dim=16,
                      attn_implementation="eager", tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval().float()
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.2) if p.dim() > 1 else torch.nn.init.normal_(p, mean=1.0, std=0.1)
    ids = torch.randint(0, 96, (2, 11))
    with torch.no_grad():
        logits = m(input_ids=ids).logits
    sd = {}
    for i, L in enumerate(m.model.layers):
        p = f"llm.layers.{i}."
        sd[p + "input_layernorm.weight"] = L.input_layernorm.weight
        sd[p + "post_attention_layernorm.weight"] = L.post_attention_layernorm.weight
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[p + f"self_attn.{n}.weight"] = getattr(L.self_attn, n).weight
        for n in ("gate_proj", "up_proj", "down_proj"):
            sd[p + f"mlp.{n}.weight"] = getattr(L.mlp, n).weight
    sd["llm.norm.weight"] = m.model.norm.weight
    sd["llm.embed_tokens.weight"] = m.model.embed_tokens.weight
    sd["lm_head.weight"] = m.lm_head.weight
    save("hf_llama_tiny", ids=ids, logits=logits, transformers_version=transformers.__version__, **{k: v.detach() for k, v in sd.items()})

    # ---------------- SigLIP vision tower
    vc = SiglipVisionConfig(hidden_size=48, intermediate_size=80, num_hidden_layers=2, num_attention_heads=4, image_size=28,
                            patch_size=14, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6, attn_implementation="eager")
    vm = SiglipVisionModel(vc).eval().float()
    for p in vm.parameters():
        torch.nn.init.normal_(p, std=0.2) if p.dim() > 1 else torch.nn.init.normal_(p, mean=0.5, std=0.2)
    px = torch.randn(2, 3, 28, 28)
    with torch.no_grad():
        out = vm(pixel_values=px, output_hidden_states=True)
    core = getattr(vm, "vision_model", vm)
    sd = {"patch.weight": core.embeddings.patch_embedding.weight.reshape(48, -1), "patch.bias": core.embeddings.patch_embedding.bias,
          "pos": core.embeddings.position_embedding.weight, "post_ln.weight": core.post_layernorm.weight,
          "post_ln.bias": core.post_layernorm.bias}
    for i, L in enumerate(core.encoder.layers):
        p = f"blocks.{i}."
        sd[p + "ln1.weight"], sd[p + "ln1.bias"] = L.layer_norm1.weight, L.layer_norm1.bias
        sd[p + "ln2.weight"], sd[p + "ln2.bias"] = L.layer_norm2.weight, L.layer_norm2.bias
        for a, b in (("q_proj", "q"), ("k_proj", "k"), ("v_proj", "v"), ("out_proj", "o")):
            sd[p + b + ".weight"], sd[p + b + ".bias"] = getattr(L.self_attn, a).weight, getattr(L.self_attn, a).bias
        sd[p + "fc1.weight"], sd[p + "fc1.bias"] = L.mlp.fc1.weight, L.mlp.fc1.bias
        sd[p + "fc2.weight"], sd[p + "fc2.bias"] = L.mlp.fc2.weight, L.mlp.fc2.bias
    save("hf_siglip_tiny", pixels=px, last_hidden=out.last_hidden_state, hidden_1=out.hidden_states[1],
         **{k: v.detach() for k, v in sd.items()})

    # ---------------- DINOv2 with registers (CLS + 4 registers, LayerScale, exact GELU)
    dc = Dinov2WithRegistersConfig(hidden_size=48, num_hidden_layers=2, num_attention_heads=4, mlp_ratio=2, image_size=28,
                                   patch_size=14, num_register_tokens=4, layer_norm_eps=1e-6, hidden_act="gelu",
                                   layerscale_value=1.0, attn_implementation="eager")
    dm = Dinov2WithRegistersModel(dc).eval().float()
    for p in dm.parameters():
        torch.nn.init.normal_(p, std=0.2) if p.dim() > 1 else torch.nn.init.normal_(p, mean=0.5, std=0.2)
    with torch.no_grad():
        dout = dm(pixel_values=px, output_hidden_states=True)
    e = dm.embeddings
    n_p = (28 // 14) ** 2
    pos = e.position_embeddings[0]                      # [1 + n_p, dim] (CLS + patches); registers carry no position
    prefix = torch.cat([e.cls_token[0], e.register_tokens[0]], 0)
    pos_full = torch.cat([pos[:1], torch.zeros(4, 48), pos[1:]], 0)
    sd = {"patch.weight": e.patch_embeddings.projection.weight.reshape(48, -1), "patch.bias": e.patch_embeddings.projection.bias,
          "pos": pos_full, "prefix": prefix}
    for i, L in enumerate(dm.encoder.layer):
        p = f"blocks.{i}."
        sd[p + "ln1.weight"], sd[p + "ln1.bias"] = L.norm1.weight, L.norm1.bias
        sd[p + "ln2.weight"], sd[p + "ln2.bias"] = L.norm2.weight, L.norm2.bias
        att = L.attention
        qm = att.attention
        sd[p + "q.weight"], sd[p + "q.bias"] = qm.query.weight, qm.query.bias
        sd[p + "k.weight"], sd[p + "k.bias"] = qm.key.weight, qm.key.bias
        sd[p + "v.weight"], sd[p + "v.bias"] = qm.value.weight, qm.value.bias
        sd[p + "o.weight"], sd[p + "o.bias"] = att.output.dense.weight, att.output.dense.bias
        sd[p + "fc1.weight"], sd[p + "fc1.bias"] = L.mlp.fc1.weight, L.mlp.fc1.bias
        sd[p + "fc2.weight"], sd[p + "fc2.bias"] = L.mlp.fc2.weight, L.mlp.fc2.bias
        sd[p + "ls1"], sd[p + "ls2"] = L.layer_scale1.lambda1, L.layer_scale2.lambda1
    save("hf_dinov2_tiny", pixels=px, hidden_2=dout.hidden_states[2], hidden_1=dout.hidden_states[1],
         **{k: v.detach() for k, v in sd.items()})
