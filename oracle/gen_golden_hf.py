"""P2 pin: outputs of HF transformers modules (this container's transformers, fp32, eager attention) on tiny random
configs, stored together with the weights in the oracle's neutral layout. The OpenVLA-7B profile has no reference code
(SURVEY.md §0), so the oracle's Llama / SigLIP / DINOv2 blocks are pinned to these public implementations instead."""
from __future__ import annotations

import numpy as np
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
    gen_hf_siglip2_bridge(save)
    gen_hf_openvla_e2e(save)


def _put(p, t):
    assert tuple(p.shape) == tuple(t.shape), (tuple(p.shape), tuple(t.shape))
    with torch.no_grad():
        p.copy_(t.to(p.dtype))


def _load_siglip_layers(layers, sd, prefix):
    for i, L in enumerate(layers):
        p = f"{prefix}{i}."
        _put(L.layer_norm1.weight, sd[p + "ln1.weight"]); _put(L.layer_norm1.bias, sd[p + "ln1.bias"])
        _put(L.layer_norm2.weight, sd[p + "ln2.weight"]); _put(L.layer_norm2.bias, sd[p + "ln2.bias"])
        for a, b in (("q_proj", "q"), ("k_proj", "k"), ("v_proj", "v"), ("out_proj", "o")):
            _put(getattr(L.self_attn, a).weight, sd[p + b + ".weight"]); _put(getattr(L.self_attn, a).bias, sd[p + b + ".bias"])
        _put(L.mlp.fc1.weight, sd[p + "fc1.weight"]); _put(L.mlp.fc1.bias, sd[p + "fc1.bias"])
        _put(L.mlp.fc2.weight, sd[p + "fc2.weight"]); _put(L.mlp.fc2.bias, sd[p + "fc2.bias"])


def siglip2_bridge_weights(seed=55):
    """Seeded verifier-backbone weights (cover_vla_amd.synth, CPU generator: reproducible in the tests) at SIGLIP2_SMALL."""
    from cover_vla_amd import synth
    c = dict(synth.SIGLIP2_SMALL)
    return c, synth.siglip2_state(c, seed=seed, std=0.08)


def openvla_e2e_weights(seed=77):
    """Seeded OpenVLA-shaped weights at OPENVLA_SMALL; the DINOv2 register tokens carry no position (HF adds positions to
    CLS + patches only), so those four position rows are zero."""
    from cover_vla_amd import synth
    c = dict(synth.OPENVLA_SMALL)
    sd = synth.openvla_state(c, seed=seed, std=0.08)
    sd["dino.pos"][1:c["dino_prefix"]] = 0
    return c, sd


def gen_hf_siglip2_bridge(save):
    """Verifier backbone pin: HF SiglipVisionModel + SiglipTextModel with the hook semantics of
    VLA_SigLIP2_Bridge.extract_features (finetune_trajectory_bridge_ddp.py:264-278, 297-355): patch features = OUTPUT OF
    THE LAST BLOCK'S ATTENTION MODULE (forward hook on encoder.layers[-1].self_attn), text features = transformer output ->
    final_layer_norm -> head applied to EVERY position (HF applies the head to the last position only: the module is called
    on all of them here, and its own pooled output is stored too). Weights: siglip2_bridge_weights() copied INTO the HF
    modules (the fixture stores outputs only)."""
    import transformers
    from transformers import SiglipTextConfig, SiglipTextModel, SiglipVisionConfig, SiglipVisionModel
    c, sd = siglip2_bridge_weights()
    dim, mlp, heads, layers, patch, image, ctx, vocab = (c[k] for k in ("dim", "mlp", "heads", "layers", "patch", "image", "context_length", "vocab"))
    vc = SiglipVisionConfig(hidden_size=dim, intermediate_size=mlp, num_hidden_layers=layers, num_attention_heads=heads, image_size=image,
                            patch_size=patch, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6, attn_implementation="eager")
    vm = SiglipVisionModel(vc).eval().float()
    tc = SiglipTextConfig(vocab_size=vocab, hidden_size=dim, intermediate_size=mlp, num_hidden_layers=layers, num_attention_heads=heads,
                          max_position_embeddings=ctx, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6, attn_implementation="eager",
                          projection_size=dim, bos_token_id=None, eos_token_id=None, pad_token_id=None)
    tm = SiglipTextModel(tc).eval().float()
    vcore, tcore = getattr(vm, "vision_model", vm), getattr(tm, "text_model", tm)
    _put(vcore.embeddings.patch_embedding.weight, sd["image.patch.weight"].view(dim, 3, patch, patch))
    _put(vcore.embeddings.patch_embedding.bias, sd["image.patch.bias"])
    _put(vcore.embeddings.position_embedding.weight, sd["image.pos"])
    _load_siglip_layers(vcore.encoder.layers, sd, "image.blocks.")
    _put(tcore.embeddings.token_embedding.weight, sd["text.tok_emb"])
    _put(tcore.embeddings.position_embedding.weight, sd["text.pos"])
    _load_siglip_layers(tcore.encoder.layers, sd, "text.blocks.")
    _put(tcore.final_layer_norm.weight, sd["text.post_ln.weight"]); _put(tcore.final_layer_norm.bias, sd["text.post_ln.bias"])
    _put(tcore.head.weight, sd["text.proj.weight"]); _put(tcore.head.bias, sd["text.proj.bias"])
    g = torch.Generator().manual_seed(5)
    px = torch.randn(2, 3, image, image, generator=g)
    ids = torch.randint(0, vocab, (2, ctx), generator=g)
    grabbed = {}
    hook = vcore.encoder.layers[-1].self_attn.register_forward_hook(lambda m, i, o: grabbed.__setitem__("attn", o[0] if isinstance(o, tuple) else o))
    with torch.no_grad():
        vm(pixel_values=px)
        to = tm(input_ids=ids)
        text_all = tcore.head(to.last_hidden_state)                   # head on every position (what the bridge's hook path does)
    hook.remove()
    pf = grabbed["attn"]
    pf = pf / pf.norm(dim=-1, keepdim=True)
    tf = text_all / text_all.norm(dim=-1, keepdim=True)
    save("hf_siglip2_bridge_tiny", pixels=px, ids=ids, patch_features=pf, text_features=tf, text_last_hidden=to.last_hidden_state,
         text_pooled=to.pooler_output, weight_seed=55, transformers_version=transformers.__version__)


def gen_hf_openvla_e2e(save):
    """END-TO-END P2 pin: HF Dinov2WithRegistersModel + SiglipVisionModel + a 3-layer GELU projector + LlamaForCausalLM
    composed per SURVEY.md Appendix D (second-to-last block features, DINOv2 drops CLS + registers, channel concat,
    [BOS | patches | prompt], greedy 7 tokens with HF's KV cache), in fp32 AND in bf16 (the same weights cast). Weights:
    openvla_e2e_weights() copied INTO the HF modules; the fixture stores the frame, the prompts, per-step logits and the
    greedy tokens of both precisions."""
    import transformers
    from transformers import (Dinov2WithRegistersConfig, Dinov2WithRegistersModel, LlamaConfig, LlamaForCausalLM,
                              SiglipVisionConfig, SiglipVisionModel)
    c, sd = openvla_e2e_weights()
    dc = Dinov2WithRegistersConfig(hidden_size=c["dino_dim"], num_hidden_layers=c["dino_layers"], num_attention_heads=c["dino_heads"],
                                   mlp_ratio=c["dino_mlp"] // c["dino_dim"], image_size=c["image"], patch_size=c["patch"],
                                   num_register_tokens=c["dino_prefix"] - 1, layer_norm_eps=1e-6, hidden_act="gelu", layerscale_value=1.0,
                                   attn_implementation="eager")
    dm = Dinov2WithRegistersModel(dc).eval().float()
    e = dm.embeddings
    _put(e.patch_embeddings.projection.weight, sd["dino.patch.weight"].view(c["dino_dim"], 3, c["patch"], c["patch"]))
    _put(e.patch_embeddings.projection.bias, sd["dino.patch.bias"])
    _put(e.position_embeddings, torch.cat([sd["dino.pos"][:1], sd["dino.pos"][c["dino_prefix"]:]], 0)[None])
    _put(e.cls_token, sd["dino.prefix"][None, :1]); _put(e.register_tokens, sd["dino.prefix"][None, 1:])
    for i, L in enumerate(dm.encoder.layer):
        p_ = f"dino.blocks.{i}."
        _put(L.norm1.weight, sd[p_ + "ln1.weight"]); _put(L.norm1.bias, sd[p_ + "ln1.bias"])
        _put(L.norm2.weight, sd[p_ + "ln2.weight"]); _put(L.norm2.bias, sd[p_ + "ln2.bias"])
        qm = L.attention.attention
        _put(qm.query.weight, sd[p_ + "q.weight"]); _put(qm.query.bias, sd[p_ + "q.bias"])
        _put(qm.key.weight, sd[p_ + "k.weight"]); _put(qm.key.bias, sd[p_ + "k.bias"])
        _put(qm.value.weight, sd[p_ + "v.weight"]); _put(qm.value.bias, sd[p_ + "v.bias"])
        _put(L.attention.output.dense.weight, sd[p_ + "o.weight"]); _put(L.attention.output.dense.bias, sd[p_ + "o.bias"])
        _put(L.mlp.fc1.weight, sd[p_ + "fc1.weight"]); _put(L.mlp.fc1.bias, sd[p_ + "fc1.bias"])
        _put(L.mlp.fc2.weight, sd[p_ + "fc2.weight"]); _put(L.mlp.fc2.bias, sd[p_ + "fc2.bias"])
        _put(L.layer_scale1.lambda1, sd[p_ + "ls1"]); _put(L.layer_scale2.lambda1, sd[p_ + "ls2"])
    vc = SiglipVisionConfig(hidden_size=c["sig_dim"], intermediate_size=c["sig_mlp"], num_hidden_layers=c["sig_layers"],
                            num_attention_heads=c["sig_heads"], image_size=c["image"], patch_size=c["patch"], hidden_act="gelu_pytorch_tanh",
                            layer_norm_eps=1e-6, attn_implementation="eager")
    vm = SiglipVisionModel(vc).eval().float()
    vcore = getattr(vm, "vision_model", vm)
    _put(vcore.embeddings.patch_embedding.weight, sd["siglip.patch.weight"].view(c["sig_dim"], 3, c["patch"], c["patch"]))
    _put(vcore.embeddings.patch_embedding.bias, sd["siglip.patch.bias"])
    _put(vcore.embeddings.position_embedding.weight, sd["siglip.pos"])
    _load_siglip_layers(vcore.encoder.layers, sd, "siglip.blocks.")
    lc = LlamaConfig(hidden_size=c["llm_dim"], intermediate_size=c["llm_mlp"], num_hidden_layers=c["llm_layers"], num_attention_heads=c["Hq"],
                     num_key_value_heads=c["Hkv"], vocab_size=c["vocab"], rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=256,
                     head_dim=c["D"], attn_implementation="eager", tie_word_embeddings=False)
    lm = LlamaForCausalLM(lc).eval().float()
    for i, L in enumerate(lm.model.layers):
        p_ = f"llm.layers.{i}."
        _put(L.input_layernorm.weight, sd[p_ + "input_layernorm.weight"])
        _put(L.post_attention_layernorm.weight, sd[p_ + "post_attention_layernorm.weight"])
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            _put(getattr(L.self_attn, n).weight, sd[p_ + f"self_attn.{n}.weight"])
        for n in ("gate_proj", "up_proj", "down_proj"):
            _put(getattr(L.mlp, n).weight, sd[p_ + f"mlp.{n}.weight"])
    _put(lm.model.norm.weight, sd["llm.norm.weight"]); _put(lm.model.embed_tokens.weight, sd["llm.embed_tokens.weight"])
    _put(lm.lm_head.weight, sd["lm_head.weight"])
    fused = c["dino_dim"] + c["sig_dim"]
    proj = torch.nn.Sequential(torch.nn.Linear(fused, 4 * fused), torch.nn.GELU(), torch.nn.Linear(4 * fused, c["llm_dim"]), torch.nn.GELU(),
                               torch.nn.Linear(c["llm_dim"], c["llm_dim"])).eval()
    for j, idx in enumerate((0, 2, 4)):
        _put(proj[idx].weight, sd[f"projector.fc{j + 1}.weight"]); _put(proj[idx].bias, sd[f"projector.fc{j + 1}.bias"])
    g = torch.Generator().manual_seed(11)
    frame = torch.randint(0, 256, (1, c["image"], c["image"], 3), generator=g, dtype=torch.uint8)
    P, Lt = 3, 9
    lens = torch.tensor([9, 6, 8], dtype=torch.int32)
    toks = torch.zeros(P, Lt, dtype=torch.long)
    for p_ in range(P):
        toks[p_, : lens[p_]] = torch.randint(2, c["tok_vocab"] - c["n_bins"], (int(lens[p_]),), generator=g)
    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

    def run(dtype):
        d_, v_, l_, pr_ = (m.to(dtype) for m in (dm, vm, lm, proj))
        x = frame.permute(0, 3, 1, 2).float() / 255.0
        xd = ((x - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1)).to(dtype)
        xs = ((x - 0.5) / 0.5).to(dtype)
        with torch.no_grad():
            hd = d_(pixel_values=xd, output_hidden_states=True).hidden_states[c["dino_layers"] - 1][:, c["dino_prefix"]:]
            hs = v_(pixel_values=xs, output_hidden_states=True).hidden_states[c["sig_layers"] - 1]
            img = pr_(torch.cat([hd, hs], dim=-1))[0]                                    # [n_patches, llm_dim]
            emb = l_.model.embed_tokens
            all_logits, all_tokens = [], []
            for p_ in range(P):
                L = int(lens[p_])
                seq = torch.cat([emb(torch.tensor([1])), img, emb(toks[p_, :L])], 0)[None]
                out = l_(inputs_embeds=seq, use_cache=True)
                past, lg = out.past_key_values, out.logits[0, -1]
                row_l, row_t = [], []
                for i in range(7):
                    row_l.append(lg.float())
                    t = int(torch.argmax(lg.float()[: c["tok_vocab"]]))
                    row_t.append(t)
                    if i == 6:
                        break
                    out = l_(inputs_embeds=emb(torch.tensor([[t]])), past_key_values=past, use_cache=True)
                    past, lg = out.past_key_values, out.logits[0, -1]
                all_logits.append(torch.stack(row_l))
                all_tokens.append(row_t)
        return torch.stack(all_logits), torch.tensor(all_tokens)

    lg32, tk32 = run(torch.float32)
    lg16, tk16 = run(torch.bfloat16)        # fp32 masters -> bf16 (one rounding, as loading a bf16 checkpoint)
    save("hf_openvla_e2e_tiny", frame=frame, toks=toks, lens=lens, logits_fp32=lg32, tokens_fp32=tk32, logits_bf16=lg16, tokens_bf16=tk16,
         weight_seed=77, transformers_version=transformers.__version__)
