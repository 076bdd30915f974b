"""G3: ONE full-width Llama-2-7B decoder layer (4096 wide, 32 x 128 MHA, MLP 11008, RMSNorm(w) eps 1e-5,
HF rotary) through HF transformers' own LlamaModel (one layer + final norm, bf16, eager attention) at the sequence geometry the
OpenVLA-7B headline decision runs: a shared causal prefix [BOS | 256 patch rows], 8 prompts of 16..23 text tokens behind it
(8 sequences of 273..280 tokens for HF; ONE 448-row two-group pass here), then ONE decode row per candidate (N = 32 = 8 prompts x 4
samples, each a different hidden row at position 257 + len) over HF's KV cache.

The OpenVLA-7B profile has no code in the reference: HF's Llama is what `north_star` calls "the reference CPU/HF
path" for it. The weights are NOT stored (202 M parameters): they are cover_vla_amd.synth.decoder_state at L7 / SEED -- the tests
regenerate them from the same seeded CPU generator. Stored (bf16 bit patterns): every 4th prefix output row, every text output
row of prompts 0 and 5, the LAST valid text row of every prompt (the row the action head reads), post-RoPE K and V of heads
0 / 13 / 31 for the same prefix rows and prompt 5's text rows, and the 32 decode output rows. Beside HF's bf16 outputs the fixture
holds the SAME bf16-rounded parameters and inputs evaluated by the same module in fp32 (f32_*: every 16th prefix row, prompt 5's text
rows, the last text rows, the decode rows) -- HF's bf16 eager attention rounds QK^T to bf16 before the softmax, so its own outputs sit
~0.8e-2 (rel-L2) from that value; a path with fp32 scores is judged against the fp32 evaluation, with HF-bf16's own distance as the bar.

Usage: PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_llama7b.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

L7 = dict(dim=4096, Hq=32, Hkv=32, D=128, mlp=11008)
SEED, N_PATCH, P, LT, S = 43, 256, 8, 24, 4
HEADS = (0, 13, 31)


def llama7b_weights():
    """Seeded full-width layer (HF key names, fp32 masters; norm weights around 1 as in a trained Llama)."""
    from cover_vla_amd import synth
    g = synth._G(SEED, True, 0.02)
    return synth.decoder_state(g, dim=L7["dim"], layers=1, Hq=L7["Hq"], Hkv=L7["Hkv"], D=L7["D"], mlp=L7["mlp"], rms_base=1.0)


def llama7b_inputs():
    """Seeded hidden rows (shared by the generator and the tests): BOS row, patch rows, text rows [P, LT] (valid up to lens[p]),
    one decode row per candidate [P * S]."""
    g = torch.Generator().manual_seed(SEED + 1)
    bf = lambda *s: torch.randn(*s, L7["dim"], generator=g).to(torch.bfloat16)
    lens = torch.tensor([16 + p for p in range(P)], dtype=torch.int32)
    return dict(bos=bf(1), patches=bf(N_PATCH), text=bf(P, LT), dec=bf(P * S), lens=lens)


def main():
    import warnings
    warnings.filterwarnings("ignore")
    import transformers
    from transformers import LlamaConfig, LlamaModel
    from gen_golden import save
    sd = llama7b_weights()
    cfg = LlamaConfig(hidden_size=L7["dim"], intermediate_size=L7["mlp"], num_hidden_layers=1, num_attention_heads=L7["Hq"],
                      num_key_value_heads=L7["Hkv"], head_dim=L7["D"], vocab_size=8, rms_norm_eps=1e-5, rope_theta=10000.0,
                      max_position_embeddings=512, attn_implementation="eager", tie_word_embeddings=False)
    m = LlamaModel(cfg).eval()
    L = m.layers[0]
    with torch.no_grad():
        L.input_layernorm.weight.copy_(sd["layers.0.input_layernorm.weight"])
        L.post_attention_layernorm.weight.copy_(sd["layers.0.post_attention_layernorm.weight"])
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            getattr(L.self_attn, n).weight.copy_(sd[f"layers.0.self_attn.{n}.weight"])
        for n in ("gate_proj", "up_proj", "down_proj"):
            getattr(L.mlp, n).weight.copy_(sd[f"layers.0.mlp.{n}.weight"])
        m.norm.weight.copy_(sd["norm.weight"])
    inv_freq = m.rotary_emb.inv_freq.clone()
    m = m.to(torch.bfloat16)            # fp32 masters -> bf16: one rounding, as loading a bf16 checkpoint
    # Module.to(bfloat16) also rounds the rotary embedding's (non-persistent, fp32) inv_freq BUFFER to bf16, which a
    # from_pretrained(torch_dtype=bfloat16) load does not do (the buffer is not in the checkpoint and is built in fp32): restore it
    m.rotary_emb.inv_freq = inv_freq
    import copy
    m32 = copy.deepcopy(m).float()      # the SAME bf16-rounded parameters evaluated in fp32: the value both bf16 paths approximate
    i = llama7b_inputs()
    T0 = 1 + N_PATCH
    bits = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)
    out = {}
    last_rows, dec_rows, last32, dec32 = [], [], [], []
    with torch.no_grad():
        for p in range(P):
            n = int(i["lens"][p])
            seq = torch.cat([i["bos"], i["patches"], i["text"][p, :n]], 0)[None]
            o32 = m32(inputs_embeds=seq.float(), use_cache=True)
            if p == 0:
                out["f32_prefix_every16"] = o32.last_hidden_state[0, 1:T0][::16].numpy()
            if p == 5:
                out["f32_text_p5"] = o32.last_hidden_state[0, T0:].numpy()
            last32.append(o32.last_hidden_state[0, -1])
            for s in range(S):
                d32 = m32(inputs_embeds=i["dec"][p * S + s][None, None].float(), past_key_values=copy.deepcopy(o32.past_key_values), use_cache=True)
                dec32.append(d32.last_hidden_state[0, 0])
            o = m(inputs_embeds=seq, use_cache=True)
            h, past = o.last_hidden_state[0], o.past_key_values
            lay = past.layers[0] if hasattr(past, "layers") else None
            k, v = (lay.keys, lay.values) if lay is not None else past[0]          # [1, H, T, D] post-RoPE
            if p == 0:
                out["prefix_every4"] = bits(h[1:T0][::4])                           # patch rows 0, 4, ... (position 1 + 4 j)
                out["k_prefix_every4"] = bits(k[0, list(HEADS), 1:T0][:, ::4])
                out["v_prefix_every4"] = bits(v[0, list(HEADS), 1:T0][:, ::4])
                out["text_p0"] = bits(h[T0:])
            if p == 5:
                out["text_p5"] = bits(h[T0:])
                out["k_text_p5"] = bits(k[0, list(HEADS), T0:])
                out["v_text_p5"] = bits(v[0, list(HEADS), T0:])
            last_rows.append(h[-1])
            for s in range(S):
                # HF's cache object is updated in place by a forward: every sample restarts from a copy of the prompt's cache
                pc = copy.deepcopy(past)
                d = m(inputs_embeds=i["dec"][p * S + s][None, None], past_key_values=pc, use_cache=True)
                dec_rows.append(d.last_hidden_state[0, 0])
    out["last_text_rows"] = bits(torch.stack(last_rows))
    out["decode_rows"] = bits(torch.stack(dec_rows))
    out["f32_last_text_rows"] = torch.stack(last32).numpy()
    out["f32_decode_rows"] = torch.stack(dec32).numpy()
    save("hf_llama7b_layer", seed=SEED, n_patch=N_PATCH, P=P, LT=LT, S=S, heads=np.array(HEADS), transformers_version=transformers.__version__,
         **{"l7_" + k: v for k, v in L7.items()}, **out)


if __name__ == "__main__":
    main()
