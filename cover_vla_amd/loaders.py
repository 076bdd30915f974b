"""On-disk formats of the reference -> the neutral state dicts the MI355X classes consume.

pi0: `config.json` + `model.safetensors` written by `PI0Policy.save_pretrained`
(lerobot_custom/lerobot/common/policies/pretrained.py:77-150). Key layout = what
`conversion_scripts/convert_pi0_to_hf_lerobot.py:67-245,384-390` produces:
    model.paligemma_with_expert.paligemma.vision_tower.vision_model.{embeddings.*, encoder.layers.N.*, post_layernorm.*}
    model.paligemma_with_expert.paligemma.multi_modal_projector.linear.{weight,bias}
    model.paligemma_with_expert.paligemma.language_model.model.{embed_tokens.weight, layers.N.*, norm.weight}
    model.paligemma_with_expert.paligemma.language_model.lm_head.weight          (tied, ignored)
    model.paligemma_with_expert.gemma_expert.model.{layers.N.*, norm.weight}     (embed_tokens / lm_head unused)
    model.{state_proj, action_in_proj, action_out_proj, action_time_mlp_in, action_time_mlp_out}.{weight,bias}
The verifier's merged checkpoint needs no conversion (cover_vla_amd.verifier reads its reference layout directly).
"""
from __future__ import annotations

import json
import os
import re
from typing import Dict, Tuple

import torch

_PWE = "model.paligemma_with_expert."
_VT = _PWE + "paligemma.vision_tower.vision_model."
_LM = _PWE + "paligemma.language_model.model."
_EX = _PWE + "gemma_expert.model."
_PROJ = ("state_proj", "action_in_proj", "action_out_proj", "action_time_mlp_in", "action_time_mlp_out")
_VIT_LAYER = {"layer_norm1": "ln1", "layer_norm2": "ln2", "self_attn.q_proj": "q", "self_attn.k_proj": "k",
              "self_attn.v_proj": "v", "self_attn.out_proj": "o", "mlp.fc1": "fc1", "mlp.fc2": "fc2"}


def pi0_reference_to_neutral(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        if not k.startswith("model."):
            k = "model." + k  # state dicts taken from PI0FlowMatching directly carry no "model." prefix
        if k.startswith(_VT):
            r = k[len(_VT):]
            if r == "embeddings.patch_embedding.weight":
                out["vision.patch.weight"] = v.reshape(v.shape[0], -1)
            elif r == "embeddings.patch_embedding.bias":
                out["vision.patch.bias"] = v
            elif r == "embeddings.position_embedding.weight":
                out["vision.pos"] = v
            elif r.startswith("post_layernorm."):
                out["vision.post_ln." + r.split(".")[-1]] = v
            else:
                m = re.match(r"encoder\.layers\.(\d+)\.(.+)\.(weight|bias)$", r)
                if m and m.group(2) in _VIT_LAYER:
                    out[f"vision.blocks.{m.group(1)}.{_VIT_LAYER[m.group(2)]}.{m.group(3)}"] = v
        elif k.startswith(_PWE + "paligemma.multi_modal_projector.linear."):
            out["projector." + k.split(".")[-1]] = v
        elif k.startswith(_LM):
            out["lm." + k[len(_LM):]] = v
        elif k.startswith(_EX):
            r = k[len(_EX):]
            if not r.startswith("embed_tokens"):
                out["expert." + r] = v
        else:
            m = re.match(r"model\.(%s)\.(weight|bias)$" % "|".join(_PROJ), k)
            if m:
                out[f"{m.group(1)}.{m.group(2)}"] = v
    return out


def neutral_to_pi0_reference(sd: Dict[str, torch.Tensor], patch: int) -> Dict[str, torch.Tensor]:
    """Inverse map (export / tests): neutral -> the reference's safetensors key layout."""
    inv = {v: k for k, v in _VIT_LAYER.items()}
    out = {}
    for k, v in sd.items():
        if k == "vision.patch.weight":
            out[_VT + "embeddings.patch_embedding.weight"] = v.reshape(v.shape[0], 3, patch, patch)
        elif k == "vision.patch.bias":
            out[_VT + "embeddings.patch_embedding.bias"] = v
        elif k == "vision.pos":
            out[_VT + "embeddings.position_embedding.weight"] = v
        elif k.startswith("vision.post_ln."):
            out[_VT + "post_layernorm." + k.split(".")[-1]] = v
        elif k.startswith("vision.blocks."):
            _, _, i, name, wb = k.split(".")
            out[f"{_VT}encoder.layers.{i}.{inv[name]}.{wb}"] = v
        elif k.startswith("projector."):
            out[_PWE + "paligemma.multi_modal_projector.linear." + k.split(".")[-1]] = v
        elif k.startswith("lm."):
            out[_LM + k[3:]] = v
        elif k.startswith("expert."):
            out[_EX + k[7:]] = v
        else:
            out["model." + k] = v
    return out


def infer_pi0_sizes(n: Dict[str, torch.Tensor], chunk: int) -> dict:
    """Size dict (cover_vla_amd.synth.PI0_FULL layout) from tensor shapes."""
    vit_dim = n["vision.patch.bias"].shape[0]
    patch = int(round((n["vision.patch.weight"].shape[1] // 3) ** 0.5))
    n_pos = n["vision.pos"].shape[0]
    layers = 1 + max(int(k.split(".")[2]) for k in n if k.startswith("lm.layers."))
    vit_layers = 1 + max(int(k.split(".")[2]) for k in n if k.startswith("vision.blocks."))
    lm_dim = n["lm.norm.weight"].shape[0]
    ex_dim = n["expert.norm.weight"].shape[0]
    kD = n["lm.layers.0.self_attn.k_proj.weight"].shape[0]
    qD = n["lm.layers.0.self_attn.q_proj.weight"].shape[0]
    return dict(lm_dim=lm_dim, lm_mlp=n["lm.layers.0.mlp.gate_proj.weight"].shape[0], ex_dim=ex_dim,
                ex_mlp=n["expert.layers.0.mlp.gate_proj.weight"].shape[0], layers=layers, vocab=n["lm.embed_tokens.weight"].shape[0],
                vit_dim=vit_dim, vit_mlp=n["vision.blocks.0.fc1.weight"].shape[0], vit_layers=vit_layers, patch=patch,
                image=int(round(n_pos ** 0.5)) * patch, chunk=chunk, _kD=kD, _qD=qD)


_NORM_KEY = re.compile(r"^(?:model\.)?(normalize_inputs|normalize_targets|unnormalize_outputs)\.buffer_(\w+)\.(mean|std|min|max)$")


def pi0_normalization(sd: Dict[str, torch.Tensor], cfg: dict) -> dict:
    """The checkpoint's Normalize / Unnormalize buffers (normalize.py:44-107: `normalize_inputs.buffer_observation_state.mean`
    ..., `unnormalize_outputs.buffer_action.std` ...) and the mode of each feature type from config.json's
    `normalization_mapping` (configs/policies.py; INT-ACT checkpoints: all IDENTITY, pi0_finetune_bridge.json:6-10).
    Returns {"state": (mode, a, b), "action": (mode, a, b)} with (a, b) = (mean, std) or (min, max) fp32, None for IDENTITY.
    Raises when a non-IDENTITY mode has no finite buffers in the checkpoint (the reference asserts the same,
    normalize.py:167-168)."""
    nm = {str(k).upper(): str(v).upper().split(".")[-1] for k, v in (cfg.get("normalization_mapping") or {}).items()}
    bufs = {}
    for k, v in sd.items():
        m = _NORM_KEY.match(k)
        if m:
            bufs[(m.group(1), m.group(2), m.group(3))] = v.to(torch.float32)
    out = {}
    for name, ftype, module, feat in (("state", "STATE", "normalize_inputs", "observation_state"),
                                      ("action", "ACTION", "unnormalize_outputs", "action")):
        mode = nm.get(ftype, "IDENTITY")
        if mode == "IDENTITY":
            out[name] = ("IDENTITY", None, None)
            continue
        ka, kb = ("mean", "std") if mode == "MEAN_STD" else ("min", "max")
        a, b = bufs.get((module, feat, ka)), bufs.get((module, feat, kb))
        if a is None or b is None or torch.isinf(a).any() or torch.isinf(b).any():
            raise ValueError(f"checkpoint uses {mode} normalisation for {ftype} but carries no finite "
                             f"{module}.buffer_{feat}.{ka}/{kb} (normalize.py:167-168 asserts the same)")
        out[name] = (mode, a, b)
    return out


def load_pi0_pretrained(path: str, head_dim: int = 256, vit_heads: int = 16) -> Tuple[Dict[str, torch.Tensor], dict, dict]:
    """Directory with config.json + model.safetensors -> (neutral state dict, size dict, raw config). The raw config gains
    `_normalization` = pi0_normalization(...) (the Normalize / Unnormalize buffers of the checkpoint)."""
    from safetensors.torch import load_file
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    raw = load_file(os.path.join(path, "model.safetensors"))
    cfg["_normalization"] = pi0_normalization(raw, cfg)
    n = pi0_reference_to_neutral(raw)
    c = infer_pi0_sizes(n, int(cfg.get("chunk_size", 50)))
    c["D"] = head_dim
    c["Hkv"] = c.pop("_kD") // head_dim
    c["Hq"] = c.pop("_qD") // head_dim
    c["vit_heads"] = vit_heads
    return n, c, cfg


# ------------------------------------------------------------------------------------------------ pi0-FAST
# `PI0FASTPolicy.save_pretrained` writes the policy's state dict: `model.pi0_paligemma.<PaliGemmaForConditionalGeneration keys>`
# (modeling_pi0fast.py:462) -- the same HF module (transformers 4.48.3 layout: vision_tower.vision_model.*, multi_modal_projector.
# linear.*, language_model.model.*, language_model.lm_head.weight tied to embed_tokens) that pi0 keeps under
# `model.paligemma_with_expert.paligemma.` (key list pinned in tests/golden/pi0_checkpoint_keys.json), plus the Normalize buffers.
_FAST = "model.pi0_paligemma."


def pi0fast_reference_to_neutral(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """-> neutral vision.* / projector.* / lm.* (what cover_vla_amd.pi0fast.PI0FASTTokens consumes)."""
    moved = {}
    for k, v in sd.items():
        kk = k if k.startswith("model.") else "model." + k
        if kk.startswith(_FAST):
            moved[_PWE + "paligemma." + kk[len(_FAST):]] = v
    return pi0_reference_to_neutral(moved)


def neutral_to_pi0fast_reference(sd: Dict[str, torch.Tensor], patch: int) -> Dict[str, torch.Tensor]:
    keep = {k: v for k, v in sd.items() if k.startswith(("vision.", "projector.", "lm."))}
    out = {}
    for k, v in neutral_to_pi0_reference(keep, patch).items():
        out[_FAST + k[len(_PWE + "paligemma."):]] = v
    out[_FAST + "language_model.lm_head.weight"] = sd["lm.embed_tokens.weight"]      # tied
    return out


def load_pi0fast_pretrained(path: str, head_dim: int = 256, vit_heads: int = 16) -> Tuple[Dict[str, torch.Tensor], dict, dict]:
    """Directory with config.json + model.safetensors of a pi0-FAST policy -> (neutral state dict, size dict, raw config)."""
    from safetensors.torch import load_file
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    raw = load_file(os.path.join(path, "model.safetensors"))
    cfg["_normalization"] = pi0_normalization(raw, cfg)
    n = pi0fast_reference_to_neutral(raw)
    if "lm.embed_tokens.weight" not in n:
        raise ValueError(f"{path}: no model.pi0_paligemma.* tensors -- not a pi0-FAST checkpoint")
    vit_dim = n["vision.patch.bias"].shape[0]
    patch = int(round((n["vision.patch.weight"].shape[1] // 3) ** 0.5))
    kD, qD = n["lm.layers.0.self_attn.k_proj.weight"].shape[0], n["lm.layers.0.self_attn.q_proj.weight"].shape[0]
    c = dict(lm_dim=n["lm.norm.weight"].shape[0], lm_mlp=n["lm.layers.0.mlp.gate_proj.weight"].shape[0],
             layers=1 + max(int(k.split(".")[2]) for k in n if k.startswith("lm.layers.")), vocab=n["lm.embed_tokens.weight"].shape[0],
             vit_dim=vit_dim, vit_mlp=n["vision.blocks.0.fc1.weight"].shape[0],
             vit_layers=1 + max(int(k.split(".")[2]) for k in n if k.startswith("vision.blocks.")), patch=patch,
             image=int(round(n["vision.pos"].shape[0] ** 0.5)) * patch, D=head_dim, Hkv=kD // head_dim, Hq=qD // head_dim, vit_heads=vit_heads)
    return n, c, cfg


# ------------------------------------------------------------------------------------------------ verifier checkpoint
# The merged verifier checkpoint is a pickled .pt (efficient_ensemble_merged.py:37-53: {"ensemble_components": [ {sub-module name ->
# state dict | scalar} ... ], optionally backbone / use_transformer / history_length / action_dim / num_models}). Unpickling arbitrary
# files is code execution, so the serving side never does it: the .pt is converted ONCE, where it was produced or downloaded and is
# trusted, into tensors in safetensors + the non-tensor fields in JSON, and EfficientEnsembleMerged loads that pair. A .pt
# path is still accepted, through torch.load(weights_only=True) (tensors, dicts, lists and plain scalars only).
_VER_META = "cover_verifier.json"
_VER_TENSORS = "cover_verifier.safetensors"


def _flatten_verifier(ck: dict):
    tensors, meta = {}, {"format": "cover-verifier-1", "top": {}, "components": []}
    for k, v in ck.items():
        if k != "ensemble_components":
            meta["top"][k] = v if isinstance(v, (str, int, float, bool)) or v is None else str(v)
    for i, comp in enumerate(ck["ensemble_components"]):
        cm = {"subs": {}, "scalars": {}}
        for name, val in comp.items():
            if isinstance(val, dict):
                keys = []
                for kk, t in val.items():
                    if not torch.is_tensor(t):
                        t = torch.as_tensor(t)
                    tensors[f"components.{i}.{name}.{kk}"] = t.detach().cpu().contiguous()
                    keys.append(kk)
                cm["subs"][name] = keys
            elif val is None:
                cm["scalars"][name] = None
            elif torch.is_tensor(val):
                tensors[f"components.{i}.{name}"] = val.detach().cpu().contiguous()
                cm["subs"][name] = None                      # a bare tensor
            else:
                cm["scalars"][name] = float(val) if isinstance(val, (int, float)) else val
        meta["components"].append(cm)
    return tensors, meta


def verifier_pt_to_safetensors(pt_path: str, out_dir: str) -> str:
    """ONE-OFF, on a machine where the .pt is trusted: unpickles the merged verifier checkpoint and writes out_dir/cover_verifier.safetensors
    (every tensor, named components.<i>.<sub-module>.<key>) + out_dir/cover_verifier.json (structure and scalars). Returns out_dir."""
    from safetensors.torch import save_file
    ck = torch.load(pt_path, map_location="cpu", weights_only=False)
    return save_verifier_checkpoint(ck, out_dir)


def save_verifier_checkpoint(ck: dict, out_dir: str) -> str:
    from safetensors.torch import save_file
    tensors, meta = _flatten_verifier(ck)
    os.makedirs(out_dir, exist_ok=True)
    save_file(tensors, os.path.join(out_dir, _VER_TENSORS))
    with open(os.path.join(out_dir, _VER_META), "w") as f:
        json.dump(meta, f, indent=1)
    return out_dir


def load_verifier_checkpoint(path: str) -> dict:
    """Directory written by verifier_pt_to_safetensors -> the reference's checkpoint dict (no unpickling). A .pt / .pth file is read with
    torch.load(weights_only=True); if that refuses it (custom classes inside), convert it once with verifier_pt_to_safetensors."""
    if os.path.isdir(path):
        from safetensors.torch import load_file
        with open(os.path.join(path, _VER_META)) as f:
            meta = json.load(f)
        if meta.get("format") != "cover-verifier-1":
            raise ValueError(f"{path}: not a converted verifier checkpoint")
        tensors = load_file(os.path.join(path, _VER_TENSORS))
        comps = []
        for i, cm in enumerate(meta["components"]):
            comp = dict(cm["scalars"])
            for name, keys in cm["subs"].items():
                comp[name] = tensors[f"components.{i}.{name}"] if keys is None else {kk: tensors[f"components.{i}.{name}.{kk}"] for kk in keys}
            comps.append(comp)
        return dict(meta["top"], ensemble_components=comps)
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:   # noqa: BLE001 -- whatever the restricted unpickler refuses
        raise ValueError(f"{path}: torch.load(weights_only=True) refused this checkpoint ({type(e).__name__}: {e}). Convert it once, on a machine where "
                         "the file is trusted, with cover_vla_amd.loaders.verifier_pt_to_safetensors(pt_path, out_dir) and pass out_dir instead") from e

