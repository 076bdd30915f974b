#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE's own modules (imported from /root/reference, read-only, in THIS
container only) on seeded synthetic weights/inputs and writes small input/output fixtures to tests/golden/.

Nothing here travels to the GPU box except the .npz data it writes. The reference cannot be imported as-is
(missing pip deps: draccus, open_clip, timm, cv2, ...), so this harness installs sys.modules SHELLS/STUBS for the
packages the reference imports at module scope but that play no role in the arithmetic being pinned
(SURVEY.md §8c lists them). No reference source is copied: modules are imported from where they lie.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [verifier] [adapter] [pi0] [hf]
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from cover_vla_amd import synth  # noqa: E402


def _shell(name, path=None, **attrs):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu()
            v = v.float().numpy() if v.dtype == torch.bfloat16 else v.numpy()
        out[k] = np.asarray(v)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ======================================================================================== verifier (P1)
def import_reference_verifier():
    """bridge_verifier/ensemble_eval/{model,efficient_ensemble_merged}.py with stubbed open_clip / timm / ijson."""
    class Mlp(torch.nn.Module):  # timm.layers.mlp.Mlp defaults: fc1 -> exact GELU -> fc2 (drop = 0)
        def __init__(self, in_features, hidden_features=None, out_features=None):
            super().__init__()
            self.fc1 = torch.nn.Linear(in_features, hidden_features or in_features)
            self.act = torch.nn.GELU()
            self.fc2 = torch.nn.Linear(hidden_features or in_features, out_features or in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    _shell("timm")
    _shell("timm.layers")
    _shell("timm.layers.mlp", Mlp=Mlp)
    _shell("open_clip", create_model_from_pretrained=None, get_tokenizer=None, create_model_and_transforms=None)
    _shell("ijson")
    if "tqdm" not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except Exception:
            _shell("tqdm", tqdm=lambda x, **k: x)
    _shell("bridge_verifier", os.path.join(REF, "bridge_verifier"))
    _shell("bridge_verifier.ensemble_eval", os.path.join(REF, "bridge_verifier", "ensemble_eval"))
    import importlib
    model = importlib.import_module("bridge_verifier.ensemble_eval.model")
    eem = importlib.import_module("bridge_verifier.ensemble_eval.efficient_ensemble_merged")
    return model, eem


def build_reference_ensemble(model, eem, ckpt, pf, tf):
    """EfficientEnsembleMerged via __new__, trainable_models filled with the reference's module classes loaded from the
    synthetic merged checkpoint exactly as efficient_ensemble_merged.py:94-184 does; encoders stubbed at the feature
    boundary (extract_features returns the supplied pf/tf)."""
    ens = eem.EfficientEnsembleMerged.__new__(eem.EfficientEnsembleMerged)
    ens.device = "cpu"
    ens.use_transformer = bool(ckpt.get("use_transformer", True))
    ens.history_length = 10
    ens.action_dim = 7
    ens.num_models = len(ckpt["ensemble_components"])
    ens.trainable_models = []
    P, D = pf.shape[1], pf.shape[2]
    for cs in ckpt["ensemble_components"]:
        ta = model.TextAwareVisualExtraction(num_img_patches=P, vision_dim=D)
        ta.load_state_dict(cs["text_aware_visual_extraction"])
        vp = model.AttentionPooling(input_dim=D, output_dim=512, num_heads=8, num_layers=4, num_readouts=1)
        vp.load_state_dict(cs["vision_poolings"])
        tp = model.AttentionPooling(input_dim=tf.shape[2], output_dim=512, num_heads=8, num_layers=4, num_readouts=1)
        tp.load_state_dict(cs["text_pooling"])
        ip = torch.nn.Linear(1024, 512)
        ip.load_state_dict(cs["input_projection"])
        se = te = ce = None
        if ens.use_transformer:
            se = torch.nn.Linear(7, 512)
            se.load_state_dict(cs["single_step_action_encoder"])
            layer = torch.nn.TransformerEncoderLayer(d_model=512, nhead=8, dim_feedforward=1024, batch_first=False, dropout=0.1)
            te = torch.nn.TransformerEncoder(layer, num_layers=4)
            te.load_state_dict(cs["trajectory_encoder"])
        else:   # the module stack of efficient_ensemble_merged.py:150-157
            ce = torch.nn.Sequential(torch.nn.Linear(10 * 7, 512), torch.nn.LayerNorm(512), torch.nn.ReLU(), torch.nn.Dropout(0.1),
                                     torch.nn.Linear(512, 512))
            ce.load_state_dict(cs["complex_action_encoder"])
        for m in (ta, vp, tp, ip, se, te, ce):
            if m is not None:
                m.eval()
        ens.trainable_models.append({
            "text_aware_visual_extraction": ta, "vision_poolings": vp, "text_pooling": tp, "input_projection": ip,
            "single_step_action_encoder": se, "trajectory_encoder": te, "complex_action_encoder": ce,
            "action_padding_value": cs["action_padding_value"]})

    class _Feat:
        def extract_features(self, img, tok):
            return pf, tf
    ens.full_model_for_features = _Feat()
    ens.preprocess = lambda img: torch.zeros(3, 8, 8)
    ens.tokenizer = lambda texts, context_length=64: torch.zeros(len(texts), context_length, dtype=torch.long)
    ens.siglip_model = types.SimpleNamespace(context_length=64)
    return ens


def gen_verifier():
    import warnings
    warnings.filterwarnings("ignore")
    model, eem = import_reference_verifier()
    img = np.zeros((8, 8, 3), dtype=np.uint8)
    for members, N, group, seed in [(2, 16, 2, 11), (3, 40, 5, 7), (3, 32, 4, 3), (1, 1, 1, 5)]:
        ckpt = synth.verifier_checkpoint(members, seed=1234 + seed)
        pf, tf, hists = synth.verifier_inputs(N, seed=seed)
        ens = build_reference_ensemble(model, eem, ckpt, pf, tf)
        with torch.no_grad():
            score, instr, hist, gidx = ens.compute_max_similarity_scores_batch(
                [img] * N, ["put the spoon on the towel"] * N, hists, cfg_repeat_language_instructions=group)
            # per-member and fused embeddings through the reference's own methods
            hb = torch.tensor(np.array([np.vstack([np.ones((10 - len(h), 7)) * -5, h]) if len(h) < 10 else h for h in hists]),
                              dtype=torch.float32)
            its, acts = [], []
            for mi in range(members):
                it, act = ens.get_embeddings_from_model_batch(mi, pf, tf, hb)
                its.append(it[0:1])
                acts.append(act)
            its, acts = torch.stack(its), torch.stack(acts)
            f_it = its.mean(0)
            f_it = f_it / f_it.norm(dim=-1, keepdim=True)
            f_act = acts.mean(0)
            f_act = f_act / f_act.norm(dim=-1, keepdim=True)
            scores = (f_it @ f_act.T)[0]
        assert isinstance(gidx, torch.Tensor) and gidx.dtype == torch.int64 and gidx.dim() == 0
        save(f"verifier_m{members}_n{N}_g{group}", members=members, N=N, group=group, ckpt_seed=1234 + seed, input_seed=seed,
             its=its, acts=acts, scores=scores, max_score=np.float32(score), global_idx=np.int64(int(gidx)),
             hist_lens=np.array([len(h) for h in hists]))
    # MLP action-encoder variant (use_transformer = False): the same public call through the reference's complex_action_encoder
    ckpt = synth.verifier_checkpoint(2, seed=1234 + 21, use_transformer=False)
    pf, tf, hists = synth.verifier_inputs(8, seed=21)
    ens = build_reference_ensemble(model, eem, ckpt, pf, tf)
    with torch.no_grad():
        score, instr, hist, gidx = ens.compute_max_similarity_scores_batch([img] * 8, ["x"] * 8, hists, cfg_repeat_language_instructions=2)
        hb = torch.tensor(np.array([np.vstack([np.ones((10 - len(h), 7)) * -5, h]) if len(h) < 10 else h for h in hists]), dtype=torch.float32)
        acts = torch.stack([ens.get_embeddings_from_model_batch(mi, pf, tf, hb)[1] for mi in range(2)])
    save("verifier_cae_m2_n8_g2", members=2, N=8, group=2, ckpt_seed=1234 + 21, input_seed=21, acts=acts, max_score=np.float32(score),
         global_idx=np.int64(int(gidx)))
    # PUBLIC API of the class (boundary row b): the 4-tuple of compute_max_similarity_scores_batch with DISTINCT instructions
    # per group (max_instruction rule :441-445), predict (:295-307) and fuse_embeddings (:249-293), stub encoders at the
    # feature boundary, ndarray image input
    ckpt = synth.verifier_checkpoint(2, seed=1234 + 31)
    pf, tf, hists = synth.verifier_inputs(12, seed=31)
    ens = build_reference_ensemble(model, eem, ckpt, pf, tf)
    instrs = [f"instruction {i // 3}" for i in range(12)]
    with torch.no_grad():
        score, instr, hist, gidx = ens.compute_max_similarity_scores_batch([img] * 12, instrs, hists, cfg_repeat_language_instructions=3)
        assert isinstance(score, float) and isinstance(instr, str) and isinstance(hist, np.ndarray)
        # predict / fuse_embeddings stack the histories with np.array (:267): equal lengths only, no padding
        _, _, h10 = synth.verifier_inputs(6, seed=32, min_hist=10)
        p_hist, p_scores = ens.predict(img, "instruction 0", h10)
        f_it, f_act = ens.fuse_embeddings(img, "instruction 0", h10)
        s1, i1, h1, g1 = ens.compute_max_similarity_scores_batch([img], ["only"], hists[4:5], cfg_repeat_language_instructions=1)
    save("verifier_api_m2_n12_g3", ckpt_seed=1234 + 31, input_seed=31, max_score=np.float32(score), instr_index=np.int64(instrs.index(instr)),
         global_idx=np.int64(int(gidx)), hist=hist, predict_index=np.int64([i for i, h in enumerate(h10) if h is p_hist][0]),
         predict_scores=np.array([p_scores[str(i)] for i in range(6)], dtype=np.float64), fused_it=f_it, fused_act=f_act,
         stage1_score=np.float32(s1), stage1_idx=np.int64(int(g1)))
    # selection edge cases (G4): exact ties and the grouped rule, through the reference's own selection code path
    # by feeding features that make every candidate identical (all scores tie -> index 0 must win)
    ckpt = synth.verifier_checkpoint(2, seed=99)
    pf, tf, hists = synth.verifier_inputs(12, seed=99)
    same = [hists[0]] * 12
    ens = build_reference_ensemble(model, eem, ckpt, pf, tf)
    with torch.no_grad():
        score, _, _, gidx = ens.compute_max_similarity_scores_batch([img] * 12, ["x"] * 12, same, cfg_repeat_language_instructions=3)
    save("verifier_ties", global_idx=np.int64(int(gidx)), max_score=np.float32(score), hist=same[0])


def gen_verifier_training():
    """Validation forward of ONE verifier model through the reference's own training module
    (finetune_trajectory_bridge_ddp.py): VLA_SigLIP2_Bridge.__init__ + forward + the loss / accuracy code of the loops.
    The frozen SigLIP2 model is a shape-only stand-in object (attributes __init__ reads: dims, patch size, the two hooked
    modules); extract_features is replaced at the feature boundary by the supplied batch of B distinct (pf, tf)."""
    import importlib
    import warnings
    warnings.filterwarnings("ignore")
    model, _ = import_reference_verifier()
    ft = importlib.import_module("bridge_verifier.ensemble_eval.finetune_trajectory_bridge_ddp")
    nn = torch.nn

    class _Trunk(nn.Module):
        def __init__(self):
            super().__init__()
            self.num_features = 1024
            self.patch_embed = nn.Module()
            self.patch_embed.proj = nn.Conv2d(3, 4, 16, 16)
            blk = nn.Module()
            blk.attn = nn.Identity()
            self.blocks = nn.ModuleList([blk])

    class _Clip(nn.Module):
        def __init__(self):
            super().__init__()
            self.visual = nn.Module()
            self.visual.trunk = _Trunk()
            self.visual.image_size = (384, 384)
            self.text = nn.Module()
            self.text.output_dim = 1024
            self.text.transformer = nn.Identity()

    for use_tr, B, seed in [(True, 12, 41), (False, 6, 43)]:
        ckpt = synth.verifier_checkpoint(1, seed=1234 + seed, use_transformer=use_tr)
        cs = ckpt["ensemble_components"][0]
        cfg = model.ModelConfig(clip_model=_Clip(), history_length=10, action_dim=7)
        net = ft.VLA_SigLIP2_Bridge(cfg, use_transformer=use_tr).set_trainable_dtype(torch.float32)
        for name in ("text_aware_visual_extraction", "vision_poolings", "text_pooling", "input_projection",
                     "single_step_action_encoder", "trajectory_encoder", "complex_action_encoder"):
            if cs.get(name) is not None:
                getattr(net, name).load_state_dict(cs[name])
        logit_scale = 2.6592 + 0.01 * seed          # a trained value, not the init
        with torch.no_grad():
            net.logit_scale.fill_(logit_scale)
        net.eval()
        pf, tf, hist = synth.verifier_batch_inputs(B, seed=seed)
        net.extract_features = lambda images, text: (pf, tf)
        with torch.no_grad():
            li, la = net(torch.zeros(B, 3, 8, 8), torch.zeros(B, 64, dtype=torch.long), hist)
            labels = torch.arange(B)
            il = torch.nn.functional.cross_entropy(li, labels)
            al = torch.nn.functional.cross_entropy(la, labels)
            acc = ft.calculate_accuracy_metrics(li, la, B, "cpu")
        save(f"verifier_train_fwd_{'tr' if use_tr else 'mlp'}_b{B}", ckpt_seed=1234 + seed, input_seed=seed, use_transformer=use_tr,
             logit_scale=np.float32(logit_scale), B=B, hist=hist, image_logits=li, action_logits=la,
             image_loss=np.float32(il), action_loss=np.float32(al), loss=np.float32((il + al) / 2),
             acc_names=np.array(sorted(acc)), acc_values=np.array([acc[k] for k in sorted(acc)], dtype=np.float64))


# ======================================================================================== adapter math (P1 host glue)
def gen_adapter():
    """BridgeSimplerAdapter.postprocess / postprocess_verifier + geometry helpers on 64 random normalised action rows."""
    _shell("cv2")
    _shell("src", os.path.join(REF, "INT-ACT", "src"))
    _shell("src.experiments", os.path.join(REF, "INT-ACT", "src", "experiments"))
    _shell("src.experiments.env_adapters", os.path.join(REF, "INT-ACT", "src", "experiments", "env_adapters"))
    _shell("src.utils", os.path.join(REF, "INT-ACT", "src", "utils"))
    import importlib
    geometry = importlib.import_module("src.utils.geometry")
    simpler = importlib.import_module("src.experiments.env_adapters.simpler")
    stats_path = os.path.join(REF, "INT-ACT", "config", "dataset", "bridge_statistics.json")
    ad = simpler.BridgeSimplerAdapter.__new__(simpler.BridgeSimplerAdapter)
    import json
    with open(stats_path) as f:
        ad.dataset_statistics = json.load(f)
    ad.action_normalization_type = "bound"
    ad.state_normalization_type = "bound"
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1.2, 1.2, size=(64, 7)).astype(np.float32)
    acts[:8, 6] = [0.5, 0.4999, 0.5001, 0.0, 1.0, -0.3, 0.75, 0.25]
    exec_rows = ad.postprocess(acts.copy())
    ver_rows = ad.postprocess_verifier(acts.copy())
    exec_arr = np.asarray(exec_rows, dtype=np.float64)
    ver_arr = np.asarray(ver_rows, dtype=np.float64)
    eul = rng.uniform(-3, 3, size=(32, 3))
    ax = np.stack([geometry.euler2axangle(*e)[0] * geometry.euler2axangle(*e)[1] for e in eul])
    quats = rng.normal(size=(16, 4))
    quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    mats = np.stack([geometry.quat2mat(q) for q in quats])
    eulers = np.stack([np.array(geometry.mat2euler(m)) for m in mats])
    save("adapter_bridge", actions=acts, exec_rows=exec_arr, verifier_rows=ver_arr, euler_in=eul, axangle=ax, quats=quats,
         quat_mats=mats, mat_eulers=eulers)


if __name__ == "__main__":
    which = sys.argv[1:] or ["verifier", "adapter", "pi0", "hf"]
    torch.manual_seed(0)
    if "verifier" in which:
        gen_verifier()
    if "verifier_train" in which or "verifier" in which:
        gen_verifier_training()
    if "adapter" in which:
        gen_adapter()
    if "pi0" in which:
        from gen_golden_pi0 import gen_pi0
        gen_pi0(save)
    if "hf" in which:
        from gen_golden_hf import gen_hf
        gen_hf(save)
