"""CPU tests: host glue against goldens produced by the reference's adapter code; the selection driver logic; the
C-ABI library loads and exports every declared symbol; candidate sharding + gather over a 2-process gloo group."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

from cover_vla_amd import host  # noqa: E402


def test_adapter_math_matches_reference_golden():
    z = np.load(os.path.join(GOLD, "adapter_bridge.npz"))
    acts = z["actions"]
    assert np.allclose(host.postprocess_verifier(acts), z["verifier_rows"], atol=1e-12)
    assert np.allclose(host.postprocess_execution(acts), z["exec_rows"], atol=1e-12)
    e = z["euler_in"]
    assert np.allclose(host.euler2axangle_sxyz(e[:, 0], e[:, 1], e[:, 2]), z["axangle"], atol=1e-12)
    # gripper thresholds: 0.5 -> 1 (verifier: a < 0.5 ? 0 : 1), execution: a > 0.5 ? +1 : -1
    v = host.postprocess_verifier(acts[:8])[:, 6]
    assert v.tolist() == [1, 0, 1, 0, 1, 0, 1, 0]
    x = host.postprocess_execution(acts[:8])[:, 6]
    assert x.tolist() == [-1, -1, 1, -1, 1, -1, 1, -1]


def test_geometry_known_answers():
    # doctest known answers carried by the reference (INT-ACT/src/utils/geometry.py:285-289): euler2axangle(0,1.5,0,'szyx')
    # -> axis [0,1,0], theta 1.5; for 'sxyz' a pure pitch gives the same
    v = host.euler2axangle_sxyz(0.0, 1.5, 0.0)
    assert np.allclose(v, [0, 1.5, 0])
    assert np.allclose(host.euler2axangle_sxyz(0.0, 0.0, 0.0), [0, 0, 0])  # identity -> zero angle


def test_process_inputs_shapes_and_history():
    rng = np.random.default_rng(0)
    q = [rng.uniform(-1, 1, size=(5, 7)).astype(np.float32) for _ in range(4)]
    hist = [rng.normal(size=7) for _ in range(9)]
    out = host.process_inputs(q, True, hist, 4)
    assert len(out) == 5 and out[0].shape == (10, 7)           # 6 past + 4 future
    assert np.array_equal(out[3][:6], np.stack(hist[-6:]))
    assert host.process_inputs(q, True, [], 4)[0].shape == (4, 7)
    assert host.process_inputs(q, False, hist[:2], 4)[0].shape == (6, 7)


class _FakeVerifier:
    """compute_max_similarity_scores_batch with scripted scores: exercises the two-stage rule and the vote."""

    def __init__(self, stage1, scores):
        self.stage1, self.scores, self.calls = stage1, np.asarray(scores, dtype=np.float32), []

    def compute_max_similarity_scores_batch(self, images, instructions, all_action_histories, cfg_repeat_language_instructions=1):
        self.calls.append((len(images), cfg_repeat_language_instructions))
        if len(all_action_histories) == 1:
            return self.stage1, instructions[0], all_action_histories[0], torch.tensor(0)
        g = cfg_repeat_language_instructions
        s = torch.from_numpy(self.scores).view(-1, g)
        bg = int(s.mean(1).argmax())
        bi = int(s[bg].argmax())
        return float(s[bg, bi]), instructions[0], all_action_histories[bg * g + bi], torch.tensor(bg * g + bi)


def test_two_stage_verification_and_gripper_vote():
    rng = np.random.default_rng(1)
    B, S = 6, 3
    q = [rng.uniform(-1, 1, size=(B, 7)).astype(np.float32) for _ in range(4)]
    for t in range(4):
        q[t][:, 6] = [0.9, 0.1, 0.2, 0.9, 0.8, 0.1]      # group 0: 1 close-vote... (>0.5 -> +1)
    tasks = ["a"] * 3 + ["b"] * 3
    hist = [rng.normal(size=7) for _ in range(3)]
    # stage 1 confident -> candidate 0, one verifier call
    fv = _FakeVerifier(0.5, np.zeros(B))
    r = host.verify_and_select(fv, None, "a", tasks, q, hist, S, process_image=False)
    assert fv.calls == [(1, 1)] and r["global_action_idx"] == 0 and r["max_instruction"] == "a"
    # group 0 grippers (+1, -1, -1): majority open (-1) overrides the winner's own +1
    assert r["execute_action"][-1] == -1.0
    assert len(r["remaining"]) == 3 and r["remaining"][0].shape == (1, 7)
    # stage 1 below 0.1 -> stage 2 over all candidates, grouped; winner in group 1
    fv = _FakeVerifier(0.05, [0.1, 0.1, 0.1, 0.2, 0.9, 0.3])
    r = host.verify_and_select(fv, None, "a", tasks, q, hist, S, process_image=False)
    assert fv.calls == [(1, 1), (B, S)] and r["global_action_idx"] == 4 and r["max_instruction"] == "b"
    assert r["execute_action"][-1] == 1.0                  # group 1 grippers (+1, +1, -1)
    assert np.array_equal(r["remaining"][1], q[2][4:5])


# ------------------------------------------------------------------------------------------------ C ABI
def test_c_abi_library_exports_every_declared_symbol():
    from cover_vla_amd import _lib
    h = _lib.lib()                                       # loads, resolves every SYMBOLS entry, checks struct sizes
    hdr = open(os.path.join(ROOT, "include", "cover_hip.h")).read()
    declared = set(re.findall(r"\b(cover_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    missing_in_binding = declared - set(_lib.SYMBOLS)
    assert not missing_in_binding, missing_in_binding
    for name in declared:
        assert hasattr(h, name), name
    assert h.cover_abi_version() == 1
    assert h.cover_packed_k(588) == 640 and h.cover_packed_weight_bytes(16, 128) == 16 * 128 * 2


def test_no_cpu_fallback_on_host_tensors():
    from cover_vla_amd import _lib, ops
    with pytest.raises(_lib.CoverError):
        ops.layernorm(torch.zeros(2, 8, dtype=torch.bfloat16), torch.ones(8), torch.zeros(8), 1e-5)


# ------------------------------------------------------------------------------------------------ multi-process
def _worker(rank, world, port, q, n_prompts=8):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cover_vla_amd.sharding import gather_records_and_select, shard_prompts
    prompts = [f"p{i}" for i in range(n_prompts)]
    mine = shard_prompts(prompts, rank, world)
    # rank r scores its candidates with a known function of the GLOBAL candidate index; the payload of a candidate is its
    # 7 action tokens (int64, a known function of the global index too) -- what the driver needs of the winner and of the
    # winner's prompt group (gripper vote, run_simpler_eval_with_openpi.py:365-401)
    S = 4
    gidx = [prompts.index(p) * S + s for p in mine for s in range(S)]
    scores = torch.tensor([((g * 37) % 101) / 101.0 for g in gidx], dtype=torch.float32)
    tokens = torch.tensor([[31744 + (g * 13 + j * 7) % 256 for j in range(7)] for g in gidx], dtype=torch.int64).reshape(len(gidx), 7)
    res = gather_records_and_select(scores, S, rank, world, n_prompts_total=len(prompts), local_payload=tokens)
    res = {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in res.items()}
    q.put((rank, res))
    dist.destroy_process_group()


def test_candidate_sharding_gloo_world2():
    """2 ranks x 4 prompts x 4 samples: one all-gather of [score | 7 tokens] records; every rank must hold the same winner
    index, the winner's tokens and its whole prompt group's tokens (winner exchange)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    S = 4
    all_scores = torch.tensor([((g * 37) % 101) / 101.0 for g in range(32)]).view(8, S)
    bg = int(all_scores.mean(1).argmax())
    bi = int(all_scores[bg].argmax())
    tok = lambda g: [31744 + (g * 13 + j * 7) % 256 for j in range(7)]
    for r in range(2):
        assert outs[r]["global_idx"] == bg * S + bi and outs[r]["group"] == bg
        assert abs(outs[r]["max_score"] - float(all_scores[bg, bi])) < 1e-7
        assert outs[r]["winner_payload"] == tok(bg * S + bi)                       # exact: token ids < 2^24 survive fp32
        assert outs[r]["group_payload"] == [tok(bg * S + s) for s in range(S)]
        assert outs[r]["payload"] == [tok(g) for g in range(32)]
        assert np.allclose(outs[r]["scores"], all_scores.view(-1).numpy(), atol=0)
    assert outs[0] == outs[1]


def test_candidate_sharding_gloo_world2_ragged_shards():
    """7 prompt groups on 2 ranks (4 + 3): the short rank pads its records for the equal-size all-gather and the padding never
    reaches the arg-max; a rank that passes the wrong number of local scores gets a ValueError naming the rule."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, 7)) for r in range(2)]
    for p in ps:
        p.start()
    outs = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    S = 4
    all_scores = torch.tensor([((g * 37) % 101) / 101.0 for g in range(28)]).view(7, S)
    bg = int(all_scores.mean(1).argmax())
    bi = int(all_scores[bg].argmax())
    tok = lambda g: [31744 + (g * 13 + j * 7) % 256 for j in range(7)]
    for r in range(2):
        assert outs[r]["global_idx"] == bg * S + bi and outs[r]["group"] == bg
        assert outs[r]["winner_payload"] == tok(bg * S + bi)
        assert outs[r]["payload"] == [tok(g) for g in range(28)]
        assert np.allclose(outs[r]["scores"], all_scores.view(-1).numpy(), atol=0)
    assert outs[0] == outs[1]


def test_candidate_sharding_gloo_world2_idle_rank():
    """More ranks than prompt groups (1 group on 2 ranks): the idle rank contributes an empty [0, 7] payload -- its record width comes
    from the trailing dimension -- pads for the equal-size all-gather, and still ends up with the winner and its group's tokens."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, 1)) for r in range(2)]
    for p in ps:
        p.start()
    outs = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    S = 4
    all_scores = torch.tensor([((g * 37) % 101) / 101.0 for g in range(4)]).view(1, S)
    bi = int(all_scores[0].argmax())
    tok = lambda g: [31744 + (g * 13 + j * 7) % 256 for j in range(7)]
    for r in range(2):
        assert outs[r]["global_idx"] == bi and outs[r]["group"] == 0
        assert outs[r]["winner_payload"] == tok(bi)
        assert outs[r]["payload"] == [tok(g) for g in range(4)]
    assert outs[0] == outs[1]


def test_gather_records_world1_fp32_payload():
    from cover_vla_amd.sharding import gather_records_and_select
    g = torch.Generator().manual_seed(0)
    scores = torch.rand(16, generator=g)
    chunk = torch.randn(16, 4, 7, generator=g)
    r = gather_records_and_select(scores, 2, 0, 1, 8, local_payload=chunk)
    gm = scores.view(8, 2).mean(1)
    bg = int(gm.argmax()); bi = int(scores.view(8, 2)[bg].argmax())
    assert r["global_idx"] == bg * 2 + bi
    assert torch.equal(r["winner_payload"], chunk[bg * 2 + bi].reshape(-1))
    assert torch.equal(r["group_payload"], chunk[bg * 2:bg * 2 + 2].reshape(2, -1))


# ------------------------------------------------------------------------------------------------ checkpoint format
def test_verifier_checkpoint_converts_once_and_loads_without_pickle(tmp_path):
    """the merged verifier .pt (efficient_ensemble_merged.py:37-53) -> safetensors + JSON once, then loaded with no unpickling;
    the .pt itself is only ever read through the restricted unpickler. Both routes give back the reference's dict, tensor for tensor,
    for the transformer and the MLP action encoder and with the optional top-level fields."""
    from cover_vla_amd import loaders, synth
    for kw in (dict(), dict(use_transformer=False)):
        try:
            ck = synth.verifier_checkpoint(2, seed=3, **kw)
        except TypeError:
            if kw:
                continue
            raise
        ck = dict(ck, backbone="hf-hub:timm/ViT-L-16-SigLIP2-384", use_transformer=not kw, history_length=10, action_dim=7, num_models=2)
        pt = tmp_path / f"merged{len(kw)}.pt"
        torch.save(ck, str(pt))
        out = loaders.verifier_pt_to_safetensors(str(pt), str(tmp_path / f"conv{len(kw)}"))
        for got in (loaders.load_verifier_checkpoint(out), loaders.load_verifier_checkpoint(str(pt))):
            assert {k: v for k, v in got.items() if k != "ensemble_components"} == {k: v for k, v in ck.items() if k != "ensemble_components"}
            assert len(got["ensemble_components"]) == 2
            for a, b in zip(got["ensemble_components"], ck["ensemble_components"]):
                assert set(a) == set(b)
                for name, val in b.items():
                    if isinstance(val, dict):
                        assert set(a[name]) == set(val)
                        assert all(torch.equal(a[name][kk], torch.as_tensor(val[kk])) for kk in val)
                    elif torch.is_tensor(val):
                        assert torch.equal(a[name], val)
                    else:
                        assert a[name] == val
    # a pickle with code in it is refused, with the conversion named in the message
    import pickle

    class Evil:
        def __reduce__(self):
            return (print, ("unpickled",))
    bad = tmp_path / "bad.pt"
    with open(bad, "wb") as f:
        pickle.dump({"ensemble_components": [Evil()]}, f)
    with pytest.raises(ValueError, match="verifier_pt_to_safetensors"):
        loaders.load_verifier_checkpoint(str(bad))



def test_pi0_checkpoint_key_layout_round_trip(tmp_path):
    """neutral -> the reference's safetensors key layout (convert_pi0_to_hf_lerobot.py:67-245,384-390) -> neutral."""
    import json
    from safetensors.torch import save_file
    from cover_vla_amd import loaders, synth
    tiny = dict(lm_dim=64, lm_mlp=128, ex_dim=32, ex_mlp=64, layers=2, Hq=4, Hkv=1, D=16, vocab=96, vit_dim=48, vit_mlp=80,
                vit_layers=2, vit_heads=4, patch=14, image=56, chunk=4)
    sd = synth.pi0_state(tiny, seed=3)
    ref = loaders.neutral_to_pi0_reference(sd, tiny["patch"])
    # spot-check names against the reference's converter
    for k in ("model.paligemma_with_expert.paligemma.vision_tower.vision_model.embeddings.patch_embedding.weight",
              "model.paligemma_with_expert.paligemma.vision_tower.vision_model.encoder.layers.1.self_attn.out_proj.bias",
              "model.paligemma_with_expert.paligemma.multi_modal_projector.linear.weight",
              "model.paligemma_with_expert.paligemma.language_model.model.layers.0.mlp.gate_proj.weight",
              "model.paligemma_with_expert.gemma_expert.model.norm.weight", "model.action_time_mlp_in.bias"):
        assert k in ref, k
    assert ref["model.paligemma_with_expert.paligemma.vision_tower.vision_model.embeddings.patch_embedding.weight"].shape == (48, 3, 14, 14)
    # extras a real checkpoint carries and the loader must ignore
    ref["model.paligemma_with_expert.paligemma.language_model.lm_head.weight"] = sd["lm.embed_tokens.weight"].clone()
    ref["model.paligemma_with_expert.gemma_expert.lm_head.weight"] = torch.zeros(4, 4)
    d = tmp_path / "ckpt"
    d.mkdir()
    save_file({k: v.contiguous() for k, v in ref.items()}, str(d / "model.safetensors"))
    (d / "config.json").write_text(json.dumps({"chunk_size": 4, "n_action_steps": 4, "tokenizer_max_length": 72, "num_steps": 10}))
    n, c, cfg = loaders.load_pi0_pretrained(str(d), head_dim=16, vit_heads=4)
    assert set(n) == set(sd)
    for k in sd:
        assert torch.equal(n[k], sd[k]), k
    for k in ("lm_dim", "lm_mlp", "ex_dim", "ex_mlp", "layers", "Hq", "Hkv", "D", "vocab", "vit_dim", "vit_mlp", "vit_layers", "patch", "image", "chunk"):
        assert c[k] == tiny[k], (k, c[k], tiny[k])


def test_pi0_normalization_buffers_from_checkpoint_keys():
    """normalize.py:44-107 buffer names; IDENTITY needs no buffers, MEAN_STD / MIN_MAX need finite ones."""
    from cover_vla_amd import loaders
    sd = {"normalize_inputs.buffer_observation_state.mean": torch.zeros(7), "normalize_inputs.buffer_observation_state.std": torch.ones(7),
          "model.unnormalize_outputs.buffer_action.mean": torch.ones(7), "model.unnormalize_outputs.buffer_action.std": torch.full((7,), 2.0)}
    n = loaders.pi0_normalization(sd, {"normalization_mapping": {"STATE": "MEAN_STD", "ACTION": "NormalizationMode.MEAN_STD", "VISUAL": "IDENTITY"}})
    assert n["state"][0] == "MEAN_STD" and n["action"][0] == "MEAN_STD" and float(n["action"][2][0]) == 2.0
    assert loaders.pi0_normalization({}, {})["state"][0] == "IDENTITY"
    with pytest.raises(ValueError):
        loaders.pi0_normalization({}, {"normalization_mapping": {"STATE": "MIN_MAX"}})
    sd["normalize_inputs.buffer_observation_state.std"] = torch.full((7,), float("inf"))
    with pytest.raises(ValueError):
        loaders.pi0_normalization(sd, {"normalization_mapping": {"STATE": "MEAN_STD"}})


def test_checkpoint_key_names_match_the_reference_converter():
    """the names loaders.neutral_to_pi0_reference writes / pi0_reference_to_neutral reads are exactly the names the
    reference's converter assigns (fixture: oracle/gen_golden_keys.py from convert_pi0_to_hf_lerobot.py:67-245,384-386), at the
    real layer counts (27 SigLIP blocks, 18 + 18 decoder layers)."""
    import json
    from cover_vla_amd import loaders
    with open(os.path.join(GOLD, "pi0_checkpoint_keys.json")) as f:
        fx = json.load(f)
    ref_keys = set(fx["keys"])
    # a structural state dict at the real layer counts with 1-element tensors (names only)
    n = {}
    for i in range(27):
        for nm in ("ln1", "ln2", "q", "k", "v", "o", "fc1", "fc2"):
            for wb in ("weight", "bias"):
                n[f"vision.blocks.{i}.{nm}.{wb}"] = torch.zeros(1)
    n.update({"vision.patch.weight": torch.zeros(1, 3 * 14 * 14), "vision.patch.bias": torch.zeros(1), "vision.pos": torch.zeros(1, 1),
              "vision.post_ln.weight": torch.zeros(1), "vision.post_ln.bias": torch.zeros(1), "projector.weight": torch.zeros(1),
              "projector.bias": torch.zeros(1), "lm.embed_tokens.weight": torch.zeros(1), "lm.norm.weight": torch.zeros(1),
              "expert.norm.weight": torch.zeros(1)})
    for pre in ("lm", "expert"):
        for i in range(18):
            for nm in ("input_layernorm", "post_attention_layernorm", "self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj",
                       "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"):
                n[f"{pre}.layers.{i}.{nm}.weight"] = torch.zeros(1)
    for nm in ("state_proj", "action_in_proj", "action_out_proj", "action_time_mlp_in", "action_time_mlp_out"):
        n[nm + ".weight"], n[nm + ".bias"] = torch.zeros(1), torch.zeros(1)
    ours = set(loaders.neutral_to_pi0_reference(n, 14).keys())
    # the converter also writes tied / dummy tensors our neutral layout does not carry: lm_head (tied to embed_tokens), the expert's
    # zero embedding and zero lm_head
    extras = {"model.paligemma_with_expert.paligemma.language_model.lm_head.weight",
              "model.paligemma_with_expert.gemma_expert.model.embed_tokens.weight", "model.paligemma_with_expert.gemma_expert.lm_head.weight"}
    assert ours <= ref_keys, sorted(ours - ref_keys)[:5]
    assert ref_keys - ours == extras, sorted(ref_keys - ours - extras)[:5]
    # and reading a dict with exactly the reference's names (extras included) gives back the neutral names
    back = loaders.pi0_reference_to_neutral({k: torch.zeros(1, 3, 14, 14) if k.endswith("patch_embedding.weight") else torch.zeros(1) for k in ref_keys})
    assert set(back.keys()) == set(n.keys())


def test_episode_log_schema_matches_the_driver(tmp_path):
    import json
    import pickle
    with open(os.path.join(GOLD, "pi0_checkpoint_keys.json")) as f:
        fields = json.load(f)["episode_fields"]
    log = host.EpisodeLog("put the spoon on the towel", "place spoon on towel")
    assert list(log.data.keys()) == fields == list(host.EpisodeLog.FIELDS)
    log.record_decision(0.31, "place spoon on towel", np.arange(7.0), 0)
    log.record_queued("place spoon on towel", np.arange(7.0) + 1, 1)
    d = log.finish(True, 2)
    assert d["verifier_scores"] == [0.31, None] and d["step_timestamps"] == [0, 1] and d["success"] is True and d["episode_length"] == 2
    log.save(str(tmp_path / "ep.pkl"))
    with open(tmp_path / "ep.pkl", "rb") as f:
        assert list(pickle.load(f).keys()) == fields


def test_pi0fast_dct_decode_matches_reference_golden():
    from cover_vla_amd.pi0fast import fast_coefficients_to_actions, fast_tokens_to_paligemma_tokens
    z = np.load(os.path.join(GOLD, "pi0fast_dct_decode.npz"))
    seqs = [z["seq0"].tolist(), z["seq1"].tolist(), z["seq2"].tolist()]
    out = fast_coefficients_to_actions(seqs, lambda t: "".join(chr(i) for i in t), min_token=int(z["min_token"]), scale=float(z["scale"]),
                                       time_horizon=4, action_dim=7)
    assert np.allclose(out, z["actions"], atol=1e-12)
    bad = fast_coefficients_to_actions([[1, 2, 3]], lambda t: (_ for _ in ()).throw(ValueError("no tokenizer")), min_token=0, scale=10.0,
                                       time_horizon=4, action_dim=7)
    assert bad.shape == (1, 4, 7) and not bad.any()                                  # undecodable sequence -> zeros, as the reference
    t = np.array([0, 5, 1000])
    assert np.array_equal(fast_tokens_to_paligemma_tokens(fast_tokens_to_paligemma_tokens(t, 257152), 257152), t)
    assert fast_tokens_to_paligemma_tokens(np.array([0]), 257152)[0] == 257152 - 1 - 128


def test_pi0fast_checkpoint_key_map_round_trip(tmp_path):
    """SURVEY 8f-3 for the pi0-FAST policy: `model.pi0_paligemma.*` = the PaliGemma module pi0 keeps under
    `model.paligemma_with_expert.paligemma.` (names pinned by tests/golden/pi0_checkpoint_keys.json, the reference converter's
    list); neutral -> reference names -> safetensors on disk -> loader -> the same tensors and sizes."""
    import json
    from safetensors.torch import save_file
    from cover_vla_amd import loaders, synth
    tiny = dict(lm_dim=64, lm_mlp=128, ex_dim=32, ex_mlp=64, layers=2, Hq=4, Hkv=1, D=16, vocab=40, vit_dim=32, vit_mlp=48, vit_layers=2,
                vit_heads=2, patch=14, image=28, chunk=4)
    sd = synth.pi0_state(tiny, seed=3)
    ref = loaders.neutral_to_pi0fast_reference(sd, tiny["patch"])
    with open(os.path.join(GOLD, "pi0_checkpoint_keys.json")) as f:
        pinned = json.load(f)
    pinned = pinned["keys"] if isinstance(pinned, dict) else pinned
    pg = "model.paligemma_with_expert.paligemma."
    suffixes = {k[len(pg):] for k in pinned if k.startswith(pg)}
    # per-layer names of the pinned list are templated on the layer index: compare with the index normalised
    norm = lambda k: re.sub(r"layers\.\d+\.", "layers.N.", k)
    ours = {norm(k[len("model.pi0_paligemma."):]) for k in ref}
    assert ours <= {norm(s_) for s_ in suffixes} | {"language_model.lm_head.weight"}, sorted(ours - {norm(s_) for s_ in suffixes})[:5]
    d = tmp_path / "fast"
    d.mkdir()
    save_file({k: v.contiguous().clone() for k, v in ref.items()}, str(d / "model.safetensors"))   # (clone: the tied lm_head shares storage)
    (d / "config.json").write_text(json.dumps({"type": "pi0fast", "normalization_mapping": {"VISUAL": "IDENTITY", "STATE": "IDENTITY", "ACTION": "IDENTITY"}}))
    n, c, cfg = loaders.load_pi0fast_pretrained(str(d), head_dim=tiny["D"], vit_heads=tiny["vit_heads"])
    for k, v in sd.items():
        if k.startswith(("vision.", "projector.", "lm.")):
            assert torch.equal(n[k].reshape(v.shape), v), k
    assert all(not k.startswith("expert.") for k in n)
    for k in ("lm_dim", "lm_mlp", "layers", "Hq", "Hkv", "D", "vocab", "vit_dim", "vit_mlp", "vit_layers", "patch", "image"):
        assert c[k] == tiny[k], (k, c[k], tiny[k])
    with pytest.raises(ValueError):
        bad = tmp_path / "bad"
        bad.mkdir()
        save_file({"model.something_else.weight": torch.zeros(2)}, str(bad / "model.safetensors"))
        (bad / "config.json").write_text("{}")
        loaders.load_pi0fast_pretrained(str(bad))


def test_pi0fast_policy_host_glue_matches_reference_golden():
    """PI0FASTPolicy's host side against the reference's own create_input_tokens / extract_actions (run by
    oracle/gen_golden_pi0fast.py with the same stand-in tokenizers): prompt text + 256-bin state discretisation -> ids and mask,
    generated ids -> cleaned text -> FAST ids -> DCT -> action chunk; then select_action's queue on a scripted token generator."""
    from cover_vla_amd import synth
    from cover_vla_amd.pi0fast import PI0FASTConfig, PI0FASTPolicy
    import types
    z = np.load(os.path.join(GOLD, "pi0fast_host_glue.npz"))
    tok = synth.CharTokenizer(vocab_size=512)
    fast = types.SimpleNamespace(bpe_tokenizer=types.SimpleNamespace(decode=lambda t: "".join(chr(max(0, i)) for i in t)), min_token=-40, scale=10.0)
    H, A = int(z["horizon"]), int(z["action_dim"])

    class _Scripted:                       # stands in for PI0FASTTokens: returns the golden's generated ids
        dev = torch.device("cpu")
        calls = 0

        def generate_tokens(self, images, img_masks, ids, mask, n_new, eos_token_id=1, pad_token_id=0):
            _Scripted.calls += 1
            assert ids.shape == mask.shape and images[0].shape[0] == ids.shape[0]
            return torch.from_numpy(z["gen_tokens"])

    cfg = PI0FASTConfig(action_dim=A, chunk_size=H, n_action_steps=3, resize_imgs_with_padding=None, device="cpu")
    pol = PI0FASTPolicy(cfg, _Scripted(), tok, fast)
    state = torch.from_numpy(z["state"])
    ids, mask = pol.create_input_tokens(state, [str(t) for t in z["tasks"]])
    assert torch.equal(ids, torch.from_numpy(z["input_ids"])) and torch.equal(mask, torch.from_numpy(z["padded_mask"]))
    assert not z["att_mask"].any()                                         # generation: every prompt token is prefix (bidirectional)
    acts = pol.extract_actions(torch.from_numpy(z["gen_tokens"]), H, A)
    assert acts.dtype == torch.float64 and np.allclose(acts.numpy(), z["actions"], atol=1e-12)
    # a left-padding tokenizer yields the same compacted rows
    pol_l = PI0FASTPolicy(cfg, _Scripted(), synth.CharTokenizer(vocab_size=512, padding_side="left"), fast)
    ids_l, mask_l = pol_l.create_input_tokens(state, [str(t) for t in z["tasks"]])
    assert torch.equal(ids_l, ids) and torch.equal(mask_l, mask)
    # select_action: one generation fills the queue with n_action_steps rows of [B, action_dim]
    batch = {"observation.state": state, "observation.images.top": torch.zeros(4, 3, 8, 8), "task": [str(t) for t in z["tasks"]]}
    a0 = pol.select_action(batch)
    a1 = pol.select_action(batch)
    a2 = pol.select_action(batch)
    assert _Scripted.calls == 1 and tuple(a0.shape) == (4, A)
    assert np.allclose(torch.stack([a0, a1, a2], 1).numpy(), z["actions"][:, :3, :A].astype(np.float32), atol=1e-6)
    pol.select_action(batch)
    assert _Scripted.calls == 2
    with pytest.raises(ValueError):
        pol.reset()
        pol.select_action({"observation.state": state, "task": ["x"] * 4})


def test_sampled_pick_decided_is_the_exact_worst_case():
    """bench.sampled_pick_decided (the criterion of cpu_baseline.agreement): a pick it calls decided survives every logit perturbation within
    +-err (random ones AND the two extremal ones that push the bin's edges furthest), and a pick it calls undecided is flipped by one of the
    two extremal perturbations -- i.e. the bound is tight, not merely sufficient."""
    import torch
    import bench
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from cover_ref import openvla as OR
    g = torch.Generator().manual_seed(5)
    n_dec = n_und = 0
    for case in range(400):
        nb = 32
        l = (torch.randn(nb, generator=g) * (0.5 + 3.0 * torch.rand(1, generator=g))).double()
        u = float(torch.rand(1, generator=g))
        err = float(10 ** (-3 + 2.5 * torch.rand(1, generator=g)))
        t = OR.select_token(l.float(), 0, nb, u, 1.0)
        dec = bool(bench.sampled_pick_decided(l[None], torch.tensor([t]), torch.tensor([u]).double(), torch.tensor([err]).double())[0])
        idx = torch.arange(nb)
        extremal = [torch.where(idx < t, err, -err), torch.where(idx <= t, -err, err)]     # lower edge up / upper edge down
        picks = [OR.select_token((l + d).float(), 0, nb, u, 1.0) for d in extremal]
        if dec:
            n_dec += 1
            assert all(p == t for p in picks), (case, t, picks, u, err)
            for _ in range(20):
                d = (torch.rand(nb, generator=g).double() * 2 - 1) * err
                assert OR.select_token((l + d).float(), 0, nb, u, 1.0) == t
        else:
            # the margin the criterion leaves to fp32 rounding (eps) is the only room for an undecided pick that no extremal perturbation flips
            lo_edge = float(torch.softmax(l, 0)[:t].sum())
            hi_edge = float(torch.softmax(l, 0)[: t + 1].sum())
            near = min(abs(u - lo_edge), abs(hi_edge - u))
            flipped = any(p != t for p in picks)
            n_und += int(flipped)
            assert flipped or near < 0.05 + 2.5 * err, (case, t, picks, u, err, near)
    assert n_dec > 50 and n_und > 50, (n_dec, n_und)


def test_oracle_mx_block_quantiser_follows_its_rule():
    """oracle/cover_ref/blocks.py::fake_quant_blocks_e4m3 (the restatement of cover_quantize_act_fp8_mx the config-5 oracle uses at the inputs of o_proj / down_proj): one
    power-of-two scale per 32 consecutive elements -- the smallest 2^e with amax / 2^e <= 448, 2^0 for an all-zero block -- RNE to e4m3, de-quantised. Checked element by
    element against a loop over the blocks, on a block amax exactly at a power-of-two boundary, an all-zero block, and blocks five decades apart inside one row (where one scale
    per row -- fake_quant_rows_e4m3 -- flushes the small block to the subnormal grid and the block scales do not)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from cover_ref import blocks as Bk
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 128, generator=g)
    x[0, :32] *= 1e-5                       # a block five decades below its neighbours
    x[1, 32:64] = 0.0                       # an all-zero block
    x[2, 64] = 448.0 * 2.0 ** -5            # amax / 448 exactly 2^-5 -> scale 2^-5, the element maps to 448
    x[2, 65:96] = x[2, 65:96].clamp(-10, 10)
    x = x.bfloat16()
    got = Bk.fake_quant_blocks_e4m3(x).float()
    want = torch.empty_like(got)
    for m in range(3):
        for b in range(4):
            blk = x[m, 32 * b: 32 * b + 32].float()
            amax = float(blk.abs().max())
            e = 0 if amax == 0 else int(np.ceil(np.log2(amax / 448.0)))
            s = 2.0 ** e
            assert amax / s <= 448.0 and (amax == 0 or amax / (s / 2) > 448.0)
            want[m, 32 * b: 32 * b + 32] = (blk / s).to(torch.float8_e4m3fn).float() * s
    assert torch.equal(got, want)
    assert float(got[2, 64]) == float(x[2, 64])                       # 448 x 2^-5 is exactly representable
    assert torch.equal(got[1, 32:64], torch.zeros(32))
    # the small block keeps e4m3's relative precision under block scales, and loses it under one scale per row
    small = x[0, :32].float()
    rel_blocks = ((got[0, :32] - small).norm() / small.norm()).item()
    rel_rows = ((Bk.fake_quant_rows_e4m3(x)[0, :32].float() - small).norm() / small.norm()).item()
    assert rel_blocks < 0.04 and rel_rows > 0.2, (rel_blocks, rel_rows)
