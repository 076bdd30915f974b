"""Episode-level parity (SURVEY 8 row a19 + the prompt drift of Appendix A.5): a scripted five-decision CoVer episode
(tests/episode_replay.py: the evaluation driver's loop over a stub environment) replayed once through the CPU oracle's classes and once
through the HIP classes -- same checkpoints, same frames, same noise. Decisions 2 and 4 force stage 2, so the instruction is replaced by a
rephrase twice and later decisions run on the drifted prompt set."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from cover_vla_amd import host, synth  # noqa: E402


def test_scripted_episode_oracle_and_hip_classes_agree(dev, tmp_path):
    from cover_vla_amd.imaging import siglip_preprocess
    from cover_vla_amd.pi0 import PI0Config, PI0FlowMatching, PI0Policy
    from cover_vla_amd.verifier import EfficientEnsembleMerged, SigLIP2Encoder
    from tests.episode_replay import OraclePI0Policy, OracleVerifier, WordTokenizer, run_episode, scripted_inputs
    from tests.helpers import pi0_case
    _, tiny, sd, _ = pi0_case(os.path.join(ROOT, "tests", "golden", "pi0_tiny_b6.npz"))
    sc = dict(synth.SIGLIP2_SMALL)
    ssd = synth.siglip2_state(sc, seed=77)
    n_patches = (sc["image"] // sc["patch"]) ** 2
    ck = synth.verifier_checkpoint(2, seed=77, num_patches=n_patches, vision_dim=sc["dim"], text_dim=sc["dim"])
    R, S, L, n_dec = 3, 2, 12, 5
    B = R * S
    ptok, vtok = WordTokenizer(tiny["vocab"]), WordTokenizer(sc["vocab"])
    pre = lambda im: siglip_preprocess(im, sc["image"])
    task = "put the spoon on the towel"
    rephrases = ["place the spoon onto the towel", "move the spoon to the cloth", "set the spoon down on the towel"]
    inputs = scripted_inputs(n_dec, B, tiny["chunk"], tiny["image"], seed=36)   # (a seed whose stage-2 decisions have score gaps well above the bf16 noise)
    stage2 = [False, True, False, True, False]

    # ---- the oracle's classes (CPU)
    o_pol = OraclePI0Policy(sd, tiny, ptok.policy, n_action_steps=4, max_lang=L)
    o_ver = OracleVerifier(ck, sc, ssd, pre, vtok.verifier)
    o_rec, o_tr = run_episode(o_pol, o_ver, inputs, task, rephrases, R, S, stage2, device="cpu")

    # ---- the HIP classes
    model = PI0FlowMatching(sd, tiny, device="cuda:0", max_batch=8, max_prompts=8, max_lang=L)
    h_pol = PI0Policy(PI0Config(n_action_steps=4, chunk_size=tiny["chunk"], tokenizer_max_length=L, resize_imgs_with_padding=None), model, ptok.policy)
    enc = SigLIP2Encoder(ssd, dim=sc["dim"], layers=sc["layers"], heads=sc["heads"], mlp=sc["mlp"], patch=sc["patch"], image=sc["image"],
                         context_length=sc["context_length"], device="cuda:0")
    h_ver = EfficientEnsembleMerged(ck, device="cuda:0", encoder=enc, preprocess=pre, tokenizer=vtok.verifier)
    h_rec, h_tr = run_episode(h_pol, h_ver, inputs, task, rephrases, R, S, stage2, device="cuda:0")

    # ---- the record: schema of run_simpler_eval_with_openpi.py:238-247, one entry per environment step
    assert set(h_rec) == set(host.EpisodeLog.FIELDS) and h_rec["episode_length"] == 4 * n_dec
    assert h_rec["step_timestamps"] == list(range(4 * n_dec)) == o_rec["step_timestamps"]
    assert [s is not None for s in h_rec["verifier_scores"]] == [t % 4 == 0 for t in range(4 * n_dec)]
    assert all(np.asarray(a).shape == (7,) and a[-1] in (-1.0, 1.0) for a in h_rec["execute_actions"])
    # ---- decisions: identical winner index, identical instruction sequence (prompt drift included), executed actions within the bf16 policy's noise
    for d, (o, h) in enumerate(zip(o_tr, h_tr)):
        assert o["prompts"] == h["prompts"], d                         # the drifted prompt set the policy saw
        upd = np.linalg.norm(o["actions"] - inputs[d]["noise"][:, :4, :7].numpy())
        assert np.linalg.norm(h["actions"] - o["actions"]) / upd < 3e-2, d
        assert abs(o["max_score"] - h["max_score"]) < 2e-2, (d, o["max_score"], h["max_score"])
        if o["stage2"]:                                                # the decision is data-decided when the oracle's score gaps exceed the difference
            sc_o = np.sort(o["scores"].reshape(R, S).mean(1))[::-1]
            in_g = np.sort(o["scores"].reshape(R, S)[o["global_action_idx"] // S])[::-1]
            assert sc_o[0] - sc_o[1] > 4e-3 and in_g[0] - in_g[1] > 4e-3, ("scripted case must be decided", d, sc_o, in_g)
        assert o["global_action_idx"] == h["global_action_idx"], (d, o["global_action_idx"], h["global_action_idx"])
    assert o_rec["selected_instructions"] == h_rec["selected_instructions"]
    assert len(set(h_rec["selected_instructions"])) == 3               # the instruction was replaced by a rephrase, twice
    n_grip = 0
    for t, (a, b) in enumerate(zip(o_rec["execute_actions"], h_rec["execute_actions"])):
        # gripper command = a vote over the winner's prompt group of 2 (a > 0.5) - 1 (decision steps) / the winner's own (queued steps): exact
        # wherever no voter's raw gripper output sits within the policies' bf16 difference of the 0.5 threshold
        d, k = divmod(t, 4)
        gi = o_tr[d]["global_action_idx"]
        voters = o_tr[d]["actions"][(gi // S) * S:(gi // S + 1) * S, 0, 6] if k == 0 else o_tr[d]["actions"][gi:gi + 1, k, 6]
        if np.abs(voters - 0.5).min() > 0.03:
            assert a[-1] == b[-1], (t, voters)
            n_grip += 1
        assert np.allclose(a[:6], b[:6], atol=2e-2 * max(1.0, float(np.abs(a[:6]).max()))), (t, a, b)
    assert n_grip >= 12, n_grip
    # ---- the pickle the analysis scripts read
    log = host.EpisodeLog(task, task)
    log.data = h_rec
    path = str(tmp_path / "episode.pkl")
    log.save(path)
    with open(path, "rb") as f:
        back = pickle.load(f)
    assert set(back) == set(host.EpisodeLog.FIELDS) and back["original_task_description"] == task and back["success"] is False
    assert back["selected_instructions"] == h_rec["selected_instructions"]
