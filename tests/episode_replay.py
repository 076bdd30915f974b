"""Scripted CoVer episode, replayed through any (policy, verifier) pair that has the reference's two call surfaces:
`policy.select_action(batch, noise=...) -> deque of n_action_steps x [B, 7]` (modeling_pi0.py:263-307) and
`verifier.compute_max_similarity_scores_batch(images, instructions, all_action_histories, cfg_repeat_language_instructions)`
(efficient_ensemble_merged.py:309-454). The loop is the evaluation driver's (run_simpler_eval_with_openpi.py:225-455) with the
simulator replaced by a stub environment: fixed seeded camera frames and robot states per step, no physics. What it keeps:

  * the batch construction of :296-319 -- `lang_rephrase_num` prompts x `policy_batch_inference_size` repeats of one observation;
  * a policy call every n_action_steps steps (:322-326) and the two-stage verification + gripper vote behind it (:329-401,
    cover_vla_amd.host.verify_and_select), with the stage-2 trigger SCRIPTED per decision (the `max_score < 0.1` rule of :355 needs a
    trained verifier to mean anything: here a decision either forces stage 2 or forbids it);
  * prompt drift: after a stage-2 decision the current instruction IS the winning rephrase (:409) and the next decision's prompt set is
    [current] + rephrases[:R-1] (:299-302);
  * the queued steps in between (:411-422), the verifier-format action history (:424-432) and the per-episode record (:238-247,
    cover_vla_amd.host.EpisodeLog).

tests/test_episode_gpu.py runs it once with the CPU oracle's classes (below) and once with the HIP classes and compares the two records.
Test infrastructure: the oracle adapters import oracle/cover_ref."""
from __future__ import annotations

import collections
import os
import sys
from typing import Callable, List, Sequence

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from cover_vla_amd import host  # noqa: E402


# ------------------------------------------------------------------------------------------------ stand-ins for un-vendored tokenizers
class WordTokenizer:
    """Deterministic text -> ids for BOTH towers of the test (the HF PaliGemma tokenizer and open_clip's SigLIP2 tokenizer are downloads):
    id = 2 + (stable hash of the word) mod (vocab - 2), right padded with 0. Policy form: (ids [B, L], mask [B, L]); verifier form:
    ids [B, context_length]."""

    def __init__(self, vocab: int):
        self.vocab = vocab

    def _ids(self, text: str) -> List[int]:
        out = []
        for w in text.replace("\n", " \n ").split(" "):
            if w:
                h = 0
                for ch in w:
                    h = (h * 131 + ord(ch)) % 1000003
                out.append(2 + h % (self.vocab - 2))
        return out

    def policy(self, texts: Sequence[str], max_length: int):
        ids = torch.zeros(len(texts), max_length, dtype=torch.long)
        mask = torch.zeros(len(texts), max_length, dtype=torch.bool)
        for i, t in enumerate(texts):
            v = self._ids(t)[:max_length]
            ids[i, :len(v)] = torch.tensor(v)
            mask[i, :len(v)] = True
        return ids, mask

    def verifier(self, texts: Sequence[str], context_length: int):
        return self.policy(texts, context_length)[0]


# ------------------------------------------------------------------------------------------------ the oracle's classes
class OraclePI0Policy:
    """PI0Policy.select_action on the CPU oracle (cover_ref.pi0.sample_actions): same batch keys, same deque contract."""

    def __init__(self, sd, tiny, tokenizer: Callable, n_action_steps=4, max_lang=12, action_dim=7, max_state_dim=32):
        from cover_ref import pi0 as PR
        from tests.helpers import _pi0_cfg
        self.cfg, self.sd = _pi0_cfg(tiny), PR.cast_like_reference(sd)
        self.tokenizer, self.n_action_steps, self.max_lang, self.action_dim, self.max_state_dim = tokenizer, n_action_steps, max_lang, action_dim, max_state_dim
        self._q = collections.deque([], maxlen=n_action_steps)

    def reset(self):
        self._q.clear()

    @torch.no_grad()
    def select_action(self, batch, noise=None, noise_std=1.0):
        from cover_ref import pi0 as PR
        if len(self._q) == 0:
            img = batch["observation.images.top"].cpu().float()
            B = img.shape[0]
            st = batch["observation.state"].cpu().float()
            state = torch.zeros(B, self.max_state_dim)
            state[:, : st.shape[1]] = st
            tasks = [t if t.endswith("\n") else f"{t}\n" for t in batch["task"]]
            toks, masks = self.tokenizer(tasks, self.max_lang)
            x = PR.sample_actions(self.cfg, self.sd, [img], [torch.ones(B, dtype=torch.bool)], toks, masks.bool(), state, noise.cpu().float())
            self._q.extend(x[:, : self.n_action_steps, : self.action_dim].transpose(0, 1))
        return self._q


class OracleVerifier:
    """EfficientEnsembleMerged.compute_max_similarity_scores_batch on the CPU oracle: the same preprocess + tokenizer callables, the
    oracle's SigLIP2 towers (cover_ref.openvla.siglip2_features) and heads (cover_ref.verifier), the reference's 4-tuple."""

    def __init__(self, ck, sc, ssd, preprocess: Callable, tokenizer: Callable):
        from cover_ref import blocks as Bk
        self.comps, self.sc, self.ssd = ck["ensemble_components"], sc, Bk.to_bf16(ssd)
        self.preprocess, self.tokenizer = preprocess, tokenizer

    @torch.no_grad()
    def compute_max_similarity_scores_batch(self, images, instructions, all_action_histories, cfg_repeat_language_instructions=1):
        from cover_ref import openvla as OR, verifier as V
        g = cfg_repeat_language_instructions
        img = self.preprocess(images[0]).unsqueeze(0)
        toks = self.tokenizer([instructions[0]], context_length=self.sc["context_length"])
        pf, tf = OR.siglip2_features(self.sc, self.ssd, img, toks)
        r = V.compute_max_similarity_scores(self.comps, pf, tf, list(all_action_histories), g)
        gidx = int(r["global_idx"])
        all_same = len(set(instructions)) == 1
        max_instruction = instructions[0] if (all_same and len(images) > 1) else instructions[min((gidx // g) * g, len(instructions) - 1)]
        self.last_scores = r["scores"].numpy().copy()
        return float(r["max_score"]), max_instruction, all_action_histories[gidx], torch.tensor(gidx, dtype=torch.int64)


# ------------------------------------------------------------------------------------------------ the scripted episode
def scripted_inputs(n_decisions: int, B: int, chunk: int, image: int, raw_hw=(96, 128), seed=11):
    """Fixed observations of the stub environment: per decision a raw camera frame (uint8 HWC, what process_raw_image_to_jpg gets), the
    policy's normalised frame (one image, repeated over the batch by the driver), a robot state, and the policy's noise [B, chunk, 32]."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_decisions):
        raw = torch.randint(0, 256, (raw_hw[0], raw_hw[1], 3), generator=g, dtype=torch.uint8).numpy()
        frame = torch.rand(1, 3, image, image, generator=g) * 2 - 1
        state = torch.randn(1, 7, generator=g) * 0.3
        noise = torch.randn(B, chunk, 32, generator=g)
        out.append(dict(raw=raw, frame=frame, state=state, noise=noise))
    return out


def run_episode(policy, verifier, inputs, task: str, rephrases: Sequence[str], R: int, S: int, stage2: Sequence[bool], device="cpu",
                n_action_steps=4):
    """Returns (EpisodeLog data, trace) -- trace = per decision: global_action_idx, the prompt list the policy saw, the verifier's score."""
    log = host.EpisodeLog(task, task)
    task_description = task
    action_history: List[np.ndarray] = []
    action_queue = collections.deque()
    trace = []
    t = 0
    policy.reset()
    for d, obs in enumerate(inputs):
        for _ in range(n_action_steps):
            if t % n_action_steps == 0:
                unique = [task_description] + list(rephrases[: R - 1]) if R > 1 else [task_description]     # :299-302
                task_list = [p for p in unique for _ in range(S)]
                B = len(task_list)
                batch = {"observation.images.top": obs["frame"].repeat(B, 1, 1, 1).to(device), "observation.state": obs["state"].repeat(B, 1).to(device),
                         "task": task_list}
                q = policy.select_action(batch, noise=obs["noise"].to(device))
                predefined = [a.detach().float().cpu().numpy() for a in q.copy()]                            # :324-326
                q.clear()
                r = host.verify_and_select(verifier, obs["raw"], task_description, task_list, predefined, list(action_history), S, n_action_steps,
                                           threshold=(float("inf") if stage2[d] else float("-inf")))
                execute_action = r["execute_action"]
                action_queue = r["remaining"]
                log.record_decision(r["max_score"], r["max_instruction"], execute_action, t)
                trace.append(dict(t=t, global_action_idx=r["global_action_idx"], prompts=unique, max_score=r["max_score"], stage2=bool(stage2[d]),
                                  actions=np.stack(predefined, 1), scores=getattr(verifier, "last_scores", None)))
                task_description = r["max_instruction"]                                                       # :409
                action_history.append(r["history_row"])                                                       # :426-427
            else:
                single = np.asarray(action_queue.popleft())                                                   # :411-416
                execute_action = host.postprocess_execution(single[0:1])[0]
                log.record_queued(task_description, execute_action, t)
                action_history.append(host.postprocess_verifier(single[0:1])[0])                             # :428-432
            t += 1
    return log.finish(False, t), trace
