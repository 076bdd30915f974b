"""Host-side glue of the evaluation loop, restated for the candidate batch (float64 numpy, vectorised).

The reference does this work in a B x 4 Python loop that rebuilds the statistics arrays 160 times per decision
(eval_utils.py:172-221 -> simpler.py:96-166); here it is a handful of array expressions with the same arithmetic.
  denormalize_bound            INT-ACT/src/experiments/env_adapters/base.py:20-31
  postprocess_verifier         .../simpler.py:96-121 (+ BridgeSimplerAdapter.postprocess_gripper_verifier :222-226)
  postprocess (execution)      .../simpler.py:123-166 (+ postprocess_gripper :211-220, euler2axangle INT-ACT/src/utils/geometry.py:261-436)
  process_inputs               CoVer_VLA/inference/experiments/robot/simpler/eval_utils.py:172-221
  two-stage verification, gripper vote, chunk extraction
                               .../run_simpler_eval_with_openpi.py:329-401
"""
from __future__ import annotations

import json
import math
import os
from collections import deque
from typing import List, Optional, Sequence

import numpy as np

_STATS = None


def bridge_statistics() -> dict:
    global _STATS
    if _STATS is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "bridge_statistics.json")) as f:
            _STATS = json.load(f)
    return _STATS


def denormalize_bound(data, data_min, data_max, clip_min=-1.0, clip_max=1.0):
    return (data - clip_min) / (clip_max - clip_min) * (data_max - data_min) + data_min


def euler2axangle_sxyz(roll, pitch, yaw):
    """euler2quat('sxyz') followed by quat2axangle, vectorised over leading dims. Returns axis*angle [..., 3]."""
    ai, aj, ak = np.asarray(roll) / 2.0, np.asarray(pitch) / 2.0, np.asarray(yaw) / 2.0
    ci, si, cj, sj, ck, sk = np.cos(ai), np.sin(ai), np.cos(aj), np.sin(aj), np.cos(ak), np.sin(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    q = np.stack([cj * cc + sj * ss, cj * sc - sj * cs, cj * ss + sj * cc, cj * cs - sj * sc], axis=-1)
    eps = np.finfo(np.float64).eps
    nq = np.sum(q ** 2, axis=-1, keepdims=True)
    qn = np.where(nq != 1, q / np.sqrt(np.where(nq > 0, nq, 1.0)), q)
    xyz = qn[..., 1:]
    len2 = np.sum(xyz ** 2, axis=-1, keepdims=True)
    ident = (len2 < (eps * 3) ** 2) | (nq < eps ** 2)
    theta = 2.0 * np.arccos(np.clip(qn[..., :1], -1.0, 1.0))
    axis = xyz / np.sqrt(np.where(ident, 1.0, len2))
    axis = np.where(ident, np.array([1.0, 0.0, 0.0]), axis)
    theta = np.where(ident, 0.0, theta)
    return axis * theta


def postprocess_verifier(actions: np.ndarray, stats: Optional[dict] = None) -> np.ndarray:
    """[n,7] normalised policy actions -> verifier format: dims 0-5 un-normalised with p01/p99, gripper 0 if a<0.5 else 1."""
    st = (stats or bridge_statistics())["action"]
    lo, hi = np.array(st["p01"])[:-1], np.array(st["p99"])[:-1]
    out = np.zeros((len(actions), 7))
    out[:, :6] = denormalize_bound(actions[:, :-1], lo, hi)
    out[:, 6] = np.where(actions[:, -1] < 0.5, 0, 1)
    return out


def postprocess_execution(actions: np.ndarray, stats: Optional[dict] = None) -> np.ndarray:
    """[n,7] -> execution format: xyz, axis-angle rotation, gripper 2*(a>0.5)-1."""
    st = (stats or bridge_statistics())["action"]
    lo, hi = np.array(st["p01"])[:-1], np.array(st["p99"])[:-1]
    raw = denormalize_bound(actions[:, :-1], lo, hi)
    out = np.zeros((len(actions), 7))
    out[:, :3] = raw[:, :3]
    out[:, 3:6] = euler2axangle_sxyz(raw[:, 3], raw[:, 4], raw[:, 5])
    out[:, 6] = 2.0 * (actions[:, -1] > 0.5) - 1.0
    return out


def process_inputs(action_queue: Sequence[np.ndarray], verifier_action: bool, action_history: Sequence[np.ndarray],
                   n_action_steps: int = 4, stats: Optional[dict] = None) -> List[np.ndarray]:
    """action_queue: n_action_steps arrays [B,7] (float32 from the policy) -> list of B trajectories [num_past + steps, 7]."""
    fn = postprocess_verifier if verifier_action else postprocess_execution
    fut = np.stack([fn(np.asarray(action_queue[i]), stats) for i in range(n_action_steps)])      # [steps, B, 7]
    fut = fut.transpose(1, 0, 2)
    B = fut.shape[0]
    num_past = min(len(action_history), 6)
    if num_past > 0:
        past = np.stack(list(action_history)[-num_past:])
        full = np.concatenate([np.repeat(past[None], B, axis=0), fut], axis=1)
    else:
        full = fut
    return [full[i] for i in range(B)]


def verify_and_select(verifier, raw_image, task_description: str, task_list: Sequence[str], action_queue, action_history,
                      samples_per_prompt: int, n_action_steps: int = 4, threshold: float = 0.1, stats=None,
                      process_image: bool = True):
    """run_simpler_eval_with_openpi.py:329-401: stage 1 scores candidate 0 under the current instruction; if its score
    is < threshold, stage 2 scores all candidates grouped per prompt; then the gripper majority vote inside the winner's
    prompt group and extraction of the winner's remaining steps.
    action_queue: n_action_steps arrays [B,7] (host). Returns dict(execute_action, max_score, max_instruction,
    global_action_idx, remaining (deque of [1,7]), history_row)."""
    B = len(task_list)
    num_past = min(len(action_history), 6)
    hist_v = process_inputs(action_queue, True, action_history, n_action_steps, stats)
    if process_image:      # images_list = [process_raw_image_to_jpg(raw_img)] * B  (run_simpler_eval_with_openpi.py:342)
        from .imaging import process_raw_image_to_jpg
        raw_image = process_raw_image_to_jpg(raw_image)
    images = [raw_image] * B
    max_score, max_instruction, max_hist, gidx = verifier.compute_max_similarity_scores_batch(
        images=images[0:1], instructions=[task_description], all_action_histories=hist_v[0:1],
        cfg_repeat_language_instructions=1)
    if max_score < threshold:
        max_score, _, max_hist, gidx = verifier.compute_max_similarity_scores_batch(
            images=images, instructions=[task_description] * B, all_action_histories=hist_v,
            cfg_repeat_language_instructions=samples_per_prompt)
        max_instruction = task_list[int(gidx)]
    gidx = int(gidx)
    hist_e = process_inputs(action_queue, False, action_history, n_action_steps, stats)
    execute_action = hist_e[gidx][num_past].copy()
    g0 = (gidx // samples_per_prompt) * samples_per_prompt
    grippers = np.stack(hist_e[g0:g0 + samples_per_prompt])[:, num_past, -1]
    close_votes, open_votes = int((grippers >= 0).sum()), int((grippers < 0).sum())
    if close_votes > open_votes:
        execute_action[-1] = 1.0
    elif open_votes > close_votes:
        execute_action[-1] = -1.0
    else:
        execute_action[-1] = 1.0 if execute_action[-1] >= 0 else -1.0
    execute_action[-1] = float(np.sign(execute_action[-1]))
    remaining = deque(np.asarray(action_queue[t])[gidx:gidx + 1] for t in range(1, n_action_steps))
    return dict(execute_action=execute_action, max_score=max_score, max_instruction=max_instruction,
                global_action_idx=gidx, remaining=remaining, history_row=max_hist[num_past].copy())


class EpisodeLog:
    """The per-episode record the driver pickles (run_simpler_eval_with_openpi.py:238-247 schema; filled at :404-407 on a
    verified decision, :419-422 on a queued step, closed at :454-455) -- the e2e parity artefact analyze_success_rate.py reads.
    Same field names and per-step value types, so records written next to this path load in the reference's analysis."""

    FIELDS = ("verifier_scores", "selected_instructions", "execute_actions", "step_timestamps", "original_task_description",
              "used_task_description", "success", "episode_length")

    def __init__(self, original_task_description: str, task_description: str):
        self.data = {"verifier_scores": [], "selected_instructions": [], "execute_actions": [], "step_timestamps": [],
                     "original_task_description": original_task_description, "used_task_description": task_description,
                     "success": False, "episode_length": 0}

    def record_decision(self, max_score: float, max_instruction: str, execute_action, t: int) -> None:
        d = self.data
        d["verifier_scores"].append(max_score)
        d["selected_instructions"].append(max_instruction)
        d["execute_actions"].append(np.asarray(execute_action).copy())
        d["step_timestamps"].append(t)

    def record_queued(self, task_description: str, execute_action, t: int) -> None:
        d = self.data
        d["verifier_scores"].append(None)
        d["selected_instructions"].append(task_description)
        d["execute_actions"].append(np.asarray(execute_action).copy())
        d["step_timestamps"].append(t)

    def finish(self, success: bool, t: int) -> dict:
        self.data["success"] = success
        self.data["episode_length"] = t
        return self.data

    def save(self, path: str) -> None:
        import pickle
        with open(path, "wb") as f:
            pickle.dump(self.data, f)
