"""Candidate sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

Candidates are independent given the observation (run_simpler_eval_with_openpi.py:305-319; the verifier scores each
history independently, efficient_ensemble_merged.py:226-245), so the path shards with ONE exchange: an all-gather of the
per-candidate fp32 scores (128 B per rank at 32 candidates: latency-bound, no weight or activation traffic), after
which every rank applies the same deterministic grouped arg-max (efficient_ensemble_merged.py:417-448 semantics need
every group's mean). Rank r owns prompt groups r, r+W, r+2W, ...; global candidate index = prompt_index * S + sample.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_prompts(prompts: Sequence, rank: int, world: int) -> List:
    return [p for i, p in enumerate(prompts) if i % world == rank]


def gather_scores_and_select(local_scores: torch.Tensor, samples_per_prompt: int, rank: int, world: int,
                             n_prompts_total: int) -> dict:
    """local_scores: fp32 [n_local_prompts * S] in this rank's prompt order. Returns the GLOBAL selection (identical on
    every rank): dict(global_idx, group, in_group, max_score, group_mean, scores)."""
    S = samples_per_prompt
    n_local = local_scores.numel() // S
    if world > 1:
        # equal shards are the contract (prompts % world == 0); a ragged tail would need all_gather with padding
        assert n_prompts_total % world == 0, "prompt count must be a multiple of the world size"
        buf = torch.empty(world * local_scores.numel(), dtype=torch.float32, device=local_scores.device)
        dist.all_gather_into_tensor(buf, local_scores.contiguous())
        # rank r's j-th prompt is global prompt r + j*world  -> scatter back into global prompt order
        g = buf.view(world, n_local, S).permute(1, 0, 2).reshape(n_prompts_total, S)
    else:
        g = local_scores.view(n_local, S)
    if g.is_cuda:
        from . import ops
        res, best = ops.group_argmax(g.reshape(-1).contiguous(), S)
        res, best = res.cpu(), best.cpu()
        return dict(global_idx=int(res[0]), group=int(res[1]), in_group=int(res[2]), max_score=float(best[0]),
                    group_mean=float(best[1]), scores=g.reshape(-1))
    # CPU process groups (tests): same rule, first maximum wins (torch.max semantics)
    gm, bg = g.mean(dim=1).max(dim=0)
    mx, bi = g[bg].max(dim=0)
    return dict(global_idx=int(bg) * S + int(bi), group=int(bg), in_group=int(bi), max_score=float(mx), group_mean=float(gm))
