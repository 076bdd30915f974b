"""Candidate sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

Candidates are independent given the observation (run_simpler_eval_with_openpi.py:305-319; the verifier scores each
history independently, efficient_ensemble_merged.py:226-245), so the path shards with ONE exchange: an all-gather of one
fp32 RECORD per candidate -- its score followed by its payload (the sampled action tokens / the action chunk, a few
dozen floats) -- after which every rank applies the same deterministic grouped arg-max
(efficient_ensemble_merged.py:417-448 semantics need every group's mean) and therefore holds, without a second
collective, everything the driver needs on the rank that steps the environment
(run_simpler_eval_with_openpi.py:365-401): the winner's index and score, the winner's chunk, and the chunks of the
winner's whole prompt group (the gripper majority vote of :374-392 runs over that group). N = 256 candidates x
(1 + 28) floats = 29 KB in total: latency-bound, no weight or activation traffic over xGMI, no all-reduce.
Rank r owns prompt groups r, r+W, r+2W, ...; global candidate index = prompt_index * S + sample.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_prompts(prompts: Sequence, rank: int, world: int) -> List:
    return [p for i, p in enumerate(prompts) if i % world == rank]


def _select(g: torch.Tensor, S: int) -> dict:
    """g fp32 [G, S] in global prompt order -> first-maximum-wins grouped arg-max (torch.max semantics).
    Device tensors -- every production caller: the gathered buffer lives where the scores were computed -- go to the library's
    group_argmax kernel. The torch branch below is TEST-ONLY: it is what the world_size-2 `gloo` tests on CPU tensors reach (there is
    no GPU in the CPU test tier); it is not a fallback of the product path, which never holds its scores on the host."""
    if g.is_cuda:
        from . import ops
        res, best = ops.group_argmax(g.reshape(-1).contiguous(), S)
        res, best = res.cpu(), best.cpu()
        return dict(global_idx=int(res[0]), group=int(res[1]), in_group=int(res[2]), max_score=float(best[0]),
                    group_mean=float(best[1]))
    gm, bg = g.mean(dim=1).max(dim=0)
    mx, bi = g[bg].max(dim=0)
    return dict(global_idx=int(bg) * S + int(bi), group=int(bg), in_group=int(bi), max_score=float(mx), group_mean=float(gm))


def gather_records_and_select(local_scores: torch.Tensor, samples_per_prompt: int, rank: int, world: int,
                              n_prompts_total: int, local_payload: Optional[torch.Tensor] = None,
                              payload_width: Optional[int] = None) -> dict:
    """local_scores fp32 [n_local_prompts * S] in this rank's prompt order; local_payload [n_local * S, P...] (any dtype
    exactly representable in fp32: token ids < 2^24, fp32 action chunks) or None -- the SAME choice on every rank.
    payload_width: P, when it cannot be read off local_payload -- a rank that owns no prompt group (world > n_prompts_total) holds an
    empty payload tensor; its trailing dimensions give P when it was built as [0, P...], otherwise pass the width every rank agrees on.
    ONE all-gather of [n_local * S, 1 + P] fp32 records. Returns, identically on every rank:
    dict(global_idx, group, in_group, max_score, group_mean, scores [N], payload [N, P] | None,
         winner_payload [P] | None, group_payload [S, P] | None)."""
    S = samples_per_prompt
    n_loc = local_scores.numel()
    n_local_prompts = n_loc // S
    if local_payload is None:
        P = 0 if payload_width is None else int(payload_width)
        if P and n_loc:
            raise ValueError("payload_width without a payload is only meaningful on a rank that owns no prompt group")
    elif n_loc > 0:
        P = int(local_payload.reshape(n_loc, -1).shape[1])
    elif payload_width is not None:
        P = int(payload_width)
    elif local_payload.dim() >= 2:
        P = 1
        for d in local_payload.shape[1:]:
            P *= int(d)
    else:
        raise ValueError("a rank without prompt groups must pass its payload as [0, P] or give payload_width (the record width of the all-gather)")
    if payload_width is not None and P != int(payload_width):
        raise ValueError(f"payload width {P} differs from payload_width={payload_width}")
    rec = local_scores.reshape(n_loc, 1).to(torch.float32)
    if P:
        pl = local_payload.reshape(n_loc, P).to(torch.float32) if local_payload is not None else torch.zeros(0, P, dtype=torch.float32, device=rec.device)
        rec = torch.cat([rec, pl], dim=1)
    rec = rec.contiguous()
    if world > 1:
        # rank r owns global prompts r, r + W, ...: ceil((G - r) / W) of them. Shards may be ragged (8 prompts on 3 or 5 GPUs): every
        # rank pads its records up to ceil(G / W) prompts for the ONE equal-size all-gather; the padded records are dropped when the
        # gathered buffer is put back into global prompt order (they never reach the arg-max).
        G = n_prompts_total
        expect = (G - rank + world - 1) // world if rank < G else 0
        if n_local_prompts * S != n_loc or n_local_prompts != expect:
            raise ValueError(f"rank {rank} of {world} owns prompts {rank}, {rank + world}, ... of {G}: expected {expect} prompt groups x {S} "
                             f"samples, got {n_loc} local scores")
        max_local = (G + world - 1) // world
        if n_local_prompts < max_local:
            rec = torch.cat([rec, torch.zeros((max_local - n_local_prompts) * S, 1 + P, dtype=torch.float32, device=rec.device)], dim=0)
        buf = torch.empty(world * max_local * S, 1 + P, dtype=torch.float32, device=rec.device)
        dist.all_gather_into_tensor(buf, rec.contiguous())
        # rank r's j-th prompt is global prompt r + j * world -> [j][r] order is global prompt order; the first G groups are the real ones
        allrec = buf.view(world, max_local, S, 1 + P).permute(1, 0, 2, 3).reshape(max_local * world * S, 1 + P)[: G * S]
    else:
        allrec = rec
    scores = allrec[:, 0].contiguous()
    sel = _select(scores.view(-1, S), S)
    sel["scores"] = scores
    if P:
        pay = allrec[:, 1:]
        if local_payload is not None and local_payload.dtype != torch.float32:
            pay = pay.round().to(local_payload.dtype)
        gi, gg = sel["global_idx"], sel["group"]
        sel["payload"] = pay
        sel["winner_payload"] = pay[gi]
        sel["group_payload"] = pay[gg * S:(gg + 1) * S]
    else:
        sel["payload"] = sel["winner_payload"] = sel["group_payload"] = None
    return sel


def gather_scores_and_select(local_scores: torch.Tensor, samples_per_prompt: int, rank: int, world: int,
                             n_prompts_total: int) -> dict:
    """Scores only (no payload): kept for callers that hold the candidates everywhere already."""
    return gather_records_and_select(local_scores, samples_per_prompt, rank, world, n_prompts_total, None)
