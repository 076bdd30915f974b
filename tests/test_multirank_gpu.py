"""N > 1 path on a GPU: two fresh rank processes (started by tests/conftest.py before this process touches the GPU) run
bench.py --small with --share-gpu --backend gloo, in weak and in strong scaling. Checks the prompt sharding, the ONE
all-gather of [score | tokens] records and the winner exchange: every rank must end with the same winner index, the
winner's tokens and its prompt group's tokens; the strong-scaling run must select exactly what an unsharded run selects."""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu


def _collect(run):
    outs = []
    for p in run["procs"]:
        try:
            log, _ = p.communicate(timeout=900)
        except Exception:
            p.kill()
            raise
        assert p.returncode == 0, log[-3000:]
        outs.append(log)
    recs = []
    for r in range(2):
        with open(f"{run['out']}.rank{r}.json") as f:
            recs.append(json.load(f))
    line = [l for l in outs[0].splitlines() if l.startswith("{")][-1]
    return recs, json.loads(line)


def test_two_ranks_weak_scaling_winner_exchange(dev, multirank_runs):
    """Weak scaling = the headline's per-GPU work on DISTINCT candidates: 8 x W prompt groups x 4 samples of one observation, rank r
    owns groups r, r + W, ... (run_simpler_eval_with_openpi.py:296-319 batch construction). The ranks' scores must differ, and the
    winner must be the grouped arg-max over the UNION of both ranks' candidates."""
    assert "weak" in multirank_runs, "rank processes were not launched (no GPU at collection time?)"
    recs, line = _collect(multirank_runs["weak"])
    S, P, W = 4, 8, 2
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["candidates_total"] == W * P * S
    assert line["config"]["prompts_per_gpu"] == P and line["config"]["candidates_per_gpu"] == P * S
    assert recs[0]["prompt_ids"] == list(range(0, W * P, W)) and recs[1]["prompt_ids"] == list(range(1, W * P, W))
    assert recs[0]["global_idx"] == recs[1]["global_idx"]
    assert recs[0]["winner_tokens"] == recs[1]["winner_tokens"] and recs[0]["group_tokens"] == recs[1]["group_tokens"]
    assert recs[0]["scores"] == recs[1]["scores"] and len(recs[0]["scores"]) == W * P * S
    # the winner's tokens are those of the OWNING rank's local candidate: global prompt g = r + j * W
    gi = recs[0]["global_idx"]
    g, s = gi // S, gi % S
    owner, j = g % W, g // W
    assert recs[owner]["local_tokens"][j * S + s] == recs[0]["winner_tokens"]
    assert recs[owner]["local_tokens"][j * S:(j + 1) * S] == recs[0]["group_tokens"]
    # the ranks did DIFFERENT work: other prompts, other uniforms -> other tokens and other scores
    assert recs[0]["local_tokens"] != recs[1]["local_tokens"]
    sc = torch.tensor(recs[0]["scores"]).view(W * P, S)            # global prompt order; rank r's groups are rows r::W
    assert not torch.equal(sc[0::W], sc[1::W])
    # the winner is the grouped arg-max of the union (first maximum wins, efficient_ensemble_merged.py:417-448)
    bg = int(sc.mean(1).argmax())
    assert gi == bg * S + int(sc[bg].argmax())
    # rank r's first prompt groups are what an unsharded headline-sized run computes for the same global prompts: prompt p and the
    # uniforms of candidate n do not depend on how many ranks there are
    import bench
    pipe = bench.Pipeline(dev, small=True)                          # 8 prompts x 4 samples, global prompts 0..7
    _, tok, _ = pipe.decision()
    tok = tok.cpu().view(P, S, 7)
    for r in range(W):
        mine = torch.tensor(recs[r]["local_tokens"]).view(P, S, 7)  # local prompt j = global prompt r + j W
        for j in range(P):
            gp = r + j * W
            if gp < P:
                assert torch.equal(mine[j], tok[gp]), (r, j)


def test_two_ranks_config3_one_prompt_group_per_rank(dev, multirank_runs):
    """BASELINE config 3 as SURVEY 8(d) defines it (N = 256 = 8 GPUs x one prompt group of 32 samples), here at W = 2: rank r owns
    prompt group r with 32 samples; one all-gather; the winner is the grouped arg-max over both groups."""
    assert "config3" in multirank_runs, "rank processes were not launched (no GPU at collection time?)"
    recs, line = _collect(multirank_runs["config3"])
    S, W = 32, 2
    assert line["scaling"] == "weak" and line["config"]["candidates_total"] == W * S and line["config"]["prompts_per_gpu"] == 1
    assert recs[0]["prompt_ids"] == [0] and recs[1]["prompt_ids"] == [1]
    assert recs[0]["global_idx"] == recs[1]["global_idx"] and recs[0]["scores"] == recs[1]["scores"]
    sc = torch.tensor(recs[0]["scores"]).view(W, S)
    assert not torch.equal(sc[0], sc[1])
    bg = int(sc.mean(1).argmax())
    gi = recs[0]["global_idx"]
    assert gi == bg * S + int(sc[bg].argmax())
    assert recs[bg]["local_tokens"][gi % S] == recs[0]["winner_tokens"] == recs[1]["winner_tokens"]
    assert recs[bg]["local_tokens"] == recs[0]["group_tokens"] == recs[1]["group_tokens"]


def test_two_ranks_strong_scaling_equals_unsharded(dev, multirank_runs):
    assert "strong" in multirank_runs, "rank processes were not launched (no GPU at collection time?)"
    recs, line = _collect(multirank_runs["strong"])
    S, P = 4, 8
    assert line["scaling"] == "strong" and line["config"]["candidates_total"] == P * S and line["config"]["prompts_per_gpu"] == P // 2
    assert recs[0]["prompt_ids"] == [0, 2, 4, 6] and recs[1]["prompt_ids"] == [1, 3, 5, 7]
    assert recs[0]["global_idx"] == recs[1]["global_idx"] and recs[0]["winner_tokens"] == recs[1]["winner_tokens"]
    import bench
    pipe = bench.Pipeline(dev, small=True)
    idx, tok, _ = pipe.decision()
    tok = tok.cpu().view(P, S, 7)
    # every rank's local tokens are the unsharded run's tokens of its prompts (candidates do not interact)
    for r in range(2):
        assert recs[r]["local_tokens"] == tok[r::2].reshape(-1, 7).tolist()
    # scores: the trajectory encoder batch differs (16 vs 32 rows) but per-candidate arithmetic is row-independent
    ref = pipe.ver.score_histories
    assert recs[0]["global_idx"] == idx
    assert recs[0]["winner_tokens"] == tok.view(-1, 7)[idx].tolist()
    assert recs[0]["group_tokens"] == tok[idx // S].tolist()


def test_rccl_backend_collectives_single_rank(dev):
    """The collectives bench.py / sharding.py issue on the real multi-GPU path -- init_process_group("nccl" = RCCL, device_id),
    all_gather_into_tensor of fp32 [n, 1 + 7] records on the device, barrier, all_reduce(MAX) of the step time -- run in a fresh
    process with world_size 1 (the box has one GPU: this checks the API surface and the RCCL stack, not the transport)."""
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = (
        "import os, torch, torch.distributed as dist\n"
        "dev = torch.device('cuda:0'); torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)\n"
        "rec = torch.arange(32 * 8, dtype=torch.float32, device=dev).view(32, 8).contiguous()\n"
        "buf = torch.empty(32, 8, dtype=torch.float32, device=dev)\n"
        "dist.all_gather_into_tensor(buf, rec)\n"
        "assert torch.equal(buf, rec)\n"
        "dist.barrier()\n"
        "t = torch.tensor([1.25], device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t[0]) == 1.25\n"
        "dist.barrier(); dist.destroy_process_group(); print('rccl ok')\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "rccl ok" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
