#!/usr/bin/env python3
"""bench.py — candidates sampled AND scored per second, OpenVLA-7B shapes, 224^2 RGB, on MI355X.

One step = one DECISION of the hot path on one observation:
    frame -> DINOv2-L + SigLIP-So400m + projector -> Llama-2-7B prefill (shared image prefix + 8 prompts) ->
    6 decode passes (7 action tokens for each of N=32 candidates = 8 prompts x 4 samples) -> de-tokenise ->
    CoVer verifier (SigLIP2-L/16-384 image+text towers, 3-member head ensemble, trajectory encoder per candidate) ->
    grouped arg-max.
Weights are synthetic (seeded N(0,0.02), random-init of the named architectures: there is no network for
checkpoints), inputs synthetic and resident in HBM before the timed region.

Multi-GPU (one process per GPU, RCCL):
  --scaling weak   (default; BASELINE config 3 at 8 GPUs): every rank runs its own 8 prompts x 4 samples of the SAME
                   observation (N = 32 per GPU, N = 32 x W in total), ONE all-gather of [score | 7 tokens] records, the
                   same grouped arg-max on every rank; value = all ranks' candidates / max-over-ranks time.
  --scaling strong the headline N = 32 itself sharded: rank r takes prompts r, r+W, ... (8/W prompts x 4 samples).

Prints ONE JSON line (rank 0). Extra objects:
  "roofline"       dominant kernel = the weight-streaming GEMM of the decode passes (HBM-bound), timed live per launch
                   with the kernels' own start/stop stamps (hipExtLaunchKernelGGL event pairs on their stream);
  "roofline_mfma"  every LDS-tiled MFMA GEMM of the decision (LLM prefill + all ViT towers + projector):
                   sum 2MNK / (their kernel time + their split-K reductions) against the dense bf16 peak;
  "end_to_end"     (HBM floor + MFMA floor) / measured step time;
  "cpu_baseline"   the CPU oracle executing ONE FULL candidate (all 32 layers, every tower block) on the host cores.
The profiled decision runs single-threaded on one stream without hipGraph replay, so that kernels neither overlap
(inflated durations) nor hide from the timer; any roofline fraction > 1 is refused (the timer did not time the work).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PROMPTS, N_SAMPLES, LT = 8, 4, 24
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TF = 2500.0      # bf16 dense
BASE_METRIC = "candidate actions scored/sec (whole node), OpenVLA-7B N=32, 224^2 RGB"


def build_inputs(dev, cfg, n_prompts, n_samples, n_cams=1, seed=0, n_gen=7):
    """All ranks build the SAME global inputs (one observation, n_prompts rephrases)."""
    g = torch.Generator().manual_seed(seed)
    frame = torch.randint(0, 256, (n_cams, cfg["image"], cfg["image"], 3), generator=g, dtype=torch.uint8)
    lens = torch.tensor([16 + (i % 8) for i in range(n_prompts)], dtype=torch.int32)
    toks = torch.zeros(n_prompts, LT, dtype=torch.long)
    for p in range(n_prompts):
        toks[p, : lens[p]] = torch.randint(3, cfg["tok_vocab"] - cfg["n_bins"], (int(lens[p]),), generator=g)
    u = torch.rand(n_prompts * n_samples, n_gen, generator=torch.Generator().manual_seed(7))
    img384 = torch.randn(1, 3, 384, 384, generator=g)
    text = torch.randint(0, 32000, (1, 64), generator=g)
    past = torch.randn(6, 7, generator=g) * 0.02
    past[:, 6] = (torch.rand(6, generator=g) > 0.5).float()
    return dict(frame=frame.to(dev), toks=toks.to(dev), lens=lens.to(dev), u=u.to(dev), img384=img384.to(dev), text=text.to(dev),
                past=past.double().numpy())


class Pipeline:
    def __init__(self, dev, small=False, n_prompts=N_PROMPTS, n_samples=N_SAMPLES, n_cams=1, members=3, prompt_ids=None,
                 weight_dtype="bf16", horizon=1):
        """prompt_ids: the global prompt indices this rank owns (strong scaling); None = all n_prompts."""
        from cover_vla_amd import synth
        from cover_vla_amd.openvla import OpenVLA
        from cover_vla_amd.verifier import EfficientEnsembleMerged, SigLIP2Encoder
        self.dev = dev
        c = dict(synth.OPENVLA_SMALL if small else synth.OPENVLA_7B)
        sc = dict(synth.SIGLIP2_SMALL if small else synth.SIGLIP2_L)
        self.c, self.sc = c, sc
        self.n_prompts_global, self.n_samples, self.n_cams = n_prompts, n_samples, n_cams
        self.prompt_ids = list(range(n_prompts)) if prompt_ids is None else list(prompt_ids)
        P = len(self.prompt_ids)
        wd = torch.bfloat16
        sd = synth.openvla_state(c, seed=1234, nontrivial=False, device=dev, wdtype=wd)
        self.horizon, self.weight_dtype = horizon, weight_dtype
        self.policy = OpenVLA(sd, c, device=str(dev), max_prompts=P, max_candidates=P * n_samples, max_text=LT, n_cams=n_cams,
                              horizon=horizon, weight_dtype=weight_dtype)
        del sd
        ssd = synth.siglip2_state(sc, seed=4321, nontrivial=False, device=dev, wdtype=wd)
        if small:
            self.enc = SigLIP2Encoder(ssd, dim=sc["dim"], layers=sc["layers"], heads=sc["heads"], mlp=sc["mlp"], patch=sc["patch"],
                                      image=sc["image"], context_length=sc["context_length"], device=str(dev))
        else:
            self.enc = SigLIP2Encoder(ssd, device=str(dev))
        del ssd
        torch.cuda.empty_cache()
        ck = synth.verifier_checkpoint(members, seed=1234, num_patches=self.enc.num_patches, vision_dim=sc["dim"], text_dim=sc["dim"])
        self.ver = EfficientEnsembleMerged(ck, device=str(dev), encoder=self.enc)
        g = build_inputs(dev, c, n_prompts, n_samples, n_cams, n_gen=7 * horizon)
        ids = torch.tensor(self.prompt_ids, device=dev)
        cand = (ids[:, None] * n_samples + torch.arange(n_samples, device=dev)[None]).reshape(-1)
        self.inp = dict(g, toks=g["toks"][ids].contiguous(), lens=g["lens"][ids].contiguous(), u=g["u"][cand].contiguous())
        self.side = None
        self.pool = None
        bins = np.linspace(-1, 1, c["n_bins"])
        self.centers = torch.tensor((bins[:-1] + bins[1:]) / 2.0, dtype=torch.float32, device=dev)   # float64 -> fp32 table
        self.past_dev = torch.tensor(self.inp["past"], dtype=torch.float32, device=dev)
        if small:
            g = torch.Generator().manual_seed(1)
            self.inp["img384"] = torch.randn(1, 3, sc["image"], sc["image"], generator=g).to(dev)
            self.inp["text"] = torch.randint(0, sc["vocab"], (1, sc["context_length"]), generator=g).to(dev)

    def decision(self, world=1, rank=0, cpu_gather=False, serial=False):
        """One decision. Returns (global winner index, local tokens, selection dict | None)."""
        i = self.inp
        S = self.n_samples
        # The verifier's frozen towers and image-text heads depend only on the observation and the instruction: they run on
        # a side stream while the policy samples. Who queues them matters as much as where they run: queued by this
        # thread before the policy they cost ~2.6 ms of host launch time with the main stream idle; queued from the
        # sampler's after-prefill hook they run underneath the decode passes, whose one-block-per-CU weight-streaming
        # grids lose ~1.4 ms to the co-tenants. A second host thread queues them from t = 0 instead (ctypes releases
        # the GIL inside the library's composites): they overlap the launch-bound vision phase and the start of the
        # prefill, and the decode passes run alone (42.2 -> 41.8 ms). COVER_SIDE_THREAD=0 selects the hook.
        # serial=True (the profiled decision): everything on the main stream from this thread, no hipGraph replay.
        main = torch.cuda.current_stream()
        if self.side is None:
            self.side = torch.cuda.Stream(device=self.dev)
        out = {}

        def side_work():
            pf, tf = self.ver.extract_shared_features(i["img384"], i["text"])
            return self.ver.image_text_embeddings(pf, tf)

        if serial:
            keep = (self.policy.vision_graph, self.policy.vision_overlap)
            self.policy.vision_graph, self.policy.vision_overlap = False, False
            try:
                its = side_work()
                tokens, _ = self.policy.sample(i["frame"], i["toks"], i["lens"], S, i["u"], 1.0)
            finally:
                self.policy.vision_graph, self.policy.vision_overlap = keep
        elif os.environ.get("COVER_SIDE_THREAD", "1") != "0":
            if self.pool is None:
                import concurrent.futures
                self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
            ev = torch.cuda.Event()
            ev.record(main)

            def threaded():
                torch.cuda.set_device(self.dev)
                self.side.wait_event(ev)
                with torch.cuda.stream(self.side):
                    return side_work()

            fut = self.pool.submit(threaded)
            tokens, _ = self.policy.sample(i["frame"], i["toks"], i["lens"], S, i["u"], 1.0)
            its = fut.result()
        else:
            def hook():
                ev = torch.cuda.Event()
                ev.record(main)
                self.side.wait_event(ev)
                with torch.cuda.stream(self.side):
                    out["its"] = side_work()

            tokens, _ = self.policy.sample(i["frame"], i["toks"], i["lens"], S, i["u"], 1.0, on_prefill_enqueued=hook)
            its = out["its"]
        # de-tokenise + assemble the verifier histories on the device: no host sync between sampler and verifier
        from cover_vla_amd import ops
        hb, pad = ops.tokens_to_histories(tokens, self.c["tok_vocab"], self.centers, self.past_dev, n_use=min(self.horizon, 4))
        if not serial:
            main.wait_stream(self.side)
        r = self.ver.score_histories(its, hb, S, pad=pad)
        self.last_scores = r["scores"]
        if world > 1:
            # ONE collective: all-gather of [score | 7 tokens] records (RCCL over xGMI; gloo only in plumbing tests), then the
            # same deterministic grouped arg-max on every rank, which thereby also holds the winner's tokens and its prompt
            # group's tokens (cover_vla_amd/sharding.py)
            from cover_vla_amd.sharding import gather_records_and_select
            sc = r["scores"].cpu() if cpu_gather else r["scores"]
            tk = tokens.cpu() if cpu_gather else tokens
            n_total = self.n_prompts_global if len(self.prompt_ids) < self.n_prompts_global else world * self.n_prompts_global
            sel = gather_records_and_select(sc, S, rank, world, n_prompts_total=n_total, local_payload=tk)
            return sel["global_idx"], tokens, sel
        return int(r["result"][0]), tokens, None


def _cpu_info():
    model, flags = "unknown", set()
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("flags") and not flags:
                    flags = set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return model, {"amx_bf16": "amx_bf16" in flags, "avx512_bf16": "avx512_bf16" in flags, "avx512f": "avx512f" in flags}


def cpu_baseline(pipe, budget_s=90.0):
    """The CPU oracle (oracle/cover_ref, PyTorch-CPU eager bf16) executing ONE FULL candidate exactly as an eager,
    un-deduplicated implementation does: both vision towers at full depth, the 3-layer projector, all 32 Llama layers for
    the T ~ 280 prefill and six single-token decode steps with a concatenated KV cache, lm_head x 7, then the verifier
    (SigLIP2-L image + text towers at full depth, 3-member heads, trajectory encoder, score). Weights: the SAME synthetic
    7B checkpoint the GPU path uses (drawn in HBM, copied to the host). The N = 32 figure is 32 x that candidate
    (as the reference executes it: no dedup; batching on a CPU changes nothing for compute-bound prefill)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from cover_ref import blocks as Bk, openvla as OR, verifier as V
    from cover_vla_amd import synth
    c, sc, dev = pipe.c, pipe.sc, pipe.dev
    t_all = time.time()
    sd = {k: v.cpu() for k, v in synth.openvla_state(c, seed=1234, nontrivial=False, device=dev, wdtype=torch.bfloat16).items()}
    sd = Bk.to_bf16(sd)
    ssd = Bk.to_bf16({k: v.cpu() for k, v in synth.siglip2_state(sc, seed=4321, nontrivial=False, device=dev, wdtype=torch.bfloat16).items()})
    torch.cuda.empty_cache()
    i = pipe.inp
    frame, toks, lens, u = i["frame"][:1].cpu(), i["toks"][:1].cpu(), i["lens"][:1].cpu(), i["u"][:1].cpu()
    ck = synth.verifier_checkpoint(3, seed=1234, num_patches=(sc["image"] // sc["patch"]) ** 2, vision_dim=sc["dim"], text_dim=sc["dim"])
    with torch.no_grad():
        t0 = time.time()
        tok = OR.sample(c, sd, frame, toks, lens, 1, u, 1.0)
        t_policy = time.time() - t0
        t0 = time.time()
        pf, tf = OR.siglip2_features(sc, ssd, i["img384"].cpu(), i["text"].cpu())
        acts = OR.tokens_to_actions(c, tok.numpy())                              # [1, 7]
        hist = [np.concatenate([i["past"], acts.astype(np.float64)], 0)]
        V.compute_max_similarity_scores(ck["ensemble_components"], pf, tf, hist, 1)
        t_ver = time.time() - t0
    per_cand = t_policy + t_ver
    model, isa = _cpu_info()
    return {"value": round(1.0 / per_cand, 4), "unit": "candidates/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"ONE full candidate as an eager un-deduplicated implementation executes it (all layers of every tower, 32 Llama "
                      f"layers prefill T={1 + 256 + int(lens[0])} + 6 decode steps, lm_head x7, verifier towers + 3-member heads): "
                      f"policy {t_policy:.1f} s + verifier {t_ver:.1f} s; N=32 as executed = 32 x this = {32 * per_cand:.0f} s per decision; "
                      f"total CPU-side time incl. copying the 7B checkpoint to the host {time.time() - t_all:.0f} s",
            "seconds_per_candidate": round(per_cand, 2), "cpu_model": model, "isa": isa, "dtype": "bf16 weights, eager PyTorch-CPU"}


def lib_hash():
    from cover_vla_amd import _lib as L
    h = hashlib.sha256()
    with open(L.LIB_PATH, "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:16]


def profile_decision(pipe, world, rank, cpu_gather):
    import ctypes as C
    from cover_vla_amd import _lib as L
    h = L.lib()
    n = 7
    ms, cnt, work = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
    L.check(h.cover_profile_begin(32768), "profile_begin")
    pipe.decision(world, rank, cpu_gather, serial=True)
    L.check(h.cover_profile_end_n(ms, cnt, work, n), "profile_end (event pool overflow = incomplete sums)")
    return list(ms), list(cnt), list(work)


def roofline_objects(ms, cnt, work, ms_per_step, lib_sha):
    out = {}

    def guard(name, obj):
        if obj["frac"] > 1.0:   # a fraction above the peak means the timer did not time the work: not evidence
            out[name] = {"invalid": f"frac {obj['frac']} > 1: refused", "launches": obj.get("launches")}
        else:
            out[name] = obj

    if cnt[0] > 0 and ms[0] > 0:
        ach = work[0] / (ms[0] * 1e-3) / 1e9
        traffic, tnote = None, "no PMC pass recorded for this build"
        try:  # HBM bytes per launch from the PMC pass of the SAME build (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction)
            with open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")) as f:
                t = json.load(f)
            if t.get("lib_sha16") == lib_sha:
                traffic, tnote = round(t["hbm_fetch_bytes_per_launch"]), f"profiles/r02_pmc_traffic.json (lib {lib_sha})"
            else:
                tnote = f"profiles/r02_pmc_traffic.json is from lib {t.get('lib_sha16')}, this run is {lib_sha}: not quoted"
        except Exception:
            pass
        guard("roofline", {"bound": "hbm", "kernel": "gemm_skinny2 / gemm_skinny3 (weight-streaming GEMMs of the 7B decode passes and lm_head, M = 32, >= 16 MB of weights each)",
                           "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                           "traffic": traffic, "traffic_source": tnote, "algorithmic_bytes_per_launch": round(work[0] / cnt[0]), "launches": int(cnt[0]),
                           "avg_launch_us": round(1e3 * ms[0] / cnt[0], 2), "algorithmic_bytes_per_decision": work[0],
                           "kernel_ms_per_decision": round(ms[0], 3), "splitk_reduce_ms_per_decision": round(ms[5], 3)})
    t_mfma = ms[1] + ms[4] + ms[6]
    if cnt[1] + cnt[4] > 0 and t_mfma > 0:
        tf_all = (work[1] + work[4]) / (t_mfma * 1e-3) / 1e12
        obj = {"bound": "mfma", "kernel": "gemm_tiled / gemm_tiled_pc (LLM prefill + DINOv2 / SigLIP / SigLIP2 towers + projector), split-K reductions included",
               "achieved": round(tf_all, 1), "peak": MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf_all / MFMA_PEAK_TF, 4),
               "launches": int(cnt[1] + cnt[4]), "kernel_ms_per_decision": round(t_mfma, 3), "flop_per_decision": work[1] + work[4]}
        if cnt[4] > 0 and ms[4] > 0:
            tf_p = work[4] / (ms[4] * 1e-3) / 1e12
            obj["prefill"] = {"achieved": round(tf_p, 1), "frac": round(tf_p / MFMA_PEAK_TF, 4), "launches": int(cnt[4]),
                              "kernel_ms_per_decision": round(ms[4], 3)}
        if cnt[1] > 0 and ms[1] > 0:
            tf_v = work[1] / ((ms[1] + ms[6]) * 1e-3) / 1e12
            obj["vit"] = {"achieved": round(tf_v, 1), "frac": round(tf_v / MFMA_PEAK_TF, 4), "launches": int(cnt[1]),
                          "kernel_ms_per_decision": round(ms[1] + ms[6], 3)}
        if any(o.get("frac", 0) > 1.0 for o in (obj, obj.get("prefill", {}), obj.get("vit", {}))):
            out["roofline_mfma"] = {"invalid": "a fraction > 1: refused"}
        else:
            out["roofline_mfma"] = obj
    if "roofline" not in out and isinstance(out.get("roofline_mfma"), dict) and "frac" in out["roofline_mfma"]:
        # no weight-streaming launch of >= 16 MB in this configuration (M > 64 decode rows, e.g. config 5: N = 512): the dominant
        # kernel class is the tiled GEMM, bounded by the matrix pipes
        m = out["roofline_mfma"]
        out["roofline"] = {"bound": "mfma", "kernel": m["kernel"], "achieved": m["achieved"], "peak": m["peak"], "unit": m["unit"], "frac": m["frac"],
                           "traffic": None, "launches": m["launches"], "kernel_ms_per_decision": m["kernel_ms_per_decision"]}
    if cnt[3] > 0:
        out["small_streaming_gemms"] = {"kernel": "weight-streaming launches with < 16 MB of weights (verifier text tower etc.): latency-bound",
                                        "launches": int(cnt[3]), "kernel_ms_per_decision": round(ms[3], 3)}
    if cnt[2] > 0:
        out["attention_kernels"] = {"launches": int(cnt[2]), "kernel_ms_per_decision": round(ms[2], 3)}
    floor_ms = 1e3 * (work[0] / (HBM_PEAK_GBS * 1e9) + (work[1] + work[4]) / (MFMA_PEAK_TF * 1e12))
    if floor_ms > 0 and ms_per_step > 0:
        e2e = floor_ms / ms_per_step
        out["end_to_end"] = ({"floor_ms": round(floor_ms, 3), "measured_ms": round(ms_per_step, 3), "frac": round(e2e, 4),
                              "floor": "decode weight bytes / 8 TB/s + tiled-GEMM FLOPs / 2.5 PFLOP/s (attention, norms, heads: 0)"}
                             if e2e <= 1.0 else {"invalid": f"frac {e2e:.3f} > 1: refused"})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--small", action="store_true", help="tiny config (plumbing check, not a valid bench line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the profiled decision (plumbing tests)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) for real runs; gloo only to test the N>1 plumbing")
    ap.add_argument("--share-gpu", action="store_true", help="plumbing test: every rank uses cuda:0")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--samples", type=int, default=N_SAMPLES, help="samples per prompt (4 = headline N=32; 2 = config 2, N=16)")
    ap.add_argument("--cams", type=int, default=1, help="cameras (2 = config 4)")
    ap.add_argument("--members", type=int, default=3, help="verifier ensemble members (2 = config 4)")
    ap.add_argument("--no-agreement", action="store_true", help="fp8: skip the comparison run through a bf16 pipeline")
    ap.add_argument("--dtype", choices=["bf16", "fp8"], default="bf16", help="fp8 = e4m3 decoder + lm_head weights (config 5)")
    ap.add_argument("--horizon", type=int, default=1, help="action-chunk horizon: 7 x horizon action tokens per candidate (config 5: 8)")
    ap.add_argument("--check-out", default=None, help="write this rank's selection (winner index / tokens) as JSON (plumbing tests)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU as CHILD processes (torch.distributed.run) before
        # anything here touches the GPU, and leave with their exit code (never exec from a process that may have initialised HIP)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and "WORLD_SIZE" in os.environ and a.gpus != 1:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started {world} rank(s)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU: bench.py measures the HIP path only (no CPU fallback)"}))
        sys.exit(2)
    if a.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    strong = a.scaling == "strong" and world > 1
    if strong and N_PROMPTS % world:
        raise SystemExit(f"--scaling strong shards the {N_PROMPTS} prompt groups: world size must divide {N_PROMPTS}")
    prompt_ids = list(range(rank, N_PROMPTS, world)) if strong else None
    pipe = Pipeline(dev, small=a.small, n_samples=a.samples, n_cams=a.cams, members=a.members, prompt_ids=prompt_ids,
                    weight_dtype=a.dtype, horizon=a.horizon)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()

    cpu_gather = a.backend != "nccl"
    for _ in range(a.warmup):
        pipe.decision(world, rank, cpu_gather)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = pipe.decision(world, rank, cpu_gather)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev if a.backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t[0])
    if a.check_out:
        gi, toks, sel = last
        rec = {"rank": rank, "global_idx": int(gi), "local_tokens": toks.cpu().tolist(), "prompt_ids": pipe.prompt_ids}
        if sel is not None:
            rec.update(winner_tokens=sel["winner_payload"].cpu().tolist(), group_tokens=sel["group_payload"].cpu().tolist(),
                       scores=[float(x) for x in sel["scores"].cpu()])
        with open(f"{a.check_out}.rank{rank}.json", "w") as f:
            json.dump(rec, f)

    n_local = len(pipe.prompt_ids) * a.samples
    n_total = N_PROMPTS * a.samples if (strong or world == 1) else world * n_local
    ms_per_step = 1000.0 * dt / a.steps
    headline = (a.samples == N_SAMPLES and a.cams == 1 and a.members == 3 and not a.small and a.dtype == "bf16" and a.horizon == 1)
    metric = BASE_METRIC
    if world > 1 and not strong:
        metric += f" [weak scaling: N=32 per GPU, N={n_total} in total at {world} GPUs]"
    if not headline:
        metric = (f"candidate actions scored/sec (whole node), OpenVLA-7B N={n_total}, 224^2 RGB x {a.cams} camera(s), verifier ensemble={a.members}, "
                  f"{a.dtype} weights, action-chunk horizon {a.horizon}")
    out = {
        "metric": metric,
        "value": round(n_total * a.steps / dt, 3), "unit": "candidates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "bf16" if a.dtype == "bf16" else "fp8 (e4m3 decoder + lm_head weights, per-channel 2^e scales; bf16 activations, KV cache, vision towers)",
        "data": "synthetic",
        "config": {"workload": ("SMALL-PLUMBING-CONFIG (invalid as a bench line)" if a.small else
                                f"OpenVLA-7B (DINOv2-L+SigLIP-So400m+Llama-2-7B) N={n_local} = {len(pipe.prompt_ids)} prompts x {a.samples} samples per GPU, "
                                f"{7 * a.horizon} action tokens, {a.cams} 224x224 RGB frame(s); CoVer verifier SigLIP2-L/16-384 + {a.members}-member ensemble; "
                                "random-init weights"),
                   "candidates_total": n_total, "candidates_per_gpu": n_local, "prompts_per_gpu": len(pipe.prompt_ids),
                   "parallelism": f"candidate-sharded x{world} ({'strong' if strong else 'weak'})", "lib_sha16": lib_hash()},
    }
    if not a.no_profile:
        # ---- rooflines: kernel start/stop stamps of every GEMM / attention launch of one extra (serialised) decision
        ms, cnt, work = profile_decision(pipe, world, rank, cpu_gather)
        out.update(roofline_objects(ms, cnt, work, ms_per_step, out["config"]["lib_sha16"]))
    if a.dtype == "fp8" and world == 1 and not a.no_agreement:
        # what quantisation changes ("report arg-max agreement rate and score RMSE"): the same decision through a bf16
        # pipeline of the same synthetic checkpoint, same frame / prompts / uniforms
        gi8, tok8, _ = pipe.decision()
        sc8 = pipe.last_scores.clone()
        ref = Pipeline(dev, small=a.small, n_samples=a.samples, n_cams=a.cams, members=a.members, prompt_ids=prompt_ids, weight_dtype="bf16",
                       horizon=a.horizon)
        gi16, tok16, _ = ref.decision()
        sc16 = ref.last_scores
        dbin = (tok8 - tok16).abs().float()
        out["fp8_vs_bf16"] = {"token_agreement": round(float((tok8 == tok16).float().mean()), 4), "mean_bin_distance": round(float(dbin.mean()), 3),
                              "first_token_agreement": round(float((tok8[:, 0] == tok16[:, 0]).float().mean()), 4),
                              "score_rmse": round(float((sc8 - sc16).pow(2).mean().sqrt()), 5), "winner_same": bool(gi8 == gi16),
                              "note": "random-init weights: logits are near-flat over the 256 action bins, so sampled picks are maximally "
                                      "sensitive to quantisation; the e4m3 stream itself is bit-identical to bf16 on the de-quantised weights "
                                      "(tests/test_fp8_gpu.py)"}
        del ref
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.small:
        out["cpu_baseline"] = cpu_baseline(pipe)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
