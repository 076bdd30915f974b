#!/usr/bin/env python3
"""bench.py — candidates sampled AND scored per second, OpenVLA-7B shapes, 224^2 RGB, on MI355X.

One step = one DECISION of the hot path on one observation:
    frame -> DINOv2-L + SigLIP-So400m + projector -> Llama-2-7B prefill (shared image prefix + 8 prompts) ->
    6 decode passes (7 action tokens for each of N=32 candidates = 8 prompts x 4 samples) -> de-tokenise ->
    CoVer verifier (SigLIP2-L/16-384 image+text towers, 3-member head ensemble, trajectory encoder per candidate) ->
    grouped arg-max.
Weights are synthetic (seeded N(0,0.02), random-init of the named architectures: there is no network for
checkpoints), inputs synthetic and resident in HBM before the timed region. Multi-GPU = weak scaling: every rank
runs its own 8 prompts x 4 samples of the SAME observation, scores are all-gathered (RCCL) and every rank runs the
same grouped arg-max; value = all ranks' candidates / max-over-ranks time.

Prints ONE JSON line (rank 0). Extra objects: "roofline" (dominant kernel = the weight-streaming GEMM of the decode
passes, timed live with hipEvents on its stream through the library's profiling hook) and "cpu_baseline" (the CPU
oracle, un-deduplicated as an eager implementation executes it, on a bounded sample scaled by layer counts).
"""
from __future__ import annotations

import argparse
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


def build_inputs(dev, cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    frame = torch.randint(0, 256, (1, cfg["image"], cfg["image"], 3), generator=g, dtype=torch.uint8)
    lens = torch.tensor([16 + (i % 8) for i in range(N_PROMPTS)], dtype=torch.int32)
    toks = torch.zeros(N_PROMPTS, LT, dtype=torch.long)
    for p in range(N_PROMPTS):
        toks[p, : lens[p]] = torch.randint(3, cfg["tok_vocab"] - cfg["n_bins"], (int(lens[p]),), generator=g)
    u = torch.rand(N_PROMPTS * N_SAMPLES, 7, generator=torch.Generator().manual_seed(7))
    img384 = torch.randn(1, 3, 384, 384, generator=g)
    text = torch.randint(0, 32000, (1, 64), generator=g)
    past = torch.randn(6, 7, generator=g) * 0.02
    past[:, 6] = (torch.rand(6, generator=g) > 0.5).float()
    return dict(frame=frame.to(dev), toks=toks.to(dev), lens=lens.to(dev), u=u.to(dev), img384=img384.to(dev), text=text.to(dev),
                past=past.double().numpy())


class Pipeline:
    def __init__(self, dev, small=False):
        from cover_vla_amd import synth
        from cover_vla_amd.openvla import OpenVLA
        from cover_vla_amd.verifier import EfficientEnsembleMerged, SigLIP2Encoder
        self.dev = dev
        c = dict(synth.OPENVLA_SMALL if small else synth.OPENVLA_7B)
        sc = dict(synth.SIGLIP2_SMALL if small else synth.SIGLIP2_L)
        self.c, self.sc = c, sc
        wd = torch.bfloat16
        sd = synth.openvla_state(c, seed=1234, nontrivial=False, device=dev, wdtype=wd)
        self.policy = OpenVLA(sd, c, device=str(dev), max_prompts=N_PROMPTS, max_candidates=N_PROMPTS * N_SAMPLES, max_text=LT)
        del sd
        ssd = synth.siglip2_state(sc, seed=4321, nontrivial=False, device=dev, wdtype=wd)
        if small:
            self.enc = SigLIP2Encoder(ssd, dim=sc["dim"], layers=sc["layers"], heads=sc["heads"], mlp=sc["mlp"], patch=sc["patch"],
                                      image=sc["image"], context_length=sc["context_length"], device=str(dev))
        else:
            self.enc = SigLIP2Encoder(ssd, device=str(dev))
        del ssd
        torch.cuda.empty_cache()
        ck = synth.verifier_checkpoint(3, seed=1234, num_patches=self.enc.num_patches, vision_dim=sc["dim"], text_dim=sc["dim"])
        self.ver = EfficientEnsembleMerged(ck, device=str(dev), encoder=self.enc)
        self.inp = build_inputs(dev, c)
        self.side = None
        self.pool = None
        bins = np.linspace(-1, 1, c["n_bins"])
        self.centers = torch.tensor((bins[:-1] + bins[1:]) / 2.0, dtype=torch.float32, device=dev)   # float64 -> fp32 table
        self.past_dev = torch.tensor(self.inp["past"], dtype=torch.float32, device=dev)
        if small:
            g = torch.Generator().manual_seed(1)
            self.inp["img384"] = torch.randn(1, 3, sc["image"], sc["image"], generator=g).to(dev)
            self.inp["text"] = torch.randint(0, sc["vocab"], (1, sc["context_length"]), generator=g).to(dev)

    def decision(self, world=1, rank=0, cpu_gather=False):
        i = self.inp
        # The verifier's frozen towers and image-text heads depend only on the observation and the instruction: they run on
        # a side stream while the policy samples. Who queues them matters as much as where they run: queued by this
        # thread before the policy they cost ~2.6 ms of host launch time with the main stream idle; queued from the
        # sampler's after-prefill hook they run underneath the decode passes, whose one-block-per-CU weight-streaming
        # grids lose ~1.4 ms to the co-tenants. A second host thread queues them from t = 0 instead (ctypes releases
        # the GIL inside the library's composites): they overlap the launch-bound vision phase and the start of the
        # prefill, and the decode passes run alone (42.2 -> 41.8 ms). COVER_SIDE_THREAD=0 selects the hook.
        main = torch.cuda.current_stream()
        if self.side is None:
            self.side = torch.cuda.Stream(device=self.dev)
        out = {}

        def side_work():
            with torch.cuda.stream(self.side):
                pf, tf = self.ver.extract_shared_features(i["img384"], i["text"])
                return self.ver.image_text_embeddings(pf, tf)

        if os.environ.get("COVER_SIDE_THREAD", "1") != "0":
            if self.pool is None:
                import concurrent.futures
                self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
            ev = torch.cuda.Event()
            ev.record(main)

            def threaded():
                torch.cuda.set_device(self.dev)
                self.side.wait_event(ev)
                return side_work()

            fut = self.pool.submit(threaded)
            tokens, _ = self.policy.sample(i["frame"], i["toks"], i["lens"], N_SAMPLES, i["u"], 1.0)
            its = fut.result()
        else:
            def hook():
                ev = torch.cuda.Event()
                ev.record(main)
                self.side.wait_event(ev)
                out["its"] = side_work()

            tokens, _ = self.policy.sample(i["frame"], i["toks"], i["lens"], N_SAMPLES, i["u"], 1.0, on_prefill_enqueued=hook)
            its = out["its"]
        # de-tokenise + assemble the verifier histories on the device: no host sync between sampler and verifier
        from cover_vla_amd import ops
        hb, pad = ops.tokens_to_histories(tokens, self.c["tok_vocab"], self.centers, self.past_dev)
        main.wait_stream(self.side)
        r = self.ver.score_histories(its, hb, N_SAMPLES, pad=pad)
        if world > 1:
            # ONE collective: all-gather of the per-candidate scores (RCCL over xGMI; gloo only in plumbing tests), then the
            # same deterministic grouped arg-max on every rank (cover_vla_amd/sharding.py)
            from cover_vla_amd.sharding import gather_scores_and_select
            sc = r["scores"].cpu() if cpu_gather else r["scores"]
            sel = gather_scores_and_select(sc, N_SAMPLES, rank, world, n_prompts_total=world * N_PROMPTS)
            return sel["global_idx"], tokens
        return int(r["result"][0]), tokens


def cpu_baseline(c, sc):
    """CPU oracle (oracle/cover_ref, PyTorch-CPU eager bf16) on a bounded sample, as an eager implementation executes
    the path: every candidate is a full forward (no dedup). Sample: 1 candidate; 2 blocks of each ViT tower and 1 Llama
    layer for prefill (T=280) and for one decode step, scaled by the layer counts; lm_head x7; verifier towers 2 blocks
    each scaled; heads measured in full for one candidate."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from cover_ref import blocks as Bk
    from cover_vla_amd import synth
    t_all = time.time()
    torch.manual_seed(0)
    bf = torch.bfloat16

    def timeit(fn, reps=2):
        fn()
        t = time.time()
        for _ in range(reps):
            fn()
        return (time.time() - t) / reps

    with torch.no_grad():
        # Llama: one layer
        g = synth._G(1, False, 0.02)
        lsd = Bk.to_bf16(synth.decoder_state(g, dim=c["llm_dim"], layers=1, Hq=c["Hq"], Hkv=c["Hkv"], D=c["D"], mlp=c["llm_mlp"], rms_base=1.0))
        cfg = Bk.DecoderCfg(c["llm_dim"], 1, c["Hq"], c["Hkv"], c["D"], c["llm_mlp"], "silu", "llama", 1e-5, "hf")
        T = 280
        x = torch.randn(1, T, c["llm_dim"]).to(bf)
        mask = torch.tril(torch.ones(T, T, dtype=torch.bool))[None]
        pos = torch.arange(T)[None]
        t_prefill_layer = timeit(lambda: Bk.decoder_forward(cfg, lsd, x, pos, mask, keep_kv=True, final_norm=False, n_pos=512))
        _, kv = Bk.decoder_forward(cfg, lsd, x, pos, mask, keep_kv=True, final_norm=False, n_pos=512)
        x1 = torch.randn(1, 1, c["llm_dim"]).to(bf)
        m1 = torch.ones(1, 1, T + 1, dtype=torch.bool)
        t_dec_layer = timeit(lambda: Bk.decoder_forward(cfg, lsd, x1, torch.tensor([[T]]), m1, past=kv, keep_kv=False, final_norm=False, n_pos=512), 4)
        head = (torch.randn(c["vocab"], c["llm_dim"]) * 0.02).to(bf)
        t_head = timeit(lambda: torch.nn.functional.linear(x1, head), 4)
        # ViT towers: 2 blocks each
        def vit_time(dim, heads, mlp, act, T, ls):
            vg = synth._G(2, False, 0.02)
            vsd = Bk.to_bf16(synth.vit_state(vg, dim=dim, layers=2, heads=heads, mlp=mlp, patch=14, n_pos=T, layerscale=ls))
            vc = Bk.VitCfg(dim, 2, heads, mlp, 14, act, 1e-6, layerscale=ls)
            xx = torch.randn(1, T, dim).to(bf)
            return timeit(lambda: Bk.vit_encode(vc, vsd, xx)) / 2
        t_dino = vit_time(c["dino_dim"], c["dino_heads"], c["dino_mlp"], "gelu_erf", 261, True) * (c["dino_layers"] - 1)
        t_sig = vit_time(c["sig_dim"], c["sig_heads"], c["sig_mlp"], "gelu_tanh", 256, False) * (c["sig_layers"] - 1)
        t_v_img = vit_time(sc["dim"], sc["heads"], sc["mlp"], "gelu_tanh", 576, False) * sc["layers"]
        t_v_txt = vit_time(sc["dim"], sc["heads"], sc["mlp"], "gelu_tanh", 64, False) * sc["layers"]
        # verifier heads for one candidate, 3 members (measured in full)
        from cover_ref import verifier as V
        ck = synth.verifier_checkpoint(3, seed=1234)
        pf, tf, hists = synth.verifier_inputs(1, seed=7)
        t_heads = timeit(lambda: V.compute_max_similarity_scores(ck["ensemble_components"], pf, tf, hists, 1), 1)
    per_cand = (t_dino + t_sig + c["llm_layers"] * (t_prefill_layer + 6 * t_dec_layer) + 7 * t_head + t_v_img + t_v_txt + t_heads)
    return {"value": round(1.0 / per_cand, 4), "unit": "candidates/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 candidate, eager bf16 PyTorch-CPU oracle, no dedup: 1 of 32 Llama layers (prefill T=280 + 1 decode step, "
                      "scaled x32 and x6 steps), 2 blocks of each ViT tower scaled to full depth, lm_head x7, verifier heads in "
                      f"full; measured {time.time() - t_all:.1f}s of CPU work",
            "seconds_per_candidate": round(per_cand, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--small", action="store_true", help="tiny config (plumbing check, not a valid bench line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) for real runs; gloo only to test the N>1 plumbing")
    ap.add_argument("--share-gpu", action="store_true", help="plumbing test: every rank uses cuda:0")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    from cover_vla_amd import _lib as L
    pipe = Pipeline(dev, small=a.small)

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
        pipe.decision(world, rank, cpu_gather)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev if a.backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t[0])

    # ---- roofline of the dominant kernel, live hipEvent timing of every launch of one extra decision
    import ctypes as C
    h = L.lib()
    ms, cnt, work = (C.c_double * 4)(), (C.c_longlong * 4)(), (C.c_double * 4)()
    L.check(h.cover_profile_begin(16384), "profile_begin")
    pipe.decision(world, rank, cpu_gather)
    L.check(h.cover_profile_end(ms, cnt, work), "profile_end")
    n_total = world * N_PROMPTS * N_SAMPLES
    out = {
        "metric": "candidate actions scored/sec (whole node), OpenVLA-7B N=32, 224^2 RGB",
        "value": round(n_total * a.steps / dt, 3), "unit": "candidates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1000.0 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": ("SMALL-PLUMBING-CONFIG (invalid as a bench line)" if a.small else
                                "OpenVLA-7B (DINOv2-L+SigLIP-So400m+Llama-2-7B) N=32 = 8 prompts x 4 samples per GPU, 7 action tokens, "
                                "one 224x224 RGB frame; CoVer verifier SigLIP2-L/16-384 + 3-member ensemble; random-init weights"),
                   "candidates_per_gpu": N_PROMPTS * N_SAMPLES, "prompts_per_gpu": N_PROMPTS, "parallelism": f"candidate-sharded x{world}"},
    }
    traffic = None
    try:  # HBM bytes per launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction); see profiles/
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            traffic = round(json.load(f)["hbm_fetch_bytes_per_launch"])
    except Exception:
        pass
    if cnt[0] > 0 and ms[0] > 0:
        ach = work[0] / (ms[0] * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "gemm_skinny2 / gemm_skinny3 (weight-streaming GEMMs of the 7B decode passes and lm_head, M = 32, >= 16 MB of weights each)",
                           "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                           "traffic": traffic, "algorithmic_bytes_per_launch": round(work[0] / cnt[0]), "launches": int(cnt[0]), "avg_launch_us": round(1e3 * ms[0] / cnt[0], 2),
                           "algorithmic_bytes_per_decision": work[0], "kernel_ms_per_decision": round(ms[0], 3)}
    if cnt[1] > 0 and ms[1] > 0:
        tf = work[1] / (ms[1] * 1e-3) / 1e12
        out["mfma_kernels"] = {"kernel": "gemm_tiled (prefill / ViT GEMMs)", "achieved": round(tf, 1), "peak": MFMA_PEAK_TF,
                               "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TF, 4), "launches": int(cnt[1]),
                               "kernel_ms_per_decision": round(ms[1], 3)}
    if cnt[3] > 0:
        out["small_streaming_gemms"] = {"kernel": "gemm_skinny2 launches with < 16 MB of weights (verifier text tower etc.)", "launches": int(cnt[3]),
                                        "kernel_ms_per_decision": round(ms[3], 3)}
    if cnt[2] > 0:
        out["attention_kernels"] = {"launches": int(cnt[2]), "kernel_ms_per_decision": round(ms[2], 3)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pipe.c, pipe.sc)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
