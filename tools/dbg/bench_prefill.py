#!/usr/bin/env python3
"""Prefill GEMMs of a 7B decoder at M = 448 (256 patch rows + 8 x 24 text rows), weights rotating over enough copies to stay out of
the Infinity Cache: us per launch and TFLOP/s per shape, plus the sum over a layer (what roofline_mfma.prefill measures).
Usage: python tools/dbg/bench_prefill.py [M] [reps]   (environment knobs of gemm_bf16.hip apply: COVER_TILE_PICK, COVER_LIB_PATH ...)
FP8=1: e4m3 weights + pre-quantised e4m3 activation rows (the config-5 decode GEMMs at M = 512; the quantiser launch is timed on its own)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cover_vla_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 448
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
shapes = [("qkv", 12288, 4096, False, False), ("o_proj", 4096, 4096, False, True), ("gate_up", 22016, 4096, True, False), ("down", 4096, 11008, False, True)]
if os.environ.get("SHAPES") == "pi0":   # Gemma-2B prefix pass of the pi0 policy (M = 8 prompts x 328 tokens = 2624)
    shapes = [("qkv", 2560, 2048, False, False), ("o_proj", 2048, 2048, False, True), ("gate_up", 32768, 2048, True, False), ("down", 2048, 16384, False, True)]
if os.environ.get("SHAPE"):
    shapes = [x for x in shapes if x[0] in os.environ["SHAPE"].split(",")]
tot_us, tot_fl = 0.0, 0.0
for name, N, K, glu, norm in shapes:
    g = torch.Generator(device=dev).manual_seed(N + K)
    copies = max(2, int(600e6 // (N * K * 2)) + 1)
    f8 = os.environ.get("FP8") == "1"
    lins = [ops.pack_linear((torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16(), glu=glu, fp8=f8) for _ in range(copies)]
    a = torch.randn(M, lins[0].kp, device=dev, generator=g).bfloat16()
    a8 = ops.quantize_act_fp8(a, K) if f8 else None
    out = torch.empty(M, lins[0].n_out, dtype=torch.bfloat16, device=dev)
    res = torch.randn(M, N, device=dev, generator=g).bfloat16() if norm else None
    kw = dict(norm_w=torch.ones(N, device=dev), norm_out=torch.empty(M, N, dtype=torch.bfloat16, device=dev), norm_style=1, norm_eps=1e-5) if norm else {}
    ws = ops.gemm_workspace(M, N, K, dev)
    run = lambda i: ops.gemm(a, lins[i % copies], act="silu" if glu else "none", out=out, ws=ws, residual=res, a8=a8, **kw)
    for i in range(copies):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = reps * copies
    e0.record()
    for i in range(n):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 2.0 * M * N * K
    tot_us += us
    tot_fl += fl
    pr = ops.gemm_probe()
    if f8:
        ops.gemm_plan_counts(reset=True)
        run(0)
        pc = ops.gemm_plan_counts()
        e0.record()
        for i in range(50):
            ops.quantize_act_fp8(a, K)
        e1.record()
        torch.cuda.synchronize()
        print(f"M={M} {name:8s} {us:7.1f} us  {fl / us / 1e6:6.0f} TF   plans {[i for i, v in enumerate(pc) if v]}   quantiser of its input {e0.elapsed_time(e1) / 50 * 1e3:.1f} us", flush=True)
        del lins
        torch.cuda.empty_cache()
        continue
    print(f"M={M} {name:8s} {us:7.1f} us  {fl / us / 1e6:6.0f} TF  (incl. its split-K reduction / norm launch)   [last v3 launch, workgroup 0: prologue {pr['prologue_us']:.1f} "
          f"loop {pr['loop_us']:.1f} epilogue {pr['epilogue_us']:.1f} us {[round(x, 1) for x in pr['epilogue_split_us']] if pr['epilogue_split_us'] else ''}, {pr['k_tiles']} k-tiles x {pr['cycles_per_k_tile']:.0f} cycles at {pr['clock_ghz']:.2f} GHz]", flush=True)
    del lins
    torch.cuda.empty_cache()
pk = 5000 if os.environ.get("FP8") == "1" else 2500
print(f"layer: {tot_us:7.1f} us  {tot_fl / tot_us / 1e6:6.0f} TF = {tot_fl / tot_us / 1e6 / pk:.3f} of {pk / 1000} PF; x32 layers = {tot_us * 32 / 1e3:.2f} ms")
