#!/usr/bin/env python3
"""o_proj of a config-5 decode pass (M = 512, N = K = 4096, MX block-scaled input, residual + post-attention RMSNorm) launch to launch over cold weights,
under whatever COVER_TILE_PICK / COVER_TILE_SPLIT the caller set (both are read once per process): the default plan is the 64 x 128 loader-wave tile
unsplit + a norm launch; a split plan folds the norm into the reduction."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cover_vla_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("M", 512)), 4096, int(os.environ.get("K", 4096))
g = torch.Generator(device=dev).manual_seed(1)
copies = 24
lins = [ops.pack_linear(torch.randn(N, K, device=dev, generator=g) * 0.02, fp8=True, klinear=True) for _ in range(copies)]
a = torch.randn(M, lins[0].kp, device=dev, generator=g).bfloat16()
q, mx = ops.quantize_act_fp8_mx(a, K)
x = torch.randn(M, N, device=dev, generator=g).bfloat16()
nw = torch.rand(N, device=dev, generator=g) + 0.5
h = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
ws = ops.gemm_workspace(M, N, K, dev)
if ws is None or ws.numel() * 4 < 8 * M * N * 4:
    ws = torch.empty(8 * M * N, dtype=torch.float32, device=dev)
run = lambda i: ops.gemm(a, lins[i % copies], residual=x, out=x, ws=ws, a8=(q, mx), norm_w=nw, norm_out=h, norm_style=1, norm_eps=1e-5)
ops.gemm_plan_counts(reset=True)
for i in range(copies):
    run(i)
torch.cuda.synchronize()
c = ops.gemm_plan_counts()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10 * copies
e0.record()
for i in range(reps):
    run(i)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"pick={os.environ.get('COVER_TILE_PICK', 'auto'):4s} split={os.environ.get('COVER_TILE_SPLIT', 'auto'):4s} M={M} K={K}: {us:6.1f} us per o_proj (+ norm), "
      f"{2.0 * M * N * K / us / 1e6:6.0f} TF; fp8 tile launches {c[21]}/{copies}")
