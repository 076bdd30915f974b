#!/usr/bin/env python3
"""Decode-row GEMMs of a 7B decoder at M = 32 (weight streaming), cold weights (rotating copies): o_proj as it runs today (split-K slabs +
the reduction / residual / RMSNorm launch) against the unsplit third-generation kernel with the residual in its epilogue and NO norm launch
(what a deferred RMSNorm would leave of it); gate_up and down for scale. us per call, launch to launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from cover_vla_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def bench(name, N, K, glu, mode):
    g = torch.Generator(device=dev).manual_seed(N + K)
    copies = max(2, int(600e6 // (N * K * 2)) + 1)
    lins = [ops.pack_linear(torch.randn(N, K, device=dev, generator=g) * 0.02, glu=glu) for _ in range(copies)]
    a = torch.randn(M, lins[0].kp, device=dev, generator=g).bfloat16()
    out = torch.empty(M, lins[0].n_out, dtype=torch.bfloat16, device=dev)
    res = torch.randn(M, N, device=dev, generator=g).bfloat16() if not glu else None
    ws = ops.gemm_workspace(M, N, K, dev)
    if mode == "split+norm":
        kw = dict(norm_w=torch.ones(N, device=dev), norm_out=torch.empty(M, N, dtype=torch.bfloat16, device=dev), norm_style=1, norm_eps=1e-5)
        run = lambda i: ops.gemm(a, lins[i % copies], out=out, ws=ws, residual=res, **kw)
    elif mode == "unsplit":
        run = lambda i: ops.gemm(a, lins[i % copies], out=out, ws=ws, residual=res, variant=6)
    elif mode == "unsplit+norm":
        kw = dict(norm_w=torch.ones(N, device=dev), norm_out=torch.empty(M, N, dtype=torch.bfloat16, device=dev), norm_style=1, norm_eps=1e-5)
        run = lambda i: ops.gemm(a, lins[i % copies], out=out, ws=ws, residual=res, variant=6, **kw)
    else:
        run = lambda i: ops.gemm(a, lins[i % copies], act="silu" if glu else "none", out=out, ws=ws)
    for i in range(copies):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 6 * copies
    ops.gemm_plan_counts(reset=True)
    e0.record()
    for i in range(n):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    pc = ops.gemm_plan_counts()
    print(f"M={M} {name:8s} {mode:13s} {us:7.1f} us  {N * K * 2 / us / 1e6:5.2f} TB/s   plans {[i for i, v in enumerate(pc) if v]}", flush=True)


bench("o_proj", 4096, 4096, False, "split+norm")
bench("o_proj", 4096, 4096, False, "unsplit")
bench("o_proj", 4096, 4096, False, "unsplit+norm")
bench("down", 4096, 11008, False, "split+norm")
bench("down", 4096, 11008, False, "unsplit")
bench("gate_up", 22016, 4096, True, "plain")
bench("qkv", 12288, 4096, False, "plain")
