#!/usr/bin/env python3
"""A/B of the LDS-tiled GEMMs on bf16 MFMA vs the MX-scaled fp8 matrix instruction at the config-5 shapes (M = 512 decode rows,
M = 448 prefill rows of a 7B decoder). Weights rotate over several copies so that they are not Infinity-Cache resident."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cover_vla_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [("qkv", 12288, 4096, False), ("o_proj", 4096, 4096, False), ("gate_up", 22016, 4096, True), ("down", 4096, 11008, False)]
for M in (512, 448):
    for name, N, K, glu in shapes:
        g = torch.Generator(device=dev).manual_seed(N + K)
        copies = max(2, int(600e6 // (N * K * 3)) + 1)
        lins = [ops.pack_linear(torch.randn(N, K, device=dev, generator=g) * 0.02, glu=glu, fp8=True) for _ in range(copies)]
        a = torch.randn(M, lins[0].kp, device=dev, generator=g).bfloat16()
        q, sc = ops.quantize_act_fp8(a, K)
        out = torch.empty(M, lins[0].n_out, dtype=torch.bfloat16, device=dev)
        ws = ops.gemm_workspace(M, N, K, dev)
        res = {}
        for tag, a8 in (("bf16", None), ("fp8", (q, sc))):
            for i in range(copies):
                ops.gemm(a, lins[i], act="silu" if glu else "none", out=out, ws=ws, a8=a8)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 6 * copies
            e0.record()
            for i in range(reps):
                ops.gemm(a, lins[i % copies], act="silu" if glu else "none", out=out, ws=ws, a8=a8)
            e1.record()
            torch.cuda.synchronize()
            res[tag] = e0.elapsed_time(e1) / reps * 1e3
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(50):
            ops.quantize_act_fp8(a, K)
        t1.record(); torch.cuda.synchronize()
        fl = 2.0 * M * N * K
        print(f"M={M:4d} {name:8s} bf16 {res['bf16']:7.1f} us ({fl / res['bf16'] / 1e6:6.0f} TF)   fp8 {res['fp8']:7.1f} us ({fl / res['fp8'] / 1e6:6.0f} TF)   "
              f"quantise {t0.elapsed_time(t1) / 50 * 1e3:5.1f} us", flush=True)
        del lins
        torch.cuda.empty_cache()
