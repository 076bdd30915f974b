#!/usr/bin/env python3
"""One fp8 tiled GEMM shape in a loop, for a rocprofv3 --pmc pass (tools/pmc_any.py reads the database)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cover_vla_amd import ops
dev = torch.device("cuda:0")
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
f8 = len(sys.argv) < 5 or sys.argv[4] == "fp8"
g = torch.Generator(device=dev).manual_seed(1)
lins = [ops.pack_linear(torch.randn(N, K, device=dev, generator=g) * 0.02, fp8=True) for _ in range(4)]
a = torch.randn(M, lins[0].kp, device=dev, generator=g).bfloat16()
q, sc = ops.quantize_act_fp8(a, K)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
ws = ops.gemm_workspace(M, N, K, dev)
for i in range(12):
    ops.gemm(a, lins[i % 4], out=out, ws=ws, a8=(q, sc) if f8 else None)
torch.cuda.synchronize()
