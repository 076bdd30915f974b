"""M = 200 GEMMs of the pi0 action expert (qkv 2560 x 1024, gate_up 8192 x 1024 GLU, o_proj 1024 x 2048 + residual, down 1024 x 4096 + residual):
us per launch (incl. the reduction launch of a split plan) for the tile / split the environment selects; weights rotate over 18 copies (the
expert's 18 layers). Usage: COVER_TILE_PICK=.. COVER_TILE_SPLIT=.. python tools/dbg/r05/bench_m200.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from cover_vla_amd import ops
dev = torch.device("cuda:0")
M = 200
shapes = [("qkv", 2560, 1024, False, False), ("gate_up", 8192, 1024, True, False), ("o_proj", 1024, 2048, False, True), ("down", 1024, 4096, False, True)]
only = os.environ.get("ONLY")
for name, N, K, glu, res in shapes:
    if only and name not in only.split(","):
        continue
    g = torch.Generator(device=dev).manual_seed(N + K)
    lins = [ops.pack_linear(torch.randn(N, K, device=dev, generator=g) * 0.02, glu=glu) for _ in range(18)]
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    out = torch.empty(M, lins[0].n_out, dtype=torch.bfloat16, device=dev)
    r = torch.randn(M, N, device=dev, generator=g).bfloat16() if res else None
    kw = dict(norm_w=torch.ones(N, device=dev), norm_out=torch.empty(M, N, dtype=torch.bfloat16, device=dev), norm_style=0, norm_w_offset=1.0, norm_eps=1e-6) if (res and os.environ.get("NORM", "1") == "1") else {}
    if res and os.environ.get("SSQ") == "1":
        kw = dict(ssq_out=torch.empty(M, N // 32, device=dev))
    ws = ops.gemm_workspace(M, N, K, dev)
    run = lambda i: ops.gemm(a, lins[i % 18], act="gelu_tanh" if glu else "none", out=out, ws=ws, residual=r, **kw)
    for i in range(18):
        run(i)
    ops.gemm_plan_counts(reset=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 18 * 20
    e0.record()
    for i in range(n):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    plans = [(i, c // n) for i, c in enumerate(ops.gemm_plan_counts()) if c]
    print(f"{name:8s} {e0.elapsed_time(e1) / n * 1e3:7.2f} us   plans {plans}", flush=True)
