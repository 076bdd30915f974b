"""Per-block timeline of gemm_tiled (the 64x64 / 64x128 / 128x128 LDS-ring kernel of the ViT-sized GEMMs; library built with
-DCOVER_PC_DEBUG): start / loop start / loop end / end, and inside the staged epilogue: barrier, LDS fill, barrier, stores."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cover_vla_amd import ops, _lib as L
dev = torch.device("cuda:0")
fn = L.lib().cover_pc_timeline
fn.argtypes = [C.c_void_p]
# EXPERT=1: the pi0 action expert's four projections at M = 200 (qkv with an fp32 slab output as the decoder requests it, o_proj / down with the
# residual + RMSNorm epilogue through split-K, gate_up with GLU): what the 8 launches per layer-step of the denoise loop look like from inside
expert = os.environ.get("EXPERT") == "1"
cases = ([(200, 2560, 1024, "f32"), (200, 1024, 2048, "norm"), (200, 8192, 1024, "glu"), (200, 1024, 4096, "norm")] if expert else
         [(261, 3072, 1024, "bias"), (261, 4096, 1024, "bias"), (256, 4608, 1152, "bias"), (261, 1024, 1024, "bias"), (576, 4096, 1024, "bias")])
for M, N, K, kind in cases:
    g = torch.Generator(device=dev).manual_seed(N)
    bias = torch.randn(N, device=dev, generator=g) if kind == "bias" else None
    lins = [ops.pack_linear((torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16(), bias, glu=kind == "glu") for _ in range(8)]
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    o = torch.empty(M, lins[0].n_out, dtype=torch.float32 if kind == "f32" else torch.bfloat16, device=dev)
    ws = ops.gemm_workspace(M, N, K, dev)
    kw = dict(act="gelu_tanh") if kind in ("bias", "glu") else {}
    if kind == "norm":
        kw = dict(residual=o, norm_w=torch.ones(N, device=dev), norm_out=torch.empty(M, N, dtype=torch.bfloat16, device=dev), norm_style=0, norm_w_offset=1.0, norm_eps=1e-6)
    for i in range(8): ops.gemm(a, lins[i], out=o, ws=ws, **kw)
    torch.cuda.synchronize()
    buf = np.zeros(8192, dtype=np.uint64)
    ops.gemm_plan_counts(reset=True)
    ops.gemm(a, lins[0], out=o, ws=ws, **kw)
    torch.cuda.synchronize()
    print(kind, "plans", [i for i, v in enumerate(ops.gemm_plan_counts()) if v], end="  ")
    fn(buf.ctypes.data)
    t = buf.reshape(1024, 8).astype(np.float64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0
    q = lambda x: f"{np.percentile(x, 0):.1f}/{np.percentile(x, 50):.1f}/{np.percentile(x, 100):.1f}"
    print(f"M={M} N={N} K={K} blocks={len(t)}: start {q(t[:,0])}  loop-start {q(t[:,1])}  loop-end {q(t[:,2])}  end {q(t[:,3])} (us min/median/max)  "
          f"loop {q(t[:,2]-t[:,1])}  epilogue {q(t[:,3]-t[:,2])} | barrier1 {q(t[:,4]-t[:,2])} fill {q(t[:,5]-t[:,4])} barrier2 {q(t[:,6]-t[:,5])} stores {q(t[:,3]-t[:,6])}", flush=True)
