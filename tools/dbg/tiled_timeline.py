"""Per-block timeline of gemm_tiled (the 64x64 / 64x128 / 128x128 LDS-ring kernel of the ViT-sized GEMMs; library built with
-DCOVER_PC_DEBUG): start / loop start / loop end / end, and inside the staged epilogue: barrier, LDS fill, barrier, stores."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cover_vla_amd import ops, _lib as L
dev = torch.device("cuda:0")
fn = L.lib().cover_pc_timeline
fn.argtypes = [C.c_void_p]
for M, N, K in [(261, 3072, 1024), (261, 4096, 1024), (256, 4608, 1152), (261, 1024, 1024), (576, 4096, 1024)]:
    g = torch.Generator(device=dev).manual_seed(N)
    bias = torch.randn(N, device=dev, generator=g)
    lins = [ops.pack_linear((torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16(), bias) for _ in range(8)]
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for i in range(8): ops.gemm(a, lins[i], out=o, act="gelu_tanh")
    torch.cuda.synchronize()
    buf = np.zeros(8192, dtype=np.uint64)
    ops.gemm(a, lins[0], out=o, act="gelu_tanh")
    torch.cuda.synchronize()
    fn(buf.ctypes.data)
    t = buf.reshape(1024, 8).astype(np.float64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0
    q = lambda x: f"{np.percentile(x, 0):.1f}/{np.percentile(x, 50):.1f}/{np.percentile(x, 100):.1f}"
    print(f"M={M} N={N} K={K} blocks={len(t)}: start {q(t[:,0])}  loop-start {q(t[:,1])}  loop-end {q(t[:,2])}  end {q(t[:,3])} (us min/median/max)  "
          f"loop {q(t[:,2]-t[:,1])}  epilogue {q(t[:,3]-t[:,2])} | barrier1 {q(t[:,4]-t[:,2])} fill {q(t[:,5]-t[:,4])} barrier2 {q(t[:,6]-t[:,5])} stores {q(t[:,3]-t[:,6])}", flush=True)
