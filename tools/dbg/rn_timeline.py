"""Per-block timeline of splitk_reduce_norm at the decode shape (M = 32 rows, N = 4096, 4 slabs, bf16 residual, Llama RMSNorm),
behind a weight-streaming GEMM as in a decode layer (library built with -DCOVER_RN_DEBUG, loaded through COVER_LIB_PATH)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cover_vla_amd import ops, _lib as L
dev = torch.device("cuda:0")
fn = L.lib().cover_rn_debug
fn.argtypes = [C.c_void_p]
M, N, K = 32, 4096, 4096
g = torch.Generator(device=dev).manual_seed(0)
lins = [ops.pack_linear((torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()) for _ in range(12)]
a = torch.randn(M, K, device=dev, generator=g).bfloat16()
x = torch.randn(M, N, device=dev, generator=g).bfloat16()
h = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
nw = torch.ones(N, device=dev)
ws = ops.gemm_workspace(M, N, K, dev)
for rep in range(3):
    for i in range(12):
        ops.gemm(a, lins[i], residual=x, out=x, ws=ws, norm_w=nw, norm_out=h, norm_style=1, norm_eps=1e-5)
    torch.cuda.synchronize()
    buf = np.zeros(4096, dtype=np.uint64); fn(buf.ctypes.data)
    t = buf.reshape(512, 8).astype(np.float64) / 100.0
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    q = lambda v: f"{np.percentile(v - t0, 5):5.1f}/{np.median(v - t0):5.1f}/{(v - t0).max():5.1f}"
    print(f"rep {rep}: blocks {len(t)}  start {q(t[:,0])}  slabs {q(t[:,1])}  epi {q(t[:,2])}  C-stored/q {q(t[:,3])}  sum {q(t[:,4])}  end {q(t[:,5])}  (p5/median/max us)", flush=True)
