"""Per-block timeline of the third-generation weight-streaming kernel (library built with -DCOVER_SK_DEBUG):
start spread, first chunk staged, last MFMA, end -- relative to the earliest block start. EXP_SHAPE=qkv|gate_up|down|lm_head"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cover_vla_amd import ops, _lib as L
dev = torch.device("cuda:0")
shapes = {"qkv": (4096, 12288, False), "gate_up": (4096, 22016, True), "down": (11008, 4096, False), "lm_head": (4096, 32064, False)}
fn = L.lib().cover_sk_debug
fn.argtypes = [C.c_void_p]
for name, (K, N, glu) in shapes.items():
    lins = [ops.pack_linear((torch.randn(N, K, device=dev) * 0.02).bfloat16(), glu=glu) for _ in range(3)]
    a = torch.randn(32, K, device=dev).bfloat16()
    o = torch.empty(32, lins[0].n_out, dtype=torch.bfloat16, device=dev)
    ws = ops.gemm_workspace(32, N, K, dev)
    flush = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
    xin = torch.randn(32, K, device=dev).bfloat16(); nwt = torch.ones(K, device=dev)
    for rep in range(3):
        if os.environ.get("MODE", "cold") == "cold":
            flush.zero_(); torch.cuda.synchronize()
        else:   # MODE=pipe: as inside a decode layer -- the activations come out of the previous kernel, only the weights are cold
            flush.zero_(); torch.cuda.synchronize()
            ops.gemm(a, lins[(rep + 1) % 3], act="silu" if glu else "none", out=o, variant=3, ws=ws)
            ops.rmsnorm(xin, nwt, 1e-5, style=1, out=a)
        ops.gemm(a, lins[rep], act="silu" if glu else "none", out=o, variant=3, ws=ws)
        torch.cuda.synchronize()
        buf = np.zeros(4096, dtype=np.uint64); fn(buf.ctypes.data)
        t = buf.reshape(1024, 4).astype(np.float64) / 100.0
        t = t[t[:, 0] > 0][:256 if name != "gate_up" else 230]
        t0 = t[:, 0].min()
        q = lambda x: f"{np.percentile(x - t0, 5):5.1f}/{np.median(x - t0):5.1f}/{(x - t0).max():5.1f}"
        print(f"{name:8s} rep {rep}: blocks {len(t)}  start {q(t[:,0])}  chunk0 staged {q(t[:,1])}  last MFMA {q(t[:,2])}  end {q(t[:,3])}   (p5/median/max us)")
