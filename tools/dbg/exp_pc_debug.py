"""Cycle breakdown inside gemm_tiled_pc (library built with -DCOVER_PC_DEBUG): loader wave 0 and MFMA wave 0 of every block.
Usage: COVER_TILE_PICK=d M=2624 python tools/dbg/exp_pc_debug.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cover_vla_amd import ops, _lib as L
dev = torch.device("cuda:0")
M = int(os.environ.get("M", "2624"))
fn = L.lib().cover_pc_debug
fn.argtypes = [C.c_void_p, C.c_int]
for K, N in [(4096, 12288), (4096, 22016), (11008, 4096)]:
    g = torch.Generator(device=dev).manual_seed(N)
    lins = [ops.pack_linear((torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()) for _ in range(4)]
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for i in range(4): ops.gemm(a, lins[i], out=o, variant=1)
    torch.cuda.synchronize()
    buf = np.zeros(8, dtype=np.uint64); fn(buf.ctypes.data, 1)
    reps = 8
    for i in range(reps): ops.gemm(a, lins[i % 4], out=o, variant=1)
    torch.cuda.synchronize()
    fn(buf.ctypes.data, 1)
    b = buf.astype(np.float64); kt = b[5]
    print(f"N={N} K={K} M={M}: per k-tile cycles  loader: wait-data {b[0]/kt:.0f} barrier {b[1]/kt:.0f} issue {b[2]/kt:.0f} | consumer: barrier {b[3]/kt:.0f} total {b[4]/kt:.0f} | clock {b[4]/max(b[6],1)*100:.0f} MHz (s_memtime per 100 MHz wall tick)")
