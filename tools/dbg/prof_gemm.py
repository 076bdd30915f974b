"""Kernel-timestamp duration (cover_profile_*) of one tiled GEMM shape under COVER_TILE_PICK / COVER_TILE_SPLIT: M N K [glu]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cover_vla_amd import ops, _lib as L
dev = torch.device("cuda:0")
M = int(os.environ.get("M", "448"))
h = L.lib()
for K, N in [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]:
    glu = N == 22016
    g = torch.Generator(device=dev).manual_seed(N + K)
    ncopy = int(600e6 // (2 * N * K)) + 1
    lins = [ops.pack_linear((torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16(), glu=glu) for _ in range(ncopy)]
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    o = torch.empty(M, lins[0].n_out, dtype=torch.bfloat16, device=dev)
    ws = ops.gemm_workspace(M, N, K, dev)
    for i in range(ncopy): ops.gemm(a, lins[i], act="silu" if glu else "none", out=o, variant=1, ws=ws)
    torch.cuda.synchronize()
    n = 7
    ms, cnt, work = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
    L.check(h.cover_profile_begin(4096), "b")
    reps = 3 * ncopy
    for i in range(reps): ops.gemm(a, lins[i % ncopy], act="silu" if glu else "none", out=o, variant=1, ws=ws)
    L.check(h.cover_profile_end_n(ms, cnt, work, n), "e")
    t = (ms[4] + ms[1]) / reps * 1e3
    r = ms[6] / reps * 1e3
    print(f"pick={os.environ.get('COVER_TILE_PICK','auto')} split={os.environ.get('COVER_TILE_SPLIT','-')} N={N} K={K}: gemm {t:.1f} us + reduce {r:.1f} us -> {2.0*M*N*K/((t+r)*1e-6)/1e12:.0f} TF", flush=True)
    del lins
