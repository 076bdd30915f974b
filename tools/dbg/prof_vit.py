"""Kernel-timestamp durations of the ViT-sized tiled GEMMs (M = 256 / 261 / 576) under COVER_TILE_PICK / COVER_TILE_SPLIT."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cover_vla_amd import ops, _lib as L
dev = torch.device("cuda:0")
h = L.lib()
shapes = [(256, 4608, 1152), (256, 1152, 1536), (256, 4352, 1152), (256, 1152, 4352), (261, 3072, 1024), (261, 1024, 1024), (261, 4096, 1024), (261, 1024, 4096),
          (576, 3072, 1024), (576, 1024, 1024), (576, 4096, 1024), (576, 1024, 4096)]
tot = 0.0
for M, N, K in shapes:
    g = torch.Generator(device=dev).manual_seed(N + K)
    lins = [ops.pack_linear((torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16(), torch.randn(N, device=dev, generator=g)) for _ in range(8)]
    a = torch.randn(M, lins[0].kp, device=dev, generator=g).bfloat16()
    res = torch.randn(M, N, device=dev, generator=g).bfloat16()
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ws = ops.gemm_workspace(M, N, K, dev)
    kw = dict(residual=res) if N <= 1152 else dict(act="gelu_tanh")
    for i in range(8): ops.gemm(a, lins[i], out=o, variant=1, ws=ws, **kw)
    torch.cuda.synchronize()
    n = 7
    ms, cnt, work = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
    L.check(h.cover_profile_begin(4096), "b")
    reps = 24
    for i in range(reps): ops.gemm(a, lins[i % 8], out=o, variant=1, ws=ws, **kw)
    L.check(h.cover_profile_end_n(ms, cnt, work, n), "e")
    t = (ms[4] + ms[1]) / reps * 1e3
    r = ms[6] / reps * 1e3
    tot += t + r
    print(f"pick={os.environ.get('COVER_TILE_PICK','auto')} split={os.environ.get('COVER_TILE_SPLIT','-')} M={M} N={N} K={K}: gemm {t:.1f} + reduce {r:.1f} us -> {2.0*M*N*K/((t+r)*1e-6)/1e12:.0f} TF", flush=True)
print(f"pick={os.environ.get('COVER_TILE_PICK','auto')} split={os.environ.get('COVER_TILE_SPLIT','-')} TOTAL {tot:.1f} us")
