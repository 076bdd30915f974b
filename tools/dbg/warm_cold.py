"""Weight-streaming GEMMs (M = 32) on weights that are resident in the 256 MB Infinity Cache (same buffer every launch) vs cold
(rotating over > 600 MB of copies): does cache residency change the kernel's duration? Kernel timestamps (cover_profile_*)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cover_vla_amd import ops, _lib as L
dev = torch.device("cuda:0")
M = 32
h = L.lib()
for K, N in [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]:
    glu = N == 22016
    g = torch.Generator(device=dev).manual_seed(N + K)
    ncopy = int(700e6 // (2 * N * K)) + 1
    lins = [ops.pack_linear((torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16(), glu=glu) for _ in range(ncopy)]
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    o = torch.empty(M, lins[0].n_out, dtype=torch.bfloat16, device=dev)
    ws = ops.gemm_workspace(M, N, K, dev)
    for mode in ("cold", "warm"):
        for i in range(ncopy): ops.gemm(a, lins[i if mode == "cold" else 0], act="silu" if glu else "none", out=o, ws=ws)
        torch.cuda.synchronize()
        n = 7
        ms, cnt, work = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
        L.check(h.cover_profile_begin(4096), "b")
        reps = 4 * ncopy
        for i in range(reps): ops.gemm(a, lins[(i % ncopy) if mode == "cold" else 0], act="silu" if glu else "none", out=o, ws=ws)
        L.check(h.cover_profile_end_n(ms, cnt, work, n), "e")
        t = (ms[0] + ms[3]) / reps * 1e3
        r = (ms[5] + ms[6]) / reps * 1e3
        print(f"{mode} N={N} K={K}: stream {t:.2f} us ({2.0*N*K/(t*1e-6)/1e12:.2f} TB/s) + reduce {r:.2f} us", flush=True)
    del lins
