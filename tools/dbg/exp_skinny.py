"""Experiment harness: weight-streaming GEMM at the decode shapes. The launches are captured into ONE hipGraph over a
rotation of distinct weight copies (> 256 MiB so the Infinity Cache cannot hold the stream; no host launch overhead in
the timing). Prints per shape: us per GEMM (kernel + reduce) and GB/s of weight bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cover_vla_amd import ops

dev = torch.device("cuda:0")
M = int(os.environ.get("EXP_M", "32"))
variant = int(os.environ.get("EXP_VARIANT", "3"))
res = []
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for K, N in [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096), (4096, 32064)]:
        glu = N == 22016
        ncopy = 1 if os.environ.get("WARM", "0") == "1" else max(2, int(600e6 // (2 * N * K)) + 1)
        lins = [ops.pack_linear((torch.randn(N, K, device=dev) * 0.02).bfloat16(), glu=glu) for _ in range(ncopy)]
        a = torch.randn(M, K, device=dev).bfloat16()
        o = torch.empty(M, lins[0].n_out, dtype=torch.bfloat16, device=dev)
        ws = ops.gemm_workspace(M, N, K, dev)
        reps = 4 * ncopy
        def body():
            for i in range(reps):
                ops.gemm(a, lins[i % ncopy], act="silu" if glu else "none", out=o, variant=variant, ws=ws)
        body()
        torch.cuda.synchronize()
        with ops.Graph() as g:
            body()
        g.launch(); torch.cuda.synchronize()
        t = ops.Timer(); t.start()
        for _ in range(5): g.launch()
        ms = t.stop() / (5 * reps)
        res.append((N, K, round(ms * 1e3, 2), round(2.0 * N * K / ms / 1e6)))
        del lins
print("variant", variant, {k: v for k, v in os.environ.items() if k.startswith("COVER_")}, res)
