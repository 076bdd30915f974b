"""fp32 GEMM (verifier heads) micro-benchmark at the ensemble shapes, hipGraph-captured."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cover_vla_amd import ops
dev = torch.device("cuda:0")
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
SHAPES = [(5120, 1536, 512), (5120, 512, 512), (5120, 2048, 512), (5120, 512, 2048)] if os.environ.get('BIG') else None
for M, N, K in SHAPES or [(320, 1536, 512), (320, 512, 512), (320, 2048, 512), (320, 512, 2048), (320, 512, 7), (64, 4096, 1024), (64, 576, 1024), (1, 1024, 1024), (1, 4096, 1024)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    f = lambda: ops.gemm_f32(a, w, bias=b)
    f(); torch.cuda.synchronize()
    with ops.Graph() as g:
        for _ in range(20): f()
    g.launch(); torch.cuda.synchronize()
    t = ops.Timer(); t.start()
    for _ in range(5): g.launch()
    ms = t.stop() / 100
    print(f"M={M} N={N} K={K}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.2f} TF", flush=True)
