"""Tiled-GEMM tile-configuration experiment at the OpenVLA-7B prefill shapes (M = 449). One process per COVER_TILE_PICK:
   for p in 0 1 2 3 4 5 6; do COVER_TILE_PICK=$p python tools/dbg/exp_tiles.py; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cover_vla_amd import ops

dev = torch.device("cuda:0")
M = int(os.environ.get("M", "448"))
pick = os.environ.get("COVER_TILE_PICK", "auto")
row = []
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
for K, N in [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]:
    glu = N == 22016
    g = torch.Generator(device=dev).manual_seed(N + K)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    lin = ops.pack_linear(w, glu=glu)
    # COLD=1: rotate over enough distinct weight copies (> 512 MB) that neither L2 nor the 256 MB Infinity Cache holds them
    ncopy = max(2, int(600e6 // (2 * N * K)) + 1) if os.environ.get("COLD", "0") == "1" else 1
    lins = [lin] + [ops.pack_linear((torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16(), glu=glu) for _ in range(ncopy - 1)]
    pad = int(os.environ.get("LDA_PAD", "0"))   # break power-of-two row strides of the activation panel
    a = torch.randn(M, K + pad, device=dev, generator=g).bfloat16()[:, :K]
    o = torch.empty(M, lin.n_out, dtype=torch.bfloat16, device=dev)
    ws = ops.gemm_workspace(M, N, K, dev)
    f = lambda: ops.gemm(a, lin, act="silu" if glu else "none", out=o, variant=int(os.environ.get("VARIANT", "1")), ws=ws)
    f(); torch.cuda.synchronize()
    # correctness against fp32 matmul of the same bf16 operands
    y = a.float().contiguous() @ w.float().T
    ref = torch.nn.functional.silu(y[:, : N // 2]) * y[:, N // 2:] if glu else y
    if glu:   # packed GLU interleaves gate/up in 16-row blocks of the ORIGINAL [gate; up] stacking
        ref = torch.nn.functional.silu(y[:, : N // 2]) * y[:, N // 2:]
    err = ((o.float() - ref).norm() / ref.norm()).item()
    reps = max(10, 2 * ncopy)
    with ops.Graph() as gr:
        for i in range(reps):
            ops.gemm(a, lins[i % ncopy], act="silu" if glu else "none", out=o, variant=int(os.environ.get("VARIANT", "1")), ws=ws)
    gr.launch(); torch.cuda.synchronize()
    t = ops.Timer(); t.start()
    for _ in range(5):
        gr.launch()
    ms = t.stop() / (5 * reps)
    row.append(f"N={N:5d} K={K:5d}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.0f} TF err={err:.1e}")
    del w, lin, lins
print(f"pick={pick} M={M} | " + " | ".join(row), flush=True)
