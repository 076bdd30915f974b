"""Per-launch budget of one candidate-decode layer-step (M = 32 rows, Llama-2-7B shapes): for each weight-streaming launch its weight bytes,
its launch-to-launch duration inside a chain of the layer's launches (hipEvents around 20 repetitions over rotating cold weight copies),
the implied TB/s, and -- from the in-kernel stamps of gemm_skinny3 (library built with -DCOVER_SK_DEBUG, COVER_LIB_PATH) -- where a
workgroup's time goes: start -> first activation chunk staged (prologue: dispatch, activation panel 32 x 1024 through LDS, first weight
window requested) -> last MFMA (the stream) -> end (k-slice reduction through LDS + epilogue / slab store).
Usage: COVER_LIB_PATH=tools/ab/libcover_hip_dbg.so python tools/decode_budget.py > profiles/r06_decode_layer_budget.txt"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cover_vla_amd import ops, _lib as L

dev = torch.device("cuda:0")
shapes = [("qkv", 4096, 12288, False, False), ("o_proj", 4096, 4096, False, True), ("gate_up", 4096, 22016, True, False), ("down", 11008, 4096, False, True)]
try:
    fn = L.lib().cover_sk_debug
    fn.argtypes = [C.c_void_p]
except AttributeError:
    fn = None
M = 32
rows = []
for name, K, N, glu, norm in shapes:
    g = torch.Generator(device=dev).manual_seed(N + K)
    copies = max(3, int(700e6 // (N * K * 2)) + 1)
    lins = [ops.pack_linear((torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16(), glu=glu) for _ in range(copies)]
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    o = torch.empty(M, lins[0].n_out, dtype=torch.bfloat16, device=dev)
    res = torch.randn(M, N, device=dev, generator=g).bfloat16() if norm else None
    kw = dict(norm_w=torch.ones(N, device=dev), norm_out=torch.empty(M, N, dtype=torch.bfloat16, device=dev), norm_style=1, norm_eps=1e-5) if norm else {}
    ws = ops.gemm_workspace(M, N, K, dev)
    run = lambda i: ops.gemm(a, lins[i % copies], act="silu" if glu else "none", out=o, ws=ws, residual=res, **kw)
    for i in range(copies):
        run(i)
    torch.cuda.synchronize()
    ops.gemm_plan_counts(reset=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 4 * copies
    e0.record()
    for i in range(n):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    pc = ops.gemm_plan_counts()
    plans = [i for i, v in enumerate(pc) if v]
    split = ""
    if fn is not None and 20 in plans:
        buf = np.zeros(4096, dtype=np.uint64)
        run(1)
        run(2)
        torch.cuda.synchronize()
        fn(buf.ctypes.data)
        t = buf.reshape(1024, 4).astype(np.float64) / 100.0
        t = t[t[:, 0] > 0]
        t = t[t[:, 0] > t[:, 0].max() - 50.0]        # the workgroups of the LAST launch (stale slots of an earlier, larger grid are older)
        t0 = t[:, 0].min()
        med = lambda x: float(np.median(x))
        split = (f"workgroups {len(t):4d}: start spread {t[:, 0].max() - t0:4.1f} | prologue (start -> chunk 0 staged) {med(t[:, 1] - t[:, 0]):4.1f} | "
                 f"stream (-> last MFMA) {med(t[:, 2] - t[:, 1]):5.1f} | reduction + epilogue {med(t[:, 3] - t[:, 2]):4.1f} | kernel {t[:, 3].max() - t0:5.1f} us")
    elif 19 in plans:
        split = "second-generation streaming kernel (gemm_skinny2: one 1024-deep K chunk per workgroup, no in-kernel stamps)"
    mb = N * K * 2 / 1e6
    rows.append((name, mb, us))
    print(f"{name:8s} weights {mb:6.1f} MB  launch-to-launch {us:5.1f} us (GEMM" + (" + its splitk_reduce_norm launch" if norm else "") +
          f") = {mb / us:5.2f} TB/s  plans {plans}\n         {split}", flush=True)
    del lins
    torch.cuda.empty_cache()
tot_mb, tot_us = sum(r[1] for r in rows), sum(r[2] for r in rows)
print(f"four projections: {tot_mb:.0f} MB in {tot_us:.1f} us = {tot_mb / tot_us:.2f} TB/s (+ the fused decode attention launch, 14.7 us in the decision's kernel trace: "
      f"no weights) -> a layer-step of ~{tot_us + 14.7:.0f} us; HBM floor of the weights alone at 8 TB/s: {tot_mb / 8.0:.1f} us")
