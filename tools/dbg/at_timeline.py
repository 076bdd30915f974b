"""Per-block timeline of the key-split attention kernel (library built with -DCOVER_AT_DEBUG, loaded through COVER_LIB_PATH):
pi0 denoise shape, D = 256 MQA. Stamps of thread 0 (wave 0) of each block."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cover_vla_amd import ops, _lib as L
dev = torch.device("cuda:0")
fn = L.lib().cover_at_debug
fn.argtypes = [C.c_void_p]
Hq, Hkv, Tq = 8, 1, 5
for D, B, Tp in ((256, 40, 328), (256, 40, 32), (64, 40, 32)):
    q = torch.randn(B, Tq, Hq, D, device=dev).bfloat16()
    cap = 352
    k = torch.randn(8, cap, Hkv, D, device=dev).bfloat16(); vt = torch.randn(8, Hkv, D, cap, device=dev).bfloat16()
    ks = torch.randn(B, 32, Hkv, D, device=dev).bfloat16(); vts = torch.randn(B, Hkv, D, 32, device=dev).bfloat16()
    slot = (torch.arange(B, device=dev) // 5).clamp(max=7).to(torch.int32)
    plen = torch.full((B,), Tp, dtype=torch.int32, device=dev)
    vis = torch.tensor([1, 5, 5, 5, 5], dtype=torch.int32, device=dev)
    segs = [ops.Segment(k, vt, (cap * Hkv * D, Hkv * D, D), (Hkv * D * cap, D * cap, cap), length=Tp, slot_of_batch=slot, len_of_batch=plen),
            ops.Segment(ks, vts, (32 * Hkv * D, Hkv * D, D), (Hkv * D * 32, D * 32, 32), length=Tq, mask=ops.MASK_VISLEN, vis_len=vis)]
    out = torch.empty(B, Tq, Hq, D, dtype=torch.bfloat16, device=dev)
    st = (Tq * Hq * D, Hq * D, D)
    for rep in range(3):
        ops.attention(q, st, out, st, B, Tq, Hq, Hkv, D, D ** -0.5, segs)
        torch.cuda.synchronize()
    buf = np.zeros(4096, dtype=np.uint64); fn(buf.ctypes.data)
    t = buf.reshape(512, 8).astype(np.float64) / 100.0
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    qq = lambda x: f"{np.percentile(x - t0, 5):5.1f}/{np.median(x - t0):5.1f}/{(x - t0).max():5.1f}"
    print(f"D={D} B={B} keys={Tp}: blocks {len(t)}  start {qq(t[:,0])}  Q {qq(t[:,1])}  firstK {qq(t[:,2])}  tiles {qq(t[:,3])}  merged {qq(t[:,4])}  end {qq(t[:,5])}  (p5/median/max us)", flush=True)
