"""Per-block timeline of the key-split attention kernel (library built with -DCOVER_AT_DEBUG, loaded through COVER_LIB_PATH):
pi0 denoise shape, D = 256 MQA. Stamps of thread 0 (wave 0) of each block."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cover_vla_amd import ops, _lib as L
dev = torch.device("cuda:0")
try:
    fn = L.lib().cover_at_debug          # (-DCOVER_AT_DEBUG builds only)
    fn.argtypes = [C.c_void_p]
except AttributeError:
    fn = None
def report(tag):
    if fn is None:
        print(tag, flush=True)
        return
    buf = np.zeros(4096, dtype=np.uint64); fn(buf.ctypes.data)
    t = buf.reshape(512, 8).astype(np.float64) / 100.0
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    qq = lambda x: f"{np.percentile(x - t0, 5):5.1f}/{np.median(x - t0):5.1f}/{(x - t0).max():5.1f}"
    raw = buf.reshape(512, 8)
    raw = raw[raw[:, 0] > 0]
    if (raw[:, 7] != 0).any():   # attn_shared_k: ticks of thread 0 summed over the tiles, per phase of a tile
        f = lambda col, sh: np.median((raw[:, col] >> np.uint64(sh)) & np.uint64(0xffff)) / 100.0
        print(f"   tile phases (us summed over the tiles, median over workgroups): wait {f(6, 0):.2f} | barrier {f(6, 16):.2f} | DMA issue {f(6, 32):.2f} | S^T {f(7, 0):.2f} | softmax {f(7, 16):.2f} | PV {f(7, 32):.2f}")
    print(f"{tag}: blocks {len(t)}  start {qq(t[:,0])}  Q {qq(t[:,1])}  firstK {qq(t[:,2])}  tiles {qq(t[:,3])}  merged {qq(t[:,4])}  end {qq(t[:,5])}  (p5/median/max us; attn_shared_k: "
          f"'Q' = first DMA batch issued, 'firstK' = Q + state landed, 'merged' unused)", flush=True)


if os.environ.get("SHAPE") == "c5":
    # the attention pass of a config-5 decode layer: 8 prompts x 64 samples, 32 heads (MHA) at D = 128 over [shared prefix 257 keys | prompt text <= 24 keys],
    # resumed from the own-token state; bf16 output and the block-scaled output (the first 512 of the 1024 workgroups are stamped)
    P, S, H, D, T0, LT = 8, 64, 32, 128, 257, 24
    N = P * S
    q = torch.randn(N, 3 * H * D, device=dev).bfloat16()
    cap0 = 288
    k0 = torch.randn(1, cap0, H, D, device=dev).bfloat16(); v0 = torch.randn(1, H, D, cap0, device=dev).bfloat16()
    k1 = torch.randn(P, 32, H, D, device=dev).bfloat16(); v1 = torch.randn(P, H, D, 32, device=dev).bfloat16()
    zero = torch.zeros(P, dtype=torch.int32, device=dev)
    len1 = (9 + (torch.arange(P, device=dev) * 5) % 16).to(torch.int32)
    segs = [ops.Segment(k0, v0, (cap0 * H * D, H * D, D), (H * D * cap0, D * cap0, cap0), length=T0, slot_of_batch=zero),
            ops.Segment(k1, v1, (32 * H * D, H * D, D), (H * D * 32, D * 32, 32), length=LT, len_of_batch=len1)]
    state = (torch.randn(N, H, D, device=dev) * 0.3, torch.stack([torch.randn(N, H, device=dev), torch.rand(N, H, device=dev) + 0.5], -1).contiguous())
    out = torch.empty(N, H * D, dtype=torch.bfloat16, device=dev)
    o8 = torch.empty(N, H * D, dtype=torch.uint8, device=dev); omx = torch.empty(H * D // 128, N, 4, dtype=torch.uint8, device=dev)
    for tag, kw in (("bf16 out", dict()), ("MX out  ", dict(out8=(o8, omx)))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(3):
            e0.record()
            ops.attention(q, (S * 3 * H * D, 3 * H * D, D), None if kw else out, (S * H * D, H * D, D), P, S, H, H, D, D ** -0.5, segs, state_in=state, **kw)
            e1.record()
            torch.cuda.synchronize()
        report(f"config-5 decode attention, {tag} ({e0.elapsed_time(e1) * 1e3:.1f} us launch to launch)")
    sys.exit(0)

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
