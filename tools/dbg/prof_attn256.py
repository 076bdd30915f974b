"""D = 256 MQA attention (pi0 denoise shape: B candidates x 5 suffix rows x 8 q heads on 1 kv head, prefix keys + 5 suffix keys):
kernel-timestamp duration against the number of prefix keys and B -- what part of the 26 us is fixed?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cover_vla_amd import ops, _lib as L
dev = torch.device("cuda:0")
h = L.lib()
Hq, Hkv, Tq = 8, 1, 5
for D, B in ((256, 40), (128, 40), (64, 40), (256, 8)):
    for Tp in (328, 32):
        q = torch.randn(B, Tq, Hq, D, device=dev).bfloat16()
        cap = 352
        k = torch.randn(8, cap, Hkv, D, device=dev).bfloat16()
        vt = torch.randn(8, Hkv, D, cap, device=dev).bfloat16()
        ks = torch.randn(B, 32, Hkv, D, device=dev).bfloat16()
        vts = torch.randn(B, Hkv, D, 32, device=dev).bfloat16()
        slot = (torch.arange(B, device=dev) // max(1, B // 8)).clamp(max=7).to(torch.int32)
        plen = torch.full((B,), Tp, dtype=torch.int32, device=dev)
        vis = torch.tensor([1, 5, 5, 5, 5], dtype=torch.int32, device=dev)
        segs = [ops.Segment(k, vt, (cap * Hkv * D, Hkv * D, D), (Hkv * D * cap, D * cap, cap), length=Tp, slot_of_batch=slot, len_of_batch=plen),
                ops.Segment(ks, vts, (32 * Hkv * D, Hkv * D, D), (Hkv * D * 32, D * 32, 32), length=Tq, mask=ops.MASK_VISLEN, vis_len=vis)]
        out = torch.empty(B, Tq, Hq, D, dtype=torch.bfloat16, device=dev)
        st = (Tq * Hq * D, Hq * D, D)
        for _ in range(3): ops.attention(q, st, out, st, B, Tq, Hq, Hkv, D, D ** -0.5, segs)
        torch.cuda.synchronize()
        n = 7
        ms, cnt, work = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
        L.check(h.cover_profile_begin(256), "b")
        for _ in range(20): ops.attention(q, st, out, st, B, Tq, Hq, Hkv, D, D ** -0.5, segs)
        L.check(h.cover_profile_end_n(ms, cnt, work, n), "e")
        print(f"D={D} B={B} prefix keys={Tp}: {ms[2] / 20 * 1e3:.1f} us", flush=True)
