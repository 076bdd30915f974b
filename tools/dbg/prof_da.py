"""Fused decode attention at config-5 sizes: N candidates, H = 32, D = 128, shared 257 + prompt 24 + own L2 keys. Kernel timestamps."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cover_vla_amd import ops, _lib as L
dev = torch.device("cuda:0")
h = L.lib()
H, D = 32, 128
def cache(S, T, cap):
    k = torch.randn(S, cap, H, D, device=dev).bfloat16()
    vt = torch.randn(S, H, D, cap, device=dev).bfloat16()
    return k, vt, (cap * H * D, H * D, D), (H * D * cap, D * cap, cap)
for N, samples in [(32, 4), (512, 64)]:
    ncol = 3 * H * D
    qkv = torch.randn(N, ncol, device=dev).bfloat16()
    pos = torch.randint(0, 300, (N,), device=dev, dtype=torch.int32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.arange(400).float()[:, None] * inv[None]
    cos, sin = ang.cos().to(dev), ang.sin().to(dev)
    c0, c1, c2 = cache(1, 257, 288), cache(8, 24, 32), cache(N, 64, 64)
    zero = torch.zeros(N, dtype=torch.int32, device=dev)
    slot1 = (torch.arange(N, device=dev) // samples).to(torch.int32)
    len1 = torch.full((N,), 20, dtype=torch.int32, device=dev)
    out = torch.empty(N, H * D, dtype=torch.bfloat16, device=dev)
    for L2 in (1, 7, 8, 16, 32, 56):
        s = [ops.Segment(c0[0], c0[1], c0[2], c0[3], length=257, slot_of_batch=zero),
             ops.Segment(c1[0], c1[1], c1[2], c1[3], length=24, slot_of_batch=slot1, len_of_batch=len1),
             ops.Segment(c2[0], c2[1], c2[2], c2[3], length=L2)]
        for _ in range(3): ops.decode_attention_fused(qkv, N, H, D, D ** -0.5, s, L2 - 1, out, positions=pos, cos=cos, sin=sin, rope_mode=2)
        torch.cuda.synchronize()
        n = 7
        ms, cnt, work = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
        L.check(h.cover_profile_begin(256), "b")
        for _ in range(20): ops.decode_attention_fused(qkv, N, H, D, D ** -0.5, s, L2 - 1, out, positions=pos, cos=cos, sin=sin, rope_mode=2)
        L.check(h.cover_profile_end_n(ms, cnt, work, n), "e")
        kv = (257 * (N // 16 + 0) * 0 + 0)
        print(f"N={N} own keys={L2}: {ms[2] / 20 * 1e3:.1f} us  (own-segment KV bytes {N * L2 * H * D * 4 / 1e6:.0f} MB)", flush=True)
