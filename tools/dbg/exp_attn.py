"""Decode-attention experiment: phase A (shared image prefix, candidates as query rows, state out) + phase B (per-candidate
segments, state in) at the OpenVLA-7B decode shape, 32 layers' worth of distinct caches captured into one hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cover_vla_amd import ops
dev = torch.device("cuda:0")
N, H, D, T0, T1, T2, L = 32, 32, 128, 257, 24, 7, 32
g = torch.Generator(device=dev).manual_seed(0)
def cache(S, T):
    tcap = (T + 31) // 32 * 32
    k = torch.randn(S, T, H, D, device=dev, generator=g).bfloat16()
    vt = torch.randn(S, H, D, tcap, device=dev, generator=g).bfloat16()
    return k, vt, (T * H * D, H * D, D), (H * D * tcap, D * tcap, tcap)
layers = [(cache(1, T0), cache(8, T1), cache(N, T2)) for _ in range(L)]
q = torch.randn(N, 1, H, D, device=dev, generator=g).bfloat16()
zero = torch.zeros(N, dtype=torch.int32, device=dev)
slot1 = (torch.arange(N, device=dev) // 4).to(torch.int32)
len1 = (16 + slot1 % 8).to(torch.int32)
so = torch.empty(N, H, D, dtype=torch.float32, device=dev); sml = torch.empty(N, H, 2, dtype=torch.float32, device=dev)
out = torch.empty(N, 1, H, D, dtype=torch.bfloat16, device=dev)
st = (H * D, H * D, D)
def segs(c):
    c0, c1, c2 = c
    return (ops.Segment(c0[0], c0[1], c0[2], c0[3], length=T0, slot_of_batch=zero),
            ops.Segment(c1[0], c1[1], c1[2], c1[3], length=T1, slot_of_batch=slot1, len_of_batch=len1),
            ops.Segment(c2[0], c2[1], c2[2], c2[3], length=4))
NS = 4
part = torch.randn(NS, N, 3 * H * D, device=dev, generator=g) * 0.5
posn = torch.full((N,), 300, dtype=torch.int32, device=dev)
ang = torch.arange(512, device=dev).float()[:, None] * (1.0 / (10000.0 ** (torch.arange(0, D, 2, device=dev).float() / D)))[None]
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
qkv = torch.randn(N, 3 * H * D, device=dev, generator=g).bfloat16()
out2 = torch.empty(N, H * D, dtype=torch.bfloat16, device=dev)
def run(mode):
    for c in layers:
        s0, s1, s2 = segs(c)
        if mode == "fused":
            ops.decode_attention_fused(qkv, N, H, D, D ** -0.5, [s0, s1, s2], 3, out2, positions=posn, cos=cos, sin=sin, rope_mode=2,
                                       partial=part)
        elif mode == "A":
            ops.attention(q, (0, H * D, D), None, st, 1, N, H, H, D, D ** -0.5, [s0], state_out=(so, sml))
        elif mode == "B":
            ops.attention(q, st, out, st, N, 1, H, H, D, D ** -0.5, [s1, s2], state_in=(so, sml))
        elif mode == "AB":
            ops.attention(q, (0, H * D, D), None, st, 1, N, H, H, D, D ** -0.5, [s0], state_out=(so, sml))
            ops.attention(q, st, out, st, N, 1, H, H, D, D ** -0.5, [s1, s2], state_in=(so, sml))
        else:
            ops.attention(q, st, out, st, N, 1, H, H, D, D ** -0.5, [s0, s1, s2])
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for mode in (sys.argv[1:] or ("A", "B", "AB", "one", "fused")):
        run(mode); torch.cuda.synchronize()
        with ops.Graph() as gr:
            run(mode)
        gr.launch(); torch.cuda.synchronize()
        t = ops.Timer(); t.start()
        for _ in range(10): gr.launch()
        ms = t.stop() / 10
        print(f"{mode:4s} {ms*1e3/L:7.2f} us per layer", {k: v for k, v in os.environ.items() if k.startswith('COVER_ATTN') or k.startswith('COVER_DA')}, flush=True)
