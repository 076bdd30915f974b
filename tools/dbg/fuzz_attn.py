"""Randomised sweep of the generic attention entry point (cover_attention_bf16) against an fp32 restatement: B, Tq, GQA ratio, head dims
64 / 96 / 128 / 256, one to three key segments with their own caches (shared slot, slot map, per-row lengths, causal with offset),
ragged lengths. Usage: python tools/dbg/fuzz_attn.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cover_vla_amd import ops


def make_cache(k, v, dev, tcap=None):
    S, T, Hkv, D = k.shape
    tcap = tcap or (T + 31) // 32 * 32
    vt = torch.zeros(S, Hkv, D, tcap, dtype=torch.bfloat16, device=dev)
    vt[..., :T] = v.permute(0, 2, 3, 1)
    return k.contiguous(), vt, (T * Hkv * D, Hkv * D, D), (Hkv * D * tcap, D * tcap, tcap)


def attn_ref(q, segs, scale):
    Hq = q.shape[2]
    k = torch.cat([s[0] for s in segs], 1).float()
    v = torch.cat([s[1] for s in segs], 1).float()
    vis = torch.cat([s[2] for s in segs], 2)
    G = Hq // k.shape[2]
    k, v = k.repeat_interleave(G, 2), v.repeat_interleave(G, 2)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k) * scale
    s = s.masked_fill(~vis[:, None], float("-inf"))
    p = torch.nan_to_num(torch.softmax(s, -1), nan=0.0)
    return torch.einsum("bhqk,bkhd->bqhd", p, v)


def run(cases, seed, dev=None, verbose=True):
    dev = dev or torch.device("cuda:0")
    rnd = random.Random(seed)
    fails = []
    for c in range(cases):
        D = rnd.choice([64, 96, 128, 256])
        Hkv = rnd.choice([1, 2, 4])
        Hq = Hkv * rnd.choice([1, 2, 4, 8])
        B = rnd.randint(1, 9)
        Tq = rnd.choice([1, 1, rnd.randint(2, 8), rnd.randint(9, 70), rnd.randint(71, 300)])
        nseg = rnd.randint(1, 3)
        g = torch.Generator(device=dev).manual_seed(seed * 100003 + c)
        rn = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
        q = rn(B, Tq, Hq, D)
        ref_segs, segs, keep = [], [], []
        for si in range(nseg):
            causal = si == nseg - 1 and rnd.random() < 0.5
            if causal:
                off = rnd.randint(0, 5)
                T = Tq + off
                k, v = rn(B, T, Hkv, D), rn(B, T, Hkv, D)
                vis = (torch.arange(T, device=dev)[None, None, :] <= (torch.arange(Tq, device=dev)[None, :, None] + off)).expand(B, Tq, T)
                cch = make_cache(k, v, dev)
                segs.append(ops.Segment(cch[0], cch[1], cch[2], cch[3], length=T, mask=ops.MASK_CAUSAL, causal_offset=off))
                ref_segs.append((k, v, vis))
            else:
                T = rnd.choice([rnd.randint(1, 40), rnd.randint(41, 300), rnd.randint(301, 900)])
                shared = rnd.random() < 0.4
                S = 1 if shared else rnd.randint(1, B)
                k, v = rn(S, T, Hkv, D), rn(S, T, Hkv, D)
                slot = torch.zeros(B, dtype=torch.int32, device=dev) if shared else torch.randint(0, S, (B,), device=dev, generator=g).to(torch.int32)
                lens = torch.randint(1, T + 1, (B,), device=dev, generator=g).to(torch.int32) if rnd.random() < 0.6 else None
                vis = torch.ones(B, Tq, T, dtype=torch.bool, device=dev) if lens is None else (torch.arange(T, device=dev)[None, None, :] < lens[:, None, None]).expand(B, Tq, T)
                cch = make_cache(k, v, dev)
                segs.append(ops.Segment(cch[0], cch[1], cch[2], cch[3], length=T, slot_of_batch=slot, len_of_batch=lens))
                ref_segs.append((k[slot.long()], v[slot.long()], vis))
            keep.append(cch)
        scale = D ** -0.5
        out = torch.empty(B, Tq, Hq, D, dtype=torch.bfloat16, device=dev)
        tag = f"case {c}: D={D} Hq={Hq} Hkv={Hkv} B={B} Tq={Tq} segs={[ (s[0].shape[1]) for s in ref_segs]}"
        try:
            ops.attention(q, (Tq * Hq * D, Hq * D, D), out, (Tq * Hq * D, Hq * D, D), B, Tq, Hq, Hkv, D, scale, segs)
            ref = attn_ref(q, ref_segs, scale)
            err = ((out.float() - ref).norm() / (ref.norm() + 1e-12)).item()
            ok = err < 1.5e-2 and bool(torch.isfinite(out.float()).all())
        except Exception as ex:   # noqa: BLE001
            err, ok = float("nan"), False
            tag += f" EXC {ex}"
        if not ok:
            fails.append(f"FAIL {tag}: rel {err:.2e}")
            if verbose:
                print(fails[-1], flush=True)
    torch.cuda.synchronize()
    return fails


if __name__ == "__main__":
    torch.cuda.set_device(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    fails = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n} cases, {len(fails)} failures")
