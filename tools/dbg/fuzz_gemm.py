"""Randomised sweep of the bf16 GEMM entry point (cover_gemm_bf16, automatic plan selection) against an fp32 matmul of the same bf16
operands: ragged M / N, every planner path (weight streaming, 64..256-row tiles, split-K), bias / activation / residual / GLU / fused
RMSNorm epilogues. Prints every case whose rel-L2 exceeds the bar. Usage: python tools/dbg/fuzz_gemm.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cover_vla_amd import ops

ACT = {"none": lambda x: x, "gelu_tanh": lambda x: torch.nn.functional.gelu(x, approximate="tanh"), "silu": torch.nn.functional.silu}


def run(cases, seed, dev=None, verbose=True):
    """Returns (failures, plans hit): failures = list of strings, one per case over the bar."""
    dev = dev or torch.device("cuda:0")
    rnd = random.Random(seed)
    fails, plans = [], {}
    for c in range(cases):
        _case(c, rnd, dev, fails, plans, verbose)
    torch.cuda.synchronize()
    return fails, plans


def _case(c, rnd, dev, fails, plans, verbose):
    bad = 0
    if True:
        kind = rnd.choice(["plain", "bias_act", "residual", "glu", "norm"])
        M = rnd.choice([rnd.randint(1, 64), rnd.randint(65, 300), rnd.randint(301, 1200), rnd.choice([1, 16, 17, 32, 33, 64, 65, 224, 225, 448, 449, 512])])
        K = 128 * rnd.choice([1, 2, 3, 5, 8, 9, 16, 17, 32, 33, 43, 86])
        N = 8 * rnd.randint(1, 1400) if rnd.random() < 0.7 else rnd.choice([16, 32, 64, 1024, 4096, 4304, 12288, 11008])
        if kind == "glu":
            N = max(32, N // 32 * 32)
        if kind == "norm":
            N = min(N, 8192)
        if M * N * K > 6e10:
            K = 128 * 8
        g = torch.Generator(device=dev).manual_seed(c + 7919 * rnd.randint(0, 1 << 20))
        a = torch.randn(M, K, device=dev, generator=g).bfloat16()
        w = (torch.randn(N, K, device=dev, generator=g) * (K ** -0.5)).bfloat16()
        y = a.float() @ w.float().T
        ops.gemm_plan_counts(reset=True)
        try:
            if kind == "plain":
                out = ops.gemm(a, ops.pack_linear(w)); ref = y
            elif kind == "bias_act":
                b = torch.randn(N, device=dev, generator=g)
                act = rnd.choice(list(ACT))
                lin = ops.pack_linear(w, b)
                out = ops.gemm(a, lin, act=act); ref = ACT[act](y + lin.bias)
            elif kind == "residual":
                r = torch.randn(M, N, device=dev, generator=g).bfloat16()
                x = r.clone()
                out = ops.gemm(a, ops.pack_linear(w), residual=x, out=x); ref = r.float() + y
            elif kind == "glu":
                out = ops.gemm(a, ops.pack_linear(w, glu=True), act="silu")
                h = N // 2
                ref = torch.nn.functional.silu(y[:, :h].bfloat16().float()).bfloat16().float() * y[:, h:].bfloat16().float()
            else:
                r = torch.randn(M, N, device=dev, generator=g).bfloat16()
                nw = torch.rand(N, device=dev, generator=g) + 0.5
                x, hn = r.clone(), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                ops.gemm(a, ops.pack_linear(w), residual=x, out=x, norm_w=nw, norm_out=hn, norm_style=1, norm_eps=1e-5)
                xr = (r.float() + y).bfloat16().float()
                ref = nw * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5)).bfloat16().float()
                out = hn
                e0 = ((x.float() - xr).norm() / xr.norm()).item()
                if not e0 < 8e-3:
                    fails.append(f"FAIL(x) case {c} {kind} M={M} N={N} K={K}: rel {e0:.2e}")
            err = ((out.float() - ref).norm() / (ref.norm() + 1e-12)).item()
            finite = bool(torch.isfinite(out.float()).all())
        except Exception as ex:   # noqa: BLE001
            err, finite = float("nan"), False
            fails.append(f"EXC case {c} {kind} M={M} N={N} K={K}: {ex}")
        pc = ops.gemm_plan_counts()
        key = tuple(i for i, v in enumerate(pc) if v)
        plans[key] = plans.get(key, 0) + 1
        if not (err < 1e-2 and finite):
            fails.append(f"FAIL case {c} {kind} M={M} N={N} K={K}: rel {err:.2e} finite {finite} plans {key}")
            if verbose:
                print(fails[-1], flush=True)


if __name__ == "__main__":
    torch.cuda.set_device(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    fails, plans = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n} cases, {len(fails)} failures; plans hit: {sorted(plans.items())}")
