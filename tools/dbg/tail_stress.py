"""Stress of the in-kernel hand-off protocols on two full-width Llama-2-7B layers (tests/test_chain_gpu.py's model): ITERS decode passes with
the tail reduction (COVER_TAIL_REDUCE=1) / the persistent chain (COVER_DECODE_CHAIN=1) against the bit pattern of the plain path /
the split-phase chain, other kernels co-running. Prints the number of passes that differ. Usage: python tools/dbg/tail_stress.py [ITERS]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_chain_gpu as T
from cover_vla_amd import ops

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
m, _ = T.build_llm(dev)
side = torch.cuda.Stream(device=dev)
a = torch.randn(2048, 2048, device=dev)
big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
for M in (32, 16):
    g = torch.Generator(device=dev).manual_seed(11 + M)
    x0 = torch.randn(M, T.L7["dim"], device=dev, generator=g).to(T.BF)
    for name, ref_args, run_args in (("tail", dict(mode="0", tail="0"), dict(mode="0", tail="1")), ("chain", dict(mode="2"), dict(mode="1"))):
        xr, kr, vr, _ = T._run(m, dev, x0, ref_args["mode"], M, tail=ref_args.get("tail", "0"))
        bad = 0
        for it in range(ITERS):
            if it % 3:
                with torch.cuda.stream(side):
                    for _ in range(4):
                        a = (a @ a).tanh()
                        big.add_(1)
            xt, kt, vt, _ = T._run(m, dev, x0, run_args["mode"], M, tail=run_args.get("tail", "0"))
            ok = torch.equal(xt, xr) and all(torch.equal(p, q) for p, q in zip(kt, kr)) and all(torch.equal(p, q) for p, q in zip(vt, vr))
            bad += 0 if ok else 1
        side.synchronize()
        print(f"M={M} {name}: {bad} of {ITERS} passes differ", flush=True)
torch.cuda.synchronize()
ops.gemm_tail_status(); ops.decode_chain_status()
