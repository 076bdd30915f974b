"""Probe of the MX block-scaled fp8 GEMM (gemm_tiled_v3_f8<.., MX = 1>): which operand / scale assumption breaks. Activations are built so that the
block scales are (a) all equal, (b) one per row, (c) one per k-tile, (d) one per k-block inside a k-tile; rel-L2 against fp64 on the de-quantised operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from cover_vla_amd import ops  # noqa: E402
from test_fp8_gpu import dequant_reference, mx_quant_reference  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("M", 256)), int(os.environ.get("N", 512)), int(os.environ.get("K", 512))
g = torch.Generator(device=dev).manual_seed(1)
w = torch.randn(N, K, device=dev, generator=g) * 0.02
lin = ops.pack_linear(w, None, fp8=True, klinear=True)
wdq, _ = dequant_reference(w.cpu())
base = torch.randint(1, 8, (M, K), device=dev, generator=g).float() * (torch.randint(0, 2, (M, K), device=dev, generator=g).float() * 2 - 1)
base[:, ::32] = 7.0   # every block's amax = 7 before scaling


def run(name, scale):
    a = torch.zeros(M, lin.kp, dtype=torch.bfloat16, device=dev)
    a[:, :K] = (base * scale).bfloat16()
    q, mx = ops.quantize_act_fp8_mx(a, K)
    ops.gemm_plan_counts(reset=True)
    y = ops.gemm(a, lin, a8=(q, mx), out_f32=True)
    c = ops.gemm_plan_counts()
    _, _, adq = mx_quant_reference(a.cpu(), K)
    assert torch.equal(adq[:, :K], a[:, :K].float().cpu()), "the probe's activations must be exactly representable"
    ref = (adq[:, :K].double() @ wdq.double().T).float()
    rel = ((y.cpu() - ref).norm() / ref.norm()).item()
    ratio = (y.cpu().norm() / ref.norm()).item()
    print(f"{name:28s} rel-L2 {rel:.3e}  |y|/|ref| {ratio:.3f}  fp8 tiles {c[21]}  mx bytes {sorted(set(mx.flatten().tolist()))[:8]}")


ones = torch.ones(M, K, device=dev)
run("(a) all scales equal", ones)
run("(a') all equal, 2^-3", ones * 0.125)
rows = torch.pow(2.0, (torch.arange(M, device=dev) % 5).float())[:, None].expand(M, K)
run("(b) one scale per row", rows)
kt = torch.pow(2.0, ((torch.arange(K, device=dev) // 128) % 3).float())[None].expand(M, K)
run("(c) one scale per k-tile", kt)
kb = torch.pow(2.0, ((torch.arange(K, device=dev) // 32) % 4).float())[None].expand(M, K)
run("(d) one scale per k-block", kb)
