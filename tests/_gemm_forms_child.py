"""Child process of tests/test_kernels_gpu.py::test_gemm_fp8_loader_wave_and_self_loading_forms_agree: a fixed list of seeded GEMMs (fp8
config-5 shapes) through cover_gemm_bf16 under whatever COVER_V3_F8 the parent set (read once per process),
outputs saved to the .pt file named on the command line together with the plan counters."""
import sys

import torch

from cover_vla_amd import ops

CASES = [  # (M, N, K, glu, fp8)
    (512, 12288, 4096, False, True), (512, 22016, 4096, True, True), (512, 4096, 11008, False, True), (530, 6144, 2304, False, True),
]


def main(path):
    dev = torch.device("cuda:0")
    out = {}
    ops.gemm_plan_counts(reset=True)
    for i, (M, N, K, glu, f8) in enumerate(CASES):
        g = torch.Generator(device=dev).manual_seed(1000 + i)
        w = torch.randn(N, K, device=dev, generator=g) * 0.02
        lin = ops.pack_linear(w, None if glu else torch.randn(N, device=dev, generator=g) * 0.1, glu=glu, fp8=f8)
        a = torch.zeros(M, lin.kp, dtype=torch.bfloat16, device=dev)
        a[:, :K] = torch.randn(M, K, device=dev, generator=g).bfloat16()
        a8 = ops.quantize_act_fp8(a, K) if f8 else None
        ws = ops.gemm_workspace(M, N, K, dev)
        y = ops.gemm(a, lin, act="silu" if glu else "none", a8=a8, ws=ws)
        out[i] = y.cpu()
    out["plans"] = ops.gemm_plan_counts()
    torch.save(out, path)


if __name__ == "__main__":
    main(sys.argv[1])
