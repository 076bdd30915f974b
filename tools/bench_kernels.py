"""Micro-benchmarks of the two GEMM kernels at the OpenVLA-7B shapes (run on the GPU box):
   tiled  -> TFLOP/s vs the 2.5 PFLOP/s bf16 MFMA peak; skinny (decode, M=32) -> GB/s of weight streaming vs 8 TB/s."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cover_vla_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = ops.Timer()
    t.start()
    for _ in range(iters):
        fn()
    return t.stop() / iters


def main():
    dev = torch.device("cuda:0")
    out = []
    shapes = [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096), (4096, 32064)]
    for M in (441,):
        for K, N in shapes[:4]:
            glu = N == 22016
            w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
            lin = ops.pack_linear(w, glu=glu)
            a = torch.randn(M, K, device=dev).bfloat16()
            o = torch.empty(M, lin.n_out, dtype=torch.bfloat16, device=dev)
            for variant in (1,):
                ms = timeit(lambda: ops.gemm(a, lin, act="silu" if glu else "none", out=o, variant=variant))
                out.append({"kernel": "tiled", "variant": variant, "M": M, "N": N, "K": K, "ms": ms,
                            "TFLOPs": 2.0 * M * N * K / ms / 1e9})
                print(out[-1], flush=True)
            del w, lin
    for M in (8, 32):
        for K, N in shapes:
            glu = N == 22016
            w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
            lin = ops.pack_linear(w, glu=glu)
            a = torch.randn(M, K, device=dev).bfloat16()
            o = torch.empty(M, lin.n_out, dtype=torch.bfloat16, device=dev)
            ws = ops.gemm_workspace(M, N, K, dev)
            for variant in (3, 5):
                ms = timeit(lambda: ops.gemm(a, lin, act="silu" if glu else "none", out=o, variant=variant, ws=ws))
                out.append({"kernel": "skinny+reduce", "variant": variant, "M": M, "N": N, "K": K, "ms": ms, "GBps": 2.0 * N * K / ms / 1e6})
                print(out[-1], flush=True)
            del w, lin
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/bench_kernels.json", "w"), indent=1)


if __name__ == "__main__":
    main()
