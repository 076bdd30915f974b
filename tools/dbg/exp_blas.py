"""Calibration only (NOT a product path): what torch.matmul (hipBLASLt / rocBLAS) reaches at the prefill GEMM shapes."""
import os, torch, time
dev = torch.device("cuda:0")
M = int(os.environ.get("M", "449"))
for K, N in [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(5):
        y = a @ w.T
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        y = a @ w.T
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"N={N} K={K}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.0f} TF")
