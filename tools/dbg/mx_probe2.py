"""Which block scale does v_mfma_scale_f32_16x16x128_f8f6f4 apply to which operand bytes? One 16-byte chunk of every activation row is 1.0 (e4m3 0x38), the
weights are all ones, the four scale bytes of a row are 2^0, 2^1, 2^2, 2^3 for the lane groups g = 0..3 as the kernel hands them over: y / 16 = the scale
the hardware applied to the chunk's products. Chunk c of the 128-byte row goes to lane group c / 2, operand bytes 16 (c % 2) .. in the kernel's k-linear read."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cover_vla_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = 128, 256, 128
lin = ops.pack_linear(torch.ones(N, K, device=dev), None, fp8=True, klinear=True)
a = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
mx = torch.tensor([127, 128, 129, 130], dtype=torch.uint8, device=dev)[None, None].expand(1, M, 4).contiguous()
for c in range(8):
    q = torch.zeros(M, K, dtype=torch.uint8, device=dev)
    q[:, 16 * c: 16 * c + 16] = 0x38
    y = ops.gemm(a, lin, a8=(q, mx), out_f32=True)
    vals = sorted(set((y / 16).flatten().tolist()))
    print(f"chunk {c} (lane group {c // 2}, operand half {c % 2}): applied scale(s) {vals[:6]}")
# and byte-granular: one byte at a time inside lane group 1's operand
for b in (32, 39, 40, 47, 48, 55, 56, 63):
    q = torch.zeros(M, K, dtype=torch.uint8, device=dev)
    q[:, b] = 0x38
    y = ops.gemm(a, lin, a8=(q, mx), out_f32=True)
    print(f"byte {b}: applied scale(s) {sorted(set(y.flatten().tolist()))[:6]}")
