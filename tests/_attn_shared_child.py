"""Child process of tests/test_kernels_gpu.py::test_attention_shared_keys_form_equals_the_per_tile_form: one seeded large-N decode attention problem (8 batch entries
x 64 query rows x 32 heads at D = 128 over [one shared slot of 257 keys | a per-entry slot of up to 24 keys], resumed from a state) through cover_attention_bf16
under whatever COVER_ATTN_SHARED the parent set (read once per process); the output rows are saved to the file named on the command line."""
import sys

import torch

from cover_vla_amd import ops


def problem(dev):
    P, S, H, D, T0, LT = 8, 64, 32, 128, 257, 24
    g = torch.Generator().manual_seed(5)
    bf = lambda x: x.to(torch.bfloat16)
    q = bf(torch.randn(P * S, 3 * H * D, generator=g)).to(dev)
    cap0 = 288
    k0 = bf(torch.randn(1, cap0, H, D, generator=g)).to(dev)
    v0 = bf(torch.randn(1, H, D, cap0, generator=g)).to(dev)
    k1 = bf(torch.randn(P, 32, H, D, generator=g)).to(dev)
    v1 = bf(torch.randn(P, H, D, 32, generator=g)).to(dev)
    zero = torch.zeros(P, dtype=torch.int32, device=dev)
    len1 = (9 + (torch.arange(P) * 5) % 16).to(torch.int32).to(dev)
    segs = [ops.Segment(k0, v0, (cap0 * H * D, H * D, D), (H * D * cap0, D * cap0, cap0), length=T0, slot_of_batch=zero),
            ops.Segment(k1, v1, (32 * H * D, H * D, D), (H * D * 32, D * 32, 32), length=LT, len_of_batch=len1)]
    state = ((torch.randn(P * S, H, D, generator=g) * 0.3).to(dev),
             torch.stack([torch.randn(P * S, H, generator=g), torch.rand(P * S, H, generator=g) + 0.5], -1).contiguous().to(dev))
    return P, S, H, D, q, segs, state, (k0, v0, k1, v1, len1, T0)


def main(path):
    dev = torch.device("cuda:0")
    P, S, H, D, q, segs, state, _ = problem(dev)
    out = torch.empty(P * S, H * D, dtype=torch.bfloat16, device=dev)
    ops.attention(q, (S * 3 * H * D, 3 * H * D, D), out, (S * H * D, H * D, D), P, S, H, H, D, D ** -0.5, segs, state_in=state)
    out2 = torch.empty_like(out)
    ops.attention(q, (S * 3 * H * D, 3 * H * D, D), out2, (S * H * D, H * D, D), P, S, H, H, D, D ** -0.5, segs)
    torch.save({"resumed": out.cpu(), "plain": out2.cpu()}, path)


if __name__ == "__main__":
    main(sys.argv[1])
