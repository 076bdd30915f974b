"""fp8 weight profile (BASELINE config 5; the reference has no fp8 path -- SURVEY.md 7 step 9 -- so the bar is: the e4m3
quantiser is exact against torch's float8_e4m3fn cast, the e4m3 weight stream is BIT-IDENTICAL to the bf16 stream of the same
quantised weights (power-of-two scales), the model on fp8 weights matches the oracle on the de-quantised weights as tightly as
the bf16 model matches its oracle, and agreement with the UNQUANTISED model is reported (token agreement rate / score RMSE)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from cover_vla_amd import ops, synth  # noqa: E402


def dequant_reference(w: torch.Tensor):
    """CPU restatement of cover_quantize_rows_fp8: per-row power-of-two scale, RNE e4m3 (torch.float8_e4m3fn), de-quantised."""
    w = w.to(torch.bfloat16).float()
    amax = w.abs().amax(dim=1)
    e = torch.ceil(torch.log2(amax.double() / 448.0))
    s = torch.where(amax > 0, torch.pow(2.0, e), torch.ones_like(e)).float()
    q = (w / s[:, None]).to(torch.float8_e4m3fn).float()
    return (q * s[:, None]).to(torch.bfloat16), s


def test_quantizer_matches_torch_float8_cast(dev):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(96, 520, generator=g) * 0.02
    w[3] *= 37.0                     # a row with another scale
    w[5] = 0                         # an all-zero row
    w[7, 0] = 448.0 * 2 ** -7        # exactly on a power-of-two boundary: max / 448 = 2^-7 -> s = 2^-7
    ref, s_ref = dequant_reference(w)
    lin = ops.pack_linear(w.to(dev), fp8=True)
    # the bf16 image of the twin is pack(Wdq): unpack by multiplying an identity
    eye = torch.zeros(520, lin.kp, dtype=torch.bfloat16, device=dev)
    eye[torch.arange(520), torch.arange(520)] = 1.0
    wdq = ops.gemm(eye, lin, variant=1).float().cpu().T            # [N, K]
    assert torch.equal(wdq, ref.float())
    assert torch.equal(lin.w8s[:96].cpu(), s_ref)
    assert lin.w8.numel() == (96 // 16) * 16 * lin.kp              # one byte per (padded) weight


@pytest.mark.parametrize("M,N,K,glu", [(32, 12288, 4096, False), (32, 22016, 4096, True), (32, 4096, 11008, False), (32, 4096, 4096, False),
                                       (7, 32064, 4096, False), (16, 1024, 2304, False), (20, 8192, 6272, False)])
def test_fp8_weight_stream_is_bit_identical_to_bf16_stream(dev, M, N, K, glu):
    """Weight-streaming kernels (second and third generation, split and unsplit plans, GLU epilogue, M < 32, ragged K) on the
    e4m3 image vs the bf16 image of the SAME quantised weights: identical bits (scales are powers of two)."""
    g = torch.Generator(device=dev).manual_seed(N + K + M)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02)
    w[: N // 3] *= 8.0
    bias = None if glu else torch.randn(N, device=dev, generator=g) * 0.1
    lin = ops.pack_linear(w, bias, glu=glu, fp8=True)
    a = torch.randn(M, lin.kp, device=dev, generator=g).bfloat16()[:, :K] if K % 128 == 0 else None
    if a is None:
        a = torch.zeros(M, lin.kp, dtype=torch.bfloat16, device=dev)
        a[:, :K] = torch.randn(M, K, device=dev, generator=g).bfloat16()
    act = "silu" if glu else "none"
    y8 = ops.gemm(a, lin, act=act, variant=3)
    lin.use_w8 = False
    y16 = ops.gemm(a, lin, act=act, variant=3)
    assert torch.equal(y8.view(torch.int16), y16.view(torch.int16))
    # and it is the GEMM of the de-quantised weight (fp32 reference)
    wdq, _ = dequant_reference(w.cpu())
    y = a[:, :K].float().cpu() @ wdq.float().T
    ref = torch.nn.functional.silu(y[:, : N // 2]) * y[:, N // 2:] if glu else y + bias.cpu().to(torch.bfloat16).float()
    assert ((y8.float().cpu() - ref).norm() / ref.norm()).item() < 6e-3
    # fp32 output + norm epilogues go through the same streaming kernels
    if not glu:
        lin.use_w8 = True
        z8 = ops.gemm(a, lin, variant=3, out_f32=True)
        lin.use_w8 = False
        z16 = ops.gemm(a, lin, variant=3, out_f32=True)
        assert torch.equal(z8, z16)


def _act_perm(kp):
    """position -> k of cover_quantize_act_fp8's row order: inside every 64-block, byte 16 g + 8 h + e holds k = 32 h + 8 g + e."""
    p = np.arange(kp)
    c, r = p // 64, p % 64
    g, h, e = r // 16, (r // 8) % 2, r % 8
    return c * 64 + 32 * h + 8 * g + e


def _dequant_act(q, sc, K):
    """e4m3 rows in operand order + row scales -> fp32 [M, K] in natural k order."""
    kp = q.shape[1]
    nat = torch.empty_like(q)
    nat[:, torch.from_numpy(_act_perm(kp))] = q
    return (nat.view(torch.float8_e4m3fn).float() * sc[:, None])[:, :K]


@pytest.mark.parametrize("M,K", [(5, 4096), (64, 11008), (17, 200), (3, 128)])
def test_activation_quantiser_matches_torch_float8_cast(dev, M, K):
    """cover_quantize_act_fp8: per-row power-of-two scale (smallest 2^e with amax / 2^e <= 448), RNE e4m3 = torch.float8_e4m3fn of
    x / s, operand k order, zero padding to a multiple of 128. Bit-exact."""
    g = torch.Generator().manual_seed(K + M)
    x = (torch.randn(M, K, generator=g) * torch.logspace(-2, 1, M)[:, None]).bfloat16()
    if M > 2:
        x[1] = 0                                           # an all-zero row: scale 1
        x[2, 0] = 448.0 * 2 ** -3                          # amax exactly on a power-of-two boundary
        x[2, 1:] = x[2, 1:].clamp(-40, 40)
    q, sc = ops.quantize_act_fp8(x.to(dev))
    kp = (K + 127) // 128 * 128
    assert q.shape == (M, kp)
    amax = x.float().abs().amax(1)
    s_ref = torch.where(amax > 0, torch.pow(2.0, torch.ceil(torch.log2(amax.double() / 448.0))).float(), torch.ones(M))
    assert torch.equal(sc.cpu(), s_ref)
    want = torch.zeros(M, kp)
    want[:, :K] = x.float() / s_ref[:, None]
    want8 = want.to(torch.float8_e4m3fn).view(torch.uint8)
    got_nat = torch.empty(M, kp, dtype=torch.uint8)
    got_nat[:, torch.from_numpy(_act_perm(kp))] = q.cpu()
    # +0 / -0 of exact zeros may differ in sign only where the input is -0: compare values, then bits away from zero
    assert torch.equal(got_nat.view(torch.float8_e4m3fn).float(), want8.view(torch.float8_e4m3fn).float())


@pytest.mark.parametrize("M,N,K,glu,norm", [(512, 12288, 4096, False, False), (512, 22016, 4096, True, False), (512, 4096, 11008, False, True),
                                            (512, 4096, 4096, False, True), (448, 12288, 4096, False, False), (448, 4096, 11008, False, True),
                                            (448, 22016, 4096, True, False), (530, 4096, 4096, False, False), (1024, 8192, 2304, False, False)])
def test_fp8_mfma_tiled_gemm_matches_fp32_on_quantised_operands(dev, M, N, K, glu, norm):
    """The MX-scaled fp8 matrix instruction path (both operands e4m3, M > 64) vs an fp32 matmul of the SAME de-quantised operands:
    exact products, fp32 accumulation in another order, bf16 output rounding -> rel-L2 <= 3e-3 (bias / residual / GLU / fused RMSNorm
    epilogues, split-K plans, ragged row tiles). Also: it must differ from the bf16-activation path (the fp8 kernel really ran)."""
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    w = torch.randn(N, K, device=dev, generator=g) * 0.02
    w[: N // 4] *= 6.0
    bias = None if glu else torch.randn(N, device=dev, generator=g) * 0.1
    lin = ops.pack_linear(w, bias, glu=glu, fp8=True)
    a = torch.zeros(M, lin.kp, dtype=torch.bfloat16, device=dev)
    a[:, :K] = (torch.randn(M, K, device=dev, generator=g) * torch.logspace(-1, 1, M, device=dev)[:, None]).bfloat16()
    q, sc = ops.quantize_act_fp8(a, K)
    res = None if (glu or not norm) else torch.randn(M, N, device=dev, generator=g).bfloat16()
    kw = {}
    if norm:
        nw = (torch.rand(N, device=dev, generator=g) + 0.5).float()
        kw = dict(norm_w=nw, norm_out=torch.empty(M, N, dtype=torch.bfloat16, device=dev), norm_style=1, norm_eps=1e-5)
    act = "silu" if glu else "none"
    y8 = ops.gemm(a, lin, act=act, residual=res, a8=(q, sc), **kw)
    n8 = kw["norm_out"].clone() if norm else None
    y16 = ops.gemm(a, lin, act=act, residual=res, **kw)
    adq = _dequant_act(q.cpu(), sc.cpu(), K)
    wdq, _ = dequant_reference(w.cpu())
    y = adq.double() @ wdq.double().T
    if glu:
        yb = y.float().bfloat16().float()
        ref = (torch.nn.functional.silu(yb[:, : N // 2]).bfloat16().float() * yb[:, N // 2:])
    else:
        ref = (y + bias.cpu().bfloat16().double()).float().bfloat16().float()
        if res is not None:
            ref = ref + res.float().cpu()
    rel = ((y8.float().cpu() - ref).norm() / ref.norm()).item()
    assert rel < 3e-3, rel
    assert not torch.equal(y8.view(torch.int16), y16.view(torch.int16)), "the fp8-activation path was not taken"
    rel16 = ((y16.float().cpu() - ref).norm() / ref.norm()).item()
    print(f"M={M} N={N} K={K}: fp8 MFMA vs fp32-on-quantised rel-L2 {rel:.2e}; bf16-activation path vs the same reference {rel16:.2e} (activation quantisation error)")
    if norm:   # fused norm of the stored bf16 rows (HF LlamaRMSNorm arithmetic), checked against the kernel's own output rows
        yo = y8.float().cpu()
        want = (nw.cpu() * (yo * torch.rsqrt(yo.pow(2).mean(-1, keepdim=True) + 1e-5)).bfloat16().float())
        assert ((n8.float().cpu() - want).norm() / want.norm()).item() < 4e-3


@pytest.mark.parametrize("M,N,K", [(448, 4096, 4096), (512, 12288, 4096), (448, 4096, 11008)])
def test_fp8_activation_error_bound_with_outliers(dev, M, N, K):
    """what per-row e4m3 ACTIVATION quantisation can do to a projection's output, bounded, on activations that look like a
    trained decoder's rather than a Gaussian: heavy-tailed rows (Student-t, 3 dof), a handful of "massive" channels 30-100 x the row's
    typical magnitude (the outlier channels of LLM residual streams), per-row dynamic range over three decades.
    Against the SAME GEMM with bf16 activations (identical e4m3 weights, so the weight quantisation cancels):
      (1) a hard element-wise bound. RNE to e4m3 with a per-row power-of-two scale s (amax / s in (224, 448]) moves an element by at
          most 2^-4 |a| in the normal range and by at most s 2^-10 below it (half a subnormal step), so for every output
          |y8 - y16| <= sum_k (2^-4 |a_k| + s 2^-10) |w_k|  (+ the bf16 rounding of both outputs);
      (2) the typical size: rel-L2 <= 4e-2 -- independent +-2^-4 relative errors have rms 2^-4 / sqrt(3) = 3.6 %; outlier channels do
          not make it worse because the error of a channel scales with the channel, and the small channels stay far above the
          subnormal floor as long as the row's dynamic range is below 448 x 2^6."""
    g = torch.Generator(device=dev).manual_seed(M + N + K + 1)
    w = torch.randn(N, K, device=dev, generator=g) * 0.02
    lin = ops.pack_linear(w, None, fp8=True)
    z = torch.randn(M, K, device=dev, generator=g)
    chi = (torch.randn(M, K, 3, device=dev, generator=g) ** 2).sum(-1) / 3.0
    a32 = z / chi.sqrt()                                                   # Student-t(3): heavy tails
    a32 = a32 * torch.logspace(-1.5, 1.5, M, device=dev)[:, None]          # per-row scale over three decades
    hot = torch.randperm(K, device=dev, generator=g)[:6]
    a32[:, hot] *= torch.tensor([30.0, 45.0, 60.0, 80.0, 100.0, 35.0], device=dev)   # massive channels, every row
    a = torch.zeros(M, lin.kp, dtype=torch.bfloat16, device=dev)
    a[:, :K] = a32.bfloat16()
    q, sc = ops.quantize_act_fp8(a, K)
    y8 = ops.gemm(a, lin, a8=(q, sc)).float()
    y16 = ops.gemm(a, lin).float()
    assert not torch.equal(y8, y16), "the fp8-activation path was not taken"
    wdq = dequant_reference(w.cpu())[0].float().to(dev)
    af = a[:, :K].float()
    bound = (af.abs() * 2.0 ** -4 + sc[:, None] * 2.0 ** -10) @ wdq.abs().T
    slack = 2.0 ** -7 * (y8.abs() + y16.abs()) + 1e-6                      # both outputs are bf16-rounded fp32 sums
    d = (y8 - y16).abs()
    assert bool((d <= bound + slack).all()), float((d - bound - slack).max())
    rel = ((y8 - y16).norm() / y16.norm()).item()
    used = float((d / (bound + 1e-12)).max())
    print(f"M={M} N={N} K={K}: e4m3-activation vs bf16-activation rel-L2 {rel:.3e}; worst element uses {used:.2f} of its bound; amax/median |a| = "
          f"{float(af.abs().amax(1).median() / af.abs().median()):.0f}")
    assert rel < 4e-2, rel


# ---------------------------------------------------------------------------------------------- MX block scales (config 5's down_proj input)
def mx_quant_reference(x: torch.Tensor, K: int):
    """CPU restatement of cover_quantize_act_fp8_mx (and of the GLU epilogue that writes the same bytes): one power-of-two scale per 32 consecutive
    k of a row -- the smallest 2^e (e >= -126) with amax_block / 2^e <= 448, 2^0 for an all-zero block -- RNE e4m3 of x / s in plain row-major order,
    zero padding to a multiple of 128; E8M0 bytes 127 + e laid out [k / 128][m][(k / 32) % 4]."""
    M = x.shape[0]
    kp = (K + 127) // 128 * 128
    xp = torch.zeros(M, kp, dtype=torch.float32)
    xp[:, :K] = x[:, :K].float()
    blk = xp.view(M, kp // 32, 32)
    amax = blk.abs().amax(-1)
    e = torch.where(amax > 0, torch.ceil(torch.log2(amax.double() / 448.0)), torch.zeros_like(amax, dtype=torch.float64)).clamp(min=-126)
    s = torch.pow(2.0, e).float()
    q = (blk / s[..., None]).to(torch.float8_e4m3fn)
    mx = (e + 127).to(torch.uint8).view(M, kp // 128, 4).permute(1, 0, 2).contiguous()
    return q.view(M, kp), mx, (q.float() * s[..., None]).view(M, kp)


@pytest.mark.parametrize("M,K", [(5, 4096), (70, 11008), (17, 200), (3, 128)])
def test_mx_activation_quantiser_matches_restatement(dev, M, K):
    g = torch.Generator().manual_seed(K + M + 7)
    x = (torch.randn(M, K, generator=g) * torch.logspace(-2, 1, M)[:, None]).bfloat16()
    x[:, 32:64] *= 50.0                                    # a block far from its neighbours' range
    if M > 2:
        x[1] = 0                                           # all-zero blocks: scale 2^0
        x[2, 0] = 448.0 * 2 ** -3                          # a block amax exactly on a power-of-two boundary
        x[2, 1:32] = x[2, 1:32].clamp(-40, 40)
    q, mx = ops.quantize_act_fp8_mx(x.to(dev))
    q_ref, mx_ref, _ = mx_quant_reference(x, K)
    assert q.shape == q_ref.shape and mx.shape == mx_ref.shape
    assert torch.equal(mx.cpu(), mx_ref)
    assert torch.equal(q.cpu().view(torch.float8_e4m3fn).float(), q_ref.float())   # (+0 / -0 of exact zeros compare equal as values)


@pytest.mark.parametrize("M,N,K,norm", [(512, 4096, 11008, True), (448, 4096, 11008, True), (200, 4096, 4096, False), (530, 1040, 2304, False),
                                        (1024, 8192, 2304, False)])
def test_fp8_mx_tiled_gemm_matches_fp32_on_quantised_operands(dev, M, N, K, norm):
    """Block-scaled activations x k-linear e4m3 weights on v_mfma_scale_f32_16x16x128_f8f6f4's OWN block scales (gemm_tiled_v3_f8<.., MX = 1>: split-K
    plans, the 256 x 128 / 224 x 128 / 128 x 256 tiles, ragged rows and columns) vs an fp64 matmul of the SAME de-quantised operands: rel-L2 <= 3e-3 --
    and tighter than the per-row-scale path on activations whose blocks differ in range (that is what the block scales are for)."""
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    w = torch.randn(N, K, device=dev, generator=g) * 0.02
    w[: N // 4] *= 6.0
    bias = torch.randn(N, device=dev, generator=g) * 0.1
    lin = ops.pack_linear(w, bias, fp8=True, klinear=True)
    lin0 = ops.pack_linear(w, bias, fp8=True)
    a = torch.zeros(M, lin.kp, dtype=torch.bfloat16, device=dev)
    a[:, :K] = (torch.randn(M, K, device=dev, generator=g) * torch.logspace(-1, 1, M, device=dev)[:, None]).bfloat16()
    a[:, 64:96] *= 40.0                                    # one block of every row with its own range
    q, mx = ops.quantize_act_fp8_mx(a, K)
    res = torch.randn(M, N, device=dev, generator=g).bfloat16() if norm else None
    kw = {}
    if norm:
        nw = (torch.rand(N, device=dev, generator=g) + 0.5).float()
        kw = dict(norm_w=nw, norm_out=torch.empty(M, N, dtype=torch.bfloat16, device=dev), norm_style=1, norm_eps=1e-5)
    ops.gemm_plan_counts(reset=True)
    y8 = ops.gemm(a, lin, residual=res, a8=(q, mx), **kw)
    assert ops.gemm_plan_counts()[21] == 1, "the fp8 tiles were not taken"
    n8 = kw["norm_out"].clone() if norm else None
    _, _, adq = mx_quant_reference(a.cpu(), K)
    wdq, _ = dequant_reference(w.cpu())
    ref = (adq[:, :K].double() @ wdq.double().T + bias.cpu().bfloat16().double()).float().bfloat16().float()
    if res is not None:
        ref = ref + res.float().cpu()
    rel = ((y8.float().cpu() - ref).norm() / ref.norm()).item()
    assert rel < 3e-3, rel
    # the un-quantised product, for the size of the quantisation error itself: block scales vs one scale per row
    exact = (a[:, :K].double().cpu() @ wdq.double().T + bias.cpu().bfloat16().double()).float()
    if res is not None:
        exact = exact + res.float().cpu()
    q0, s0 = ops.quantize_act_fp8(a, K)
    ops.gemm_plan_counts(reset=True)
    y0 = ops.gemm(a, lin0, residual=res, a8=(q0, s0), **({**kw, "norm_out": torch.empty_like(kw["norm_out"])} if norm else {}))
    row_on_fp8 = ops.gemm_plan_counts()[21] == 1             # (shapes outside the planner's fp8 range run the row-scale twin on the bf16 kernels)
    e_mx = ((y8.float().cpu() - exact).norm() / exact.norm()).item()
    e_row = ((y0.float().cpu() - exact).norm() / exact.norm()).item()
    print(f"M={M} N={N} K={K}: MX GEMM vs fp64-on-quantised rel-L2 {rel:.2e}; quantisation error vs bf16 activations: block scales {e_mx:.3e}, "
          f"row scales {e_row:.3e}{'' if row_on_fp8 else ' (bf16 kernel)'}")
    # e4m3 is a floating-point format: its relative step does not depend on the scale, so on activations whose blocks differ by a factor of 40 the block
    # scales buy NO accuracy over one scale per row (what they buy is a quantiser that needs only the producer's own tile) ...
    assert e_mx <= 1.02 * e_row or not row_on_fp8
    if row_on_fp8:   # ... until a row's range exceeds e4m3's: a block 3e4 x the rest pushes the rest of the row below the row scale's subnormals
        a2 = a.clone()
        a2[:, 64:96] *= 750.0
        q2, mx2 = ops.quantize_act_fp8_mx(a2, K)
        q3, s3 = ops.quantize_act_fp8(a2, K)
        wd = wdq.double()[:, 96:K]                                   # the part of the product the hot block does not dominate
        ex2 = a2[:, 96:K].double().cpu() @ wd.T
        _, _, adq2 = mx_quant_reference(a2.cpu(), K)
        e2_mx = ((adq2[:, 96:K].double() @ wd.T - ex2).norm() / ex2.norm()).item()
        adq3 = _dequant_act(q3.cpu(), s3.cpu(), K)
        e2_row = ((adq3[:, 96:K].double() @ wd.T - ex2).norm() / ex2.norm()).item()
        print(f"   with one block 3e4 x the rest: the rest of the product, block scales {e2_mx:.3e}, row scales {e2_row:.3e}")
        assert e2_mx < 0.5 * e2_row
    if norm:
        yo = y8.float().cpu()
        want = (nw.cpu() * (yo * torch.rsqrt(yo.pow(2).mean(-1, keepdim=True) + 1e-5)).bfloat16().float())
        assert ((n8.float().cpu() - want).norm() / want.norm()).item() < 4e-3


@pytest.mark.parametrize("M,N,K,act", [(512, 22016, 4096, "silu"), (448, 22016, 4096, "silu"), (300, 2 * 2080, 2304, "gelu_tanh")])
def test_glu_gemm_writes_the_mx_form_of_its_bf16_output(dev, M, N, K, act):
    """out8 of a GLU GEMM on the fp8 tiles = cover_quantize_act_fp8_mx of the bf16 rows the same GEMM stores without it: e4m3 bytes and E8M0 scales
    bit for bit (ragged last column tile, 128 x 192 / 224 x 128 / 128 x 256 tiles)."""
    g = torch.Generator(device=dev).manual_seed(M + N + K + 3)
    w = torch.randn(N, K, device=dev, generator=g) * 0.02
    lin = ops.pack_linear(w, None, glu=True, fp8=True)
    a = torch.zeros(M, lin.kp, dtype=torch.bfloat16, device=dev)
    a[:, :K] = (torch.randn(M, K, device=dev, generator=g) * torch.logspace(-1, 1, M, device=dev)[:, None]).bfloat16()
    q, sc = ops.quantize_act_fp8(a, K)
    ops.gemm_plan_counts(reset=True)
    y = ops.gemm(a, lin, act=act, a8=(q, sc))
    same_kernel = ops.gemm_plan_counts()[21] == 1            # (out8 keeps ANY shape on the fp8 tiles; without it the planner sends small ones to the bf16 kernels)
    n_out = N // 2
    kp_o = (n_out + 127) // 128 * 128
    o8 = torch.zeros(M, kp_o, dtype=torch.uint8, device=dev)
    omx = torch.full((kp_o // 128, M, 4), 127, dtype=torch.uint8, device=dev)
    ops.gemm(a, lin, act=act, a8=(q, sc), out8=(o8, omx))
    yp = torch.zeros(M, kp_o, dtype=torch.bfloat16, device=dev)
    yp[:, :n_out] = y
    q_ref, mx_ref = ops.quantize_act_fp8_mx(yp, n_out)
    nb = n_out // 32                                            # whole blocks the GEMM owns (n_out % 32 == 0 is required)
    if not same_kernel:   # bf16-activation reference: the bytes differ by the activation quantisation; the de-quantised rows must agree to e4m3 precision
        e = omx.permute(1, 0, 2).reshape(M, -1)[:, :nb].float().cpu() - 127.0
        deq = (o8[:, :n_out].cpu().view(torch.float8_e4m3fn).float().view(M, nb, 32) * torch.pow(2.0, e)[..., None]).view(M, n_out)
        rel = ((deq - y.float().cpu()).norm() / y.float().cpu().norm()).item()
        assert rel < 6e-2, rel
        de = (e - (mx_ref.permute(1, 0, 2).reshape(M, -1)[:, :nb].float().cpu() - 127.0)).abs().max().item()
        assert de <= 1.0, de
        return
    assert torch.equal(omx.permute(1, 0, 2).reshape(M, -1)[:, :nb].cpu(), mx_ref.permute(1, 0, 2).reshape(M, -1)[:, :nb].cpu())
    assert torch.equal(o8[:, :n_out].cpu().view(torch.float8_e4m3fn).float(), q_ref[:, :n_out].cpu().view(torch.float8_e4m3fn).float())
    # and the restatement agrees with both
    q_cpu, mx_cpu, _ = mx_quant_reference(y.cpu(), n_out)
    assert torch.equal(mx_ref.cpu(), mx_cpu) and torch.equal(q_ref.cpu().view(torch.float8_e4m3fn).float(), q_cpu.float())


@pytest.mark.parametrize("M,N,K", [(32, 4096, 11008), (32, 4096, 4096), (7, 4096, 11008), (16, 1024, 2304), (20, 8192, 6272)])
def test_klinear_weight_stream_equals_the_bf16_stream(dev, M, N, K):
    """The weight-streaming kernels on the k-linear e4m3 image (second and third generation, split and unsplit plans, ragged K): the lanes multiply
    other k than with the default image, the SET of products per output is the same -- identical to the bf16 stream of the same quantised weights up
    to the summation order (rel-L2 <= 2e-3 against each other and <= 6e-3 against fp32)."""
    g = torch.Generator(device=dev).manual_seed(N + K + M + 11)
    w = torch.randn(N, K, device=dev, generator=g) * 0.02
    w[: N // 3] *= 8.0
    bias = torch.randn(N, device=dev, generator=g) * 0.1
    lin = ops.pack_linear(w, bias, fp8=True, klinear=True)
    a = torch.zeros(M, lin.kp, dtype=torch.bfloat16, device=dev)
    a[:, :K] = torch.randn(M, K, device=dev, generator=g).bfloat16()
    y8 = ops.gemm(a, lin, variant=3)
    lin.use_w8 = False
    y16 = ops.gemm(a, lin, variant=3)
    assert ((y8.float() - y16.float()).norm() / y16.float().norm()).item() < 2e-3
    wdq, _ = dequant_reference(w.cpu())
    ref = a[:, :K].float().cpu() @ wdq.float().T + bias.cpu().to(torch.bfloat16).float()
    assert ((y8.float().cpu() - ref).norm() / ref.norm()).item() < 6e-3


@pytest.mark.parametrize("B,Tq,Tk,chained", [(4, 64, 300, False), (8, 64, 281, True), (3, 50, 77, False)])
def test_attention_writes_the_mx_form_of_its_bf16_output(dev, B, Tq, Tk, chained):
    """cover_attn_args.out8: the key-split attention kernel at D = 128 (32 heads: a head = one 128-deep k-tile of o_proj) writes e4m3 rows + E8M0 block scales
    that are bit for bit cover_quantize_act_fp8_mx of the bf16 rows the same call stores without it -- also resumed from a state (the config-5 decode pass:
    8 prompts x 64 samples = 1 024 query tiles) and with ragged query tiles / per-batch key lengths."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_kernels_gpu import bf, make_cache
    H, D = 32, 128
    g = torch.Generator().manual_seed(B + Tq + Tk)
    q = bf(torch.randn(B, Tq, H, D, generator=g))
    k = bf(torch.randn(B, Tk, H, D, generator=g))
    v = bf(torch.randn(B, Tk, H, D, generator=g) * torch.logspace(-1, 1, H)[None, None, :, None])
    lens = torch.tensor([max(1, Tk - 7 * i) for i in range(B)], dtype=torch.int32)
    kc, vt, ks, vs = make_cache(k, v, dev)
    seg = ops.Segment(kc, vt, ks, vs, length=Tk, len_of_batch=lens.to(dev))
    rows = B * Tq
    st = (Tq * H * D, H * D, D)
    state = None
    if chained:
        state = ((torch.randn(B, Tq, H, D, generator=g) * 0.3).to(dev), torch.stack([torch.randn(B, Tq, H, generator=g), torch.rand(B, Tq, H, generator=g) + 0.5], -1).to(dev))
    out = torch.empty(rows, H * D, dtype=torch.bfloat16, device=dev)
    ops.attention(q.to(dev), st, out, st, B, Tq, H, H, D, D ** -0.5, [seg], state_in=state)
    o8 = torch.zeros(rows, H * D, dtype=torch.uint8, device=dev)
    omx = torch.zeros(H * D // 128, rows, 4, dtype=torch.uint8, device=dev)
    ops.attention(q.to(dev), st, None, st, B, Tq, H, H, D, D ** -0.5, [seg], state_in=state, out8=(o8, omx))
    q_ref, mx_ref = ops.quantize_act_fp8_mx(out)
    assert torch.equal(omx.cpu(), mx_ref.cpu())
    assert torch.equal(o8.cpu().view(torch.float8_e4m3fn).float(), q_ref.cpu().view(torch.float8_e4m3fn).float())
    # a problem the key-split kernel does not take is refused, not silently written as bf16
    with pytest.raises(RuntimeError):
        ops.attention(q.to(dev), st, None, st, B, Tq, H, 8, D, D ** -0.5, [seg], out8=(o8, omx))           # GQA


def test_decoder_mx_down_input_fused_equals_unfused_and_matches_oracle(dev):
    """cover_decoder_forward with the MX block-scaled down_proj and o_proj inputs (two Llama-style layers at 2048 wide, MLP 4096, one causal pass of 448 rows -- the
    smallest geometry that takes the fp8 tiles): (1) the GLU epilogue of gate_up writes the SAME operand bytes as the standalone quantiser launch it
    replaces (COVER_FP8_MX_FUSE=0) -- hidden rows bit-identical; (2) against the oracle on the de-quantised weights with the same quantisers at the
    projections' inputs (per row at qkv / gate_up, per 32-block at o_proj / down_proj) the device path deviates from the bf16-activation oracle no more than the oracle's own
    fake-quantised evaluation does (the statistical bar of tests/fp8_mfma_model_case.py); (3) plan counters: every projection ran on the fp8 tiles."""
    from cover_ref import blocks as Bk
    from cover_vla_amd.models import Decoder, KvGeometry
    dim, Hq, D, mlp, T, layers = 2048, 16, 128, 4096, 448, 2
    g = synth._G(21, True)
    sd = synth.decoder_state(g, dim=dim, layers=layers, Hq=Hq, Hkv=Hq, D=D, mlp=mlp, rms_base=1.0)
    x0 = (torch.randn(T, dim, generator=torch.Generator().manual_seed(3)) * 0.5).bfloat16()
    pos = torch.arange(T, dtype=torch.int32, device=dev)
    outs = {}
    for fuse in ("1", "0"):
        os.environ["COVER_FP8_MX_FUSE"] = fuse
        try:
            llm = Decoder(sd, dim=dim, layers=layers, Hq=Hq, Hkv=Hq, D=D, mlp=mlp, act="silu", norm="llama", eps=1e-5, rope="hf", n_pos=T + 8,
                          device="cuda:0", cache=KvGeometry(Hq, D, [1], [T]), fp8_weights=True)
            assert llm._arr[0].down_klinear == 1
            x = x0.clone().to(dev)
            ops.gemm_plan_counts(reset=True)
            llm.forward(x, [llm.group(1, T, pos, [dict(region=0, length=T, mask=ops.MASK_CAUSAL)], 0)], final_norm=True)
            counts = ops.gemm_plan_counts()
            assert counts[21] == 4 * layers and sum(counts) == 4 * layers, counts
            outs[fuse] = x.cpu()
        finally:
            os.environ.pop("COVER_FP8_MX_FUSE", None)
    assert torch.equal(outs["1"].view(torch.int16), outs["0"].view(torch.int16))
    cfg = Bk.DecoderCfg(dim, layers, Hq, Hq, D, mlp, "silu", "llama", 1e-5, "hf")
    osd = Bk.to_bf16({k: (dequant_reference(v)[0].float() if k.endswith("_proj.weight") else v) for k, v in sd.items()})
    mask = torch.tril(torch.ones(T, T, dtype=torch.bool))[None]
    with torch.no_grad():
        ref = Bk.decoder_forward(cfg, osd, x0[None].clone(), pos.cpu()[None].long(), mask, n_pos=T + 8)[0][0].float()
        rq = Bk.decoder_forward(cfg, osd, x0[None].clone(), pos.cpu()[None].long(), mask, n_pos=T + 8, act_fp8=True, act_mx_down=True, act_mx_o=True)[0][0].float()
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    e_h, e_q, e_hq = rel(outs["1"].float(), ref), rel(rq, ref), rel(outs["1"].float(), rq)
    print(f"MX down input, 2 layers x 448 rows: device vs bf16-activation oracle {e_h:.4f}, fake-quant oracle vs the same {e_q:.4f}, device vs fake-quant oracle {e_hq:.4f}")
    assert e_q > 0.005 and e_h > 0.005
    assert e_h <= 1.25 * e_q + 0.005 and e_hq <= 1.6 * e_q, (e_h, e_q, e_hq)


def _dequant_sd(sd):
    out = dict(sd)
    for k, v in sd.items():
        if (k.startswith("llm.layers.") and k.endswith("_proj.weight")) or k == "lm_head.weight":
            out[k] = dequant_reference(v)[0].float()
    return out


@pytest.mark.parametrize("greedy", [True, False])
def test_openvla_fp8_matches_oracle_on_dequantised_weights(dev, greedy):
    """Same criteria as tests/test_openvla_gpu.py (logit tolerances, exact selection rule, data-decided picks exact), the oracle
    holding the DE-QUANTISED weights; plus the reported agreement with the unquantised oracle."""
    from cover_ref import blocks as Bk, openvla as OR
    from cover_vla_amd.openvla import OpenVLA
    from tests.test_openvla_gpu import _case
    c, sd, frame, toks, lens, u = _case(seed=5)
    n_samples = 1 if greedy else 2
    P = toks.shape[0]
    un = None if greedy else u[: P * n_samples]
    sdq = _dequant_sd(sd)
    model = OpenVLA(sd, c, device="cuda:0", max_prompts=4, max_candidates=8, max_text=toks.shape[1], weight_dtype="fp8")
    assert model.llm.fp8_weights and model.lm_head.w8 is not None
    otr, utr = {}, {}
    with torch.no_grad():
        ref = OR.sample(c, Bk.to_bf16(sdq), frame, toks, lens, n_samples, un, 0.9, trace=otr)
        ref_unq = OR.sample(c, Bk.to_bf16(sd), frame, toks, lens, n_samples, un, 0.9, trace=utr)
    tr = {}
    tokens, _ = model.sample(frame.to(dev), toks.to(dev), lens.to(dev), n_samples, None if greedy else un.to(dev), 0.9, trace=tr,
                             force_tokens=ref.to(dev))
    tokens = tokens.cpu()
    rl = otr["logits"]
    gl = torch.stack([l.cpu() for l in tr["logits"]], 1)
    lo, hi = (0, c["tok_vocab"]) if greedy else (c["tok_vocab"] - c["n_bins"], c["tok_vocab"])
    n_dec = 0
    for n in range(tokens.shape[0]):
        for i in range(7):
            err = (gl[n, i] - rl[n, i]).abs().max().item()
            scale = rl[n, i].abs().max().item()
            rel = ((gl[n, i] - rl[n, i]).norm() / rl[n, i].norm()).item()
            assert err < max(5e-2, 4e-2 * scale) and rel < 3.5e-2, (n, i, err, scale, rel)
            assert int(tokens[n, i]) == OR.select_token(gl[n, i], lo, hi, None if greedy else float(un[n, i]), 0.9)
            if greedy:
                top2 = torch.topk(rl[n, i, lo:hi], 2).values
                if (top2[0] - top2[1]).item() > 2 * err:
                    n_dec += 1
                    assert tokens[n, i] == ref[n, i]
    # reported (not a parity bar): what quantisation itself changes -- de-quantised oracle vs unquantised oracle
    agree_q = (ref == ref_unq).float().mean().item()
    rmse = (otr["logits"] - utr["logits"]).pow(2).mean().sqrt().item()
    print(f"fp8 vs oracle(dequantised): token agreement {(tokens == ref).float().mean().item():.3f}, data-decided {n_dec}; "
          f"quantisation effect (oracle fp8 vs oracle bf16): token agreement {agree_q:.3f}, logit RMSE {rmse:.4f}")


def test_openvla_fp8_mfma_decode_rows_match_oracle(dev):
    """Model-level run of the fp8 MFMA path (more than 64 decode rows on e4m3 weights) in a child process with the tile knob that
    gives the small config an fp8-capable tile; see tests/fp8_mfma_model_case.py."""
    import subprocess
    env = dict(os.environ, COVER_TILE_PICK="a", COVER_FP8_MX="0")   # (per-row activation scales on every projection: what the oracle's act_fp8 flags restate)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fp8_mfma_model_case.py")], env=env, capture_output=True, text=True, timeout=900)
    print(p.stdout[-1500:])
    assert p.returncode == 0 and "FP8_MFMA_MODEL_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
