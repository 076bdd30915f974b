"""Persistent decode chain (cover_vla_amd/csrc/decode_chain.hip): one launch runs o_proj -> gate_up -> down -> qkv(next layer) of a
candidate-decode layer with in-kernel grid barriers and cross-phase weight prefetch. Checked here, at the Llama-2-7B layer shapes it
is built for (4096 wide, 32 x 128 MHA, MLP 11008), through cover_decoder_forward:
  * fused launch == the same phases as separate launches of the same kernel (COVER_DECODE_CHAIN=2), BIT FOR BIT, over repeated passes
    with other kernels co-running: the in-kernel hand-offs (write-through stores, barrier, acquire) add nothing and lose nothing;
  * against the separate-kernel path (COVER_DECODE_CHAIN=0: split-K GEMMs + reduce/norm launches) within the bf16 tolerance of two
    fp32 summation orders, and against an fp32 restatement of the layer on the same bf16 weights;
  * row independence (M = 1 / 8 / 32 give the same rows) and the barrier status word.
The HF pin of the same pass is tests/test_openvla_gpu.py::test_full_width_llama7b_layer_matches_hf_g3."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from cover_vla_amd import ops, synth  # noqa: E402
from cover_vla_amd.models import BF, Decoder, KvGeometry  # noqa: E402

L7 = dict(dim=4096, Hq=32, Hkv=32, D=128, mlp=11008)
T0, LT, P, S = 257, 24, 8, 4


class _Env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def build_llm(dev):
    """Two full-width layers with caches holding a random prefix / prompt text (what a prefill would have left)."""
    g = synth._G(91, True, 0.02)
    sd = synth.decoder_state(g, dim=L7["dim"], layers=2, Hq=L7["Hq"], Hkv=L7["Hkv"], D=L7["D"], mlp=L7["mlp"], rms_base=1.0)
    N = P * S
    geom = KvGeometry(L7["Hkv"], L7["D"], [1, P, N], [T0, LT, 7])
    m = Decoder(sd, dim=L7["dim"], layers=2, Hq=L7["Hq"], Hkv=L7["Hkv"], D=L7["D"], mlp=L7["mlp"], act="silu", norm="llama", eps=1e-5, rope="hf",
                n_pos=T0 + LT + 16, device="cuda:0", cache=geom)
    gg = torch.Generator(device=dev).manual_seed(5)
    for l in range(2):   # finite, bf16-sized cache contents everywhere
        m.k_cache[l].copy_((torch.randn(geom.elems, device=dev, generator=gg) * 0.5).to(BF))
        m.vt_cache[l].copy_((torch.randn(geom.elems, device=dev, generator=gg) * 0.5).to(BF))
    return m, sd


@pytest.fixture(scope="module")
def llm(dev):
    return build_llm(dev)


def _group(m, dev, N, write_t=0):
    zero = torch.zeros(N, dtype=torch.int32, device=dev)
    prompt_of = (torch.arange(N, device=dev) // S).to(torch.int32)
    lens = (16 + prompt_of % 8).to(torch.int32).contiguous()
    pos = (T0 + lens + write_t).to(torch.int32).contiguous()
    return m.group(N, 1, pos, [dict(region=0, length=T0, slot_of_batch=zero), dict(region=1, length=LT, len_of_batch=lens, slot_of_batch=prompt_of),
                               dict(region=2, length=write_t + 1)], 2, write_t_off=write_t, seg0_shared=True)


def _run(m, dev, x0, mode, N, write_t=0, tail="0", head="0"):
    """One decode pass of the two layers from hidden rows x0; mode = COVER_DECODE_CHAIN value, tail = COVER_TAIL_REDUCE value, head =
    COVER_HEAD_REDUCE value. Returns (x, own K region, own V^T region, plan counters)."""
    with _Env(COVER_DECODE_CHAIN=mode, COVER_TAIL_REDUCE=tail, COVER_HEAD_REDUCE=head):
        x = x0.clone()
        g = _group(m, dev, N, write_t)
        ops.gemm_plan_counts(reset=True)
        m.forward(x, [g], final_norm=True)
        counts = ops.gemm_plan_counts()
    torch.cuda.synchronize()
    o2 = m.geom.k_off[2]
    return x, [kc[o2:].clone() for kc in m.k_cache], [vc[o2:].clone() for vc in m.vt_cache], counts


def test_chain_fused_equals_split_phases_and_matches_separate_kernels(llm, dev):
    m, sd = llm
    N = P * S
    g = torch.Generator(device=dev).manual_seed(3)
    x0 = torch.randn(N, L7["dim"], device=dev, generator=g).to(BF)
    xs, ks, vs, c2 = _run(m, dev, x0, "2", N)
    xf, kf, vf, c1 = _run(m, dev, x0, "1", N)
    ops.decode_chain_status()
    assert sum(c1) == 0 and sum(c2) == 0, (c1, c2)                     # no GEMM launcher ran: the chain did
    assert torch.equal(xf, xs) and all(torch.equal(a, b) for a, b in zip(kf, ks)) and all(torch.equal(a, b) for a, b in zip(vf, vs))
    xl, kl, vl, c0 = _run(m, dev, x0, "0", N)
    assert c0[19] + c0[20] == 8, c0                                   # two layers x four weight-streaming launches
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    r = rel(xf, xl)
    print(f"chain vs separate kernels: hidden rel-L2 {r:.2e}, own K {rel(kf[1], kl[1]):.2e}")
    assert r < 6e-3 and rel(kf[0], kl[0]) < 2e-3 and rel(kf[1], kl[1]) < 6e-3 and rel(vf[1], vl[1]) < 6e-3
    assert torch.isfinite(xf.float()).all() and xf.float().abs().max() > 0


def test_chain_is_deterministic_under_co_running_load(llm, dev):
    """Repeated passes while another stream keeps the chip busy with unrelated kernels (uneven arrival at the barriers, other tenants in
    L2): every pass bit-identical to the split-phase reference."""
    m, _ = llm
    N = P * S
    g = torch.Generator(device=dev).manual_seed(4)
    x0 = torch.randn(N, L7["dim"], device=dev, generator=g).to(BF)
    xs, ks, vs, _ = _run(m, dev, x0, "2", N)
    side = torch.cuda.Stream(device=dev)
    a = torch.randn(2048, 2048, device=dev)
    big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for it in range(12):
        with torch.cuda.stream(side):
            for _ in range(6):
                a = (a @ a).tanh()               # MFMA tenants
                big.add_(1)                      # a streaming tenant dirtying L2 lines
        xf, kf, vf, _ = _run(m, dev, x0, "1", N)
        assert torch.equal(xf, xs), it
        assert all(torch.equal(p, q) for p, q in zip(kf, ks)) and all(torch.equal(p, q) for p, q in zip(vf, vs)), it
    side.synchronize()
    ops.decode_chain_status()


@pytest.mark.parametrize("M", [1, 8, 20])
def test_chain_rows_do_not_depend_on_the_batch(llm, dev, M):
    """The first M candidates of the 32-row pass, run as an M-row pass, give the same hidden rows bit for bit (fixed geometry: a row's
    sums never depend on M) -- the property behind the sampler's M = 1 vs M = 8 greedy test."""
    m, _ = llm
    g = torch.Generator(device=dev).manual_seed(6)
    x0 = torch.randn(P * S, L7["dim"], device=dev, generator=g).to(BF)
    x32, *_ = _run(m, dev, x0, "1", P * S)
    xm, *_ = _run(m, dev, x0[:M].contiguous(), "1", M)
    assert torch.equal(xm, x32[:M])
    ops.decode_chain_status()


def test_chain_layer_matches_fp32_restatement(llm, dev):
    """One layer + the next layer's qkv through the chain vs an fp32 torch restatement with the bf16 rounding points of the eager graph
    (RMSNorm -> bf16, projections -> bf16, attention over [prefix | text | own] from the caches, residual adds in bf16)."""
    m, sd = llm
    N = P * S
    g = torch.Generator(device=dev).manual_seed(8)
    x0 = torch.randn(N, L7["dim"], device=dev, generator=g).to(BF)
    geom, H, D = m.geom, L7["Hq"], L7["D"]
    k_before = [kc.clone() for kc in m.k_cache]
    v_before = [vc.clone() for vc in m.vt_cache]
    xf, *_ = _run(m, dev, x0, "1", N)
    # ---- fp32 restatement on the device
    bf = lambda t: t.to(BF).float()
    W = {k: v.to(dev).to(BF).float() for k, v in sd.items()}
    prompt_of = torch.arange(N, device=dev) // S
    lens = 16 + prompt_of % 8
    pos = T0 + lens
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, device=dev).float() / D))
    ang = pos.float()[:, None] * inv[None]
    cos, sin = bf(torch.cat([ang.cos(), ang.cos()], -1)), bf(torch.cat([ang.sin(), ang.sin()], -1))

    def rms(x, w):
        xf_ = x.float()
        return bf(w * bf(xf_ * torch.rsqrt(xf_.pow(2).mean(-1, keepdim=True) + 1e-5)))

    def rope(t):   # [N, H, D], HF rotate_half in bf16 arithmetic
        rot = torch.cat([-t[..., D // 2:], t[..., : D // 2]], -1)
        return bf(bf(t * cos[:, None]) + bf(rot * sin[:, None]))

    x = x0.float()
    c0, c1 = geom.caps[0], geom.caps[1]
    for l in range(2):
        pfx = f"layers.{l}."
        h = rms(x, W[pfx + "input_layernorm.weight"])
        q = bf(h @ W[pfx + "self_attn.q_proj.weight"].T).view(N, H, D)
        k = bf(h @ W[pfx + "self_attn.k_proj.weight"].T).view(N, H, D)
        v = bf(h @ W[pfx + "self_attn.v_proj.weight"].T).view(N, H, D)
        q, k = rope(q), rope(k)
        kc, vc = k_before[l].float(), v_before[l].float()
        K0 = kc[: c0 * H * D].view(c0, H, D)[:T0]
        V0 = vc[: H * D * c0].view(H, D, c0)[:, :, :T0].permute(2, 0, 1)
        o1 = geom.k_off[1]
        K1 = kc[o1: o1 + P * c1 * H * D].view(P, c1, H, D)
        V1 = vc[o1: o1 + P * H * D * c1].view(P, H, D, c1).permute(0, 3, 1, 2)
        out = torch.empty(N, H, D, device=dev)
        for n in range(N):
            p_, ln = int(prompt_of[n]), int(lens[n])
            kk = torch.cat([K0, K1[p_, :ln], k[n][None]], 0)
            vv = torch.cat([V0, V1[p_, :ln], v[n][None]], 0)
            sc = torch.einsum("hd,thd->ht", q[n], kk) * D ** -0.5
            out[n] = torch.einsum("ht,thd->hd", bf(torch.softmax(sc, -1)), vv)
        a = bf(out.reshape(N, H * D))
        x = bf(bf(a @ W[pfx + "self_attn.o_proj.weight"].T) + x)
        h2 = rms(x, W[pfx + "post_attention_layernorm.weight"])
        act = bf(bf(torch.nn.functional.silu(bf(h2 @ W[pfx + "mlp.gate_proj.weight"].T))) * bf(h2 @ W[pfx + "mlp.up_proj.weight"].T))
        x = bf(bf(act @ W[pfx + "mlp.down_proj.weight"].T) + x)
    ref = rms(x, W["norm.weight"])
    rel = ((xf.float() - ref).norm() / ref.norm()).item()
    print(f"chain (2 layers) vs fp32 restatement: rel-L2 {rel:.2e}")
    assert rel < 1.2e-2
    ops.decode_chain_status()


@pytest.mark.parametrize("M", [32, 20, 8, 1])
def test_tail_reduction_equals_the_reduction_launches(llm, dev, M):
    """COVER_TAIL_REDUCE=1: the split-K slabs of o_proj / down are folded (+ residual + RMSNorm) by the last workgroups of the launch that
    wrote them (gemm_bf16.hip "Tail reduction") instead of by splitk_reduce_norm launches. Same code on the same slabs: hidden rows and
    the K / V^T rows of the second layer bit-identical, over repeated passes with other kernels co-running (arrival order at the ticket
    counter changes from pass to pass), and the bounded wait never gives up."""
    m, _ = llm
    g = torch.Generator(device=dev).manual_seed(11)
    x0 = torch.randn(M, L7["dim"], device=dev, generator=g).to(BF)
    xr, kr, vr, c0 = _run(m, dev, x0, "0", M, tail="0")
    side = torch.cuda.Stream(device=dev)
    a = torch.randn(2048, 2048, device=dev)
    big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for it in range(8):
        if it >= 2:
            with torch.cuda.stream(side):
                for _ in range(6):
                    a = (a @ a).tanh()
                    big.add_(1)
        xt, kt, vt, c1 = _run(m, dev, x0, "0", M, tail="1")
        assert c1 == c0, (c0, c1)                                       # the same GEMM plans ran
        assert torch.equal(xt, xr), it
        assert all(torch.equal(p, q) for p, q in zip(kt, kr)) and all(torch.equal(p, q) for p, q in zip(vt, vr)), it
    side.synchronize()
    torch.cuda.synchronize()
    ops.gemm_tail_status()


@pytest.mark.parametrize("M", [32, 20, 8, 1])
def test_head_reduction_equals_the_reduction_launches(llm, dev, M):
    """COVER_HEAD_REDUCE=1 (gemm_bf16.hip "Head reduction", opt-in): the split-K slabs of o_proj / down are folded (+ residual + RMSNorm) by
    the FIRST M workgroups of the next weight-streaming launch (gate_up / the next layer's qkv), every workgroup of that launch waiting at the
    hand-off's release word before it loads activation rows. Default = the reduction launches. Same code on the
    same slabs: hidden rows and the K / V^T rows bit-identical, over repeated passes with other kernels co-running (which workgroups reach
    the counter first changes from pass to pass), fewer launches, and the bounded wait never gives up."""
    m, _ = llm
    g = torch.Generator(device=dev).manual_seed(13)
    x0 = torch.randn(M, L7["dim"], device=dev, generator=g).to(BF)
    xr, kr, vr, c0 = _run(m, dev, x0, "0", M, head="0")
    side = torch.cuda.Stream(device=dev)
    a = torch.randn(2048, 2048, device=dev)
    big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for it in range(8):
        if it >= 2:
            with torch.cuda.stream(side):
                for _ in range(6):
                    a = (a @ a).tanh()
                    big.add_(1)
        xt, kt, vt, c1 = _run(m, dev, x0, "0", M, head="1", write_t=0)
        assert c1 == c0, (c0, c1)                                       # the same GEMM plans ran
        assert torch.equal(xt, xr), it
        assert all(torch.equal(p, q) for p, q in zip(kt, kr)) and all(torch.equal(p, q) for p, q in zip(vt, vr)), it
    side.synchronize()
    torch.cuda.synchronize()
    ops.gemm_tail_status()
