"""Full-size (BASELINE.json shapes) checks on the GPU through size-independent properties: the CPU oracle cannot run a
7B-parameter decision in seconds, so at OpenVLA-7B N=32 the HIP path is checked for determinism, row independence,
permutation equivariance, agreement between the two independent GEMM kernels, and softmax normalisation."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cover_vla_amd import ops, synth  # noqa: E402


@pytest.fixture(scope="module")
def pipe(dev):
    import bench
    return bench.Pipeline(dev, small=False)


def test_fullsize_gemm_kernels_agree(dev):
    """Weight-streaming kernel vs LDS-tiled kernel on the Llama-2-7B decode shapes (two independent code paths)."""
    g = torch.Generator(device=dev).manual_seed(0)
    for N, K, glu in [(12288, 4096, False), (22016, 4096, True), (4096, 11008, False)]:
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        lin = ops.pack_linear(w, glu=glu)
        a = torch.randn(32, K, device=dev, generator=g).bfloat16()
        y3 = ops.gemm(a, lin, act="silu" if glu else "none", variant=3).float()
        y1 = ops.gemm(a, lin, act="silu" if glu else "none", variant=1).float()
        rel = ((y3 - y1).norm() / y1.norm()).item()
        assert rel < 4e-3, (N, K, rel)      # both round the same fp32 sums to bf16; only summation order differs
        # linearity: f(2a) == 2 f(a) exactly for the linear (non-GLU) layers (power-of-two scaling commutes with rounding)
        if not glu:
            y2 = ops.gemm((a.float() * 2).bfloat16(), lin, variant=3).float()
            assert torch.equal(y2, 2 * y3)


@pytest.mark.parametrize("M,N,K", [(32, 12288, 4224), (7, 8192, 6272), (32, 32064, 4096), (20, 4096, 11008)])
def test_fullsize_streaming_plans_with_ragged_k(dev, M, N, K):
    """Third-generation weight streaming with K slices / chunks that do not divide evenly (ragged last chunk and slice),
    M < 32 and the fused-epilogue (unsplit) plan of the lm_head shape: vs the LDS-tiled kernel and an fp32 matmul."""
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev, generator=g) * 0.1
    lin = ops.pack_linear(w, bias)
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    y3 = ops.gemm(a, lin, variant=3).float()
    y1 = ops.gemm(a, lin, variant=1).float()
    ref = a.float() @ w.float().T + bias
    assert ((y3 - ref).norm() / ref.norm()).item() < 4e-3
    assert ((y3 - y1).norm() / y1.norm()).item() < 4e-3
    y32 = ops.gemm(a, lin, variant=3, out_f32=True)
    assert torch.allclose(y32, ref.bfloat16().float(), atol=0.03, rtol=2e-2)


def test_fullsize_attention_rows_are_convex_combinations(dev):
    """V = all-ones -> every output element is exactly 1 (softmax rows sum to 1) at the decode shape, 3 segments."""
    N, H, D = 32, 32, 128
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(N, 1, H, D, device=dev, generator=g).bfloat16()
    segs = []
    for S, T in [(1, 257), (8, 24), (N, 7)]:
        k = torch.randn(S, T, H, D, device=dev, generator=g).bfloat16()
        tcap = (T + 31) // 32 * 32
        vt = torch.zeros(S, H, D, tcap, dtype=torch.bfloat16, device=dev)
        vt[..., :T] = 1.0
        slot = (torch.arange(N, device=dev) * S // N).to(torch.int32) if S > 1 else torch.zeros(N, dtype=torch.int32, device=dev)
        segs.append(ops.Segment(k, vt, (T * H * D, H * D, D), (H * D * tcap, D * tcap, tcap), length=T, slot_of_batch=slot))
    out = torch.empty(N, 1, H, D, dtype=torch.bfloat16, device=dev)
    st = (H * D, H * D, D)
    ops.attention(q, st, out, st, N, 1, H, H, D, D ** -0.5, segs)
    assert torch.allclose(out.float(), torch.ones_like(out.float()), atol=8e-3)   # bf16 P rounding only


def _decision_properties(pipe, dev, P, S):
    i = pipe.inp
    # (a) determinism: bit-identical tokens and scores on a repeat
    idx1, tok1, _ = pipe.decision()
    idx2, tok2, _ = pipe.decision()
    assert idx1 == idx2 and torch.equal(tok1, tok2)
    assert 0 <= idx1 < P * S
    assert tok1.shape == (P * S, 7)
    lo, hi = pipe.c["tok_vocab"] - pipe.c["n_bins"], pipe.c["tok_vocab"]
    assert int(tok1.min()) >= lo and int(tok1.max()) < hi
    # (b) row independence: samples of one prompt given IDENTICAL uniforms must produce identical token rows
    u = i["u"].view(P, S, 7)[:, :1].expand(P, S, 7).reshape(P * S, 7).contiguous()
    tok, _ = pipe.policy.sample(i["frame"], i["toks"], i["lens"], S, u, 1.0)
    tv = tok.view(P, S, 7)
    assert torch.equal(tv, tv[:, :1].expand_as(tv))
    # (c) prompt-permutation equivariance: the slot a prompt occupies in the batch must not matter
    if P == 8:
        perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device=dev)
        tokp, _ = pipe.policy.sample(i["frame"], i["toks"][perm], i["lens"][perm], S, u.view(P, S, 7)[perm].reshape(P * S, 7).contiguous(), 1.0)
        assert torch.equal(tokp.view(P, S, 7), tv[perm])
    # (d) greedy decoding of every prompt ALONE vs its row of the batched greedy run (M = 1 vs M = P decode rows; the prefill passes have
    # ~280 vs 448 rows and run on other GEMM tiles, so the two runs' logits differ by bf16-level noise). Data-decided criterion: a step whose
    # top-1 / top-2 margin exceeds twice the logit difference must pick the same token, bit for bit. On the checkpoint with decision margins
    # (bench.Pipeline(peaked=True), the default) at least 6 of 7 steps must be decided on average over the prompts; the flat i.i.d. checkpoint
    # of rounds 1-4 (logit noise 0.14 against margins of 0.03 .. 1.7) decided 3 of 7.
    t8 = {}
    g8, _ = pipe.policy.sample(i["frame"], i["toks"], i["lens"], 1, trace=t8)
    V = pipe.c["tok_vocab"]
    decided, worst_err = 0, 0.0
    for p_ in range(P):
        t1 = {}
        # the M = 1 run is teacher-forced on the batched run's tokens (it still returns its own picks), so that every step compares like with like
        g1, _ = pipe.policy.sample(i["frame"], i["toks"][p_:p_ + 1], i["lens"][p_:p_ + 1], 1, trace=t1, force_tokens=g8[p_:p_ + 1].contiguous())
        for s_ in range(7):
            a8, a1 = t8["logits"][s_][p_, :V].float(), t1["logits"][s_][0, :V].float()
            err = float((a8 - a1).abs().max())
            top = a8.topk(2).values
            worst_err = max(worst_err, err / max(float(a8.std()), 1e-6))
            if float(top[0] - top[1]) > 2 * err:
                assert int(g1[0, s_]) == int(g8[p_, s_]), (p_, s_, err, float(top[0] - top[1]))
                decided += 1
    print(f"greedy M = 1 vs M = {P}: {decided} of {7 * P} steps data-decided (all equal), worst logit difference {worst_err:.3f} of the logit spread")
    assert worst_err < 0.5, worst_err
    assert decided >= (6 if getattr(pipe, "peaked", False) else 3) * P, decided
    # (e) the serialised decision bench.py profiles (one stream, no hipGraph replay, both towers on one stream) selects the same
    idx3, tok3, _ = pipe.decision(serial=True)
    assert idx3 == idx1 and torch.equal(tok3, tok1)


def test_fullsize_decision_properties(pipe, dev):
    """Headline configuration: OpenVLA-7B, N = 32 = 8 prompts x 4 samples, one camera, 3-member ensemble."""
    _decision_properties(pipe, dev, 8, 4)


def test_fullsize_config2_n16(pipe, dev):
    """BASELINE config 2: OpenVLA-7B N = 16 = 8 prompts x 2 samples, bf16, one MI355X (same weights, M = 16 decode rows)."""
    keep = (pipe.n_samples, pipe.inp["u"])
    pipe.n_samples = 2
    pipe.inp["u"] = keep[1].view(8, 4, 7)[:, :2].reshape(16, 7).contiguous()
    try:
        _decision_properties(pipe, dev, 8, 2)
        # candidate (p, s) of the N = 16 run draws the uniforms of candidate (p, s) of the N = 32 run: same tokens
        _, tok16, _ = pipe.decision()
        pipe.n_samples, pipe.inp["u"] = keep
        _, tok32, _ = pipe.decision()
        assert torch.equal(tok16.view(8, 2, 7), tok32.view(8, 4, 7)[:, :2])
    finally:
        pipe.n_samples, pipe.inp["u"] = keep


def test_fullsize_config3_one_prompt_group_of_32_samples(pipe, dev):
    """BASELINE config 3, the per-rank workload at FULL size (W prompt groups x 32 samples, rank r owns group r): ONE
    prompt group, 32 samples -- one prefill row group of 24 text rows behind the shared prefix (M = 280 prefill rows), M = 32 decode
    rows that all attend ONE prompt slot (both 16-candidate tiles of the fused decode attention on the same slot). Properties as the
    headline's, plus a cross-configuration sanity check against the headline run's prompt-0 candidates (same frame, prompt, uniforms)."""
    import bench
    p3 = bench.Pipeline(dev, small=False, n_prompts=1, n_samples=32)
    assert len(p3.prompt_ids) == 1 and p3.inp["u"].shape == (32, 7) and p3.policy.max_prompts == 1
    _decision_properties(p3, dev, 1, 32)
    _, tok3, _ = p3.decision()
    _, tok32, _ = pipe.decision()
    assert torch.equal(p3.inp["toks"][0], pipe.inp["toks"][0]) and torch.equal(p3.inp["u"][:4], pipe.inp["u"][:4])
    # Samples 0..3 here and the headline's prompt-0 candidates see the same frame, prompt and uniforms, but NOT the same arithmetic: this
    # prefill pass has 280 rows (other GEMM tiles, other fp32 summation order than the headline's 448), so the prompt's K / V differ in
    # the last bf16 bit and a uniform that sits on a CDF edge may pick the neighbouring bin, after which the histories diverge. Measured
    # on MI355X: 23 of 28 tokens equal. The bar only says the two runs sample the same distribution from the same state.
    same = float((tok3[:4] == tok32[:4]).float().mean())
    print(f"config 3 vs headline, prompt 0 samples 0..3: {same:.2f} of the tokens equal")
    assert same >= 0.25          # (measured 0.5-0.8 box to box; 1 / 256 would be chance)
    assert len({tuple(r) for r in tok3.tolist()}) > 8          # the 32 samples are not copies of each other
    del p3
    torch.cuda.empty_cache()


def test_fullsize_pi0_profile_b40(dev):
    """Profile P1 at FULL size -- the reference's shipped run (modeling_pi0.py:672-715 on the batch of
    run_simpler_eval_with_openpi.py:296-319): PI0_FULL (SigLIP-So400m, 18-layer Gemma-2B prefix, 300 M expert), B = 40 = 8 prompts
    x 5 samples, chunk 4, 10 Euler steps, 3-member verifier. The CPU oracle cannot run this in seconds; size-independent properties:
    determinism; rows of a prompt with IDENTICAL noise produce identical actions; the de-duplicated schedule (tower once, prefix
    once per distinct prompt) == the reference's schedule (every row its own tower + prefix pass: _image_classes forced to B classes)
    to bf16 tolerance; the three shortcuts (trailing pad columns dropped, suffix embedding folded, QKV slabs folded by the RoPE launch)
    on vs off: fold / trim change only the order of fp32 sums or the GEMM tile (bf16 tolerance), the slab fold is bit-identical."""
    import os
    import bench
    torch.cuda.empty_cache()
    pp = bench.Pi0Pipeline(dev, max_prompts=40)
    i, B, P, S = pp.inp, 40, 8, 5
    idx1, x1 = pp.decision()
    idx2, x2 = pp.decision()
    assert idx1 == idx2 and torch.equal(x1, x2) and 0 <= idx1 < B
    assert x1.shape == (B, 4, 32) and torch.isfinite(x1).all() and torch.isfinite(pp.last_scores).all()
    rel = lambda a, b: ((a - b).norm() / (b - i["noise"]).norm()).item()          # relative to the size of the update, as the golden tests
    # the serialised (profiled) decision runs the denoise loop eagerly, the default replays it as a hipGraph: bit-identical. Cut into
    # independent row-group chains (parallel branches of the graph; measured slower, so not the default) the rows agree up to the order
    # of the fp32 sums (the GEMMs see 100 / 50 instead of 200 rows)
    _, xs = pp.decision(serial=True)
    assert torch.equal(xs, x1)
    keep_ch = pp.model.n_chains
    for n_ch in (2, 4):
        pp.model.n_chains = n_ch
        for _ in range(3):
            _, xc = pp.decision()
        assert rel(xc, x1) < 5e-3, (n_ch, rel(xc, x1))
    pp.model.n_chains = keep_ch
    # rows of one prompt with identical noise are identical; different noise gives different rows
    nz = i["noise"].view(P, S, 4, 32)[:, :1].expand(P, S, 4, 32).reshape(B, 4, 32).contiguous()
    _, xe = pp.decision(noise=nz)
    xv = xe.view(P, S, 4, 32)
    assert torch.equal(xv, xv[:, :1].expand_as(xv))
    assert not torch.equal(x1.view(P, S, 4, 32)[:, 0], x1.view(P, S, 4, 32)[:, 1])
    # dedup == no dedup
    keep = pp.model._image_classes
    try:
        pp.model._image_classes = lambda cams, B_: np.arange(B_, dtype=np.int64)
        _, xn = pp.decision()
    finally:
        pp.model._image_classes = keep
    r_dedup = rel(xn, x1)
    # shortcuts on vs off
    r = {}
    for var in ("COVER_PI0_TRIM_PAD", "COVER_PI0_SUFFIX_FOLD", "COVER_QKV_FOLD"):
        os.environ[var] = "0"
        try:
            _, xo = pp.decision()
        finally:
            os.environ.pop(var, None)
        r[var] = rel(xo, x1)
        if var == "COVER_QKV_FOLD":
            assert torch.equal(xo, x1)
    print(f"P1 full size: dedup vs per-row rel {r_dedup:.2e}; shortcuts off vs on {r}")
    assert r_dedup < 2e-2 and all(v < 2e-2 for v in r.values())
    del pp
    torch.cuda.empty_cache()


def test_fullsize_config4_two_cameras_n64_ensemble2(dev):
    """BASELINE config 4: Prismatic dual encoder, 2-camera 224^2 observation (512 patch rows in the shared prefix), N = 64 =
    8 prompts x 8 samples, verifier ensemble = 2. Size-independent properties at full size (small-size oracle parity with
    two cameras: tests/test_openvla_gpu.py; 2-member verifier vs the reference golden: tests/test_models_gpu.py)."""
    import bench
    torch.cuda.empty_cache()
    p4 = bench.Pipeline(dev, small=False, n_samples=8, n_cams=2, members=2)
    assert p4.policy.T0 == 1 + 512 and len(p4.ver.trainable_models) == 2
    _decision_properties(p4, dev, 8, 8)
    # the second camera matters: another frame in camera 1 changes the sampled tokens of at least one candidate
    i = p4.inp
    _, t0, _ = p4.decision()
    f2 = i["frame"].clone()
    f2[1] = 255 - f2[1]
    t1, _ = p4.policy.sample(f2, i["toks"], i["lens"], 8, i["u"], 1.0)
    assert not torch.equal(t0, t1)
    del p4
    torch.cuda.empty_cache()


def test_fullsize_verifier_permutation(pipe, dev):
    """Scores are per-candidate: permuting the candidate histories permutes the scores; arg-max rule re-derived on host."""
    i = pipe.inp
    pf, tf = pipe.ver.extract_shared_features(i["img384"], i["text"])
    its = pipe.ver.image_text_embeddings(pf, tf)
    g = torch.Generator().manual_seed(5)
    hists = []
    for n in range(32):
        h = torch.randn(4 + n % 7, 7, generator=g) * 0.02
        h[:, 6] = (torch.rand(h.shape[0], generator=g) > 0.5).float()
        hists.append(h.double().numpy())
    r = pipe.ver.score_histories(its, hists, 4)
    s = r["scores"].cpu()
    perm = torch.randperm(32, generator=g)
    rp = pipe.ver.score_histories(its, [hists[j] for j in perm.tolist()], 4)
    assert torch.allclose(rp["scores"].cpu(), s[perm], atol=1e-6)
    gm = s.view(8, 4).mean(1)
    bg = int(gm.argmax())
    bi = int(s.view(8, 4)[bg].argmax())
    assert r["result"].cpu().tolist()[:3] == [bg * 4 + bi, bg, bi]
    assert abs(float((pf[0] ** 2).sum(-1).mean()) - 1.0) < 1e-3 and abs(float((tf[0] ** 2).sum(-1).mean()) - 1.0) < 1e-3


def test_fullsize_config5_fp8_n512_horizon8(pipe, dev):
    """BASELINE config 5 at FULL size: OpenVLA-7B, N = 512 = 8 prompts x 64 samples, action-chunk horizon 8 (56 action tokens per
    candidate), e4m3 weights AND per-row e4m3 activations on the fp8 MFMA (M = 512 decode rows), e4m3 own-token KV cache, verifier
    histories from the first four actions of every chunk. Size-independent properties (the CPU oracle cannot run this in seconds; the
    small-size oracle parity of every piece is tests/test_openvla_gpu.py, tests/test_fp8_gpu.py, tests/test_kernels_gpu.py):
    determinism, token range, row independence (samples of a prompt given IDENTICAL uniforms produce identical rows -- through the fp8
    GEMMs' row tiles, the per-row activation scales and the per-candidate own-token cache), prompt-permutation equivariance."""
    import bench
    del pipe                                                   # (fixture ordering only: the 7B bf16 pipeline is built once per module)
    torch.cuda.empty_cache()
    p5 = bench.Pipeline(dev, small=False, n_samples=64, weight_dtype="fp8", horizon=8)
    assert p5.own_kv == "fp8" and p5.policy.llm.fp8_weights and p5.policy.n_gen == 56
    i, P, S, G = p5.inp, 8, 64, 56
    idx1, tok1, _ = p5.decision()
    idx2, tok2, _ = p5.decision()
    assert idx1 == idx2 and torch.equal(tok1, tok2)
    assert tok1.shape == (P * S, G) and 0 <= idx1 < P * S
    lo, hi = p5.c["tok_vocab"] - p5.c["n_bins"], p5.c["tok_vocab"]
    assert int(tok1.min()) >= lo and int(tok1.max()) < hi
    assert torch.isfinite(p5.last_scores).all()
    # row independence: every sample of a prompt draws the uniforms of the prompt's first sample
    u = i["u"].view(P, S, G)[:, :1].expand(P, S, G).reshape(P * S, G).contiguous()
    tok, _ = p5.policy.sample(i["frame"], i["toks"], i["lens"], S, u, 1.0)
    tv = tok.view(P, S, G)
    assert torch.equal(tv, tv[:, :1].expand_as(tv))
    # different uniforms do give different candidates (the check above is not vacuous)
    assert not torch.equal(tok1.view(P, S, G)[:, 0], tok1.view(P, S, G)[:, 1])
    # prompt-permutation equivariance
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device=dev)
    tokp, _ = p5.policy.sample(i["frame"], i["toks"][perm], i["lens"][perm], S, u.view(P, S, G)[perm].reshape(P * S, G).contiguous(), 1.0)
    assert torch.equal(tokp.view(P, S, G), tv[perm])
    # MX block scales (down_proj input written by gate_up's GLU epilogue, o_proj input written by the attention kernel): the fused producers write the
    # bytes of the quantiser launches they replace -- the same decision, token for token and score for score, with the launches back in
    llm = p5.policy.llm
    assert llm._arr[0].down_klinear == 1 and llm._arr[0].o_klinear == 1
    s1 = p5.last_scores.clone()
    os.environ["COVER_FP8_MX_FUSE"] = "0"
    keep_graph = p5.policy.decode_graph
    p5.policy.decode_graph = False                           # (the knob is read per call: a replayed graph would still hold the fused launches)
    try:
        idx3, tok3, _ = p5.decision()
    finally:
        os.environ.pop("COVER_FP8_MX_FUSE", None)
        p5.policy.decode_graph = keep_graph
    assert idx3 == idx1 and torch.equal(tok3, tok1) and torch.equal(p5.last_scores, s1)
    del p5
    torch.cuda.empty_cache()


def _oracle_agreement(pipe, n_prompts):
    """The bench's own full-size check (bench.oracle_agreement, emitted as cpu_baseline.agreement): the CPU oracle executes ONE batched
    decision on the pipeline's first `n_prompts` prompt groups (same checkpoint, frame, prompts, uniforms), the HIP sampler runs
    teacher-forced on the oracle's tokens and the HIP verifier scores the oracle's tokens."""
    import bench
    sd, ssd, ck = bench.oracle_state(pipe)
    tok_o, tr, sel, *_ = bench.oracle_batched_decision(pipe, sd, ssd, ck, n_prompts)
    del sd, ssd
    ag = bench.oracle_agreement(pipe, tok_o, tr["logits"], sel, n_prompts)
    print(ag)
    return ag


def test_small_config_oracle_agreement(dev):
    """Plumbing of the agreement record at the small config (seconds of CPU): every data-decided pick equals the oracle's, scores 2e-2."""
    import bench
    p = bench.Pipeline(dev, small=True)
    ag = _oracle_agreement(p, 8)
    assert ag["of"] == 32 * 7 and ag["decided"] >= 0.5 * ag["of"], ag
    assert ag["agree_on_decided"] == 1.0, ag
    assert ag["score_max_abs"] < 2e-2 and (ag["winner_same"] or not ag["winner_decided"]), ag


def test_fullsize_oracle_agreement_one_prompt_group(pipe, dev):
    """OpenVLA-7B against the CPU oracle at full size: one prompt group x 4 samples, all 7 steps (run_simpler_eval_with_openpi.py:305-326,
    346-365). Every pick the data decide must be the oracle's pick bit for bit; logits within the bf16 bars of the small-config test;
    verifier scores on the oracle's tokens within 2e-2; same winner whenever the oracle's margins exceed twice the score difference."""
    ag = _oracle_agreement(pipe, 1)
    assert ag["of"] == 4 * 7 and ag["decided"] >= 14, ag
    assert ag["agree_on_decided"] == 1.0, ag
    assert ag["logit_rel_l2_max"] < 3.5e-2, ag
    assert ag["score_max_abs"] < 2e-2 and (ag["winner_same"] or not ag["winner_decided"]), ag
