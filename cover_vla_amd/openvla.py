"""OpenVLA-7B / Prismatic candidate sampler (profile P2: the shapes BASELINE.json's metric is quoted on).

No OpenVLA code exists in the reference; this profile assembles the same kernel library
into DINOv2-L/14 + SigLIP-So400m/14 (features of the second-to-last block) -> 3-layer GELU projector -> Llama-2-7B ->
7 action tokens per step (256 bins, the de-tokeniser arithmetic the reference carries at
INT-ACT/src/experiments/policies/policy_wrapper.py:259-266). Its parity pin is the CPU oracle
(oracle/cover_ref/openvla.py), itself pinned to HF transformers modules.

Work that is provably identical across candidates is done once:
  * both vision towers and the projector: once per camera frame
  * Llama prefill: the [BOS + 256 patch] prefix is causal, so its K/V do not depend on the prompt -> ONE shared
    prefix segment; each distinct prompt only adds its ~20 text tokens. Prefix rows and all prompts' text rows go
    through the 32 layers in a single pass (one read of the 13.5 GB of weights).
  * decode: all N candidates advance together (M = N rows per weight pass); every candidate attends
    [shared image prefix | its prompt's text | its own generated tokens] as three KV segments, no copies.
"""
from __future__ import annotations

from typing import Dict, Optional

import os

import numpy as np
import torch

from . import _lib as L
from . import ops
from .models import BF, Decoder, KvGeometry, VitTower, _f32

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


class OpenVLA:
    def __init__(self, sd: Dict[str, torch.Tensor], c: dict, *, device="cuda:0", max_prompts=8, max_candidates=32,
                 max_text=32, horizon=1, n_cams=1, weight_dtype="bf16", own_kv=None):
        """weight_dtype "fp8" (BASELINE config 5): the Llama projections and the lm_head are quantised to e4m3 with per-channel
        power-of-two scales (cover_vla_amd.ops.pack_linear). Passes with at most 64 rows (the decode passes up to N = 64) stream the e4m3
        weight image with bf16 activations -- bit-identical to a bf16 GEMM on the de-quantised weights. Passes with MORE than 64 rows --
        the 448-row prefill of every fp8 run, and every decode pass of config 5 (N = 512) -- also quantise the input rows of each
        projection to e4m3 (per-row power-of-two scale) and run on the MX-scaled fp8 MFMA; COVER_FP8_MFMA=0 keeps those passes on the
        bf16 MFMA with the bf16 image of the same quantised weights (weights-only quantisation). The vision towers and the projector
        stay bf16.
        own_kv "bf16" / "fp8" (large N, config 5): the candidates' own-token KV segment is kept head-major (K and V [slot][h][t][d],
        bf16 or e4m3 with per-row scales = the "fp8 KV" of config 5) and every decode pass runs the own-token VALU pass + ONE MFMA pass
        over [shared prefix | prompt text] instead of the fused 16-candidate kernel (cover_decode_own_attention). None = legacy layout."""
        self.c, self.dev = dict(c), torch.device(device)
        dev = self.dev
        sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
        self.n_patches = (c["image"] // c["patch"]) ** 2
        self.n_cams = n_cams
        # second-to-last block features: only layers-1 blocks are ever executed, so only those are packed
        self.dino = VitTower(sub("dino."), dim=c["dino_dim"], layers=c["dino_layers"], heads=c["dino_heads"], mlp=c["dino_mlp"],
                             patch=c["patch"], act="gelu_erf", eps=1e-6, layerscale=True, prefix_tokens=c["dino_prefix"],
                             device=device, n_layers_used=c["dino_layers"] - 1)
        self.siglip = VitTower(sub("siglip."), dim=c["sig_dim"], layers=c["sig_layers"], heads=c["sig_heads"], mlp=c["sig_mlp"],
                               patch=c["patch"], act="gelu_tanh", eps=1e-6, device=device, n_layers_used=c["sig_layers"] - 1)
        self.fused = c["dino_dim"] + c["sig_dim"]
        self.proj = [ops.pack_linear(sd[f"projector.fc{i}.weight"].to(dev), sd[f"projector.fc{i}.bias"]) for i in (1, 2, 3)]
        self.embed = sd["llm.embed_tokens.weight"].to(BF).contiguous().to(dev)
        fp8 = weight_dtype == "fp8"
        if weight_dtype not in ("bf16", "fp8"):
            raise ValueError("weight_dtype must be 'bf16' or 'fp8'")
        self.weight_dtype = weight_dtype
        if own_kv not in (None, "bf16", "fp8"):
            raise ValueError("own_kv must be None, 'bf16' or 'fp8'")
        self.own_kv = own_kv
        self.lm_head = ops.pack_linear(sd["lm_head.weight"].to(dev), fp8=fp8)
        # Sampling draws from the softmax over the n_bins ACTION tokens only (the last n_bins entries of the tokenizer vocabulary,
        # policy_wrapper.py:259-266): their logits are n_bins rows of the lm_head -- 2 MB instead of the 262 MB the full head streams
        # per step. Same rows, same arithmetic, same logits for those tokens; greedy decoding (arg-max over the whole vocabulary)
        # and traced runs keep the full head. Opt-in (slice_action_head below).
        lo, hi = c["tok_vocab"] - c["n_bins"], c["tok_vocab"]
        self.lm_head_actions = ops.pack_linear(sd["lm_head.weight"][lo:hi].to(dev), fp8=fp8)
        self.n_gen = 7 * horizon
        self.T0 = 1 + self.n_patches * n_cams          # [BOS] + patches
        geom = KvGeometry(c["Hkv"], c["D"], [1, max_prompts, max_candidates], [self.T0, max_text, self.n_gen])
        self.llm = Decoder(sub("llm."), dim=c["llm_dim"], layers=c["llm_layers"], Hq=c["Hq"], Hkv=c["Hkv"], D=c["D"],
                           mlp=c["llm_mlp"], act="silu", norm="llama", eps=1e-5, rope="hf", n_pos=self.T0 + max_text + self.n_gen + 8,
                           device=device, cache=geom, fp8_weights=fp8)
        self.max_prompts, self.max_candidates, self.max_text = max_prompts, max_candidates, max_text
        # the largest pass this model can run (patch rows + every prompt's text rows): sized once, so that the decode graphs captured for
        # one (P, Lt) never see the workspace move when a later decision has more prompt rows
        self.llm.reserve(self.T0 + max_prompts * max_text)
        D = c["llm_dim"]
        self.action_lo = c["tok_vocab"] - c["n_bins"]
        self.action_hi = c["tok_vocab"]
        # static buffers (fixed addresses -> the whole decision can be captured in a hipGraph)
        self.x_pre = torch.empty(self.T0 + max_prompts * max_text, D, dtype=BF, device=dev)
        self.x_dec = torch.empty(max_candidates, D, dtype=BF, device=dev)
        self.h_sel = torch.empty(max_candidates, D, dtype=BF, device=dev)
        self.logits = torch.empty(max_candidates, c["vocab"], dtype=torch.float32, device=dev)
        self.head_ws = ops.gemm_workspace(max_candidates, c["vocab"], D, dev)
        self.logits_actions = torch.empty(max_candidates, c["n_bins"], dtype=torch.float32, device=dev)
        self.head_ws_actions = ops.gemm_workspace(max_candidates, c["n_bins"], D, dev)
        self.hn = torch.empty(max_candidates, D, dtype=BF, device=dev)
        self.zero_slots = torch.zeros(max(max_prompts, max_candidates), dtype=torch.int32, device=dev)
        self.bos = torch.tensor([1], dtype=torch.int64, device=dev)
        self._side = None
        self._cap = None
        self._vis = {}
        self._bos_ready = False
        # measurement switches (bench.py's profiled decision): hipGraph replay hides launches from the in-library kernel
        # timer, and the SigLIP tower on a side stream inflates the durations of the kernels it overlaps
        self.vision_graph = os.environ.get("COVER_VISION_GRAPH", "1") != "0"
        # opt-in (COVER_ACTION_HEAD=1 or the attribute): measured 34.18-34.48 vs 34.26-34.28 ms per decision at N = 32 -- the seven
        # lm_head launches hide behind the host work between decode passes -- so the default keeps the full head
        self.slice_action_head = os.environ.get("COVER_ACTION_HEAD", "0") == "1"
        self.vision_overlap = True
        # the head + decode loop of a decision (first action token, then n_gen - 1 passes of 32 layers + head: ~1 350 launches at 7 B) replayed
        # as ONE hipGraph over static buffers: bit-identical to the eager loop; the GPU time is the same (the loop is
        # GPU-bound), the host is done with a decision's policy side after the prefill. COVER_DECODE_GRAPH=0 keeps the eager loop.
        self.decode_graph = os.environ.get("COVER_DECODE_GRAPH", "1") != "0"
        self._dec = {}

    def _ensure_bos_kv(self):
        """The BOS token sits at position 0 of a causal prefix: it attends only to itself, so its hidden states and its K/V
        in every layer depend on nothing but the weights. They are computed once (a 1-row pass) and stay in slot 0 /
        position 0 of the shared cache segment; every decision then prefills 256 patch rows + the text rows =
        448 = 7 x 64 rows instead of 449 (a whole 64-row GEMM tile for one row, 12.5 % of the prefill)."""
        if self._bos_ready:
            return
        x = ops.embed_gather(self.embed, self.bos)
        pos = torch.zeros(1, dtype=torch.int32, device=self.dev)
        g = self.llm.group(1, 1, pos, [dict(region=0, length=1, mask=ops.MASK_CAUSAL)], 0)
        self.llm.forward(x, [g], final_norm=False)
        self._bos_ready = True

    # ---------------------------------------------------------------------------------------------- vision
    def _vision_static(self, n, H, W):
        key = (n, H, W)
        st = self._vis.get(key)
        if st is None:
            c, P, dev = self.c, self.n_patches, self.dev
            st = dict(frame=torch.empty(n, H, W, 3, dtype=torch.uint8, device=dev),
                      d=self.dino.static_bufs(n, H, W), s=self.siglip.static_bufs(n, H, W),
                      fused=torch.empty(n * P, self.fused, dtype=BF, device=dev),
                      h=[torch.empty(n * P, lin.n_out, dtype=BF, device=dev) for lin in self.proj],
                      ws=[ops.gemm_workspace(n * P, lin.N, lin.K, dev) for lin in self.proj], graph=None)
            self._vis[key] = st
        return st

    def _encode_static(self, st) -> torch.Tensor:
        """Both towers + projector on persistent buffers (no allocation: recordable into a hipGraph)."""
        c = self.c
        n, P = st["frame"].shape[0], self.n_patches
        mul_d = [1.0 / (255.0 * s) for s in IMAGENET_STD]
        add_d = [-m / s for m, s in zip(IMAGENET_MEAN, IMAGENET_STD)]
        # the two towers are independent and individually too small to fill 256 CUs: run SigLIP on a side stream
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        side = self._side if self.vision_overlap else main
        side.wait_stream(main)
        with torch.cuda.stream(side):
            xs = self.siglip.embed(st["frame"], [1.0 / (255.0 * 0.5)] * 3, [-1.0] * 3, bufs=st["s"])
            xs = self.siglip.forward(xs)
        xd = self.dino.embed(st["frame"], mul_d, add_d, bufs=st["d"])
        xd = self.dino.forward(xd)
        main.wait_stream(side)
        fused = st["fused"].view(n, P, self.fused)                       # channel concat (strided device copies)
        fused[:, :, :c["dino_dim"]].copy_(xd[:, c["dino_prefix"]:, :])
        fused[:, :, c["dino_dim"]:].copy_(xs)
        h = ops.gemm(st["fused"], self.proj[0], act="gelu_erf", out=st["h"][0], ws=st["ws"][0])
        h = ops.gemm(h, self.proj[1], act="gelu_erf", out=st["h"][1], ws=st["ws"][1])
        return ops.gemm(h, self.proj[2], out=st["h"][2], ws=st["ws"][2])

    def encode_image(self, frame_u8: torch.Tensor) -> torch.Tensor:
        """frame uint8 [n_cams, H, W, 3] -> projected patch embeddings bf16 [n_cams*256, llm_dim] (a persistent buffer,
        overwritten by the next call). The ~400 launches of the two towers are recorded into a hipGraph on first use
        (COVER_VISION_GRAPH=0 disables): queued one after the other from the host, the second tower starts ~1 ms late."""
        n, H, W, _ = frame_u8.shape
        st = self._vision_static(n, H, W)
        st["frame"].copy_(frame_u8)
        gen = (self.dino.ws_gen, self.siglip.ws_gen)
        if st["graph"] is not None and st.get("ws_gen") != gen:
            st["graph"] = None                                           # a tower workspace moved (a larger frame batch ran since): re-capture
        if st["graph"] is not None and self.vision_graph:
            st["graph"].launch()
            return st["h"][2]
        out = self._encode_static(st)                                    # eager (also sizes the towers' workspaces)
        if self.vision_graph and st["graph"] is None:
            cur = torch.cuda.current_stream()
            if self._cap is None:
                self._cap = torch.cuda.Stream(device=self.dev)
            self._cap.wait_stream(cur)
            with torch.cuda.stream(self._cap):
                with ops.Graph() as g:
                    self._encode_static(st)
            st["graph"] = g
            st["ws_gen"] = (self.dino.ws_gen, self.siglip.ws_gen)
            cur.wait_stream(self._cap)
        return out

    # ---------------------------------------------------------------------------------------------- sampler
    def sample(self, frame_u8: torch.Tensor, prompt_tokens: torch.Tensor, prompt_lens: torch.Tensor, n_samples: int,
               uniforms: Optional[torch.Tensor] = None, temperature: float = 1.0, trace: Optional[dict] = None,
               force_tokens: Optional[torch.Tensor] = None, on_prefill_enqueued=None, on_vision_enqueued=None):
        """frame_u8 [n_cams,H,W,3] uint8; prompt_tokens int64 [P, Lt] right padded, prompt_lens int32 [P] (device);
        n_samples candidates per prompt (N = P*n_samples, candidate i belongs to prompt i // n_samples);
        uniforms fp32 [N, n_gen] in [0,1) for inverse-CDF sampling over the 256 action tokens, None = greedy over the
        tokenizer vocabulary. force_tokens int64 [N, n_gen] (tests): teacher-force the fed-back tokens while still
        returning this path's own picks. on_prefill_enqueued: optional callable invoked once the prefill launches are
        queued -- the point where a caller should queue independent side-stream work (the verifier towers): the
        HBM-bound decode passes that follow tolerate concurrent kernels, the MFMA-bound prefill does not. Returns (tokens int64 [N, n_gen], selected-logit fp32 [N, n_gen])."""
        c, dev = self.c, self.dev
        P, Lt = prompt_tokens.shape
        N = P * n_samples

        def mark(name):   # optional phase timing (tools/phases.py): trace["events"] collects (name, event) pairs
            if trace is not None and "events" in trace:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                trace["events"].append((name, ev))

        if P > self.max_prompts or N > self.max_candidates or Lt > self.max_text:
            raise ValueError("prompts/candidates/text length exceed the sizes this model was built for")
        D, T0 = c["llm_dim"], self.T0
        Tp = T0 - 1                                   # patch rows; the BOS row is input-independent (see _ensure_bos_kv)
        self._ensure_bos_kv()
        # ---- prefill input rows: the patches (positions 1..Tp) then P x Lt text rows
        x = self.x_pre[: Tp + P * Lt]
        mark("start")
        x[:Tp].copy_(self.encode_image(frame_u8))
        if on_vision_enqueued is not None:
            on_vision_enqueued()
        mark("vision")
        ops.embed_gather(self.embed, prompt_tokens.reshape(-1).contiguous(), out=x[Tp:])
        pos0 = 1 + torch.arange(Tp, dtype=torch.int32, device=dev)
        pos1 = (T0 + torch.arange(Lt, dtype=torch.int32, device=dev))[None].expand(P, Lt).contiguous()
        # patch row t (position t+1) sees keys 0..t+1 of the shared segment: BOS (cached) + patches up to itself
        g0 = self.llm.group(1, Tp, pos0, [dict(region=0, length=T0, mask=ops.MASK_CAUSAL, causal_offset=1)], 0, write_t_off=1)
        g1 = self.llm.group(P, Lt, pos1.view(-1),
                            [dict(region=0, length=T0, slot_of_batch=self.zero_slots),
                             dict(region=1, length=Lt, mask=ops.MASK_CAUSAL)], 1)
        self.llm.forward(x, [g0, g1], final_norm=False)
        mark("prefill")
        if on_prefill_enqueued is not None:
            on_prefill_enqueued()
        # ---- first action token: last valid text position of each candidate's prompt
        prompt_of_cand = (torch.arange(N, device=dev) // n_samples).to(torch.int32)
        cand_len = prompt_lens.to(torch.int32)[prompt_of_cand.long()].contiguous()
        last_row = (Tp + prompt_of_cand * Lt + cand_len - 1).to(torch.int32)
        pos_all = ((T0 + cand_len)[None, :] + torch.arange(self.n_gen, dtype=torch.int32, device=dev)[:, None]).contiguous()
        u_t = None if uniforms is None else uniforms.to(torch.float32).t().contiguous()
        if self.decode_graph and trace is None and force_tokens is None and not self.slice_action_head:
            # static buffers per batch shape; the per-decision values (prompt lengths -> rows / positions, uniforms) are copied in
            key = (P, n_samples, Lt, uniforms is None, float(temperature), self.slice_action_head)
            st = self._dec.get(key)
            if st is None:
                st = dict(graph=None, prompt_of_cand=prompt_of_cand.clone(), cand_len=torch.empty_like(cand_len), last_row=torch.empty_like(last_row),
                          pos_all=torch.empty_like(pos_all), tokens=torch.empty(self.n_gen, N, dtype=torch.int64, device=dev),
                          sel=torch.empty(self.n_gen, N, dtype=torch.float32, device=dev),
                          u=None if u_t is None else torch.empty_like(u_t),
                          prompt_slots=torch.arange(P, dtype=torch.int32, device=dev), prompt_lens=torch.empty(P, dtype=torch.int32, device=dev))
                self._dec[key] = st
            st["cand_len"].copy_(cand_len); st["last_row"].copy_(last_row); st["pos_all"].copy_(pos_all)
            st["prompt_lens"].copy_(prompt_lens.to(torch.int32))
            if u_t is not None:
                st["u"].copy_(u_t)
            body = lambda: self._decode_body(x, N, n_samples, Lt, st["prompt_of_cand"], st["cand_len"], st["last_row"], st["pos_all"], st["u"], temperature,
                                             st["tokens"], st["sel"], st["tokens"], None, st["prompt_slots"], st["prompt_lens"])
            if st["graph"] is not None and st.get("ws_gen") != self.llm.ws_gen:
                st["graph"] = None                                      # the decoder workspace moved under the captured pointer: re-capture
            if st["graph"] is None:
                body()                                                  # eager once (also sizes every workspace), then recorded
                cur = torch.cuda.current_stream()
                if self._cap is None:
                    self._cap = torch.cuda.Stream(device=self.dev)
                self._cap.wait_stream(cur)
                with torch.cuda.stream(self._cap):
                    with ops.Graph() as g:
                        body()
                st["graph"] = g
                st["ws_gen"] = self.llm.ws_gen
                cur.wait_stream(self._cap)
            else:
                st["graph"].launch()
                return st["tokens"].t().contiguous(), st["sel"].t().contiguous()
        # step-major buffers: row i of each is contiguous, so the kernels of step i read / write them in place (no per-step slice copies)
        tokens = torch.empty(self.n_gen, N, dtype=torch.int64, device=dev)
        sel = torch.empty(self.n_gen, N, dtype=torch.float32, device=dev)
        fed = tokens if force_tokens is None else force_tokens.t().contiguous()
        self._decode_body(x, N, n_samples, Lt, prompt_of_cand, cand_len, last_row, pos_all, u_t, temperature, tokens, sel, fed, trace,
                          torch.arange(P, dtype=torch.int32, device=dev), prompt_lens.to(torch.int32).contiguous(), mark)
        return tokens.t().contiguous(), sel.t().contiguous()

    def _decode_body(self, x, N, n_samples, Lt, prompt_of_cand, cand_len, last_row, pos_all, uniforms, temperature, tokens, sel, fed, trace,
                     prompt_slots, prompt_lens_i32, mark=lambda name: None):
        """Head on the last prompt rows, then n_gen - 1 decode passes + heads. Launches only (no allocation, no host read): recordable."""
        D, T0 = self.c["llm_dim"], self.T0
        ops.copy_rows(x, self.h_sel, N, D, last_row, None)
        self._head_select(self.h_sel[:N], uniforms, 0, temperature, tokens, sel, trace)
        xd = self.x_dec[:N]
        own = {}
        if self.own_kv is not None:   # regular structure of the batch: the n_samples candidates of prompt p are rows [p S, (p + 1) S)
            own = dict(own_kv=self.own_kv, seg1_group=n_samples, seg1_slot_of_group=prompt_slots, seg1_len_of_group=prompt_lens_i32)
        for i in range(1, self.n_gen):
            ops.embed_gather(self.embed, fed[i - 1], out=xd)
            g = self.llm.group(N, 1, pos_all[i - 1],
                               [dict(region=0, length=T0, slot_of_batch=self.zero_slots),
                                dict(region=1, length=Lt, len_of_batch=cand_len, slot_of_batch=prompt_of_cand),
                                dict(region=2, length=i)], 2, write_t_off=i - 1, seg0_shared=True, **own)
            self.llm.forward(xd, [g], final_norm=False)
            self._head_select(xd, uniforms, i, temperature, tokens, sel, trace)
            mark(f"decode{i}")

    def _head_select(self, h, uniforms, i, temperature, tokens, sel, trace):
        N = h.shape[0]
        hn = ops.rmsnorm(h, self.llm.final_norm, 1e-5, w_offset=0.0, style=1, out=self.hn[:N])
        if uniforms is not None and (trace is None or "events" in trace) and self.slice_action_head:
            lg = ops.gemm(hn, self.lm_head_actions, out=self.logits_actions[:N], ws=self.head_ws_actions)
            t, _ = ops.token_select(lg, 0, self.c["n_bins"], uniform=uniforms[i], temperature=temperature, out_logit=sel[i])
            torch.add(t, self.action_lo, out=tokens[i])
            return
        lg = ops.gemm(hn, self.lm_head, out=self.logits[:N], ws=self.head_ws)
        if trace is not None and "events" not in trace:
            trace.setdefault("logits", []).append(lg.clone())
        if uniforms is None:
            ops.token_select(lg, 0, self.c["tok_vocab"], out_tok=tokens[i], out_logit=sel[i])
        else:
            ops.token_select(lg, self.action_lo, self.action_hi, uniform=uniforms[i], temperature=temperature, out_tok=tokens[i],
                             out_logit=sel[i])

    # ---------------------------------------------------------------------------------------------- de-tokeniser
    def tokens_to_actions(self, tokens: np.ndarray) -> np.ndarray:
        """256-bin de-tokeniser (policy_wrapper.py:259-266 arithmetic): discretized = vocab - token, clip(d-1, 0, 254),
        bin centres of linspace(-1, 1, 256). Host numpy on N x 7 integers."""
        bins = np.linspace(-1, 1, self.c["n_bins"])
        centers = (bins[:-1] + bins[1:]) / 2.0
        d = self.c["tok_vocab"] - np.asarray(tokens)
        d = np.clip(d - 1, a_min=0, a_max=centers.shape[0] - 1)
        return centers[d]
