"""pi0 sampler on MI355X behind the reference's API.

  PI0FlowMatching.sample_actions(images, img_masks, lang_tokens, lang_masks, state, noise=None, noise_std=1.0)
      mirrors lerobot_custom/lerobot/common/policies/pi0/modeling_pi0.py:672-715 (tensor-level boundary)
  PI0Policy.select_action(batch, noise=None, noise_std=1.0) -> collections.deque
      mirrors modeling_pi0.py:263-307 (the drop-in the evaluation driver calls,
      CoVer_VLA/inference/experiments/robot/simpler/run_simpler_eval_with_openpi.py:322-326)

What changes underneath (results are unchanged): the reference runs vision + prefix B times on identical inputs
(run_simpler_eval_with_openpi.py:305-313); here the SigLIP tower runs once per distinct set of camera frames and the
PaliGemma prefix once per DISTINCT (frames, prompt) pair, the KV cache is a static allocation the 10 denoise steps append to
instead of torch.cat (paligemma_with_expert.py:305-308), masks are lengths, and every layer loop is one C call.
All arithmetic is libcover_hip; torch here allocates buffers and does index bookkeeping only.
"""
from __future__ import annotations

import collections
import math
import os
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import _lib as L
from . import ops
from .models import BF, Decoder, KvGeometry, VitTower, _f32


class PI0FlowMatching:
    def __init__(self, sd: Dict[str, torch.Tensor], c: dict, *, device="cuda:0", max_batch=64, max_prompts=16,
                 max_lang=72, n_cams=1, num_steps=10, max_state_dim=32, max_action_dim=32):
        """sd: neutral pi0 state dict (cover_vla_amd.synth.pi0_state layout / loader output), c: size dict."""
        self.c, self.dev = dict(c), torch.device(device)
        self.chunk, self.num_steps = c["chunk"], num_steps
        self.max_state_dim, self.max_action_dim = max_state_dim, max_action_dim
        self.W = c["ex_dim"]
        self.n_img = (c["image"] // c["patch"]) ** 2
        self.n_cams = n_cams
        dev = self.dev
        sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
        self.vit = VitTower(sub("vision."), dim=c["vit_dim"], layers=c["vit_layers"], heads=c["vit_heads"], mlp=c["vit_mlp"],
                            patch=c["patch"], act="gelu_tanh", eps=1e-6, device=device)
        self.projector = ops.pack_linear(sd["projector.weight"].to(dev), sd["projector.bias"])
        self.embed = sd["lm.embed_tokens.weight"].to(BF).contiguous().to(dev)
        self.Tp_cap = self.n_img * n_cams + max_lang
        S = 1 + self.chunk
        geom = KvGeometry(c["Hkv"], c["D"], [max_prompts, max_batch], [self.Tp_cap, S])
        n_pos = self.Tp_cap + S + 8
        self.lm = Decoder(sub("lm."), dim=c["lm_dim"], layers=c["layers"], Hq=c["Hq"], Hkv=c["Hkv"], D=c["D"], mlp=c["lm_mlp"],
                          act="gelu_tanh", norm="gemma", eps=1e-6, rope="pi0", n_pos=n_pos, device=device, cache=geom)
        self.expert = Decoder(sub("expert."), dim=c["ex_dim"], layers=c["layers"], Hq=c["Hq"], Hkv=c["Hkv"], D=c["D"],
                              mlp=c["ex_mlp"], act="gelu_tanh", norm="gemma", eps=1e-6, rope="pi0", n_pos=n_pos, device=device,
                              share_cache_with=self.lm, final_norm_bf16=False)
        # pi0's own projections stay fp32 (modeling_pi0.py:488-494)
        self.p = {n: (_f32(sd[n + ".weight"], dev), _f32(sd[n + ".bias"], dev))
                  for n in ("state_proj", "action_in_proj", "action_out_proj", "action_time_mlp_in", "action_time_mlp_out")}
        self.vis_len = torch.tensor([1] + [S] * self.chunk, dtype=torch.int32, device=dev)
        self.max_batch, self.max_prompts, self.max_lang = max_batch, max_prompts, max_lang
        self._den = {}     # static buffers (+ hipGraph) of the denoise loop per (batch size, chains)
        # denoise loop: independent row-group chains on streams of their own, replayed as one hipGraph (see sample_actions)
        self.n_chains = int(os.environ.get("COVER_PI0_CHAINS", "1"))
        self.denoise_graph = os.environ.get("COVER_PI0_GRAPH", "1") != "0"
        self._fold = {}    # folded suffix-embedding constants per step size (_suffix_fold)
        self._cap = None   # capture stream

    # ---------------------------------------------------------------------------------------------- prefix
    def _image_tokens(self, img: torch.Tensor) -> torch.Tensor:
        """img fp32 [n,3,H,W] in [-1,1] -> bf16 [n, n_img, lm_dim]: tower -> projector -> (/sqrt(D))*bf16(sqrt(D))
        (modeling_pi0.py:529-538 with HF-4.48.3 get_image_features; Appendix A.3 double rounding)."""
        x = self.vit.embed(img.float().contiguous())
        x = self.vit.forward(x, post_ln=True)
        n, T, _ = x.shape
        D = self.c["lm_dim"]
        y = ops.gemm(x.view(n * T, -1), self.projector)
        ops.scale_bf16(y, D ** 0.5, float(torch.tensor(D ** 0.5, dtype=BF)))
        return y.view(n, T, D)

    @staticmethod
    def _image_classes(cams: List[torch.Tensor], B: int) -> np.ndarray:
        """Equality classes of the rows' camera frames (all cameras together): int64 [B], class ids in order of first
        occurrence. Rows are grouped by a float64 fingerprint (two fixed projections per camera) and every group is then
        VERIFIED element-wise against its first row, so a collision can only split work, never merge different frames."""
        if B == 1:
            return np.zeros(1, dtype=np.int64)
        if all(bool(torch.equal(im[:1].expand_as(im), im)) for im in cams):
            return np.zeros(B, dtype=np.int64)                                   # the evaluation driver's case: one frame
        fps = []
        for im in cams:
            flat = im.reshape(B, -1).to(torch.float64)
            n = flat.shape[1]
            w1 = torch.cos(torch.arange(n, device=im.device, dtype=torch.float64) * 0.7548776662466927)
            w2 = torch.sin(torch.arange(n, device=im.device, dtype=torch.float64) * 0.5698402909980532)
            fps += [flat @ w1, flat @ w2]
        fp = torch.stack(fps, 1).cpu().numpy()
        cls = np.full(B, -1, dtype=np.int64)
        reps: List[int] = []
        for r in range(B):
            for k, rr in enumerate(reps):
                if np.array_equal(fp[r], fp[rr]) and all(bool(torch.equal(im[r], im[rr])) for im in cams):
                    cls[r] = k
                    break
            if cls[r] < 0:
                cls[r] = len(reps)
                reps.append(r)
        return cls

    def _suffix_fold(self, dt: float):
        """Constants of the folded suffix embedding, built once per step size on the device with the library's own fp32 GEMM.
        embed_suffix (modeling_pi0.py:593-609) computes, all in fp32, hid = silu(W1 [W_in x + b_in ; bf16(time_emb(t))] + b1). Nothing
        between x and the SiLU is non-linear or rounded to bf16, and the Euler schedule t = 1, 1 + dt, ... is fixed (:697-715), so with
        W1 = [W1a | W1b]:  hid = silu(x (W1a W_in)^T + c_t),  c_t = W1a b_in + W1b bf16(time_emb(t)) + b1 -- a 32-deep contraction per
        step instead of a 2048-deep one, and no time-embedding / concat launches. Same value up to the order of the fp32 sums."""
        key = round(float(dt), 9)
        f = self._fold.get(key)
        if f is not None:
            return f
        W = self.W
        w_in, b_in = self.p["action_in_proj"]
        w1, b1 = self.p["action_time_mlp_in"]
        w1a, w1b = w1[:, :W], w1[:, W:]
        times, t, dt32 = [], torch.tensor(1.0, dtype=torch.float32), torch.tensor(dt, dtype=torch.float32)
        while t >= -dt32 / 2:
            times.append(float(t))
            t = t + dt32
        tv = torch.tensor(times, dtype=torch.float32, device=self.dev)
        temb = ops.cast_bf16_to_f32(ops.sincos_time_embed(tv, W, 4e-3, 4.0))                   # [steps, W], bf16-rounded as the reference's
        wc = ops.gemm_f32(w1a, w_in, b_is_kn=True)                                               # [W, A] = W1a @ W_in
        v0 = ops.gemm_f32(b_in.view(1, W).contiguous(), w1a)                                      # [1, W] = W1a b_in
        ctab = ops.gemm_f32(temb, w1b, bias=b1, residual=v0.expand(len(times), W).contiguous())  # [steps, W]
        f = dict(Wc=wc.contiguous(), ctab=ctab.contiguous(), steps=len(times))
        self._fold[key] = f
        return f

    def sample_actions(self, images: List[torch.Tensor], img_masks: List[torch.Tensor], lang_tokens: torch.Tensor,
                       lang_masks: torch.Tensor, state: torch.Tensor, noise: Optional[torch.Tensor] = None,
                       noise_std: float = 1.0, trace: Optional[dict] = None, on_prefix_enqueued: Optional[Callable] = None) -> torch.Tensor:
        """on_prefix_enqueued (optional): called once the vision tower + prefix pass are queued and before the Euler loop (bench.py's
        profiled decision splits its kernel timers there; not part of the reference's signature)."""
        dev, c = self.dev, self.c
        B = state.shape[0]
        Lg = lang_tokens.shape[1]
        if B > self.max_batch or Lg > self.max_lang or len(images) != len(img_masks):
            raise ValueError(f"batch {B}/lang {Lg}/cams {len(images)} exceed the sizes this model was built for")
        if noise is None:
            noise = torch.normal(mean=0.0, std=noise_std, size=(B, self.chunk, self.max_action_dim), dtype=torch.float32,
                                 device=dev)
        # ---- cameras: the reference marks an absent camera with an all-False mask over an all -1 image
        # (modeling_pi0.py:372-385); its tokens are padding -- never attended by a valid query, positions do not advance
        # over them (cumsum of the pad mask, :685), their own rows are never read back -- so dropping that camera from the
        # prefix leaves every used output unchanged. Per-row mixed masks do not occur on this path.
        # Masks that differ across the rows of one camera (embed_prefix carries them per row, :529-547; the CoVer driver never produces
        # them): the same argument row by row -- the rows are partitioned by their camera pattern and every part is sampled with
        # exactly its present cameras (its own prefix width; same noise rows, so the parts are what one call would have produced).
        mk_h = torch.stack([m.to(torch.bool).reshape(-1) for m in img_masks], dim=1).cpu().numpy()            # [B, cameras]
        if mk_h.shape[0] != B:
            raise ValueError("img_masks must have one entry per batch row")
        pats = np.unique(mk_h, axis=0)
        if pats.shape[0] > 1:
            if trace is not None:
                raise NotImplementedError("trace with camera masks that differ across rows")
            out = torch.empty(B, self.chunk, self.max_action_dim, dtype=torch.float32, device=dev)
            for pat in pats:
                if not pat.any():
                    raise ValueError("every camera is masked out")
                idx = torch.from_numpy(np.nonzero((mk_h == pat[None]).all(axis=1))[0]).to(dev)
                on = [ci for ci in range(len(images)) if pat[ci]]
                ones = torch.ones(idx.numel(), dtype=torch.bool, device=dev)
                out[idx] = self.sample_actions([images[ci][idx] for ci in on], [ones for _ in on], lang_tokens[idx], lang_masks[idx],
                                               state[idx], noise=noise[idx], noise_std=noise_std)
            return out
        cams = [im for ci, im in enumerate(images) if pats[0][ci]]
        if not cams:
            raise ValueError("every camera is masked out")
        if len(cams) > self.n_cams:
            raise ValueError(f"{len(cams)} cameras > n_cams={self.n_cams} this model was built for")
        # ---- dedup (index bookkeeping): a prefix is computed once per distinct (camera frames, prompt) pair, the SigLIP
        # tower once per distinct set of camera frames
        # (on the host: the count U is needed there anyway, and a device row-sort of B x 2Lg integers costs ~0.6 ms of
        # rocprim kernels against ~50 us for the round trip of a few tens of KB)
        img_class = self._image_classes(cams, B)                                   # host int64 [B], 0 for a shared frame
        key = torch.cat([lang_tokens, lang_masks.to(lang_tokens.dtype)], dim=1).cpu().numpy()
        key = np.concatenate([img_class[:, None].astype(key.dtype), key], axis=1)
        _, first_h, inv_h = np.unique(key, axis=0, return_index=True, return_inverse=True)
        U = int(first_h.shape[0])
        if U > self.max_prompts:
            raise ValueError(f"{U} distinct (frames, prompt) pairs > max_prompts={self.max_prompts}")
        prompt_of_row = torch.from_numpy(np.ascontiguousarray(inv_h.reshape(-1)).astype(np.int64)).to(dev)
        first_row = torch.from_numpy(np.ascontiguousarray(first_h).astype(np.int64)).to(dev)
        n_ic = int(img_class.max()) + 1
        ic_first = torch.from_numpy(np.array([int(np.nonzero(img_class == k)[0][0]) for k in range(n_ic)], dtype=np.int64)).to(dev)
        ic_of_group = torch.from_numpy(img_class[first_h].astype(np.int64)).to(dev)
        n_img_all = self.n_img * len(cams)
        # ---- trailing pad columns: the tokenizer pads every prompt on the right to max_length (modeling_pi0.py:398-407: L = 48..72 for
        # ~20 real tokens). A pad token is never a key (pad mask, :547-560 / :731-735), positions do not advance over it (:685) and its own
        # rows are never read back, so the prefix pass only needs the columns up to the longest real prompt of the batch: at P1 sizes
        # 8 x (256 + 23) = 2232 rows instead of 8 x 328 = 2624. Only when every mask row is a contiguous run from column 0 (checked
        # on the host copy that the dedup above already made); otherwise all L columns are kept.
        Lfull = Lg
        mh = key[:, 1 + Lg:] != 0
        lens_h = mh.sum(axis=1)
        if os.environ.get("COVER_PI0_TRIM_PAD", "1") != "0" and bool(np.array_equal(mh, np.arange(Lg)[None, :] < lens_h[:, None])):
            Lg = max(int(lens_h.max()), 1)
            lang_tokens, lang_masks = lang_tokens[:, :Lg], lang_masks[:, :Lg]
        Tp = n_img_all + Lg
        D = c["lm_dim"]
        prefix = torch.empty(U, Tp, D, dtype=BF, device=dev)
        for ci, im in enumerate(cams):
            tok = self._image_tokens(im[ic_first])  # [n_ic, n_img, D]
            rows = torch.arange(self.n_img, device=dev)
            sidx = (rows[None] + ic_of_group[:, None] * self.n_img).reshape(-1)
            didx = (torch.arange(U, device=dev)[:, None] * Tp + ci * self.n_img + rows[None]).reshape(-1)
            ops.copy_rows(tok.view(-1, D), prefix.view(-1, D), U * self.n_img, D, sidx.to(torch.int32), didx.to(torch.int32))
        utok = lang_tokens[first_row].contiguous()
        umask = lang_masks[first_row]
        lang = ops.embed_gather(self.embed, utok.view(-1), math.sqrt(D))
        didx = (torch.arange(U, device=dev)[:, None] * Tp + n_img_all + torch.arange(Lg, device=dev)[None]).reshape(-1)
        ops.copy_rows(lang, prefix.view(-1, D), U * Lg, D, None, didx.to(torch.int32))
        if trace is not None:   # at the reference's full width (trimmed pad columns: zeros)
            pe = torch.zeros(B, n_img_all + Lfull, D, dtype=BF, device=dev)
            pe[:, :Tp] = prefix[prompt_of_row]
            trace["prefix_embs"] = pe
        # lengths / positions (prompts are right padded: valid keys are a contiguous prefix)
        plen = (n_img_all + umask.sum(dim=1)).to(torch.int32)
        pad = torch.cat([torch.ones(U, n_img_all, dtype=torch.bool, device=dev), umask.bool()], dim=1)
        ppos = (torch.cumsum(pad, dim=1) - 1).clamp(min=0).to(torch.int32).contiguous()
        g0 = self.lm.group(U, Tp, ppos.view(-1), [dict(region=0, length=Tp, len_of_batch=plen)], 0)
        self.lm.forward(prefix.view(U * Tp, D), [g0], final_norm=False)
        if on_prefix_enqueued is not None:
            on_prefix_enqueued()

        # ---- denoise loop (modeling_pi0.py:697-752). The rows of the batch never interact inside it (every row attends its own prompt's prefix
        # and its own suffix), and at B = 40 a layer-step is eight dependent launches of <= 224 workgroups that last 5-15 us each: the chip is
        # mostly fill and drain. The batch is therefore cut into `n_chains` contiguous row groups, each an independent chain of launches on a
        # stream of its own (own buffers = row slices of the batch's, own decoder workspace, own suffix-KV slots), forked from and joined to the
        # caller's stream, and the whole fork / loop / join is replayed as ONE hipGraph with n_chains parallel branches from the third call
        # on (call 1 runs eagerly and sizes everything, call 2 captures). Row results do not depend on the cut (tests: chains 1 vs 2 vs 4).
        # self.n_chains (COVER_PI0_CHAINS); self.denoise_graph = False (COVER_PI0_GRAPH=0) keeps the eager loop (host-bound with more than one
        # chain). MEASURED (MI355X, B = 40, profiles/r06_pi0_chains_ab.txt): 1 chain eager 29.2 ms per decision, 1 chain replayed 29.0,
        # 2 chains 29.7-30.3, 4 chains 40.4-41.5, 8 chains 54.6-55.1 -- the branches of the graph do not overlap: a launch at 100 rows lasts
        # nearly as long as one at 200 (it is fill / drain and a dependent round trip either way) and the branches overlap little, so the
        # decision grows with the number of launches. The default is therefore ONE chain, replayed.
        S, W, A = 1 + self.chunk, self.W, self.max_action_dim
        n_ch = max(1, min(self.n_chains, B))
        if trace is not None:
            n_ch = 1
        # (the A/B knobs that are read per call are part of the key: a captured graph has their value baked in)
        knobs = (os.environ.get("COVER_PI0_SUFFIX_FOLD", "1"), os.environ.get("COVER_QKV_FOLD", "1"))
        st = self._den.get((B, n_ch, knobs))
        if st is None:
            st = dict(calls=0, graph=None, ws_gen=-1,
                      row_prompt=torch.empty(B, dtype=torch.int32, device=dev), row_plen=torch.empty(B, dtype=torch.int32, device=dev),
                      spos=torch.empty(B, S, dtype=torch.int32, device=dev),
                      slots=torch.arange(B, dtype=torch.int32, device=dev),
                      suffix=torch.empty(B, S, W, dtype=torch.float32, device=dev),
                      x_t=torch.empty(B, self.chunk, A, dtype=torch.float32, device=dev),
                      cat=torch.empty(B * self.chunk, 2 * W, dtype=torch.float32, device=dev),
                      hid=torch.empty(B * self.chunk, W, dtype=torch.float32, device=dev),
                      xb=torch.empty(B * S, W, dtype=BF, device=dev), out32=torch.empty(B, S, W, dtype=torch.float32, device=dev),
                      tvec=torch.empty(B * self.chunk, dtype=torch.float32, device=dev),
                      temb=torch.empty(B * self.chunk, W, dtype=BF, device=dev))
            cuts = [(B * i) // n_ch for i in range(n_ch + 1)]
            st["chains"] = []
            for ci in range(n_ch):
                b0, b1 = cuts[ci], cuts[ci + 1]
                sl = st["slots"][b0:b1]
                g = self.expert.group(b1 - b0, S, st["spos"][b0:b1].view(-1),
                                      [dict(region=0, length=Tp, len_of_batch=st["row_plen"][b0:b1], slot_of_batch=st["row_prompt"][b0:b1]),
                                       dict(region=1, length=S, mask=ops.MASK_VISLEN, vis_len=self.vis_len, slot_of_batch=sl)], 1,
                                      write_slot=sl, write_scratch=True)   # suffix K/V: per-step temporaries in the rows' own slots
                st["chains"].append(dict(b0=b0, b1=b1, g=g, ws=self.expert.workspace((b1 - b0) * S),
                                         stream=torch.cuda.Stream(device=dev) if ci > 0 else None))
            self._den[(B, n_ch, knobs)] = st
        st["calls"] += 1
        if st.get("Tp") != Tp:                      # the prefix width is baked into the groups (and into a captured graph)
            for ch in st["chains"]:
                ch["g"].segs[0].len = Tp
            st["Tp"], st["graph"] = Tp, None
        st["row_prompt"].copy_(prompt_of_row)
        st["row_plen"].copy_(plen[prompt_of_row])
        st["spos"].copy_(st["row_plen"][:, None] + torch.arange(S, device=dev, dtype=torch.int32)[None])
        suffix, x_t, cat, hid, xb, out32, tvec, temb = (st[k] for k in ("suffix", "x_t", "cat", "hid", "xb", "out32", "tvec", "temb"))
        stp = ops.gemm_f32(state.float().contiguous(), self.p["state_proj"][0], bias=self.p["state_proj"][1])
        ops.cast_bf16_to_f32(ops.cast_f32_to_bf16(stp), out=suffix.view(B, S * W)[:, :W])  # bf16-rounded state token
        x_t.copy_(noise.to(torch.float32))
        dt = -1.0 / self.num_steps
        vs = []

        fold = self._suffix_fold(dt) if os.environ.get("COVER_PI0_SUFFIX_FOLD", "1") != "0" else None
        ch_rows = self.chunk

        def euler_chain(ch):
            """The whole Euler loop for the rows [b0, b1) of the batch, queued on the current stream."""
            b0, b1 = ch["b0"], ch["b1"]
            n = b1 - b0
            x_c, hid_c, suf_c = x_t[b0:b1], hid[b0 * ch_rows:b1 * ch_rows], suffix[b0:b1]
            xb_c, out_c = xb[b0 * S:b1 * S], out32[b0:b1]
            # the reference loops `while time >= -dt/2` on an fp32 tensor: exactly num_steps iterations (modeling_pi0.py:697-715)
            time = torch.tensor(1.0, dtype=torch.float32)
            dt32 = torch.tensor(dt, dtype=torch.float32)
            step = 0
            while time >= -dt32 / 2:
                if fold is not None:
                    # embed_suffix (modeling_pi0.py:569-629) with its two linear maps in front of the SiLU folded (see _suffix_fold):
                    # hid = silu(x_t Wc^T + c_step): two launches per step instead of six
                    ops.gemm_f32(x_c.view(n * ch_rows, A), fold["Wc"], bias=fold["ctab"][step], act="silu", out=hid_c)
                else:
                    tv_c, te_c, cat_c = tvec[b0 * ch_rows:b1 * ch_rows], temb[b0 * ch_rows:b1 * ch_rows], cat[b0 * ch_rows:b1 * ch_rows]
                    tv_c.fill_(float(time))
                    ops.sincos_time_embed(tv_c, W, 4e-3, 4.0, out=te_c)
                    ops.cast_bf16_to_f32(te_c, out=cat_c[:, W:])
                    ops.gemm_f32(x_c.view(n * ch_rows, A), self.p["action_in_proj"][0], bias=self.p["action_in_proj"][1], out=cat_c[:, :W])
                    ops.gemm_f32(cat_c, self.p["action_time_mlp_in"][0], bias=self.p["action_time_mlp_in"][1], act="silu", out=hid_c)
                step += 1
                wo, bo = self.p["action_time_mlp_out"]
                ops.gemm_f32_raw(hid_c.data_ptr(), W, 1, wo.data_ptr(), W, 1, suf_c.data_ptr() + 4 * W, W, ch_rows, W, W,
                                 bias=bo, batch=n, a_bs=ch_rows * W, c_bs=S * W)  # rows 1..chunk of every suffix
                if trace is not None and not vs:
                    trace["suffix_embs_t1"] = suffix.clone()
                self.expert.forward(xb_c, [ch["g"]], final_norm=True, x_f32=suf_c.view(n * S, W), ws=ch["ws"])
                ops.cast_bf16_to_f32(xb_c, out=out_c.view(n * S, W))
                # v_t = action_out_proj(suffix_out[:, -chunk:]) ; x_t += dt * v_t   (modeling_pi0.py:748-751, 713)
                wp, bp = self.p["action_out_proj"]
                ops.gemm_f32_raw(out_c.data_ptr() + 4 * W, W, 1, wp.data_ptr(), W, 1, x_c.data_ptr(), A, ch_rows, A, W,
                                 bias=bp, residual_ptr=x_c.data_ptr(), ld_res=A, alpha=float(dt32), batch=n, a_bs=S * W,
                                 c_bs=ch_rows * A)
                if trace is not None:
                    vs.append(x_t.clone())
                time = time + dt32

        def euler_loop():
            cur = torch.cuda.current_stream()
            for ch in st["chains"][1:]:
                ch["stream"].wait_stream(cur)
                with torch.cuda.stream(ch["stream"]):
                    euler_chain(ch)
            euler_chain(st["chains"][0])
            for ch in st["chains"][1:]:
                cur.wait_stream(ch["stream"])

        use_graph = trace is None and self.denoise_graph and st["calls"] >= 2
        if not use_graph:
            euler_loop()
        else:
            if st["graph"] is None:
                cur = torch.cuda.current_stream()
                if self._cap is None:
                    self._cap = torch.cuda.Stream(device=dev)
                self._cap.wait_stream(cur)
                with torch.cuda.stream(self._cap):
                    with ops.Graph() as gr:
                        euler_loop()
                cur.wait_stream(self._cap)
                st["graph"] = gr
            st["graph"].launch()
        x_t = x_t.clone()   # the static buffer is overwritten by the next decision
        if trace is not None:
            trace["x_steps"] = vs
        return x_t


# --------------------------------------------------------------------------------------------------- policy wrapper
def pad_vector(vector: torch.Tensor, new_dim: int) -> torch.Tensor:
    """modeling_pi0.py:153-164."""
    if vector.shape[-1] == new_dim:
        return vector
    shape = list(vector.shape)
    shape[-1] = new_dim
    out = torch.zeros(*shape, dtype=vector.dtype, device=vector.device)
    out[..., : vector.shape[-1]] = vector
    return out


class PI0Config:
    """The fields of lerobot's PI0Config the evaluation path touches (configuration_pi0.py:29-71, configs/policies.py)."""

    def __init__(self, *, image_keys=("observation.images.top",), n_action_steps=4, chunk_size=4, max_state_dim=32,
                 max_action_dim=32, tokenizer_max_length=72, num_steps=10, action_dim=7, device="cuda:0",
                 resize_imgs_with_padding=(224, 224), empty_cameras=0):
        self.image_features = list(image_keys)
        self.n_action_steps, self.chunk_size = n_action_steps, chunk_size
        self.max_state_dim, self.max_action_dim = max_state_dim, max_action_dim
        self.tokenizer_max_length, self.num_steps = tokenizer_max_length, num_steps
        self.action_dim, self.device = action_dim, device
        self.resize_imgs_with_padding = resize_imgs_with_padding
        self.empty_cameras = empty_cameras          # configuration_pi0.py:45


class PI0Policy:
    """Drop-in for lerobot's PI0Policy on the CoVer evaluation path (modeling_pi0.py:226-307)."""

    def __init__(self, config: PI0Config, model: PI0FlowMatching, tokenizer: Callable,
                 action_mean: Optional[torch.Tensor] = None, action_std: Optional[torch.Tensor] = None,
                 state_mean: Optional[torch.Tensor] = None, state_std: Optional[torch.Tensor] = None,
                 normalization: Optional[dict] = None):
        """normalization: {"state": (mode, a, b), "action": (mode, a, b)} with mode IDENTITY / MEAN_STD (a, b = mean, std) /
        MIN_MAX (a, b = min, max) -- loaders.pi0_normalization reads it from a checkpoint's Normalize / Unnormalize buffers
        (normalize.py:152-183, 226-254). INT-ACT checkpoints use IDENTITY (pi0_finetune_bridge.json:6-10)."""
        self.config, self.model, self.tokenizer = config, model, tokenizer
        norm = dict(normalization or {})
        if state_mean is not None:
            norm["state"] = ("MEAN_STD", state_mean, state_std)
        if action_mean is not None:
            norm["action"] = ("MEAN_STD", action_mean, action_std)
        dev = model.dev
        mv = lambda t: None if t is None else torch.as_tensor(t, dtype=torch.float32).to(dev)
        self._norm = {k: (m, mv(a), mv(b)) for k, (m, a, b) in norm.items()}
        self._preprocess_adapter = None
        self.reset()

    def _normalize_state(self, x):
        mode, a, b = self._norm.get("state", ("IDENTITY", None, None))
        if mode == "MEAN_STD":
            return (x - a) / (b + 1e-8)                       # normalize.py:169
        if mode == "MIN_MAX":
            return (x - a) / (b - a + 1e-8) * 2 - 1           # :177-179
        return x

    def _unnormalize_action(self, x):
        mode, a, b = self._norm.get("action", ("IDENTITY", None, None))
        if mode == "MEAN_STD":
            return x * b + a                                  # :243
        if mode == "MIN_MAX":
            return (x + 1) / 2 * (b - a) + a                  # :250-251
        return x

    @classmethod
    def from_pretrained(cls, pretrained_name_or_path: str, *, tokenizer: Callable, device="cuda:0", max_batch=64,
                        max_prompts=16, image_keys=("observation.images.top",), **kwargs) -> "PI0Policy":
        """Local directory holding the reference's `config.json` + `model.safetensors` (pretrained.py:77-150). The HF
        tokenizer the reference downloads at modeling_pi0.py:251 is injected (no network on the GPU box)."""
        from .loaders import load_pi0_pretrained
        sd, c, cfg = load_pi0_pretrained(pretrained_name_or_path)
        max_lang = int(cfg.get("tokenizer_max_length", 48))
        model = PI0FlowMatching(sd, c, device=device, max_batch=max_batch, max_prompts=max_prompts, max_lang=max_lang,
                                n_cams=len(image_keys),
                                num_steps=int(cfg.get("num_steps", 10)), max_state_dim=int(cfg.get("max_state_dim", 32)),
                                max_action_dim=int(cfg.get("max_action_dim", 32)))
        pc = PI0Config(image_keys=image_keys, n_action_steps=int(cfg.get("n_action_steps", c["chunk"])), chunk_size=c["chunk"],
                       max_state_dim=int(cfg.get("max_state_dim", 32)), max_action_dim=int(cfg.get("max_action_dim", 32)),
                       tokenizer_max_length=max_lang, num_steps=int(cfg.get("num_steps", 10)), device=device,
                       resize_imgs_with_padding=tuple(cfg.get("resize_imgs_with_padding", (c["image"], c["image"]))),
                       empty_cameras=int(cfg.get("empty_cameras", 0)))
        kwargs.setdefault("normalization", cfg.get("_normalization"))
        return cls(pc, model, tokenizer, **kwargs)

    def to(self, device):
        return self

    def eval(self):
        return self

    def reset(self):
        """modeling_pi0.py:256-258."""
        self._action_queue = collections.deque([], maxlen=self.config.n_action_steps)

    def prepare_images(self, batch):
        """modeling_pi0.py:344-387: every configured camera present in the batch is resized with padding to
        `resize_imgs_with_padding` (a no-op at the evaluated 224 x 224; otherwise the bilinear + left/top pad of
        resize_with_pad :131-150, on the device) and carries an all-True mask; up to `empty_cameras` missing cameras are
        appended as all -1 images with all-False masks."""
        from .imaging import resize_with_pad
        present = [k for k in self.config.image_features if k in batch]
        missing = [k for k in self.config.image_features if k not in batch]
        if not present:
            raise ValueError(f"All image features are missing from the batch. At least one expected. "
                             f"(batch: {batch.keys()}) (image_features:{self.config.image_features})")
        images, masks = [], []
        for k in present:
            img = batch[k]
            if self.config.resize_imgs_with_padding is not None:
                img = resize_with_pad(img, *self.config.resize_imgs_with_padding, pad_value=0)
            mask = torch.ones(img.shape[0], dtype=torch.bool, device=img.device)
            images.append(img)
            masks.append(mask)
        for num_empty in range(len(missing)):
            if num_empty >= self.config.empty_cameras:
                break
            img = torch.ones_like(img) * -1
            mask = torch.zeros_like(mask)
            images.append(img)
            masks.append(mask)
        return images, masks

    def prepare_language(self, batch):
        """modeling_pi0.py:389-409: "<task>\\n", right padded to tokenizer_max_length, truncated."""
        tasks = batch["task"]
        tasks = [t if t.endswith("\n") else f"{t}\n" for t in tasks]
        ids, mask = self.tokenizer(tasks, self.config.tokenizer_max_length)
        dev = self.model.dev
        return ids.to(dev), mask.to(dev).bool()

    @torch.no_grad()
    def select_action(self, batch: dict, noise: Optional[torch.Tensor] = None, noise_std: float = 1.0) -> collections.deque:
        """Returns the deque itself (the local modification of modeling_pi0.py:303-307); the caller copies and clears it."""
        if len(self._action_queue) == 0:
            state = self._normalize_state(batch["observation.state"])
            images, img_masks = self.prepare_images(batch)
            state = pad_vector(state, self.config.max_state_dim)
            lang_tokens, lang_masks = self.prepare_language(batch)
            actions = self.model.sample_actions(images, img_masks, lang_tokens, lang_masks, state, noise=noise,
                                                noise_std=noise_std)
            actions = actions[:, : self.config.n_action_steps, : self.config.action_dim]
            actions = self._unnormalize_action(actions)
            self._action_queue.extend(actions.transpose(0, 1))
        return self._action_queue
