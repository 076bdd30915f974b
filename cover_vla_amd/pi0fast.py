"""pi0-FAST token path on MI355X: the autoregressive action-token head of the reference's second policy family
(lerobot_custom/lerobot/common/policies/pi0fast/modeling_pi0fast.py).

  PI0FAST.generate_actions(batch)                                          :861-884
      embed_inputs(images, img_masks, tokens, pad_mask, ...)               :888-946   image tokens then token embeddings
      pi0_paligemma.generate(inputs_embeds, attention_mask, position_ids, max_new_tokens, do_sample=False)
          block_causal_update_causal_mask                                   :236-330   prefix bidirectional, generated tokens causal
          prepare_inputs_for_generation                                     :333-386   positions 1-indexed
      extract_actions -> decode_actions_with_fast                           :794-859, :735-792

Boundary here = token ids in, token ids out: `PI0FASTTokens.generate_tokens` takes what `embed_inputs` takes (camera
frames, prompt token ids, pad mask) and returns what `generate` returns (the greedy new tokens, pad after EOS). The two
tokenizers on either side -- PaliGemma's sentencepiece model and the `physical-intelligence/fast` BPE + DCT processor --
are un-vendored downloads: the text side is the caller's, the DCT half of the de-tokeniser is `fast_coefficients_to_actions`
with the BPE decoder injected.

Same kernels as the OpenVLA profile's decode loop (prefill GEMMs, weight-streaming decode GEMMs, KV cache segments, lm_head +
arg-max on the device) on PaliGemma's Gemma-2B geometry: MQA 8:1, head_dim 256, tied lm_head over the 257 152-entry
vocabulary. Work that is done once instead of per step: the prefix (image + prompt) is prefilled once and cached; HF
`generate` does the same through its KV cache.
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .models import BF, Decoder, KvGeometry, VitTower


class PI0FASTTokens:
    def __init__(self, sd: Dict[str, torch.Tensor], c: dict, *, device="cuda:0", max_batch=64, max_prompt=96, max_new_tokens=256,
                 n_cams=1):
        """sd: neutral state dict with vision.*, projector.*, lm.* (cover_vla_amd.synth.pi0_state layout / loader output);
        c: size dict (lm_dim, lm_mlp, layers, Hq, Hkv, D, vocab, vit_*, patch, image)."""
        self.c, self.dev = dict(c), torch.device(device)
        dev = self.dev
        sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
        self.n_img = (c["image"] // c["patch"]) ** 2
        self.n_cams = n_cams
        self.vit = VitTower(sub("vision."), dim=c["vit_dim"], layers=c["vit_layers"], heads=c["vit_heads"], mlp=c["vit_mlp"],
                            patch=c["patch"], act="gelu_tanh", eps=1e-6, device=device)
        self.projector = ops.pack_linear(sd["projector.weight"].to(dev), sd["projector.bias"])
        self.embed = sd["lm.embed_tokens.weight"].to(BF).contiguous().to(dev)
        self.lm_head = ops.pack_linear(self.embed)                       # tied (PaliGemma ties lm_head to embed_tokens)
        self.Tp_cap = self.n_img * n_cams + max_prompt
        geom = KvGeometry(c["Hkv"], c["D"], [max_batch, max_batch], [self.Tp_cap, max_new_tokens])
        self.lm = Decoder(sub("lm."), dim=c["lm_dim"], layers=c["layers"], Hq=c["Hq"], Hkv=c["Hkv"], D=c["D"], mlp=c["lm_mlp"],
                          act="gelu_tanh", norm="gemma", eps=1e-6, rope="hf", n_pos=self.Tp_cap + max_new_tokens + 8, device=device,
                          cache=geom)
        self.max_batch, self.max_prompt, self.max_new = max_batch, max_prompt, max_new_tokens
        self.eos_check_every = 8          # decode steps between `done.all()` read-backs (0 = never stop early)
        D = c["lm_dim"]
        self.emb_scale = float(torch.tensor(D ** 0.5, dtype=BF))          # GemmaModel: normalizer in the embedding dtype

    def _image_tokens(self, img: torch.Tensor) -> torch.Tensor:
        """[n,3,H,W] in [-1,1] -> bf16 [n, n_img, dim]: tower -> projector -> / sqrt(dim) (get_image_features, HF 4.48.3), then
        GemmaModel's * bf16(sqrt(dim)) on the embedded sequence -- two bf16 roundings, as the reference's dtype flow has them."""
        x = self.vit.embed(img.float().contiguous())
        x = self.vit.forward(x, post_ln=True)
        n, T, _ = x.shape
        D = self.c["lm_dim"]
        y = ops.gemm(x.view(n * T, -1), self.projector)
        ops.scale_bf16(y, D ** 0.5, self.emb_scale)
        return y.view(n, T, D)

    def generate_tokens(self, images: List[torch.Tensor], img_masks: List[torch.Tensor], tokens: torch.Tensor, pad_mask: torch.Tensor,
                        max_new_tokens: int, eos_token_id: int = 1, pad_token_id: int = 0,
                        force_tokens: Optional[torch.Tensor] = None, trace: Optional[dict] = None) -> torch.Tensor:
        """images: list (cameras) of [B,3,H,W]; tokens int64 [B,L] RIGHT padded with pad_mask [B,L] (the reference pads left for
        generation: positions come from the cumulative pad mask and padded keys are masked, so the side does not enter the
        arithmetic). Returns int64 [B, max_new_tokens] on the device: the greedy continuation, `pad_token_id` after a row's EOS
        (what `generate(do_sample=False)` returns after the prompt). force_tokens (tests): teacher-force the fed-back tokens."""
        dev, c = self.dev, self.c
        # Greedy decoding is a function of (frames, prompt): candidates that share both (the samples of one rephrased prompt)
        # are generated once and the tokens broadcast -- index bookkeeping on the host, B x 2L integers
        if force_tokens is None and tokens.shape[0] > 1 and all(bool(torch.equal(im[:1].expand_as(im), im)) for im in images):
            key = torch.cat([tokens, pad_mask.to(tokens.dtype)], dim=1).cpu().numpy()
            _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
            if first.shape[0] < tokens.shape[0]:
                order = np.argsort(first)                     # distinct rows in order of first occurrence
                rank = np.empty_like(order)
                rank[order] = np.arange(order.shape[0])
                first, inv = first[order], rank[inv.reshape(-1)]
                fi = torch.from_numpy(np.ascontiguousarray(first)).to(dev)
                sub_out = self.generate_tokens([im[fi] for im in images], [m[fi] for m in img_masks], tokens[fi], pad_mask[fi],
                                               max_new_tokens, eos_token_id, pad_token_id, None, trace)
                return sub_out[torch.from_numpy(np.ascontiguousarray(inv.reshape(-1))).to(dev)]
        B, L = tokens.shape
        if B > self.max_batch or L > self.max_prompt or max_new_tokens > self.max_new or len(images) > self.n_cams:
            raise ValueError("batch / prompt length / new tokens / cameras exceed the sizes this model was built for")
        if len(images) != len(img_masks) or not all(bool(m.to(torch.bool).all()) for m in img_masks):
            raise NotImplementedError("masked-out cameras are not supported on the pi0-FAST path (prepare_images :494-536 "
                                      "produces all-True masks for present cameras)")
        D = c["lm_dim"]
        n_img_all = self.n_img * len(images)
        Tp = n_img_all + L
        x = torch.empty(B, Tp, D, dtype=BF, device=dev)
        for ci, im in enumerate(images):
            same = bool(torch.equal(im[:1].expand_as(im), im))            # the evaluation driver's case: one frame for all rows
            tok = self._image_tokens(im[:1] if same else im)
            x[:, ci * self.n_img:(ci + 1) * self.n_img].copy_(tok.expand(B, -1, -1) if same else tok)   # device copy, no arithmetic
        te = ops.embed_gather(self.embed, tokens.reshape(-1).contiguous(), self.emb_scale)
        x[:, n_img_all:].copy_(te.view(B, L, D))
        if trace is not None:
            trace["prefix_embs"] = x.clone()
        plen = (n_img_all + pad_mask.to(torch.int32).sum(dim=1)).to(torch.int32).contiguous()          # valid keys: a contiguous prefix
        pos = (1 + torch.arange(Tp, dtype=torch.int32, device=dev))[None].expand(B, Tp).contiguous()   # 1-indexed (:352-354)
        g0 = self.lm.group(B, Tp, pos.view(-1), [dict(region=0, length=Tp, len_of_batch=plen)], 0)
        xf = x.view(B * Tp, D)
        self.lm.forward(xf, [g0], final_norm=False)
        # ---- first new token: the last valid prefix position of every row
        last = (torch.arange(B, device=dev, dtype=torch.int32) * Tp + plen - 1).to(torch.int32)
        h = torch.empty(B, D, dtype=BF, device=dev)
        ops.copy_rows(xf, h, B, D, last, None)
        out = torch.empty(B, max_new_tokens, dtype=torch.int64, device=dev)
        done = torch.zeros(B, dtype=torch.bool, device=dev)
        logits = torch.empty(B, self.lm_head.N, dtype=torch.float32, device=dev)
        head_ws = ops.gemm_workspace(B, self.lm_head.N, self.lm_head.K, dev)

        def pick(hidden, i):
            hn = ops.rmsnorm(hidden, self.lm.final_norm, 1e-6, w_offset=1.0, style=0)
            lg = ops.gemm(hn, self.lm_head, out=logits, ws=head_ws)
            if trace is not None:
                trace.setdefault("logits", []).append(lg[:, :c["vocab"]].clone())
            t, _ = ops.token_select(lg, 0, c["vocab"])                                # greedy over the vocabulary
            if force_tokens is not None:
                t = force_tokens[:, i].to(dev)
            t = torch.where(done, torch.full_like(t, pad_token_id), t)               # index bookkeeping: finished rows emit pad
            out[:, i].copy_(t)
            done.logical_or_(t == eos_token_id)

        pick(h, 0)
        xd = torch.empty(B, D, dtype=BF, device=dev)
        # HF generate(do_sample=False) stops once every row has emitted EOS (modeling_pi0fast.py:861-946 runs it with
        # max_new_tokens = max_decoding_steps = 256, a FAST sequence is a few dozen tokens): `done.all()` is read back every
        # `eos_check_every` steps (one 1-byte D2H) and the rest of `out` is the pad the finished rows would have emitted anyway
        out[:, 1:].fill_(pad_token_id)
        for i in range(1, max_new_tokens):
            if force_tokens is None and self.eos_check_every > 0 and i % self.eos_check_every == 0 and bool(done.all()):
                break
            ops.embed_gather(self.embed, out[:, i - 1].contiguous(), self.emb_scale, out=xd)
            pos_i = (plen + i).to(torch.int32).contiguous()                           # token i-1 sits at 1-indexed position plen + i
            g = self.lm.group(B, 1, pos_i, [dict(region=0, length=Tp, len_of_batch=plen), dict(region=1, length=i)], 1,
                              write_t_off=i - 1)
            self.lm.forward(xd, [g], final_norm=False)
            pick(xd, i)
        return out


@dataclass
class PI0FASTConfig:
    """The fields of configuration_pi0fast.PI0FASTConfig the inference path reads."""
    image_keys: Tuple[str, ...] = ("observation.images.top",)
    state_key: str = "observation.state"
    action_dim: int = 7                 # config.action_feature.shape[0]
    chunk_size: int = 10                # action horizon handed to the FAST decoder
    n_action_steps: int = 5
    max_state_dim: int = 32
    max_decoding_steps: int = 256
    fast_skip_tokens: int = 128
    relaxed_action_decoding: bool = True
    resize_imgs_with_padding: Optional[Tuple[int, int]] = (224, 224)
    device: str = "cuda:0"


class PI0FASTPolicy:
    """PI0FASTPolicy.select_action (modeling_pi0fast.py:193-233) on `PI0FASTTokens`: state discretisation + prompt text
    (`create_input_tokens` :570-640), greedy generation on the device, `extract_actions` (:794-859) and the action queue. The two
    tokenizers are the caller's objects (HF `AutoTokenizer("google/paligemma-3b-pt-224")` and the `physical-intelligence/fast`
    processor in production; `cover_vla_amd.synth.CharTokenizer` in the tests): only the methods the reference calls are used."""

    def __init__(self, config: PI0FASTConfig, model: PI0FASTTokens, paligemma_tokenizer, fast_processor, normalization: Optional[dict] = None):
        self.config, self.model = config, model
        self.paligemma_tokenizer, self.fast_tokenizer = paligemma_tokenizer, fast_processor
        self.normalization = normalization or {"state": ("IDENTITY", None, None), "action": ("IDENTITY", None, None)}
        self.pad_token_id = paligemma_tokenizer.pad_token_id if hasattr(paligemma_tokenizer, "pad_token_id") else paligemma_tokenizer.eos_token_id
        self.reset()

    def reset(self):
        self._action_queue = deque([], maxlen=self.config.n_action_steps)

    # ---- create_input_tokens(state, lang_text, actions=None) :570-640 (generation: the prefix only)
    def create_input_tokens(self, state: torch.Tensor, lang_text: Sequence[str]):
        bins = torch.linspace(-1, 1, 256 + 1, device=state.device)[:-1]
        discretized = (torch.bucketize(state, bins) - 1)[:, :32]
        prefix_texts = []
        for txt, disc in zip(lang_text, discretized):
            cleaned = txt.lower().strip().replace("_", " ")
            state_str = " ".join(str(val.detach()) for val in disc)     # (sic: the reference joins the tensors' repr, :582-585)
            prefix_texts.append(f"Task: {cleaned}, State: {state_str};\n")
        out = self.paligemma_tokenizer(prefix_texts, add_special_tokens=True, return_tensors="pt", padding="longest", truncation=False)
        ids, mask = out["input_ids"], out["attention_mask"]
        # compact every row's valid tokens to the left (a left-padding tokenizer: the side does not enter the arithmetic)
        B, Lp = ids.shape
        order = torch.argsort((mask == 0).to(torch.int8), dim=1, stable=True)
        return torch.gather(ids, 1, order), torch.gather(mask, 1, order)

    def _normalize_state(self, state):
        mode, a, b = self.normalization["state"]
        if mode == "IDENTITY":
            return state
        a, b = a.to(state.device), b.to(state.device)
        return (state - a) / (b + 1e-8) if mode == "MEAN_STD" else (state - a) / (b - a + 1e-8) * 2 - 1

    def _unnormalize_action(self, act):
        mode, a, b = self.normalization["action"]
        if mode == "IDENTITY":
            return act
        a, b = a.to(act.device), b.to(act.device)
        return act * b + a if mode == "MEAN_STD" else (act + 1) / 2 * (b - a) + a

    # ---- extract_actions(tokens, action_horizon, action_dim) :794-859
    def extract_actions(self, tokens: torch.Tensor, action_horizon: int, action_dim: int) -> torch.Tensor:
        decoded = self.paligemma_tokenizer.batch_decode(tokens, skip_special_tokens=True)
        cleaned = [seq.replace("Action:", "").replace(":", "").strip().split("|")[0].strip() for seq in decoded]
        outs = []
        for text in cleaned:
            raw = self.paligemma_tokenizer.encode(text, return_tensors="pt", padding=False)
            fast_ids = self.paligemma_tokenizer.vocab_size - 1 - self.config.fast_skip_tokens - raw      # _act_tokens_to_paligemma_tokens
            ft = self.fast_tokenizer
            acts = fast_coefficients_to_actions(fast_ids.tolist(), ft.bpe_tokenizer.decode, min_token=ft.min_token, scale=ft.scale,
                                                time_horizon=action_horizon, action_dim=action_dim,
                                                relaxed_decoding=self.config.relaxed_action_decoding)
            outs.append(torch.tensor(acts, device=tokens.device).squeeze(0))
        return torch.stack(outs, dim=0)

    @torch.no_grad()
    def select_action(self, batch: dict) -> torch.Tensor:
        if len(self._action_queue) == 0:
            dev = self.model.dev
            state = self._normalize_state(batch[self.config.state_key].to(torch.float32))
            present = [k for k in self.config.image_keys if k in batch]
            if not present:
                raise ValueError(f"All image features are missing from the batch. At least one expected. (batch: {batch.keys()})")
            images = []
            for k in present:
                img = batch[k]
                if self.config.resize_imgs_with_padding is not None and tuple(img.shape[-2:]) != tuple(self.config.resize_imgs_with_padding):
                    from . import imaging
                    img = imaging.resize_with_pad(img, *self.config.resize_imgs_with_padding, pad_value=0)
                images.append(img.to(dev))
            ids, mask = self.create_input_tokens(state, batch["task"])
            B = ids.shape[0]
            toks = self.model.generate_tokens(images, [torch.ones(B, dtype=torch.bool, device=dev) for _ in images], ids.to(dev), mask.to(dev),
                                              self.config.max_decoding_steps, eos_token_id=self.paligemma_tokenizer.eos_token_id,
                                              pad_token_id=self.pad_token_id)
            actions = self.extract_actions(toks.cpu(), self.config.chunk_size, self.config.action_dim)
            actions = actions[:, : self.config.n_action_steps, : self.config.action_dim]
            actions = self._unnormalize_action(actions.to(torch.float32))
            self._action_queue.extend(actions.transpose(0, 1))
        return self._action_queue.popleft()


def fast_coefficients_to_actions(token_lists: Sequence[Sequence[int]], bpe_decode: Callable[[Sequence[int]], str], *, min_token: int,
                                 scale: float, time_horizon: int, action_dim: int, relaxed_decoding: bool = True) -> np.ndarray:
    """decode_actions_with_fast (:735-792): FAST token ids -> BPE-decoded string whose code points + min_token are the quantised
    DCT coefficients -> (relaxed) truncate / zero-pad to time_horizon x action_dim -> idct(coeff / scale, axis 0, ortho).
    `bpe_decode` is the FAST processor's `bpe_tokenizer.decode` (un-vendored); a sequence that fails to decode yields zeros, as
    in the reference. Host numpy on a few hundred integers."""
    from scipy.fft import idct
    outs = []
    for toks in token_lists:
        try:
            coeff = np.array(list(map(ord, bpe_decode(toks)))) + min_token
            if relaxed_decoding:
                want = time_horizon * action_dim
                diff = want - coeff.shape[0]
                if diff < 0:
                    coeff = coeff[:want]
                elif diff > 0:
                    coeff = np.pad(coeff, (0, diff), mode="constant", constant_values=0)
            coeff = coeff.reshape(-1, action_dim)
            if coeff.shape != (time_horizon, action_dim):
                raise ValueError(f"decoded DCT coefficients have shape {coeff.shape}, expected ({time_horizon}, {action_dim})")
        except Exception:
            coeff = np.zeros((time_horizon, action_dim))
        outs.append(idct(coeff / scale, axis=0, norm="ortho"))
    return np.stack(outs)


def fast_tokens_to_paligemma_tokens(tokens: np.ndarray, vocab_size: int, fast_skip_tokens: int = 128) -> np.ndarray:
    """_act_tokens_to_paligemma_tokens (:538-540): FAST ids live at the top of PaliGemma's vocabulary, mirrored (an involution)."""
    return vocab_size - 1 - fast_skip_tokens - np.asarray(tokens)
