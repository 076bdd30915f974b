"""CoVer verifier on MI355X behind the reference's API (bridge_verifier/ensemble_eval/efficient_ensemble_merged.py).

  EfficientEnsembleMerged(merged_checkpoint, device=...)                      :25-186
      .extract_shared_features(img_tensor, text_tokens) -> (patch_features, text_features)      :188-192
      .get_embeddings_from_model_batch(model_idx, pf, tf, histories)                            :194-247
      .fuse_embeddings / .predict                                                               :249-307
      .compute_max_similarity_scores_batch(images, instructions, histories, group) ->
            (max_score: float, max_instruction: str, max_action_history: ndarray, global_action_idx: 0-dim int64) :309-454

Same results, different schedule: the reference repeats the (image, text) heads N times on identical inputs
(:213-214) and loops members sequentially; here the image-text embedding of a member is computed once per distinct
(image, text) pair and only the trajectory encoder runs per candidate. Heads, fusion and scoring stay fp32 as in
the reference; the frozen SigLIP2 towers run in bf16 through the shared ViT kernels.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import os

import numpy as np
import torch

from . import _lib as L
from . import ops
from .models import BF, VitTower, _f32


def _dev_sd(sd, dev):
    return {k: (_f32(v, dev) if torch.is_tensor(v) else v) for k, v in sd.items()}


class _Pooling:
    """AttentionPooling (model.py:76-112) with num_readouts = 1: one learned query cross-attends the kv tokens."""

    def __init__(self, sd, dev, heads=8):
        self.sd = _dev_sd(sd, dev)
        self.heads = heads
        self.n_layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
        self.dim = sd["query"].shape[-1]
        # K and V projections of all blocks read the same kv tokens: one stacked weight -> one GEMM per pooling
        ws, bs = [], []
        for i in range(self.n_layers):
            p = f"blocks.{i}.attention."
            b_in = self.sd[p + "in_proj_bias"]
            ws += [self.sd[p + "k_proj_weight"], self.sd[p + "v_proj_weight"]]
            bs += [b_in[self.dim:2 * self.dim], b_in[2 * self.dim:]]
        self.w_kv = torch.cat(ws, 0).contiguous()
        self.b_kv = torch.cat(bs, 0).contiguous()

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """x fp32 [T, input_dim] -> [1, dim]."""
        sd, E, H = self.sd, self.dim, self.heads
        T = x.shape[0]
        kv = ops.gemm_f32(x, self.w_kv, bias=self.b_kv)  # [T, n_layers*2*E]
        q = sd["query"].view(1, E)
        ld = kv.shape[1]
        for i in range(self.n_layers):
            p = f"blocks.{i}."
            q = ops.layernorm_f32(q, sd[p + "q_layer_norm.weight"], sd[p + "q_layer_norm.bias"])
            qp = ops.gemm_f32(q, sd[p + "attention.q_proj_weight"], bias=sd[p + "attention.in_proj_bias"][:E])
            k = kv[:, (2 * i) * E:(2 * i + 1) * E]
            v = kv[:, (2 * i + 1) * E:(2 * i + 2) * E]
            a = ops.mha_f32(qp, k, v, 1, 1, T, H, E // H, (E, E), (T * ld, ld), (T * ld, ld))
            q = ops.gemm_f32(a.view(1, E), sd[p + "attention.out_proj.weight"], bias=sd[p + "attention.out_proj.bias"], residual=q)
            q = ops.layernorm_f32(q, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"])
            h = ops.gemm_f32(q, sd[p + "mlp.fc1.weight"], bias=sd[p + "mlp.fc1.bias"], act="gelu_erf")
            q = ops.gemm_f32(h, sd[p + "mlp.fc2.weight"], bias=sd[p + "mlp.fc2.bias"], residual=q)
        return ops.layernorm_f32(q, sd["layer_norm.weight"], sd["layer_norm.bias"])


class _Member:
    def __init__(self, comp: dict, dev):
        self.dev = dev
        ta = comp["text_aware_visual_extraction"]
        self.inv_temp = 1.0 / float(torch.as_tensor(ta["temperature"]).clamp(0, 100))
        self.pos_emb = _f32(ta["pos_emb"], dev)
        self.vision = _Pooling(comp["vision_poolings"], dev)
        self.text = _Pooling(comp["text_pooling"], dev)
        self.ip = _dev_sd(comp["input_projection"], dev)
        self.mlp = None
        if comp.get("trajectory_encoder") is not None:
            self.se = _dev_sd(comp["single_step_action_encoder"], dev)
            self.traj = _dev_sd(comp["trajectory_encoder"], dev)
            self.traj_layers = 1 + max(int(k.split(".")[1]) for k in comp["trajectory_encoder"])
        else:
            # MLP action encoder (use_transformer = False): Sequential(Linear(h*a, 512), LayerNorm, ReLU, Dropout, Linear)
            # -> state-dict keys 0.*, 1.*, 4.* (efficient_ensemble_merged.py:148-184)
            self.mlp = _dev_sd(comp["complex_action_encoder"], dev)
            self.se, self.traj, self.traj_layers = None, None, 0
        self.pad_value = float(comp["action_padding_value"])

    def image_text(self, pf: torch.Tensor, tf: torch.Tensor) -> torch.Tensor:
        """pf fp32 [P, D], tf fp32 [T, D] unit rows -> unit [1, 512] (efficient_ensemble_merged.py:216-223)."""
        sim = ops.gemm_f32(tf, pf)                               # [T, P]
        ops.softmax_rows_f32(sim, self.inv_temp)
        pfpe = ops.add_f32(pf, self.pos_emb)
        taf = ops.gemm_f32(sim, pfpe, b_is_kn=True)              # [T, D]
        vt = self.vision(taf)
        tt = self.text(tf)
        comb = torch.empty(1, vt.shape[1] + tt.shape[1], dtype=torch.float32, device=self.dev)
        comb[:, :tt.shape[1]].copy_(tt)   # cat([text_token, vision_token]) (device copy, no arithmetic)
        comb[:, tt.shape[1]:].copy_(vt)
        y = ops.gemm_f32(comb, self.ip["weight"], bias=self.ip["bias"])
        return ops.l2norm_rows_f32(y)

    def trajectory(self, hist: torch.Tensor, pad: torch.Tensor) -> torch.Tensor:
        """hist fp32 [N, 10, 7], pad uint8 [N, 10] -> unit [N, 512] (efficient_ensemble_merged.py:226-245)."""
        N, T, A = hist.shape
        if self.mlp is not None:   # flat_actions -> complex_action_encoder -> L2 norm (:241-245); padding rows are plain inputs
            m = self.mlp
            h = ops.gemm_f32(hist.reshape(N, T * A), m["0.weight"], bias=m["0.bias"])
            h = ops.act_f32(ops.layernorm_f32(h, m["1.weight"], m["1.bias"]), "relu")
            return ops.l2norm_rows_f32(ops.gemm_f32(h, m["4.weight"], bias=m["4.bias"]))
        sd, E, H = self.traj, self.se["weight"].shape[0], 8
        x = ops.gemm_f32(hist.view(N * T, A), self.se["weight"], bias=self.se["bias"])
        for i in range(self.traj_layers):
            p = f"layers.{i}."
            qkv = ops.gemm_f32(x, sd[p + "self_attn.in_proj_weight"], bias=sd[p + "self_attn.in_proj_bias"])  # [N*T, 3E]
            a = ops.mha_f32(qkv, qkv[:, E:], qkv[:, 2 * E:], N, T, T, H, E // H, (T * 3 * E, 3 * E), (T * 3 * E, 3 * E),
                            (T * 3 * E, 3 * E), key_pad=pad)
            y = ops.gemm_f32(a.view(N * T, E), sd[p + "self_attn.out_proj.weight"], bias=sd[p + "self_attn.out_proj.bias"], residual=x)
            x = ops.layernorm_f32(y, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
            h = ops.gemm_f32(x, sd[p + "linear1.weight"], bias=sd[p + "linear1.bias"], act="relu")
            y = ops.gemm_f32(h, sd[p + "linear2.weight"], bias=sd[p + "linear2.bias"], residual=x)
            x = ops.layernorm_f32(y, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        m = ops.masked_mean_f32(x, pad, N, T, E)
        return ops.l2norm_rows_f32(m)


class _PoolingStack:
    """`_Pooling` of all members in batched launches (member = batch index)."""

    def __init__(self, pools: List["_Pooling"]):
        p0 = pools[0]
        self.G, self.L, self.E, self.H = len(pools), p0.n_layers, p0.dim, p0.heads
        st = lambda k: torch.stack([pp.sd[k] for pp in pools]).contiguous()
        self.sd = {k: st(k) for k in p0.sd if torch.is_tensor(p0.sd[k])}
        self.w_kv = torch.stack([pp.w_kv for pp in pools]).contiguous()
        self.b_kv = torch.stack([pp.b_kv for pp in pools]).contiguous()
        self.q_b = [torch.stack([pp.sd[f"blocks.{i}.attention.in_proj_bias"][:p0.dim] for pp in pools]).contiguous()
                    for i in range(p0.n_layers)]

    def __call__(self, x: torch.Tensor, shared: bool) -> torch.Tensor:
        """x fp32 [T, Din] (shared by every member) or [G, T, Din] -> [G, E]."""
        sd, G, E, H = self.sd, self.G, self.E, self.H
        T, Din = x.shape[-2], x.shape[-1]
        ld = self.w_kv.shape[1]
        kv = ops.gemm_f32(x, self.w_kv, bias=self.b_kv, batch=G, a_bs=0 if shared else T * Din, b_bs=ld * Din, c_bs=T * ld,
                          bias_bs=ld, M=T, N=ld, K=Din)                                  # [G, T, L*2E]
        q = sd["query"].reshape(G, E)
        bg = lambda a_, w_, b_, N_, K_, **kw: ops.gemm_f32(a_, w_, bias=b_, batch=G, a_bs=K_, b_bs=N_ * K_, c_bs=N_, bias_bs=N_,
                                                           M=1, N=N_, K=K_, **kw).view(G, N_)
        for i in range(self.L):
            p = f"blocks.{i}."
            q = ops.layernorm_f32_grouped(q, sd[p + "q_layer_norm.weight"], sd[p + "q_layer_norm.bias"], 1)
            qp = bg(q.view(G, 1, E), sd[p + "attention.q_proj_weight"], self.q_b[i], E, E)
            k = kv[:, :, (2 * i) * E:(2 * i + 1) * E]
            v = kv[:, :, (2 * i + 1) * E:(2 * i + 2) * E]
            a = ops.mha_f32(qp, k, v, G, 1, T, H, E // H, (E, E), (T * ld, ld), (T * ld, ld))
            q = bg(a.view(G, 1, E), sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"], E, E, residual=q.view(G, 1, E))
            q = ops.layernorm_f32_grouped(q, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1)
            F = sd[p + "mlp.fc1.weight"].shape[1]
            h = bg(q.view(G, 1, E), sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], F, E, act="gelu_erf")
            q = bg(h.view(G, 1, F), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], E, F, residual=q.view(G, 1, E))
        return ops.layernorm_f32_grouped(q, sd["layer_norm.weight"], sd["layer_norm.bias"], 1)


class _ImageTextStack:
    """`_Member.image_text` of all members in batched launches: a third of the ~150 tiny fp32 launches of the member loop,
    same per-element arithmetic (bit-identical)."""

    def __init__(self, members: List["_Member"]):
        m0 = members[0]
        self.G, self.dev = len(members), m0.dev
        same = lambda a, b: set(a) == set(b) and all((not torch.is_tensor(a[k])) or a[k].shape == b[k].shape for k in a)
        self.ok = all(same(m0.vision.sd, mm.vision.sd) and same(m0.text.sd, mm.text.sd) and mm.pos_emb.shape == m0.pos_emb.shape and
                      mm.ip["weight"].shape == m0.ip["weight"].shape and mm.vision.heads == m0.vision.heads and
                      mm.text.heads == m0.text.heads for mm in members)
        if not self.ok:
            return
        self.inv_temp = [mm.inv_temp for mm in members]
        self.pos = torch.cat([mm.pos_emb.reshape(-1, mm.pos_emb.shape[-1]) for mm in members], 0).contiguous()   # [G*P, D]
        self.vision, self.text = _PoolingStack([mm.vision for mm in members]), _PoolingStack([mm.text for mm in members])
        self.ip_w = torch.stack([mm.ip["weight"] for mm in members]).contiguous()
        self.ip_b = torch.stack([mm.ip["bias"] for mm in members]).contiguous()

    def __call__(self, pf: torch.Tensor, tf: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """pf fp32 [P, D], tf fp32 [T, D] unit rows -> out fp32 [G, 512] unit rows."""
        G = self.G
        P, D = pf.shape
        T = tf.shape[0]
        sim = ops.gemm_f32(tf, pf, batch=G, a_bs=0, b_bs=0, c_bs=T * P, M=T, N=P, K=D)   # the same product, one copy per member
        for gi in range(G):
            ops.softmax_rows_f32(sim[gi], self.inv_temp[gi])
        pfpe = ops.add_f32(self.pos, pf)                                                 # pos_emb[g] + pf (b rows cycle)
        taf = ops.gemm_f32(sim, pfpe.view(G, P, D), b_is_kn=True, batch=G, a_bs=T * P, b_bs=P * D, c_bs=T * D, M=T, N=D, K=P)
        vt = self.vision(taf, shared=False)
        tt = self.text(tf, shared=True)
        E1, E2 = tt.shape[1], vt.shape[1]
        comb = torch.empty(G, E1 + E2, dtype=torch.float32, device=self.dev)
        comb[:, :E1].copy_(tt)     # cat([text_token, vision_token]) (device copies, no arithmetic)
        comb[:, E1:].copy_(vt)
        No = self.ip_w.shape[1]
        y = ops.gemm_f32(comb.view(G, 1, E1 + E2), self.ip_w, bias=self.ip_b, batch=G, a_bs=E1 + E2, b_bs=No * (E1 + E2), c_bs=No,
                         bias_bs=No, M=1, N=No, K=E1 + E2)
        return ops.l2norm_rows_f32(y.view(G, No), out=out)


class _TrajectoryStack:
    """The trajectory encoders of ALL ensemble members as batched launches (members have one architecture): every GEMM /
    attention / norm of `_Member.trajectory` runs once with the member index as the batch dimension, i.e. a third of the
    launches of the member loop with the same per-element arithmetic (bit-identical results). The chains are latency-bound
    (~10 us per launch), so this is ~3x off the verifier tail."""

    def __init__(self, members: List["_Member"]):
        m0 = members[0]
        self.n, self.layers, self.dev = len(members), m0.traj_layers, m0.dev
        self.ok = all(mm.mlp is None for mm in members) and all(mm.traj_layers == m0.traj_layers and mm.se["weight"].shape == m0.se["weight"].shape and
                      all(mm.traj[k].shape == m0.traj[k].shape for k in m0.traj) for mm in members)
        if not self.ok:
            return
        st = lambda d, k: torch.stack([mm_[k] for mm_ in d]).contiguous()
        self.se_w, self.se_b = st([mm.se for mm in members], "weight"), st([mm.se for mm in members], "bias")
        self.w = {k: st([mm.traj for mm in members], k) for k in m0.traj}
        self.E = m0.se["weight"].shape[0]

    def __call__(self, hist: torch.Tensor, pad: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """hist fp32 [N,10,7], pad uint8 [N,10] -> out fp32 [M, N, E] unit rows (member m = `_Member.trajectory` of member m)."""
        G, E, H = self.n, self.E, 8
        N, T, A = hist.shape
        R = N * T
        w = self.w
        x = ops.gemm_f32(hist.view(R, A), self.se_w, bias=self.se_b, batch=G, a_bs=0, b_bs=E * A, c_bs=R * E, bias_bs=E,
                         M=R, N=E, K=A)                                                # [G, R, E]
        pad_g = pad.repeat(G, 1).contiguous()
        for i in range(self.layers):
            p = f"layers.{i}."
            qkv = ops.gemm_f32(x, w[p + "self_attn.in_proj_weight"], bias=w[p + "self_attn.in_proj_bias"], batch=G,
                               a_bs=R * E, b_bs=3 * E * E, c_bs=R * 3 * E, bias_bs=3 * E, M=R, N=3 * E, K=E)
            q2 = qkv.view(G * R, 3 * E)
            a = ops.mha_f32(q2, q2[:, E:], q2[:, 2 * E:], G * N, T, T, H, E // H, (T * 3 * E, 3 * E), (T * 3 * E, 3 * E),
                            (T * 3 * E, 3 * E), key_pad=pad_g)
            y = ops.gemm_f32(a.view(G, R, E), w[p + "self_attn.out_proj.weight"], bias=w[p + "self_attn.out_proj.bias"],
                             residual=x, batch=G, a_bs=R * E, b_bs=E * E, c_bs=R * E, bias_bs=E, M=R, N=E, K=E)
            x = ops.layernorm_f32_grouped(y.view(G * R, E), w[p + "norm1.weight"], w[p + "norm1.bias"], R).view(G, R, E)
            F = w[p + "linear1.weight"].shape[1]
            h = ops.gemm_f32(x, w[p + "linear1.weight"], bias=w[p + "linear1.bias"], act="relu", batch=G, a_bs=R * E,
                             b_bs=F * E, c_bs=R * F, bias_bs=F, M=R, N=F, K=E)
            y = ops.gemm_f32(h, w[p + "linear2.weight"], bias=w[p + "linear2.bias"], residual=x, batch=G, a_bs=R * F,
                             b_bs=E * F, c_bs=R * E, bias_bs=E, M=R, N=E, K=F)
            x = ops.layernorm_f32_grouped(y.view(G * R, E), w[p + "norm2.weight"], w[p + "norm2.bias"], R).view(G, R, E)
        m = ops.masked_mean_f32(x.view(G * R, E), pad_g, G * N, T, E)
        return ops.l2norm_rows_f32(m, out=out.view(G * N, E))


class VLASigLIP2Bridge:
    """One verifier model as it is trained and validated (bridge_verifier/ensemble_eval/finetune_trajectory_bridge_ddp.py:182-421):
    forward(image, text, action_histories) -> (image_logits, action_logits), both [B, B], for B DISTINCT (image, text, history)
    triples, and the symmetric InfoNCE loss + retrieval accuracies of its validation loop (:446-469, :1081-1090). Inference
    (eval) mode only: dropout is the identity, no gradients -- the optimiser side of training is outside the hot path.

    `component` is one entry of the merged checkpoint's `ensemble_components` (the trainable heads, merge_ensemble_checkpoints
    layout); `logit_scale` is the model's learned log-temperature (init 2.6592, :210)."""

    def __init__(self, component: dict, logit_scale: float = 2.6592, device="cuda:0", encoder: Optional["SigLIP2Encoder"] = None):
        L.lib()
        self.device = torch.device(device)
        self.member = _Member(component, self.device)
        self.logit_scale = float(logit_scale)
        self.encoder = encoder
        self.history_length, self.action_dim = 10, 7

    def forward_features(self, patch_features, text_features, action_histories):
        """patch_features [B, P, D], text_features [B, T, D] (unit rows, what extract_features returns :297-355),
        action_histories [B, H, A] padded with the member's padding value -> (image_logits, action_logits) fp32 [B, B]."""
        pf = _f32(patch_features, self.device)
        tf = _f32(text_features, self.device)
        hist = _f32(torch.as_tensor(np.asarray(action_histories.cpu() if torch.is_tensor(action_histories) else action_histories)), self.device)
        B = pf.shape[0]
        if tf.shape[0] != B or hist.shape[0] != B:
            raise ValueError(f"batch sizes differ: images {B}, texts {tf.shape[0]}, histories {hist.shape[0]}")
        it = torch.empty(B, 512, dtype=torch.float32, device=self.device)
        for b in range(B):                                   # per-sample text-aware heads (:368-377)
            it[b:b + 1].copy_(self.member.image_text(pf[b], tf[b]))
        pad = (hist[:, :, 0] == self.member.pad_value).to(torch.uint8).contiguous()   # :384 (a comparison, no arithmetic)
        act = self.member.trajectory(hist.contiguous(), pad)                           # :380-412
        scale = float(np.exp(np.float32(self.logit_scale)))                            # :414
        image_logits = ops.gemm_f32(it, act, alpha=scale)                              # :416
        action_logits = ops.gemm_f32(act, it, alpha=scale)                             # :417
        return image_logits, action_logits

    def forward(self, image, text, action_histories):
        if self.encoder is None:
            raise RuntimeError("VLASigLIP2Bridge.forward needs a SigLIP2Encoder; call forward_features with extracted features instead")
        pf, tf = self.encoder.extract_features(image, text)
        return self.forward_features(pf, tf, action_histories)

    __call__ = forward

    @staticmethod
    def contrastive_metrics(image_logits, action_logits, k_values=(1, 5)) -> Dict[str, float]:
        """loss = (CE(image_logits, arange) + CE(action_logits, arange)) / 2 (:895-899) and the top-k retrieval accuracies of
        calculate_accuracy_metrics (:446-469). Row statistics come from one kernel per direction; only the B-element means run on
        the host. Ties at the k-th place are broken towards the lower column (torch.topk leaves the order of equal values open)."""
        B = image_logits.shape[0]
        li, ri = ops.xent_diag_f32(image_logits)
        la, ra = ops.xent_diag_f32(action_logits)
        li, ri, la, ra = (t.cpu().numpy() for t in (li, ri, la, ra))
        out = {"image_loss": float(li.astype(np.float64).mean()), "action_loss": float(la.astype(np.float64).mean())}
        out["loss"] = 0.5 * (out["image_loss"] + out["action_loss"])
        for k in k_values:
            if k <= B:
                out[f"img2act_top{k}_acc"] = float((ri < k).mean())
                out[f"act2img_top{k}_acc"] = float((ra < k).mean())
        return out


class SigLIP2Encoder:
    """The frozen shared encoder: SigLIP2 ViT-L/16-384 image tower (patch features = the LAST block's attention-module
    output, forward hook at finetune_trajectory_bridge_ddp.py:272-274) and text tower (transformer output -> ln_final
    -> text_projection on all 64 positions, :318-330), both bf16, features cast to fp32 and L2-normalised per token.
    State dict: image.* / text.* in synth.vit_state layout; text.tok_emb [vocab, dim]; text.proj.{weight,bias}."""

    def __init__(self, sd: Dict[str, torch.Tensor], *, dim=1024, layers=24, heads=16, mlp=4096, patch=16, image=384,
                 context_length=64, device="cuda:0"):
        dev = torch.device(device)
        self.dev, self.dim, self.context_length, self.image_size = dev, dim, context_length, image
        self.num_patches = (image // patch) ** 2
        sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
        self.layers = layers
        self.image = VitTower(sub("image."), dim=dim, layers=layers, heads=heads, mlp=mlp, patch=patch, act="gelu_tanh",
                              eps=1e-6, device=device)
        tsd = sub("text.")
        tsd.setdefault("patch.weight", torch.zeros(dim, 3 * patch * patch))  # text tower has no patch embedding
        tsd.setdefault("patch.bias", torch.zeros(dim))
        self.text = VitTower(tsd, dim=dim, layers=layers, heads=heads, mlp=mlp, patch=patch, act="gelu_tanh", eps=1e-6,
                             device=device)
        self.tok_emb = sd["text.tok_emb"].to(BF).contiguous().to(dev)
        self.text_pos = sd["text.pos"].to(BF).contiguous().to(dev)
        self.text_proj = ops.pack_linear(sd["text.proj.weight"].to(dev), sd["text.proj.bias"])

    def extract_features(self, images: torch.Tensor, text: torch.Tensor, text_stream: Optional[torch.cuda.Stream] = None):
        """images fp32 [B,3,384,384] (open_clip-preprocessed), text int64 [B,64] -> (pf fp32 [B,P,D], tf fp32 [B,64,D]).
        text_stream: run the text tower there, beside the image tower on the current stream (the two towers are independent chains of
        launches that each fill a fraction of the chip); joined before returning."""
        B = images.shape[0]
        cur = torch.cuda.current_stream()
        ts = text_stream if text_stream is not None else cur
        if ts is not cur:
            ts.wait_stream(cur)
        with torch.cuda.stream(ts):
            T = text.shape[1]
            t = ops.embed_gather(self.tok_emb, text.reshape(-1).contiguous())
            ops.add_rows(t, self.text_pos[:T])
            t = self.text.forward(t.view(B, T, self.dim), post_ln=True)      # transformer -> ln_final
            tp = ops.gemm(t.view(B * T, self.dim), self.text_proj)           # text_projection (Linear)
            tf = ops.l2norm_rows_f32(ops.cast_bf16_to_f32(tp)).view(B, T, self.dim)
        x = self.image.embed(images.float().contiguous())
        a = self.image.forward(x, n_layers=self.layers, last_attn_only=True)
        pf = ops.cast_bf16_to_f32(a.view(B * self.num_patches, self.dim))
        pf = ops.l2norm_rows_f32(pf).view(B, self.num_patches, self.dim)
        if ts is not cur:
            cur.wait_stream(ts)
        return pf, tf


class EfficientEnsembleMerged:
    def __init__(self, merged_checkpoint, device="cuda:0", encoder: Optional[SigLIP2Encoder] = None,
                 preprocess: Optional[Callable] = None, tokenizer: Optional[Callable] = None):
        """merged_checkpoint: the already-loaded dict, or a path -- the directory written once by loaders.verifier_pt_to_safetensors, or the
        merged .pt itself, read with torch.load(weights_only=True) (efficient_ensemble_merged.py:37-53 unpickles it; this side never does). encoder/preprocess/tokenizer: the SigLIP2 towers and the open_clip
        CPU transforms are injected (they are un-vendored pip dependencies of the reference, :57,69)."""
        self.device = device
        dev = torch.device(device)
        if isinstance(merged_checkpoint, str):   # a converted directory (safetensors + JSON) or a .pt read WITHOUT arbitrary unpickling
            from .loaders import load_verifier_checkpoint
            ck = load_verifier_checkpoint(merged_checkpoint)
        else:
            ck = merged_checkpoint
        if "ensemble_components" in ck and "backbone" not in ck:
            self.backbone, self.use_transformer, self.history_length, self.action_dim = \
                "hf-hub:timm/ViT-L-16-SigLIP2-384", True, 10, 7
            self.num_models = len(ck["ensemble_components"])
        else:
            self.backbone, self.use_transformer = ck["backbone"], ck["use_transformer"]
            self.history_length, self.action_dim, self.num_models = ck["history_length"], ck["action_dim"], ck["num_models"]
        self.trainable_models = [_Member(c, dev) for c in ck["ensemble_components"]]
        self._traj_stack = _TrajectoryStack(self.trainable_models) if self.num_models > 1 else None
        self._it_stack = _ImageTextStack(self.trainable_models) if self.num_models > 1 else None
        if preprocess is None:
            # open_clip's transform for the SigLIP2 checkpoints (:69) restated: Resize(BICUBIC, squash) + ToTensor + Normalize(.5)
            from .imaging import siglip_preprocess
            size = encoder.image_size if encoder is not None else 384
            preprocess = lambda im: siglip_preprocess(im, size)
        self.encoder, self.preprocess, self.tokenizer = encoder, preprocess, tokenizer
        self._dev = dev

    # ---- feature-level API (what the parity tests drive: tokenisation / resampling are inputs)
    def extract_shared_features(self, img_tensor, text_tokens):
        return self.encoder.extract_features(img_tensor.to(self._dev), text_tokens.to(self._dev))

    def _pad_histories(self, all_action_histories):
        """efficient_ensemble_merged.py:378-390 (front pad with -5 to length 10) + the padding mask of :229."""
        max_len = 10
        out = []
        for ah in all_action_histories:
            ah = np.array(ah)
            if len(ah) < max_len:
                ah = np.vstack([np.ones((max_len - len(ah), ah.shape[1])) * -5, ah])
            out.append(ah)
        hb = torch.tensor(np.array(out), dtype=torch.float32)
        return hb

    def get_embeddings_from_model_batch(self, model_idx, patch_features, text_features, action_histories_batch):
        m = self.trainable_models[model_idx]
        hb = action_histories_batch.float().to(self._dev).contiguous()
        pad = (action_histories_batch[:, :, 0].cpu() == m.pad_value).to(torch.uint8).to(self._dev).contiguous()
        it = m.image_text(patch_features[0].contiguous(), text_features[0].contiguous())
        act = m.trajectory(hb, pad)
        return it.expand(hb.shape[0], -1), act

    def shared_embeddings_graph(self, img_tensor: torch.Tensor, text_tokens: torch.Tensor) -> torch.Tensor:
        """image_text_embeddings(*extract_shared_features(img, text)) for ONE (image, instruction) pair, replayed as one hipGraph from the
        third call on, with the SigLIP2 image tower and text tower as parallel branches (ops.PooledGraph): ~600 launches become one host
        call, and the two towers -- 576 and 64 rows, a fraction of the chip each -- overlap. Same kernels, same results as the eager pair of
        calls. Returns the per-member embeddings [M, 512] (a static tensor, overwritten by the next call)."""
        key = (tuple(img_tensor.shape), tuple(text_tokens.shape))
        st = self._shared_graphs.get(key) if hasattr(self, "_shared_graphs") else None
        if st is None:
            if not hasattr(self, "_shared_graphs"):
                self._shared_graphs = {}
            img = torch.empty_like(img_tensor, device=self._dev)
            txt = torch.empty_like(text_tokens, device=self._dev)
            side = torch.cuda.Stream(device=self._dev)

            def fn():
                pf, tf = self.encoder.extract_features(img, txt, text_stream=side)
                return self.image_text_embeddings(pf, tf)

            st = dict(img=img, txt=txt, g=ops.PooledGraph(fn, self._dev))
            self._shared_graphs[key] = st
        st["img"].copy_(img_tensor)
        st["txt"].copy_(text_tokens)
        return st["g"]()

    def image_text_embeddings(self, patch_features, text_features) -> torch.Tensor:
        """Per-member image-text embeddings [M, 512] of ONE (image, text) pair. Independent of the candidates, so a caller
        may run it (and extract_shared_features) on a side stream while the policy is still sampling."""
        pf = patch_features[0].to(self._dev).contiguous()
        tf = text_features[0].to(self._dev).contiguous()
        its = torch.empty(self.num_models, 512, dtype=torch.float32, device=self._dev)
        if self._it_stack is not None and self._it_stack.ok and os.environ.get("COVER_MEMBER_BATCH", "1") != "0":
            return self._it_stack(pf, tf, its)          # all members per launch
        for i, m in enumerate(self.trainable_models):
            its[i] = m.image_text(pf, tf)[0]
        return its

    def score_histories(self, its: torch.Tensor, all_action_histories, group_size=1, pad: Optional[torch.Tensor] = None):
        """Trajectory encoder per candidate + fusion + scoring + grouped arg-max against precomputed image-text embeddings.
        all_action_histories: list of [h<=10, 7] host arrays (reference format), OR an already padded DEVICE tensor
        fp32 [N,10,7] together with its padding mask `pad` uint8 [N,10] (ops.tokens_to_histories): no host round trip."""
        if torch.is_tensor(all_action_histories) and all_action_histories.is_cuda:
            hb = all_action_histories.contiguous()
        else:
            hb = self._pad_histories(all_action_histories)
            pad = (hb[:, :, 0] == self.trainable_models[0].pad_value).to(torch.uint8).to(self._dev).contiguous()
            hb = hb.to(self._dev).contiguous()
        N = hb.shape[0]
        acts = torch.empty(self.num_models, N, 512, dtype=torch.float32, device=self._dev)

        # (members on separate HIP streams were tried: the tail shrinks 1.95 -> 1.47 ms but every decode pass of the policy
        # slows by ~0.17 ms with the extra queues alive -- a net loss, so the members stay sequential)
        if self._traj_stack is not None and self._traj_stack.ok and os.environ.get("COVER_MEMBER_BATCH", "1") != "0":
            self._traj_stack(hb, pad, acts)           # all members per launch
        else:
            for i, m in enumerate(self.trainable_models):
                acts[i] = m.trajectory(hb, pad)
        scores, result, best, fit, fact = ops.score_select(its, acts, group_size)
        return {"scores": scores, "result": result, "best": best, "its": its, "acts": acts, "fused_it": fit, "fused_act": fact}

    def score_features(self, patch_features, text_features, all_action_histories, group_size=1):
        """Scores every candidate history against ONE (image, text) pair given its features. Returns a dict with the
        device tensors (scores [N], result int32[4], best f32[2]) plus per-member embeddings."""
        return self.score_histories(self.image_text_embeddings(patch_features, text_features), all_action_histories, group_size)

    def fuse_embeddings(self, image, instruction, action_histories):
        """efficient_ensemble_merged.py:249-293 -> (fused_image_text [N,512] (the one pair's embedding, a row per history),
        fused_action [N,512]). The reference stacks the histories with np.array (equal lengths, :267); shorter histories are
        front-padded here, which gives the same valid rows."""
        pf, tf = self._encode_pair(image, instruction)
        r = self.score_features(pf, tf, action_histories, 1)
        return r["fused_it"].view(1, -1).expand(r["fused_act"].shape[0], -1), r["fused_act"]

    def predict(self, image, instruction, possible_action_histories):
        """efficient_ensemble_merged.py:295-307."""
        pf, tf = self._encode_pair(image, instruction)
        scores = self.score_features(pf, tf, possible_action_histories, 1)["scores"].cpu().numpy()
        idx = scores.argmax()
        return possible_action_histories[idx], {str(i): float(scores[i]) for i in range(len(scores))}

    def _encode_pair(self, image, instruction):
        if self.encoder is None:
            raise L.CoverError("EfficientEnsembleMerged needs the SigLIP2 encoder for image/text inputs")
        if isinstance(image, np.ndarray):                      # efficient_ensemble_merged.py:252-253, 334-337
            from PIL import Image
            image = Image.fromarray(image.astype("uint8"))
        img_tensor = self.preprocess(image).unsqueeze(0)
        if isinstance(instruction, str):
            if self.tokenizer is None:
                raise L.CoverError("EfficientEnsembleMerged needs a tokenizer for string instructions (open_clip's is un-vendored)")
            toks = self.tokenizer([instruction], context_length=self.encoder.context_length)
        else:
            toks = instruction if instruction.ndim > 1 else instruction.unsqueeze(0)
        return self.extract_shared_features(img_tensor, toks)

    def compute_max_similarity_scores_batch(self, images, instructions, all_action_histories,
                                            cfg_repeat_language_instructions=1):
        """efficient_ensemble_merged.py:309-454. The reference scores candidates against the FIRST (image, text) pair only
        (reference_scores = similarity_matrix[0], :421-425), on both of its encode paths, so one pair is encoded here."""
        group_size = cfg_repeat_language_instructions
        pf, tf = self._encode_pair(images[0], instructions[0])
        r = self.score_features(pf, tf, all_action_histories, group_size)
        result = r["result"].cpu()
        best = r["best"].cpu()
        gidx, gbest = int(result[0]), int(result[1])
        all_same = len(set(instructions)) == 1 if isinstance(instructions[0], str) else False
        if all_same and len(images) > 1:
            max_instruction = instructions[0]
        else:
            max_instruction = instructions[min(gbest * group_size, len(instructions) - 1)]
        return float(best[0]), max_instruction, all_action_histories[gidx], torch.tensor(gidx, dtype=torch.int64)
