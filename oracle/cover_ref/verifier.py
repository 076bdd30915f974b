"""CPU restatement of the CoVer verifier heads, fusion, scoring and selection (fp32, as the reference).

Follows
  bridge_verifier/ensemble_eval/model.py:7-38      CrossAttentionBlock (nn.MultiheadAttention kdim=vdim=1024, timm Mlp)
  bridge_verifier/ensemble_eval/model.py:50-73     TextAwareVisualExtraction
  bridge_verifier/ensemble_eval/model.py:76-112    AttentionPooling
  bridge_verifier/ensemble_eval/efficient_ensemble_merged.py:194-247   get_embeddings_from_model_batch
  bridge_verifier/ensemble_eval/efficient_ensemble_merged.py:378-454   padding, fusion, scoring, grouped arg-max
Weights are the reference's own state-dict entries (see cover_vla_amd/synth.py for the key layout). Written as
explicit matmuls rather than nn.Module calls so that it is a restatement, not a re-import.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _ln(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _mha(q_in, kv_in, wq, wk, wv, b_in, wo, bo, heads, key_pad=None):
    """torch.nn.MultiheadAttention forward (batch_first semantics), eval mode. q_in [B,Tq,E], kv_in [B,Tk,Ekv]."""
    E = wq.shape[0]
    bq, bk, bv = b_in[:E], b_in[E:2 * E], b_in[2 * E:]
    q = q_in @ wq.T + bq
    k = kv_in @ wk.T + bk
    v = kv_in @ wv.T + bv
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    dh = E // heads
    q = q.view(B, Tq, heads, dh).transpose(1, 2) * (dh ** -0.5)
    k = k.view(B, Tk, heads, dh).transpose(1, 2)
    v = v.view(B, Tk, heads, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if key_pad is not None:
        s = s.masked_fill(key_pad[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, Tq, E)
    return o @ wo.T + bo


def text_aware_visual_extraction(sd, pf, tf):
    """model.py:58-73. pf [B,P,D] unit rows, tf [B,T,D] unit rows -> [B,T,D]."""
    sim = torch.einsum("bij,bkj->bik", tf, pf)
    attn = torch.softmax(sim / sd["temperature"].clamp(0, 100), dim=-1)
    return torch.einsum("bik,bkj->bij", attn, pf + sd["pos_emb"])


def cross_attention_block(sd, p, q, kv, heads=8):
    """model.py:25-38: residuals add to the NORMALISED stream."""
    q = _ln(q, sd[p + "q_layer_norm.weight"], sd[p + "q_layer_norm.bias"])
    a = _mha(q, kv, sd[p + "attention.q_proj_weight"], sd[p + "attention.k_proj_weight"], sd[p + "attention.v_proj_weight"],
             sd[p + "attention.in_proj_bias"], sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"], heads)
    q = q + a
    q = _ln(q, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"])
    h = F.gelu(q @ sd[p + "mlp.fc1.weight"].T + sd[p + "mlp.fc1.bias"])  # timm Mlp: exact GELU
    return q + (h @ sd[p + "mlp.fc2.weight"].T + sd[p + "mlp.fc2.bias"])


def attention_pooling(sd, x, heads=8):
    """model.py:97-112 (num_readouts = 1)."""
    B = x.shape[0]
    q = sd["query"].expand(B, -1, -1)
    n_layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(n_layers):
        q = cross_attention_block(sd, f"blocks.{i}.", q, x, heads)
    q = _ln(q, sd["layer_norm.weight"], sd["layer_norm.bias"])
    return q.reshape(B, -1)


def transformer_encoder_layer(sd, p, x, key_pad, heads=8):
    """nn.TransformerEncoderLayer defaults: post-LN, ReLU, eps 1e-5 (efficient_ensemble_merged.py:140-147). x [B,T,E]."""
    E = x.shape[-1]
    w_in, b_in = sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]
    a = _mha(x, x, w_in[:E], w_in[E:2 * E], w_in[2 * E:], b_in, sd[p + "self_attn.out_proj.weight"],
             sd[p + "self_attn.out_proj.bias"], heads, key_pad)
    x = _ln(x + a, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    h = torch.relu(x @ sd[p + "linear1.weight"].T + sd[p + "linear1.bias"])
    h = h @ sd[p + "linear2.weight"].T + sd[p + "linear2.bias"]
    return _ln(x + h, sd[p + "norm2.weight"], sd[p + "norm2.bias"])


def image_text_embedding(member, pf, tf):
    """efficient_ensemble_merged.py:216-223 for ONE (image, text) pair: [1,512] unit vector."""
    taf = text_aware_visual_extraction(member["text_aware_visual_extraction"], pf, tf)
    vision_token = attention_pooling(member["vision_poolings"], taf)
    text_token = attention_pooling(member["text_pooling"], tf)
    comb = torch.cat([text_token, vision_token], dim=-1)
    comb = comb @ member["input_projection"]["weight"].T + member["input_projection"]["bias"]
    return comb / comb.norm(dim=-1, keepdim=True)


def trajectory_embedding(member, histories):
    """efficient_ensemble_merged.py:226-245. histories fp32 [N,10,7] (front padded with -5) -> [N,512] unit rows."""
    a = histories.float()
    if member.get("trajectory_encoder") is None:
        # MLP variant (:148-184, 241-243): flat actions -> Linear -> LayerNorm -> ReLU -> (Dropout) -> Linear; no padding mask
        m = member["complex_action_encoder"]
        h = a.reshape(a.shape[0], -1) @ m["0.weight"].T + m["0.bias"]
        h = torch.relu(_ln(h, m["1.weight"], m["1.bias"]))
        traj = h @ m["4.weight"].T + m["4.bias"]
        return traj / traj.norm(dim=-1, keepdim=True)
    pad = a[:, :, 0] == member["action_padding_value"]
    x = a @ member["single_step_action_encoder"]["weight"].T + member["single_step_action_encoder"]["bias"]
    sd = member["trajectory_encoder"]
    n_layers = 1 + max(int(k.split(".")[1]) for k in sd)
    for i in range(n_layers):
        x = transformer_encoder_layer(sd, f"layers.{i}.", x, pad)
    keep = (~pad).unsqueeze(-1).float()
    summed = (x * keep).sum(dim=1)
    cnt = keep.sum(dim=1).clamp(min=1e-9)
    traj = summed / cnt
    return traj / traj.norm(dim=-1, keepdim=True)


def pad_histories(histories, max_len=10, pad_value=-5):
    """efficient_ensemble_merged.py:378-390: front-pad with -5 to length 10, float64 numpy -> fp32 tensor."""
    out = []
    for ah in histories:
        ah = np.array(ah)
        if len(ah) < max_len:
            ah = np.vstack([np.ones((max_len - len(ah), ah.shape[1])) * pad_value, ah])
        out.append(ah)
    return torch.tensor(np.array(out), dtype=torch.float32)


def fuse_and_score(members, pf, tf, hist_batch):
    """efficient_ensemble_merged.py:396-414 (encode-once path): returns (scores [N], fused_it [1,512], fused_act [N,512],
    per-member it [M,1,512], per-member act [M,N,512])."""
    its = torch.stack([image_text_embedding(m, pf, tf) for m in members])
    acts = torch.stack([trajectory_embedding(m, hist_batch) for m in members])
    f_it = its.mean(dim=0)
    f_act = acts.mean(dim=0)
    f_it = f_it / f_it.norm(dim=-1, keepdim=True)
    f_act = f_act / f_act.norm(dim=-1, keepdim=True)
    sim = f_it @ f_act.T
    return sim[0], f_it, f_act, its, acts


def select(scores, group_size):
    """efficient_ensemble_merged.py:417-448: arg-max of group means, then arg-max inside the group (first max wins)."""
    n_groups = scores.numel() // group_size
    rs = scores.view(n_groups, group_size)
    best_group_score, best_group = rs.mean(dim=1).max(dim=0)
    max_score, best_in = rs[best_group].max(dim=0)
    return int(best_group * group_size + best_in), int(best_group), int(best_in), float(max_score), float(best_group_score)


def compute_max_similarity_scores(members, pf, tf, histories, group_size=1):
    """Feature-level equivalent of compute_max_similarity_scores_batch (:309-454): histories = list of [h<=10, 7]."""
    hb = pad_histories(histories)
    scores, f_it, f_act, its, acts = fuse_and_score(members, pf, tf, hb)
    gidx, g, i, mx, gm = select(scores, group_size)
    return {"scores": scores, "global_idx": gidx, "group": g, "in_group": i, "max_score": mx, "group_mean": gm,
            "fused_it": f_it, "fused_act": f_act, "its": its, "acts": acts}


def contrastive_forward(member, logit_scale, pf, tf, histories):
    """finetune_trajectory_bridge_ddp.py:357-421 in eval mode for a batch of B DISTINCT triples: pf [B,P,D], tf [B,T,D],
    histories [B,10,7] -> (image_logits, action_logits) [B,B]."""
    it = torch.cat([image_text_embedding(member, pf[b:b + 1], tf[b:b + 1]) for b in range(pf.shape[0])], dim=0)
    act = trajectory_embedding(member, histories)
    scale = torch.tensor(float(logit_scale), dtype=torch.float32).exp()
    return scale * (it @ act.T), scale * (act @ it.T)


def contrastive_metrics(image_logits, action_logits, k_values=(1, 5)):
    """Loss of the training / validation loops (:895-899) and calculate_accuracy_metrics (:446-469)."""
    B = image_logits.shape[0]
    labels = torch.arange(B)

    def ce(x):
        return float((torch.logsumexp(x, dim=1) - x[labels, labels]).mean())

    out = {"image_loss": ce(image_logits), "action_loss": ce(action_logits)}
    out["loss"] = 0.5 * (out["image_loss"] + out["action_loss"])
    for name, x in (("img2act", image_logits), ("act2img", action_logits)):
        for k in k_values:
            if k <= B:
                top = torch.topk(x, k, dim=1).indices
                out[f"{name}_top{k}_acc"] = float((top == labels.view(-1, 1)).any(dim=1).float().mean())
    return out
