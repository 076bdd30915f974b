"""pi0-FAST token path at full size on one MI355X (secondary measurement; bench.py is the contractual line): PaliGemma-3B geometry
(SigLIP-So400m 224^2 + Gemma-2B, vocabulary 257152), B = 40 candidates = 8 rephrased prompts x 5 samples (greedy decoding is a
function of the prompt: 8 distinct generations), prompt 48 tokens, NEW action tokens per candidate (default 32), synthetic weights."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cover_vla_amd import synth
from cover_vla_amd.pi0fast import PI0FASTTokens

dev = torch.device("cuda:0")
NEW = int(os.environ.get("NEW", "32"))
c = dict(synth.PI0_FULL)
g = synth._G(1234, False, 0.02, dev, torch.bfloat16)
sd = {}
n_patches = (c["image"] // c["patch"]) ** 2
for k, v in synth.vit_state(g, dim=c["vit_dim"], layers=c["vit_layers"], heads=c["vit_heads"], mlp=c["vit_mlp"], patch=c["patch"], n_pos=n_patches).items():
    sd["vision." + k] = v
sd["projector.weight"] = g.w(c["lm_dim"], c["vit_dim"]); sd["projector.bias"] = g.b(c["lm_dim"])
for k, v in synth.decoder_state(g, dim=c["lm_dim"], layers=c["layers"], Hq=c["Hq"], Hkv=c["Hkv"], D=c["D"], mlp=c["lm_mlp"], rms_base=0.0, vocab=c["vocab"]).items():
    sd["lm." + k] = v
B, P, L = 40, 8, 48
model = PI0FASTTokens(sd, c, device="cuda:0", max_batch=P, max_prompt=L, max_new_tokens=max(NEW, 8))
del sd; torch.cuda.empty_cache()
gen = torch.Generator().manual_seed(0)
img = (torch.rand(1, 3, 224, 224, generator=gen) * 2 - 1).repeat(B, 1, 1, 1).to(dev)
toks = torch.zeros(B, L, dtype=torch.long); pad = torch.zeros(B, L, dtype=torch.long)
for p in range(P):
    n = 30 + p
    row = torch.randint(2, 257000, (n,), generator=gen)
    for s in range(B // P):
        toks[p * (B // P) + s, :n] = row; pad[p * (B // P) + s, :n] = 1
toks, pad = toks.to(dev), pad.to(dev)
ones = [torch.ones(B, dtype=torch.bool, device=dev)]
def step():
    return model.generate_tokens([img], ones, toks, pad, NEW, eos_token_id=-1)      # no early stop: NEW tokens for every row
for _ in range(2): out = step()
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n): out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
assert out.shape == (B, NEW)
wbytes = 2.0 * (c["layers"] * (c["lm_dim"] * (c["Hq"] + 2 * c["Hkv"]) * c["D"] + c["Hq"] * c["D"] * c["lm_dim"] + 3 * c["lm_dim"] * c["lm_mlp"]) + c["vocab"] * c["lm_dim"])
print(json.dumps({"profile": "pi0-FAST tokens", "B": B, "distinct_prompts": P, "new_tokens": NEW, "ms_per_decision": round(dt * 1e3, 2),
                  "candidates_per_s": round(B / dt, 1), "decode_weight_GB_per_step": round(wbytes / 1e9, 2),
                  "hbm_floor_ms_decode": round((NEW - 1) * wbytes / 8e12 * 1e3, 2)}))
