"""BASELINE.json configs[3]: Prismatic dual encoder, 2-camera 224^2 observation, N = 64 (8 prompts x 8 samples), verifier
ensemble = 2 -- full-size random weights, one GPU. Plumbing + timing of a non-headline configuration."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cover_vla_amd import synth, ops
from cover_vla_amd.openvla import OpenVLA
from cover_vla_amd.verifier import EfficientEnsembleMerged, SigLIP2Encoder
dev = torch.device("cuda:0")
c = dict(synth.OPENVLA_7B)
P, S, LT = 8, 8, 24
sd = synth.openvla_state(c, seed=1234, nontrivial=False, device=dev, wdtype=torch.bfloat16)
pol = OpenVLA(sd, c, device="cuda:0", max_prompts=P, max_candidates=P * S, max_text=LT, n_cams=2); del sd
ssd = synth.siglip2_state(dict(synth.SIGLIP2_L), seed=4321, nontrivial=False, device=dev, wdtype=torch.bfloat16)
enc = SigLIP2Encoder(ssd, device="cuda:0"); del ssd
ver = EfficientEnsembleMerged(synth.verifier_checkpoint(2, seed=1234), device="cuda:0", encoder=enc)
g = torch.Generator().manual_seed(0)
frames = torch.randint(0, 256, (2, 224, 224, 3), generator=g, dtype=torch.uint8).to(dev)
lens = torch.tensor([16 + i for i in range(P)], dtype=torch.int32)
toks = torch.zeros(P, LT, dtype=torch.long)
for p in range(P): toks[p, :lens[p]] = torch.randint(3, 31000, (int(lens[p]),), generator=g)
u = torch.rand(P * S, 7, generator=g).to(dev)
img384 = torch.randn(1, 3, 384, 384, generator=g).to(dev); text = torch.randint(0, 32000, (1, 64), generator=g).to(dev)
bins = np.linspace(-1, 1, 256); centers = torch.tensor((bins[:-1] + bins[1:]) / 2, dtype=torch.float32, device=dev)
past = (torch.randn(6, 7, generator=g) * 0.02).to(dev)
def decision():
    pf, tf = ver.extract_shared_features(img384, text)
    its = ver.image_text_embeddings(pf, tf)
    tok, _ = pol.sample(frames, toks.to(dev), lens.to(dev), S, u, 1.0)
    hb, pad = ops.tokens_to_histories(tok, c["tok_vocab"], centers, past)
    r = ver.score_histories(its, hb, S, pad=pad)
    return int(r["result"][0]), tok
idx, tok = decision(); idx2, tok2 = decision()
assert idx == idx2 and torch.equal(tok, tok2)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): decision()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(json.dumps({"config": "2 cameras, N=64 (8x8), ensemble=2", "ms_per_decision": round(dt * 1e3, 2), "candidates_per_s": round(P * S / dt, 1), "selected": idx}))
