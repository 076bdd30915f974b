"""pi0 profile (P1, what the reference actually runs) at full size on one MI355X: PI0FlowMatching.sample_actions for
B = 40 candidates = 8 prompts x 5 samples, one 224x224 camera, L = 72, chunk 4, 10 Euler steps, synthetic weights;
then the verifier on the 40 chunks. Prints decisions/s and candidates/s. (Secondary measurement; bench.py is the
contractual line on the OpenVLA-7B shapes.)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cover_vla_amd import synth, ops
from cover_vla_amd.pi0 import PI0FlowMatching
from cover_vla_amd.verifier import EfficientEnsembleMerged, SigLIP2Encoder

dev = torch.device("cuda:0")
c = dict(synth.PI0_FULL)
t0 = time.time()
sd = synth_sd = None
g = synth._G(1234, False, 0.02, dev, torch.bfloat16)
# build the neutral state dict directly on the device (bf16 weights)
sd = {}
n_patches = (c["image"] // c["patch"]) ** 2
for k, v in synth.vit_state(g, dim=c["vit_dim"], layers=c["vit_layers"], heads=c["vit_heads"], mlp=c["vit_mlp"], patch=c["patch"], n_pos=n_patches).items():
    sd["vision." + k] = v
sd["projector.weight"] = g.w(c["lm_dim"], c["vit_dim"]); sd["projector.bias"] = g.b(c["lm_dim"])
for k, v in synth.decoder_state(g, dim=c["lm_dim"], layers=c["layers"], Hq=c["Hq"], Hkv=c["Hkv"], D=c["D"], mlp=c["lm_mlp"], rms_base=0.0, vocab=c["vocab"]).items():
    sd["lm." + k] = v
for k, v in synth.decoder_state(g, dim=c["ex_dim"], layers=c["layers"], Hq=c["Hq"], Hkv=c["Hkv"], D=c["D"], mlp=c["ex_mlp"], rms_base=0.0).items():
    sd["expert." + k] = v
pw = c["ex_dim"]
for n, (o, i) in {"state_proj": (pw, 32), "action_in_proj": (pw, 32), "action_out_proj": (32, pw), "action_time_mlp_in": (pw, 2 * pw), "action_time_mlp_out": (pw, pw)}.items():
    sd[n + ".weight"] = g.w(o, i).float(); sd[n + ".bias"] = g.b(o)
B, P, L = 40, 8, 72
model = PI0FlowMatching(sd, c, device="cuda:0", max_batch=B, max_prompts=P, max_lang=L)
del sd; torch.cuda.empty_cache()
ssd = synth.siglip2_state(dict(synth.SIGLIP2_L), seed=4321, nontrivial=False, device=dev, wdtype=torch.bfloat16)
enc = SigLIP2Encoder(ssd, device="cuda:0"); del ssd
ver = EfficientEnsembleMerged(synth.verifier_checkpoint(3, seed=1234), device="cuda:0", encoder=enc)
print(f"build {time.time()-t0:.1f}s", flush=True)
gen = torch.Generator().manual_seed(0)
img = (torch.rand(1, 3, 224, 224, generator=gen) * 2 - 1).repeat(B, 1, 1, 1).to(dev)
toks = torch.zeros(B, L, dtype=torch.long); masks = torch.zeros(B, L, dtype=torch.bool)
for p in range(P):
    n = 16 + p
    row = torch.randint(1, 257000, (n,), generator=gen)
    for s in range(B // P):
        toks[p * (B // P) + s, :n] = row; masks[p * (B // P) + s, :n] = True
state = torch.zeros(B, 32); state[:, :7] = torch.rand(1, 7, generator=gen) * 2 - 1
noise = torch.randn(B, 4, 32, generator=gen)
img384 = torch.randn(1, 3, 384, 384, generator=gen).to(dev); text = torch.randint(0, 32000, (1, 64), generator=gen).to(dev)
past = (torch.randn(6, 7, generator=gen) * 0.02).double().numpy()
toks, masks, state, noise = toks.to(dev), masks.to(dev), state.to(dev), noise.to(dev)
side = torch.cuda.Stream()
from concurrent.futures import ThreadPoolExecutor
from cover_vla_amd import host
pool = ThreadPoolExecutor(1)
st = host.bridge_statistics()["action"]
lo_hi = torch.tensor(list(st["p01"][:6]) + list(st["p99"][:6]), dtype=torch.float32, device=dev)
past_d = torch.tensor(past, dtype=torch.float32, device=dev)
history = [past[i] for i in range(6)]
MODE = os.environ.get("PI0_MODE", "device")     # "host": chunks -> host numpy post-processing -> histories (reference flow)
all_valid = torch.ones(B, dtype=torch.bool, device=dev)

def side_work(ev):
    torch.cuda.set_device(0)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        pf, tf = ver.extract_shared_features(img384, text)
        return ver.image_text_embeddings(pf, tf)

def decision():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    if MODE == "host":
        its = side_work(ev)
    else:
        fut = pool.submit(side_work, ev)            # queued from a second host thread while this one queues the policy
    x = model.sample_actions([img], [all_valid], toks, masks, state, noise=noise)
    if MODE == "host":
        xa = x[:, :4, :7].cpu().numpy()
        hists = host.process_inputs([xa[:, t] for t in range(4)], True, history, 4)
        pad = None
    else:
        hists, pad = ops.actions_to_histories(x, 4, past_d, lo_hi)
        its = fut.result()
    main.wait_stream(side)
    r = ver.score_histories(its, hists, B // P, pad=pad)
    return int(r["result"][0]), x

idx, x = decision()
assert torch.isfinite(x).all(), "non-finite actions"
torch.cuda.synchronize()
t = time.perf_counter(); K = 5
for _ in range(K): decision()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
print(json.dumps({"profile": "pi0-cover (P1)", "mode": MODE, "B": B, "prompts": P, "L": L, "ms_per_decision": round(dt * 1e3, 2),
                  "candidates_per_s": round(B / dt, 1), "selected": idx, "action_absmax": float(x.abs().max())}))
