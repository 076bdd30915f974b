import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from cover_ref import blocks as Bk, openvla as OR
from cover_vla_amd.openvla import OpenVLA
from tests.test_fp8_gpu import _dequant_sd
from tests.test_openvla_gpu import _case
dev = torch.device("cuda:0")
n_samples = 24
c, sd, frame, toks, lens, _ = _case(seed=9, n_samples=n_samples)
P = toks.shape[0]; N = P * n_samples
u = torch.rand(N, 7, generator=torch.Generator().manual_seed(9))
model = OpenVLA(sd, c, device="cuda:0", max_prompts=4, max_candidates=N, max_text=toks.shape[1], weight_dtype="fp8")
o1, o0 = {}, {}
with torch.no_grad():
    ref = OR.sample(c, Bk.to_bf16(_dequant_sd(sd)), frame, toks, lens, n_samples, u, 0.9, trace=o1, act_fp8_decode=True)
    OR.sample(c, Bk.to_bf16(_dequant_sd(sd)), frame, toks, lens, n_samples, u, 0.9, trace=o0, act_fp8_decode=False)  # free-running differs after a flip; compare step 1 only
tr = {}
model.sample(frame.to(dev), toks.to(dev), lens.to(dev), n_samples, u.to(dev), 0.9, trace=tr, force_tokens=ref.to(dev))
gl = torch.stack([l.cpu() for l in tr["logits"]], 1)
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
for n in (0, 1, 24, 25, 48, 71):
    print(n, [round(rel(gl[n, i], o1["logits"][n, i]), 4) for i in range(7)], "oracle q vs no-q @1:", round(rel(o1["logits"][n, 1], o0["logits"][n, 1]), 4),
          "hip vs no-q @1:", round(rel(gl[n, 1], o0["logits"][n, 1]), 4))
