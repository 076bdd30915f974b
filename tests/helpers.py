"""Shared test helpers (config builders / seeded cases used by both the CPU and the GPU test files)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from cover_vla_amd import synth  # noqa: E402


def _pi0_cfg(tiny):
    from cover_ref import blocks as Bk, pi0 as P
    vit = Bk.VitCfg(tiny["vit_dim"], tiny["vit_layers"], tiny["vit_heads"], tiny["vit_mlp"], tiny["patch"], "gelu_tanh", 1e-6)
    lm = Bk.DecoderCfg(tiny["lm_dim"], tiny["layers"], tiny["Hq"], tiny["Hkv"], tiny["D"], tiny["lm_mlp"], "gelu_tanh", "gemma", 1e-6, "pi0")
    ex = Bk.DecoderCfg(tiny["ex_dim"], tiny["layers"], tiny["Hq"], tiny["Hkv"], tiny["D"], tiny["ex_mlp"], "gelu_tanh", "gemma", 1e-6, "pi0")
    return P.Pi0Cfg(vit, lm, ex, proj_width=tiny["ex_dim"], chunk_size=tiny["chunk"], n_img_tokens=(tiny["image"] // tiny["patch"]) ** 2)


def pi0_case(path):
    from gen_golden_pi0 import pi0_inputs
    z = np.load(path)
    tiny = {k[5:]: int(z[k]) for k in z.files if k.startswith("tiny_")}
    B, L, seed = int(z["B"]), int(z["L"]), int(z["seed"])
    sd = synth.pi0_state(tiny, seed=seed)
    return z, tiny, sd, pi0_inputs(tiny, B, L, seed)
