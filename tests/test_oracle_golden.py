"""The oracle (oracle/cover_ref) against golden vectors produced by the REFERENCE's own modules
(oracle/gen_golden.py, run in the build container). CPU only. Tolerances: fp32 heads/scores atol 1e-5,
indices exact."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")

from cover_ref import verifier as V  # noqa: E402
from cover_vla_amd import synth  # noqa: E402


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "verifier_m*.npz"))))
def test_verifier_oracle_matches_reference(path):
    z = np.load(path)
    members, N, group = int(z["members"]), int(z["N"]), int(z["group"])
    ckpt = synth.verifier_checkpoint(members, seed=int(z["ckpt_seed"]))
    pf, tf, hists = synth.verifier_inputs(N, seed=int(z["input_seed"]))
    assert [len(h) for h in hists] == z["hist_lens"].tolist()
    with torch.no_grad():
        r = V.compute_max_similarity_scores(ckpt["ensemble_components"], pf, tf, hists, group)
    assert np.allclose(r["its"].numpy(), z["its"], atol=1e-5)
    assert np.allclose(r["acts"].numpy(), z["acts"], atol=1e-5)
    assert np.allclose(r["scores"].numpy(), z["scores"], atol=1e-5)
    assert r["global_idx"] == int(z["global_idx"])
    assert abs(r["max_score"] - float(z["max_score"])) < 1e-5


def test_verifier_ties_first_index_wins():
    z = np.load(os.path.join(GOLD, "verifier_ties.npz"))
    ckpt = synth.verifier_checkpoint(2, seed=99)
    pf, tf, _ = synth.verifier_inputs(12, seed=99)
    with torch.no_grad():
        r = V.compute_max_similarity_scores(ckpt["ensemble_components"], pf, tf, [z["hist"]] * 12, 3)
    assert r["global_idx"] == int(z["global_idx"]) == 0
    assert abs(r["max_score"] - float(z["max_score"])) < 1e-5
