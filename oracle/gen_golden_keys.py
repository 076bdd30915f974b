"""Key-name fixture for the on-disk checkpoint format: the parameter names the reference's converter ASSIGNS
(lerobot_custom/lerobot/common/policies/pi0/conversion_scripts/convert_pi0_to_hf_lerobot.py:67-245, prefixes :384-386), extracted
from its source text with a regular expression and expanded over the layer loops. Only the resulting NAME LIST (data) is
committed -> tests/golden/pi0_checkpoint_keys.json. Also the per-episode pickle schema the driver writes
(CoVer_VLA/inference/experiments/robot/simpler/run_simpler_eval_with_openpi.py:238-247) as a list of field names.
Usage: python oracle/gen_golden_keys.py"""
import json
import os
import re

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = open(os.path.join(REF, "lerobot_custom/lerobot/common/policies/pi0/conversion_scripts/convert_pi0_to_hf_lerobot.py")).read()
    keys = set()
    for m in re.finditer(r"""state_dict\[f?["']([A-Za-z_.{}0-9]+)["']\]\s*=""", src):
        k = m.group(1)
        if "/" in k:
            continue
        if "{i}" in k:
            n = 27 if "vision_tower" in k else 18       # SigLIP-So400m blocks / Gemma-2B and expert layers
            keys.update(k.replace("{i}", str(i)) for i in range(n))
        else:
            keys.add(k)
    full = sorted("model.paligemma_with_expert." + k for k in keys)
    # pi0's own projections (modeling_pi0.py:486-494), prefixed "model." (:386)
    proj = [f"model.{n}.{wb}" for n in ("state_proj", "action_in_proj", "action_out_proj", "action_time_mlp_in", "action_time_mlp_out")
            for wb in ("weight", "bias")]
    drv = open(os.path.join(REF, "CoVer_VLA/inference/experiments/robot/simpler/run_simpler_eval_with_openpi.py")).read()
    blk = drv[drv.index("episode_data = {"):]
    blk = blk[: blk.index("}")]
    fields = re.findall(r"""["']([a-z_]+)["']\s*:""", blk)
    out = {"source": "convert_pi0_to_hf_lerobot.py:67-245,384-386 (assigned keys, layer loops expanded: 27 vision / 18 decoder layers)",
           "keys": full + sorted(proj), "episode_fields_source": "run_simpler_eval_with_openpi.py:238-247", "episode_fields": fields}
    path = os.path.join(ROOT, "tests", "golden", "pi0_checkpoint_keys.json")
    json.dump(out, open(path, "w"), indent=0)
    print(f"wrote {path}: {len(out['keys'])} keys, episode fields {fields}")


if __name__ == "__main__":
    main()
