import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cover_vla_amd import imaging as IM
from tests.test_imaging_cpu import _images
raw = _images()["noise"]
pipe = IM.DeviceImagePipeline(device="cuda:0")
x = pipe.policy_image(raw, (224, 224)).cpu()
h = IM.simpler_preprocess_image(raw, (224, 224))
print(x.shape, h.shape, x.dtype, h.dtype, (x - h).abs().max().item(), (x != h).sum().item())
i = (x != h).nonzero()[:5]
for idx in i:
    idx = tuple(idx.tolist())
    print(idx, x[idx].item(), h[idx].item(), np.float32(x[idx].item()).view(np.uint32), np.float32(h[idx].item()).view(np.uint32))
