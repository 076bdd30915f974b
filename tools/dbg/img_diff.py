import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cover_vla_amd import imaging as IM
from tests.test_imaging_cpu import _images
dev = torch.device("cuda:0")
pipe = IM.DeviceImagePipeline(device="cuda:0")
for name, raw in list(_images().items()) + [("up", np.random.default_rng(3).integers(0, 256, size=(224, 224, 3), dtype=np.uint8))]:
    d = pipe.raw_to_jpg(torch.from_numpy(raw).to(dev)).cpu().numpy().astype(int)
    h = IM.process_raw_image_to_jpg(raw).astype(int)
    bad = np.argwhere(d != h)
    print(name, len(bad), bad[:5].tolist(), [(int(d[tuple(b)]), int(h[tuple(b)])) for b in bad[:5]])
    # stage by stage
    H, W, _ = raw.shape
    bh, wh, sh = IM.tf_spans(H, 256)
    rows_h = IM._gather_f32(raw, bh, wh, 0)
    tb = pipe._tables(H, W)
    rows_d = torch.empty(256, W, 3, dtype=torch.float32, device=dev)
    from cover_vla_amd import _lib as L
    L.check(L.lib().cover_resample_axis(torch.from_numpy(raw).to(dev).data_ptr(), 0, rows_d.data_ptr(), 1, H, W, 3, 256, W, 0, tb[0].data_ptr(), tb[1].data_ptr(), tb[4], 0, torch.cuda.current_stream().cuda_stream), "x")
    rd = rows_d.cpu().numpy()
    print("  rows stage equal:", np.array_equal(rd, rows_h), np.abs(rd - rows_h).max())
