"""What the two policy towers of the OpenVLA profile cost alone, back to back on one stream, and overlapped (the default graph):
upper bound of what grouped (two-tower) launches could gain. Usage: python tools/dbg/vision_split.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cover_vla_amd import ops
from cover_vla_amd.openvla import IMAGENET_MEAN, IMAGENET_STD

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
pipe = bench.Pipeline(dev, small=False)
pol, i = pipe.policy, pipe.inp
for _ in range(3):
    pol.encode_image(i["frame"])
torch.cuda.synchronize()
st = pol._vision_static(*i["frame"].shape[:3])
mul_d = [1.0 / (255.0 * s) for s in IMAGENET_STD]
add_d = [-m / s for m, s in zip(IMAGENET_MEAN, IMAGENET_STD)]


def timed(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def dino():
    pol.dino.forward(pol.dino.embed(st["frame"], mul_d, add_d, bufs=st["d"]))


def sig():
    pol.siglip.forward(pol.siglip.embed(st["frame"], [1.0 / (255.0 * 0.5)] * 3, [-1.0] * 3, bufs=st["s"]))


def graphed(f):
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with ops.Graph() as g:
            f()
    torch.cuda.current_stream().wait_stream(s)
    return g.launch


print(f"DINOv2 alone  eager {timed(dino):.3f} ms | graph {timed(graphed(dino)):.3f} ms")
print(f"SigLIP alone  eager {timed(sig):.3f} ms | graph {timed(graphed(sig)):.3f} ms")
print(f"both, one stream    {timed(lambda: (dino(), sig())):.3f} ms | graph {timed(graphed(lambda: (dino(), sig()))):.3f} ms")
print(f"encode_image (default graph: towers overlapped + projector) {timed(lambda: pol.encode_image(i['frame'])):.3f} ms")
