"""Experiment: the verifier's side stream confined to a subset of the CUs (hipExtStreamCreateWithCUMask) so that its tower kernels stop
competing with the launch-bound vision chains of the policy on every CU. Prints ms per decision per mask."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from cover_vla_amd import _lib as L
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
pipe = bench.Pipeline(dev, small=False)
h = L.lib()
fn = h.hipExtStreamCreateWithCUMask
fn.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
fn.restype = C.c_int

def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = C.c_void_p()
    rc = fn(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)

def run(tag):
    for _ in range(3):
        pipe.decision()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        pipe.decision()
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per decision", flush=True)

run("unmasked side stream")
full = (1 << 256) - 1
masks = {"low 64 bits": (1 << 64) - 1, "low 128 bits": (1 << 128) - 1, "every 4th bit (64 CUs)": sum(1 << i for i in range(0, 256, 4)),
         "every 2nd bit (128 CUs)": sum(1 << i for i in range(0, 256, 2)), "high 32 bits": ((1 << 32) - 1) << 224}
for name, m in masks.items():
    pipe.side = masked_stream(m)
    run(name)
pipe.side = torch.cuda.Stream(device=dev)
run("unmasked again")
