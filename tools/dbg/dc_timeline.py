"""Per-workgroup timeline of the persistent decode chain (library built with -DCOVER_DC_DEBUG, selected with COVER_LIB_PATH):
stamps of thread 0 per phase of the LAST 4-phase launch: 0 phase start, 1 weight window issued, 2 seam passed (barrier + acquire + block
barrier), 3 rstd done + activation window issued, 4 first step done, 5 main loop done, 6 k-slice sums in LDS, 7 stores drained.
Prints, per phase, the median / max over workgroups of each interval in us.  MODE=1 fused (default) / 2 split."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import test_chain_gpu as T
from cover_vla_amd import _lib as L
dev = torch.device("cuda:0")
m, sd = T.build_llm(dev)
g = torch.Generator(device=dev).manual_seed(3)
x0 = torch.randn(32, 4096, device=dev, generator=g).to(torch.bfloat16)
mode = os.environ.get("MODE", "1")
for _ in range(3):
    T._run(m, dev, x0, mode, 32)
buf = (C.c_ulonglong * (256 * 4 * 8))()
h = L.lib()
h.cover_dc_debug.argtypes = [C.POINTER(C.c_ulonglong)]
assert h.cover_dc_debug(buf) == 0
v = list(buf)
names = ["o_proj", "gate_up", "down", "(last launch: no qkv)"]
# the last launch of the pass is [o_proj, gate_up, down] (3 phases): phase slot 3 holds the qkv phase of the launch before it
t0 = min(v[(b * 4 + 0) * 8 + 0] for b in range(256))
for p in range(3):
    rows = [[(v[(b * 4 + p) * 8 + s] - t0) / 100.0 for s in range(8)] for b in range(256)]
    act = [r for r in rows if r[5] > r[3]]
    if not act:
        continue
    med = lambda i: statistics.median(r[i] for r in act)
    mx = lambda i: max(r[i] for r in act)
    mn = lambda i: min(r[i] for r in act)
    print(f"{names[p]:8s} ({len(act)} active workgroups)   stamp: median / min / max  [us from the launch's first stamp]")
    for i, nm in enumerate(["phase start", "window issued", "seam passed", "x window issued", "first step done", "loop done", "k-sums in LDS", "stores drained"]):
        print(f"    {i} {nm:18s} {med(i):8.2f} {mn(i):8.2f} {mx(i):8.2f}")
