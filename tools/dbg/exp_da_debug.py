"""Per-block timeline of the fused decode attention (library built with -DCOVER_DA_DEBUG): start, phase 1 done (q / k_new /
v_new in LDS), tile phase done, end -- relative to the earliest block start. OpenVLA-7B decode shape: N = 32, H = 32, D = 128.
MODE=cold: 600 MB written between launches (data and code evicted); MODE=back2back: the same launch three times in a row."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from cover_vla_amd import ops, _lib as L
from test_kernels_gpu import bf, make_cache
dev = torch.device("cuda:0")
H, D, N, write_t = 32, 128, 32, 3
T0, T1, cap2, npos = 257, 24, 32, 320
g = torch.Generator().manual_seed(1)
ncol = 3 * H * D
part = (torch.randn(2, N, ncol, generator=g) * 0.6).to(dev)
bias = (torch.randn(ncol, generator=g) * 0.1).to(dev)
qkv = torch.zeros(N, ncol, dtype=torch.bfloat16, device=dev)
pos = torch.randint(0, npos, (N,), generator=g, dtype=torch.int32).to(dev)
inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
ang = torch.arange(npos).float()[:, None] * inv[None]
cos, sin = ang.cos().to(dev), ang.sin().to(dev)
k0, v0 = bf(torch.randn(1, T0, H, D, generator=g)), bf(torch.randn(1, T0, H, D, generator=g))
k1, v1 = bf(torch.randn(8, T1, H, D, generator=g)), bf(torch.randn(8, T1, H, D, generator=g))
k2, v2 = bf(torch.randn(N, cap2, H, D, generator=g)), bf(torch.randn(N, cap2, H, D, generator=g))
slot1 = (torch.arange(N) // 4 % 8).to(torch.int32)
zero = torch.zeros(N, dtype=torch.int32)
c0, c1, c2 = make_cache(k0, v0, dev), make_cache(k1, v1, dev), make_cache(k2, v2, dev, cap2)
segs = [ops.Segment(c0[0], c0[1], c0[2], c0[3], length=T0, slot_of_batch=zero.to(dev)),
        ops.Segment(c1[0], c1[1], c1[2], c1[3], length=T1, slot_of_batch=slot1.to(dev)),
        ops.Segment(c2[0], c2[1], c2[2], c2[3], length=write_t + 1)]
out = torch.empty(N, H * D, dtype=torch.bfloat16, device=dev)
fn = L.lib().cover_da_debug
fn.argtypes = [C.c_void_p]
flush = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
mode = os.environ.get("MODE", "cold")
def launch():
    ops.decode_attention_fused(qkv, N, H, D, D ** -0.5, segs, write_t, out, positions=pos, cos=cos, sin=sin, rope_mode=1, partial=part, bias=bias)
for rep in range(3):
    if mode == "cold":
        flush.zero_(); launch()
    else:
        launch(); launch(); launch()
    torch.cuda.synchronize()
    buf = np.zeros(4096, dtype=np.uint64); fn(buf.ctypes.data)
    t = buf.reshape(512, 8).astype(np.float64) / 100.0
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    q = lambda x: f"{np.percentile(x - t0, 5):5.1f}/{np.median(x - t0):5.1f}/{(x - t0).max():5.1f}"
    try:
        fw = L.lib().cover_da_debug_waves
        fw.argtypes = [C.c_void_p]
        wb = np.zeros(128, dtype=np.uint64); fw(wb.ctypes.data)
        wv = wb.reshape(16, 8).astype(np.float64)
        w0 = wv[:, 6].min() / 100.0
        print("   waves of workgroup (tile 1, head 7, value block 2) -- role: at barrier / past barrier / first tile folded / all folded (tiles), us from the workgroup's start:")
        print("   " + "  ".join(f"w{i}[{'ABC'[int(wv[i, 4])]}] {wv[i, 0] / 100.0 - w0:4.1f}/{wv[i, 1] / 100.0 - w0:4.1f}/{wv[i, 2] / 100.0 - w0:4.1f}/{wv[i, 3] / 100.0 - w0:4.1f}({int(wv[i, 5])})" for i in range(16)))
    except AttributeError:
        pass
    print(f"{mode} rep {rep}: blocks {len(t)}  start {q(t[:,0])}  idx {q(t[:,4])}  partials {q(t[:,5])}  lds-written {q(t[:,6])}  phase1 {q(t[:,1])}  tiles {q(t[:,2])}  end {q(t[:,3])}  (p5/median/max us)")
