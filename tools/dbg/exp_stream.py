"""HBM ceiling probe: pure streaming read of B bytes with the weight-streaming kernel's launch shape, captured in a graph
over rotating buffers (> Infinity Cache)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cover_vla_amd import ops, _lib as L
h = L.lib()
h.cover_debug_stream_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(4, dtype=torch.int32, device=dev)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for mb in (33.5, 100, 180, 262, 1000):
        n = int(mb * 1e6) // 16 * 16
        ncopy = max(2, int(700e6 // n) + 1)
        bufs = [torch.randint(0, 2**31 - 1, (n // 4,), dtype=torch.int32, device=dev) for _ in range(ncopy)]
        for blocks in (256, 512, 1024, 2048):
            for nt in (0, 1):
                reps = 4 * ncopy
                def body():
                    for i in range(reps):
                        h.cover_debug_stream_read(bufs[i % ncopy].data_ptr(), n, blocks, nt, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
                body(); torch.cuda.synchronize()
                with ops.Graph() as g:
                    body()
                g.launch(); torch.cuda.synchronize()
                t = ops.Timer(); t.start()
                for _ in range(3): g.launch()
                ms = t.stop() / (3 * reps)
                print(f"{mb:7.1f} MB blocks {blocks:5d} nt {nt}: {ms*1e3:7.2f} us  {n/ms/1e6:7.0f} GB/s", flush=True)
        del bufs
