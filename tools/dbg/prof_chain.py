"""Per-phase durations of the persistent decode chain: runs decode passes of a 2-layer full-width Llama-2-7B decoder (the geometry of
tests/test_chain_gpu.py) with COVER_DECODE_CHAIN = 2 (every phase its own launch), 1 (fused) and 0 (separate kernels).
   rocprofv3 --kernel-trace -d gpurun_out/pc -o pc -- python3 tools/dbg/prof_chain.py ; python tools/dbg/prof_chain.py --parse gpurun_out/pc/pc_results.db"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import re, sqlite3, statistics
    db = sqlite3.connect(sys.argv[2])
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    ks = [(re.sub(r"\(anonymous namespace\)::", "", n), s, e) for n, s, e in rows]
    ks = [(re.sub(r"\(.*", "", n)[:40], (e - s) / 1e3, s, e) for n, s, e in ks]
    idx = [i for i, k in enumerate(ks) if "decode_chain_k" in k[0] or "decode_attn_fused" in k[0] or "gemm_skinny" in k[0] or "splitk_reduce" in k[0]]
    ks = [ks[i] for i in idx]
    # split mode: pattern per pass = ssq, qkv0, [attn, o, gu, down, qkv1], [attn, o, gu, down]; fused: chain(2), [attn, chain(4)], [attn, chain(3)]
    names = [k[0] for k in ks]
    def runs(pattern_len, first):
        out = {}
        i = first
        while i + pattern_len <= len(ks):
            for j in range(pattern_len):
                out.setdefault(j, []).append(ks[i + j][1])
            i += pattern_len
        return out
    print("dispatches:", len(ks))
    marks = [i for i in range(len(ks))]
    # print the first 40 launches with durations and gaps to eyeball the pattern
    for i in range(min(len(ks), 150)):
        gap = (ks[i][2] - ks[i - 1][3]) / 1e3 if i else 0.0
        print(f"{i:3d} {ks[i][0]:40s} {ks[i][1]:8.2f} us  gap {gap:6.2f}")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_chain_gpu as T
dev = torch.device("cuda:0")
m, sd = T.build_llm(dev)
g = torch.Generator(device=dev).manual_seed(3)
x0 = torch.randn(32, 4096, device=dev, generator=g).to(torch.bfloat16)
for mode in ("2", "1", "0"):
    for _ in range(3):
        T._run(m, dev, x0, mode, 32)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    os.environ["COVER_DECODE_CHAIN"] = mode
    grp = T._group(m, dev, 32, 0)
    xs = [x0.clone() for _ in range(20)]
    e0.record()
    for x in xs:
        m.forward(x, [grp], final_norm=False)
    e1.record()
    torch.cuda.synchronize()
    print(f"COVER_DECODE_CHAIN={mode}: {e0.elapsed_time(e1) / 20 * 1e3 / 2:.1f} us per layer (2-layer pass, incl. attention)")
from cover_vla_amd import ops
ops.decode_chain_status()
