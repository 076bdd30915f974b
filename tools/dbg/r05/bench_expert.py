"""The pi0 action expert's denoise passes alone (200 rows = 40 candidates x 5 suffix tokens through 18 layers, as one Euler step runs them):
us per layer-step with the deferred RMSNorm (five launches per layer) against the eight-launch path, alternating inside one process, eager and
as a replayed hipGraph of ten passes. COVER_DEFER_NORM is read per call by the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from cover_vla_amd import ops, synth
from cover_vla_amd.models import BF, Decoder, KvGeometry

dev = torch.device("cuda:0")
c = dict(synth.PI0_FULL) if hasattr(synth, "PI0_FULL") else None
dim, layers, Hq, Hkv, D, mlp = 1024, 18, 8, 1, 256, 4096
B, S, Tp = 40, 5, 279
g = synth._G(5, False, 0.02, dev, torch.bfloat16)
sd = synth.decoder_state(g, dim=dim, layers=layers, Hq=Hq, Hkv=Hkv, D=D, mlp=mlp, rms_base=0.0, vocab=8)
geom = KvGeometry(Hkv, D, [8, B], [320, S])
ex = Decoder(sd, dim=dim, layers=layers, Hq=Hq, Hkv=Hkv, D=D, mlp=mlp, act="gelu_tanh", norm="gemma", eps=1e-6, rope="pi0", n_pos=400, device="cuda:0",
             cache=geom, final_norm_bf16=False, fold_norm=True)
row_prompt = (torch.arange(B, device=dev) // 5).to(torch.int32)
row_plen = torch.full((B,), Tp, dtype=torch.int32, device=dev)
spos = (row_plen[:, None] + torch.arange(S, device=dev, dtype=torch.int32)[None]).contiguous()
vis = torch.tensor([1] + [S] * (S - 1), dtype=torch.int32, device=dev)
g1 = ex.group(B, S, spos.view(-1), [dict(region=0, length=Tp, len_of_batch=row_plen, slot_of_batch=row_prompt),
                                    dict(region=1, length=S, mask=ops.MASK_VISLEN, vis_len=vis)], 1, write_scratch=True)
x32 = torch.randn(B * S, dim, device=dev)
xb = torch.empty(B * S, dim, dtype=BF, device=dev)


def passes(n):
    for _ in range(n):
        ex.forward(xb, [g1], final_norm=True, x_f32=x32)


def timed(n_pass, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    passes(2)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        passes(n_pass)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n_pass * layers)


outs = {}
for dn in ("0", "1"):
    os.environ["COVER_DEFER_NORM"] = dn
    ops.gemm_plan_counts(reset=True)
    passes(1)
    torch.cuda.synchronize()
    outs[dn] = xb.float().clone()
    print(f"COVER_DEFER_NORM={dn}: plans per pass {[(i, n) for i, n in enumerate(ops.gemm_plan_counts()) if n]}")
rel = ((outs["1"] - outs["0"]).norm() / outs["0"].norm()).item()
print(f"deferred vs eight-launch path: rel-L2 of the pass output {rel:.2e}")
for rnd in range(3):
    for dn in ("0", "1"):
        os.environ["COVER_DEFER_NORM"] = dn
        print(f"eager  COVER_DEFER_NORM={dn}: {timed(10, 5):7.2f} us per layer-step", flush=True)
for dn in ("0", "1"):
    os.environ["COVER_DEFER_NORM"] = dn
    passes(1)
    cap = torch.cuda.Stream(device=dev)
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        with ops.Graph() as gr:
            passes(10)
    torch.cuda.current_stream().wait_stream(cap)
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr.launch(); torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            gr.launch()
        e1.record(); torch.cuda.synchronize()
        print(f"graph  COVER_DEFER_NORM={dn}: {e0.elapsed_time(e1) * 1e3 / (5 * 10 * layers):7.2f} us per layer-step", flush=True)
