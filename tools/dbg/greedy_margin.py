"""Why can M = 1 and M = 8 greedy runs differ? Prints, per decode step of prompt 0, the top-1 / top-2 logit margin of both runs and the
largest logit difference between them (COVER_DECODE_CHAIN as set by the caller)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda:0")
pipe = bench.Pipeline(dev, small=False)
i = pipe.inp
for mode in ("0", "1"):
    os.environ["COVER_DECODE_CHAIN"] = mode
    t8, t1 = {}, {}
    g8, _ = pipe.policy.sample(i["frame"], i["toks"], i["lens"], 1, trace=t8)
    g1, _ = pipe.policy.sample(i["frame"], i["toks"][:1], i["lens"][:1], 1, trace=t1)
    print(f"CHAIN={mode}: g8[0] {g8[0].tolist()}  g1[0] {g1[0].tolist()}")
    for s in range(7):
        a, b = t8["logits"][s][0, :32000].float(), t1["logits"][s][0, :32000].float()
        top = a.topk(2).values
        print(f"   step {s}: margin(M=8) {float(top[0] - top[1]):.4f}  max |logit(M=8) - logit(M=1)| {float((a - b).abs().max()):.4f}  argmax {int(a.argmax())} / {int(b.argmax())}")
        if int(a.argmax()) != int(b.argmax()):
            break
