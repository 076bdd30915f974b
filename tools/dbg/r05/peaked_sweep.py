"""Round 5: which re-scaling of the synthetic 7B checkpoint gives decisions a margin? For each (residual scale, embedding scale, head sigma)
the greedy M = 1 vs M = 8 comparison of tests/test_fullsize_gpu.py (d): steps whose top-1 / top-2 margin exceeds twice the logit difference."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import bench
dev = torch.device("cuda:0")
combos = [("1.0", "1.0", "1.5"), ("1.0", "1.0", "2.5"), ("0.5", "1.0", "1.5"), ("1.0", "2.0", "1.5"), ("0.5", "2.0", "2.0"), ("1.0", "3.0", "1.5"), ("0.25", "1.0", "1.5")]   # (residual x (2L)^-1/2, embedding scale, head sigma); the last one is the default
for res, emb, sig in combos:
    os.environ.update(COVER_SYNTH_RES=res, COVER_SYNTH_EMBED=emb, COVER_SYNTH_SIGMA=sig)
    pipe = bench.Pipeline(dev, small=False, members=1)
    i, P = pipe.inp, 8
    t8 = {}
    g8, _ = pipe.policy.sample(i["frame"], i["toks"], i["lens"], 1, trace=t8)
    V = pipe.c["tok_vocab"]
    dec, dec10, errs, distinct = 0, 0, [], set()
    for p_ in range(P):
        t1 = {}
        g1, _ = pipe.policy.sample(i["frame"], i["toks"][p_:p_ + 1], i["lens"][p_:p_ + 1], 1, trace=t1, force_tokens=g8[p_:p_ + 1].contiguous())
        for s_ in range(7):
            a8, a1 = t8["logits"][s_][p_, :V].float(), t1["logits"][s_][0, :V].float()
            err = float((a8 - a1).abs().max()); top = a8.topk(2).values; m = float(top[0] - top[1])
            errs.append(err / float(a8.std())); dec += m > 2 * err; dec10 += m > 10 * err; distinct.add(int(g8[p_, s_]))
    # sampled decision: diversity of the N = 32 candidates
    _, tok, _ = pipe.decision()
    print(f"res {res} embed {emb} sigma {sig}: decided(2x) {dec}/56 decided(10x) {dec10}/56 median err/std {sorted(errs)[28]:.4f} max {max(errs):.3f} distinct greedy tokens {len(distinct)} "
          f"distinct sampled rows {len(set(map(tuple, tok.cpu().tolist())))}/32 action-token share {float(((g8 >= pipe.c['tok_vocab'] - pipe.c['n_bins'])).float().mean()):.2f}", flush=True)
    del pipe
    torch.cuda.empty_cache()
