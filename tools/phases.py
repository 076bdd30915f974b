"""Phase breakdown of one OpenVLA-7B + CoVer decision (bench.py's pipeline) with HIP events on the main stream:
vision towers, LLM prefill, the six decode passes, the verifier tail. Usage: python tools/phases.py [--small]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cover_vla_amd import ops

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
pipe = bench.Pipeline(dev, small="--small" in sys.argv)
for _ in range(3):
    pipe.decision()
torch.cuda.synchronize()
i = pipe.inp
acc = {}
_cache = {}
R = 5
for _ in range(R):
    main = torch.cuda.current_stream()
    tr = {"events": []}
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    out = {}

    # 4 (default) = what bench.py's timed decisions do: the verifier's side work queued by a second host thread from t = 0.
    # Diagnostics: 0 = side work queued from the sampler's after-prefill hook (it then runs UNDER decode passes 1-2 and slows them:
    # the "decode1 / decode2 excess" of profiles/r03_phases.txt was this mode, not a property of the bench's schedule),
    # 1 = no side work (stale embeddings), 2 = towers on the side stream, heads in the tail, 3 = side work queued before the policy
    # 5 (default since round 6) = the verifier's towers + heads as ONE replayed hipGraph on the side stream, launched by this thread before the policy
    # 6 = the same graph launched by the second host thread (bench.py's default)
    MODE = int(os.environ.get("SIDE_MODE", "6"))

    def side_work():
        if MODE == 1 and "its" in globals().get("_cache", {}):
            out["its"] = _cache["its"]
            return
        ev = torch.cuda.Event(); ev.record(main)
        pipe.side.wait_event(ev)
        with torch.cuda.stream(pipe.side):
            pf, tf = pipe.ver.extract_shared_features(i["img384"], i["text"])
            if MODE == 2:
                out["pf"], out["tf"] = pf, tf
            else:
                out["its"] = pipe.ver.image_text_embeddings(pf, tf)
                _cache["its"] = out["its"]

    fut = None
    if MODE == 5:
        ev5 = torch.cuda.Event(); ev5.record(main)
        pipe.side.wait_event(ev5)
        with torch.cuda.stream(pipe.side):
            out["its"] = pipe.ver.shared_embeddings_graph(i["img384"], i["text"])
    if MODE == 3:   # diagnostic: side work queued BEFORE the policy (overlaps the vision phase and the start of the prefill)
        side_work()
    if MODE in (4, 6):   # side work queued by a second host thread while this one queues the policy
        import concurrent.futures
        if "pool" not in _cache:
            _cache["pool"] = concurrent.futures.ThreadPoolExecutor(max_workers=1)
        ev0 = torch.cuda.Event(); ev0.record(main)

        def threaded():
            torch.cuda.set_device(dev)
            pipe.side.wait_event(ev0)
            with torch.cuda.stream(pipe.side):
                if MODE == 6:
                    return pipe.ver.shared_embeddings_graph(i["img384"], i["text"])
                pf, tf = pipe.ver.extract_shared_features(i["img384"], i["text"])
                return pipe.ver.image_text_embeddings(pf, tf)

        fut = _cache["pool"].submit(threaded)
    tokens, _ = pipe.policy.sample(i["frame"], i["toks"], i["lens"], bench.N_SAMPLES, i["u"], 1.0, trace=tr,
                                   on_prefill_enqueued=None if MODE in (3, 4, 5, 6) else side_work)
    if fut is not None:
        out["its"] = fut.result()
    if "its" not in out:
        main.wait_stream(pipe.side)
        out["its"] = pipe.ver.image_text_embeddings(out["pf"], out["tf"])
    its = out["its"]
    hb, pad = ops.tokens_to_histories(tokens, pipe.c["tok_vocab"], pipe.centers, pipe.past_dev)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    main.wait_stream(pipe.side)
    e2 = torch.cuda.Event(enable_timing=True); e2.record()
    r = pipe.ver.score_histories(its, hb, bench.N_SAMPLES, pad=pad)
    e3 = torch.cuda.Event(enable_timing=True); e3.record()
    idx = int(r["result"][0])
    torch.cuda.synchronize()
    evs = [("t0", e0)] + tr["events"] + [("histories", e1), ("join_side", e2), ("verifier_tail", e3)]
    for (n0, a), (n1, b) in zip(evs[:-1], evs[1:]):
        acc[n1] = acc.get(n1, 0.0) + a.elapsed_time(b)
    acc["total"] = acc.get("total", 0.0) + e0.elapsed_time(e3)
print(f"SIDE_MODE={MODE}", {k: round(v / R, 3) for k, v in acc.items()})
