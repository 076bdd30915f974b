"""Per-stream timeline of the LAST decision in a rocprofv3 kernel trace (rocpd sqlite): busy time per stream, idle gaps on the
main stream, and the kernels grouped by phase. Usage: python tools/timeline.py <results.db> [n_decisions_in_trace]"""
import re, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, stream_id, queue_id, start, end from kernels order by start").fetchall()
# decisions are delimited by the patch-embedding kernel of the first vision tower ('patchify' appears twice+ per decision)
# a decision ends with the verifier's score kernel (score_rows_k, older traces: score_select_k) and the grouped arg-max behind it
idx = [i for i, r in enumerate(rows) if "score_rows" in r[0] or "score_select" in r[0]]
print("score dispatches:", len(idx))
tail = lambda i: i + 2 if i + 1 < len(rows) and "group_argmax" in rows[i + 1][0] else i + 1
lo = tail(idx[-2]) if len(idx) >= 2 else 0
hi = tail(idx[-1])
sel = rows[lo:hi]
t0, t1 = sel[0][3], max(r[4] for r in sel)
print(f"last decision: {len(sel)} dispatches, span {(t1 - t0) / 1e6:.3f} ms")
streams = {}
for n, s, q, a, b in sel:
    streams.setdefault((s, q), []).append((a, b, n))
for k, v in sorted(streams.items(), key=lambda kv: kv[1][0][0]):
    busy = sum(b - a for a, b, _ in v)
    print(f"stream {k}: {len(v):5d} kernels, busy {busy / 1e6:7.3f} ms, first +{(v[0][0] - t0) / 1e6:7.3f} ms, last end +{(v[-1][1] - t0) / 1e6:7.3f} ms")
main = max(streams.values(), key=len)
gaps = [(main[i + 1][0] - main[i][1], main[i][2], main[i + 1][2], main[i][1] - t0) for i in range(len(main) - 1)]
tot_gap = sum(g[0] for g in gaps if g[0] > 0)
print(f"main stream: idle between kernels {tot_gap / 1e6:.3f} ms over {len(gaps)} gaps (avg {tot_gap / len(gaps) / 1e3:.2f} us)")
big = sorted(gaps, key=lambda g: -g[0])[:12]
def short(n):
    return re.sub(r"\(.*", "", n)[:50]


for g in big:
    print(f"  gap {g[0] / 1e3:8.1f} us at +{g[3] / 1e6:7.3f} ms after {short(g[1])} before {short(g[2])}")
# coarse phases on the main stream by time buckets of 2 ms
print("main-stream kernel time by 4 ms bucket (ms busy):")
B = 4e6
nb = int((t1 - t0) / B) + 1
acc = [0.0] * nb
for a, b, n in main:
    acc[int((a - t0) / B)] += b - a
print("  " + " ".join(f"{x / 1e6:.2f}" for x in acc))

# phase markers on the main stream (offsets in ms from the first kernel of the decision)
def first(pred, seq=main):
    for a, b, n in seq:
        if pred(n):
            return (a - t0) / 1e6
    return None


def last(pred, seq=main):
    r = None
    for a, b, n in seq:
        if pred(n):
            r = (b - t0) / 1e6
    return r


print("markers (ms): first patchify %.3f | first decode_attn %.3f | last token_select end %.3f | score / arg-max end %.3f" % (
    first(lambda n: "patchify" in n) or -1, first(lambda n: "decode_attn" in n) or -1,
    last(lambda n: "token_select" in n) or -1, last(lambda n: "score_rows" in n or "score_select" in n or "group_argmax" in n) or -1))
tail0 = last(lambda n: "token_select" in n)
tail = [(a, b, n) for a, b, n in main if (a - t0) / 1e6 >= tail0]
agg = {}
for a, b, n in tail:
    k = short(n)
    agg.setdefault(k, [0, 0])
    agg[k][0] += 1
    agg[k][1] += b - a
print(f"tail after the last token_select: {len(tail)} kernels, busy {sum(v[1] for v in agg.values()) / 1e6:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"   {k:50s} x{v[0]:4d} {v[1] / 1e3:9.1f} us")

# ---- per-pass table: the main stream cut at the token_select launches -- phase 0 = vision + prefill + the first
# head, phases 1..6 = the six decode passes (32 layers + lm_head + token_select each), then the verifier tail
cuts = [k for k, (a, b, n) in enumerate(main) if "token_select" in n]
if len(cuts) >= 2:
    print("per-pass table (main stream, cut at token_select): kernels | sum of kernel durations ms | sum of gaps ms | span ms | co-running side-stream kernel ms")
    others = [v for k, v in streams.items() if v is not main]
    lo_k = 0
    for pi, c in enumerate(cuts):
        seg = main[lo_k:c + 1]
        if seg:
            k_ms = sum(b - a for a, b, _ in seg) / 1e6
            span = (seg[-1][1] - seg[0][0]) / 1e6
            side = sum(max(0, min(b, seg[-1][1]) - max(a, seg[0][0])) for v in others for a, b, _ in v) / 1e6
            print(f"  phase {pi}: {len(seg):5d} | {k_ms:7.3f} | {span - k_ms:6.3f} | {span:7.3f} | {side:6.3f}")
        lo_k = c + 1
    # the slowest kernels of the first two decode passes against the same kernel's median over passes 3-6
    import statistics
    if len(cuts) >= 7:
        def by_name(seg):
            d = {}
            for a, b, n in seg:
                d.setdefault(short(n), []).append((b - a) / 1e3)
            return d
        late = by_name(main[cuts[2] + 1:cuts[6] + 1])
        for pi in (1, 2):
            d = by_name(main[cuts[pi - 1] + 1:cuts[pi] + 1])
            rows = []
            for n, v in d.items():
                if n in late and len(v) >= 8:
                    rows.append((sum(v) - statistics.median(late[n]) * len(v), n, statistics.median(v), statistics.median(late[n])))
            rows.sort(reverse=True)
            print(f"  decode pass {pi} vs passes 3-6 (excess us over the late median x launches | kernel | median here | median late):")
            for ex, n, m0, m1 in rows[:5]:
                print(f"     {ex:8.1f} | {n:50s} | {m0:7.2f} | {m1:7.2f}")
