"""FETCH_SIZE (rocprofv3 --pmc FETCH_SIZE --kernel-trace) per kernel, with the gfx950 correction from
/opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE reports half the bytes of a wide coalesced stream -> x2.
Usage: python tools/pmc_stats.py <db> [out.json [lib_sha16]]   (lib_sha16 = sha256(libcover_hip.so)[:16] of the build that ran: bench.py quotes
the traffic figure only when it matches the library it is running)"""
import collections, json, re, sqlite3, sys

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select kernel_name, grid_size_x, grid_size_y, workgroup_size_x, value from counters_collection where counter_name='FETCH_SIZE'").fetchall()
agg = collections.defaultdict(list)
for n, gx, gy, wx, v in rows:
    agg[(re.sub(r"\(.*", "", n)[:60], gx // max(wx, 1), gy)].append(v)
print(f"# {sys.argv[1]}: FETCH_SIZE in KiB as reported; 'corrected MiB' = 2 x reported (gfx950 wide-load under-count)")
print(f"{'kernel':62s} {'grid':>12s} {'calls':>6s} {'avg_KiB':>10s} {'corrected_MiB':>14s}")
sk_bytes, sk_n = 0.0, 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    avg = sum(v) / len(v)
    shown = globals().get("shown", 0)
    if shown < 40:
        print(f"{k[0]:62s} {str(k[1])+'x'+str(k[2]):>12s} {len(v):6d} {avg:10.0f} {2*avg/1024:14.1f}")
        shown += 1
    # bench.py's roofline class: weight-streaming launches with >= 16 MB of weights (the 7B decode GEMMs and lm_head)
    if ("gemm_skinny2" in k[0] or "gemm_skinny3" in k[0]) and 2 * avg * 1024 >= 16.0e6:
        sk_bytes += 2 * sum(v) * 1024
        sk_n += len(v)
if len(sys.argv) > 2 and sk_n:
    json.dump({"kernel": "gemm_skinny2 + gemm_skinny3", "launches": sk_n, "hbm_fetch_bytes_per_launch": sk_bytes / sk_n,
               "correction": "FETCH_SIZE x2 (gfx950 wide coalesced loads, MI355X_MICROARCH.md HBM section)",
               "source": sys.argv[1], "lib_sha16": sys.argv[3] if len(sys.argv) > 3 else None}, open(sys.argv[2], "w"), indent=1)
    print(f"# gemm_skinny2 + gemm_skinny3: {sk_n} launches, corrected HBM fetch {sk_bytes/sk_n/1e6:.1f} MB per launch")
