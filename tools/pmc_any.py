"""Average of every collected PMC counter per kernel from a rocprofv3 --pmc rocpd database.
Usage: python tools/pmc_any.py <db> [kernel-substring]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for n, c, v in rows:
    n = re.sub(r"\(.*", "", n)[:70]
    if flt in n:
        agg[n][c].append(v)
for n, cs in agg.items():
    print(n)
    for c, v in sorted(cs.items()):
        print(f"    {c:36s} n={len(v):5d} avg={sum(v)/len(v):16.1f}")
