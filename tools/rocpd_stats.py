"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (markdown/CSV-ish).
Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name[:110]


def main(path, after=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {ncol}, start, end from kernels order by start").fetchall()
    if after:  # drop everything before the first dispatch whose name contains `after` (model build / weight packing)
        for i, r in enumerate(rows):
            if after in r[0]:
                rows = rows[i:]
                break
        print(f"# restricted to dispatches from the first '{after}' on")
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 10**18, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"# {path}: {len(rows)} dispatches, total kernel time {tot/1e6:.3f} ms")
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:110s} {a[0]:7d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:9.2f} {a[2]/1e3:9.2f} {a[3]/1e3:9.2f} {100*a[1]/tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
