"""Summarises a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table:
calls, total / average / min / max duration (us).  Usage: python tools/rocpd_stats.py <db> [skip_first_n_per_kernel]"""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
agg = defaultdict(list)
for n, s, e in rows:
    agg[n].append((e - s) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':70s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    short = n if len(n) <= 70 else n[:67] + "..."
    print(f"{short:70s} {len(v):6d} {sum(v):11.1f} {sum(v)/len(v):9.2f} {min(v):9.2f} {max(v):9.2f} {100*sum(v)/tot:6.2f}")
print(f"total kernel time {tot/1e3:.3f} ms over {len(rows)} dispatches")
