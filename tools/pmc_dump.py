"""Prints per-kernel PMC counter averages from a rocprofv3 rocpd database (kernel name filter optional)."""
import sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
q = "select kernel_name, counter_name, value from counters_collection" if "kernel_name" in cols else None
if q is None:
    print(cols); sys.exit(0)
agg = defaultdict(list)
for k, n, v in c.execute(q):
    if flt in k: agg[(k[:60], n)].append(v)
for (k, n), v in sorted(agg.items()):
    print(f"{k:60s} {n:28s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
