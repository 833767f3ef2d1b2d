"""Prints the last N kernel dispatches of a rocprofv3 rocpd sqlite database in launch order:
start offset (us), duration (us), idle gap before the launch (us), kernel name.  Shows where a step's wall
time goes when the sum of kernel durations is well below it.  Usage: python tools/rocpd_timeline.py <db> [N]"""
import sqlite3
import sys

db = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()[-n:]
t0, prev_end = rows[0][1], rows[0][1]
busy = 0.0
for name, s, e in rows:
    busy += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {max(0, s - prev_end) / 1e3:8.1f}  {name[:90]}")
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {busy:.1f} us")
