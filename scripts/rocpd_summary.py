#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output) into the per-kernel table that
`--stats` prints for CSV output: calls, total / average / min / max duration (us), share.  Also dumps PMC counters
per kernel when the run collected any.   usage: rocpd_summary.py results.db [> profiles/rNN_kernel_stats.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'pct':>6s}")
for n, c, s, a, mn, mx in rows:
    print(f"{n[:70]:70s} {c:6d} {s/1e3:12.1f} {a/1e3:12.1f} {mn/1e3:12.1f} {mx/1e3:12.1f} {100*s/tot:6.2f}")
try:
    pm = cur.execute("select k.name, p.name, count(*), avg(e.value) from pmc_events e join kernels k on k.id = e.event_id "
                     "join pmc_info p on p.id = e.pmc_id group by 1, 2").fetchall()
except Exception:
    pm = []
    try:
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        pm = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by 1, 2").fetchall() if "counter_name" in ccols else []
    except Exception:
        pm = []
if pm:
    print("\nPMC counters (average per dispatch)")
    for k, p, c, v in pm:
        print(f"{k[:60]:60s} {p:28s} n={c:4d} avg={v:.1f}")
