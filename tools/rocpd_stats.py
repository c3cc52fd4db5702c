"""Summarise a rocprofv3 rocpd .db (kernel trace) as per-kernel stats CSV: python tools/rocpd_stats.py results.db [skip_first_n_per_kernel]"""
import sqlite3
import sys
from collections import OrderedDict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
stats = OrderedDict()
for n, s, e in rows:
    stats.setdefault(n, []).append(e - s)
tot = sum(sum(v) for v in stats.values())
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for n, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
    print(f'"{n}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100 * sum(v) / tot:.2f},{min(v)},{max(v)}')
