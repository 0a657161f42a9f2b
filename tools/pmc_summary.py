#!/usr/bin/env python3
"""Per-kernel sums of the PMC counters in a rocprofv3 rocpd .db:  tools/pmc_summary.py <db> [kernel-substring]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
rows = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
agg = {}
for k, c, v, n in rows:
    if sub and sub not in k: continue
    name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    agg.setdefault(name, {})[c] = (v, n)
for k, d in agg.items():
    n = max(x[1] for x in d.values())
    print(f"{k} (dispatches {n})")
    for c, (v, _) in sorted(d.items()):
        print(f"   {c:28s} {v:16.0f}  per-dispatch {v / n:14.0f}")
