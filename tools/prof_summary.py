#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) into a small CSV for profiles/.

usage: tools/prof_summary.py gpurun_out/<dir>/<name>_results.db profiles/<name>.csv
Columns: kernel, calls, total_us, avg_us, min_us, max_us, pct, vgpr, lds_bytes, workgroup, grid(example)
"""
import csv
import sqlite3
import sys


def main(db, out):
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
        "max(vgpr_count), max(lds_size), max(workgroup_x), max(grid_x) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "lds_bytes", "workgroup_x", "max_grid_x"])
        for r in rows:
            name = r[0].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            w.writerow([name, r[1], f"{r[2]:.1f}", f"{r[3]:.2f}", f"{r[4]:.2f}", f"{r[5]:.2f}", f"{100 * r[2] / total:.2f}",
                        r[6], r[7], r[8], r[9]])
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
