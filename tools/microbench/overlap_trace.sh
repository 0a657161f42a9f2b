#!/bin/bash
# INVESTIGATION (GPU box): how the three streams of the bench loop share the GPU -- from a kernel trace of `python bench.py`:
# wall time, time with 0 / 1 / 2 / 3+ kernels running, per-stream busy time.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ovl; timeout 200 rocprofv3 --kernel-trace -d gpurun_out/ovl -o t -- python bench.py --no-cpu-baseline --no-configs --sustain-seconds 0 --steps 10 --warmup 2 > gpurun_out/ovl.log 2>&1 < /dev/null
tail -1 gpurun_out/ovl.log | cut -c1-120
python - <<'P'
import sqlite3
con = sqlite3.connect("gpurun_out/ovl/t_results.db")
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
print(cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = con.execute("select start, end, name%s from kernels where name like '%%anonymous%%' order by start" % ((", " + qcol) if qcol else "")).fetchall()
# the timed region: skip the first 20 % and the last 25 % of the launches (warm-up, isolated kernels + latency loop after it)
n = len(rows); rows = rows[int(0.2 * n):int(0.55 * n)]
ev = []
for r in rows: ev.append((r[0], 1)); ev.append((r[1], -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
hist = {}; cur = 0; last = t0
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
wall = t1 - t0
print("wall %.2f ms, kernels %d" % (wall / 1e6, len(rows)))
for k in sorted(hist): print("   %d kernel(s) running: %5.1f %%" % (k, 100.0 * hist[k] / wall))
print("   sum of kernel durations / wall = %.2f" % (sum(r[1] - r[0] for r in rows) / wall))
if qcol:
    per = {}
    for r in rows: per[r[3]] = per.get(r[3], 0) + (r[1] - r[0])
    print("   busy per %s: %s" % (qcol, {k: round(100.0 * v / wall, 1) for k, v in per.items()}))
P
rm -rf gpurun_out/ovl
