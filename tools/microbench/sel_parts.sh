cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/microbench/level_stats.py
rocprofv3 --kernel-trace -d gpurun_out/prof_sel -o sel -- python tools/microbench/select_parts.py > gpurun_out/sel.log 2>&1
python - <<'P'
import sqlite3
con = sqlite3.connect("gpurun_out/prof_sel/sel_results.db")
for r in con.execute("select name, duration/1e3 from kernels where name like '%select_kernel%' order by start").fetchall(): print(r)
P
rm -rf gpurun_out/prof_sel
