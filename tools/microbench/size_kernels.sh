#!/bin/bash
# Per-kernel average time of detectAndCompute BAD512 on one frame size (one stream, sync per call): tools/microbench/size_kernels.sh fhd 4k
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for sz in "$@"; do
  rm -rf gpurun_out/szk; timeout 120 rocprofv3 --kernel-trace -d gpurun_out/szk -o t -- python tools/microbench/fhd_prof.py $sz > gpurun_out/szk.log 2>&1 < /dev/null
  echo "== $sz: $(grep -E '^(vga|720p|fhd|2.7k|4k|8k) ' gpurun_out/szk.log)"
  python - <<'P'
import sqlite3
con = sqlite3.connect("gpurun_out/szk/t_results.db")
tot = 0
for n, c, a in con.execute("select name, count(*), avg(duration)/1e3 from kernels where name not like '%at::%' and name not like '%rocclr%' group by name order by min(start)"):
    print("   %-34s x%d avg %6.1f us" % (n.replace("(anonymous namespace)::", "").replace("void ", "")[:34], c, a)); tot += a
print("   sum %.1f us" % tot)
P
done
rm -rf gpurun_out/szk
