#!/bin/bash
# Tower tile edge A/B (plan_tower, EFX_TOWER_TT pins the edge): per-kernel time of pyramid_tower_kernel on FHD / 4K / 8K.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; : > $O/tower_tt.log
for tt in ${TTS:-0 32 36 40}; do
  for sz in ${SIZES:-fhd 4k}; do
    rm -rf $O/tw; EFX_TOWER_TT=$tt timeout 90 rocprofv3 --kernel-trace -d $O/tw -o t -- python tools/microbench/fhd_prof.py $sz > $O/tw.log 2>&1 < /dev/null
    echo "tt=$tt $sz rc=$? $(grep -E "^(vga|720p|fhd|2.7k|4k|8k) " $O/tw.log < /dev/null)" >> $O/tower_tt.log
    python - >> $O/tower_tt.log <<'P'
import sqlite3
con = sqlite3.connect("gpurun_out/tw/t_results.db")
for n, c, a in con.execute("select name, count(*), avg(duration)/1e3 from kernels where name like '%pyramid%' group by name"):
    print("   %-40s x%d avg %.1f us" % (n[:40], c, a))
tot = con.execute("select sum(duration)/1e3 from kernels").fetchone()[0]
print("   all kernels / 21 frames: %.1f us" % (tot / 21))
P
  done
done
cat $O/tower_tt.log
