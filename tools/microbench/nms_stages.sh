#!/bin/bash
# INVESTIGATION (GPU box): nms_kernel's time on the density cases with the stage knobs of a debug build
# (EFX_DEBUG = 16: prologue only, 32: no exact scans, 48: no survivor write; results are NOT valid with a knob set)
cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc" && rm -f detect_kernels.o efx_api.o && make -s EXTRA=-DEFX_DEBUG_BUILD 2>&1 | grep -E " error" | head -3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in 0 16 32 48; do
  EFX_DEBUG=$k rocprofv3 --kernel-trace -d gpurun_out/prof_nmss -o nms -- python tools/microbench/nms_density.py > gpurun_out/nmss.log 2>&1
  python - <<P
import sqlite3
con = sqlite3.connect("gpurun_out/prof_nmss/nms_results.db")
r = [x[0] for x in con.execute("select duration/1e3 from kernels where name like '%nms_kernel%' order by start").fetchall()]
print("EFX_DEBUG=$k", " ".join("%s %.1f" % (lab, sum(r[11*i+1:11*i+11]) / 10) for i, lab in enumerate(("c34_r5", "3x", "4k", "8k"))))
P
  rm -rf gpurun_out/prof_nmss
done
