#!/bin/bash
# INVESTIGATION (GPU box): nms_kernel on the density cases + FHD for builds with extra -D flags
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for fl in "$@"; do
  if [ -n "$fl" ]; then (cd cuda-efficient-features_amd/csrc && rm -f detect_kernels.o && make -s EXTRA="$fl" 2>&1 | grep -E " error" | head -3); fi
  rocprofv3 --kernel-trace -d gpurun_out/prof_nms -o nms -- python tools/microbench/nms_density.py > gpurun_out/nms_density.log 2>&1
  python - "$fl" <<'P'
import sqlite3, sys
con = sqlite3.connect("gpurun_out/prof_nms/nms_results.db")
r = [x[0] for x in con.execute("select duration/1e3 from kernels where name like '%nms_kernel%' order by start").fetchall()]
print("flags '%s':" % sys.argv[1], " ".join("%s %.1f" % (lab, sum(r[11*i+1:11*i+11]) / 10) for i, lab in enumerate(("c34_r5", "3x", "4k", "8k"))))
P
  rm -rf gpurun_out/prof_nms
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fhd -o f -- python tools/microbench/fhd_prof.py fhd 40000 > gpurun_out/fhd.log 2>&1
  python tools/prof_summary.py gpurun_out/prof_fhd/f_results.db gpurun_out/fhd.csv | grep nms_kernel | cut -c1-60; rm -rf gpurun_out/prof_fhd
done
