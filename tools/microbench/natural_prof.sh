#!/bin/bash
# rocprofv3 kernel stats of 8K detectAndCompute BAD512 on the natural-statistics frames: tools/microbench/natural_prof.sh [beta ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for beta in ${@:-1.3 1.0}; do
  rm -rf gpurun_out/prof_nat
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_nat -o nat -- python tools/microbench/natural_prof.py $beta 12 > gpurun_out/prof_nat.log 2>&1
  echo "== 1/f^$beta: $(grep keypoints gpurun_out/prof_nat.log)"
  python tools/prof_summary.py gpurun_out/prof_nat/nat_results.db gpurun_out/natural_${beta}_kernel_stats.csv | cut -d, -f1,2,4 | head -13
done
rm -rf gpurun_out/prof_nat
