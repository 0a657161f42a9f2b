#!/bin/bash
# INVESTIGATION (GPU box): HashSIFT C4 kernel times for the shipped build and for builds with extra -D flags
# usage: tools/microbench/hs_ab.sh "" "-DHS_NO_WIDE" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for fl in "$@"; do
  if [ -n "$fl" ]; then (cd cuda-efficient-features_amd/csrc && rm -f hashsift_kernels.o && make -s EXTRA="$fl" 2>&1 | grep -E "error" | head -3); fi
  rocprofv3 --kernel-trace --stats -d $O/prof_hsab -o hs -- python tools/microbench/hs_stage.py --only 0 > $O/prof_hsab.log 2>&1
  echo "== flags: '$fl'"; python tools/prof_summary.py $O/prof_hsab/hs_results.db $O/hsab.csv | grep -E "patch_sift|project_sign|hs_record" | cut -c1-90; rm -rf $O/prof_hsab
done
