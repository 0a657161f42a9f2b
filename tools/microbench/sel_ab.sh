#!/bin/bash
# INVESTIGATION (GPU box): select_kernel at 8K under build flags: tools/microbench/sel_ab.sh "<flags>" ...
for flags in "$@"; do
  cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc"
  rm -f detect_kernels.o && make -s -j8 EXTRA="$flags" 2>&1 | grep -E "error" | head
  cd "$GRAFT_REPO_ROOT"
  echo "== flags: $flags"
  timeout 100 tools/microbench/batch_prof.sh 8k 1 1 hx > /dev/null 2>&1; python tools/prof_summary.py gpurun_out/prof_hx/b_results.db /tmp/x.csv | cut -d, -f1,4 | grep -E "select|emit"
  rm -rf gpurun_out/prof_hx
done
