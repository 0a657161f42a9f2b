#!/bin/bash
# INVESTIGATION (GPU box): harris_kernel's time at 8K with parts compiled out (results WRONG: timing only)
for flags in "" "-DEFX_X_NO_RANK" "-DEFX_X_NO_BITMAP" "-DEFX_X_NO_RANK -DEFX_X_NO_BITMAP"; do
  cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc"
  rm -f detect_kernels.o && make -s -j8 EXTRA="$flags" 2>&1 | grep -E "error" | head
  cd "$GRAFT_REPO_ROOT"
  echo "== flags: $flags"
  tools/microbench/batch_prof.sh 8k 1 1 hx > /dev/null 2>&1
  python tools/prof_summary.py gpurun_out/prof_hx/b_results.db /tmp/x.csv | grep -E "fast_kernel|harris|nms_kernel" | cut -d, -f1,4
  rm -rf gpurun_out/prof_hx
done
