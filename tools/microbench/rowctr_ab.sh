#!/bin/bash
# INVESTIGATION (GPU box): kernel times of a single 8K frame by the stride of the per-row counters (EFX_ROWCTR_PAD ints of padding)
for pad in 30 14 6 2; do
  cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc"
  rm -f detect_kernels.o efx_api.o bad_kernel.o && make -s -j8 EXTRA=-DEFX_ROWCTR_PAD=$pad 2>&1 | grep -E "error" | head
  cd "$GRAFT_REPO_ROOT"
  echo "== pad $pad (stride $((8 + 4 * pad)) bytes)"
  tools/microbench/batch_prof.sh 8k 1 1 rc$pad > /dev/null 2>&1
  python tools/prof_summary.py gpurun_out/prof_rc$pad/b_results.db /tmp/x.csv | grep -E "fast_kernel|harris|nms_kernel|select_kernel" | cut -d, -f1,4
done
