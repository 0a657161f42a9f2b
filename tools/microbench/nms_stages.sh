#!/bin/bash
# INVESTIGATION (GPU box): nms_kernel's time on the 1/f^1.3 8K frame (and the corner-rich one) with its stages cut short (debug build;
# EFX_DEBUG = 16 x stage: 1 stop behind the prologue, 2 no exact scans, 3 stop before the survivors are written, 4 no histogram
# updates, 5 no row sums either; results invalid)
cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc"
rm -f detect_kernels.o efx_api.o && make -s -j8 EXTRA="-DEFX_DEBUG_BUILD" 2>&1 | grep -E "error" | head
cd "$GRAFT_REPO_ROOT"
for dbg in ${STAGES:-0 16 32 48 64 80}; do
  echo "== EFX_DEBUG=$dbg"
  EFX_DEBUG=$dbg timeout 200 tools/microbench/natural_prof.sh 1.3 2>&1 | grep -E "nms"
  EFX_DEBUG=$dbg timeout 100 tools/microbench/batch_prof.sh 8k 1 1 hx > /dev/null 2>&1; python tools/prof_summary.py gpurun_out/prof_hx/b_results.db /tmp/x.csv | cut -d, -f1,4 | grep -E "nms"; rm -rf gpurun_out/prof_hx
done
