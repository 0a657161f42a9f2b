#!/bin/bash
# INVESTIGATION (GPU box): nms_kernel against nms_packed_kernel on the natural-statistics 8K frames (EFX_PACK_NMS = 0 / unset)
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  for v in 0 1; do
    echo "== EFX_PACK_NMS=$v"
    EFX_PACK_NMS=$v tools/microbench/natural_prof.sh ${1:-1.3} 2>&1 | grep -E "==|nms|harris|fast_k|select"
  done
done
