#!/bin/bash
# usage (GPU box): tools/microbench/sweep_hs.sh v1 v2 ...  -- rebuilds with -DHS_NT=v and prints the HashSIFT timings
for v in "$@"; do
  make -s -C cuda-efficient-features_amd/csrc clean >/dev/null 2>&1
  make -s -C cuda-efficient-features_amd/csrc EXTRA="-DHS_NT=$v" 2>&1 | grep -E "error" | head -3
  echo "== HS_NT=$v"; python tools/microbench/hs_stage_dac.py 2>&1 | tail -1; python tools/microbench/hs_stage.py 2>&1 | tail -1
done
