#!/bin/bash
# usage (GPU box): tools/microbench/sweep_blur_shapes.sh "6 5" "8 8" ...  -- rebuilds with the row-pass / column-pass item
# shapes of blur_window.h (EFX_BLUR_RO outputs x 2 rows, 2 columns x EFX_BLUR_CR rows) and prints the kernel times
for rc in "$@"; do
  set -- $rc
  make -s -C cuda-efficient-features_amd/csrc clean >/dev/null 2>&1
  make -s -C cuda-efficient-features_amd/csrc EXTRA="-DEFX_BLUR_RO=$1 -DEFX_BLUR_CR=$2" 2>&1 | grep -E "error" | head -3
  echo "== RO=$1 CR=$2"
  tools/microbench/prof_dbg.sh 0 sw_$1_$2 9 | cut -d, -f1,4 | grep -E "bad_kernel|patch_sift"
done
make -s -C cuda-efficient-features_amd/csrc clean >/dev/null 2>&1; make -s -C cuda-efficient-features_amd/csrc 2>&1 | grep error
