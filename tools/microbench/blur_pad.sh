#!/bin/bash
# usage (GPU box, repo root): tools/microbench/blur_pad.sh "<flags 1>" ...  -- the window-blur users (8K detectAndCompute HashSIFT512: patch_sift_kernel<true, 48>) under build flags: kernel time + LDS counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for f in "$@"; do
  (cd cuda-efficient-features_amd/csrc && rm -f hashsift_kernels.o bad_kernel.o efx_api.o && make -s EXTRA="$f" 2>&1 | grep -E " error" | head -3)
  rocprofv3 --kernel-trace --stats -d $O/prof_bp -o bp -- python tools/microbench/dac_run.py 8k HASH_SIFT_512 12 > $O/prof_bp.log 2>&1
  a="$(python tools/prof_summary.py $O/prof_bp/bp_results.db $O/bp.csv | grep patch_sift | cut -d'"' -f3 | cut -d, -f4)"; rm -rf $O/prof_bp
  b="$(tools/microbench/lds_conflicts.sh python tools/microbench/dac_run.py 8k HASH_SIFT_512 3 | grep patch_sift | cut -c45-110)"
  echo "== [$f] patch_sift<true,48> avg $a us | $b"
done
(cd cuda-efficient-features_amd/csrc && rm -f hashsift_kernels.o bad_kernel.o efx_api.o && make -s 2>&1 | grep -E " error" | head -3)
