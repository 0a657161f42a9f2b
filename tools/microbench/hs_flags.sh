#!/bin/bash
# usage (GPU box, repo root): tools/microbench/hs_flags.sh "<flags 1>" "<flags 2>" ...  -- patch_sift_kernel's average launch (C4 workload) and LDS counters under build flags
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for f in "$@"; do
  (cd cuda-efficient-features_amd/csrc && rm -f hashsift_kernels.o && make -s EXTRA="$f" 2>&1 | grep -E " error" | head -3)
  rocprofv3 --kernel-trace --stats -d $O/prof_hs -o hs -- python tools/microbench/hs_stage.py --only ${HS_STAGE:-0} > $O/prof_hs.log 2>&1
  a="$(python tools/prof_summary.py $O/prof_hs/hs_results.db $O/hs_dbg.csv | grep patch_sift | cut -d, -f1,4)"; rm -rf $O/prof_hs
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace -d $O/pmc_hsp -o pmc -- python tools/microbench/hs_stage.py --only ${HS_STAGE:-0} > $O/pmc_hsp.log 2>&1
  b="$(python tools/pmc_summary.py $O/pmc_hsp/pmc_results.db patch_sift | grep -E 'IDX_ACTIVE|BANK_CONFLICT|INSTS_LDS|INSTS_VALU' | awk '{printf "%s %.1fM  ", $1, $NF/1e6}')"; rm -rf $O/pmc_hsp
  echo "== [$f] $a | $b"
done
(cd cuda-efficient-features_amd/csrc && rm -f hashsift_kernels.o && make -s 2>&1 | grep -E " error" | head -3)
