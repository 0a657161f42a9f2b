#!/bin/bash
# INVESTIGATION (GPU box): patch_sift_kernel's VALU / LDS instructions and time per stage knob (debug build, EFX_DEBUG_HS:
# 5 after blur / window, 1 + warp patch, 6 + gradients and votes without the atomics, 2 + votes, 3 + fold, 4 + normalise, 0 all)
cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc" && rm -f *.o && make -s -j8 EXTRA=-DEFX_DEBUG_BUILD 2>&1 | grep -E " error" | head -3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in 5 1 6 2 3 4 0; do
  rm -rf gpurun_out/pmc_hs
  timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d gpurun_out/pmc_hs -o pmc -- python tools/microbench/hs_stage.py --only $k > gpurun_out/pmc_hs.log 2>&1 < /dev/null
  echo "== EFX_DEBUG_HS=$k: $(python tools/pmc_summary.py gpurun_out/pmc_hs/pmc_results.db patch_sift | grep -E 'SQ_INSTS' | awk '{printf "%s %.2f M  ", $1, $NF/1e6}') $(grep -o 'compute ms.*' gpurun_out/pmc_hs.log | tail -1)"
done
rm -rf gpurun_out/pmc_hs
