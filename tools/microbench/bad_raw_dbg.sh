#!/bin/bash
# (needs tools/experiments/bad_raw_stage_knobs.patch applied)
# usage (GPU box, repo root): tools/microbench/bad_raw_dbg.sh  -- what bad_raw_kernel<8>'s time is made of (config C3: 40 000 keypoints of
# a 4K frame): stages taken out, -DBAD_RAW_DBG bits 1 window loads, 2 row prefix, 4 column prefix, 8 box phase, 16 the boxes' LDS gathers only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for d in 0 1 2 4 8 16 6 7 15 24; do
  (cd cuda-efficient-features_amd/csrc && rm -f bad_kernel.o && make -s EXTRA="-DBAD_RAW_DBG=$d" 2>&1 | grep -E " error" | head -3)
  rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python tools/microbench/c3_run.py --nbits 512 > $O/c3_dbg.log 2>&1
  echo "== DBG $d: $(python tools/prof_summary.py $O/prof_c3/c3_results.db $O/c3_dbg.csv | grep bad_raw | cut -d, -f1-4)"; rm -rf $O/prof_c3
done
(cd cuda-efficient-features_amd/csrc && rm -f bad_kernel.o && make -s 2>&1 | grep -E " error" | head -3)
