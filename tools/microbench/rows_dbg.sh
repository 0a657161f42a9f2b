#!/bin/bash
# usage (GPU box, repo root): tools/microbench/rows_dbg.sh "<RW_DBG values>"  -- investigation builds of resize_rows_kernel (results invalid), kernel times
# RW_DBG bits: 1 no stores, 2 no arithmetic (the first version also had 4 no loads in the loop, 8 no conversion of source rows, 16 no second level,
# 32 per-wave time stamps: docs/history/round5.md has what they showed)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
    timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sw -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/bench_sw.log 2>&1
    python tools/prof_summary.py gpurun_out/prof_sw/bench_results.db gpurun_out/sw_kernel_stats.csv > /dev/null; rm -rf gpurun_out/prof_sw
    echo "== $1: $(grep resize_rows gpurun_out/sw_kernel_stats.csv | cut -d, -f1-6 | tr '\n' ' ')"
}
for d in $1; do
  rm -f cuda-efficient-features_amd/csrc/detect_kernels.o
  make -s -C cuda-efficient-features_amd/csrc EXTRA="-DRW_DBG=$d" 2>&1 | grep -E "error" | head -3
  run "RW_DBG=$d"
done
rm -f cuda-efficient-features_amd/csrc/detect_kernels.o
