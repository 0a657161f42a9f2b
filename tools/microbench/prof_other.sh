#!/bin/bash
# rocprofv3 kernel stats for the HashSIFT (configs[3]) and matcher paths -> gpurun_out/*.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_hs -o hs -- python tools/microbench/hs_stage.py > gpurun_out/prof_hs.log 2>&1
python tools/prof_summary.py gpurun_out/prof_hs/hs_results.db gpurun_out/r01_hashsift_kernel_stats.csv | head -6 | cut -c1-120
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_m -o m -- python tools/microbench/matcher_bench.py > gpurun_out/prof_m.log 2>&1
python tools/prof_summary.py gpurun_out/prof_m/m_results.db gpurun_out/r01_matcher_kernel_stats.csv | head -5 | cut -c1-120
