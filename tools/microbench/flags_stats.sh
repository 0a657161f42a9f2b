#!/bin/bash
# usage (GPU box, repo root): tools/microbench/flags_stats.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...  -- rebuilds the detector / BAD translation units with compiler flags and prints the one-stream kernel times of the headline frame (results of flags that change float semantics are NOT checked here)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for f in "$@"; do
  (cd cuda-efficient-features_amd/csrc && rm -f detect_kernels.o bad_kernel.o && make -s EXTRA="$f" 2>&1 | grep -E " error|unknown" | head -3)
  rocprofv3 --kernel-trace --stats -d $O/prof_fl -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > $O/bench_fl.log 2>&1
  python tools/prof_summary.py $O/prof_fl/bench_results.db $O/fl_kernel_stats.csv > /dev/null; rm -rf $O/prof_fl
  echo "== [$f] $(awk -F, 'NR>1 && NR<12 {gsub(/_kernel.*/,"",$1); printf "%s %.1f  ", $1, $4}' $O/fl_kernel_stats.csv)"
done
(cd cuda-efficient-features_amd/csrc && rm -f detect_kernels.o bad_kernel.o && make -s 2>&1 | grep -E " error" | head -3)
