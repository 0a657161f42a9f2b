#!/bin/bash
# INVESTIGATION (GPU box): single-stream detector kernel times for the shipped build and for builds with extra -D flags
# usage: tools/microbench/det_ab.sh "" "-DEFX_NO_RANGE_CHECKS" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for fl in "$@"; do
  if [ -n "$fl" ]; then (cd cuda-efficient-features_amd/csrc && rm -f detect_kernels.o bad_kernel.o && make -s EXTRA="$fl" 2>&1 | grep -E "error" | head -3); fi
  rocprofv3 --kernel-trace --stats -d $O/prof_dab -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > $O/bench_dab.log 2>&1
  echo "== flags: '$fl'"; python tools/prof_summary.py $O/prof_dab/bench_results.db $O/dab.csv | sed -n 2,9p | cut -c1-60; rm -rf $O/prof_dab
done
