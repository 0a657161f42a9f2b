#!/bin/bash
# usage (GPU box, repo root): tools/microbench/rows_sweep.sh "<RW_D values>" "<EFX_ROWS_WAVES values>" ["<extra -D flags>"]
# rebuilds the library with -DRW_D=<d> (kernel and plan), runs a short one-stream bench under rocprofv3 per EFX_ROWS_WAVES value
# and prints the resize_rows_kernel rows of the kernel stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in $1; do
  rm -f cuda-efficient-features_amd/csrc/detect_kernels.o cuda-efficient-features_amd/csrc/efx_api.o
  make -s -C cuda-efficient-features_amd/csrc EXTRA="-DRW_D=$d $3" 2>&1 | grep -E "error" | head -3
  for w in $2; do
    export EFX_ROWS_WAVES=$w
    timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sw -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/bench_sw.log 2>&1
    python tools/prof_summary.py gpurun_out/prof_sw/bench_results.db gpurun_out/sw_kernel_stats.csv > /dev/null; rm -rf gpurun_out/prof_sw
    echo "== RW_D=$d $3 EFX_ROWS_WAVES=$w: $(grep resize_rows gpurun_out/sw_kernel_stats.csv | cut -d, -f1-6 | tr '\n' ' ')"
  done
done
rm -f cuda-efficient-features_amd/csrc/detect_kernels.o cuda-efficient-features_amd/csrc/efx_api.o
