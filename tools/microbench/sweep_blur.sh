#!/bin/bash
# usage (GPU box): tools/microbench/sweep_blur.sh "RO CR" ["RO CR" ...]  -- rebuilds the two describers with the blur item
# shapes -DEFX_BLUR_RO / -DEFX_BLUR_CR and prints the descriptor kernels' single-stream times (headline workload + HashSIFT 8K)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rc in "$@"; do
  set -- $rc; ro=$1; cr=$2
  rm -f cuda-efficient-features_amd/csrc/bad_kernel.o cuda-efficient-features_amd/csrc/hashsift_kernels.o
  make -s -C cuda-efficient-features_amd/csrc EXTRA="-DEFX_BLUR_RO=$ro -DEFX_BLUR_CR=$cr" 2>&1 | grep -E "error" | head -3
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sw -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/bench_sw.log 2>&1
  echo "== RO=$ro CR=$cr  parity $(grep -o '"parity_8k_frame0": [a-z]*' gpurun_out/bench_sw.log | tail -1)"
  python tools/prof_summary.py gpurun_out/prof_sw/bench_results.db gpurun_out/sw.csv | grep bad_det | cut -d, -f1,4
  rm -rf gpurun_out/prof_sw
done
