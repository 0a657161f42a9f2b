#!/bin/bash
# usage (GPU box, repo root): tools/microbench/blur_sweep.sh "<-DBLV_PF=3 -DBLV_WAVES=6>" ...   -- rebuilds bad_kernel.hip with the given
# defines and prints blur_levels_kernel's single-stream average (rocprofv3) and two bench lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for defs in "$@"; do
  rm -f cuda-efficient-features_amd/csrc/bad_kernel.o
  make -s -C cuda-efficient-features_amd/csrc EXTRA="$defs" 2>&1 | grep -E "error" | head -3
  echo "== $defs"
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bs -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/bench_bs.log 2>&1
  python tools/prof_summary.py gpurun_out/prof_bs/bench_results.db gpurun_out/bs.csv | grep -E "blur_levels|bad_raw" | cut -d, -f1,4,8
  rm -rf gpurun_out/prof_bs
  python bench.py --no-cpu-baseline --no-configs --sustain-seconds 0 --steps 100 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'lat', d['latency']['ms_per_frame'], 'parity', d.get('parity_8k_frame0'))"
done
rm -f cuda-efficient-features_amd/csrc/bad_kernel.o; make -s -C cuda-efficient-features_amd/csrc 2>&1 | grep error
