#!/bin/bash
# usage (GPU box): tools/microbench/ab.sh "<ENV=1 ...>" ["<ENV=2 ...>" ...]: single-stream rocprofv3 kernel averages per environment (A/B)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for envs in "$@"; do
  env $envs rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ab -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/bench_ab.log 2>&1
  echo "== $envs"; python tools/prof_summary.py gpurun_out/prof_ab/bench_results.db gpurun_out/ab.csv | sed -n 2,${ABN:-3}p | cut -c1-80
  rm -rf gpurun_out/prof_ab
done
