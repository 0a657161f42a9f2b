#!/bin/bash
# usage (GPU box, repo root): tools/microbench/rows_iter.sh <tag>   -- pyramid parity tests + one-stream kernel stats of a short bench run
tag=${1:-x}
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "pyramid" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/bench_$tag.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$tag/bench_results.db gpurun_out/${tag}_kernel_stats.csv > /dev/null; rm -rf gpurun_out/prof_$tag
head -14 gpurun_out/${tag}_kernel_stats.csv | cut -c1-110
tail -1 gpurun_out/bench_$tag.log | cut -c1-300
