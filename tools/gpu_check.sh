#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_check.sh <tag> [pytest-args]
# runs the GPU parity tests, then a rocprofv3 kernel trace of a short bench run into gpurun_out/prof_<tag>
tag=${1:-x}
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/bench_$tag.log 2>&1
tail -1 gpurun_out/bench_$tag.log | cut -c1-260
