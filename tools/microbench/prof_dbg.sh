#!/bin/bash
# usage: tools/microbench/prof_dbg.sh <EFX_DEBUG value> <tag>: rocprofv3 kernel stats of a short bench run with a debug knob set
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
EFX_DEBUG=$1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$2 -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/bench_$2.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$2/bench_results.db gpurun_out/$2.csv | head -${3:-6} | cut -c1-110
