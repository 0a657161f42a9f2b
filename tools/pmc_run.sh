#!/bin/bash
# usage (GPU box): tools/pmc_run.sh <tag> <counter> [<counter> ...]   -- one rocprofv3 --pmc pass over a short bench run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/pmc_$tag -o pmc -- python bench.py --steps 1 --warmup 1 --frames-per-step 2 --no-cpu-baseline --no-configs --sustain-seconds 0 > gpurun_out/pmc_$tag.log 2>&1
tail -2 gpurun_out/pmc_$tag.log | cut -c1-200
ls gpurun_out/pmc_$tag
