#!/bin/bash
# INVESTIGATION (GPU box): the matrix-core matcher's default choice (chunks from the occupancy of the kernel that runs) against
# fixed chunk counts and tiles per step.  usage: tools/microbench/match_sweep.sh
echo "default"; python tools/microbench/matcher_bench.py 2>&1 | cut -c1-64
for tt in 1 2; do for ch in 3 4 6; do echo "TT=$tt CHUNKS=$ch"; EFX_MATCH_TT=$tt EFX_MATCH_CHUNKS=$ch python tools/microbench/matcher_bench.py 2>&1 | grep "40000x" | cut -c1-64; done; done
