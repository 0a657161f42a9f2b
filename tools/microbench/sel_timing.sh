#!/bin/bash
# INVESTIGATION (GPU box): rebuild detect_kernels with -DEFX_SEL_TIMING and print select_kernel's phase times of level 0
# (leader and the counting workgroup that finishes the level last) for a few single 8K frames: sel_timing.sh [8k | nat]
cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc"
rm -f detect_kernels.o && make -s EXTRA=-DEFX_SEL_TIMING 2>&1 | grep -E "error" | head
cd "$GRAFT_REPO_ROOT"
if [ "${1:-8k}" = "nat" ]; then timeout 120 python tools/microbench/natural_prof.py 1.3 4 2>&1 | grep -E "select|keypoints" | tail -6
else timeout 120 python tools/microbench/call_latency.py ${1:-8k} 4 2>&1 | grep -E "select|ms" | tail -6; fi
