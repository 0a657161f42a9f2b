#!/bin/bash
# INVESTIGATION (GPU box): rebuild detect_kernels with -DEFX_SEL_TIMING and print select_kernel's phase times of level 0
# (leader and the counting workgroup that finishes the level last) for a few single 8K frames
cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc"
rm -f detect_kernels.o && make -s EXTRA=-DEFX_SEL_TIMING 2>&1 | grep -E "error" | head
cd "$GRAFT_REPO_ROOT"
python tools/microbench/call_latency.py ${1:-8k} 4 2>&1 | grep -E "select|ms" | tail -14
