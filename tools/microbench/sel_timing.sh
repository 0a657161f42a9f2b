#!/bin/bash
# INVESTIGATION (GPU box): rebuild detect_kernels with -DEFX_SEL_TIMING and print select_kernel's phase times of level 0
cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc"
rm -f detect_kernels.o && make -s EXTRA=-DEFX_SEL_TIMING 2>&1 | grep -E "error" | head
cd "$GRAFT_REPO_ROOT"
python tools/microbench/select_parts.py 2>&1 | grep -E "select l0|^[0-9]" | tail -12
