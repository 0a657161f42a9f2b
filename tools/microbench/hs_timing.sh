#!/bin/bash
# INVESTIGATION (GPU box): project_sign_kernel's phase times (-DHS_PROJ_TIMING build), workgroup 0
cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc"
rm -f hashsift_kernels.o && make -s EXTRA=-DHS_PROJ_TIMING 2>&1 | grep -E "error" | head
cd "$GRAFT_REPO_ROOT"
python tools/microbench/hs_stage.py --only 0 2>&1 | grep -E "proj wg0|compute ms" | head -12
