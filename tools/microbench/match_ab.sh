#!/bin/bash
# usage (GPU box, repo root): tools/microbench/match_ab.sh "<EXTRA flags of variant B>" (e.g. -DEFX_MATCH_NO_PRELOAD=1)  -- the matcher with and without a build flag, alternating, then full-size parity of B
cd "$GRAFT_REPO_ROOT"
build() { rm -f cuda-efficient-features_amd/csrc/match_kernels.o; make -s -C cuda-efficient-features_amd/csrc EXTRA="$1" 2>&1 | grep -E "error" | head -3; }
for rep in 1 2; do
  build ""; echo "== A (default)"; python tools/microbench/match_bench.py 2>/dev/null | grep bit
  build "$1"; echo "== B ($1)"; python tools/microbench/match_bench.py 2>/dev/null | grep bit
done
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -x -k "matcher" 2>&1 | tail -1
build ""
