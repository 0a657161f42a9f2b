#!/bin/bash
# usage (GPU box, repo root): tools/microbench/match_dbg.sh  -- what knn2_dma_kernel's time is made of: stages taken out one at a time
# (1 epilogue, 2 MFMAs, 4 LDS fragment reads, 8 DMA fetches, 16 barrier; the results of these builds are wrong by construction)
cd "$GRAFT_REPO_ROOT"
build() { rm -f cuda-efficient-features_amd/csrc/match_kernels.o; make -s -C cuda-efficient-features_amd/csrc EXTRA="$1" 2>&1 | grep -E "error" | head -3; }
for d in 0 1 2 4 8 16 3 7 15 31 24 28; do
  build "-DKNN_DMA_DBG=$d"; echo "== DBG $d: $(python tools/microbench/match_bench.py 2>/dev/null | grep '512 bit' | cut -c1-40)"
done
build ""
