#!/bin/bash
# usage (GPU box, repo root): tools/microbench/match_flags.sh "<env assignments>" "<flags 1>" "<flags 2>" ...  -- the matcher bench under each set of build flags (no parity run: investigation flags may break results)
cd "$GRAFT_REPO_ROOT"
envs=$1; shift
build() { rm -f cuda-efficient-features_amd/csrc/match_kernels.o; make -s -C cuda-efficient-features_amd/csrc EXTRA="$1" 2>&1 | grep -E "error" | head -3; }
for f in "$@"; do build "$f"; echo "== [$f] $(env $envs python tools/microbench/match_bench.py 2>/dev/null | grep bit | cut -c1-24 | tr '\n' ' ')"; done
build ""
