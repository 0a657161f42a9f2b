#!/bin/bash
# INVESTIGATION (GPU box): fast_kernel by tiles per workgroup (EFX_FAST_TPW) on the headline frame and on the 1/f^1.3 frame
cd "$GRAFT_REPO_ROOT"
for t in ${@:-1 2 4 8 1 4}; do
  echo "== EFX_FAST_TPW=$t"
  EFX_FAST_TPW=$t tools/microbench/prof_dbg.sh 0 tpw_$t 12 | grep -E "fast_kernel" | cut -d, -f1,2,4,5
  EFX_FAST_TPW=$t tools/microbench/natural_prof.sh 1.3 2>&1 | grep -E "fast_k"
  EFX_FAST_TPW=$t python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_frame'])"
done
