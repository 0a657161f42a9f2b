#!/bin/bash
# usage (GPU box, repo root): tools/microbench/tower_vs_rows.sh  -- FHD / 4K detect and detectAndCompute (reference protocol) with the
# tower launch and with the row-walking chain (EFX_NO_TOWER=1), plus the pyramid kernels' times under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in tower rows; do
  if [ $mode = rows ]; then export EFX_NO_TOWER=1; else unset EFX_NO_TOWER; fi
  echo "== $mode"
  python tools/bench_configs.py --sizes fhd,4k --no-cpu-baseline --iters 40 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['rows']:
    if r['config']=='readme':
        if r['mode']=='detect': print('  detect', r['size'], r['ms'], r['ms_min'])
        else: print('  dAC', r['size'], r['descriptor'], r['detectAndCompute']['ms'], r['detectAndCompute']['ms_min'])
    elif r['config']=='C2': print('  C2 4K detect', r['ms'], r['ms_min'])
"
done
