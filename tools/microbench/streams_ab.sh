#!/bin/bash
# INVESTIGATION (GPU box): headline throughput by the number of contexts / streams and frames per step
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
for cfg in "3 8" "2 8" "4 8" "3 12" "4 12" "2 12" "3 9"; do
  set -- $cfg
  echo -n "streams=$1 frames_per_step=$2  "
  python bench.py --steps 150 --warmup 10 --streams $1 --frames-per-step $2 --no-cpu-baseline --no-configs --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_frame'], d['config'].get('frames_per_launch'))"
done
done
