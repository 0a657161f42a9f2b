#!/bin/bash
# INVESTIGATION (GPU box): the driver's --steps 20 line by the number of warm-up steps (is the gap to the sustained figure a ramp?)
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do
for w in 5 50 400; do
  echo -n "warmup=$w  "
  python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-configs --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_frame'], d['ms_per_step_min_median_max'])"
done
done
