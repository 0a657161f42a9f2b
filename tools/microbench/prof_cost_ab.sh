#!/bin/bash
# INVESTIGATION (GPU box): what the live HIP-event pairs inside the timed region cost the driver's --steps 20 line
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do
for p in 24 0 4; do
  echo -n "prof_steps=$p  "
  EFX_BENCH_PROF_STEPS=$p python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_frame'], d['ms_per_step_min_median_max'])"
done
done
