#!/bin/bash
# usage (GPU box, repo root): tools/microbench/ab_values.sh VAR v1 v2 ...   -- the bench line (value, ms per frame, one-call latency) for
# each value of an environment knob, twice, alternating (same box)
var=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    env $var=$v python bench.py --no-cpu-baseline --no-configs --sustain-seconds 0 --steps ${STEPS:-200} --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', d['value'], d['ms_per_frame'], 'lat', d['latency']['ms_per_frame'])"
  done
done
