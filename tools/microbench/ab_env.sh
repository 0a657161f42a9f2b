#!/bin/bash
# usage (GPU box, repo root): tools/microbench/ab_env.sh VAR [reps] -- same-box A/B of the bench line with and without an environment knob
# (EFX_NO_LEVEL_BLUR, EFX_NO_TOWER, ...): value, ms per frame, one-call latency; alternating runs
var=$1; reps=${2:-3}
for i in $(seq $reps); do
  for v in 1 0; do
    if [ $v = 1 ]; then export $var=1; else unset $var; fi
    python bench.py --no-cpu-baseline --no-configs --sustain-seconds 0 --steps ${STEPS:-200} --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$var', os.environ.get('$var'), d['value'], d['ms_per_frame'], d['latency']['ms_per_frame'], {k:v['avg_launch_ms'] for k,v in r['kernels_isolated'].items()})"
  done
done
