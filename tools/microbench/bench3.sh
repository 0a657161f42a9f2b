#!/bin/bash
# usage (GPU box): tools/microbench/bench3.sh  -- three short bench.py runs, prints Mkeypoints/s of each (A/B comparisons)
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-configs --sustain-seconds 0 --steps 12 --warmup 3 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_frame'])"; done
