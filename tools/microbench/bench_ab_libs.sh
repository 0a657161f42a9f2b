#!/bin/bash
# INVESTIGATION (GPU box): the bench line for the shipped library against the one built from another git revision's csrc
# (pre-built in the container as gpurun_out/ab/base.so):  tools/microbench/bench_ab_libs.sh   (REPS, STEPS)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp cuda-efficient-features_amd/libefx_hip.so /tmp/new.so
for rep in $(seq 1 ${REPS:-4}); do
  for which in new base; do
    if [ $which = new ]; then cp /tmp/new.so cuda-efficient-features_amd/libefx_hip.so; else cp tools/microbench/ab_base.so cuda-efficient-features_amd/libefx_hip.so; fi
    v=$(python bench.py --no-cpu-baseline --no-configs --sustain-seconds 0 --steps ${STEPS:-20} --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_frame'], d['latency']['ms_per_frame'])")
    echo "rep $rep $which: $v"
  done
done
cp /tmp/new.so cuda-efficient-features_amd/libefx_hip.so
