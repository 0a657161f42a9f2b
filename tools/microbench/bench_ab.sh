#!/bin/bash
# INVESTIGATION (GPU box): the bench line (three frames in flight) for the shipped build and for builds with extra -D flags,
# REPS runs each, alternating.  usage: tools/microbench/bench_ab.sh "" "-DEFX_RESIZE_SCALAR" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O/ab
i=0
for fl in "$@"; do
  (cd cuda-efficient-features_amd/csrc && rm -f detect_kernels.o bad_kernel.o && make -s EXTRA="$fl" 2>&1 | grep -E "error" | head -3)
  cp cuda-efficient-features_amd/libefx_hip.so $O/ab/lib$i.so; i=$((i+1))
done
for rep in $(seq 1 ${REPS:-4}); do
  i=0
  for fl in "$@"; do
    cp $O/ab/lib$i.so cuda-efficient-features_amd/libefx_hip.so
    v=$(python bench.py --no-cpu-baseline --no-configs --sustain-seconds 0 --steps ${STEPS:-20} --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_frame'])")
    echo "rep $rep flags '$fl': $v"
    i=$((i+1))
  done
done
rm -rf $O/ab
