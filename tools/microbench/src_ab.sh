#!/bin/bash
# INVESTIGATION (GPU box): A/B of two versions of detect_kernels.hip inside one call: the tree's against the file given as $1
# (a copy of an older version placed in the snapshot, e.g. `git show HEAD~1:.../detect_kernels.hip > gpurun_old_detect.hip`)
cd "$GRAFT_REPO_ROOT"
D=cuda-efficient-features_amd/csrc
cp $D/detect_kernels.hip /tmp/new_detect.hip
for v in old new old new; do
  if [ $v = old ]; then cp "$1" $D/detect_kernels.hip; else cp /tmp/new_detect.hip $D/detect_kernels.hip; fi
  rm -f $D/detect_kernels.o; make -s -C $D 2>&1 | grep -E " error" | head -3
  echo "== $v"
  tools/microbench/prof_dbg.sh 0 sw_$v 12 | grep -E "${ABK:-fast_kernel|harris|nms}" | cut -d, -f1,4
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_frame'])"
done
cp /tmp/new_detect.hip $D/detect_kernels.hip
