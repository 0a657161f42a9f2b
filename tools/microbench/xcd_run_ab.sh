#!/bin/bash
# INVESTIGATION (GPU box): the run length of xcd_interleaved (-DEFX_XCD_RUN: consecutive tiles an XCD takes before the next XCD's
# turn) on the tile kernels, the headline, the one-call latencies and the small-frame batches: tools/microbench/xcd_run_ab.sh 32 128 ...
cd "$GRAFT_REPO_ROOT"
D=cuda-efficient-features_amd/csrc
for v in ${@:-32 128 32 128}; do
  rm -f $D/*.o; make -s -C $D EXTRA="-DEFX_XCD_RUN=$v" 2>&1 | grep -E " error" | head -3
  echo "== EFX_XCD_RUN=$v"
  EFX_NO_BATCH=1 timeout 100 tools/microbench/prof_dbg.sh 0 xr 14 | grep -E "fast_kernel|harris|nms_k" | cut -d, -f1,4,5
  timeout 100 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_frame'])"
  for s in fhd 4k 8k; do timeout 60 python tools/microbench/call_latency.py $s 200 2>/dev/null | tail -1; done
  timeout 60 python tools/microbench/batch_throughput.py fhd 16 2 1.5 40000 2>&1 | grep -v amdgpu | tail -1 | cut -c1-110
  timeout 60 python tools/microbench/batch_throughput.py 4k 8 2 1.5 40000 2>&1 | grep -v amdgpu | tail -1 | cut -c1-110
done
