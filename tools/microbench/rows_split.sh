#!/bin/bash
# usage (GPU box, repo root): tools/microbench/rows_split.sh "<split> <split> ..."   e.g. "2,2,2,1 3,4 2,3,2"
# per EFX_ROWS_SPLIT value (levels made per launch of resize_rows_kernel): the pyramid parity tests, then the kernel times of a short
# one-stream bench run under rocprofv3 and the bench line's value / latency
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for sp in $1; do
  export EFX_ROWS_SPLIT=$sp
  echo "== EFX_ROWS_SPLIT=$sp: $(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k 'pyramid' 2>&1 | tail -1)"
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sw -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/bench_sw.log 2>&1
  python tools/prof_summary.py gpurun_out/prof_sw/bench_results.db gpurun_out/sw_kernel_stats.csv > /dev/null; rm -rf gpurun_out/prof_sw
  grep resize_rows gpurun_out/sw_kernel_stats.csv | cut -d, -f1-6
  python - <<'PY'
import csv
rows=[r for r in csv.reader(open("gpurun_out/sw_kernel_stats.csv")) if r and r[0].startswith("resize_rows")]
frames=53
print("   chain us per frame:", round(sum(float(r[2]) for r in rows)/frames,1), " launches per frame:", sum(int(r[1]) for r in rows)/frames)
PY
  python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-configs --sustain-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value', d['value'], 'ms/frame', d['ms_per_frame'], 'latency', d['latency']['ms_per_frame'], 'parity-free')"
done
