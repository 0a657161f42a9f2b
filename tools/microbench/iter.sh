#!/bin/bash
# usage (GPU box, repo root): tools/microbench/iter.sh <tag> [pytest targets]   -- one optimisation iteration:
# parity tests, single-stream rocprofv3 kernel stats (gpurun_out/<tag>.csv), two default bench runs (three frames in flight)
tag=${1:-x}; shift
targets=${@:-tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_golden.py tests/test_mask_provided.py}
python -m pytest $targets -m gpu -q -x 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/bench_$tag.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$tag/bench_results.db gpurun_out/$tag.csv | head -14 | cut -c1-100
rm -rf gpurun_out/prof_$tag
for i in 1 2; do python bench.py --no-cpu-baseline --no-configs --sustain-seconds 0 --steps 12 --warmup 3 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/frame', d['ms_per_frame'], 'lat', d['latency']['ms_per_frame'])"; done
if [ -n "$PMC" ]; then
  tools/pmc_run.sh $tag SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_$tag/pmc_results.db > gpurun_out/pmc_$tag.txt; rm -rf gpurun_out/pmc_$tag
  grep -A9 "$PMC" gpurun_out/pmc_$tag.txt | head -12
fi
