#!/bin/bash
# one headline iteration (GPU box): BAD / detectAndCompute parity, single-stream kernel stats, two bench runs
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_mask_provided.py -m gpu -q -x -k "bad or compute or golden or provided or extremes or batched or determinism" 2>&1 | tail -3
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "c3 or c5_8k_detect_and_compute_bad512" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > $O/bench_$tag.log 2>&1
python tools/prof_summary.py $O/prof_$tag/bench_results.db $O/$tag.csv | head -11 | cut -c1-100
rm -rf $O/prof_$tag
for i in 1 2; do python bench.py --no-cpu-baseline --no-configs --sustain-seconds 0 --steps 12 --warmup 3 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/frame', d['ms_per_frame'], 'lat', d['latency']['ms_per_frame'])"; done
