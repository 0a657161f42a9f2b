#!/bin/bash
# one C3 iteration (GPU box): parity of the computeAsync kernels, then kernel stats of BAD512 / BAD256
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wave_per_keypoint or compute_async or compute_bad" 2>&1 | tail -3
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "c3" 2>&1 | tail -2
for nb in 512 256; do
  rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python tools/microbench/c3_run.py --nbits $nb > $O/c3_raw_$nb.log 2>&1
  python tools/prof_summary.py $O/prof_c3/c3_results.db $O/c3_bad${nb}.csv | sed -n 2,3p | cut -c1-110; rm -rf $O/prof_c3
done
python tools/microbench/c3_run.py --nbits 512 --iters 100; python tools/microbench/c3_run.py --nbits 256 --iters 100
