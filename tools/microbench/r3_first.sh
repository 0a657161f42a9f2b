#!/bin/bash
# round 3, first GPU call: parity of the new code + C3 profiles (wave-per-keypoint kernel vs the generic one)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_arena.py tests/test_reference_table_pins.py tests/test_gpu_fullsize.py tests/test_golden.py tests/test_mask_provided.py -m gpu -q -x -s 2>&1 | tail -15 > $O/r3_first_tests.txt
cat $O/r3_first_tests.txt
for nb in 512 256; do
  rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python tools/microbench/c3_run.py --nbits $nb > $O/c3_raw_$nb.log 2>&1
  tail -1 $O/c3_raw_$nb.log
  python tools/prof_summary.py $O/prof_c3/c3_results.db $O/r03_c3_bad${nb}_kernel_stats.csv | head -8 | cut -c1-110; rm -rf $O/prof_c3
done
EFX_BAD_NO_RAW=1 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python tools/microbench/c3_run.py --nbits 512 > $O/c3_generic_512.log 2>&1
tail -1 $O/c3_generic_512.log
python tools/prof_summary.py $O/prof_c3/c3_results.db $O/r03_c3_bad512_generic_kernel_stats.csv | head -8 | cut -c1-110; rm -rf $O/prof_c3
python bench.py --no-cpu-baseline --no-configs --sustain-seconds 0 --steps 12 --warmup 3 | tail -1 | cut -c1-400
