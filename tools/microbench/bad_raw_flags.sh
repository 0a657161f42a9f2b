#!/bin/bash
# usage (GPU box, repo root): tools/microbench/bad_raw_flags.sh "<flags 1>" "<flags 2>" ...  -- bad_raw_kernel's average launch (config C3, BAD512 and BAD256) under build flags
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for f in "$@"; do
  (cd cuda-efficient-features_amd/csrc && rm -f bad_kernel.o && make -s EXTRA="$f" 2>&1 | grep -E " error" | head -3)
  line="== [$f]"
  for nb in 512 256; do
    rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python tools/microbench/c3_run.py --nbits $nb > $O/c3_dbg.log 2>&1
    line="$line $(python tools/prof_summary.py $O/prof_c3/c3_results.db $O/c3_dbg.csv | grep bad_raw | cut -d, -f1,4)"; rm -rf $O/prof_c3
  done
  echo "$line"
done
(cd cuda-efficient-features_amd/csrc && rm -f bad_kernel.o && make -s 2>&1 | grep -E " error" | head -3)
