#!/bin/bash
# usage (GPU box): tools/microbench/sweep_define.sh NAME v1 v2 ...  -- rebuilds the library with -DNAME=v and prints the kernel times
name=$1; shift
for v in "$@"; do
  make -s -C cuda-efficient-features_amd/csrc clean >/dev/null 2>&1
  make -s -C cuda-efficient-features_amd/csrc EXTRA="-D$name=$v" 2>&1 | grep -E "error" | head -3
  echo "== $name=$v"
  tools/microbench/prof_dbg.sh 0 sw_$v 9 | tail -8 | cut -d, -f1,4
done
