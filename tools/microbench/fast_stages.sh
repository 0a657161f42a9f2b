#!/bin/bash
# INVESTIGATION (GPU box): fast_kernel's time with its phases cut off (debug build, EFX_DEBUG: 1 = tile load only, 2 = no survivors,
# 8 = no 16-point test; results invalid) on the headline frame and on the 1/f^1.3 frame.  Every run under `timeout`: downstream
# kernels of a frame whose FAST stage was cut short may take long.
cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc"
rm -f detect_kernels.o efx_api.o && make -s EXTRA="-DEFX_DEBUG_BUILD" 2>&1 | grep -E " error" | head
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for dbg in ${@:-0 1 2 8}; do
  echo "== EFX_DEBUG=$dbg"
  rm -rf gpurun_out/prof_fs
  EFX_DEBUG=$dbg timeout 90 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fs -o b -- python bench.py --steps 1 --warmup 1 --frames-per-step 2 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/fs.log 2>&1 < /dev/null
  echo -n "headline frame: "; python tools/prof_summary.py gpurun_out/prof_fs/b_results.db gpurun_out/fs.csv | grep -E "fast_kernel" | cut -d, -f1,2,4,5
  rm -rf gpurun_out/prof_fs
  EFX_DEBUG=$dbg timeout 90 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fs -o b -- python tools/microbench/natural_prof.py 1.3 4 > gpurun_out/fs.log 2>&1 < /dev/null
  echo -n "1/f^1.3 frame:  "; python tools/prof_summary.py gpurun_out/prof_fs/b_results.db gpurun_out/fs.csv | grep -E "fast_kernel" | cut -d, -f1,2,4,5
done
rm -rf gpurun_out/prof_fs
