#!/bin/bash
# INVESTIGATION (GPU box): fast_kernel's FETCH_SIZE (KiB per launch, uncorrected) and time under build flags:
# tools/microbench/fast_fetch_ab.sh "" "-DEFX_FAST_QUICK16=1" "-DEFX_XCD_RUN=8" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=cuda-efficient-features_amd/csrc
for fl in "$@"; do
  rm -f $D/detect_kernels.o $D/efx_api.o; make -s -C $D EXTRA="$fl" 2>&1 | grep -E " error" | head -3
  echo "== flags: '$fl'"
  rm -rf gpurun_out/pmc_ff
  timeout 100 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_ff -o pmc -- python bench.py --steps 1 --warmup 1 --frames-per-step 2 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/pmc_ff.log 2>&1 < /dev/null
  python tools/pmc_summary.py gpurun_out/pmc_ff/pmc_results.db 2>/dev/null | awk '/^fast_kernel/{f=1;next} /^[^ ]/{f=0} f{print}'
  EFX_NO_BATCH=1 timeout 100 tools/microbench/prof_dbg.sh 0 ff 12 | grep -E "fast_kernel" | cut -d, -f1,2,4,5
done
rm -rf gpurun_out/pmc_ff
