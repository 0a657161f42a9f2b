#!/bin/bash
# INVESTIGATION (GPU box): VALU wave-instructions per launch of the detector kernels on the headline frame with the stage
# knobs of a debug build (EFX_DEBUG: see the kernels; results are NOT valid with a knob set).
# usage: tools/microbench/stage_insts.sh 0 16 32 48
cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc" && rm -f detect_kernels.o efx_api.o && make -s EXTRA=-DEFX_DEBUG_BUILD 2>&1 | grep -E " error" | head -3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in "$@"; do
  rm -rf gpurun_out/pmc_st
  EFX_DEBUG=$k timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d gpurun_out/pmc_st -o pmc -- python bench.py --steps 1 --warmup 1 --frames-per-step 2 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/pmc_st.log 2>&1 < /dev/null
  echo "== EFX_DEBUG=$k"
  python tools/pmc_summary.py gpurun_out/pmc_st/pmc_results.db | python -c "
import sys
k = None
for l in sys.stdin:
    if not l.startswith(' '): k = l.split(' (')[0].strip(); continue
    f = l.split()
    if k and not k.startswith('at::') and f[0] in ('SQ_INSTS_VALU', 'SQ_INSTS_LDS'): print('   %-28s %-14s %12.0f per launch' % (k[:28], f[0], float(f[-1])))
"
done
rm -rf gpurun_out/pmc_st
