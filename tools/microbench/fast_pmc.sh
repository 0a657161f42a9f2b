#!/bin/bash
# INVESTIGATION (GPU box): where fast_kernel's wave-cycles go (SQ counters, one --pmc pass each over a short bench run)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "SQ_IFETCH SQ_INSTS_BRANCH" "SQ_INSTS_SENDMSG SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  rm -rf gpurun_out/pmc_f
  timeout 100 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc_f -o pmc -- python bench.py --steps 1 --warmup 1 --frames-per-step 2 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/pmc_f.log 2>&1 < /dev/null
  python tools/pmc_summary.py gpurun_out/pmc_f/pmc_results.db 2>/dev/null | awk '/^fast_kernel/{f=1;next} /^[^ ]/{f=0} f{print}'
done
rm -rf gpurun_out/pmc_f
