#!/bin/bash
# INVESTIGATION (GPU box): matrix-pipe and wait counters of the HashSIFT kernels on the C4 workload
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --kernel-trace -d $O/pmc_hsp -o pmc -- python tools/microbench/hs_stage.py --only 0 > $O/pmc_hsp.log 2>&1
python tools/pmc_summary.py $O/pmc_hsp/pmc_results.db project_sign; rm -rf $O/pmc_hsp
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace -d $O/pmc_hsp -o pmc -- python tools/microbench/hs_stage.py --only 0 > $O/pmc_hsp.log 2>&1
python tools/pmc_summary.py $O/pmc_hsp/pmc_results.db project_sign; rm -rf $O/pmc_hsp
tail -3 $O/pmc_hsp.log
