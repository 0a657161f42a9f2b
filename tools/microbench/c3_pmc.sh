#!/bin/bash
# PMC passes over config C3's kernels (GPU box): tools/microbench/c3_pmc.sh <tag> -> gpurun_out/<tag>_pmc_{sq,lds}.txt
tag=${1:-c3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --kernel-trace -d $O/pmc_c3a -o pmc -- python tools/microbench/c3_run.py --iters 3 > $O/pmc_c3a.log 2>&1
python tools/pmc_summary.py $O/pmc_c3a/pmc_results.db bad_ > $O/${tag}_pmc_sq.txt; rm -rf $O/pmc_c3a
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace -d $O/pmc_c3b -o pmc -- python tools/microbench/c3_run.py --iters 3 > $O/pmc_c3b.log 2>&1
python tools/pmc_summary.py $O/pmc_c3b/pmc_results.db bad_ > $O/${tag}_pmc_lds.txt; rm -rf $O/pmc_c3b
cat $O/${tag}_pmc_sq.txt $O/${tag}_pmc_lds.txt
tail -3 $O/pmc_c3b.log
